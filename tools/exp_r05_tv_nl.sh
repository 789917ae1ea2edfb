export SPCIES_TVR_RTC=1
for nl in 13 10 8 6; do echo "NL=$nl"; SPCIES_TVR_RTC_FLAGS="-DSPCIES_TVR_NL=$nl" python3 tools/bench_tv.py 65536 auto 0 C2_lax; SPCIES_TVR_RTC_FLAGS="-DSPCIES_TVR_NL=$nl" python3 tools/bench_tv.py 65536 auto 0 C2_lax_FISTA; done
