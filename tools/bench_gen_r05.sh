for c in C2_lax_N30_gen C4_lax_ADMM_gen; do for v in mfma4r mfma4g; do python3 tools/bench_one.py $c $v 65536 4; done; done
python3 tools/bench_one.py C2_lax_N30 mfma4r 65536 4
