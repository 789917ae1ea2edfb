#!/bin/bash
# Sanitizer recipe for the HOST side of libspcies_hip.so (CPU builds only: this pool has no GPU sanitizer, no XNACK).
#
#   tools/sanitize.sh asan    AddressSanitizer + UndefinedBehaviorSanitizer: the CPU test hooks under pytest (blob fuzz, C-ABI symbol /
#                             error-path tests, the on-disk code-object cache tests) and the thread stress (tools/sanitize_stress.cpp)
#   tools/sanitize.sh tsan    ThreadSanitizer: the thread stress (8 threads: code-object cache with eviction + pruning + in-flight sharing,
#                             parser / packers behind create and create_multi, last-error strings)
#   tools/sanitize.sh oracle  the C oracle under ASan + UBSan against its golden tests
#   tools/sanitize.sh all     the three, log in profiles/r05_sanitizers.log
#
# Device code is compiled as usual (-Xarch_host restricts the instrumentation to the host half); each variant builds into its own
# object directory next to the product (spcies_amd/csrc/build_<mode>, spcies_amd/libspcies_hip_<mode>.so: git-ignored).
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
MODE=${1:-all}
LOG=${SANITIZE_LOG:-$R/gpurun_out/sanitize_$MODE.log}
mkdir -p "$(dirname "$LOG")"
CLANG=/opt/rocm/lib/llvm/bin/clang++
JOBS=${JOBS:-6}

blobs() {  # a few controllers' blobs as files for the stress driver
    python3 - "$1" <<'PY'
import os, sys
sys.path.insert(0, os.environ["SPCIES_ROOT"])
from spcies_amd import benchmarks, blob
for name in ("C1", "C1_lax_gen", "C1_MPCT", "C1_soc", "C1_HMPC_SADMM", "C1_MPCT_cs", "C1_lax_FISTA"):
    open(os.path.join(sys.argv[1], name + ".blob"), "wb").write(blob.pack(benchmarks.ingredients(benchmarks.config(name))))
PY
}

build_lib() {  # $1 = mode name, $2 = sanitizer list
    make -j$JOBS -C $R/spcies_amd/csrc BUILD=build_$1 OUT=../libspcies_hip_$1.so EXTRA="-Xarch_host -fsanitize=$2 -Xarch_host -fno-omit-frame-pointer -Xarch_host -g1" \
        > $R/gpurun_out/sanitize_build_$1.log 2>&1 || { echo "build of the $1 library failed: gpurun_out/sanitize_build_$1.log"; tail -5 $R/gpurun_out/sanitize_build_$1.log; return 1; }
    # (hipcc links the shared object without the sanitizer runtime: it comes from the executable - the stress driver - or LD_PRELOAD under python)
}

run_stress() {  # $1 = mode, $2 = sanitizer list
    local T=$(mktemp -d /tmp/spcies_san.XXXXXX)
    mkdir -p $T/cache $T/blobs
    SPCIES_ROOT=$R blobs $T/blobs || return 1
    $CLANG -O1 -g -std=c++17 -fsanitize=$2 -fno-omit-frame-pointer -o $T/stress $R/tools/sanitize_stress.cpp -ldl -lpthread || return 1
    $T/stress $R/spcies_amd/libspcies_hip_$1.so $T/cache ${STRESS_THREADS:-8} ${STRESS_ROUNDS:-150} $T/blobs/*.blob
    local rc=$?
    rm -rf $T
    return $rc
}

asan() {
    echo "== asan + ubsan: host side of libspcies_hip.so"
    build_lib asan address,undefined || return 1
    local RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
    ( cd $R && LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
        SPCIES_HIP_LIB=$R/spcies_amd/libspcies_hip_asan.so python3 -m pytest tests/test_blob_fuzz.py tests/test_cabi_symbols.py tests/test_rtc_disk_cache.py \
        tests/test_no_library_gemm.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15 ) || return 1
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 run_stress asan address,undefined
}

tsan() {
    echo "== tsan: host side of libspcies_hip.so, thread stress"
    build_lib tsan thread || return 1
    TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 run_stress tsan thread
}

oracle() {
    echo "== asan + ubsan: oracle/*.c against its golden tests"
    local T=$(mktemp -d /tmp/spcies_san.XXXXXX)
    gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fopenmp -shared -fPIC -o $T/liboracle.so $R/oracle/*.c -lm || return 1
    ( cd $R && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 SPCIES_ORACLE_LIB=$T/liboracle.so \
        python3 -m pytest tests/test_oracle_golden.py -q -x -p no:cacheprovider 2>&1 | tail -5 )
    local rc=$?
    rm -rf $T
    return $rc
}

rc=0
{
    echo "# tools/sanitize.sh $MODE - $(date -u +%Y-%m-%dT%H:%MZ) - $(git -C $R rev-parse --short HEAD) - $($CLANG --version | head -1)"
    case $MODE in
        asan) asan || rc=1 ;;
        tsan) tsan || rc=1 ;;
        oracle) oracle || rc=1 ;;
        all) asan || rc=1; tsan || rc=1; oracle || rc=1 ;;
        *) echo "usage: $0 asan|tsan|oracle|all"; rc=2 ;;
    esac
    echo "# exit code $rc"
} 2>&1 | tee $LOG
exit $(tail -1 $LOG | awk '{print $NF}')
