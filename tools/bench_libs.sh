#!/bin/bash
# timing of diagnostic library builds: tools/bench_libs.sh lib1.so lib2.so ...
for lib in "$@"; do
  SPCIES_HIP_LIB=$PWD/spcies_amd/$lib timeout 300 python bench.py --variant ${VARIANT:-mfma} --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), 'solves/s  kernel_ms', round(d['roofline']['kernel_ms'],3), d['config']['all_k_200_eflag_-1'])"
done
