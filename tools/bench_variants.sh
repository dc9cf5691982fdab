#!/bin/bash
# usage: tools_bench_variants.sh lib1.so lib2.so ...   (run on the GPU box; prints value + kernel_ms per build)
for lib in "$@"; do
  for i in 1 2; do
    YAGI_HIP_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['roofline']['kernel_ms'], d['roofline']['fp32']['achieved_tflops'], d['parity_rel_l2_vs_f64'])"
  done
done
