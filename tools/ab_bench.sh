#!/bin/bash
# A/B of library builds on the headline IN THE BENCH CONTEXT (GPU box): alternating `bench.py --no-extras` processes,
# one pipelined stream object per process (several pipelined objects in one process share hardware queues and stop
# overlapping: profiles/r03_notes.md).   usage: tools/ab_bench.sh <rounds> <lib> [<lib> ...]   lib = "-" (the tree's
# build) or a variant name under yagi_amd/variants/
rounds=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" != "-" ]; then export YAGI_HIP_LIB=$root/yagi_amd/variants/libyagi_$lib.so; else unset YAGI_HIP_LIB; fi
    python3 $root/bench.py --no-extras --no-cpu-baseline $AB_BENCH_ARGS 2>/dev/null | python3 -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib', o['value'], o['roofline']['frac'], o['roofline']['kernel_ms'])"
  done
done
