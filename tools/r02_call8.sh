#!/bin/bash
set -u
O=gpurun_out/r02h
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wino or bf3" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -h "passed\|failed\|rc=\|FAILED" $O/tests.log | tail -5
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu --dump-ops $O/c2_ops.md > $O/bench_c2.json 2> $O/bench_c2.err ); python - <<PY
import json
d=json.load(open('$O/bench_c2.json')); print('c2', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, round(d['roofline']['frac'],3), round(d['roofline']['frac_step'],3))
PY
grep "winograd_input\|winograd_output" $O/c2_ops.md | sort -t'|' -k5 -n -r | head -8
