#!/bin/bash
set -u
O=gpurun_out/r02g
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -s -m gpu -k "bf3 or golden or full_size" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -h "passed\|failed\|rc=\|FAILED\|Error\|rel err" $O/tests.log | tail -12
for d in 1 2; do echo "== BBDM_BF3_DEPTH=$d"; BBDM_BF3_DEPTH=$d timeout 200 python tools/gemm_bench.py --bf3 1 2>&1 | grep -v amdgpu.ids | tail -4
( BBDM_BF3_DEPTH=$d timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu --dump-ops $O/c2_ops_d$d.md > $O/bench_c2_d$d.json 2> $O/bench_c2_d$d.err ); python - <<PY
import json
d=json.load(open('$O/bench_c2_d$d.json')); print('c2 depth $d', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, round(d['roofline']['frac'],3), round(d['roofline']['frac_step'],3))
PY
done
