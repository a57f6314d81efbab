#!/bin/bash
set -u
O=gpurun_out/r02j
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -s -m gpu -k "bf3" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -h "gemm_bf3\|passed\|failed\|rc=\|FAILED" $O/tests.log | tail -6
timeout 200 python tools/gemm_bench.py --bf3 1 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench.log | tail -9
( timeout 400 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu > $O/bench_c2.json 2> $O/bench_c2.err ); python - <<PY
import json
d=json.load(open('$O/bench_c2.json')); print('c2', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})
PY
