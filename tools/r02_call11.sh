#!/bin/bash
set -u
O=gpurun_out/r02k
mkdir -p $O
export TMPDIR=/tmp
( BBDM_BF3_K32=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -s -m gpu -k "bf3" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -h "gemm_bf3\|passed\|failed\|rc=\|FAILED" $O/tests.log | tail -6
for k in 0 1; do echo "== K32=$k"; BBDM_BF3_K32=$k timeout 200 python tools/gemm_bench.py --bf3 1 2>&1 | grep -v amdgpu.ids | tail -9
( BBDM_BF3_K32=$k timeout 400 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu > $O/bench_c2_k$k.json 2> $O/bench_c2_k$k.err ); python - <<PY
import json
d=json.load(open('$O/bench_c2_k$k.json')); print('c2 k32=$k', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})
PY
done
