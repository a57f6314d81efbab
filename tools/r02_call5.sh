#!/bin/bash
set -u
O=gpurun_out/r02e
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -s -m gpu -k "bf3 or wino" > $O/bf3_tests.log 2>&1; echo "rc=$?" >> $O/bf3_tests.log )
grep -h "gemm_bf3\|passed\|failed\|rc=\|FAILED\|Error" $O/bf3_tests.log | tail -12
for b in 0 1; do echo "== gemm_bench bf3=$b"; timeout 200 python tools/gemm_bench.py --bf3 $b 2>&1 | grep -v amdgpu.ids; done | tee $O/gemm_bench.log
for b in 0 1; do ( BBDM_GEMM_BF3=$b timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_c2_bf3_$b.json 2> $O/bench_c2_bf3_$b.err ); python - <<PY
import json
try:
    d=json.load(open('$O/bench_c2_bf3_$b.json')); print('bf3=$b', round(d['ms_per_step'],2),'ms/step', d['parity'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})
except Exception as e: print('bf3=$b failed', e)
PY
done
tail -3 $O/bench_c2_bf3_1.err
