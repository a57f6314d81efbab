#!/bin/bash
set -u
O=gpurun_out/r02l
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_loop_parity_gpu.py tests/test_training_gpu.py -q -s -m gpu -k "statistics or wino or golden or full_size or loop or gradients" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -h "passed\|failed\|rc=\|FAILED\|Error\|rel err\|drift" $O/tests.log | tail -14
for f in 1 0; do for w in c2 c3; do ( BBDM_FUSE_STATS=$f timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu > $O/bench_${w}_fs$f.json 2> $O/bench_${w}_fs$f.err ); python - <<PY
import json
d=json.load(open('$O/bench_${w}_fs$f.json')); print('$w fuse_stats=$f', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items() if 'stats' in k or 'output' in k or 'conv2d' in k})
PY
done; done
