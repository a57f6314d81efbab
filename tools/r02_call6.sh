#!/bin/bash
set -u
O=gpurun_out/r02f
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_first_stage_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_latent_gpu.py -q -s -m gpu -k "first_stage or vq or golden or xattn or cross_attention or layernorm or conv2d or fused or latent or encode" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -h "VQ-f4\|passed\|failed\|rc=\|FAILED\|Error\|rel err" $O/tests.log | tail -25
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu --dump-ops $O/c2_ops.md > $O/bench_c2.json 2> $O/bench_c2.err ); python - <<PY
import json
d=json.load(open('$O/bench_c2.json')); print('c2', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, d['roofline']['frac'], d['roofline']['frac_step'])
PY
for w in c1 c3 c5; do ( timeout 300 python bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err ); python -c "
import json
d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],2), 'ms/step')"; done
