#!/bin/bash
# Round-2 bring-up of the experimental F(6x6,3x3) path (inside one gpurun call; ~2 GPU-minutes):
#   kernel parity tests, per-layer timing against direct / F(2x2) / F(4x4), then the whole C2 step with the cap raised.
set -u
export BBDM_TEST_EXPERIMENTAL=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "m6" 2>&1 | tail -3
timeout 300 python tools/wino_bench.py --reps 3 2>&1 | grep -v amdgpu.ids
for cap in 4 6; do
    echo "== bench c2, BBDM_WINOGRAD=$cap"
    BBDM_WINOGRAD=$cap timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']
print('ms/step %.2f  steps/s %.3f  conv_igemm %.1f TFLOP/s' % (d['ms_per_step'], d['value'], r['achieved']))
print({k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})"
done
