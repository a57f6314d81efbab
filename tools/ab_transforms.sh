#!/bin/bash
# Round-2 A/B of the experimental branch-free Winograd transforms (BBDM_WINO_BL=1; inside one gpurun call, ~2 GPU-minutes):
# parity tests under the switch, then the C2 step with and without it (compare winograd_input / winograd_output ms).
set -u
BBDM_WINO_BL=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "wino" 2>&1 | tail -2
for bl in 0 1; do
    echo "== bench c2, BBDM_WINO_BL=$bl"
    BBDM_WINO_BL=$bl timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('ms/step %.2f  steps/s %.3f' % (d['ms_per_step'], d['value']))
print({k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})"
done
