#!/bin/bash
# A/B of the tile-GEMM loop variants on the Winograd shapes of the 256^2 / batch-16 step (seconds of GPU time each):
#   bash tools/ab_gemm.sh            (inside one gpurun call)
set -u
for cfg in "default" "BBDM_GEMM_PREFETCH2=1" "BBDM_CONV_GLDS=1"; do
    echo "== $cfg"
    if [ "$cfg" = default ]; then timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
    else env $cfg timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids; fi
done
echo "== correctness of the opt-in loop (Winograd kernel tests under BBDM_GEMM_PREFETCH2=1)"
BBDM_GEMM_PREFETCH2=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k wino 2>&1 | tail -2
