#!/bin/bash
O=gpurun_out/r02w3; mkdir -p $O
timeout 900 python -m pytest tests/test_backward_kernels_gpu.py tests/test_training_gpu.py -x -q > $O/tests.log 2>&1; echo rc=$?
grep -E "passed|failed|Error|error|assert" $O/tests.log | tail -10
