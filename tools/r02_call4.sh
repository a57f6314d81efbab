#!/bin/bash
set -u
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -s -m gpu > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log )
grep -h "rel err\|passed\|failed\|rc=\|C4 full\|FAILED\|fused Adam\|egress:" $O/gpu_suite.log | tail -40
