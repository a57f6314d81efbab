#!/bin/bash
set -u
O=gpurun_out/r02c
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_backward_kernels_gpu.py tests/test_optim_gpu.py -q -s -m gpu > $O/bwd_kernels.log 2>&1; echo "rc=$?" >> $O/bwd_kernels.log )
grep -h "FAILED\|passed\|failed\|rc=\|fused Adam" $O/bwd_kernels.log | tail -30
( timeout 600 python tools/diag_c4_grads.py 64 > $O/diag64.log 2>&1 )
( timeout 300 python tools/diag_c4_grads.py 16 > $O/diag16.log 2>&1 )
grep -c . $O/diag64.log; awk '{print $0}' $O/diag64.log | head -80
for a in "" "--torch-adam"; do ( timeout 300 python bench.py --workload c4 --no-cpu $a > $O/bench_c4$a.json 2> $O/bench_c4$a.err ); tail -c 300 $O/bench_c4$a.json; done
