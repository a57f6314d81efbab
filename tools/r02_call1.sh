#!/bin/bash
# Round-2 GPU call 1: real-size parity tests, the three A/B runs round 1 left unmeasured, and a first bench line with
# the new parity / cpu_baseline fields.  Everything lands in gpurun_out/r02a/.
set -u
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -s -m gpu > $O/fullsize.log 2>&1; echo "rc=$?" >> $O/fullsize.log ) 
tail -5 $O/fullsize.log
( timeout 400 bash tools/ab_gemm.sh > $O/ab_gemm.log 2>&1 )
( timeout 500 bash tools/ab_transforms.sh > $O/ab_transforms.log 2>&1 )
( timeout 600 bash tools/ab_winograd6.sh > $O/ab_winograd6.log 2>&1 )
( timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err )
tail -c 1500 $O/bench_c2.json
grep -h "rel err\|passed\|failed\|rc=" $O/fullsize.log | tail -20
cat $O/ab_gemm.log | tail -30
cat $O/ab_transforms.log | tail -12
cat $O/ab_winograd6.log | tail -40
