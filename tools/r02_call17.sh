#!/bin/bash
O=gpurun_out/r02w2; mkdir -p $O; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_backward_kernels_gpu.py -x -q -k "winograd_wgrad or gemm_tn" 2>&1 | tail -2
timeout 300 python tools/wgrad_bench.py > $O/wgrad_bench.txt 2>&1; cat $O/wgrad_bench.txt
timeout 300 python bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; tail -3 $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4', d['ms_per_step'], json.dumps(d['roofline'])[:900]); print(d['kernel_ms_per_step'])"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4 -o c4 -- python $R/bench.py --workload c4 --steps 4 --warmup 1 --no-cpu > $R/$O/prof_c4.log 2>&1 )
python tools/rocprof_summary.py $(find $O/prof_c4 -name "*.db" | head -1) "python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu" > $O/c4_kernel_stats.md 2>&1
rm -rf $O/prof_c4; head -40 $O/c4_kernel_stats.md | cut -c1-200
