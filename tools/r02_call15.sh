#!/bin/bash
set -u
O=gpurun_out/r02o
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_first_stage_gpu.py tests/test_kernels_gpu.py -q -s -m gpu -k "first_stage or vq or conv2d or bf3" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -h "VQ-f4\|passed\|failed\|rc=\|FAILED" $O/tests.log | tail -6
for w in c3 c5; do ( timeout 400 python bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err ); python - <<PY
import json
d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],2), d.get('pipeline'))
PY
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4 -o c4 -- python $R/bench.py --workload c4 --steps 8 --warmup 4 --no-cpu > $R/$O/prof_c4.log 2>&1 )
python tools/rocprof_summary.py $(find $O/prof_c4 -name "*.db" | head -1) "python bench.py --workload c4 --steps 8 --warmup 4 --no-cpu" > $O/c4_kernel_stats.md 2>&1
rm -rf $O/prof_c4
head -28 $O/c4_kernel_stats.md | cut -c1-200
