#!/bin/bash
# Round-2 GPU call 2: whole GPU suite on the cleaned-up build (m = 6 adopted, GLDS / PREFETCH2 / WINO_BL deleted),
# bench lines for every workload, rocprofv3 kernel stats of the C2 step.
set -u
O=gpurun_out/r02b
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -s -m gpu > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log )
grep -h "rel err\|passed\|failed\|rc=\|C4 full\|worst" $O/gpu_suite.log | tail -30
( timeout 400 python bench.py --steps 20 --warmup 3 --dump-ops $O/c2_ops.md > $O/bench_c2.json 2> $O/bench_c2.err )
tail -c 1200 $O/bench_c2.json
for w in c1 c3 c5 c4; do ( timeout 300 python bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err ); python -c "
import json,sys
d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],2), 'ms/step', round(d['value'],2), 'steps/s')" ; done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_c2.log 2>&1 )
python tools/rocprof_summary.py $(find $O/prof_c2 -name "*.db" | head -1) > $O/c2_kernel_stats.md 2>&1 || true
head -30 $O/c2_kernel_stats.md
