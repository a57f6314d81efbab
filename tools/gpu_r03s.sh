set -u
O=gpurun_out/r03s
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in c3 c4; do ( timeout 300 python bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err ); python -c "
import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],3), d['parity'] and {k:v for k,v in d['parity'].items() if k.startswith('rel')}); print({k: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 0.3})"; done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4 -o c4 -- python $R/bench.py --workload c4 --steps 4 --warmup 1 --no-cpu --no-parity > $R/$O/prof_c4.log 2>&1 )
DB=$(find $O/prof_c4 -name "*.db" | head -1)
python tools/rocprof_summary.py $DB "python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu --no-parity" > $O/c4_kernel_stats.md 2>&1
python tools/rocprof_gaps.py $DB > $O/c4_gaps.md 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect("$DB")
cols=[r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
print(cols)
PY
head -40 $O/c4_gaps.md
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c5 -o c5 -- python $R/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu --no-parity --no-pipeline > $R/$O/prof_c5.log 2>&1 )
DB5=$(find $O/prof_c5 -name "*.db" | head -1)
python tools/rocprof_gaps.py $DB5 > $O/c5_gaps.md 2>&1; head -12 $O/c5_gaps.md
python tools/rocprof_summary.py $DB5 "python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu --no-parity --no-pipeline" > $O/c5_kernel_stats.md 2>&1
rm -rf $O/prof_c4 $O/prof_c5
