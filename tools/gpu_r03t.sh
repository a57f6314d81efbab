set -u
O=gpurun_out/r03t
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_latent_gpu.py -q -x > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log; tail -4 $O/tests1.log
timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py -q -x -k "c4 or c3 or c5 or c1" > $O/tests2.log 2>&1; echo "rc=$?" >> $O/tests2.log; tail -4 $O/tests2.log
( timeout 300 python bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err ); tail -2 $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4', round(d['ms_per_step'],3), d['parity'] and {k:v for k,v in d['parity'].items() if k.startswith('rel')})"
( BBDM_TRAIN_GRAPH=0 timeout 300 python bench.py --workload c4 --no-cpu --no-parity > $O/bench_c4_eager.json 2> $O/bench_c4_eager.err ); python -c "
import json; d=json.load(open('$O/bench_c4_eager.json')); print('c4 eager', round(d['ms_per_step'],3))"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4 -o c4 -- python $R/bench.py --workload c4 --steps 8 --warmup 4 --no-cpu --no-parity > $R/$O/prof_c4.log 2>&1 )
DB=$(find $O/prof_c4 -name "*.db" | head -1)
python tools/rocprof_gaps.py $DB 250 > $O/c4_gaps.md 2>&1
head -36 $O/c4_gaps.md
rm -rf $O/prof_c4
