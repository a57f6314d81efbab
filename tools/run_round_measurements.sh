#!/bin/bash
# The round's measurement set (inside one gpurun call): bench lines of every workload, rocprofv3 kernel stats of the C2 step,
# and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each with --kernel-trace only) behind roofline.traffic.
#   bash tools/run_round_measurements.sh r02        -> gpurun_out/<tag>/..., summaries to copy into profiles/
set -u
TAG=${1:-r02}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 500 python bench.py --steps 20 --warmup 3 --dump-ops $O/c2_per_launch.md > $O/bench_c2.json 2> $O/bench_c2.err )
for w in c1 c3 c5 c4; do ( timeout 300 python bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err ); done
( BBDM_GEMM_BF3=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_c2_f32mfma.json 2> /dev/null )
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c2 -o c2 -- python $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu > $R/$O/prof_c2.log 2>&1 )
python tools/rocprof_summary.py $(find $O/prof_c2 -name "*.db" | head -1) "python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu" > $O/c2_kernel_stats.md 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4 -o c4 -- python $R/bench.py --workload c4 --steps 4 --warmup 1 --no-cpu > $R/$O/prof_c4.log 2>&1 )
python tools/rocprof_summary.py $(find $O/prof_c4 -name "*.db" | head -1) "python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu (training; the profiled second pass doubles the launches)" > $O/c4_kernel_stats.md 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pmc_fetch -o pmc -- python $R/bench.py --workload c2 --steps 1 --warmup 1 --no-cpu > $R/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pmc_write -o pmc -- python $R/bench.py --workload c2 --steps 1 --warmup 1 --no-cpu > $R/$O/pmc_write.log 2>&1 )
python tools/rocprof_pmc.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) "python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu" > $O/pmc_c2_traffic.json 2> $O/pmc_err.log
rm -rf $O/prof_c2 $O/prof_c4 $O/pmc_fetch $O/pmc_write
head -12 $O/c2_kernel_stats.md; head -14 $O/c4_kernel_stats.md; tail -c 600 $O/bench_c2.json; python -c "
import json
for w in ('c1','c3','c5','c4','c2_f32mfma'):
    d=json.load(open('$O/bench_%s.json' % w)); print(w, round(d['ms_per_step'],2), 'ms', round(d['value'],3))
d=json.load(open('$O/pmc_c2_traffic.json')); print({k: round(v['fabric_bytes_per_launch_corrected']/1e9,3) for k,v in d['kernels'].items() if v['fabric_bytes_per_launch_corrected']>1e8})"
