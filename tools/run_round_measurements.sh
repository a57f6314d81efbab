#!/bin/bash
# The round's measurement set (inside one gpurun call): bench lines of every workload, rocprofv3 kernel stats of the C2 step and the C4
# training step, the two PMC traffic passes (FETCH_SIZE / WRITE_SIZE, each with --kernel-trace only) behind roofline.traffic /
# traffic_step, and the SQ pass behind roofline.mfma_util.
#   bash tools/run_round_measurements.sh r03        -> gpurun_out/<tag>/..., summaries to copy into profiles/
set -u
TAG=${1:-r04}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# PMC passes FIRST: their summaries are copied into profiles/ of this (box-side) copy of the tree, so that the bench lines written
# after them carry the PMC-derived fields (roofline.traffic, traffic_step, mfma_util) checked against the hash of this very library.
# The c3 / c4 passes (tools/run_pmc_latent.sh) run here too.
CMD="python bench.py --set hip_graph=0 --workload c2 --steps 1 --warmup 1 --no-cpu --no-parity --no-f32mfma --no-extras"
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pmc_fetch -o pmc -- python $R/bench.py --set hip_graph=0 --workload c2 --steps 1 --warmup 1 --no-cpu --no-parity --no-f32mfma --no-extras > $R/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pmc_write -o pmc -- python $R/bench.py --set hip_graph=0 --workload c2 --steps 1 --warmup 1 --no-cpu --no-parity --no-f32mfma --no-extras > $R/$O/pmc_write.log 2>&1 )
# 2 forward passes were profiled (1 warm-up + 1 timed); SURVEY.md 8(d): 92.3 GB of algorithmic HBM traffic per C2 step
python tools/rocprof_pmc.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) "$CMD" 2 92.3e9 > $O/pmc_c2_traffic.json 2> $O/pmc_err.log
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/$O/pmc_sq -o pmc -- python $R/bench.py --set hip_graph=0 --workload c2 --steps 1 --warmup 1 --no-cpu --no-parity --no-f32mfma --no-extras > $R/$O/pmc_sq.log 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $R/$O/pmc_lds -o pmc -- python $R/bench.py --set hip_graph=0 --workload c2 --steps 1 --warmup 1 --no-cpu --no-parity --no-f32mfma --no-extras > $R/$O/pmc_lds.log 2>&1 )
python tools/rocprof_counters.py --json $(find $O/pmc_sq -name "*.db" | head -1) > $O/pmc_c2_mfma_util.json 2>> $O/pmc_err.log
python tools/rocprof_counters.py $(find $O/pmc_sq -name "*.db" | head -1) $(find $O/pmc_lds -name "*.db" | head -1) > $O/pmc_c2_counters.md 2>> $O/pmc_err.log
bash tools/run_pmc_latent.sh $TAG > $O/pmc34.log 2>&1
for f in pmc_c2_traffic pmc_c2_mfma_util pmc_c3_traffic pmc_c3_mfma_util pmc_c4_traffic pmc_c4_mfma_util; do
  [ -s $O/$f.json ] && cp $O/$f.json profiles/${TAG}_$f.json
done
# the driver's own command first (c2 headline + the four other configs in `workloads`), then c2 alone with the per-launch table
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err )
( timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --dump-ops $O/c2_per_launch.md > $O/bench_c2.json 2> $O/bench_c2.err )
for w in c1 c3 c5; do ( timeout 300 python bench.py --workload $w --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err ); done
( timeout 600 python bench.py --workload c4 --no-cpu --dump-ops $O/c4_per_launch.md > $O/bench_c4.json 2> $O/bench_c4.err )
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c2 -o c2 -- python $R/bench.py --set hip_graph=0 --workload c2 --steps 3 --warmup 1 --no-cpu --no-parity --no-f32mfma --no-extras > $R/$O/prof_c2.log 2>&1 )
python tools/rocprof_summary.py $(find $O/prof_c2 -name "*.db" | head -1) "python bench.py --set hip_graph=0 --workload c2 --steps 3 --warmup 1 --no-cpu --no-parity --no-f32mfma --no-extras (eager launches: a graph capture would run the first forward twice)" > $O/c2_kernel_stats.md 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c4 -o c4 -- python $R/bench.py --workload c4 --steps 4 --warmup 1 --no-cpu --no-parity --no-f32mfma > $R/$O/prof_c4.log 2>&1 )
python tools/rocprof_summary.py $(find $O/prof_c4 -name "*.db" | head -1) "python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu --no-parity --no-f32mfma (training: 4 priming + 1 warm-up + 4 timed + 4 per-op-profiled micro-steps = 13)" > $O/c4_kernel_stats.md 2>&1
rm -rf $O/prof_c2 $O/prof_c4 $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds
head -14 $O/c2_kernel_stats.md; head -16 $O/c4_kernel_stats.md; python -c "
import json
for w in ('c2','c1','c3','c5','c4'):
    d=json.load(open('$O/bench_%s.json' % w)); print(w, round(d['ms_per_step'],2), 'ms', round(d['value'],3), 'frac', round(d['roofline']['frac'],3), 'frac_step', round(d['roofline']['frac_step'],3), {k:v for k,v in (d['parity'] or {}).items() if k.startswith('rel_err')})
d=json.load(open('$O/bench_default.json')); print('default line:', round(d['ms_per_step'],2), {w: (round(v['ms_per_step'],3), round(v['frac_step'],3)) for w, v in d['workloads'].items()})
d=json.load(open('$O/pmc_c2_traffic.json')); print({k: round(v['fabric_bytes_per_launch_corrected']/1e9,3) for k,v in d['kernels'].items() if v['fabric_bytes_per_launch_corrected']>1e8}); print(d['totals'])
d=json.load(open('$O/pmc_c2_mfma_util.json')); print({k[:40]: (round(v['MfmaUtil%'],1) if v.get('MfmaUtil%') else None, v.get('clock_GHz')) for k,v in d['kernels'].items() if v.get('MfmaUtil%')})"
tail -3 $O/pmc_err.log
