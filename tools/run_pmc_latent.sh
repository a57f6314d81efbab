#!/bin/bash
# The three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters; each with --kernel-trace only) of the LBBDM-f4 sampling step (c3) and training
# micro-step (c4) inside one gpurun call -> gpurun_out/<tag>/pmc_c{3,4}_traffic.json, pmc_c{3,4}_mfma_util.json, to be copied into profiles/
#   bash tools/run_pmc_latent.sh r05
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
O=gpurun_out/$TAG
mkdir -p $O
C3="--set hip_graph=0 --workload c3 --steps 1 --warmup 1 --no-cpu --no-parity --no-f32mfma --no-pipeline --no-op-profile"
C4="--workload c4 --steps 4 --warmup 0 --no-cpu --no-parity --no-f32mfma --no-op-profile"
for w in c3 c4; do
  if [ $w = c3 ]; then A="$C3"; else A="$C4"; fi
  ( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pmc_fetch_$w -o pmc -- python $R/bench.py $A > $R/$O/pmc_fetch_$w.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pmc_write_$w -o pmc -- python $R/bench.py $A > $R/$O/pmc_write_$w.log 2>&1 )
  ( cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/$O/pmc_sq_$w -o pmc -- python $R/bench.py $A > $R/$O/pmc_sq_$w.log 2>&1 )
done
python tools/rocprof_pmc.py $(find $O/pmc_fetch_c3 -name "*.db" | head -1) $(find $O/pmc_write_c3 -name "*.db" | head -1) "python bench.py $C3" 2 12.4e9 > $O/pmc_c3_traffic.json 2> $O/pmc34_err.log
python tools/rocprof_pmc.py $(find $O/pmc_fetch_c4 -name "*.db" | head -1) $(find $O/pmc_write_c4 -name "*.db" | head -1) "python bench.py $C4 (4 priming + 4 timed micro-steps = 8; the weight planes are re-packed after each of the 2 optimizer steps: excluded from the per-step figure like the one-time packing)" 8 > $O/pmc_c4_traffic.json 2>> $O/pmc34_err.log
for w in c3 c4; do python tools/rocprof_counters.py --json $(find $O/pmc_sq_$w -name "*.db" | head -1) > $O/pmc_${w}_mfma_util.json 2>> $O/pmc34_err.log; done
rm -rf $O/pmc_fetch_c3 $O/pmc_write_c3 $O/pmc_sq_c3 $O/pmc_fetch_c4 $O/pmc_write_c4 $O/pmc_sq_c4
python -c "
import json
for w in ('c3','c4'):
    d=json.load(open('$O/pmc_%s_traffic.json' % w)); print(w, d['totals'])
    d=json.load(open('$O/pmc_%s_mfma_util.json' % w)); print({k[:44]: (round(v['MfmaUtil%'],1), v.get('clock_GHz')) for k,v in d['kernels'].items() if v.get('MfmaUtil%') and v['MfmaUtil%']>20})"
tail -3 $O/pmc34_err.log
