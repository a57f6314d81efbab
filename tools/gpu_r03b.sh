set -u
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf3p" > $O/tests_bf3p.log 2>&1; echo "tests rc=$?" >> $O/tests_bf3p.log
tail -3 $O/tests_bf3p.log
timeout 600 python tools/bf3p_bench.py --reps 10 --kernels 0,1,3,4,5 > $O/bf3p_bench.txt 2>&1
tail -20 $O/bf3p_bench.txt
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/$O/pmc1 -o pmc -- python $R/tools/bf3p_bench.py --shapes 0,2 --kernels 3,4 --reps 3 > $R/$O/pmc1.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $R/$O/pmc2 -o pmc -- python $R/tools/bf3p_bench.py --shapes 0,2 --kernels 3,4 --reps 3 > $R/$O/pmc2.log 2>&1 )
python tools/rocprof_counters.py $(find $O/pmc1 -name "*.db" | head -1) $(find $O/pmc2 -name "*.db" | head -1) > $O/pmc_gemm.md 2> $O/pmc_gemm.err
cat $O/pmc_gemm.md | cut -c1-400 | head -12; tail -3 $O/pmc_gemm.err
rm -rf $O/pmc1 $O/pmc2
( BBDM_BF3P_KERNEL=4 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_c2_k4.json 2> $O/bench_c2_k4.err )
( BBDM_BF3P_KERNEL=3 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_c2_k3.json 2> $O/bench_c2_k3.err )
python - <<'PY'
import json
for n in ("k4","k3"):
    try:
        d=json.load(open("gpurun_out/r03b/bench_c2_%s.json"%n))
        print(n, round(d["ms_per_step"],2), "ms")
        print("   ", {k: round(v,2) for k,v in d["kernel_ms_per_step"].items() if v>0.5})
    except Exception as e:
        print(n, "FAILED", e)
PY
