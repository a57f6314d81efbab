set -u
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf3p" > $O/tests_bf3p.log 2>&1; echo "tests rc=$?" >> $O/tests_bf3p.log
tail -3 $O/tests_bf3p.log
BBDM_WINO_XCD_RUNS=0 timeout 600 python tools/bf3p_bench.py --reps 10 --kernels 6 > $O/bf3p_bench_runs0.txt 2>&1
tail -3 $O/bf3p_bench_runs0.txt
timeout 600 python tools/bf3p_bench.py --reps 10 --kernels 4,5,6 > $O/bf3p_bench.txt 2>&1
tail -19 $O/bf3p_bench.txt | cut -c1-260
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/$O/pmc1 -o pmc -- python $R/tools/bf3p_bench.py --shapes 0 --kernels 6 --reps 3 > $R/$O/pmc1.log 2>&1 )
python tools/rocprof_counters.py --schema $(find $O/pmc1 -name "*.db" | head -1) > $O/pmc_gemm.md 2> $O/pmc_gemm.err
head -5 $O/pmc_gemm.md | cut -c1-300; cat $O/pmc_gemm.err | head -40
rm -rf $O/pmc1
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --dump-ops $O/c2_ops.md > $O/bench_c2.json 2> $O/bench_c2.err )
python - <<'PY'
import json
for n in ("bench_c2",):
    try:
        d=json.load(open("gpurun_out/r03c/%s.json"%n))
        print(n, round(d["ms_per_step"],2), "ms")
        print("   ", {k: round(v,2) for k,v in d["kernel_ms_per_step"].items() if v>0.5})
    except Exception as e:
        print(n, "FAILED", e)
PY
