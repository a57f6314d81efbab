set -u
O=gpurun_out/r03a
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf3p" > $O/tests_bf3p.log 2>&1; echo "tests rc=$?" >> $O/tests_bf3p.log
tail -3 $O/tests_bf3p.log
timeout 600 python tools/bf3p_bench.py --reps 10 > $O/bf3p_bench.txt 2>&1
tail -22 $O/bf3p_bench.txt
( BBDM_GEMM_BF3P=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_c2_bf3.json 2> $O/bench_c2_bf3.err )
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --dump-ops $O/c2_ops_k0.md > $O/bench_c2_bf3p_k0.json 2> $O/bench_c2_bf3p_k0.err )
( BBDM_BF3P_KERNEL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_c2_bf3p_k1.json 2> $O/bench_c2_bf3p_k1.err )
python - <<'PY'
import json
for n in ("bf3","bf3p_k0","bf3p_k1"):
    try:
        d=json.load(open("gpurun_out/r03a/bench_c2_%s.json"%n))
        print(n, round(d["ms_per_step"],2), "ms frac", round(d["roofline"]["frac"],3), "frac_step", round(d["roofline"]["frac_step"],3), "parity", d.get("parity"))
        print("   ", {k: round(v,2) for k,v in d["kernel_ms_per_step"].items() if v>0.5})
    except Exception as e:
        print(n, "FAILED", e)
PY
