set -u
O=gpurun_out/r03e
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf3p" > $O/tests_bf3p.log 2>&1; echo "tests rc=$?" >> $O/tests_bf3p.log
tail -2 $O/tests_bf3p.log
BBDM_WINO_INPUT_LDS=0 timeout 600 python tools/bf3p_bench.py --reps 10 --kernels 6 > $O/bf3p_bench_lds0.txt 2>&1
timeout 600 python tools/bf3p_bench.py --reps 10 --kernels 6 > $O/bf3p_bench_lds1.txt 2>&1
paste -d'\n' <(cut -c1-70 $O/bf3p_bench_lds0.txt) <(cut -c1-70 $O/bf3p_bench_lds1.txt) | grep -v "^$" | tail -40
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --dump-ops $O/c2_ops.md > $O/bench_c2.json 2> $O/bench_c2.err )
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03e/bench_c2.json"))
print("c2", round(d["ms_per_step"],2), "ms", {k: round(v,2) for k,v in d["kernel_ms_per_step"].items() if v>0.5})
PY
