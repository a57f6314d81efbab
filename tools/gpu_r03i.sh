set -u
O=gpurun_out/r03i
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "upsampled_residual or winograd" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -2 $O/tests.log
timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_model_gpu.py -q -x > $O/tests2.log 2>&1; echo "tests rc=$?" >> $O/tests2.log; tail -2 $O/tests2.log
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-f32mfma --dump-ops $O/c2_ops.md > $O/bench_c2.json 2> $O/err.txt ); python - <<'PY'
import json
d=json.load(open("gpurun_out/r03i/bench_c2.json"))
print("c2", round(d["ms_per_step"],2), "ms", {k: round(v,2) for k,v in d["kernel_ms_per_step"].items() if v>0.3}, d["parity"] and {k:v for k,v in d["parity"].items() if k.startswith("rel")})
PY
