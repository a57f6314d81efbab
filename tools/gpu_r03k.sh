set -u
O=gpurun_out/r03k
mkdir -p $O
timeout 1500 python -m pytest tests/test_backward_kernels_gpu.py -q -x -s -k "wgrad" > $O/tests1.log 2>&1; echo "tests rc=$?" >> $O/tests1.log; grep -E "bf16x3 TN|passed|failed|rc=|Error" $O/tests1.log | tail -16
timeout 1500 python -m pytest tests/test_training_gpu.py tests/test_first_stage_gpu.py -q -x > $O/tests2.log 2>&1; echo "tests rc=$?" >> $O/tests2.log; tail -3 $O/tests2.log
timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py -q -x -s -k "c4_full" > $O/tests3.log 2>&1; echo "tests rc=$?" >> $O/tests3.log; grep -E "sign flips|passed|failed|rc=|Error" $O/tests3.log | tail -8
( timeout 900 python bench.py --workload c4 --no-cpu --dump-ops $O/c4_ops.md > $O/bench_c4.json 2> $O/bench_c4.err ); tail -2 $O/bench_c4.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03k/bench_c4.json"))
print("c4", round(d["ms_per_step"],2), "ms", {k:v for k,v in d["parity"].items() if k.startswith("rel")})
print({k: round(v,2) for k,v in d["kernel_ms_per_step"].items() if v>0.8})
PY
