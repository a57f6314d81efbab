set -u
O=gpurun_out/r03g
mkdir -p $O
( timeout 900 python bench.py --workload c4 --no-cpu --dump-ops $O/c4_ops.md > $O/bench_c4.json 2> $O/bench_c4.err ); tail -2 $O/bench_c4.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03g/bench_c4.json"))
print("c4", round(d["ms_per_step"],2), "ms")
print("   parity", {k:(round(v,9) if isinstance(v,float) else v) for k,v in (d["parity"] or {}).items() if k not in ("metric",)})
print({k: round(v,2) for k,v in d["kernel_ms_per_step"].items() if v>0.8})
PY
