set -u
O=gpurun_out/r03f
mkdir -p $O
( timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_c2.json 2> $O/bench_c2.err ); tail -2 $O/bench_c2.err
for w in c1 c3 c5; do ( timeout 300 python bench.py --workload $w --no-cpu --no-pipeline > $O/bench_$w.json 2> $O/bench_$w.err ); tail -1 $O/bench_$w.err; done
( timeout 900 python bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err ); tail -2 $O/bench_c4.err
python - <<'PY'
import json
for w in ("c2","c1","c3","c5","c4"):
    try:
        d=json.load(open("gpurun_out/r03f/bench_%s.json"%w))
        print(w, round(d["ms_per_step"],2), "ms frac", round(d["roofline"]["frac"],3), "frac_step", round(d["roofline"]["frac_step"],3), "f32mfma", d.get("f32mfma_ms_per_step"))
        print("   parity", {k:(round(v,9) if isinstance(v,float) else v) for k,v in (d["parity"] or {}).items() if k not in ("metric","grad_errors")})
        print("   kernel", d["roofline"]["kernel"][:90])
    except Exception as e:
        print(w, "FAILED", e)
PY
