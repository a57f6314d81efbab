#!/bin/bash
# What the part sustains on a bare MFMA stream, with the power / clock the driver reports beside it, and the same telemetry during the
# C2 bench:   bash tools/run_mfma_peak.sh <outdir>     (inside a gpurun call)
set -u
O=${1:-gpurun_out/mfma_peak}
mkdir -p $O
hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_peak.hip -o tools/microbench/mfma_peak || exit 1
smi() {   # one line per sample: time, average socket power (W), sclk (MHz), as rocm-smi prints them
    while true; do
        echo "$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr -s ' ' | tr '\n' ';')"
        sleep 0.2
    done
}
smi > $O/smi_mfma_peak.txt & SMI=$!
./tools/microbench/mfma_peak 3 3 > $O/mfma_peak.txt 2>&1
./tools/microbench/mfma_peak 3 2 >> $O/mfma_peak.txt 2>&1
kill $SMI
# burst mode (round 5): the step's duty cycle instead of a continuous stream -- see burst_main() in mfma_peak.hip
smi > $O/smi_mfma_burst.txt & SMI=$!
./tools/microbench/mfma_peak burst 3 > $O/mfma_burst.txt 2>&1
kill $SMI
smi > $O/smi_bench_c2.txt & SMI=$!
python bench.py --steps 20 --warmup 5 --no-cpu --no-parity --no-f32mfma --no-extras > $O/bench_c2.json 2> $O/bench_c2.err
kill $SMI
cat $O/mfma_peak.txt $O/mfma_burst.txt
python - <<P
import re, json
for f in ("smi_mfma_peak", "smi_mfma_burst", "smi_bench_c2"):
    pw, ck = [], []
    for ln in open("$O/%s.txt" % f):
        m = re.search(r"Power[^;]*?:\s*([0-9.]+)", ln); c = re.search(r"sclk[^;]*?\(([0-9.]+)Mhz\)", ln)
        if m: pw.append(float(m.group(1)))
        if c: ck.append(float(c.group(1)))
    if pw: print(f, "power W: max %.0f  p50 %.0f  n %d" % (max(pw), sorted(pw)[len(pw)//2], len(pw)), "| sclk MHz: max %.0f min %.0f" % ((max(ck), min(ck)) if ck else (0, 0)))
    else: print(f, "no power samples (rocm-smi unavailable to this user?)"); print(open("$O/%s.txt" % f).read()[:400])
d = json.load(open("$O/bench_c2.json")); print("c2 ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
P
