#!/bin/bash
# The first run on a multi-GPU node (none was available to rounds 1-5): the north-star's scaling table in one go.
#   bash tools/run_multigpu_measurements.sh [outdir]        (on a node with 8 MI355X; one process per GPU, RCCL over xGMI)
# What to compare afterwards (rank 0 prints ONE JSON line per command):
#   C2 sampling, N = 1, 2, 4, 8 (weak scaling, no data-path collective):  value (steps/s, whole job) ~ N x the N = 1 value;
#       ms_per_step is the max over ranks; roofline.frac / frac_step are rank 0's kernel numbers (should not move with N).
#   C4 training, N = 1, 2, 4, 8 (DDP gradient all-reduce over RCCL, only on the accumulation boundary):
#       training.boundary_micro_step_ms vs training.non_boundary_micro_step_ms   -> what the 948 MB all-reduce adds to 1 micro-step of 4,
#       training.allreduce_exposed_ms = boundary - non_boundary - optimizer_step -> the part NOT hidden behind the 4-segment backward,
#       training.sync_every_micro_step_ms_per_step                               -> the reference's behaviour (all-reduce on every
#                                                                                    micro-step, runners/BaseRunner.py:412-417) for the A/B,
#       training.rccl_version, devices[*]                                         -> what ran where.
#   2-rank RCCL correctness first: tests/test_dist_gpu.py (skipped on 1-GPU boxes) -- parameters identical across ranks after a step,
#       reduced gradients against the oracle on the combined batch.
set -u
O=${1:-gpurun_out/multigpu}
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_dist_gpu.py -x -q -m gpu > $O/pytest_dist_gpu.txt 2>&1; tail -3 $O/pytest_dist_gpu.txt
for N in 1 2 4 8; do
    python bench.py --gpus $N --steps 20 --warmup 5 --no-extras --no-cpu --no-f32mfma > $O/scale_c2_n$N.json 2> $O/scale_c2_n$N.err
    python bench.py --gpus $N --workload c4 --steps 8 --warmup 1 --no-cpu > $O/scale_c4_n$N.json 2> $O/scale_c4_n$N.err
done
python - <<P
import json
for w in ("c2", "c4"):
    base = None
    for n in (1, 2, 4, 8):
        try:
            d = json.load(open("$O/scale_%s_n%d.json" % (w, n)))
        except Exception as e:
            print(w, n, "no line:", e); continue
        base = base or d["value"]
        t = d.get("training") or {}
        print(w, "N", n, "value", round(d["value"], 3), "efficiency", round(d["value"] / (n * base), 3), "ms_per_step", round(d["ms_per_step"], 2),
              {k: t.get(k) for k in ("boundary_micro_step_ms", "non_boundary_micro_step_ms", "allreduce_exposed_ms", "sync_every_micro_step_ms_per_step")} if t else "")
P
