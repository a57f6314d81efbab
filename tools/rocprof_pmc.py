#!/usr/bin/env python
"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one pass each, with
--kernel-trace only -- the combination gpurun allows).

    python tools/rocprof_pmc.py FETCH_results.db WRITE_results.db "<command that was profiled>" [steps [algorithmic_bytes_per_step]] \
        > profiles/rNN_pmc_<wl>_traffic.json
(steps = forward passes the command ran: adds `totals` with the whole-step fabric bytes, one-time weight packing excluded)

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE are
reported in KiB; FETCH_SIZE under-counts by 2x on gfx950 (64 B requests are counted as 32 B), so
fetch_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as reported.  Both count L2 <-> fabric requests, so
Infinity-Cache (MALL) hits are included: an upper bound on HBM bytes."""
import json
import re
import sqlite3
import sys


def library_stamp():
    """sha256 of the kernel library these counters were taken on (bench.py shows the counter fields only while it matches the library it
    loaded: a PMC file of another build is reported as "stale", round-5 verdict item 6)."""
    import hashlib
    import os
    path = os.environ.get("BBDM_HIP_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bbdm_amd", "libbbdm_hip.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    q = "select kernel_name, value from counters_collection where counter_name = ?"
    for name, value in db.execute(q, (counter,)):
        short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
        short = re.sub(r"\(.*$", "", short)
        d = out.setdefault(short, [0, 0.0])
        d[0] += 1
        d[1] += float(value)
    return out


def main():
    fetch_db, write_db, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        n = max(nf, nw)
        fk = sf / nf if nf else 0.0
        wk = sw / nw if nw else 0.0
        kernels[k] = {"launches": n, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
                      "fabric_bytes_per_launch_corrected": (2.0 * fk + wk) * 1024.0}
    steps = float(sys.argv[4]) if len(sys.argv) > 4 else None         # forward passes the profiled command ran (warm-up + timed)
    totals = None
    if steps:
        allb = sum(v["fabric_bytes_per_launch_corrected"] * v["launches"] for v in kernels.values())
        packb = sum(v["fabric_bytes_per_launch_corrected"] * v["launches"] for k, v in kernels.items() if "pack" in k or "weight_kernel" in k or "weight_planes_kernel" in k)
        totals = {"steps_profiled": steps, "all_kernels_bytes": allb, "one_time_weight_packing_bytes": packb,
                  "bytes_per_step_excl_packing": (allb - packb) / steps,
                  "algorithmic_bytes_per_step": float(sys.argv[5]) if len(sys.argv) > 5 else None}
    json.dump({"command": cmd, "library_sha256": library_stamp(), "totals": totals,
               "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; fetch_bytes_corrected = 2 x FETCH_SIZE x 1024 (gfx950 "
                        "half-count, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported. They count L2<->fabric "
                        "requests: Infinity-Cache (MALL) hits are INCLUDED, so this is an upper bound on HBM bytes",
               "kernels": kernels}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
