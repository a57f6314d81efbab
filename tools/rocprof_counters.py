#!/usr/bin/env python
"""Per-kernel averages of the hardware counters of one or more rocprofv3 PMC passes (each pass: --pmc <counters> --kernel-trace).

    python tools/rocprof_counters.py pass1_results.db [pass2_results.db ...] > profiles/rNN_pmc_<what>.md
Prints one markdown table: kernel, launches, then the per-launch mean of every counter, plus derived columns when their inputs
are present: MfmaUtil % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), the SQ wave-cycle split
(WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY over WAVE_CYCLES: parked at a waitcnt or barrier / issue-stalled / issuing), the LDS
bank-conflict share of the LDS-array cycles, and the effective clock GRBM_GUI_ACTIVE / kernel time."""
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


def main():
    data, order = {}, []
    dur = {}
    for path in [a for a in sys.argv[1:] if not a.startswith('--')]:
        db = sqlite3.connect(path)
        for name, cname, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
            short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
            short = re.sub(r"\(.*$", "", short)
            d = data.setdefault(short, {})
            c = d.setdefault(cname, [0, 0.0])
            c[0] += 1
            c[1] += float(value)
            if cname not in order:
                order.append(cname)
        if "--schema" in sys.argv:
            for r in db.execute("select name from sqlite_master where type='table'"):
                print("table", r[0], file=sys.stderr)
        try:                                             # kernel durations (the `kernels` view of the rocpd schema, ns)
            for name, ns in db.execute("select name, duration from kernels"):
                short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
                short = re.sub(r"\(.*$", "", short)
                t = dur.setdefault(short, [0, 0.0])
                t[0] += 1
                t[1] += float(ns)
        except sqlite3.Error:
            pass
    derived = ["MfmaUtil%", "parked%", "issue_stall%", "issuing%", "lds_conflict%", "clock_GHz"]
    as_json = "--json" in sys.argv
    jout = {}
    if not as_json:
        print("| kernel | launches | us | " + " | ".join(order + derived) + " |")
        print("|---|---|---|" + "---|" * (len(order) + len(derived)))
    for k in sorted(data, key=lambda k: -sum(v[1] for v in data[k].values())):
        d = {c: v[1] / v[0] for c, v in data[k].items()}
        n = max(v[0] for v in data[k].values())
        us = dur[k][1] / dur[k][0] / 1e3 if k in dur and dur[k][0] else float("nan")
        g = lambda c: d.get(c)
        row = [f"{d[c]:.4g}" if c in d else "" for c in order]
        # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs (32 cycles per
        # v_mfma_f32_32x32x16_bf16: MI355X_MICROARCH.md)
        mf = 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / (g("GRBM_GUI_ACTIVE") / 8) if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE") else None
        wc = g("SQ_WAVE_CYCLES")
        pct = lambda c: 100 * g(c) / wc if wc and g(c) is not None else None
        ldc = 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None else None
        clk = g("GRBM_GUI_ACTIVE") / 8 / (us * 1e3) if g("GRBM_GUI_ACTIVE") and us == us and us > 0 else None
        dv = [mf, pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), pct("SQ_ACTIVE_INST_ANY"), ldc, clk]
        if as_json:
            jout[k] = dict({"launches": n, "avg_us": None if us != us else us}, **d, **{nm: v for nm, v in zip(derived, dv)})
        else:
            print(f"| {k} | {n} | {us:.1f} | " + " | ".join(row + [("" if v is None else f"{v:.3g}") for v in dv]) + " |")
    if as_json:
        import json
        json.dump({"library_sha256": library_stamp(), "units": "per-launch means; GRBM_GUI_ACTIVE summed over the 8 XCDs, SQ_* summed over the chip; MfmaUtil% = "
                            "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)", "kernels": jout}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
