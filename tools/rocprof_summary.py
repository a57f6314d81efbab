#!/usr/bin/env python
"""Summarise a rocprofv3 results database (rocpd sqlite, what `rocprofv3 --kernel-trace --stats` writes on ROCm 7.2)
into the per-kernel table kept under profiles/:  calls, total / average / min / max duration, share of GPU time.

    python tools/rocprof_summary.py gpurun_out/prof_c2/c2_results.db > profiles/r01_c2_kernel_stats.md
"""
import sqlite3
import sys


def main(path, title=""):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary {title}\n")
    print(f"source db: `{path}`; total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds, scr, wg in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{short}` | {n} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.2f} | "
              f"{vg} | {ag} | {sg} | {lds} | {scr} | {wg} |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
