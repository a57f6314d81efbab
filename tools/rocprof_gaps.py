#!/usr/bin/env python
"""GPU idle time inside a rocprofv3 kernel trace (rocpd sqlite): where does a launch-bound step wait for the host?

    python tools/rocprof_gaps.py <results.db> [last_ms]
Prints span / busy / idle of the LAST `last_ms` milliseconds of the trace (the timed steps; default: everything) and the idle time
aggregated by the kernel that FOLLOWS the gap (the launch the GPU waited for), plus a histogram of gap lengths."""
import collections
import re
import sqlite3
import sys


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"(?:[A-Za-z_0-9]+::)*([A-Za-z_0-9]+)(<[^(]{0,60})?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


def main(path, last_ms=0.0):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = db.execute(f"select name, {s}, {e} from kernels order by {s}").fetchall()
    if not rows:
        print("no kernels")
        return
    if last_ms > 0:
        t0 = rows[-1][2] - last_ms * 1e6
        rows = [r for r in rows if r[1] >= t0]
    span = rows[-1][2] - rows[0][1]
    busy, idle_by, hist = 0, collections.Counter(), collections.Counter()
    cnt_by = collections.Counter()
    cur_end = rows[0][1]
    for name, a, b in rows:
        if a > cur_end:
            gap = a - cur_end
            key = short(name)
            idle_by[key] += gap
            cnt_by[key] += 1
            hist[min(int(gap / 1e3).bit_length(), 12)] += gap
        if b > cur_end:
            busy += b - max(a, cur_end)
            cur_end = b
    idle = span - busy
    print(f"# GPU idle time in `{path}` (last {last_ms} ms of the trace; 0 = all)\n")
    print(f"span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms ({100 * idle / span:.1f} %) over {len(rows)} dispatches\n")
    print("| idle before kernel | gaps | idle ms | avg gap us |\n|---|---|---|---|")
    for k, v in idle_by.most_common(25):
        print(f"| `{k}` | {cnt_by[k]} | {v / 1e6:.3f} | {v / cnt_by[k] / 1e3:.1f} |")
    print("\n| gap length | idle ms |\n|---|---|")
    for b in sorted(hist):
        lo = 0 if b == 0 else 2 ** (b - 1)
        print(f"| {'>= ' if b == 12 else ''}{lo}{'' if b == 12 else f' .. {2 ** b}'} us | {hist[b] / 1e6:.3f} |")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
