#!/usr/bin/env python
"""Idle time between the kernels of a hipGraph-replayed forward (rocprofv3 --kernel-trace rocpd database): the forwards are delimited by
their first kernel (nchw_to_nhwc_kernel); prints, for the last `n` forwards, span / busy / idle per forward and the gap histogram.
    python tools/graph_gaps.py <results.db> [n]"""
import collections
import sqlite3
import sys


def main(path, n=10):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = db.execute(f"select name, {s}, {e} from kernels order by {s}").fetchall()
    starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc_kernel" in r[0]]
    if len(starts) < n + 1:
        print("too few forwards in the trace:", len(starts))
        return
    spans, busys, counts = [], [], []
    hist = collections.Counter()
    for a, b in zip(starts[-n - 1:-1], starts[-n:]):
        seg = rows[a:b]
        span = seg[-1][2] - seg[0][1]
        busy = sum(r[2] - r[1] for r in seg)
        spans.append(span); busys.append(busy); counts.append(len(seg))
        for x, y in zip(seg[:-1], seg[1:]):
            g = max(0, y[1] - x[2])
            hist[min(int(g / 250), 40)] += 1
    m = lambda v: sum(v) / len(v)
    print(f"last {n} forwards: {m(counts):.0f} kernels each, span {m(spans) / 1e6:.3f} ms, busy {m(busys) / 1e6:.3f} ms, "
          f"idle {(m(spans) - m(busys)) / 1e6:.3f} ms ({100 * (1 - m(busys) / m(spans)):.1f} %), mean gap {(m(spans) - m(busys)) / max(1, m(counts) - 1) / 1e3:.2f} us")
    print("gap histogram (0.25 us bins):", {f"{k * 0.25:.2f}": v for k, v in sorted(hist.items())})


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
