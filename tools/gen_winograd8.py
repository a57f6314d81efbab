#!/usr/bin/env python
"""Generate the m = 8 branches of csrc/winograd_math.h (bt_transform, at_transform, g_transform, a_transform, gt_transform and the gain
of the input transform) from a point set {0, +-p1, +-p2, +-p3, +-p4, inf} by the Cook-Toom construction of
tests/test_winograd_math_cpu.py::cook_toom, in exact rational arithmetic: rows of B^T / columns of A^T scaled to dyadic rationals (exact in
fp32, printed as hex floats), G carrying the reciprocals (printed as quotients of integers, evaluated in fp64).

    python tools/gen_winograd8.py 5/4 9/4 2/5 4/5            # prints the five code blocks; --patch rewrites winograd_math.h in place
"""
import math
import os
import re
import sys
from fractions import Fraction as Fr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cook_toom_exact(pairs, m=8, r=3):
    pts = [Fr(0)] + [s * Fr(p) for p in pairs for s in (1, -1)]
    n = m + r - 1

    def polymul(a, b):
        c = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                c[i + j] += x * y
        return c
    AT = [[Fr(0)] * n for _ in range(m)]
    G = [[Fr(0)] * r for _ in range(n)]
    BT = [[Fr(0)] * n for _ in range(n)]
    for j, p in enumerate(pts):
        for i in range(m):
            AT[i][j] = p ** i
        Nj = math.prod([p - q for k, q in enumerate(pts) if k != j], start=Fr(1))
        for k in range(r):
            G[j][k] = p ** k / Nj
        f = [Fr(1)]
        for k, q in enumerate(pts):
            if k != j:
                f = polymul(f, [-q, Fr(1)])
        BT[j][:len(f)] = f
    AT[m - 1][n - 1] = Fr(1)
    G[n - 1][r - 1] = Fr(1)
    f = [Fr(1)]
    for q in pts:
        f = polymul(f, [-q, Fr(1)])
    BT[n - 1][:len(f)] = f

    def odd_den(vals):
        den = 1
        for c in vals:
            d = c.denominator
            while d % 2 == 0:
                d //= 2
            den = den * d // math.gcd(den, d)
        return Fr(den)

    def pow2_norm(vals):
        mx, e = max(abs(c) for c in vals), 0
        while mx >= 2:
            mx, e = mx / 2, e - 1
        while mx < 1:
            mx, e = mx * 2, e + 1
        return Fr(2) ** e
    for j in range(n):
        s = odd_den(BT[j])
        s *= pow2_norm([c * s for c in BT[j]])
        BT[j] = [c * s for c in BT[j]]
        G[j] = [c / s for c in G[j]]
        col = [AT[i][j] for i in range(m)]
        s = odd_den(col)
        s *= pow2_norm([c * s for c in col])
        for i in range(m):
            AT[i][j] *= s
        G[j] = [c / s for c in G[j]]
    return BT, G, AT


def hexf(c: Fr) -> str:
    """A dyadic rational as a C hex-float literal (exact), e.g. 0x1.2p-3f."""
    v = float(c)
    assert Fr(v) == c and float.fromhex(float(v).hex()) == v, c
    import struct
    assert struct.unpack("f", struct.pack("f", v))[0] == v, f"{c} is not exact in fp32"
    h = abs(v).hex()                         # 0x1.2000000000000p-3
    mant, exp = h.split("p")
    mant = mant.rstrip("0").rstrip(".")
    return ("-" if v < 0 else "") + mant + "p" + exp + "f"


def quo(c: Fr) -> str:
    return f"(T)({c.numerator}.0 / {c.denominator}.0)"


def term(c: Fr, var: str, first: bool) -> str:
    if c == 0:
        return ""
    mag = hexf(abs(c))
    body = var if abs(c) == 1 else f"{mag} * {var}"
    if first:
        return ("(-" + body + ")") if c < 0 and abs(c) != 1 else (("-1.f * " + var) if c < 0 else body)
    return (" - " if c < 0 else " + ") + body


def lin(coeffs, names) -> str:
    out, first = "", True
    for c, nm in zip(coeffs, names):
        t = term(c, nm, first)
        if t:
            out += t
            first = False
    return out or "0.f"


def generate(pairs):
    BT, G, AT = cook_toom_exact(pairs)
    n, m = 10, 8
    d = [f"d[{k}]" for k in range(n)]
    blocks = {}
    # ---- bt_transform<8>: point 0 (even polynomial), pairs (ev +- od), infinity (odd polynomial)
    L = []
    L.append(f"        t[0] = {lin(BT[0], d)};")
    for k in range(4):
        rp, rm = BT[1 + 2 * k], BT[2 + 2 * k]
        ev = [(rp[c] + rm[c]) / 2 for c in range(n)]
        od = [(rp[c] - rm[c]) / 2 for c in range(n)]
        assert all(ev[c] == 0 for c in range(1, n, 2)) and all(od[c] == 0 for c in range(0, n, 2))
        L.append("        {")
        L.append(f"            const T ev = {lin(ev, d)};")
        L.append(f"            const T od = {lin(od, d)};")
        L.append(f"            t[{1 + 2 * k}] = ev + od;")
        L.append(f"            t[{2 + 2 * k}] = ev - od;")
        L.append("        }")
    L.append(f"        t[9] = {lin(BT[9], d)};")
    blocks["bt"] = "\n".join(L)
    # ---- at_transform<8>
    L = ["        const T p0 = m[1] + m[2], q0 = m[1] - m[2], p1 = m[3] + m[4], q1 = m[3] - m[4], p2 = m[5] + m[6], q2 = m[5] - m[6],",
         "                p3 = m[7] + m[8], q3 = m[7] - m[8];"]
    for i in range(m):
        names = [f"{'p' if i % 2 == 0 else 'q'}{k}" for k in range(4)]
        co = [AT[i][1 + 2 * k] for k in range(4)]
        for k in range(4):
            assert AT[i][2 + 2 * k] == (AT[i][1 + 2 * k] if i % 2 == 0 else -AT[i][1 + 2 * k])
        expr = lin(co, names)
        if AT[i][0] != 0:
            expr = lin([AT[i][0]], ["m[0]"]) + " + " + expr
        if AT[i][9] != 0:
            expr = expr + " + " + lin([AT[i][9]], ["m[9]"])
        L.append(f"        s[{i}] = {expr};")
    blocks["at"] = "\n".join(L)
    # ---- g_transform<8>
    gname = ["g[0]", "g[1]", "g[2]"]
    L = []
    for j in range(n):
        parts = [f"{quo(G[j][k])} * {gname[k]}" for k in range(3) if G[j][k] != 0]
        L.append(f"        u[{j}] = " + " + ".join(parts) + ";")
    blocks["g"] = "\n".join(L)
    # ---- a_transform<8> (transpose of A^T)
    s_ = [f"s[{i}]" for i in range(m)]
    L = []
    for k in range(4):
        col = [AT[i][1 + 2 * k] for i in range(m)]
        ev = [col[i] if i % 2 == 0 else Fr(0) for i in range(m)]
        od = [col[i] if i % 2 == 1 else Fr(0) for i in range(m)]
        L.append(f"        const T e{k} = {lin(ev, s_)};")
        L.append(f"        const T o{k} = {lin(od, s_)};")
    assert all(AT[i][0] == (1 if i == 0 else 0) for i in range(m)) and all(AT[i][9] == (1 if i == 7 else 0) for i in range(m))
    L.append("        t[0] = s[0];")
    for k in range(4):
        L.append(f"        t[{1 + 2 * k}] = e{k} + o{k};")
        L.append(f"        t[{2 + 2 * k}] = e{k} - o{k};")
    L.append("        t[9] = s[7];")
    blocks["a"] = "\n".join(L)
    # ---- gt_transform<8> (transpose of G)
    L = []
    for k in range(3):
        parts = [f"{quo(G[j][k])} * u[{j}]" for j in range(n) if G[j][k] != 0]
        L.append(f"        g[{k}] = " + " + ".join(parts) + ";")
    blocks["gt"] = "\n".join(L)
    gain = max(sum(abs(c) for c in row) for row in BT) ** 2
    blocks["gain"] = float(gain)
    blocks["ggain"] = float(max(sum(abs(c) for c in row) for row in G) ** 2)          # |G g G^T| <= ggain max |g|
    blocks["dygain"] = float(max(sum(abs(AT[i][j]) for i in range(m)) for j in range(n)) ** 2)      # |A dY A^T| <= dygain max |dY|
    return blocks


def patch(pairs):
    b = generate(pairs)
    path = os.path.join(ROOT, "bbdm_amd", "csrc", "winograd_math.h")
    s = open(path).read()
    pts = ", ".join("+-" + p for p in pairs)
    for key in ("bt", "at", "g", "a", "gt"):
        pat = re.compile(r"(// <gen8:%s>[^\n]*\n)(.*?)(\n\s*// </gen8:%s>)" % (key, key), re.S)
        assert pat.search(s), key
        s = pat.sub(lambda mo: mo.group(1) + b[key] + mo.group(3), s, count=1)
    gain = math.ceil(b["gain"] * 100) / 100
    s = re.sub(r"m == 8 \? [0-9.]+f : 225\.f;[^\n]*", f"m == 8 ? {gain}f : 225.f;      // (m = 8, points {{0, {pts}, inf}}: {b['gain']:.4f})", s)
    gg = math.ceil(b["ggain"] * 100) / 100
    s = re.sub(r"m == 8 \? [0-9.]+f : 1\.55f;[^\n]*", f"m == 8 ? {gg}f : 1.55f;      // <gen8:ggain> (m = 8: {b['ggain']:.4f})", s)
    dg = math.ceil(b["dygain"] * 10) / 10
    s = re.sub(r"m == 8 \? [0-9.]+f : 3969\.f;[^\n]*", f"m == 8 ? {dg}f : 3969.f;      // (m = 8: {b['dygain']:.3f})", s)
    open(path, "w").write(s)
    print("patched", path, "gains", b["gain"], b["ggain"], b["dygain"])


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--patch" in sys.argv:
        patch(args)
    else:
        for k, v in generate(args).items():
            print(f"---- {k} ----\n{v}")
