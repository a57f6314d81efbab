#!/usr/bin/env python
"""What bounds the Winograd transforms?  Builds ablated copies of csrc/winograd.hip ON THE GPU BOX (hipcc, ~15 s each; the other
objects of the in-tree build are re-linked unchanged) and times the plane-writing input transform and the output transform with
each: full kernel / stores suppressed / loads suppressed / non-temporal stores.  Ablated variants compute garbage by
construction -- this is a timing probe (tools/, not product).

    python tools/wino_variants.py [--reps 10]"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bbdm_amd", "csrc")
SRC = open(os.path.join(CSRC, "winograd.hip")).read()

# anchors in winograd_input_split2_kernel (the two-phase kernel, the default) and its launcher
IN_STORE = """            *reinterpret_cast<unsigned*>(o) = pl[jj][0];
            *reinterpret_cast<unsigned*>(o + 1024) = pl[jj][1];
            *reinterpret_cast<unsigned*>(o + 2048) = pl[jj][2];"""
IN_LOAD = "                d[i] = *reinterpret_cast<const float2*>(x + ((size_t)(n * Hs + hs) * Ws + wsrc) * ldx + c);"
IN_MAP = "    const int chunk = q % nchunks, tg = (q / nchunks) * 8 + (L & 7);"
IN_BLOCKS = "        const long long blocks = 8ll * ((TG + 7) / 8) * nchunks;"
IN_PLANE = "    const size_t plane = Tp * (size_t)CinPad * 6;"
PAD_BYTES = 4352            # 17 x 256 B


def in_variant(kind, s=None):
    s = SRC if s is None else s
    for a in (IN_STORE, IN_LOAD, IN_MAP, IN_BLOCKS, IN_PLANE) if s is SRC else ():
        assert a in s, "csrc/winograd.hip changed: update the patch anchors: " + a
    if kind == "no_store":          # keep the arithmetic alive: store only for a value that never occurs
        s = s.replace(IN_STORE, "            if (pl[jj][0] == 0x7fc12345u && pl[jj][1] == 0x7fc54321u && pl[jj][2] == 0x7fc99999u) *reinterpret_cast<unsigned*>(o) = pl[jj][0];")
    elif kind == "no_load":
        s = s.replace(IN_LOAD, "                d[i] = make_float2((float)(hs + wsrc) * 1e-3f, (float)(c + n) * 1e-3f);")
    elif kind == "nt_store":
        s = s.replace(IN_STORE, "\n".join("            __builtin_nontemporal_store(pl[jj][%d], reinterpret_cast<unsigned*>(o + %d));" % (p, 1024 * p) for p in range(3)))
    elif kind == "remap":           # the four 8-tile groups of a 32-row fragment unit run back to back on ONE XCD
        s = s.replace(IN_MAP, "    const int chunk = (q >> 2) % nchunks, tg = (((q >> 2) / nchunks) * 8 + (L & 7)) * 4 + (q & 3);")
        s = s.replace(IN_BLOCKS, "        const long long blocks = 8ll * ((TG + 31) / 32) * 4 * nchunks;")
    elif kind == "remapc":          # ... chunk-major inside the XCD: a unit's four pieces are written further apart, the input line halves closer
        s = s.replace(IN_MAP, "    const int chunk = q % nchunks, tg = (((q / nchunks) >> 2) * 8 + (L & 7)) * 4 + ((q / nchunks) & 3);")
        s = s.replace(IN_BLOCKS, "        const long long blocks = 8ll * ((TG + 31) / 32) * 4 * nchunks;")
    elif kind == "pad":             # plane pitch off the power-of-two multiples
        s = s.replace(IN_PLANE, IN_PLANE.replace(";", " + %d;" % PAD_BYTES))
    elif "+" in kind:               # combinations: apply the parts in turn
        for part in kind.split("+"):
            s = in_variant(part, s)
    elif kind.startswith("lb"):     # min waves per SIMD -> VGPR cap 512 / n
        s = s.replace("__global__ void __launch_bounds__((MO + 2) * 64) winograd_input_split2_kernel", f"__global__ void __launch_bounds__((MO + 2) * 64, {kind[2:]}) winograd_input_split2_kernel")
    elif kind != "full":
        raise ValueError(kind)
    assert kind == "full" or s != SRC
    return s


OUT_LOAD = "            for (int i = 0; i < AL; ++i) v[i] = *reinterpret_cast<const float2*>(m + (size_t)(i * AL + j) * plane);"
OUT_STORE = "                    *reinterpret_cast<float2*>(y + pix * ldy + c) = val;"


def out_variant(kind):
    s = SRC
    assert OUT_LOAD in s and OUT_STORE in s, "csrc/winograd.hip changed: update the patch anchors"
    if kind == "no_store":
        s = s.replace(OUT_STORE, "                    if (val.x == 1.2345e33f) *reinterpret_cast<float2*>(y + pix * ldy + c) = val;")
    elif kind == "no_load":
        s = s.replace(OUT_LOAD, "            for (int i = 0; i < AL; ++i) v[i] = make_float2((float)(i + j + c) * 1e-3f, (float)(tw + th) * 1e-3f);")
    elif kind == "nt_load":
        s = s.replace(OUT_LOAD, "            for (int i = 0; i < AL; ++i) { const float* q = m + (size_t)(i * AL + j) * plane; "
                                "v[i] = make_float2(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1)); }")
    elif kind == "no_stats":
        assert s.count("    const bool stats = st.s[0] != nullptr || st.s[1] != nullptr;") >= 2
        s = s.replace("    const bool stats = st.s[0] != nullptr || st.s[1] != nullptr;", "    const bool stats = false;")
    elif kind == "st_noflush":      # statistics: per-thread sums + LDS atomics, no global atomics
        assert s.count("        stat_flush(lsum, st, n0, N);") >= 2
        s = s.replace("        stat_flush(lsum, st, n0, N);", "        if (lsum[threadIdx.x] == 1.2345e300) stat_flush(lsum, st, n0, N);")
    elif kind == "st_noadd":        # statistics: per-thread sums only
        s = s.replace("        if (stats) stat_add(lsum, st, n - n0, n, c, psum, psq);", "        if (stats && psum == 1.2345e300 && psq == 1.0) stat_add(lsum, st, n - n0, n, c, psum, psq);")
    elif kind == "st_f32row":       # statistics: the 6 pixels of an output row summed in fp32, rows in fp64
        a = """                    if (stats) {
                        psum += (double)val.x + (double)val.y;
                        psq += (double)val.x * val.x + (double)val.y * val.y;
                    }
                }
            }
        }
        if (stats) stat_add(lsum, st, n - n0, n, c, psum, psq);"""
        assert a in s
        s = s.replace(a, """                    if (stats) {
                        rsum += val.x + val.y;
                        rsq = fmaf(val.x, val.x, fmaf(val.y, val.y, rsq));
                    }
                }
            }
            psum += (double)rsum; psq += (double)rsq;
        }
        if (stats) stat_add(lsum, st, n - n0, n, c, psum, psq);""")
        b = "            float2 rv[MO];\n            if (RES) {"
        assert b in s
        s = s.replace(b, "            float rsum = 0.f, rsq = 0.f;\n" + b)
    elif kind == "pad":
        assert s.count("Tp * (size_t)Cm, Cm, bias") == 2
        s = s.replace("Tp * (size_t)Cm, Cm, bias", "Tp * (size_t)Cm + %d, Cm, bias" % (PAD_BYTES // 4))
    elif kind.startswith("lb"):
        s = s.replace("__global__ void __launch_bounds__(256) winograd_output6_kernel", f"__global__ void __launch_bounds__(256, {kind[2:]}) winograd_output6_kernel")
    elif kind != "full":
        raise ValueError(kind)
    assert kind == "full" or s != SRC
    return s


def build(name, text):
    d = f"/tmp/winov/{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/winograd.hip", "w").write(text)
    others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if os.path.basename(o) != "winograd.o"]
    so = f"{d}/lib.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I", CSRC, "-c",
                           f"{d}/winograd.hip", "-o", f"{d}/winograd.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", f"{d}/winograd.o"] + others + ["-o", so])
    return ctypes.CDLL(so)


def _time(fn, reps):
    assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default=None, help="comma-separated variant names to run (default: all)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    P_, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    shapes = [(16, 64, 64, 1024), (16, 256, 256, 128), (16, 128, 128, 512), (16, 128, 128, 1024)]
    m, P = 6, 64
    sides = (("input (writes 3 bf16 planes)", ("full", "nt_store", "nt_store+pad", "nt_store+remap", "nt_store+pad+remap", "pad+remap"), in_variant),
             ("output (+ residual, + GroupNorm statistics)", ("full", "no_store", "no_load", "nt_load", "no_stats", "st_noflush", "st_noadd", "st_f32row", "pad"), out_variant))
    only = args.only.split(",") if args.only else None
    from concurrent.futures import ThreadPoolExecutor
    jobs = [(("in_" if side.startswith("input") else "out_") + kind, variant(kind)) for side, kinds, variant in sides for kind in kinds
            if not only or kind in only]
    with ThreadPoolExecutor(len(jobs)) as ex:            # hipcc runs side by side (~15 s each)
        libs = dict(zip((j[0] for j in jobs), ex.map(lambda j: build(*j), jobs)))
    for side, kinds, variant in sides:
        print(side)
        for kind in kinds:
            if only and kind not in only:
                continue
            lib = libs[("in_" if side.startswith("input") else "out_") + kind]
            lib.bbdm_winograd_tiles.restype = ctypes.c_size_t
            lib.bbdm_winograd_input_bf3p_f32.argtypes = [I, P_, I, P_, P_, P_, I, I, I, I, I, I, I, P_]
            lib.bbdm_winograd_output_stats_f32.argtypes = [I, P_, P_, P_, I, P_, I, I, I, I, I, I, P_, I, I, P_, I, I, P_]
            line = f"  {kind:9s}"
            for N, H, W, C in shapes:
                tiles = lib.bbdm_winograd_tiles(m, N, H, W)
                x = torch.randn(N, H, W, C, device=dev)
                if side.startswith("input"):
                    sc, bi = torch.rand(N, C, device=dev) + 0.5, torch.randn(N, C, device=dev) * 0.1
                    Vp = torch.empty(P * (tiles * C * 6 + PAD_BYTES), dtype=torch.uint8, device=dev)
                    fn = lambda: lib.bbdm_winograd_input_bf3p_f32(m, x.data_ptr(), C, Vp.data_ptr(), sc.data_ptr(), bi.data_ptr(), C, 1, 0,
                                                                  N, H, W, C, st)
                    gb = (x.numel() * 4 + P * tiles * C * 6) / 1e9
                else:
                    M = torch.randn(P * (tiles * C + PAD_BYTES // 4), device=dev)
                    b = torch.randn(C, device=dev)
                    y = torch.empty_like(x)
                    stats = torch.zeros(N * 64 * 4, dtype=torch.int64, device=dev)      # exact limb accumulator (csrc/stats_acc.h)
                    fn = lambda: lib.bbdm_winograd_output_stats_f32(m, M.data_ptr(), b.data_ptr(), x.data_ptr(), C, y.data_ptr(), C, 0, N, H,
                                                                    W, C, stats.data_ptr(), max(4, C // 32), 0, None, 0, 0, st)
                    gb = (P * tiles * C * 4 + 2 * x.numel() * 4) / 1e9
                ms = _time(fn, args.reps)
                line += f" | N{N} {H}x{W} C{C}: {ms:6.3f} ms ({gb / ms:5.2f} TB/s of the full kernel's bytes)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
