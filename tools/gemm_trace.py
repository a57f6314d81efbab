#!/usr/bin/env python
"""Per-workgroup timeline of one Winograd tile-GEMM launch (needs a libbbdm_hip.so built with the bbdm_debug_conv_trace hook)."""
import ctypes
import os
import sys
from collections import defaultdict

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]      # kernel_ops lives with the tests
from bbdm_amd import _lib
import kernel_ops as ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.bbdm_debug_conv_trace.argtypes = [ctypes.c_void_p]
    raw.bbdm_debug_conv_trace.restype = None
    st = torch.cuda.current_stream().cuda_stream
    m = 4
    for N, H, W, Cin, Cout in ((16, 64, 64, 1024, 1024), (16, 128, 128, 512, 512)):
        P = (m + 2) ** 2
        tiles = lib.bbdm_winograd_tiles(m, N, H, W)
        nblk = (tiles // 256) * ((Cout + 127) // 128) * P
        V = torch.randn(P * tiles * Cin, device=dev)
        M = torch.empty(P * tiles * Cout, device=dev)
        pw = ops.pack_winograd_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02, m=m)
        call = lambda: _lib.call("bbdm_winograd_gemm_f32", m, V.data_ptr(), pw.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
        call(); call()
        torch.cuda.synchronize()
        buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        raw.bbdm_debug_conv_trace(ctypes.c_void_p(buf.data_ptr()))
        call()
        torch.cuda.synchronize()
        raw.bbdm_debug_conv_trace(None)
        t = buf.view(nblk, 8).cpu()
        t0, t1, t2, t3 = (t[:, i].double() for i in range(4))
        base = float(t0.min())
        us = lambda x: x / 100.0          # 100 MHz ticks -> us
        print(f"== N{N} {H}x{W} {Cin}->{Cout}: {nblk} workgroups; launch span {us(float(t3.max()) - base):.1f} us")
        print(f"   setup+prologue {us((t1 - t0).mean()):.2f} us (max {us((t1 - t0).max()):.2f}), main loop {us((t2 - t1).mean()):.2f} us "
              f"(min {us((t2 - t1).min()):.2f} max {us((t2 - t1).max()):.2f}), epilogue {us((t3 - t2).mean()):.2f} us (max {us((t3 - t2).max()):.2f})")
        # group by CU: xcc id + (se, sh, cu) bits of HW_ID
        cu = defaultdict(list)
        for i in range(nblk):
            hw, xcc = int(t[i, 4]), int(t[i, 5]) & 0xF
            key = (xcc, (hw >> 8) & 0xFF)          # cu_id[11:8], sh_id[12], se_id[15:13]
            cu[key].append((float(t0[i]), float(t3[i])))
        gaps, conc = [], []
        for key, lst in cu.items():
            lst.sort()
            ends = sorted(e for _, e in lst)
            starts = [s for s, _ in lst]
            # each start after the first two slots follows some end: match k-th start with (k-2)-th end
            for k in range(2, len(starts)):
                gaps.append(starts[k] - ends[k - 2])
            conc.append(len(lst))
        gaps = torch.tensor(gaps)
        print(f"   {len(cu)} CUs seen, workgroups per CU min {min(conc)} max {max(conc)}; "
              f"slot turnaround (end of a workgroup -> start of its successor on that CU): mean {us(gaps.mean()):.2f} us, "
              f"p50 {us(gaps.median()):.2f}, p95 {us(gaps.quantile(0.95)):.2f}")
        first = sorted(float(x) - base for x in t0.tolist())
        print(f"   first-wave start skew: 256th start {us(first[255]):.1f} us, 512th start {us(first[511]):.1f} us; "
              f"last workgroup starts {us(first[-1]):.1f} us, last 512 ends spread "
              f"{us(float(t3.max()) - sorted(t3.tolist())[-512]):.1f} us")
        del V, M, pw


if __name__ == "__main__":
    main()
