#!/usr/bin/env python
"""Probe: is gemm_bf3_kernel bound by the latency of its fp32 A feed?  Same GEMM shape per workgroup (K = Cin), A either
small enough to live in L2 / Infinity Cache (few rows, re-read by many Cout tiles) or streamed from HBM."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for pixels, Cin, Cout in [(4096, 512, 4096), (16384, 512, 1024), (65536, 512, 512), (262144, 512, 512), (1048576, 512, 512),
                          (4096, 1024, 4096), (65536, 1024, 1024), (262144, 1024, 1024)]:
    x = torch.randn(pixels, Cin, device=dev)
    w = torch.randn(Cout, Cin, device=dev) * 0.05
    pf = torch.empty(lib.bbdm_conv_packed_floats(Cout, Cin, 1), device=dev)
    _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), pf.data_ptr(), Cout, Cin, Cin, 1, st)
    pk = torch.empty(lib.bbdm_gemm_bf3_packed_halfs(1, Cin, Cout), dtype=torch.int16, device=dev)
    _lib.call("bbdm_gemm_bf3_pack_f32", pf.data_ptr(), pk.data_ptr(), 1, Cin, Cout, st)
    out = torch.empty(pixels, Cout, device=dev)
    call = lambda: _lib.call("bbdm_conv1x1_bf3_f32", x.data_ptr(), Cin, pk.data_ptr(), None, None, 0, out.data_ptr(), Cout,
                             pixels, Cin, Cout, st)
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * pixels * Cin * Cout
    print(f"pixels {pixels:8d} {Cin}->{Cout}: A {pixels * Cin * 4 / 1e6:7.1f} MB, {pixels // 256 * ((Cout + 127) // 128):6d} workgroups, "
          f"{ms:7.3f} ms, {fl / ms / 1e9:6.1f} TFLOP/s fp32-eq", flush=True)
    del x, w, pf, pk, out
