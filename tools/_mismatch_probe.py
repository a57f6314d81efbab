import os, sys, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib
import kernel_ops as ops
dev = torch.device("cuda:0"); lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
m, P = 6, 64
for (N, H, W, Cin, Cout) in [(16, 128, 128, 1024, 1024), (16, 64, 64, 512, 512), (16, 256, 256, 256, 128), (16, 64, 64, 1024, 1024)]:
    tiles = lib.bbdm_winograd_tiles(m, N, H, W)
    torch.manual_seed(1)
    x = torch.randn(N, H, W, Cin, device=dev)
    sc = torch.rand(N, Cin, device=dev) + 0.5; bi = torch.randn(N, Cin, device=dev) * 0.1
    pw = ops.pack_winograd_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02, m=m)
    Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
    _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
    Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
    Vf = torch.empty(lib.bbdm_gemm_bf3q_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
    _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, Vp.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin, 1, 0, N, H, W, Cin, st)
    _lib.call("bbdm_winograd_input_bf3q_f32", m, x.data_ptr(), Cin, Vf.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin, 1, 0, N, H, W, Cin, st)
    res = {}
    for name, entry, V in (("p", "bbdm_winograd_gemm_bf3p_f32", Vp), ("q", "bbdm_winograd_gemm_bf3q_f32", Vf)):
        for rep in range(3):
            M = torch.full((P, tiles, Cout), float("nan"), device=dev)
            _lib.call(entry, m, V.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
            torch.cuda.synchronize()
            res[(name, rep)] = M
    def diff(a, b):
        d = (res[a] != res[b]) & ~(torch.isnan(res[a]) & torch.isnan(res[b]))
        n = int(d.sum())
        if not n: return "equal"
        idx = d.nonzero()
        return f"{n} differ; xi {sorted(set(idx[:,0].tolist()))[:6]} rows {int(idx[:,1].min())}..{int(idx[:,1].max())} cols {int(idx[:,2].min())}..{int(idx[:,2].max())} max|d| {float((res[a]-res[b])[d].abs().max()):.3e}"
    print(f"N{N} {H}x{W} {Cin}->{Cout} tiles {tiles}: p0/p1 {diff(('p',0),('p',1))} | p1/p2 {diff(('p',1),('p',2))} | q0/q1 {diff(('q',0),('q',1))} | p0/q0 {diff(('p',0),('q',0))} | nan p {int(torch.isnan(res[('p',0)]).sum())} q {int(torch.isnan(res[('q',0)]).sum())}", flush=True)
    del x, pw, Bp, Vp, Vf, res
