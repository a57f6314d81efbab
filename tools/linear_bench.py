import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import kernel_ops as ops
from bbdm_amd import _lib
dev = torch.device("cuda")
lib = _lib.load()
for (N, In, Out) in ((32, 512, 25088), (4, 512, 25088), (32, 512, 512), (32, 128, 512), (4, 512, 512)):
    x = torch.randn(N, In, device=dev); w = torch.randn(Out, In, device=dev) * 0.05; b = torch.randn(Out, device=dev)
    y = torch.empty(N, Out, device=dev)
    wp = torch.empty(lib.bbdm_linear_packed_bytes(Out, In), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("bbdm_linear_pack_f32", w.data_ptr(), wp.data_ptr(), Out, In, st)
    # several weight copies so that nothing is cache resident
    ws = [w.clone() for _ in range(6)]; wps = [wp.clone() for _ in range(6)]
    def t(fn):
        for i in range(6): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60): fn(i % 6)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 60 * 1000
    a = t(lambda i: _lib.call("bbdm_linear_f32", x.data_ptr(), ws[i].data_ptr(), b.data_ptr(), y.data_ptr(), N, In, Out, 1, 0, st))
    p = t(lambda i: _lib.call("bbdm_linear_packed_f32", x.data_ptr(), wps[i].data_ptr(), b.data_ptr(), y.data_ptr(), N, In, Out, 1, 0, st))
    print(f"N={N} In={In} Out={Out}: linear_f32 {a:.1f} us, packed {p:.1f} us")
