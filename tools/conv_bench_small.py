import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bbdm_amd import ops
SH=[(32,16,16,1024,1024,3),(32,16,16,2048,1024,3),(32,32,32,512,512,3),(4,16,16,1024,1024,3),(32,4,4,1024,1024,3),(32,8,8,512,512,3),(32,16,16,1024,3072,1)]
dev=torch.device("cuda:0")
for N,H,W,Ci,Co,ks in SH:
    x=torch.randn(N,H,W,Ci,device=dev); w=torch.randn(Co,Ci,ks,ks,device=dev)*0.02; b=torch.randn(Co,device=dev)
    pw=ops.pack_conv_weight(w); out=torch.empty(N,H,W,Co,device=dev)
    ops.conv2d_nhwc(x,pw,b,Co,ks,out=out); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d_nhwc(x,pw,b,Co,ks,out=out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10; fl=2.0*N*H*W*Co*Ci*ks*ks
    print(f"N{N} {H}x{W} {Ci}->{Co} k{ks}: {ms:7.3f} ms {fl/ms/1e9:6.1f} TF")
