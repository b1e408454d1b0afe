import math, sys, os
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops
dev = torch.device("cuda:0")
def run(B,H,W,Cin,Cout,k,s,p,nstage):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin*k*k)).bfloat16().float()
    ref = F.conv2d(x, w, None, s, p)
    xd = x.permute(0,2,3,1).contiguous().to(dev, torch.bfloat16)
    wd = w.permute(0,2,3,1).contiguous().to(dev, torch.bfloat16)
    y = torch.empty(B, ref.shape[2], ref.shape[3], Cout, device=dev, dtype=torch.bfloat16)
    rc = _lib.load().nopesac_conv2d_nhwc_bfrag(xd.data_ptr(), ops._frag_weights(wd).data_ptr(), None, None, None, y.data_ptr(), B,H,W,Cin,Cout,k,k,s,p,Cin,Cout,0,0,1,nstage, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    yy = y.float().permute(0,3,1,2).cpu()
    err = (yy-ref).abs().max().item()/ref.abs().max().item()
    # per output-channel-tile and per row tile error
    e = (yy-ref).abs()
    print((B,H,W,Cin,Cout,k,s,p,nstage), "rc", rc, "rel err %.4f" % err, "by n-tile:", [round(e[:, i*32:(i+1)*32].max().item(),3) for i in range(Cout//32)][:8])
for rep in range(3):
    for nst in (3,4):
        run(2,30,40,256,256,3,1,1,nst)
        run(8,30,40,256,256,3,1,1,nst)
        run(64,30,40,256,256,3,1,1,nst)
