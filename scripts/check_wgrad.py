"""Correctness + speed of hc_conv_wgrad on a list of shapes against torch's CPU convolution gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from holocron_amd.ops import conv as cv

shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ.get(
    "SHAPES", "4,192,192,14,1;4,192,1280,14,2;2,1280,1280,7,1;2,256,256,20,1;2,256,512,19,2;2,512,256,10,1;3,128,128,9,1;2,192,384,11,1").split(";")]
g = torch.Generator().manual_seed(0)
for (N, Cin, Cout, H, s) in shapes:
    for k in (3, 1):
        pad = k // 2
        x = torch.randn((N, Cin, H, H), generator=g).to(torch.bfloat16).float()
        OH = (H + 2 * pad - k) // s + 1
        dy = torch.randn((N, Cout, OH, OH), generator=g).to(torch.bfloat16).float()
        w = torch.zeros((Cout, Cin, k, k), requires_grad=True)
        (F.conv2d(x, w, None, s, pad) * dy).sum().backward()
        got = cv.conv_wgrad(cv.to_cl_bf16(x.cuda()), cv.to_cl_bf16(dy.cuda()), Cin, Cout, k, k, s, pad).cpu()
        rel = float((got - w.grad).norm() / w.grad.norm())
        print(f"N{N} {Cin}->{Cout} @{H} s{s} k{k}: rel {rel:.2e}", "OK" if rel < 2e-3 else "BAD", flush=True)
