"""Does the bf16-pipe accumulation round or truncate?  All-positive operands expose a bias."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pg-alignments-gan_amd"))
import torch
from pgan_hip import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
torch.manual_seed(0)
for (Cs, Cl, hs) in [(128, 64, 16), (512, 256, 4)]:
    N = 16
    for kind in ("randn", "positive", "wide"):
        x = torch.randn(N, Cl, 2*hs, 2*hs, device=dev); w = torch.randn(Cs, Cl, 4, 4, device=dev) * 0.05
        if kind == "positive": x, w = x.abs(), w.abs()
        if kind == "wide": x = x * torch.exp(3 * torch.randn_like(x)); w = w * torch.exp(3 * torch.randn_like(w))
        ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=1)
        for prec in ("fp32", "x3"):
            ops.precision = prec
            y = torch.empty(N, Cs, hs, hs, device=dev); ops.conv_down(x, w, y)
            d = (y.double() - ref)
            print(f"down {Cs}x{Cl} {kind:8s} {prec:4s}: rel {float(d.norm()/ref.norm()):.2e}  mean signed err / mean |ref| {float(d.mean()/ref.abs().mean()):+.2e}")
        yy = torch.randn(N, Cs, hs, hs, device=dev)
        if kind == "positive": yy = yy.abs()
        if kind == "wide": yy = yy * torch.exp(3 * torch.randn_like(yy))
        refu = torch.nn.functional.conv_transpose2d(yy.double(), w.double(), stride=2, padding=1)
        for prec in ("fp32", "x3"):
            ops.precision = prec
            xo = torch.empty(N, Cl, 2*hs, 2*hs, device=dev); ops.conv_up(yy, w, xo)
            d = (xo.double() - refu)
            print(f"up   {Cs}x{Cl} {kind:8s} {prec:4s}: rel {float(d.norm()/refu.norm()):.2e}  mean signed err / mean |ref| {float(d.mean()/refu.abs().mean()):+.2e}")
