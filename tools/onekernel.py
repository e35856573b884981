"""Launch one conv kernel shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pg-alignments-gan_amd"))
import torch
from pgan_hip import HipOps, set_option
kind, N, hs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    if k == "precision": os.environ["PGAN_CONV_PRECISION"] = v
    else: set_option(k, int(v))
Cs, Cl = {16: (128, 64), 8: (256, 128), 4: (512, 256)}[hs]
dev = torch.device("cuda:0"); ops = HipOps(dev); ops.static_weights = True
y = torch.randn(N, Cs, hs, hs, device=dev); x = torch.randn(N, Cl, 2*hs, 2*hs, device=dev)
w = torch.randn(Cs, Cl, 4, 4, device=dev) * 0.05; dw = torch.zeros_like(w)
for _ in range(reps):
    if kind == "down": ops.conv_down(x, w, y)
    elif kind == "up": ops.conv_up(y, w, x)
    else: ops.conv_wgrad(y, x, dw)
torch.cuda.synchronize()
