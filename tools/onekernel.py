"""Launch one split-operand convolution shape a few times, as the engine launches it (fp16 split, operand bounds from
the producer), for rocprofv3 --pmc passes.  usage: onekernel.py down|up|wgrad N hs [reps] [precision=x3|h2] [bounds=0]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pg-alignments-gan_amd"))
import torch
from pgan_hip import HipOps, set_x3_option

kind, N, hs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
prec, bounds = "h2", True
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    if k == "precision":
        prec = v
    elif k == "bounds":
        bounds = v != "0"
    else:
        set_x3_option(k, int(v))
Cs, Cl = {32: (64, 32), 16: (128, 64), 8: (256, 128), 4: (512, 256)}[hs]
dev = torch.device("cuda:0")
ops = HipOps(dev, precision=prec)
ops.static_weights = True
y = torch.randn(N, Cs, hs, hs, device=dev)
x = torch.randn(N, Cl, 2 * hs, 2 * hs, device=dev)
w = torch.randn(Cs, Cl, 4, 4, device=dev) * 0.05
dw = torch.zeros_like(w)
amax = lambda t: t.reshape(t.shape[0], 8, -1).abs().amax(2).contiguous() if bounds else None
bx, by = amax(x), amax(y)
for _ in range(reps):
    if kind == "down":
        ops.conv_down(x, w, y, bound=bx)
    elif kind == "up":
        ops.conv_up(y, w, x, bound=by)
    else:
        ops.conv_wgrad(y, x, dw, bound_s=by, bound_l=bx)
torch.cuda.synchronize()
