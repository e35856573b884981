"""Timing of the DOWN bf16-pipe kernel with one ingredient removed at a time (results are wrong by design)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pg-alignments-gan_amd"))
import torch
from pgan_hip import HipOps, set_x3_option
dev = torch.device("cuda:0"); ops = HipOps(dev); ops.static_weights = True
N, Cs, Cl, hs = 192, 128, 64, 16
x = torch.randn(N, Cl, 2*hs, 2*hs, device=dev); w = torch.randn(Cs, Cl, 4, 4, device=dev) * 0.05
y = torch.empty(N, Cs, hs, hs, device=dev)
fl = 2.0 * N * Cs * Cl * 16 * hs * hs
def timeit(fn, reps=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
set_x3_option("down_variant", 2); set_x3_option("down_ks", 1)
names = {0: "full", 1: "no global loads in loop", 2: "no staging stores/split", 3: "no fragment reads", 4: "no MFMA", 5: "no barrier"}
for ab in (0, 1, 2, 3, 4, 5, 0):
    set_x3_option("ablate", ab)
    us = timeit(lambda: ops.conv_down(x, w, y))
    print(f"ablate {ab} ({names[ab]:28s}): {us:7.1f} us  {fl/us/1e6:6.1f} TF-eq")
