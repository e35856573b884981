"""Which neighbour makes the packed-fp32 build of the single-channel DOWN kernel return wrong values?

The kernel under test (conv_down, Cl = 1, 64 x 64 inputs, 64 samples) runs on one stream, a neighbour kernel of one class
at a time on a second stream, both enqueued back to back many times; every output of the kernel under test is compared bit
for bit with its standalone result.  Neighbour classes: the 16-bit-pipe DOWN / UP / WGRAD convolutions of the larger
stages (matrix pipe + LDS + 16-byte weight-fragment loads), GroupNorm forward (memory streaming), a second copy of the
kernel under test, a plain device-to-device copy.  Run with PGAN_HIP_LIBRARY pointing at a packed-fp32 build and
PGAN_PACKED_FP32=1; with the normal library every line reads 0.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch

from pgan_hip import HipOps

dev = torch.device("cuda:0")
o = HipOps(dev)
g = torch.Generator(device="cpu").manual_seed(5)
N = 64
REPS = int(os.environ.get("REPS", "300"))


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


x = rnd(N, 1, 64, 64)
w1 = rnd(64, 1, 4, 4, scale=0.25)
want = torch.empty(N, 64, 32, 32, device=dev)
o.conv_down(x, w1, want)
torch.cuda.synchronize()
outs = [torch.empty_like(want) for _ in range(8)]

side = torch.cuda.Stream()
o.reserve_stream(side.cuda_stream)
# operands of the neighbours (stage 2 and stage 4 of the discriminator at 128 samples)
a2, wt2, y2 = rnd(128, 64, 32, 32), rnd(128, 64, 4, 4, scale=0.03), torch.empty(128, 128, 16, 16, device=dev)
a4, wt4, y4 = rnd(128, 256, 8, 8), rnd(512, 256, 4, 4, scale=0.02), torch.empty(128, 512, 4, 4, device=dev)
dw2 = torch.empty_like(wt2)
gam, bet = torch.ones(64, device=dev), torch.zeros(64, device=dev)
h1, st1 = torch.empty(128, 64, 32, 32, device=dev), torch.empty(128, 2, device=dev)
big_a, big_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
x_b, out_b = rnd(N, 1, 64, 64), torch.empty_like(want)
for fn in (lambda: o.conv_down(a2, wt2, y2), lambda: o.conv_down(a4, wt4, y4)):      # weight planes exist before timing
    fn()
torch.cuda.synchronize()

neighbours = {
    "nothing": None,
    "DOWN 64->128 at hs 16 (16-bit pipe)": lambda: o.conv_down(a2, wt2, y2),
    "DOWN 256->512 at hs 4 (16-bit pipe, K split)": lambda: o.conv_down(a4, wt4, y4),
    "UP 128->64 at hs 16 (16-bit pipe)": lambda: o.conv_up(y2, wt2, a2),
    "WGRAD 64x128 at hs 16 (16-bit pipe)": lambda: o.conv_wgrad(y2, a2, dw2),
    "GroupNorm forward, 128 x 64 x 32 x 32": lambda: o.gn_fwd(h1, gam, bet, h1, st1),
    "the same single-channel DOWN": lambda: o.conv_down(x_b, w1, out_b),
    "device-to-device copy, 256 MB": lambda: big_b.copy_(big_a),
}
print(f"single-channel DOWN, {N} samples, {REPS} x {len(outs)} launches per neighbour; library: {os.environ.get('PGAN_HIP_LIBRARY', 'default')}")
for name, nb in neighbours.items():
    bad_launches = bad_values = 0
    rows = torch.zeros(32, dtype=torch.long)
    cols = torch.zeros(32, dtype=torch.long)
    for rep in range(REPS):
        if nb is not None:
            with torch.cuda.stream(side):
                for _ in range(4):
                    nb()
        for out in outs:
            o.conv_down(x, w1, out)
        torch.cuda.synchronize()
        for out in outs:
            d = out != want
            n = int(d.sum())
            if n:
                bad_launches += 1
                bad_values += n
                nz = torch.nonzero(d).cpu()
                rows += torch.bincount(nz[:, 2], minlength=32)
                cols += torch.bincount(nz[:, 3], minlength=32)
    line = f"beside {name:46s}: {bad_launches:5d} of {REPS * len(outs)} launches differ, {bad_values} values"
    if bad_values:
        line += f"; rows mod 8 {[int(rows[i::8].sum()) for i in range(8)]}; columns mod 4 {[int(cols[i::4].sum()) for i in range(4)]}"
    print(line, flush=True)
