"""Third neighbour matrix (real kernels of the normal library as neighbours; see pk_neighbours2.py for the set-up).

Second neighbour matrix for the packed-fp32 fault.  Kernel under test: the single-channel DOWN kernel of a packed-fp32
build (tools/_pk/libpgan_hip_pk.so, called through its C entry point), launched on one stream while a second stream runs
 * the normal library's (no packed instructions) 16-bit-pipe DOWN kernel -- does the neighbour need packed code itself? --
 * synthetic neighbours of tools/dbg/pk_nb.hip that keep one kind of unit busy: matrix pipe from registers, matrix pipe
   fed from LDS, LDS alone, matrix pipe fed by 16-byte global loads, a 12-accumulator matrix loop, plain fp32 FMAs.
Every output is compared bit for bit with the standalone result of the same packed kernel."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch

from pgan_hip import HipOps

dev = torch.device("cuda:0")
o = HipOps(dev)                                            # the normal library
pk = ctypes.CDLL(os.path.join(ROOT, "tools", "_pk", "libpgan_hip_pk.so"))
nbl = ctypes.CDLL(os.path.join(ROOT, "tools", "_pk", "libpk_nb.so"))
pk.pgan_build_info.restype = ctypes.c_char_p
print("kernel under test from a library built as:", pk.pgan_build_info().decode())
VP, I = ctypes.c_void_p, ctypes.c_int
pk.pgan_conv_down.argtypes = [VP, VP, VP, I, I, I, I, VP]
nbl.pk_nb_launch.argtypes = [I, VP, VP, I, I, VP]
g = torch.Generator(device="cpu").manual_seed(5)
N, REPS = 64, int(os.environ.get("REPS", "100"))


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


x, w1 = rnd(N, 1, 64, 64), rnd(64, 1, 4, 4, scale=0.25)
want = torch.empty(N, 64, 32, 32, device=dev)
outs = [torch.empty_like(want) for _ in range(8)]
main = torch.cuda.current_stream().cuda_stream


def victim(out):
    rc = pk.pgan_conv_down(x.data_ptr(), w1.data_ptr(), out.data_ptr(), N, 64, 1, 32, main)
    assert rc == 0, rc


victim(want)
torch.cuda.synchronize()
side = torch.cuda.Stream()
o.reserve_stream(side.cuda_stream)
a2, wt2, y2 = rnd(128, 64, 32, 32), rnd(128, 64, 4, 4, scale=0.03), torch.empty(128, 128, 16, 16, device=dev)
src, sink = rnd(1 << 20), torch.zeros(16, device=dev)
with torch.cuda.stream(side):
    o.conv_down(a2, wt2, y2)
torch.cuda.synchronize()


def on_side(fn, n=4):
    def run():
        with torch.cuda.stream(side):
            for _ in range(n):
                fn()
    return run


o32 = HipOps(dev, precision="fp32")
ox3 = HipOps(dev, precision="x3")
for b_ in (o32, ox3):
    b_.reserve_stream(side.cuda_stream)
a3, wt3, y3 = rnd(256, 128, 16, 16), rnd(256, 128, 4, 4, scale=0.02), torch.empty(256, 256, 8, 8, device=dev)
a4, wt4, y4 = rnd(512, 256, 8, 8), rnd(512, 256, 4, 4, scale=0.02), torch.empty(512, 512, 4, 4, device=dev)
dw2 = torch.empty_like(wt2)
xb, yb, dwb = rnd(128, 1, 64, 64), rnd(128, 64, 32, 32), torch.empty(64, 1, 4, 4, device=dev)
pflat, gflat, m_, v_ = rnd(1 << 22), rnd(1 << 22), torch.zeros(1 << 22, device=dev), torch.zeros(1 << 22, device=dev)
stepdev = torch.zeros(1, dtype=torch.int32, device=dev)
for fn in (lambda: o.conv_down(a3, wt3, y3), lambda: o.conv_down(a4, wt4, y4), lambda: ox3.conv_down(a2, wt2, y2)):
    with torch.cuda.stream(side):
        fn()
torch.cuda.synchronize()
neighbours = {
    "nothing": None,
    "fp16-split DOWN 64->128, hs 16, 128 samples": on_side(lambda: o.conv_down(a2, wt2, y2)),
    "fp16-split DOWN 128->256, hs 8, 256 samples": on_side(lambda: o.conv_down(a3, wt3, y3)),
    "fp16-split DOWN 256->512, hs 4, 512 samples": on_side(lambda: o.conv_down(a4, wt4, y4)),
    "bf16-split DOWN 64->128, hs 16": on_side(lambda: ox3.conv_down(a2, wt2, y2)),
    "fp32-pipe DOWN 64->128, hs 16": on_side(lambda: o32.conv_down(a2, wt2, y2)),
    "fp32-pipe UP 128->64, hs 16": on_side(lambda: o32.conv_up(y2, wt2, a2)),
    "fp32-pipe WGRAD 64x128, hs 16": on_side(lambda: o32.conv_wgrad(y2, a2, dw2)),
    "single-channel UP (64 -> 1, 128 samples)": on_side(lambda: o.conv_up(yb, w1, xb)),
    "single-channel WGRAD (128 samples)": on_side(lambda: o.conv_wgrad(yb, xb, dwb)),
    "Adam over 4 M parameters": on_side(lambda: o.adam(pflat, gflat, m_, v_, 1, 1e-4, 0.5, 0.999, 1e-8, stepdev)),
}
for name, nb in neighbours.items():
    bad_launches = bad_values = 0
    rows = torch.zeros(32, dtype=torch.long)
    cols = torch.zeros(32, dtype=torch.long)
    for rep in range(REPS):
        if nb is not None:
            nb()
        for out in outs:
            victim(out)
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
    line = f"beside {name:42s}: {bad_launches:5d} of {REPS * len(outs)} launches differ, {bad_values} values"
    if bad_values:
        line += f"; rows mod 8 {[int(rows[i::8].sum()) for i in range(8)]}; columns mod 4 {[int(cols[i::4].sum()) for i in range(4)]}"
    print(line, flush=True)
