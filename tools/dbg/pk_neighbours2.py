"""Second neighbour matrix for the packed-fp32 fault.  Kernel under test: the single-channel DOWN kernel of a packed-fp32
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
src, sink = rnd(1 << 22), torch.zeros(16, device=dev)
with torch.cuda.stream(side):
    o.conv_down(a2, wt2, y2)
torch.cuda.synchronize()


def synthetic(kind, blocks=1024, iters=3000):
    def run():
        rc = nbl.pk_nb_launch(kind, src.data_ptr(), sink.data_ptr(), blocks, iters, side.cuda_stream)
        assert rc == 0, rc
    return run


def real_down():
    with torch.cuda.stream(side):
        for _ in range(4):
            o.conv_down(a2, wt2, y2)


neighbours = {
    "nothing": None,
    "normal library's DOWN 64->128 at hs 16": real_down,
    "matrix pipe, operands in registers": synthetic(0),
    "matrix pipe fed from 32 KB of LDS": synthetic(1),
    "matrix pipe fed from 64 KB of LDS": synthetic(6),
    "LDS reads and writes only": synthetic(2),
    "matrix pipe fed by 16-byte global loads": synthetic(3),
    "matrix pipe, 12 accumulators": synthetic(4, iters=600),
    "fp32 FMA chains only": synthetic(5, iters=40000),
    "fp16 -> fp32 conversions with SDWA (WORD_1)": synthetic(7, iters=40000),
    "packed conversions fp32 pair -> fp16 / bf16": synthetic(8, iters=40000),
    "private (scratch) memory traffic": synthetic(9, iters=4000),
    "SDWA compares": synthetic(10, iters=40000),
    "matrix pipe at full rate on random fp16 bit patterns": synthetic(11, iters=1500),
    "the same, 2048 blocks": synthetic(11, blocks=2048, iters=1500),
    "MFMA triples as in the split kernels (2 dependent + 1), 1 tile": synthetic(12, iters=4000),
    "the same, 2 tiles": synthetic(13, iters=2000),
    "the same, 4 tiles": synthetic(14, iters=1000),
    "short blocks: MFMA triples, 2 tiles, 24 steps, 60000 blocks": synthetic(13, blocks=60000, iters=24),
    "short blocks: MFMA triples, 4 tiles, 8 steps, 60000 blocks": synthetic(14, blocks=60000, iters=8),
    "short blocks: triples, 2 tiles, 24 steps, barriers, stores, 60000 blocks": synthetic(15, blocks=60000, iters=24),
    "short blocks: triples, 4 tiles, 16 steps, barriers, stores, 60000 blocks": synthetic(16, blocks=60000, iters=16),
    "short blocks: triples, 2 tiles, 64 steps, barriers, stores, 20000 blocks": synthetic(15, blocks=20000, iters=64),
    "short blocks: plain MFMA loop, 50 steps, 60000 blocks": synthetic(0, blocks=60000, iters=50),
    "short blocks: fp32 FMA chains, 400 steps, 60000 blocks": synthetic(5, blocks=60000, iters=400),
}
only = os.environ.get("ONLY")
for name, nb in neighbours.items():
    if only and only not in name and name != "nothing":
        continue
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
