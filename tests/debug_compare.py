"""Diagnostic (not a test): run the engine with the HIP backend on the GPU and with the plain-torch
backend on the CPU on the same inputs and print the relative error of every workspace buffer."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pg-alignments-gan_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import dcwgan_oracle as orc
from helpers import rel_err
from pgan_engine import Engine
from pgan_hip import HipOps
from torch_ops import TorchOps

def zd(seed, b):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal((b, 128, 1, 1)).astype(np.float32))

def main(loss="gp", act="tanh", B=8, seed=1234, mode="critic"):
    dev = torch.device("cuda:0")
    eh = Engine(HipOps(dev), B, dev, loss=loss, act=act)
    ec = Engine(TorchOps(), B, torch.device("cpu"), loss=loss, act=act)
    gp_ = orc.make_params(orc.gen_param_shapes(), seed); dp = orc.make_params(orc.disc_param_shapes(), seed + 1)
    for e in (eh, ec):
        e.G.load(gp_); e.D.load(dp)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    if act == "tanh":
        rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
    z = zd(seed + 2, B)
    t = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 4)).random((B, 1, 1, 1))).float()
    if mode == "critic":
        eh.critic_grads(rs.to(dev), rp.to(dev), z.to(dev), t.to(dev)); ec.critic_grads(rs, rp, z, t)
    else:
        eh.generator_grads(z.to(dev)); ec.generator_grads(z)
    torch.cuda.synchronize()
    B_ = B
    def cmp(name, a, b):
        if isinstance(a, list):
            for i, (x, y) in enumerate(zip(a, b)):
                if x is not None:
                    cmp(f"{name}[{i}]", x, y)
            return
        if a.shape[0] % B_ == 0 and a.shape[0] >= 2 * B_ and a.dim() > 1:
            errs = " ".join(f"{rel_err(a[i*B_:(i+1)*B_], b[i*B_:(i+1)*B_]):.1e}" for i in range(a.shape[0] // B_))
        else:
            errs = f"{rel_err(a, b):.1e}"
        print(f"{name:12s} {errs}")
    names = ["X0", "P0", "dA", "dH", "dStat", "dU", "dJ", "dStatB", "dStatT", "dPG", "dPB", "G0", "d1A", "F1", "score", "pen", "closs",
             "gA", "gH", "gStat", "gdH", "gdA", "gXs", "gXp", "gdX", "hA", "hH", "hdH", "hdA", "hdX", "dP0", "gloss"]
    for n in names:
        cmp(n, getattr(eh, n), getattr(ec, n))
    for k in range(1, 5):
        a, b = eh.dH[k].cpu(), ec.dH[k]
        mism = ((a > 0) != (b > 0)).flatten(1).sum(1)
        print(f"mask mismatches dH[{k}] per sample:", mism.tolist())
        e = [f"{rel_err(eh.dJ[k][i], ec.dJ[k][i]):.0e}" for i in range(a.shape[0])]
        print(f"dJ[{k}] per-sample err:", " ".join(e))
        idx = ((a > 0) != (b > 0)).nonzero()
        for r in idx[:6]:
            print("   at", r.tolist(), "hip", float(a[tuple(r)]), "cpu", float(b[tuple(r)]))
    for k in range(1, 5):
        a, b = eh.gH[k].cpu(), ec.gH[k]
        print(f"mask mismatches gH[{k}]:", int(((a > 0) != (b > 0)).sum()))
    st_h, st_c = (eh.D, ec.D) if mode == "critic" else (eh.G, ec.G)
    for k in st_h.g:
        print(f"grad {k:32s} {rel_err(st_h.g[k], st_c.g[k]):.1e}")

if __name__ == "__main__":
    main(*sys.argv[1:3], mode=sys.argv[3] if len(sys.argv) > 3 else "critic")
