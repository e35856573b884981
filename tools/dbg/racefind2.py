"""Repeat one critic step on the same inputs (four streams) and compare every workspace tensor with the first run's;
for the first mismatch print where inside the tensor the differing elements lie."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch

import dcwgan_oracle as orc
import pgan_hip
from pgan_engine import Engine
from pgan_hip import HipOps

dev = torch.device("cuda:0")
for item in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = item.split("=")
    pgan_hip.set_x3_option(k, int(v))
B = int(os.environ.get("B", "8"))
eng = Engine(HipOps(dev, precision=os.environ.get("PREC", "h2")), B, dev, loss=os.environ.get("LOSS", "gp"))
eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
rs, rp = orc.synthetic_alignments(B, 60)
g = np.random.Generator(np.random.PCG64(62))
t = torch.from_numpy(g.random((B, 1, 1, 1))).float().to(dev)
z = torch.from_numpy(g.standard_normal((B, 128, 1, 1))).float().to(dev)
rs, rp = rs.to(dev), rp.to(dev)
names = ["dA", "dH", "dU", "dJ", "dStat", "dStatB", "dStatT", "dPGB"]


def snap():
    out = {"X0": eng.X0.clone(), "G0": eng.G0.clone(), "Dg": eng.D.gflat.clone()}
    for n in names:
        for k, v in enumerate(getattr(eng, n)):
            if isinstance(v, torch.Tensor):
                out[f"{n}[{k}]"] = v.clone()
    return out


eng.critic_grads(rs, rp, z, t)
torch.cuda.synchronize()
ref = snap()
bad = 0
for rep in range(int(os.environ.get("REPS", "300"))):
    eng.critic_grads(rs, rp, z, t)
    torch.cuda.synchronize()
    cur = snap()
    diff = [k for k in ref if not torch.equal(torch.nan_to_num(ref[k]), torch.nan_to_num(cur[k]))]
    if diff:
        bad += 1
        print(f"rep {rep}: {diff}", flush=True)
        for k in diff[:12]:
            d = (torch.nan_to_num(ref[k]) - torch.nan_to_num(cur[k])).abs()
            nz = torch.nonzero(d)
            lo, hi = nz.min(0).values.tolist(), nz.max(0).values.tolist()
            print(f"   {k} shape {tuple(d.shape)} differing elements {nz.shape[0]} max {float(d.max()):.2e} index range {lo} .. {hi}",
                  flush=True)
        if "dJ[3]" in diff:
            d = (ref["dJ[3]"] - cur["dJ[3]"]).abs()
            nz = torch.nonzero(d).tolist()
            print("   dJ[3] differing (n, c, h, w):", nz[:200], flush=True)
            for n_, c_, h_, w_ in nz[:6]:
                print("     ", (n_, c_, h_, w_), float(ref["dJ[3]"][n_, c_, h_, w_]), float(cur["dJ[3]"][n_, c_, h_, w_]),
                      "u", float(ref["dU[3]"][n_, c_, h_, w_]), float(cur["dU[3]"][n_, c_, h_, w_]),
                      "a", float(ref["dA[3]"][n_, c_, h_, w_]), "h", float(ref["dH[3]"][n_, c_, h_, w_]), flush=True)
        if bad >= 2:
            break
print("mismatching repetitions:", bad)
