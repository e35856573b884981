"""One critic step on fixed inputs, recorded on a launch tape and replayed INNER times back to back (dense launches, as
in training) before every comparison with the first run's workspace; prints which sample slots of which tensor differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch

import dcwgan_oracle as orc
import pgan_hip
import pgan_tape
from pgan_engine import Engine
from pgan_hip import HipOps

dev = torch.device("cuda:0")
for item in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = item.split("=")
    pgan_hip.set_x3_option(k, int(v))
B = int(os.environ.get("B", "8"))
INNER = int(os.environ.get("INNER", "5"))
eng = Engine(HipOps(dev, precision=os.environ.get("PREC", "h2")), B, dev, loss=os.environ.get("LOSS", "gp"))
eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
rs, rp = orc.synthetic_alignments(2 * B, 60)
g = np.random.Generator(np.random.PCG64(62))
eng.X0[:2 * B].copy_(rs.to(dev))
eng.P0[:2 * B].copy_(rp.to(dev))
eng.tbuf.copy_(torch.from_numpy(g.random(B)).float().to(dev))
names = ["dA", "dH", "dU", "dJ", "dStat", "dStatB", "dStatT", "dHmax", "dJmax", "dPG", "dPB"]


def snap():
    out = {"X0": eng.X0.clone(), "G0": eng.G0.clone(), "score": eng.score.clone(), "closs": eng.closs.clone()}
    for n in names:
        for k, v in enumerate(getattr(eng, n, [])):
            if isinstance(v, torch.Tensor):
                out[f"{n}[{k}]"] = v.clone()
    for k, v in eng.D.g.items():
        out["Dg." + k] = v.clone()
    return out


eng.critic_grads_resident(run_generator=False)
torch.cuda.synchronize()
ref = snap()
pgan_tape.tape = entries = []
eng.critic_grads_resident(run_generator=False)
pgan_tape.tape = None
torch.cuda.synchronize()
print("tape entries", len(entries), flush=True)
bad = 0
for rep in range(int(os.environ.get("REPS", "100"))):
    for _ in range(INNER):
        if os.environ.get("REPLAY", "tape") == "tape":
            pgan_tape.replay(entries)
        else:
            eng.critic_grads_resident(run_generator=False)
    torch.cuda.synchronize()
    cur = snap()
    diff = [k for k in ref if not torch.equal(torch.nan_to_num(ref[k]), torch.nan_to_num(cur[k]))]
    if diff:
        bad += 1
        print(f"rep {rep}: {len(diff)} tensors differ", flush=True)
        for k in diff:
            d = (torch.nan_to_num(ref[k]) - torch.nan_to_num(cur[k])).abs()
            if d.dim() > 1:
                slots = torch.nonzero(d.reshape(d.shape[0], -1).amax(1) > 0).flatten().tolist()
            else:
                slots = torch.nonzero(d).flatten().tolist()[:12]
            print(f"   {k} {tuple(d.shape)} n={int((d > 0).sum())} max={float(d.max()):.1e} slots={slots}", flush=True)
        if bad >= 3:
            break
print("mismatching repetitions:", bad)
