"""Twin trainers in lockstep -- one launched eagerly, one through ROUTE (graph | tape) -- compared tensor by tensor
after every iteration; prints the tensors that differ first, with the sample slots that differ (B, CI, LOSS, TRIALS, ITERS)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch

import dcwgan_oracle as orc
import pgan_hip
from pgan_hip import HipOps
from pgan_trainer import FusedTrainer

dev = torch.device("cuda:0")
for item in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = item.split("=")
    pgan_hip.set_x3_option(k, int(v))
loss, B, CI = os.environ.get("LOSS", "gp"), int(os.environ.get("B", "8")), int(os.environ.get("CI", "2"))
rs, rp = orc.synthetic_alignments(64, 9)


ROUTE = os.environ.get("ROUTE", "graph")


def make(serial):
    tr = FusedTrainer(HipOps(dev, precision=os.environ.get("PREC", "h2")), B, dev, loss=loss, critic_iter=CI, seed=77)
    tr.use_graph, tr.use_tape = (not serial) and ROUTE == "graph", (not serial) and ROUTE == "tape"
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    tr.weights_changed()
    tr.set_pool(rs, rp)
    return tr


def tensors(obj, prefix, out, seen):
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda and obj.is_floating_point():
            out[prefix] = obj
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            tensors(v, f"{prefix}[{i}]", out, seen)
    elif isinstance(obj, dict):
        for k, v in obj.items():
            tensors(v, f"{prefix}.{k}", out, seen)
    elif hasattr(obj, "__dict__") and type(obj).__module__.startswith("pgan") and not isinstance(obj, HipOps):
        for k, v in vars(obj).items():
            tensors(v, f"{prefix}.{k}", out, seen)


found = 0
for trial in range(int(os.environ.get("TRIALS", "6"))):
    a, b = make(True), make(False)
    ta, tb = {}, {}
    tensors(a, "tr", ta, set())
    tensors(b, "tr", tb, set())
    for it in range(int(os.environ.get("ITERS", "12"))):
        a.step()
        torch.cuda.synchronize()
        b.step()
        torch.cuda.synchronize()
        bad = []
        for k in ta:
            if ".p." in k or ".m." in k or ".v." in k or k.endswith("flat"):
                continue
            u, v = ta[k], tb[k]
            if u.shape == v.shape and not torch.equal(torch.nan_to_num(u, nan=-7.0), torch.nan_to_num(v, nan=-7.0)):
                d = (torch.nan_to_num(u, nan=-7.0) - torch.nan_to_num(v, nan=-7.0)).abs()
                idx = torch.nonzero(d.reshape(d.shape[0], -1).amax(1) > 0).flatten().tolist() if d.dim() > 1 else []
                bad.append(f"{k.replace('tr.eng.', '')}:{float(d.max()):.1e}:{idx[:16] if len(idx) < 16 else 'many'}")
        if bad:
            print(f"trial {trial} iteration {it}: {len(bad)} tensors differ", flush=True)
            print("  " + " | ".join(bad), flush=True)
            found += 1
            break
    else:
        print(f"trial {trial}: no difference", flush=True)
    del a, b, ta, tb
print("found", found)
