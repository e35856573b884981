"""Which buffers of a critic + generator step depend on what was in memory before?  Runs the step twice on engines whose
torch.empty buffers start as NaN / as a large finite value and lists every engine tensor that differs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch

import dcwgan_oracle as orc
from pgan_engine import Engine
from pgan_hip import HipOps

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "8"))
loss = os.environ.get("LOSS", "gp")


def poison(val):
    big = torch.full((768 << 18,), val, device=dev)
    small = [torch.full((1 << 16,), val, device=dev) for _ in range(512)]
    tiny = [torch.full((256,), val, device=dev) for _ in range(2048)]
    torch.cuda.synchronize()
    del big, small, tiny


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
    elif hasattr(obj, "__dict__") and type(obj).__module__.startswith("pgan"):
        for k, v in vars(obj).items():
            tensors(v, f"{prefix}.{k}", out, seen)


def run_trainer(val):
    from pgan_trainer import FusedTrainer
    poison(val)
    tr = FusedTrainer(HipOps(dev), B, dev, loss=loss, critic_iter=2, seed=77)
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    tr.weights_changed()
    rs, rp = orc.synthetic_alignments(64, 9)
    tr.set_pool(rs, rp)
    snap = {}
    for it in range(int(os.environ.get("ITERS", "2"))):
        tr.iteration()
        torch.cuda.synchronize()
        out = {}
        tensors(tr, "tr", out, set())
        snap.update({f"it{it}:" + k: v.clone() for k, v in out.items()})
    return snap


def run(val):
    if os.environ.get("TRAINER", "1") == "1":
        return run_trainer(val)
    poison(val)
    ops = HipOps(dev)
    eng = Engine(ops, B, dev, loss=loss)
    eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    rs, rp = orc.synthetic_alignments(B, 60)
    g = np.random.Generator(np.random.PCG64(62))
    t = torch.from_numpy(g.random((B, 1, 1, 1))).float()
    z = torch.from_numpy(g.standard_normal((B, 128, 1, 1))).float()
    eng.critic_grads(rs.to(dev), rp.to(dev), z.to(dev), t.to(dev) if loss == "gp" else None)
    torch.cuda.synchronize()
    out = {}
    tensors(eng, "eng", out, set())
    snap = {k: v.clone() for k, v in out.items()}
    eng.generator_grads(z.to(dev))
    torch.cuda.synchronize()
    snap.update({"G:" + k: v.clone() for k, v in out.items() if k.startswith("eng.G.") or k.startswith("eng.g")})
    return snap


a = run(float("nan"))
b = run(3.0e4)
c = run(float("nan"))
for name, x, y in (("nan vs 3e4", a, b), ("nan vs nan", a, c)):
    print(name)
    for k in a:
        u, v = x[k], y[k]
        if u.shape != v.shape:
            continue
        same = torch.equal(torch.nan_to_num(u, nan=-7.0), torch.nan_to_num(v, nan=-7.0))
        if not same:
            m = torch.isfinite(u) & torch.isfinite(v)
            d = float((u[m] - v[m]).abs().max()) if bool(m.any()) else float("nan")
            print(f"  {k:50s} shape {tuple(u.shape)} finite-diff {d:.3e} nonfinite {int((~torch.isfinite(u)).sum())}/{int((~torch.isfinite(v)).sum())}")
