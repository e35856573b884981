"""Two-chain critic step against the single-chain order on the same inputs (run on the GPU box): forward activations
of every slot must agree bit for bit, gradients to rounding."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pg-alignments-gan_amd")]
import torch
import numpy as np
from bench import synthetic_pool
from pgan_engine import disc_param_shapes, gen_param_shapes
from pgan_engine import Engine
from pgan_hip import HipOps
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ops = HipOps(dev)
engs = []
for two in (True, False):
    e = Engine(ops, B, dev)
    e.two_chains = two
    gen = torch.Generator().manual_seed(7)
    for store, shapes in ((e.G, gen_param_shapes()), (e.D, disc_param_shapes())):
        for k, sh in shapes.items():
            v = ((torch.ones(sh) if k.endswith("weight") and not k.startswith("out") else torch.zeros(sh)) if len(sh) == 1
                 else (torch.rand(sh, generator=gen) * 2 - 1) / int(np.prod(sh[1:])) ** 0.5)
            store.p[k].copy_(v)
    engs.append(e)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    rs, rp = synthetic_pool(B, 100 + it)
    g = torch.Generator().manual_seed(it)
    z, t = torch.randn(B, 128, 1, 1, generator=g), torch.rand(B, generator=g)
    for e in engs:
        e.critic_grads(rs.to(dev), rp.to(dev), z.to(dev), t.to(dev))
    torch.cuda.synchronize()
    a, b = engs
    for k in range(0, 5):
        for name in ("dH", "dA"):
            if name == "dA" and k == 0: continue
            x, y = getattr(a, name)[k], getattr(b, name)[k]
            for lo, hi, tag in ((0, 2 * B, "real|fake"), (2 * B, 3 * B, "interp"), (3 * B, 4 * B, "tangent")):
                if not torch.equal(x[lo:hi], y[lo:hi]):
                    d = float((x[lo:hi] - y[lo:hi]).abs().max())
                    print(f"iter {it} {name}[{k}] {tag}: max abs diff {d:.3e}"); bad += 1
    e = float((a.D.gflat - b.D.gflat).norm() / b.D.gflat.norm())
    if e > 2e-6: print(f"iter {it} grad rel diff {e:.2e}"); bad += 1
print("mismatches:", bad)
