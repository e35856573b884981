"""Host enqueue time of one eager iteration against its GPU time (run on the GPU box from the repo root)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pg-alignments-gan_amd")]
import torch
from pgan_hip import HipOps
from pgan_trainer import FusedTrainer
from bench import synthetic_pool

dev = torch.device("cuda", 0)
pg = None
torch.cuda.set_device(dev)
if os.environ.get("PGAN_FORCE_DIST"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    pg = dist.group.WORLD
ops = HipOps(dev)
if os.environ.get("PGAN_PROXY"):
    from bench import RecordingOps
    ops = RecordingOps(ops)
tr = FusedTrainer(ops, 64, dev, process_group=pg)
if os.environ.get("PGAN_NO_EARLY"):
    tr.eng.grad_ready = None
import numpy as np
from pgan_engine import disc_param_shapes, gen_param_shapes
gen = torch.Generator().manual_seed(20260)
for store, shapes in ((tr.eng.G, gen_param_shapes()), (tr.eng.D, disc_param_shapes())):
    for k, sh in shapes.items():
        if len(sh) == 1:
            v = torch.ones(sh) if k.endswith("weight") and not k.startswith("out") else torch.zeros(sh)
        else:
            v = (torch.rand(sh, generator=gen) * 2 - 1) / int(np.prod(sh[1:])) ** 0.5
        store.p[k].copy_(v)
tr.weights_changed()
ps, pp = synthetic_pool(1024, 1)
tr.set_pool(ps, pp)
for _ in range(5):
    tr.iteration()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.iteration()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/iter, total {1e3 * (t2 - t0) / n:.3f} ms/iter")
if pg is not None:
    x = tr.eng.D.gflat
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(x, group=pg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"all_reduce alone: host {1e6 * (t1 - t0) / 50:.1f} us, total {1e6 * (t2 - t0) / 50:.1f} us")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.iteration()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
