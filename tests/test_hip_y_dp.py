"""Data-parallel path on the GPU kernels: two processes, each with half of the batch -- over RCCL (backend "nccl") on
two GPUs where the box has them, and sharing the one GPU of a single-GPU box with gloo moving the CUDA tensors.
(Collected after the other GPU tests.)  Reduced gradients, losses and the generator's running statistics equal one
process at the whole batch: the critic has no batch coupling, the generator's BatchNorm layers reduce their per-channel
sums over the ranks (src/layers.py:93-108,119-134).  A third test runs the trainer's own iteration on one rank over
RCCL (collectives inside the launch tape) against the same iteration without a process group."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dcwgan_oracle as orc

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
B = 8
SEED = int(os.environ.get("PGAN_DP_SEED", "60"))


def _inputs(seed):
    rs, rp = orc.synthetic_alignments(B, seed)
    g = np.random.Generator(np.random.PCG64(seed + 2))
    t = torch.from_numpy(g.random((B, 1, 1, 1))).float()
    z = torch.from_numpy(g.standard_normal((B, 128, 1, 1))).float()
    return rs, rp, t, z


def _run(tr, rs, rp, t, z, sl, dev):
    """One critic step's gradients (fakes from this rank's z rows), then one generator step's; both reduced."""
    e = tr.eng
    e.critic_grads(rs[sl].to(dev), rp[sl].to(dev), z[sl].to(dev), t[sl].to(dev) if e.loss == "gp" else None)
    tr._allreduce(e.D.gflat)        # leaves the sum over the ranks; Adam reads it times Engine.grad_scale = 1 / ranks
    out = {"d": (e.D.gflat * e.grad_scale).cpu(), "closs": e.closs.clone().cpu()}
    e.generator_grads(z[sl].flip(0).contiguous().to(dev))
    tr._allreduce(e.G.gflat)
    out.update(g=(e.G.gflat * e.grad_scale).cpu(), gloss=e.gloss.clone().cpu().reshape(-1),
               buf={k: v.clone().cpu() for k, v in e.Gbuf.p.items()})
    return out


def _load(tr):
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    tr.weights_changed()


def _worker(rank, world, port, loss, out_dir, backend="gloo"):
    for p in (os.path.join(ROOT, "pg-alignments-gan_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    from pgan_hip import HipOps
    from pgan_trainer import FusedTrainer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    from helpers import poison_gpu_cache
    poison_gpu_cache(dev, 256)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = FusedTrainer(HipOps(dev), B // world, dev, loss=loss, process_group=dist.group.WORLD, rank=rank, world=world)
    assert tr.sync_bn
    _load(tr)
    rs, rp, t, z = _inputs(SEED)
    # rank r holds the global batch's rows [r B/2, (r+1) B/2); generator_grads sees them flipped within the rank
    out = _run(tr, rs, rp, t, z, slice(rank * B // world, (rank + 1) * B // world), dev)
    for k in ("closs", "gloss"):
        v = out[k].to(dev) if backend == "nccl" else out[k].clone()
        dist.all_reduce(v)
        out[k] = (v / world).cpu()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save(out, os.path.join(out_dir, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("loss", ["gp", "div"])
def test_two_ranks_equal_one_process_at_the_whole_batch(tmp_path, loss, backend):
    from pgan_hip import HipOps
    from pgan_trainer import FusedTrainer
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL between two ranks needs two GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, loss, str(tmp_path), backend), nprocs=2, join=True)
    got = torch.load(tmp_path / "dp.pt", weights_only=True)
    dev = torch.device("cuda", 0)
    tr = FusedTrainer(HipOps(dev), B, dev, loss=loss)
    _load(tr)
    rs, rp, t, z = _inputs(SEED)
    e = tr.eng
    e.critic_grads(rs.to(dev), rp.to(dev), z.to(dev), t.to(dev) if loss == "gp" else None)
    want_d, want_closs = e.D.gflat.clone().cpu(), e.closs.clone().cpu()
    zg = torch.cat([z[:B // 2].flip(0), z[B // 2:].flip(0)]).contiguous()
    e.generator_grads(zg.to(dev))
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    def per_tensor(store, a, b):       # which parameters disagree, for the failure message
        out = {}
        for k, off in store.offsets.items():
            n = store.p[k].numel()
            r = float((a[off:off + n] - b[off:off + n]).norm() / max(float(b[off:off + n].norm()), 1e-30))
            if r > 1e-6:
                out[k] = r
        return out
    assert rel(got["d"], want_d) < 1e-5, (rel(got["d"], want_d), per_tensor(e.D, got["d"], want_d))
    assert torch.allclose(got["closs"], want_closs, rtol=2e-5, atol=1e-6)
    assert rel(got["g"], e.G.gflat.cpu()) < 1e-5, rel(got["g"], e.G.gflat.cpu())
    assert torch.allclose(got["gloss"], e.gloss.cpu().reshape(-1), rtol=2e-5, atol=1e-6)
    for k, v in e.Gbuf.p.items():
        assert torch.allclose(got["buf"][k], v.cpu(), rtol=2e-5, atol=1e-6), k



def _one_rank_worker(rank, port, out_dir):
    for p in (os.path.join(ROOT, "pg-alignments-gan_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    from pgan_hip import HipOps
    from pgan_trainer import FusedTrainer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rs, rp = orc.synthetic_alignments(64, 9)
    out = {}
    for dp in (False, True):
        tr = FusedTrainer(HipOps(dev), B, dev, critic_iter=2, seed=77)
        if dp:      # after the trainer, as the CLI and the benchmark do
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            tr.attach_process_group(dist.group.WORLD)
        _load(tr)
        tr.set_pool(rs, rp)
        tr.use_tape = True
        for _ in range(3):
            tr.step()
        torch.cuda.synchronize()
        out[dp] = dict(d=tr.eng.D.flat.clone().cpu(), g=tr.eng.G.flat.clone().cpu(), losses=list(tr.losses()),
                       buf={k: v.clone().cpu() for k, v in tr.eng.Gbuf.p.items()})
    torch.save(out, os.path.join(out_dir, "one.pt"))
    dist.destroy_process_group()


def test_trainer_iteration_over_rccl_on_one_rank(tmp_path):
    """The trainer's whole iteration with every collective of the data-parallel path issued over RCCL (one rank:
    gradient all-reduces with the per-layer early reductions of both networks, BatchNorm partial-sum all-reduces,
    all recorded on and replayed from the launch tape) trains like the same iteration without a process group."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_one_rank_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    out = torch.load(tmp_path / "one.pt", weights_only=True)
    a, b = out[False], out[True]
    lr, steps = 5e-5, 3 * 2
    for k in ("d", "g"):
        assert float((a[k] - b[k]).abs().max()) <= 2 * lr * steps, k
        assert float((a[k] - b[k]).norm() / a[k].norm()) < 1e-3, k
    for x, y in zip(a["losses"], b["losses"]):
        assert abs(x - y) <= 1e-2 * max(1.0, abs(x))


def _split_trainer(FusedTrainer, ops, dev, loss, batch, rank, world, pg):
    tr = FusedTrainer(ops, batch, dev, loss=loss, critic_iter=2, seed=99, process_group=pg, rank=rank, world=world)
    _load(tr)
    rs, rp = orc.synthetic_alignments(24, 70)
    tr.set_pool(rs, rp)
    tr.enable_augment(label_noise=0.05, mono_switch=(0, 10))
    tr.shuffle_pool(rows=True)
    return tr


def _split_worker(rank, world, port, loss, steps, out_dir, backend):
    for p in (os.path.join(ROOT, "pg-alignments-gan_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    from pgan_hip import HipOps
    from pgan_trainer import FusedTrainer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = _split_trainer(FusedTrainer, HipOps(dev), dev, loss, B // world, rank, world, dist.group.WORLD)
    assert tr.global_stream and tr.sync_bn
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    e = tr.eng
    torch.save({"g": e.G.flat.cpu(), "d": e.D.flat.cpu(), "Zw": e.Zw.cpu(), "t": e.tbuf.cpu(), "pool": tr.pool_s.cpu(),
                "cursor": tr.cursor.cpu(), "rng": torch.stack([tr.rng_z, tr.rng_t, tr.rng_aug, tr.rng_mono]).cpu(),
                "buf": {k: v.cpu() for k, v in e.Gbuf.p.items()}}, os.path.join(out_dir, f"split{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("loss", ["gp", "div"])
def test_two_ranks_split_one_run_sample_for_sample(tmp_path, loss, backend):
    """FusedTrainer.step() on two ranks against one process at the whole batch, through the HIP kernels: the ranks share
    pool order, cursor and generator streams and take their rows of every batch and draw (z, t, label-smoothing noise,
    monomorphic-site draws) -- bit-identical inputs -- with BatchNorm statistics over the global batch; after three
    iterations the weights agree within the band stated below.  The one process takes its BatchNorm statistics through the same sums / apply launches as the
    ranks (a no-op reduction), so both sides round them alike."""
    from pgan_hip import HipOps
    from pgan_trainer import FusedTrainer
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL between two ranks needs two GPUs")
    steps = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_split_worker, args=(2, port, loss, steps, str(tmp_path), backend), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / f"split{r}.pt", weights_only=True) for r in (0, 1))
    assert torch.equal(a["g"], b["g"]) and torch.equal(a["d"], b["d"])
    dev = torch.device("cuda", 0)
    one = _split_trainer(FusedTrainer, HipOps(dev), dev, loss, B, 0, 1, None)
    one.eng.bn_sync = lambda part: None
    start = {"g": one.eng.G.flat.cpu(), "d": one.eng.D.flat.cpu()}
    for _ in range(steps):
        one.step()
    torch.cuda.synchronize()
    e, h = one.eng, B // 2
    rows = lambda t, per: torch.cat([x.reshape(-1, h, per) for x in t], dim=1)
    assert torch.equal(rows((a["Zw"], b["Zw"]), 128), e.Zw.cpu().reshape(-1, B, 128))
    if loss == "gp":
        assert torch.equal(rows((a["t"], b["t"]), 1), e.tbuf.cpu().reshape(1, B, 1))
    want_rng = torch.stack([one.rng_z, one.rng_t, one.rng_aug, one.rng_mono]).cpu()
    for r in (a, b):
        assert torch.equal(r["pool"], one.pool_s.cpu()) and torch.equal(r["cursor"], one.cursor.cpu())
        assert torch.equal(r["rng"], want_rng)
    # Weights: the ranks' launches hold other samples than the one process's (the fp16 split scales a block's operands by
    # its own samples' magnitudes; reductions add in another order), so values differ in the last bits and, over three
    # iterations of 8 samples, now and then a pre-activation lands on the other side of a LeakyReLU kink.  Bounds: every
    # weight within two Adam steps per update, 1e-3 of the norm, and a small part of the distance travelled.
    lr = 5e-5
    for key, ref, updates in (("g", e.G.flat.cpu(), steps), ("d", e.D.flat.cpu(), 2 * steps)):
        err = float((a[key] - ref).norm() / ref.norm())
        moved = float((ref - start[key]).norm() / ref.norm())
        print(f"{loss} {backend} {key}: |ranks - one process| / |w| = {err:.2e}, distance travelled {moved:.2e}")
        assert float((a[key] - ref).abs().max()) <= 2 * lr * updates, key
        assert moved > 1e-3 and err <= 1e-3 and err <= 0.1 * moved, (key, err, moved)
    for k, v in e.Gbuf.p.items():       # running statistics of generators whose weights differ as bounded above
        assert float((a["buf"][k] - v.cpu()).norm() / v.cpu().norm()) < 2e-3, k


@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_runs_under_the_launcher_with_several_ranks(ranks, tmp_path):
    """bench.py as the driver launches it for N > 1 (python -m torch.distributed.run, one process per rank): with one rank
    over RCCL (PGAN_FORCE_DIST) and with two ranks that share this box's GPU over gloo (--dist-backend gloo), the run ends
    with rank 0's one JSON line -- the multi-rank control flow (route timing agreed between the ranks, barriers, maximum
    over the ranks, the per-shape timing after the timed region beside a live process group) is the part no other test
    reaches."""
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-fp32-pipe", "--profile-iters", "3"]
    if ranks == 1:
        env["PGAN_FORCE_DIST"] = "1"
    else:
        cmd += ["--dist-backend", "gloo"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["config"]["global_batch"] == 64 * ranks and out["scaling"] == "weak"
    assert all(np.isfinite(v) for v in out["losses"].values())


def test_command_line_trainer_under_the_launcher_with_two_ranks(tmp_path):
    """train_wgan_v2.py as `python -m torch.distributed.run --nproc-per-node 2` runs it (the two ranks share this box's GPU;
    collectives over gloo): --batch_size stays the global batch, every rank loads the whole pool and takes its rows of
    every global batch, rank 0 writes the checkpoints, and the replicas' weights are the same numbers at the end."""
    import ast
    import subprocess
    from test_hip_cli import _write_sims
    idir, odir = tmp_path / "sims", tmp_path / "out"
    idir.mkdir()
    _write_sims(str(idir), 96)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PGAN_DIST_BACKEND="gloo", PGAN_DUMP_WEIGHT_SUMS=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "pg-alignments-gan_amd", "train_wgan_v2.py"), "--idir", str(idir),
           "--odir", str(odir), "--use_cuda", "--num_in", "96", "--batch_size", "16", "--epochs", "2", "--save_freq", "1",
           "--critic_iter", "2", "--label_smooth", "--mono_switch"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    dumps = [[ast.literal_eval(l) for l in open(tmp_path / f"weights_rank{r}.txt")] for r in (0, 1)]
    # the weights; the generator's running statistics differ by design: only rank 0 makes the 1000-sample plotting forward
    # before a save (src/train_wgan_v2.py:686-698), which steps them, and training never reads them
    assert dumps[0][0] == dumps[1][0], dumps
    assert all(np.isfinite(v) for d_ in dumps for row in d_ for v in row)
    g = torch.load(odir / "generator_model_0002.pt", weights_only=True)
    d = torch.load(odir / "discriminator_model_0002.pt", weights_only=True)
    assert all(bool(torch.isfinite(v.float()).all()) for v in list(g.values()) + list(d.values()))
    # 96 alignments / (global batch 16) / 2 critic steps = 3 iterations per epoch, 3 generator passes each, + 2 plotting forwards
    assert int(g["twod_conv_layers.1.num_batches_tracked"]) == 2 * 3 * 3 + 2
