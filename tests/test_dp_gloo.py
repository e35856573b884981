"""Data-parallel path on the CPU (gloo, world_size 2): per-rank gradients over a half batch,
all-reduced by FusedTrainer._allreduce, equal the single-process gradient over the whole batch.
The discriminator has no batch coupling (GroupNorm(1,C)), every loss term is a batch mean."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dcwgan_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _inputs(B, seed):
    rs, rp = orc.synthetic_alignments(B, seed)
    fs, fp = orc.synthetic_alignments(B, seed + 1)
    fs = fs * 0.8 + 0.1
    t = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 2)).random((B, 1, 1, 1))).float()
    return rs, rp, fs, fp, t


def _worker(rank, world, port, loss, B, out_dir):
    for p in (os.path.join(ROOT, "pg-alignments-gan_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    from pgan_trainer import FusedTrainer
    from torch_ops import TorchOps
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = FusedTrainer(TorchOps(), B // world, torch.device("cpu"), loss=loss, process_group=dist.group.WORLD,
                      rank=rank, world=world)
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    rs, rp, fs, fp, t = _inputs(B, 40)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    tr.eng.critic_grads_given_fake(rs[sl], rp[sl], fs[sl], fp[sl], t[sl])
    tr._allreduce(tr.eng.D.gflat)
    loss_part = tr.eng.closs.clone()
    dist.all_reduce(loss_part)
    if rank == 0:
        # _allreduce leaves the SUM over the ranks; the Adam step that follows reads it times Engine.grad_scale
        assert tr.eng.grad_scale == 1.0 / world
        torch.save({"g": tr.eng.D.gflat * tr.eng.grad_scale, "loss": loss_part / world}, os.path.join(out_dir, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("loss", ["gp", "div"])
def test_two_rank_gradients_equal_single_process(tmp_path, loss):
    from pgan_engine import Engine
    from torch_ops import TorchOps
    B = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, loss, B, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "dp.pt", weights_only=True)
    eng = Engine(TorchOps(), B, torch.device("cpu"), loss=loss)
    eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    eng.critic_grads_given_fake(*_inputs(B, 40))
    err = float((got["g"] - eng.D.gflat).norm() / eng.D.gflat.norm())
    assert err < 2e-6, err
    assert torch.allclose(got["loss"], eng.closs, rtol=1e-5, atol=1e-6)


def _gen_worker(rank, world, port, B, out_dir):
    for p in (os.path.join(ROOT, "pg-alignments-gan_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    from pgan_trainer import FusedTrainer
    from torch_ops import TorchOps
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = FusedTrainer(TorchOps(), B // world, torch.device("cpu"), process_group=dist.group.WORLD, rank=rank, world=world)
    assert tr.sync_bn
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    z = torch.from_numpy(np.random.Generator(np.random.PCG64(50)).standard_normal((B, 128, 1, 1))).float()
    tr.eng.generator_grads(z[rank * B // world:(rank + 1) * B // world])
    tr._allreduce(tr.eng.G.gflat)
    loss_part = tr.eng.gloss.clone()
    dist.all_reduce(loss_part)
    if rank == 1:     # the other rank holds the same reduced gradients and running statistics
        torch.save({"g": tr.eng.G.gflat * tr.eng.grad_scale, "loss": loss_part / world,
                    "buf": {k: v.clone() for k, v in tr.eng.Gbuf.p.items()}}, os.path.join(out_dir, "dpg.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_generator_step_with_global_batch_statistics(tmp_path):
    """BatchNorm couples the generator's samples (src/layers.py:93-108,119-134): with the per-channel sums reduced
    over the ranks, two ranks of B/2 give the gradients, loss and running statistics of one process at B."""
    from pgan_engine import Engine
    from torch_ops import TorchOps
    B = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gen_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "dpg.pt", weights_only=True)
    eng = Engine(TorchOps(), B, torch.device("cpu"))
    eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    z = torch.from_numpy(np.random.Generator(np.random.PCG64(50)).standard_normal((B, 128, 1, 1))).float()
    eng.generator_grads(z)
    err = float((got["g"] - eng.G.gflat).norm() / eng.G.gflat.norm())
    assert err < 5e-6, err
    assert torch.allclose(got["loss"].reshape(-1), eng.gloss.reshape(-1), rtol=1e-5, atol=1e-6)
    for k, v in eng.Gbuf.p.items():
        assert torch.allclose(got["buf"][k], v, rtol=1e-5, atol=1e-6), k


def _iter_worker(rank, world, port, sync_bn, out_dir):
    for p in (os.path.join(ROOT, "pg-alignments-gan_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    from pgan_trainer import FusedTrainer
    from torch_ops import TorchOps
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = FusedTrainer(TorchOps(), 2, torch.device("cpu"), critic_iter=2, process_group=dist.group.WORLD, rank=rank,
                      world=world, sync_bn=sync_bn, global_stream=False)
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    rs, rp = orc.synthetic_alignments(16, 70)
    tr.set_pool(rs[rank::world], rp[rank::world])
    for _ in range(2):
        tr.step()
    torch.save({"g": tr.eng.G.flat.clone(), "d": tr.eng.D.flat.clone(), "losses": tr.losses(),
                "buf": {k: v.clone() for k, v in tr.eng.Gbuf.p.items()}}, os.path.join(out_dir, f"it{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sync_bn", [True, False])
def test_two_rank_iterations_keep_the_replicas_identical(tmp_path, sync_bn):
    """Two full iterations of FusedTrainer on two ranks (each its own real batches, z and t; the early reduction of
    the critic's largest gradient, generator forwards one step ahead, both Adam steps): the replicas' weights stay
    bit-identical and move; with statistics over the global batch the BatchNorm running statistics agree too, with
    per-rank statistics they differ."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_iter_worker, args=(2, port, sync_bn, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / f"it{r}.pt", weights_only=True) for r in (0, 1))
    assert torch.equal(a["g"], b["g"]) and torch.equal(a["d"], b["d"])
    assert all(np.isfinite(v) for v in a["losses"])
    same = all(torch.equal(a["buf"][k], b["buf"][k]) for k in a["buf"])
    assert same == sync_bn
    d0 = torch.cat([v.reshape(-1) for v in orc.make_params(orc.disc_param_shapes(), 7).values()])
    assert float(a["d"].abs().sum()) != float(d0.abs().sum())


def _split_run_worker(rank, world, port, loss, B, steps, out_dir):
    for p in (os.path.join(ROOT, "pg-alignments-gan_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    from pgan_trainer import FusedTrainer
    from torch_ops import TorchOps
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = _split_run_trainer(FusedTrainer, TorchOps, loss, B // world, rank, world, dist.group.WORLD)
    assert tr.global_stream and tr.sync_bn
    for _ in range(steps):
        tr.step()
    e = tr.eng
    torch.save({"g": e.G.flat.clone(), "d": e.D.flat.clone(), "losses": tr.losses(),
                "buf": {k: v.clone() for k, v in e.Gbuf.p.items()},
                # this rank's latent and interpolation draws of the last steps; the state its next inputs come from
                "Zw": e.Zw.clone(), "t": e.tbuf.clone(), "pool": tr.pool_s.clone(), "cursor": tr.cursor.clone(),
                "rng": torch.stack([tr.rng_z, tr.rng_t, tr.rng_aug, tr.rng_mono])},
               os.path.join(out_dir, f"split{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _split_run_trainer(FusedTrainer, TorchOps, loss, batch, rank, world, pg):
    tr = FusedTrainer(TorchOps(), batch, torch.device("cpu"), loss=loss, critic_iter=2, seed=99, process_group=pg,
                      rank=rank, world=world)
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    rs, rp = orc.synthetic_alignments(24, 70)
    tr.set_pool(rs, rp)                      # every rank holds the whole pool
    tr.enable_augment(label_noise=0.05, mono_switch=(0, 10))
    tr.shuffle_pool(rows=True)
    return tr


@pytest.mark.parametrize("loss", ["gp", "div"])
def test_two_ranks_split_one_run_sample_for_sample(tmp_path, loss, monkeypatch):
    """step() itself on two ranks equals step() of one process at the whole batch: the ranks keep one pool cursor and
    one set of generator streams (z, t, label-smoothing noise, monomorphic-site draws, pool shuffles) and take rows
    [r B/P, (r+1) B/P) of every batch and draw; the generator's BatchNorm statistics run over the global batch.  Three
    iterations of 2 critic steps + 1 generator step.  The inputs of the last steps are bit-identical.  The weights agree
    within a band (1e-3 of their norm, a tenth of the distance travelled) always, and up to the summation order of the
    reductions (1e-5 of their norm after Adam's near-sign first steps, a thousandth of the distance travelled) unless a
    pre-activation within rounding of the LeakyReLU kink took the other slope for one sample in one of the two batch
    splits -- the run is therefore made under both orders of the weight-gradient sums (one pass per stage, two half
    passes), and the tight bound must hold under at least one.  The one-process run takes its BatchNorm statistics
    through the same sums / apply pair as the ranks do (a no-op reduction): the one-launch form rounds the statistics
    differently in the last bit."""
    from pgan_trainer import FusedTrainer
    from torch_ops import TorchOps
    B, steps = 4, 3
    tight_runs = 0
    for split in ("1", "0"):
        monkeypatch.setenv("PGAN_SPLIT_WGRAD", split)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_split_run_worker, args=(2, port, loss, B, steps, str(tmp_path)), nprocs=2, join=True)
        a, b = (torch.load(tmp_path / f"split{r}.pt", weights_only=True) for r in (0, 1))
        assert torch.equal(a["g"], b["g"]) and torch.equal(a["d"], b["d"])
        one = _split_run_trainer(FusedTrainer, TorchOps, loss, B, 0, 1, None)
        assert not one.global_stream and not one.dist_active
        one.eng.bn_sync = lambda part: None
        start = {"g": one.eng.G.flat.clone(), "d": one.eng.D.flat.clone()}
        for _ in range(steps):
            one.step()
        e, h = one.eng, B // 2
        rows = lambda t, per: torch.cat([x.reshape(-1, h, per) for x in t], dim=1)      # ranks' rows side by side
        assert torch.equal(rows((a["Zw"], b["Zw"]), 128), e.Zw.reshape(-1, B, 128))
        if loss == "gp":
            assert torch.equal(rows((a["t"], b["t"]), 1), e.tbuf.reshape(1, B, 1))
        for r in (a, b):      # same shuffled pool, same cursor, same generator states on every rank as in the one process
            assert torch.equal(r["pool"], one.pool_s) and torch.equal(r["cursor"], one.cursor)
            assert torch.equal(r["rng"], torch.stack([one.rng_z, one.rng_t, one.rng_aug, one.rng_mono]))
        tight = True
        for key, got, ref in (("g", a["g"], e.G.flat), ("d", a["d"], e.D.flat)):
            err = float((got - ref).norm() / ref.norm())
            moved = float((ref - start[key]).norm() / ref.norm())
            print(f"{loss}, PGAN_SPLIT_WGRAD={split}, {key}: error {err:.2e} of the norm, weights moved {moved:.2e}")
            assert moved > 1e-3 and err <= 1e-3 and err <= 0.1 * moved, (key, err, moved)
            tight = tight and err <= 1e-5 and err <= 2e-3 * moved
        tight_runs += tight
        for k, v in one.eng.Gbuf.p.items():
            assert torch.allclose(a["buf"][k], v, rtol=2e-3, atol=1e-4), k
        # every loss term is a batch mean: the ranks' values average to the one process's
        assert np.allclose((np.array(a["losses"]) + np.array(b["losses"])) / 2, one.losses(), rtol=2e-3, atol=1e-4)
    assert tight_runs >= 1, "the weights agree to summation order under neither order of the weight-gradient sums"
