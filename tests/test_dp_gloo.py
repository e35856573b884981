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
        torch.save({"g": tr.eng.D.gflat.clone(), "loss": loss_part / world}, os.path.join(out_dir, "dp.pt"))
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
