"""Host logic of the fused step (buffer layout, op order, seeds, analytic double backward) checked
on the CPU: the engine runs on the plain-torch op backend from tests/torch_ops.py and must
reproduce the autograd oracle.  The HIP kernels themselves are covered by the gpu-marked tests."""
import numpy as np
import pytest
import torch

import dcwgan_oracle as orc
from helpers import rel_err
from pgan_engine import Engine
from torch_ops import TorchOps


def z_draw(seed, b):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal((b, orc.LATENT, 1, 1)).astype(np.float32))


def make_engine(B, loss, act, seed, data=64):
    eng = Engine(TorchOps(), B, torch.device("cpu"), loss=loss, act=act, data=data)
    gp_ = orc.make_params(orc.gen_param_shapes(data=data), seed)
    dp = orc.make_params(orc.disc_param_shapes(data), seed + 1)
    eng.G.load(gp_)
    eng.D.load(dp)
    return eng, gp_, dp


@pytest.mark.parametrize("loss", ["gp", "div"])
def test_critic_grads_match_autograd(loss):
    B, seed = 3, 11
    eng, gp_, dp = make_engine(B, loss, "sigmoid", seed)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    z = z_draw(seed + 2, B)
    t = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 4)).random((B, 1, 1, 1))).float()
    gbuf = orc.make_gen_buffers()
    w, grads = orc.critic_grads(dp, gp_, gbuf, rs, rp, z, loss, 10.0, t, "sigmoid")
    eng.critic_grads(rs, rp, z, t)
    assert abs(float(eng.closs[0]) - float(w)) <= 1e-5 * max(1.0, abs(float(w)))
    for k, g in grads.items():
        e = rel_err(eng.D.g[k], g) if float(g.norm()) > 0 else float(eng.D.g[k].abs().max())
        assert e < 2e-5, (k, e)
    for k, v in gbuf.items():
        if "running" in k:
            assert rel_err(eng.Gbuf.p[k], v) < 1e-6, k


@pytest.mark.parametrize("act", ["sigmoid", "tanh"])
def test_generator_grads_match_autograd(act):
    B, seed = 3, 21
    eng, gp_, dp = make_engine(B, "gp", act, seed)
    z = z_draw(seed + 2, B)
    loss, grads = orc.generator_grads(dp, gp_, orc.make_gen_buffers(), z, act)
    eng.generator_grads(z)
    assert abs(float(eng.gloss) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    for k, g in grads.items():
        assert rel_err(eng.G.g[k], g) < 2e-5, k


def test_train_iteration_matches_oracle():
    B, seed = 2, 31
    eng, gp_, dp = make_engine(B, "gp", "sigmoid", seed)
    gbuf = orc.make_gen_buffers()
    d_opt, g_opt = orc.AdamState(dp), orc.AdamState(gp_)
    rs, rp = orc.synthetic_alignments(B * 2, seed + 3)
    batches = [(rs[:B], rp[:B]), (rs[B:], rp[B:])]
    zs = [z_draw(seed + 10 + i, B) for i in range(3)]
    ts = [torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 20 + i)).random((B, 1, 1, 1))).float()
          for i in range(2)]
    orc.train_iteration(dp, gp_, gbuf, d_opt, g_opt, batches, zs, ts, "gp", 10.0, "sigmoid")
    eng.train_iteration(batches, zs, ts)
    for k, v in dp.items():
        assert rel_err(eng.D.p[k], v) < 1e-5, k
    for k, v in gp_.items():
        assert rel_err(eng.G.p[k], v) < 1e-5, k
    assert eng.g_batches_tracked == 3


@pytest.mark.parametrize("loss", ["gp", "div"])
def test_scaled_network_128_matches_autograd(loss):
    """The 128 x 128 networks (one more stride-2 stage; build-defined, the reference has none): critic and generator
    gradients of the engine's host logic against the oracle's autograd on the same layer tables."""
    B, seed, S = 2, 41, 128
    eng, gp_, dp = make_engine(B, loss, "tanh", seed, data=S)
    assert eng.nS == 5 and eng.nG == 5 and eng.feat1 == 64 * 119
    rs, rp = orc.synthetic_alignments(B, seed + 3, size=S)
    rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
    z = z_draw(seed + 2, B)
    t = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 4)).random((B, 1, 1, 1))).float()
    gbuf = orc.make_gen_buffers(data=S)
    w, grads = orc.critic_grads(dp, gp_, gbuf, rs, rp, z, loss, 10.0, t, "tanh")
    eng.critic_grads(rs, rp, z, t)
    assert abs(float(eng.closs[0]) - float(w)) <= 1e-5 * max(1.0, abs(float(w)))
    for k, g in grads.items():
        e = rel_err(eng.D.g[k], g) if float(g.norm()) > 0 else float(eng.D.g[k].abs().max())
        assert e < 2e-5, (k, e)
    for k, v in gbuf.items():
        if "running" in k:
            assert rel_err(eng.Gbuf.p[k], v) < 1e-6, k
    gl, ggrads = orc.generator_grads(dp, gp_, None, z, "tanh")
    eng.generator_grads(z)
    assert abs(float(eng.gloss) - float(gl)) <= 1e-5 * max(1.0, abs(float(gl)))
    for k, g in ggrads.items():
        assert rel_err(eng.G.g[k], g) < 2e-5, k
