"""The reference's own training-loop statements (src/train_wgan_v2.py:534-634) executed on the
drop-in modules on the GPU, compared with the CPU oracle."""
import numpy as np
import pytest
import torch

import dcwgan_oracle as orc
from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def z_draw(seed, b):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal((b, 128, 1, 1)).astype(np.float32))


def build(act, seed):
    import layers
    gp_ = orc.make_params(orc.gen_param_shapes(), seed)
    dp = orc.make_params(orc.disc_param_shapes(), seed + 1)
    G = (layers.DC_Generator_Leaky_Sig if act == "sigmoid" else layers.DC_Generator_Leaky_Tan)(64, 128, 0.1, 1, k_size=4)
    D = layers.DC_Discriminator(64, 0.1, 1, k_size=4)
    G.load_state_dict({**gp_, **orc.make_gen_buffers()})
    D.load_state_dict(dp)
    return layers, G.to(DEV), D.to(DEV), gp_, dp


# direct fp32-vs-fp64 gradient comparison (no shared LeakyReLU region): 5e-3, see test_hip_engine
GTOL = 5e-3


@pytest.mark.parametrize("loss", ["gp", "div"])
def test_reference_critic_statements(loss, monkeypatch):
    layers, G, D, gp_, dp = build("sigmoid", 310)
    B = 8
    rs, rp = orc.synthetic_alignments(B, 313)
    z = z_draw(312, B)
    t = np.random.Generator(np.random.PCG64(314)).random((B, 1, 1, 1))
    monkeypatch.setattr(np.random, "random", lambda shape: t)
    d_opt = torch.optim.Adam(D.parameters(), lr=5e-5, betas=(0.5, 0.999))
    X_real_sites, X_real_pos = rs.to(DEV), rp.to(DEV)
    X_fake_sites, X_fake_pos = G(z.to(DEV))
    if loss == "div":
        X_real_sites.requires_grad_(True)
        X_fake_sites = X_fake_sites.detach().requires_grad_(True)
        X_fake_pos = X_fake_pos.detach().requires_grad_(True)
    d_opt.zero_grad()
    d_loss = D(torch.cat([X_real_sites, X_fake_sites]), torch.cat([X_real_pos, X_fake_pos]))
    if loss == "gp":
        w = d_loss[B:].mean() - d_loss[:B].mean() + layers.gp_gradient_penalty(
            D, X_real_sites, X_real_pos, X_fake_sites, X_fake_pos, 10, DEV)
    else:
        w = d_loss[B:].mean() - d_loss[:B].mean() + layers.div_gradient_penalty(
            d_loss[:B], d_loss[B:], X_real_sites, X_real_pos, X_fake_sites, X_fake_pos, DEV)
    w.backward()
    want_w, want = orc.critic_grads({k: v.double() for k, v in dp.items()}, {k: v.double() for k, v in gp_.items()},
                                    None, rs.double(), rp.double(), z.double(), loss, 10.0,
                                    torch.from_numpy(t), "sigmoid")
    assert abs(float(w) - float(want_w)) <= 2e-5 * max(1.0, abs(float(want_w)))
    for k, p in D.named_parameters():
        if float(want[k].norm()) == 0:
            continue
        assert rel_err(p.grad, want[k]) < GTOL, k
    before = {k: p.detach().clone() for k, p in D.named_parameters()}
    d_opt.step()
    moved = max(float((p.detach() - before[k]).abs().max()) for k, p in D.named_parameters())
    assert 4e-5 < moved < 6e-5          # first Adam step = lr * sign(grad)


@pytest.mark.parametrize("act", ["sigmoid", "tanh"])
def test_reference_generator_statements(act):
    layers, G, D, gp_, dp = build(act, 320)
    B = 8
    z = z_draw(322, B)
    g_opt = torch.optim.Adam(G.parameters(), lr=5e-5, betas=(0.5, 0.999))
    g_opt.zero_grad()
    xs, xp = G(z.to(DEV))
    gen_loss = -1 * D(xs, xp).mean()
    gen_loss.backward()
    g_opt.step()
    want_l, want = orc.generator_grads({k: v.double() for k, v in dp.items()}, {k: v.double() for k, v in gp_.items()},
                                       None, z.double(), act)
    assert abs(float(gen_loss) - float(want_l)) <= 2e-5 * max(1.0, abs(float(want_l)))
    for k, p in G.named_parameters():
        assert rel_err(p.grad, want[k]) < GTOL, k
    sd = G.state_dict()
    assert int(sd["twod_conv_layers.1.num_batches_tracked"]) == 1
    gbuf = orc.make_gen_buffers()
    orc.gen_forward(gp_, gbuf, z, act)
    assert rel_err(sd["twod_conv_layers.4.running_var"], gbuf["twod_conv_layers.4.running_var"]) < 1e-5
    with pytest.raises(RuntimeError):
        G(z)      # CPU input: no fallback
