"""End-to-end parity of the fused HIP step on the GPU: losses, every parameter gradient (including
the analytic gradient-penalty / W-div terms), Adam-updated weights and BatchNorm running statistics
against the CPU oracle on identical injected inputs, and against the golden vectors the reference's
own modules produced (tests/golden)."""
import numpy as np
import pytest
import torch

import dcwgan_oracle as orc
from helpers import check_summary, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def z_draw(seed, b):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal((b, orc.LATENT, 1, 1)).astype(np.float32))


def t_draw(seed, b):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.random((b, 1, 1, 1))).float()


@pytest.fixture(scope="module")
def hip():
    from pgan_hip import HipOps
    return HipOps(torch.device(DEV))


def make_engine(hip, B, loss, act, seed):
    from pgan_engine import Engine
    eng = Engine(hip, B, torch.device(DEV), loss=loss, act=act)
    gp_ = orc.make_params(orc.gen_param_shapes(), seed)
    dp = orc.make_params(orc.disc_param_shapes(), seed + 1)
    eng.G.load(gp_)
    eng.D.load(dp)
    return eng, gp_, dp


@pytest.mark.parametrize("fname,loss", [("critic_gp_B4.npz", "gp"), ("critic_div_B4.npz", "div"),
                                        ("critic_gp_B64.npz", "gp")])
def test_critic_step_vs_reference_golden(hip, fname, loss):
    g = load_golden(fname)
    B, seed, lam = [int(v) for v in g["meta"]]
    eng, gp_, dp = make_engine(hip, B, loss, "sigmoid", seed)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    z, t = z_draw(seed + 2, B), torch.from_numpy(g["t"])
    eng.critic_grads(rs.to(DEV), rp.to(DEV), z.to(DEV), t.to(DEV))
    torch.cuda.synchronize()
    ref = g["w"]
    got = eng.closs.cpu().double().numpy()
    print("loss", got, ref)
    assert abs(got[0] - ref[0]) <= 2e-5 * max(1.0, abs(ref[0]))
    assert abs(got[3] - ref[1]) <= 2e-5 * max(1.0, abs(ref[1]))
    assert abs(got[1] - ref[2]) <= 2e-5 * max(1.0, abs(ref[2]))
    assert abs(got[2] - ref[3]) <= 2e-5 * max(1.0, abs(ref[3]))
    for k in dp:
        check_summary(g, f"grad.{k}", eng.D.g[k], rtol=2e-5, what=f"{fname}:")


@pytest.mark.parametrize("loss", ["gp", "div"])
def test_critic_grads_vs_oracle(hip, loss):
    B, seed = 8, 1234
    eng, gp_, dp = make_engine(hip, B, loss, "tanh", seed)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
    z, t = z_draw(seed + 2, B), t_draw(seed + 4, B)
    gbuf = orc.make_gen_buffers()
    P64 = {k: v.double() for k, v in dp.items()}
    G64 = {k: v.double() for k, v in gp_.items()}
    w, grads = orc.critic_grads(P64, G64, None, rs.double(), rp.double(), z.double(), loss, 10.0, t.double(), "tanh")
    orc.gen_forward(gp_, gbuf, z, "tanh")
    eng.critic_grads(rs.to(DEV), rp.to(DEV), z.to(DEV), t.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(eng.closs[0]) - float(w)) <= 1e-5 * max(1.0, abs(float(w)))
    for k, gr in grads.items():
        if float(gr.norm()) == 0:
            assert float(eng.D.g[k].abs().max()) < 1e-9, k
            continue
        e = rel_err(eng.D.g[k], gr)
        print(f"{loss} {k}: {e:.2e}")
        assert e < 1e-5, (k, e)
    for k, v in gbuf.items():
        if "running" in k:
            assert rel_err(eng.Gbuf.p[k], v) < 1e-5, k


@pytest.mark.parametrize("act", ["sigmoid", "tanh"])
def test_generator_step_vs_reference_golden(hip, act):
    g = load_golden(f"generator_{act}_B4.npz")
    B, seed = [int(v) for v in g["meta"]]
    eng, gp_, dp = make_engine(hip, B, "gp", act, seed)
    eng.generator_grads(z_draw(seed + 2, B).to(DEV))
    torch.cuda.synchronize()
    assert abs(float(eng.gloss) - g["loss"][0]) <= 2e-5 * max(1.0, abs(g["loss"][0]))
    for k in gp_:
        check_summary(g, f"grad.{k}", eng.G.g[k], rtol=2e-5, what=f"gen {act}:")


def test_generator_grads_vs_oracle_fp64(hip):
    B, seed = 8, 4321
    eng, gp_, dp = make_engine(hip, B, "gp", "sigmoid", seed)
    z = z_draw(seed + 2, B)
    loss, grads = orc.generator_grads({k: v.double() for k, v in dp.items()}, {k: v.double() for k, v in gp_.items()},
                                      None, z.double(), "sigmoid")
    eng.generator_grads(z.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(eng.gloss) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    for k, gr in grads.items():
        e = rel_err(eng.G.g[k], gr)
        print(f"gen {k}: {e:.2e}")
        assert e < 1e-5, (k, e)


@pytest.mark.parametrize("fname,loss,act", [("train_gp_sigmoid_B8.npz", "gp", "sigmoid"),
                                            ("train_gp_tanh_norm_B8.npz", "gp", "tanh"),
                                            ("train_div_sigmoid_B8.npz", "div", "sigmoid")])
def test_training_trajectory_vs_reference_golden(hip, fname, loss, act):
    g = load_golden(fname)
    B, seed, lam, iters, critic_iter, normalize = [int(v) for v in g["meta"]]
    eng, gp_, dp = make_engine(hip, B, loss, act, seed)
    real_s, real_p = orc.synthetic_alignments(B * critic_iter * iters, seed + 3)
    if normalize:
        real_s, real_p = (real_s - 0.5) / 0.5, (real_p - 0.5) / 0.5
    traj, ctr = [], 0
    for it in range(iters):
        batches, zs, ts = [], [], []
        for h in range(critic_iter):
            batches.append((real_s[ctr * B:(ctr + 1) * B].to(DEV), real_p[ctr * B:(ctr + 1) * B].to(DEV)))
            zs.append(z_draw(seed + 1000 + it * 16 + h, B).to(DEV))
            ts.append(t_draw(seed + 5000 + it * 16 + h, B).to(DEV))
            ctr += 1
        zs.append(z_draw(seed + 1000 + it * 16 + critic_iter, B).to(DEV))
        eng.train_iteration(batches, zs, ts if loss == "gp" else None)
        torch.cuda.synchronize()
        traj.append((float(eng.closs[0]), float(eng.gloss)))
    ref = g["traj"]
    print(traj, ref)
    # same stated band as the oracle-vs-reference test: first iteration 1e-5, later ones 1e-2
    for i, ((w, gl), r) in enumerate(zip(traj, ref)):
        tol = 2e-5 if i == 0 else 1e-2
        assert abs(w - r[0]) <= tol * max(1.0, abs(r[0])), (i, w, r)
        assert abs(gl - r[1]) <= tol * max(1.0, abs(r[1])), (i, gl, r)
    for k in dp:
        check_summary(g, f"D.{k}", eng.D.p[k], rtol=5e-3)
    for k in gp_:
        check_summary(g, f"G.{k}", eng.G.p[k], rtol=5e-3)
    for k, v in eng.Gbuf.p.items():
        check_summary(g, f"G.{k}", v, rtol=5e-3)
    assert eng.g_batches_tracked == 6 * iters


def test_one_adam_step_matches_oracle(hip):
    """After exactly one critic step the weights must match the oracle's Adam to fp32 rounding."""
    B, seed = 4, 99
    eng, gp_, dp = make_engine(hip, B, "gp", "sigmoid", seed)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    z, t = z_draw(seed + 2, B), t_draw(seed + 4, B)
    w, grads = orc.critic_grads(dp, gp_, orc.make_gen_buffers(), rs, rp, z, "gp", 10.0, t, "sigmoid")
    opt = orc.AdamState(dp)
    before = {k: v.clone() for k, v in dp.items()}
    opt.apply(dp, grads)
    eng.critic_step(rs.to(DEV), rp.to(DEV), z.to(DEV), t.to(DEV))
    torch.cuda.synchronize()
    for k, v in dp.items():
        step = (v - before[k])
        got = eng.D.p[k].cpu() - before[k]
        # the first Adam step is -lr * sign(g) (|m/sqrt(v)| = 1): compare the steps themselves
        bad = (got - step).abs() > 1e-6
        frac = float(bad.float().mean())
        assert frac < 2e-3, (k, frac)   # sign flips only where the gradient is rounding noise
