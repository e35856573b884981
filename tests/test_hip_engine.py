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
GOLDEN_TIGHT = 2e-5    # bound on the reference-made fp32 gradients (after removing the effect of differing LeakyReLU slopes)
FLIPS = {}             # fixture -> LeakyReLU slopes that differed from the fp32 oracle's in this session


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


def _mask(h):
    return torch.where(h > 0, 1.0, 0.1).double().cpu()


def engine_masks(eng, mode):
    """The LeakyReLU slope pattern the HIP run actually took, keyed as the oracle expects."""
    B = eng.B
    m = {}
    for i in range(eng.nG):
        m[f"g2d.{i}"] = _mask(eng.gH[i + 1])
        m[f"g1d.{i}"] = _mask(eng.hH[i + 1])
    for i in range(eng.nS):
        if mode == "generator":
            m[f"d_main.{i}"] = _mask(eng.dH[i + 1][:B])
        else:
            m[f"d_main.{i}"] = _mask(eng.dH[i + 1][:2 * B])
            if eng.loss == "gp":
                m[f"d_gp.{i}"] = _mask(eng.dH[i + 1][2 * B:3 * B])
    return m


def check_region_choice(masks, acts):
    """The HIP run may sit on a different side of a LeakyReLU kink than the fp64 oracle only where
    the oracle's own activation is within 2e-5 of the kink (relative to the tensor's rms)."""
    names = {"g2d": "g2d.act", "g1d": "g1d.act", "d_main": "d_main2d.act", "d_gp": "d_gp2d.act"}
    flips = 0
    for k, mk in masks.items():
        tag, i = k.rsplit(".", 1)
        h = acts[f"{names[tag]}{i}"].detach().double()
        mine = torch.where(h > 0, 1.0, 0.1).double()
        diff = mine != mk
        if diff.any():
            rms = float(h.pow(2).mean().sqrt())
            worst = float(h[diff].abs().max())
            assert worst <= 2e-5 * rms, (k, worst, rms)
            flips += int(diff.sum())
    return flips


def count_flips(masks, acts):
    """Activations whose LeakyReLU slope in the HIP run differs from the oracle run that recorded `acts`."""
    names = {"g2d": "g2d.act", "g1d": "g1d.act", "d_main": "d_main2d.act", "d_gp": "d_gp2d.act"}
    n = 0
    for k, mk in masks.items():
        tag, i = k.rsplit(".", 1)
        h = acts[f"{names[tag]}{i}"].detach()
        n += int((torch.where(h > 0, 1.0, 0.1).double() != mk).sum())
    return n


def held_tight(g, grads, delta, flips, what):
    """Hold `grads` to the fixture's gradients at GOLDEN_TIGHT: as they are, or with the flipped slopes' effect (delta)
    taken out; returns which one matched.  With flips and neither matching, the formula-level bound 5e-3 is all a direct
    fp32-to-fp32 comparison can give (the 1e-5 bound is enforced against the fp64 oracle on the same linear region)."""
    def ok(sub, rtol):
        try:
            for k, v in grads.items():
                check_summary(g, f"grad.{k}", v - (delta[k] if sub else 0.0), rtol=rtol, what=what)
            return True
        except AssertionError:
            return False
    if ok(False, GOLDEN_TIGHT):
        return f"{GOLDEN_TIGHT:g} as computed"
    if flips and ok(True, GOLDEN_TIGHT):
        return f"{GOLDEN_TIGHT:g} after removing the flipped slopes' effect"
    assert flips, f"{what} no slope flip, yet the gradients miss the reference's at {GOLDEN_TIGHT:g}"
    for k, v in grads.items():
        check_summary(g, f"grad.{k}", v, rtol=5e-3, what=what)
    return "5e-3 (flips on both sides)"


def make_engine(hip, B, loss, act, seed, data=64):
    from pgan_engine import Engine
    eng = Engine(hip, B, torch.device(DEV), loss=loss, act=act, data=data)
    gp_ = orc.make_params(orc.gen_param_shapes(data=data), seed)
    dp = orc.make_params(orc.disc_param_shapes(data), seed + 1)
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
    # Gradients against the reference's fp32 run: a LeakyReLU pre-activation within rounding of 0
    # (expected about once per 2.5e6 activations) may take the other slope in one of the two fp32
    # implementations, which shifts a whole sample's gradient chain by O(1e-2) and the batch
    # gradient by O(1e-4) norm-wise.  With millions of activations per step some always lie that
    # close to the kink, so whether a flip HAPPENED is counted (slope pattern of the HIP run against
    # that of the fp32 oracle, the reference's own arithmetic) and its effect is taken out: the fp32
    # oracle is evaluated on both patterns and the difference of the two gradients -- exactly what the
    # flipped slopes change, in the reference's arithmetic -- is subtracted from the HIP gradient.
    # What remains is held to the reference-made gradients at 2e-5.  The fixture itself was made on another CPU than
    # the one this test's oracle runs on, so for a pre-activation that close to the kink the reference's run may have sided
    # with the HIP run rather than with today's oracle: the uncorrected gradient is accepted as well (held_tight below).
    acts = {}
    _, g_own = orc.critic_grads(dp, gp_, None, rs, rp, z, loss, float(lam), t, "sigmoid", acts=acts)
    masks = engine_masks(eng, "critic")
    flips = count_flips(masks, acts)
    delta = {k: 0.0 for k in dp}
    if flips:
        _, g_hip = orc.critic_grads(dp, gp_, None, rs, rp, z, loss, float(lam), t, "sigmoid", masks=masks)
        delta = {k: (g_hip[k] - g_own[k]) for k in dp}
    how = held_tight(g, {k: eng.D.g[k].cpu() for k in dp}, delta, flips, f"{fname}:")
    FLIPS[fname] = (flips, how)
    print(f"{fname}: {flips} slope flips against the fp32 oracle; gradients held to the reference's at {how}")


def test_forward_activations_vs_reference_golden(hip):
    """Per-layer activations of both networks against what the reference's own modules produced (forward_B4.npz, written
    from layers.py): every BatchNorm + LeakyReLU stage of both generator towers, the generated sites and positions, the
    running statistics after one forward (sigmoid and tanh generators), then the discriminator on cat(real, fake): its
    four GroupNorm + LeakyReLU stages, the 1-D tower's output features and the scores.  1e-5 (forward values do not
    depend on which side of a LeakyReLU kink a rounding error falls)."""
    g = load_golden("forward_B4.npz")
    B, seed = [int(v) for v in g["meta"]]
    from pgan_engine import Engine
    z = z_draw(seed + 2, B)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    fake = None
    for act in ("sigmoid", "tanh"):
        eng, gp_, dp = make_engine(hip, B, "gp", act, seed)
        xs, xp = torch.empty(B, 1, 64, 64, device=DEV), torch.empty(B, 1, 64, device=DEV)
        eng.gen_forward(z.to(DEV), xs, xp)
        torch.cuda.synchronize()
        check_summary(g, f"{act}.xs", xs)
        check_summary(g, f"{act}.xp", xp)
        for i in range(4):
            check_summary(g, f"{act}.g2d.act{i}", eng.gH[i + 1])
            check_summary(g, f"{act}.g1d.act{i}", eng.hH[i + 1])
        for k, v in eng.Gbuf.p.items():
            check_summary(g, f"{act}.buf.{k}", v)
        assert eng.g_batches_tracked == 1
        if act == "sigmoid":
            fake = (xs.clone(), xp.clone())
            score = torch.empty(2 * B, device=DEV)
            eng.disc_forward(torch.cat([rs.to(DEV), fake[0]]), torch.cat([rp.to(DEV), fake[1]]), score)
            torch.cuda.synchronize()
            check_summary(g, "d.score", score.view(2 * B, 1))
            for i in range(4):
                check_summary(g, f"d.d2d.act{i}", eng.dH[i + 1][:2 * B])
            check_summary(g, "d.d1d.act2", eng.F1[:2 * B].view(2 * B, 64, 55))      # the 1-D tower's last stage, flattened


@pytest.mark.parametrize("loss,act,B", [("gp", "tanh", 8), ("div", "tanh", 8), ("gp", "sigmoid", 64),
                                        ("div", "sigmoid", 32), ("gp", "sigmoid", 1), ("div", "sigmoid", 3),
                                        ("gp", "tanh", 5)])
def test_critic_grads_vs_oracle(hip, loss, act, B):
    """Every discriminator gradient within 1e-5 (norm-wise) of the fp64 oracle evaluated on the same
    LeakyReLU linear region, at test size and at the benchmark's batch size."""
    seed = 1234
    eng, gp_, dp = make_engine(hip, B, loss, act, seed)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    if act == "tanh":
        rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
    z, t = z_draw(seed + 2, B), t_draw(seed + 4, B)
    eng.critic_grads(rs.to(DEV), rp.to(DEV), z.to(DEV), t.to(DEV))
    torch.cuda.synchronize()
    masks = engine_masks(eng, "critic")
    P64 = {k: v.double() for k, v in dp.items()}
    G64 = {k: v.double() for k, v in gp_.items()}
    args = (P64, G64, None, rs.double(), rp.double(), z.double(), loss, 10.0, t.double(), act)
    w, grads = orc.critic_grads(*args, masks=masks)
    acts = {}
    orc.critic_grads(*args, acts=acts)
    flips = check_region_choice(masks, acts)
    print(f"{loss} B={B}: {flips} activations on the other side of the kink")
    assert abs(float(eng.closs[0]) - float(w)) <= 1e-5 * max(1.0, abs(float(w)))
    for k, gr in grads.items():
        if float(gr.norm()) == 0:
            # exactly 0 in the oracle (+1/B and -1/B seeds cancel); fp32 sums of 1/B leave rounding when B is not 2^k
            assert float(eng.D.g[k].abs().max()) < (1e-9 if B & (B - 1) == 0 else 1e-6), k
            continue
        e = rel_err(eng.D.g[k], gr)
        print(f"{loss} {k}: {e:.2e}")
        assert e < 1e-5, (k, e)
    gbuf = orc.make_gen_buffers()
    orc.gen_forward(gp_, gbuf, z, act)
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
    acts = {}
    _, g_own = orc.generator_grads(dp, gp_, None, z_draw(seed + 2, B), act, acts=acts)
    masks = engine_masks(eng, "generator")
    flips = count_flips(masks, acts)
    delta = {k: 0.0 for k in gp_}
    if flips:       # see the critic golden test: the flipped slopes' effect, in the reference's arithmetic, is taken out
        _, g_hip = orc.generator_grads(dp, gp_, None, z_draw(seed + 2, B), act, masks=masks)
        delta = {k: (g_hip[k] - g_own[k]) for k in gp_}
    how = held_tight(g, {k: eng.G.g[k].cpu() for k in gp_}, delta, flips, f"gen {act}:")
    FLIPS[f"generator_{act}_B4.npz"] = (flips, how)
    print(f"generator {act}: {flips} slope flips against the fp32 oracle; gradients held to the reference's at {how}")


def test_reference_gradient_comparisons_ran_tight():
    """The direct comparisons with the reference's fp32 gradients above ran at 2e-5 for all four batch-4 fixtures (gp and
    div critic steps, both generator steps), as computed or after removing the effect of the slopes that differed from
    the fp32 oracle's; this prints the flip count and the bound reached per fixture, the batch-64 one included (several
    flips there, some on the fixture's side and some on today's oracle's: 5e-3, with the 1e-5 bound enforced against the
    fp64 oracle on the same linear region by test_critic_grads_vs_oracle).  (Runs after those tests.)"""
    assert len(FLIPS) == 5, sorted(FLIPS)
    print("fixture -> (slope flips against the fp32 oracle, how the reference's gradients were matched):", dict(FLIPS))
    assert all(how.startswith("2e-05") for k, (_, how) in FLIPS.items() if "_B4" in k), FLIPS


@pytest.mark.parametrize("act,B", [("sigmoid", 8), ("tanh", 8), ("sigmoid", 64), ("sigmoid", 1), ("tanh", 3)])
def test_generator_grads_vs_oracle_fp64(hip, act, B):
    seed = 4321
    eng, gp_, dp = make_engine(hip, B, "gp", act, seed)
    z = z_draw(seed + 2, B)
    eng.generator_grads(z.to(DEV))
    torch.cuda.synchronize()
    masks = engine_masks(eng, "generator")
    P64 = {k: v.double() for k, v in dp.items()}
    G64 = {k: v.double() for k, v in gp_.items()}
    loss, grads = orc.generator_grads(P64, G64, None, z.double(), act, masks=masks)
    acts = {}
    orc.generator_grads(P64, G64, None, z.double(), act, acts=acts)
    print("flips", check_region_choice(masks, acts))
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
    # Stated band: 2e-3 on the first iteration's losses, 1e-2 afterwards.  Adam's early steps are
    # sign-like (|m/sqrt(v)| ~ 1), so weights whose gradient is rounding noise step +-lr differently
    # in any two fp32 implementations; an fp64 run of the oracle differs from the fp32 reference by
    # 3e-3 after two iterations.  Single-step agreement is checked exactly in the tests above.
    for i, ((w, gl), r) in enumerate(zip(traj, ref)):
        tol = 2e-3 if i == 0 else 1e-2
        assert abs(w - r[0]) <= tol * max(1.0, abs(r[0])), (i, w, r)
        assert abs(gl - r[1]) <= tol * max(1.0, abs(r[1])), (i, gl, r)
    # weights: an entry whose gradient is rounding noise may step the other way on every update
    lr = 5e-5
    for k in dp:
        check_summary(g, f"D.{k}", eng.D.p[k], rtol=1e-4, atol=2 * lr * critic_iter * iters)
    for k in gp_:
        check_summary(g, f"G.{k}", eng.G.p[k], rtol=1e-4, atol=2 * lr * iters)
    for k, v in eng.Gbuf.p.items():
        check_summary(g, f"G.{k}", v, rtol=5e-3)
    assert eng.g_batches_tracked == 6 * iters
    # Adam's second-moment estimates at the end of the run against the reference optimizer's state (exp_avg_sq of
    # torch.optim.Adam: with beta2 = 0.999 every step's squared gradient carries the same weight, so the estimate is an
    # average over the run and two fp32 runs that drift apart by a few sign-like steps still agree on it; exp_avg, with
    # beta1 = 0.5, is the last two gradients and is pinned by the one-step fixture below instead)
    worst = 0.0
    for name, store in (("D", eng.D), ("G", eng.G)):
        for k in store.p:
            st = g[f"{name}.exp_avg_sq.{k}#stat"]
            mine = float(store.v[k].double().sum())
            worst = max(worst, abs(mine - st[0]) / max(st[0], 1e-300))
    print(f"{fname}: Adam second moments vs the reference optimizer state, worst relative difference of a tensor's sum {worst:.2e}")
    assert worst < 0.05, worst


def test_one_step_of_each_optimizer_vs_reference_golden(hip):
    """One critic step and one generator step (critic_iter 1) against the reference's modules driven by torch.optim.Adam
    (train_gp_sigmoid_onestep_B8.npz): after exactly one step of each optimizer exp_avg = (1 - beta1) g and
    exp_avg_sq = (1 - beta2) g^2, so the reference optimizer's state pins the fused Adam's moments (and through them the
    first gradients) directly; the updated weights move by lr times a near-sign, so entries whose gradient is rounding
    noise may step the other way (absolute bound 2 lr)."""
    g = load_golden("train_gp_sigmoid_onestep_B8.npz")
    B, seed, lam, iters, critic_iter, normalize = [int(v) for v in g["meta"]]
    assert (iters, critic_iter, normalize) == (1, 1, 0)
    eng, gp_, dp = make_engine(hip, B, "gp", "sigmoid", seed)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    z0, t0, z1 = z_draw(seed + 1000, B), t_draw(seed + 5000, B), z_draw(seed + 1001, B)
    eng.train_iteration([(rs.to(DEV), rp.to(DEV))], [z0.to(DEV), z1.to(DEV)], [t0.to(DEV)])
    torch.cuda.synchronize()
    ref = g["traj"][0]
    assert abs(float(eng.closs[0]) - ref[0]) <= 2e-5 * max(1.0, abs(ref[0]))
    assert abs(float(eng.gloss) - ref[1]) <= 1e-4 * max(1.0, abs(ref[1]))        # after the critic's Adam step
    # slope flips against the fp32 oracle's critic step (the generator step runs on weights that already differ by the
    # critic's near-sign Adam step, so its moments get the looser bound)
    acts = {}
    orc.critic_grads(dp, gp_, None, rs, rp, z0, "gp", float(lam), t0, "sigmoid", acts=acts)
    lr = 5e-5
    for k in dp:
        check_summary(g, f"D.exp_avg.{k}", eng.D.m[k], rtol=5e-3)
        check_summary(g, f"D.exp_avg_sq.{k}", eng.D.v[k], rtol=1e-2, atol=1e-16)
        check_summary(g, f"D.{k}", eng.D.p[k], rtol=1e-5, atol=2 * lr)
    for k in gp_:
        check_summary(g, f"G.exp_avg.{k}", eng.G.m[k], rtol=2e-2)
        check_summary(g, f"G.exp_avg_sq.{k}", eng.G.v[k], rtol=4e-2, atol=1e-16)
        check_summary(g, f"G.{k}", eng.G.p[k], rtol=1e-5, atol=2 * lr)
    # tight where it can be: moments are the gradients of the first step, scaled
    d_m = torch.cat([eng.D.m[k].reshape(-1) for k in dp]).double().cpu()
    d_g = torch.cat([eng.D.g[k].reshape(-1) for k in dp]).double().cpu()
    assert float((d_m - 0.5 * d_g).norm() / d_m.norm()) < 1e-6
    d_v = torch.cat([eng.D.v[k].reshape(-1) for k in dp]).double().cpu()
    one_minus_b2 = float(np.float32(1) - np.float32(0.999))          # as fp32 arithmetic has it
    assert float((d_v - one_minus_b2 * d_g * d_g).norm() / d_v.norm()) < 1e-6


def test_one_adam_step_matches_oracle(hip):
    """After exactly one critic step from identical weights: Adam moments match the oracle's, and
    every weight whose gradient is not rounding noise moved by the same amount."""
    B, seed = 4, 99
    eng, gp_, dp = make_engine(hip, B, "gp", "sigmoid", seed)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    z, t = z_draw(seed + 2, B), t_draw(seed + 4, B)
    eng.critic_step(rs.to(DEV), rp.to(DEV), z.to(DEV), t.to(DEV))
    torch.cuda.synchronize()
    masks = engine_masks(eng, "critic")
    P64 = {k: v.double() for k, v in dp.items()}
    w, grads = orc.critic_grads(P64, {k: v.double() for k, v in gp_.items()}, None, rs.double(), rp.double(),
                                z.double(), "gp", 10.0, t.double(), "sigmoid", masks=masks)
    opt = orc.AdamState(P64)
    before = {k: v.clone() for k, v in P64.items()}
    opt.apply(P64, grads)
    for k, v in P64.items():
        if float(grads[k].norm()) == 0:
            continue
        assert rel_err(eng.D.m[k], opt.m[k]) < 1e-5, k
        assert rel_err(eng.D.v[k], opt.v[k]) < 2e-5, k
        g = grads[k]
        solid = g.abs() > 1e-3 * g.pow(2).mean().sqrt()
        step = (v - before[k])[solid]
        got = (eng.D.p[k].cpu().double() - before[k])[solid]
        ulp = 1.2e-7 * before[k][solid].abs() + 1e-9   # the step is added to an fp32 weight
        assert bool(((got - step).abs() <= 2 * ulp).all()), k


def test_two_chain_critic_step_equals_single_chain(hip):
    """gp critic step at the benchmark batch: the two concurrent sample chains (Wasserstein | penalty, on two
    streams) give what the one-chain order gives on the same inputs, on repeated calls: stage activations of the
    primal slots to rounding (launch sizes differ, so the split-K summation order does).  The tangent slots start
    from v = d penalty / d g, which jumps by 2 lambda / (B 4096) where a rounding-level difference flips the sign of
    a g element next to 0; gradients therefore agree to 1e-4 (measured: below 7e-5 in 150 draws, 2e-6 without a flip)."""
    from pgan_engine import Engine
    B = 64
    engs = []
    for two in (True, False):
        e = Engine(hip, B, DEV)
        e.two_chains = two
        e.D.load(orc.make_params(orc.disc_param_shapes(), 7))
        e.G.load(orc.make_params(orc.gen_param_shapes(), 8))
        engs.append(e)
    a, b = engs
    for it in range(4):
        rs, rp = orc.synthetic_alignments(B, 100 + it)
        g = torch.Generator().manual_seed(it)
        z, t = torch.randn(B, 128, 1, 1, generator=g), torch.rand(B, generator=g)
        for e in engs:
            e.critic_grads(rs.to(DEV), rp.to(DEV), z.to(DEV), t.to(DEV))
        torch.cuda.synchronize()
        assert torch.equal(a.X0[:3 * B], b.X0[:3 * B])
        for k in range(1, 5):
            for buf in ("dA", "dH"):
                x, y = getattr(a, buf)[k][:3 * B], getattr(b, buf)[k][:3 * B]
                err = float((x - y).abs().max()) / float(y.abs().max())
                assert err < 2e-5, (it, buf, k, err)
        assert rel_err(a.D.gflat, b.D.gflat) < 1e-4
        assert torch.allclose(a.closs, b.closs, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("augment", [False, True])
@pytest.mark.parametrize("loss", ["gp", "div"])
def test_trainer_routes_agree(hip, loss, augment, monkeypatch):
    """FusedTrainer: eager iterations, hipGraph replay and launch-tape replay (walked in C by the library, and call by
    call from Python) enqueue the same kernels on the same streams -- same device RNG draws, same pool cursor, and, every
    kernel's sums being taken in a fixed order (K splits are in-block or two-addend), bit-identical losses, weights and
    running statistics.  augment: with per-batch label smoothing and monomorphic-site switching inside the step."""
    import pgan_tape
    from pgan_trainer import FusedTrainer
    B, n_it, lr = 8, 3, 5e-5
    rs, rp = orc.synthetic_alignments(64, 9)
    out = {}
    for route in ("eager", "graph", "tape", "pytape"):
        monkeypatch.setenv("PGAN_NATIVE_TAPE", "0" if route == "pytape" else "1")
        tr = FusedTrainer(hip, B, DEV, loss=loss, critic_iter=2, seed=77)
        tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
        tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
        tr.weights_changed()
        tr.set_pool(rs, rp)
        if augment:
            tr.enable_augment(label_noise=0.05, mono_switch=(0, 10))
        tr.use_graph, tr.use_tape = route == "graph", route in ("tape", "pytape")
        for _ in range(n_it):
            tr.step()
        torch.cuda.synchronize()
        assert (tr.graph is not None) == (route == "graph") and (tr.tape is not None) == (route in ("tape", "pytape"))
        if route == "tape":      # one native tape holds the whole iteration: no Python between its launches
            assert isinstance(tr.tape, pgan_tape.CompiledTape) and tr.tape.python_steps == 0
            assert len(tr.tape.segments) == 1 and tr.tape.native_calls == len(tr.tape.entries) > 200
        if route == "pytape":
            assert isinstance(tr.tape, list)
        out[route] = dict(d=tr.eng.D.flat.clone(), g=tr.eng.G.flat.clone(), losses=tr.losses(), cursor=int(tr.cursor),
                          rng=torch.stack([tr.rng_z, tr.rng_t, tr.rng_aug, tr.rng_mono]), nb=tr.eng.g_batches_tracked,
                          step=(tr.eng.D.step, tr.eng.G.step),
                          run=tr.eng.Gbuf.p["twod_conv_layers.1.running_mean"].clone())
    a = out["eager"]
    for route in ("graph", "tape", "pytape"):
        b = out[route]
        assert b["cursor"] == a["cursor"] and torch.equal(b["rng"], a["rng"]) and b["nb"] == a["nb"] and b["step"] == a["step"]
        assert tuple(a["losses"]) == tuple(b["losses"]), (route, a["losses"], b["losses"])
        for k in ("d", "g", "run"):
            assert torch.equal(a[k], b[k]), (route, k, rel_err(b[k], a[k]))


@pytest.mark.parametrize("act", ["sigmoid", "tanh"])
def test_generator_passes_in_one_forward(hip, act):
    """gen_forward_chunks: three forward passes as one launch per layer equal three separate forwards on the same z
    (outputs, per-pass BatchNorm statistics, running statistics after the three, batch counter), and the last pass
    is where the single-pass workspace points, so a backward after it works as after a forward of its own."""
    from pgan_engine import Engine
    B, m = 8, 3
    engs = []
    for chunks in (m, 0):
        e = Engine(hip, B, DEV, act=act)
        e.D.load(orc.make_params(orc.disc_param_shapes(), 7))
        e.G.load(orc.make_params(orc.gen_param_shapes(), 8))
        if chunks:
            e.enable_gen_chunks(chunks)
        engs.append(e)
    a, b = engs
    z = torch.randn(m * B, 128, generator=torch.Generator().manual_seed(3)).to(DEV)
    a.Zw.copy_(z)
    a.gen_forward_chunks(0, m)
    outs = []
    for c in range(m):
        b.Z.copy_(z[c * B:(c + 1) * B])
        xs, xp = torch.zeros(B, 1, 64, 64, device=DEV), torch.zeros(B, 1, 64, device=DEV)
        b.gen_forward_resident(xs, xp)
        outs.append((xs, xp, [t.clone() for t in b.gStat[1:]], [t.clone() for t in b.hStat[1:]]))
    torch.cuda.synchronize()
    assert a.g_batches_tracked == b.g_batches_tracked == m
    for c, (xs, xp, gst, hst) in enumerate(outs):
        assert rel_err(a.fake_s[c * B:(c + 1) * B], xs) < 1e-5 and rel_err(a.fake_p[c * B:(c + 1) * B], xp) < 1e-5
        for i in range(4):
            assert torch.allclose(a.gStatw[i + 1][c], gst[i], rtol=1e-5, atol=1e-6)
            assert torch.allclose(a.hStatw[i + 1][c], hst[i], rtol=1e-5, atol=1e-6)
    for k, v in b.Gbuf.p.items():
        assert torch.allclose(a.Gbuf.p[k], v, rtol=1e-5, atol=1e-6), k
    for i in range(1, 5):       # the last pass is the single-pass workspace
        assert rel_err(a.gH[i], b.gH[i]) < 1e-5 and rel_err(a.hH[i], b.hH[i]) < 1e-5
    dxs, dxp = torch.randn(B, 1, 64, 64, device=DEV), torch.randn(B, 1, 64, device=DEV)
    a.gen_backward(a.gXs, a.gXp, dxs, dxp)
    b.gen_backward(outs[-1][0], outs[-1][1], dxs, dxp)
    torch.cuda.synchronize()
    assert rel_err(a.G.gflat, b.G.gflat) < 1e-5


def test_trainer_batched_generator_equals_separate_forwards(hip):
    """FusedTrainer with the generator forwards of an iteration batched (default) against one forward per step: same
    z stream, losses, pool cursor, batch counter; weights equal up to rounding-noise +-lr steps."""
    from pgan_trainer import FusedTrainer
    B, n_it, lr = 8, 3, 5e-5
    rs, rp = orc.synthetic_alignments(64, 9)
    out = {}
    for batched in (True, False):
        tr = FusedTrainer(hip, B, DEV, critic_iter=3, seed=78, batched_generator=batched)
        tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
        tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
        tr.weights_changed()
        tr.set_pool(rs, rp)
        for _ in range(n_it):
            tr.iteration()
        torch.cuda.synchronize()
        out[batched] = dict(d=tr.eng.D.flat.clone(), g=tr.eng.G.flat.clone(), losses=tr.losses(), cursor=int(tr.cursor),
                            rng=tr.rng_z.clone(), nb=tr.eng.g_batches_tracked, z=tr.eng.Z.clone(),
                            run={k: v.clone() for k, v in tr.eng.Gbuf.p.items()})
    a, b = out[True], out[False]
    assert a["cursor"] == b["cursor"] and torch.equal(a["rng"], b["rng"]) and a["nb"] == b["nb"]
    assert torch.equal(a["z"], b["z"])                  # the generator step's z: same draw either way
    for x, y in zip(a["losses"], b["losses"]):
        assert abs(x - y) <= 1e-2 * max(1.0, abs(x)), (a["losses"], b["losses"])
    for k in ("d", "g"):
        assert float((a[k] - b[k]).abs().max()) <= 2 * lr * 3 * n_it and rel_err(a[k], b[k]) < 1e-3, k
    for k in a["run"]:                                   # running statistics follow the weights' +-lr steps
        assert torch.allclose(a["run"][k], b["run"][k], rtol=1e-3, atol=2 * lr * 3 * n_it), k


# ------------------------------------------------------------------------------------------------------------------
# 128 x 128 alignments: the scaled networks (one more stride-2 stage in each 2-D tower and in the generator's 1-D
# tower).  PARITY UNPINNED by the reference, which has no model of that size (its discriminator head is
# Linear(11712, 1), /root/reference/src/layers.py:482): the checker is the oracle's restatement of the same layer
# types on the build's own layer tables, held to the same 1e-5 bound.
@pytest.mark.parametrize("loss,act,B", [("gp", "tanh", 4), ("div", "sigmoid", 3), ("gp", "sigmoid", 16)])
def test_scaled_128_critic_grads_vs_oracle(hip, loss, act, B):
    seed, S = 2468, 128
    eng, gp_, dp = make_engine(hip, B, loss, act, seed, data=S)
    rs, rp = orc.synthetic_alignments(B, seed + 3, size=S)
    if act == "tanh":
        rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
    z, t = z_draw(seed + 2, B), t_draw(seed + 4, B)
    eng.critic_grads(rs.to(DEV), rp.to(DEV), z.to(DEV), t.to(DEV))
    torch.cuda.synchronize()
    masks = engine_masks(eng, "critic")
    P64 = {k: v.double() for k, v in dp.items()}
    G64 = {k: v.double() for k, v in gp_.items()}
    args = (P64, G64, None, rs.double(), rp.double(), z.double(), loss, 10.0, t.double(), act)
    w, grads = orc.critic_grads(*args, masks=masks)
    acts = {}
    orc.critic_grads(*args, acts=acts)
    check_region_choice(masks, acts)
    assert abs(float(eng.closs[0]) - float(w)) <= 1e-5 * max(1.0, abs(float(w)))
    for k, gr in grads.items():
        if float(gr.norm()) == 0:
            assert float(eng.D.g[k].abs().max()) < (1e-9 if B & (B - 1) == 0 else 1e-6), k
            continue
        e = rel_err(eng.D.g[k], gr)
        print(f"128 {loss} {k}: {e:.2e}")
        assert e < 1e-5, (k, e)
    gbuf = orc.make_gen_buffers(data=S)
    orc.gen_forward(gp_, gbuf, z, act)
    for k, v in gbuf.items():
        if "running" in k:
            assert rel_err(eng.Gbuf.p[k], v) < 1e-5, k


@pytest.mark.parametrize("act,B", [("sigmoid", 4), ("tanh", 16)])
def test_scaled_128_generator_grads_vs_oracle(hip, act, B):
    seed, S = 8642, 128
    eng, gp_, dp = make_engine(hip, B, "gp", act, seed, data=S)
    z = z_draw(seed + 2, B)
    eng.generator_grads(z.to(DEV))
    torch.cuda.synchronize()
    masks = engine_masks(eng, "generator")
    P64 = {k: v.double() for k, v in dp.items()}
    G64 = {k: v.double() for k, v in gp_.items()}
    loss, grads = orc.generator_grads(P64, G64, None, z.double(), act, masks=masks)
    acts = {}
    orc.generator_grads(P64, G64, None, z.double(), act, acts=acts)
    check_region_choice(masks, acts)
    assert abs(float(eng.gloss) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    for k, gr in grads.items():
        e = rel_err(eng.G.g[k], gr)
        print(f"128 gen {k}: {e:.2e}")
        assert e < 1e-5, (k, e)


def test_scaled_128_trainer_routes_agree(hip):
    """FusedTrainer on 128 x 128 alignments: eager, launch tape and step graph train alike."""
    from pgan_trainer import FusedTrainer
    B, n_it, lr, S = 4, 2, 5e-5, 128
    rs, rp = orc.synthetic_alignments(32, 9, size=S)
    out = {}
    for route in ("eager", "tape", "graph"):
        tr = FusedTrainer(hip, B, DEV, critic_iter=2, seed=77, data=S)
        tr.eng.D.load(orc.make_params(orc.disc_param_shapes(S), 7))
        tr.eng.G.load(orc.make_params(orc.gen_param_shapes(data=S), 8))
        tr.weights_changed()
        tr.set_pool(rs, rp)
        tr.use_graph, tr.use_tape = route == "graph", route == "tape"
        for _ in range(n_it):
            tr.step()
        torch.cuda.synchronize()
        out[route] = dict(d=tr.eng.D.flat.clone(), g=tr.eng.G.flat.clone(), losses=tr.losses())
        assert all(np.isfinite(v) for v in out[route]["losses"])
    for route in ("tape", "graph"):
        for k in ("d", "g"):
            assert float((out["eager"][k] - out[route][k]).abs().max()) <= 2 * lr * 2 * n_it, (route, k)
            assert rel_err(out[route][k], out["eager"][k]) < 1e-3, (route, k)


@pytest.mark.parametrize("loss,sync_bn,data", [("gp", False, 64), ("div", False, 64), ("gp", True, 64), ("gp", False, 128),
                                               ("div", True, 128)])
def test_operand_bounds_hold_in_every_split_convolution(hip, loss, sync_bn, data, monkeypatch):
    """The fp16-split convolutions trust the magnitude bounds their producers report (a bound below the truth could
    overflow fp16 silently).  With the backend's checking mode on, every bounded convolution of a critic step and a
    generator step -- forward, both backward passes, tangent pass, weight gradients; BatchNorm statistics in one launch
    and as the data-parallel sums / apply pair -- compares its bounds with the operand's true magnitudes first."""
    from pgan_hip import HipOps
    B, seed = (8 if data == 64 else 4), 99
    eng, gp_, dp = make_engine(hip, B, loss, "sigmoid", seed, data=data)
    if sync_bn:
        eng.bn_sync = lambda part: None          # one rank's share is the whole batch: the sums are already global
    # weights far from their initial scale, so that the magnitudes differ by orders between stages
    with torch.no_grad():
        for k, v in eng.D.p.items():
            if k.endswith("weight") and v.dim() == 4:
                v.mul_(6.0)
    eng.ops.invalidate_weight_planes(reset=True)
    monkeypatch.setattr(HipOps, "check_bounds", True)
    before = HipOps.bounds_checked
    rs, rp = orc.synthetic_alignments(B, seed + 3, size=data)
    eng.critic_grads(rs.to(DEV), rp.to(DEV), z_draw(seed + 2, B).to(DEV), t_draw(seed + 4, B).to(DEV))
    eng.generator_grads(z_draw(seed + 5, B).to(DEV))
    torch.cuda.synchronize()
    made = HipOps.bounds_checked - before
    print(f"{loss}, sync_bn={sync_bn}, {data} x {data}: {made} bounds compared with their operands")
    assert made >= 30, made
    assert torch.isfinite(eng.D.gflat).all() and torch.isfinite(eng.G.gflat).all()


def test_bound_check_catches_a_bound_below_the_operand(hip, monkeypatch):
    from pgan_hip import HipOps, PganError
    monkeypatch.setattr(HipOps, "check_bounds", True)
    x = torch.randn(4, 64, 32, 32, device=DEV)
    w = torch.randn(128, 64, 4, 4, device=DEV) * 0.05
    y = torch.empty(4, 128, 16, 16, device=DEV)
    good = x.abs().reshape(4, 8, -1).amax(dim=2).contiguous()
    hip.conv_down(x, w, y, bound=good)
    bad = good.clone()
    bad[2] *= 0.5
    with pytest.raises(PganError, match="below the operand"):
        hip.conv_down(x, w, y, bound=bad)
    torch.cuda.synchronize()


@pytest.mark.parametrize("loss", ["gp", "div"])
def test_operand_bounds_hold_through_a_whole_iteration(loss, monkeypatch):
    """The same check over the trainer's own iteration, launched eagerly: the batched generator forwards (several passes in
    one launch per layer, whose per-channel bounds are combined over the passes), the two sample chains of a critic step
    on their streams, five critic steps and the generator step with weights that move between them."""
    from pgan_hip import HipOps
    from pgan_trainer import FusedTrainer
    ops = HipOps(torch.device(DEV))
    tr = FusedTrainer(ops, 8, torch.device(DEV), loss=loss, critic_iter=5, seed=31)
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    tr.weights_changed()
    rs, rp = orc.synthetic_alignments(96, 9)
    tr.set_pool(rs, rp)
    tr.iteration()
    torch.cuda.synchronize()
    monkeypatch.setattr(HipOps, "check_bounds", True)
    before = HipOps.bounds_checked
    for _ in range(2):
        tr.iteration()
    torch.cuda.synchronize()
    made = HipOps.bounds_checked - before
    print(f"{loss}: {made} bounds compared with their operands in two iterations")
    assert made >= 200, made
    assert all(np.isfinite(v) for v in tr.losses())
