"""Ownership of scratch and of sample slots on the multi-stream engine, and the steps that must stay inside a
recorded iteration: per-stream norm scratch at large batches (both generator towers on two streams), slice-confined
writes of every kernel that works on a slot range, allocation-free graph capture, a launch tape recorded right after
an evaluation forward, the device-resident replay buffer against the reference's host deque
(/root/reference/src/train_wgan_v2.py:98-117), and device-side genotype sampling (:162-192)."""
import random

import numpy as np
import pytest
import torch

import dcwgan_oracle as orc
from helpers import rel_err
from test_hip_engine import DEV, make_engine, z_draw

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pgan_hip import HipOps
    return HipOps(torch.device(DEV))


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [256, 512])
def test_generator_towers_on_two_streams_at_large_batch(hip, B):
    """gen_forward + gen_backward with the 1-D tower on the side stream beside the 2-D tower (both take the sliced
    BatchNorm path at these sizes, whose partial sums live in per-stream scratch) equal the same calls with every
    side stream switched off, bit for bit in the forward, on repeated calls; forward and running statistics also
    match the oracle."""
    from pgan_engine import Engine
    engs = []
    for streams in (True, False):
        e = Engine(hip, B, torch.device(DEV), generator_only=True)
        e.G.load(orc.make_params(orc.gen_param_shapes(), 8))
        if not streams:
            e.side.enabled = e.wside.enabled = e.gside.enabled = False
        engs.append(e)
    a, b = engs
    gp_ = orc.make_params(orc.gen_param_shapes(), 8)
    gbuf = orc.make_gen_buffers()
    for it in range(3):
        z = z_draw(500 + it, B)
        outs = []
        for e in engs:
            xs, xp = torch.zeros(B, 1, 64, 64, device=DEV), torch.zeros(B, 1, 64, device=DEV)
            e.gen_forward(z.to(DEV), xs, xp)
            outs.append((xs, xp))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), it
        for k in a.Gbuf.p:
            assert torch.equal(a.Gbuf.p[k], b.Gbuf.p[k]), (it, k)
        os_, op_ = orc.gen_forward(gp_, gbuf, z, "sigmoid")
        assert rel_err(outs[0][0], os_) < 1e-5 and rel_err(outs[0][1], op_) < 1e-5
        for k, v in gbuf.items():
            if "running" in k:
                assert rel_err(a.Gbuf.p[k], v) < 1e-5, (it, k)
        g = torch.Generator().manual_seed(it)
        dxs, dxp = torch.randn(B, 1, 64, 64, generator=g).to(DEV), torch.randn(B, 1, 64, generator=g).to(DEV)
        for e, (xs, xp) in zip(engs, outs):
            e.gen_backward(xs, xp, dxs, dxp)
        torch.cuda.synchronize()
        assert rel_err(a.G.gflat, b.G.gflat) < 2e-6, it


def test_plotting_forward_n500_matches_oracle(hip):
    """The evaluation forward of the trainer (500 samples, BatchNorm over all of them, both towers concurrently)
    against the oracle's generator: activations to 1e-5, and the 2DSWD of the two sets of genotypes (same uniform
    draws, same real sample, seeded projections) inside the stated band of 2 % + 0.01."""
    import data_processing_v2 as dpv
    from pgan_engine import Engine
    n, seed = 500, 31
    ev = Engine(hip, n, torch.device(DEV), generator_only=True)
    gp_ = orc.make_params(orc.gen_param_shapes(), seed)
    ev.G.load(gp_)
    z = z_draw(seed + 1, n)
    fs, fp = torch.empty(n, 1, 64, 64, device=DEV), torch.empty(n, 1, 64, device=DEV)
    ev.gen_forward(z.to(DEV), fs, fp)
    torch.cuda.synchronize()
    os_, op_ = orc.gen_forward(gp_, None, z, "sigmoid")
    assert rel_err(fs, os_) < 1e-5 and rel_err(fp, op_) < 1e-5
    u = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 9)).random((n, 1, 64, 64))).float()
    real_eval, _ = orc.synthetic_alignments(n, seed + 11)
    d_hip = dpv.plot_sfs_pca((u < fs.cpu()).float().numpy(), 0, real_eval.numpy(), 64, seed=0)
    d_orc = dpv.plot_sfs_pca((u < os_.detach()).float().numpy(), 0, real_eval.numpy(), 64, seed=0)
    assert abs(d_hip - d_orc) <= 0.02 * d_orc + 0.01, (d_hip, d_orc)


# ------------------------------------------------------------------------------------------------------------------
SENT = 12345.678


def _guarded(shape, lo, hi, pad=3):
    """A buffer of `pad` + shape[0] + `pad` samples filled with a sentinel; returns (whole, view of [lo:hi))."""
    whole = torch.full((shape[0] + 2 * pad,) + tuple(shape[1:]), SENT, device=DEV)
    return whole, whole[pad + lo:pad + hi]


def _untouched(whole, lo, hi, pad=3):
    keep = torch.ones(whole.shape[0], dtype=torch.bool, device=DEV)
    keep[pad + lo:pad + hi] = False
    return bool((whole[keep] == SENT).all())


@pytest.mark.parametrize("N,lo,hi", [(5, 1, 4), (13, 3, 13), (67, 0, 67), (70, 33, 66)])
def test_kernels_write_only_their_slot_range(hip, N, lo, hi):
    """Every kernel the critic step runs on a slot range [lo:hi) of a per-layer buffer leaves the samples on either
    side of the range untouched (sample counts that are not multiples of any pixel tile, ranges that start and end
    inside a tile)."""
    g = torch.Generator().manual_seed(N)
    n = hi - lo
    for (cl, cs, hs) in ((1, 64, 32), (64, 128, 16), (128, 256, 8), (256, 512, 4)):
        w = (torch.randn(cs, cl, 4, 4, generator=g) * 0.05).to(DEV)
        x = torch.randn(N, cl, 2 * hs, 2 * hs, generator=g).to(DEV)
        y = torch.randn(N, cs, hs, hs, generator=g).to(DEV)
        whole, out = _guarded((N, cs, hs, hs), lo, hi)
        hip.conv_down(x[lo:hi], w, out)
        torch.cuda.synchronize()
        assert _untouched(whole, lo, hi), ("conv_down", cl, cs, hs)
        whole, out = _guarded((N, cl, 2 * hs, 2 * hs), lo, hi)
        hip.conv_up(y[lo:hi], w, out)
        torch.cuda.synchronize()
        assert _untouched(whole, lo, hi), ("conv_up", cl, cs, hs)
        # GroupNorm family on [N, cs, hs, hs]
        gam, bet = torch.rand(cs, generator=g).to(DEV) + 0.5, torch.randn(cs, generator=g).to(DEV)
        wh_h, h = _guarded((N, cs, hs, hs), lo, hi)
        wh_s, st = _guarded((N, 2), lo, hi)
        hip.gn_fwd(y[lo:hi], gam, bet, h, st)
        wh_d, d = _guarded((N, cs, hs, hs), lo, hi)
        wh_sb, sb = _guarded((N, 2), lo, hi)
        wh_pg, pgb = _guarded((N, 2 * cs), lo, hi)
        u = torch.randn(n, cs, hs, hs, generator=g).to(DEV)
        hip.gn_bwd(u, h, y[lo:hi], st, gam, d, sb, pgb[:, :cs], pgb[:, cs:])
        t = torch.randn(n, cs, hs, hs, generator=g).to(DEV)
        wh_hd, hd = _guarded((N, cs, hs, hs), lo, hi)
        wh_st, stt = _guarded((N, 2), lo, hi)
        hip.gn_jvp(t, h, y[lo:hi], st, gam, hd, stt)
        wh_p, p = _guarded((N, cs, hs, hs), lo, hi)
        wh_pg2, pgb2 = _guarded((N, 2 * cs), lo, hi)
        hip.gn_bwd2(None, u, t, h, y[lo:hi], st, sb, stt, gam, p, pgb2[:, :cs], pgb2[:, cs:])
        torch.cuda.synchronize()
        for name, whl in (("gn_fwd h", wh_h), ("gn_fwd stats", wh_s), ("gn_bwd d", wh_d), ("gn_bwd stats", wh_sb),
                          ("gn_bwd partials", wh_pg), ("gn_jvp", wh_hd), ("gn_jvp stats", wh_st), ("gn_bwd2", wh_p),
                          ("gn_bwd2 partials", wh_pg2)):
            assert _untouched(whl, lo, hi), (name, cs, hs)
            assert not bool((whl[3 + lo:3 + hi] == SENT).any()), (name, "range not fully written")
    # elementwise steps of the critic step
    real, fake = torch.rand(n, 1, 64, 64, generator=g).to(DEV), torch.rand(n, 1, 64, 64, generator=g).to(DEV)
    whole, out = _guarded((N, 1, 64, 64), lo, hi)
    hip.interpolate(torch.rand(n, generator=g).to(DEV), real, fake, out)
    wh_v, v = _guarded((N, 1, 64, 64), lo, hi)
    hip.gp_penalty(real, 10.0, v, torch.zeros(1, device=DEV))
    wh_u, u4 = _guarded((N, 512, 4, 4), lo, hi)
    hip.head_seed(torch.randn(1, 8192 + 3520, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV), u4)
    torch.cuda.synchronize()
    assert _untouched(whole, lo, hi) and _untouched(wh_v, lo, hi) and _untouched(wh_u, lo, hi)


# ------------------------------------------------------------------------------------------------------------------
def test_kernels_beside_other_streams_equal_their_standalone_runs(hip):
    """One gp critic step at batch 64 replayed from a launch tape (dense launches, both sample chains and the weight
    gradients side by side): the stage-1 activations of the penalised and of the tangent slots, produced beside the other
    chain's kernels, equal the same convolution run alone on the step's final inputs, bit for bit, and every
    repetition reproduces the first one's gradients.  (Guards the build against packed-fp32 vector instructions:
    with them the single-channel convolution returned wrong low halves of a register pair in one 16-lane row, only
    while other kernels ran beside it.)"""
    import pgan_tape
    B = 64
    eng, _, _ = make_engine(hip, B, "gp", "sigmoid", 31)
    o = eng.ops
    rs, rp = orc.synthetic_alignments(2 * B, 60)
    eng.X0[:2 * B].copy_(rs.to(DEV))
    eng.P0[:2 * B].copy_(rp.to(DEV))
    eng.tbuf.copy_(torch.from_numpy(np.random.Generator(np.random.PCG64(62)).random(B)).float().to(DEV))
    eng.critic_grads_resident(run_generator=False)
    torch.cuda.synchronize()
    pgan_tape.tape = entries = []
    try:
        eng.critic_grads_resident(run_generator=False)
    finally:
        pgan_tape.tape = None
    torch.cuda.synchronize()
    first = eng.D.gflat.clone()
    tmp = torch.empty_like(eng.dA[1][:B])
    NS, NT = eng.NS, eng.NT
    for rep in range(12):
        for k in range(1, eng.nS + 1):       # what the step itself fills starts as NaN
            eng.dA[k][2 * B:].fill_(float("nan"))
            eng.dH[k][2 * B:].fill_(float("nan"))
        eng.X0[2 * B:].fill_(float("nan"))
        torch.cuda.synchronize()
        pgan_tape.replay(entries)
        torch.cuda.synchronize()
        assert torch.equal(eng.D.gflat, first), rep
        for lo, hi in ((2 * B, NS), (NS, NT)):
            o.conv_down(eng.X0[lo:hi], eng.dconv_w(1), tmp)
            torch.cuda.synchronize()
            assert torch.equal(tmp, eng.dA[1][lo:hi]), (rep, lo, int((tmp != eng.dA[1][lo:hi]).sum()))


def _trainer(hip, B, loss="gp", critic_iter=2, seed=77, **kw):
    from pgan_trainer import FusedTrainer
    tr = FusedTrainer(hip, B, torch.device(DEV), loss=loss, critic_iter=critic_iter, seed=seed, **kw)
    tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
    tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
    tr.weights_changed()
    rs, rp = orc.synthetic_alignments(64, 9)
    tr.set_pool(rs, rp)
    return tr


@pytest.mark.parametrize("loss", ["gp", "div"])
def test_step_boundary_overlap_does_not_change_results(loss, monkeypatch):
    """The next critic step's input launches beside Adam (PGAN_INPUTS_AHEAD) and the loss / norm-gradient / head-gradient
    launches beside the last weight-gradient passes (PGAN_TAIL_BESIDE_WGRAD) only move launches between streams: with
    both switched off the weights, Adam moments, running statistics, losses, pool cursor and generator state after
    several tape-replayed iterations are the same bits."""
    from pgan_hip import HipOps
    ops = HipOps(torch.device(DEV))      # its own backend: the trainers' streams would use up the shared one's stream pool
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PGAN_INPUTS_AHEAD", mode)
        monkeypatch.setenv("PGAN_TAIL_BESIDE_WGRAD", mode)
        tr = _trainer(ops, 8, loss=loss, critic_iter=3)
        assert tr.eng.tail_beside_wgrad == (mode == "1")
        tr.use_tape = True
        for _ in range(5):
            tr.step()
        torch.cuda.synchronize()
        e = tr.eng
        out[mode] = dict(d=e.D.flat.clone(), dm=e.D.mflat.clone(), dv=e.D.vflat.clone(), g=e.G.flat.clone(),
                         gm=e.G.mflat.clone(), run=e.Gbuf.p["twod_conv_layers.1.running_var"].clone(),
                         rz=tr.rng_z.clone(), rt=tr.rng_t.clone(), cur=tr.cursor.clone())
        out[mode + "l"] = tr.losses()
    assert tuple(out["1l"]) == tuple(out["0l"])
    for k in out["1"]:
        assert torch.equal(out["1"][k], out["0"][k]), (k, float((out["1"][k].float() - out["0"][k].float()).abs().max()))


def test_capture_allocates_nothing_and_refuses_unreserved_streams(hip):
    """Graph capture: the allocator's request count does not move between begin and end (capture() checks it), the op
    backend is back out of capture mode afterwards, and a launch on a stream whose scratch was never reserved is an
    error while a capture is open instead of an allocation."""
    from pgan_hip import PganError
    tr = _trainer(hip, 8)
    tr.use_graph = True
    tr.step()          # eager pass + capture
    tr.step()          # replay
    torch.cuda.synchronize()
    assert tr.graph is not None and hip.capturing is False
    a = torch.randn(4, 64, 8, 8, device=DEV)
    h, st = torch.empty_like(a), torch.empty(4, 2, device=DEV)
    gam, bet = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    fresh = torch.cuda.Stream()
    while fresh.cuda_stream in hip._scratch:          # torch hands out pooled streams round-robin
        fresh = torch.cuda.Stream()
    hip.capturing = True
    try:
        with torch.cuda.stream(fresh), pytest.raises(PganError):
            hip.gn_fwd(a, gam, bet, h, st)
    finally:
        hip.capturing = False
    with torch.cuda.stream(fresh):
        hip.gn_fwd(a, gam, bet, h, st)       # outside a capture the scratch is made on demand
    torch.cuda.synchronize()


def test_tape_recorded_after_an_evaluation_forward(hip):
    """A launch tape recorded while the prepared weight planes are fresh (an evaluation forward ran after the last
    optimiser step) still starts by preparing them: replays follow the eager route instead of convolving with the
    planes of the weights at recording time."""
    B, n_it, lr = 8, 4, 5e-5
    out = {}
    for route in ("eager", "tape"):
        tr = _trainer(hip, B, seed=91)
        tr.iteration()
        xs, xp = torch.zeros(B, 1, 64, 64, device=DEV), torch.zeros(B, 1, 64, device=DEV)
        z = z_draw(3, B).to(DEV)
        snap = {k: v.clone() for k, v in tr.eng.Gbuf.p.items()}
        zkeep = tr.eng.Z.clone()
        tr.eng.gen_forward(z, xs, xp)       # leaves the generator's planes fresh, as the trainer's plotting forward does
        tr.eng.g_batches_tracked -= 1
        tr.eng.Z.copy_(zkeep)
        for k, v in snap.items():
            tr.eng.Gbuf.p[k].copy_(v)
        tr.use_tape = route == "tape"
        for _ in range(n_it):
            tr.step()
        torch.cuda.synchronize()
        out[route] = (tr.eng.D.flat.clone(), tr.eng.G.flat.clone(), tr.losses())
    for i in range(2):
        assert float((out["eager"][i] - out["tape"][i]).abs().max()) <= 2 * lr * 2 * (n_it + 1)
        assert rel_err(out["tape"][i], out["eager"][i]) < 1e-3
    for x, y in zip(out["eager"][2], out["tape"][2]):
        assert abs(x - y) <= 1e-2 * max(1.0, abs(x))


def test_tape_replay_reports_a_failed_call(hip):
    import pgan_tape
    from pgan_hip import PganError, _lib
    bad = [(_lib.pgan_affine.fn, (None, None, -1, 1.0, 0.0, None), True)]
    with pytest.raises(PganError):
        pgan_tape.replay(bad)


# ------------------------------------------------------------------------------------------------------------------
def test_device_replay_buffer_equals_host_deque(hip):
    """pgan_fakebuf_step driven with injected indices against FakeBuffer (the reference's deque semantics): 12 steps
    into a buffer of 3.5 batches, so it fills, wraps and overwrites its oldest entries."""
    import train_wgan_v2 as tw
    B, size = 8, 28
    buf_s, buf_p = torch.zeros(size, 1, 64, 64, device=DEV), torch.zeros(size, 1, 64, device=DEV)
    state = torch.zeros(4, dtype=torch.int32, device=DEV)
    slots = torch.zeros(B, dtype=torch.int32, device=DEV)
    rng_state = torch.tensor([5, 0, 0, 0], dtype=torch.int32, device=DEV)
    host = tw.FakeBuffer(size)
    rs = np.random.RandomState(3)
    g = torch.Generator().manual_seed(1)
    for step in range(12):
        ns, npos = torch.rand(B, 1, 64, 64, generator=g), torch.rand(B, 1, 64, generator=g)
        host.add(ns.numpy(), npos.numpy())
        ix = rs.choice(range(host.length), B, replace=False)
        want_s = torch.stack([torch.as_tensor(host.sites[u]) for u in ix])
        want_p = torch.stack([torch.as_tensor(host.pos[u]) for u in ix])
        out_s, out_p = torch.zeros(B, 1, 64, 64, device=DEV), torch.zeros(B, 1, 64, device=DEV)
        hip.fakebuf_step(buf_s, buf_p, state, slots, ns.to(DEV), npos.to(DEV), out_s, out_p, rng_state,
                         picks=torch.from_numpy(ix.astype(np.int32)).to(DEV))
        torch.cuda.synchronize()
        assert torch.equal(out_s.cpu(), want_s) and torch.equal(out_p.cpu(), want_p), step
        assert state.tolist()[:3] == [((step + 1) * B) % size, min(size, (step + 1) * B), 0]
    # an index outside the buffer raises the sticky flag
    hip.fakebuf_step(buf_s, buf_p, state, slots, ns.to(DEV), npos.to(DEV), out_s, out_p, rng_state,
                     picks=torch.full((B,), size, dtype=torch.int32, device=DEV))
    assert int(state[2]) == 1


def test_device_replay_buffer_draws_without_replacement(hip):
    """Random draws: B distinct valid slots every step, the Philox offset advances, and over many steps every logical
    position of a full buffer is drawn about equally often (within 5 sigma of B / size per step)."""
    B, size, steps = 16, 64, 400
    buf_s, buf_p = torch.zeros(size, 1, 64, 64, device=DEV), torch.zeros(size, 1, 64, device=DEV)
    state = torch.zeros(4, dtype=torch.int32, device=DEV)
    slots = torch.zeros(B, dtype=torch.int32, device=DEV)
    rng_state = torch.tensor([11, 0, 0, 0], dtype=torch.int32, device=DEV)
    out_s, out_p = torch.zeros(B, 1, 64, 64, device=DEV), torch.zeros(B, 1, 64, device=DEV)
    hits = np.zeros(size)
    for step in range(steps):
        ns = torch.full((B, 1, 64, 64), float(step), device=DEV)
        ns[:, 0, 0, 0] = torch.arange(B, device=DEV, dtype=torch.float32)
        hip.fakebuf_step(buf_s, buf_p, state, slots, ns, torch.zeros(B, 1, 64, device=DEV), out_s, out_p, rng_state)
        s = slots.cpu().numpy()
        assert len(set(s.tolist())) == B and s.min() >= 0 and s.max() < min(size, (step + 1) * B)
        # what came out is what sits in those slots: (step it was added at, row in that batch)
        age = (step - out_s[:, 0, 0, 1].cpu().numpy()).astype(int)
        assert age.min() >= 0 and age.max() < size // B
        if (step + 1) * B >= size:
            head = int(state[0])
            hits[(s - head) % size] += 1        # logical position = slot relative to the oldest entry
    n_full = steps - size // B + 1
    p = B / size
    assert np.all(np.abs(hits - n_full * p) < 5 * np.sqrt(n_full * p * (1 - p))), hits
    assert int(rng_state[2]) == steps * (B // 4)


@pytest.mark.parametrize("loss", ["div", "gp"])
def test_buffered_trainer_routes_agree(hip, loss):
    """--use_buffer on every route: eager, launch tape and step graph give the same draws (buffer state, slots, Philox
    offsets) and the same training up to rounding-noise +-lr steps."""
    B, n_it, lr = 8, 4, 5e-5
    out = {}
    for route in ("eager", "tape", "graph"):
        tr = _trainer(hip, B, loss=loss, seed=55)
        tr.enable_fake_buffer(3 * B)
        tr.use_graph, tr.use_tape = route == "graph", route == "tape"
        for _ in range(n_it):
            tr.step()
        torch.cuda.synchronize()
        assert (tr.graph is not None) == (route == "graph") and (tr.tape is not None) == (route == "tape")
        out[route] = dict(d=tr.eng.D.flat.clone(), g=tr.eng.G.flat.clone(), st=tr.buf_state.tolist(),
                          slots=tr.buf_slots.tolist(), rng=tr.rng_buf.tolist(), losses=tr.losses())
    a = out["eager"]
    assert a["st"][:3] == [(2 * n_it * B) % (3 * B), 3 * B, 0]
    for route in ("tape", "graph"):
        b = out[route]
        assert b["st"] == a["st"] and b["slots"] == a["slots"] and b["rng"] == a["rng"], route
        for k in ("d", "g"):
            assert float((a[k] - b[k]).abs().max()) <= 2 * lr * 2 * n_it and rel_err(b[k], a[k]) < 1e-3, (route, k)


# ------------------------------------------------------------------------------------------------------------------
def test_sample_genotypes_on_the_device(hip):
    """pgan_sample_genotypes: 0/1 output, Bernoulli means, every column polymorphic when variation is required, the
    forced individual drawn in proportion to its probability of the missing allele; unconstrained draws may leave
    monomorphic columns."""
    import train_wgan_v2 as tw
    random.seed(5)
    tw._GENOTYPE_RNG.clear()
    n = 300
    g = torch.Generator().manual_seed(2)
    p = torch.rand(n, 1, 64, 64, generator=g).to(DEV)
    out = tw.sample_genotypes_bernoulli(p)
    assert out.is_cuda and tuple(out.shape) == (n, 1, 64, 64)
    assert bool(((out == 0) | (out == 1)).all())
    colsum = out.sum(dim=2)
    assert bool(((colsum > 0) & (colsum < 64)).all())
    m, pm = float(out.mean()), float(p.mean())
    assert abs(m - pm) < 5 * (0.25 / p.numel()) ** 0.5 + 1e-4, (m, pm)
    err = float((out - p).mean(dim=0).abs().max())          # per-cell frequency follows its probability
    assert err < 6 * 0.5 / n ** 0.5
    # columns that almost surely come out all-ancestral: one derived individual is forced, in proportion to p
    q = torch.full((n, 1, 64, 64), 1e-4, device=DEV)
    q[:, :, 32:, :] = 3e-4
    out = tw.sample_genotypes_bernoulli(q)
    colsum = out.sum(dim=2)
    assert bool((colsum >= 1).all()) and float((colsum == 1).float().mean()) > 0.95
    upper = float(out[:, :, 32:, :].sum() / out.sum())
    assert abs(upper - 0.75) < 0.02, upper
    # all-derived columns: one ancestral individual is forced
    out = tw.sample_genotypes_bernoulli(1 - q)
    assert bool((out.sum(dim=2) <= 63).all())
    free = tw.sample_genotypes_bernoulli(q, require_variation=False)
    assert float((free.sum(dim=2) == 0).float().mean()) > 0.9
    # 128 individuals (the scaled networks' alignments)
    p128 = torch.rand(50, 1, 128, 128, generator=g).to(DEV)
    o128 = tw.sample_genotypes_bernoulli(p128)
    cs = o128.sum(dim=2)
    assert bool(((o128 == 0) | (o128 == 1)).all()) and bool(((cs > 0) & (cs < 128)).all())
    assert abs(float(o128.mean()) - float(p128.mean())) < 5 * (0.25 / p128.numel()) ** 0.5 + 1e-4
    lo = torch.full((50, 1, 128, 128), 1e-5, device=DEV)
    assert bool((tw.sample_genotypes_bernoulli(lo).sum(dim=2) >= 1).all())
    # two calls draw different numbers (the state advances)
    assert not torch.equal(tw.sample_genotypes_bernoulli(p), tw.sample_genotypes_bernoulli(p))
