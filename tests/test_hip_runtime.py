"""The step's runtime pieces on the GPU, each through the C-ABI: the device generator against the published
Philox4x32-10 (restated in numpy, tests/torch_ops.py), slices of a global draw / batch (what data-parallel ranks take),
per-batch label smoothing (/root/reference/src/train_wgan_v2.py:265-268) and monomorphic-site switching (:129-160) with
injected draws against the reference's statements, events and stream edges of the library, the native launch tape, the
Adam gradient scale."""
import ctypes
import random

import numpy as np
import pytest
import torch

import dcwgan_oracle as orc
import torch_ops as T
from test_hip_engine import DEV

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pgan_hip import HipOps
    return HipOps(torch.device(DEV))


def _state(seed, hi=0, off=0):
    return torch.tensor([seed, hi, off & 0x7FFFFFFF, off >> 31], dtype=torch.int32, device=DEV)


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [5, 64, 8192])
def test_device_generator_is_philox(hip, n):
    """pgan_philox: uniforms are the generator's words >> 8 times 2^-24, bit for bit; normals their Box-Muller pairs
    (to the accuracy of the device's logf / sincosf); the state's offset moves by the calls consumed."""
    for normal in (False, True):
        st = _state(1234, 3, 17)
        before = st.tolist()
        out = torch.empty(n, device=DEV)
        hip.philox(out, st, normal)
        torch.cuda.synchronize()
        want = T.philox_values(before, 0, n, normal)
        got = out.cpu().numpy()
        if normal:
            assert np.abs(got - want).max() < 4e-6
        else:
            assert np.array_equal(got, want) and got.min() >= 0 and got.max() < 1
        assert st.tolist() == [1234, 3, 17 + (n + 3) // 4, 0]


@pytest.mark.parametrize("skip,n,total", [(0, 12, 40), (12, 12, 40), (7, 9, 33), (1024, 1024, 8192)])
def test_slices_of_a_global_draw(hip, skip, n, total):
    """pgan_philox_slice: values [skip, skip + n) of a draw of `total`, bit-identical to that stretch of the whole draw,
    with the state advanced by the whole draw -- P ranks that call it with their own skip draw what one device would."""
    for normal in (False, True):
        whole, st0 = torch.empty(total, device=DEV), _state(99, 0, 5)
        hip.philox(whole, st0, normal)
        part, st1 = torch.full((n + 2,), -7.0, device=DEV), _state(99, 0, 5)
        hip.philox(part[1:n + 1], st1, normal, skip=skip, total=total)
        torch.cuda.synchronize()
        assert torch.equal(part[1:n + 1], whole[skip:skip + n]) and st0.tolist() == st1.tolist()
        assert float(part[0]) == -7.0 and float(part[-1]) == -7.0           # nothing outside the slice is written


def test_batch_rows_of_a_global_batch(hip):
    pool_s = torch.arange(10 * 64, dtype=torch.float32, device=DEV).reshape(10, 1, 8, 8).contiguous()
    pool_p = torch.arange(10 * 8, dtype=torch.float32, device=DEV).reshape(10, 1, 8).contiguous()
    cur = torch.tensor([7], dtype=torch.int32, device=DEV)
    out_s, out_p = torch.empty(2, 1, 8, 8, device=DEV), torch.empty(2, 1, 8, device=DEV)
    hip.take_batch(pool_s, pool_p, out_s, out_p, cur, offset=2, advance=4)       # rank 1 of 2, global batch 4
    torch.cuda.synchronize()
    assert torch.equal(out_s, pool_s[[9, 0]]) and torch.equal(out_p, pool_p[[9, 0]]) and int(cur) == 1


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("normalized", [False, True])
def test_label_smoothing_with_given_draws_equals_the_reference_statement(hip, normalized):
    """sites * (1 - ey) + 0.05 * ey (the loader's statement, on the raw 0 / 1 values; --normalize maps them afterwards)."""
    rs, _ = orc.synthetic_alignments(6, 3)
    ey = torch.from_numpy(np.random.default_rng(1).uniform(0.0, 0.05, tuple(rs.shape))).float()
    want = rs.double() * (1 - ey.double()) + 0.05 * ey.double()
    x = rs.clone()
    if normalized:
        x, want = (x - 0.5) / 0.5, (want - 0.5) / 0.5
    x = x.to(DEV).contiguous()
    hip.label_smooth(x, 0.05, 0.05, normalized, None, ey=ey.to(DEV))
    torch.cuda.synchronize()
    assert float((x.cpu().double() - want).abs().max()) < 2e-7


def test_label_smoothing_draws_are_the_generator_stream(hip):
    """Device draws: ey = noise * uniform of the Philox stream, value v from call v / 4; a rank's slice of a global batch
    equals that stretch of the whole batch's smoothing, and the next batch gets fresh noise."""
    rs, _ = orc.synthetic_alignments(4, 5)
    st = _state(555, 0, 0)
    whole = rs.to(DEV).contiguous()
    hip.label_smooth(whole, 0.05, 0.05, False, st)
    u = torch.from_numpy(T.philox_values([555, 0, 0, 0], 0, rs.numel(), False)).reshape(rs.shape)
    want = rs * (1 - 0.05 * u) + 0.05 * (0.05 * u)
    torch.cuda.synchronize()
    assert float((whole.cpu() - want).abs().max()) < 2e-7
    assert st.tolist() == [555, 0, rs.numel() // 4, 0]
    st2 = _state(555, 0, 0)
    half = rs[2:].to(DEV).contiguous()
    hip.label_smooth(half, 0.05, 0.05, False, st2, skip=2 * 4096, total=4 * 4096)
    again = rs.to(DEV).contiguous()
    hip.label_smooth(again, 0.05, 0.05, False, st2)              # the following batch
    torch.cuda.synchronize()
    assert torch.equal(half, whole[2:]) and st2.tolist() == [555, 0, 2 * (rs.numel() // 4), 0]
    assert not torch.equal(again, whole)


def _reference_switch(sites, pos, n, cols, u, monkeypatch):
    """The package's statement-for-statement restatement of switch_monomorphic, fed the given draws in the order it asks
    for them: random.randint per alignment, np.random.choice per alignment, np.random.uniform per chosen column."""
    import train_wgan_v2 as tw
    state = {"a": -1, "k": 0}

    def randint(lo, hi):
        state["a"] += 1
        state["k"] = 0
        return int(n[state["a"]])

    def choice(rng, size, replace=False):
        assert not replace and size == int(n[state["a"]])
        return np.array([int(c) for c in cols[state["a"]][:size]], dtype=np.int64)

    def uniform(lo, hi):
        v = float(lo) + (float(hi) - float(lo)) * float(u[state["a"]][state["k"]])
        state["k"] += 1
        return v
    monkeypatch.setattr(tw.random, "randint", randint)
    monkeypatch.setattr(tw.np.random, "choice", choice)
    monkeypatch.setattr(tw.np.random, "uniform", uniform)
    return tw.switch_monomorphic(sites, pos, 0, 10)


@pytest.mark.parametrize("normalized", [False, True])
def test_monomorphic_switch_with_given_draws_equals_the_reference_statements(hip, normalized, monkeypatch):
    """Per alignment: n columns, the first n // 2 all-ancestral (0.0) and the rest all-derived (1.0) -- these literals
    also on a normalised batch, as in the reference, whose switch runs after Normalize -- positions redrawn one after
    the other between the current neighbours; first and last columns, neighbouring columns and n = 0 / 1 / 10 included.
    Bit for bit."""
    B = 6
    rs, rp = orc.synthetic_alignments(B, 11)
    if normalized:
        rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
    rng = np.random.default_rng(7)
    n = np.array([0, 1, 10, 7, 4, 10], np.int32)
    cols = np.zeros((B, 10), np.int32)
    for a in range(B):
        cols[a] = rng.permutation(64)[:10]
    lead = lambda first, row: np.array(first + [c for c in row.tolist() + list(range(40, 50)) if c not in first][:10 - len(first)])
    cols[2] = lead([0, 63, 1, 62], cols[2])     # both ends and their neighbours
    cols[3] = lead([30, 31, 29], cols[3])       # neighbours redrawn in turn: the later ones see the earlier results
    cols[5] = np.arange(20, 30)[::-1]
    assert all(len(set(r.tolist())) == 10 for r in cols)
    u = rng.random((B, 10)).astype(np.float32)
    want_s, want_p = _reference_switch(rs.clone(), rp.clone(), n, cols, u, monkeypatch)
    s, p = rs.to(DEV).contiguous(), rp.to(DEV).contiguous()
    hip.mono_switch(s, p, 0, 10, None, given=(torch.from_numpy(n).to(DEV), torch.from_numpy(cols).to(DEV),
                                              torch.from_numpy(u).to(DEV)))
    torch.cuda.synchronize()
    assert torch.equal(s.cpu(), want_s) and torch.equal(p.cpu(), want_p)
    mono = (want_s.amax(dim=2) == want_s.amin(dim=2))[:, 0].sum(dim=1)
    assert mono.tolist() == n.tolist()


def test_monomorphic_switch_device_draws(hip):
    """The kernel's own draws are the generator stream's (restated in numpy: n = lower + floor(word (upper - lower + 1) /
    2^32), a partial Fisher-Yates over the columns, one uniform per column), keyed by the alignment's index in the global
    batch -- a rank's rows equal those rows of the whole batch -- and n covers its range with distinct columns."""
    B = 64
    rs, rp = orc.synthetic_alignments(B, 21)
    st = _state(4242, 0, 3)
    s, p = rs.to(DEV).contiguous(), rp.to(DEV).contiguous()
    hip.mono_switch(s, p, 0, 10, st)
    n, cols, u = T.mono_switch_draws([4242, 0, 3, 0], 0, B, 0, 10, 64)
    want_s, want_p = T.mono_switch_apply(rs.clone(), rp.clone(), n, cols, u)
    torch.cuda.synchronize()
    assert torch.equal(s.cpu(), want_s) and torch.equal(p.cpu(), want_p)
    assert st.tolist() == [4242, 0, 3 + B * T.MONO_CALLS, 0]
    assert set(n.tolist()) >= {0, 10} and n.min() == 0 and n.max() == 10
    assert all(len(set(cols[a][:n[a]].tolist())) == n[a] for a in range(B))
    st2 = _state(4242, 0, 3)
    s2, p2 = rs[16:32].to(DEV).contiguous(), rp[16:32].to(DEV).contiguous()
    hip.mono_switch(s2, p2, 0, 10, st2, first=16, total=B)
    torch.cuda.synchronize()
    assert torch.equal(s2, s[16:32]) and torch.equal(p2, p[16:32]) and st2.tolist() == st.tolist()


def test_augmentations_are_drawn_afresh_for_every_batch(hip):
    """FusedTrainer.enable_augment: the real batch a critic step trains on is smoothed and switched inside the step, with
    new draws every time the same alignments come round again (the reference draws them in get_batch / per critic batch;
    noise applied once to the pool would repeat)."""
    from pgan_trainer import FusedTrainer
    B = 8
    rs, rp = orc.synthetic_alignments(B, 9)            # pool of exactly one batch: every step sees the same alignments
    tr = FusedTrainer(hip, B, DEV, critic_iter=2, seed=5)
    tr.set_pool(rs, rp)
    tr.enable_augment(label_noise=0.05, mono_switch=(0, 10))
    seen = []
    for _ in range(3):
        tr._take_real()
        torch.cuda.synchronize()
        seen.append((tr.eng.X0[:B].clone(), tr.eng.P0[:B].clone()))
    for s, p in seen:
        x = s.cpu()
        assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
        # 0 -> 0.05 ey in [0, 0.0025], 1 -> 1 - 0.95 ey in [0.9525, 1]; switched columns are exactly 0 / 1
        frac = float(((x > 0.0026) & (x < 0.9524)).float().mean())
        assert frac == 0.0 and float(((x != 0) & (x != 1)).float().mean()) > 0.7
    assert not torch.equal(seen[0][0], seen[1][0]) and not torch.equal(seen[1][0], seen[2][0])
    assert not torch.equal(seen[0][1], seen[1][1])


# ------------------------------------------------------------------------------------------------------------------
def test_events_and_edges_order_streams(hip):
    """pgan_stream_edge / pgan_event_record / pgan_stream_wait_event on events that exist beforehand: a consumer on
    another stream sees the producer's result, over many reuses of the same events; none of the calls creates one."""
    import pgan_hip
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    ev, mark = pgan_hip.Event(), pgan_hip.Event()
    made = pgan_hip.events_created()
    x = torch.zeros(1 << 22, device=DEV)
    y = torch.zeros_like(x)
    torch.cuda.synchronize()
    for i in range(1, 21):
        with torch.cuda.stream(a):
            hip.affine(x, 1.0, 1.0, x)                 # x += 1 on stream a
        ev.edge(a.cuda_stream, b.cuda_stream)
        with torch.cuda.stream(b):
            hip.affine(x, 2.0, 0.0, y)                 # y = 2 x on stream b, after the increment
        mark.record(b.cuda_stream)
        mark.wait(a.cuda_stream)                       # the next increment waits for the read
    torch.cuda.synchronize()
    assert float(y.min()) == 40.0 and float(y.max()) == 40.0 and pgan_hip.events_created() == made


def test_native_tape_replays_library_calls(hip):
    """A run of enqueue calls noted through pgan_tape_append and replayed from C equals making the calls: floats, 64-bit
    sizes, pointers and host-array arguments all arrive intact; a call that fails stops the replay with its name."""
    import pgan_hip
    import pgan_tape
    p = torch.full((4096,), 1.0, device=DEV)
    g = torch.full((4096,), 0.5, device=DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    out = torch.zeros(4096, device=DEV)
    parts = [torch.rand(16, 64, device=DEV), torch.rand(16, 128, device=DEV)]
    sums = [torch.zeros(64, device=DEV), torch.zeros(128, device=DEV)]
    pgan_tape.tape = entries = []
    try:
        hip.adam(p, g, m, v, 3, 1e-3, 0.5, 0.999, 1e-8, grad_scale=0.5)
        hip.affine(p, 3.0, -1.0, out)
        hip.rows_reduce_multi(parts, sums, 2, 9, 11, 16)
    finally:
        pgan_tape.tape = None
    torch.cuda.synchronize()
    first = (p.clone(), out.clone(), sums[0].clone(), sums[1].clone())
    p.fill_(1.0), m.zero_(), v.zero_(), out.zero_(), sums[0].zero_(), sums[1].zero_()
    tape = pgan_tape.compile(entries)
    assert tape.python_steps == 0 and tape.native_calls == 3
    tape.replay()
    torch.cuda.synchronize()
    assert torch.equal(p, first[0]) and torch.equal(out, first[1])
    assert torch.equal(sums[0], first[2]) and torch.equal(sums[1], first[3])
    want = torch.cat([parts[0][2:9], parts[0][11:16]]).sum(0)
    assert torch.allclose(sums[0], want, rtol=1e-6)
    # Adam read g times 0.5: the same step as g = 0.25 at scale 1
    q, m2, v2 = torch.full((4096,), 1.0, device=DEV), torch.zeros(4096, device=DEV), torch.zeros(4096, device=DEV)
    hip.adam(q, g * 0.5, m2, v2, 3, 1e-3, 0.5, 0.999, 1e-8)
    torch.cuda.synchronize()
    assert torch.equal(q, first[0])
    bad = pgan_hip.NativeTape([(pgan_hip._lib.pgan_affine.fn, (p.data_ptr(), out.data_ptr(), 0, 1.0, 0.0, 0), True)])
    with pytest.raises(pgan_hip.PganError, match="pgan_affine"):
        bad.replay()
