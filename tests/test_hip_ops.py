"""Per-kernel parity on the GPU: every HIP op (through the C-ABI) against the plain-torch statement
of the same op (tests/torch_ops.py) evaluated in float64 on the CPU, same inputs.

Tolerance: 1e-5 norm-wise relative error in fp32 (the north star's bound), with the elementwise
error measured against the tensor's rms allowed 5e-5 for long reductions."""
import numpy as np
import pytest
import torch

from helpers import max_rel_err, rel_err
from torch_ops import TorchOps

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def hip():
    from pgan_hip import HipOps
    return HipOps(torch.device("cuda:0"), precision="x3")      # the split-kernel tests switch to "h2" themselves


REF = TorchOps()
DEV = "cuda:0"


def rnd(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def d64(*ts):
    return [None if t is None else ([d64(x)[0] for x in t] if isinstance(t, (list, tuple)) else t.double()) for t in ts]


def gpu(*ts):
    return [None if t is None else ([x.to(DEV) for x in t] if isinstance(t, (list, tuple)) else t.to(DEV)) for t in ts]


def close(name, got, want, rtol=RTOL, etol=5e-5):
    r, m = rel_err(got, want), max_rel_err(got, want)
    print(f"{name}: rel {r:.2e} max/rms {m:.2e}")
    assert r < rtol, f"{name}: norm-wise rel err {r:.3e}"
    assert m < etol, f"{name}: max err / rms {m:.3e}"


LAYERS = [(128, 64, 16), (256, 128, 8), (512, 256, 4)]   # (Cs, Cl, hs)
LAYERS_X3 = LAYERS + [(64, 32, 32)]                      # + the extra stage of the 128 x 128 networks (bf16 pipe only)


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS)
@pytest.mark.parametrize("N", [5, 8])
@pytest.mark.parametrize("kg", [0, 1, 2, 4, 12, 14])
def test_conv_down(hip, Cs, Cl, hs, N, kg):
    from pgan_hip import set_option
    rng = np.random.default_rng(1)
    x, w = rnd(rng, N, Cl, 2 * hs, 2 * hs), rnd(rng, Cs, Cl, 4, 4, scale=0.05)
    want = torch.zeros(N, Cs, hs, hs, dtype=torch.float64)
    REF.conv_down(*d64(x, w), want)
    got = torch.full((N, Cs, hs, hs), float("nan"), device=DEV)
    set_option("down_kg", kg % 10)           # 12 / 14: K also split over two blocks
    set_option("down_ks", 2 if kg >= 10 else (1 if kg else 0))
    hip.precision = "fp32"
    try:
        hip.conv_down(*gpu(x, w), got)
    finally:
        set_option("down_kg", 0)
        set_option("down_ks", 0)
        hip.precision = "x3"
    close("conv_down", got, want)


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS_X3)
@pytest.mark.parametrize("N", [5, 8])
@pytest.mark.parametrize("variant,ks", [(0, 0), (1, 1), (2, 1), (2, 2), (2, 3), (3, 1), (4, 1), (4, 3), (4, 4)])
@pytest.mark.parametrize("split", ["x3", "h2"])
def test_conv_down_bf16_pipe(hip, Cs, Cl, hs, N, variant, ks, split):
    """Split operands on the 16-bit matrix pipe (six products of three bf16 planes / three products of two fp16
    planes): same fp32 tolerance as the fp32-pipe kernels."""
    from pgan_hip import set_x3_option
    if Cs % 128 and variant in (1, 2):
        pytest.skip("128-row tiles need Cs % 128 == 0")
    rng = np.random.default_rng(1)
    x, w = rnd(rng, N, Cl, 2 * hs, 2 * hs), rnd(rng, Cs, Cl, 4, 4, scale=0.05)
    want = torch.zeros(N, Cs, hs, hs, dtype=torch.float64)
    REF.conv_down(*d64(x, w), want)
    got = torch.full((N, Cs, hs, hs), float("nan"), device=DEV)
    assert hip.precision == "x3"
    hip.precision = split
    set_x3_option("down_variant", variant)
    set_x3_option("down_ks", ks)
    try:
        hip.conv_down(*gpu(x, w), got)
    finally:
        set_x3_option("down_variant", 0)
        set_x3_option("down_ks", 0)
        hip.precision = "x3"
    close("conv_down_x3", got, want)


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS)
@pytest.mark.parametrize("N", [5, 8])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_conv_up(hip, Cs, Cl, hs, N, mode):
    from pgan_hip import set_option
    rng = np.random.default_rng(2)
    y, w = rnd(rng, N, Cs, hs, hs), rnd(rng, Cs, Cl, 4, 4, scale=0.05)
    want = torch.zeros(N, Cl, 2 * hs, 2 * hs, dtype=torch.float64)
    REF.conv_up(*d64(y, w), want)
    got = torch.full((N, Cl, 2 * hs, 2 * hs), float("nan"), device=DEV)
    set_option("up_mode", mode)
    hip.precision = "fp32"
    try:
        hip.conv_up(*gpu(y, w), got)
    finally:
        set_option("up_mode", 0)
        hip.precision = "x3"
    close("conv_up", got, want)


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS_X3)
@pytest.mark.parametrize("N", [5, 8])
@pytest.mark.parametrize("variant,ks", [(0, 0), (1, 1), (1, 4), (2, 1), (2, 2), (3, 1), (4, 1), (5, 1), (6, 1), (6, 2), (7, 1)])
@pytest.mark.parametrize("split", ["x3", "h2"])
def test_conv_up_bf16_pipe(hip, Cs, Cl, hs, N, variant, ks, split):
    from pgan_hip import set_x3_option
    rng = np.random.default_rng(2)
    y, w = rnd(rng, N, Cs, hs, hs), rnd(rng, Cs, Cl, 4, 4, scale=0.05)
    want = torch.zeros(N, Cl, 2 * hs, 2 * hs, dtype=torch.float64)
    REF.conv_up(*d64(y, w), want)
    got = torch.full((N, Cl, 2 * hs, 2 * hs), float("nan"), device=DEV)
    set_x3_option("up_variant", variant)
    set_x3_option("up_ks", ks)
    hip.precision = split
    try:
        hip.conv_up(*gpu(y, w), got)
    finally:
        set_x3_option("up_variant", 0)
        set_x3_option("up_ks", 0)
        hip.precision = "x3"
    close("conv_up_x3", got, want)


@pytest.mark.parametrize("split", ["x3", "h2"])
def test_bf16_pipe_extremes(hip, split):
    """Both splits keep fp32's exponent range (bf16 has it; the fp16 planes are scaled per block and per weight
    tensor): tiny and huge magnitudes, samples of very different magnitude side by side, exact zeros, and weights that
    are updated in place between calls (prepared planes must follow them)."""
    rng = np.random.default_rng(12)
    N, Cs, Cl, hs = 3, 128, 64, 16
    hip.precision = split
    try:
        _extremes(hip, rng, N, Cs, Cl, hs)
    finally:
        hip.precision = "x3"


def _extremes(hip, rng, N, Cs, Cl, hs):
    for scale_x, scale_w in [(1e-18, 1e-12), (1e15, 1e12), (1.0, 0.0), (3e-7, 2.0)]:
        x, w = rnd(rng, N, Cl, 2 * hs, 2 * hs, scale=scale_x), rnd(rng, Cs, Cl, 4, 4, scale=scale_w)
        want = torch.zeros(N, Cs, hs, hs, dtype=torch.float64)
        REF.conv_down(*d64(x, w), want)
        got = torch.full((N, Cs, hs, hs), float("nan"), device=DEV)
        hip.conv_down(*gpu(x, w), got)
        if scale_w == 0.0:
            assert float(got.abs().max()) == 0.0
        else:
            close(f"conv_down_x3 scale {scale_x:g}", got, want)
    x, w = rnd(rng, N, Cl, 2 * hs, 2 * hs), rnd(rng, Cs, Cl, 4, 4, scale=0.05)
    xg, wg = gpu(x, w)
    got = torch.empty(N, Cs, hs, hs, device=DEV)
    hip.conv_down(xg, wg, got)
    wg.mul_(-2.0)                      # in-place update, same storage
    hip.conv_down(xg, wg, got)
    want = torch.zeros(N, Cs, hs, hs, dtype=torch.float64)
    REF.conv_down(x.double(), -2.0 * w.double(), want)
    close("conv_down_x3 after in-place weight update", got, want)
    # samples whose magnitudes differ by 2^40: every sample keeps its own relative accuracy, in all three kernels
    mags = torch.tensor([1e-6, 1.0, 1e6]).view(N, 1, 1, 1)
    x, y = rnd(rng, N, Cl, 2 * hs, 2 * hs) * mags, rnd(rng, N, Cs, hs, hs) * mags
    w = rnd(rng, Cs, Cl, 4, 4, scale=0.05)
    want = torch.zeros(N, Cs, hs, hs, dtype=torch.float64)
    REF.conv_down(*d64(x, w), want)
    got = torch.empty(N, Cs, hs, hs, device=DEV)
    hip.conv_down(*gpu(x, w), got)
    wantu = torch.zeros(N, Cl, 2 * hs, 2 * hs, dtype=torch.float64)
    REF.conv_up(*d64(y, w), wantu)
    gotu = torch.empty(N, Cl, 2 * hs, 2 * hs, device=DEV)
    hip.conv_up(*gpu(y, w), gotu)
    for n in range(N):
        close(f"conv_down sample of magnitude {float(mags[n]):g}", got[n], want[n])
        close(f"conv_up sample of magnitude {float(mags[n]):g}", gotu[n], wantu[n])
    for a, b in ((x / mags, y), (x, y / mags), (x * 1e-20, y * 1e-12)):      # weight gradient: one operand, then both, spread out
        wantw = torch.zeros(Cs, Cl, 4, 4, dtype=torch.float64)
        REF.conv_wgrad(*d64(b, a), wantw, N, 0)
        gotw = torch.empty(Cs, Cl, 4, 4, device=DEV)
        hip.conv_wgrad(*gpu(b, a), gotw)
        close("conv_wgrad over samples of different magnitude", gotw, wantw)


@pytest.mark.parametrize("split", ["x3", "h2"])
def test_weights_prepared_together(hip, split):
    """prepare_weights: the prepared planes of several weight tensors built by one launch pair give the same
    convolution results, bit for bit, as planes built tensor by tensor."""
    rng = np.random.default_rng(31)
    N = 3
    layers = [(128, 64, 16), (256, 128, 8), (512, 256, 4)]
    xs = [rnd(rng, N, Cl, 2 * hs, 2 * hs).to(DEV) for _, Cl, hs in layers]
    ws = [rnd(rng, Cs, Cl, 4, 4, scale=0.05 * (i + 1)).to(DEV) for i, (Cs, Cl, _) in enumerate(layers)]
    hip.precision = split
    try:
        hip.invalidate_weight_planes(reset=True)
        one = []
        for x, w, (Cs, _, hs) in zip(xs, ws, layers):
            y = torch.empty(N, Cs, hs, hs, device=DEV)
            hip.conv_down(x, w, y)
            one.append(y)
        hip.invalidate_weight_planes(reset=True)
        hip.prepare_weights(ws)
        try:
            for x, w, (Cs, _, hs), want in zip(xs, ws, layers, one):
                y = torch.empty(N, Cs, hs, hs, device=DEV)
                hip.conv_down(x, w, y)
                assert torch.equal(y, want), (Cs, hs)
        finally:
            hip.release_weights()
    finally:
        hip.precision = "x3"
        hip.invalidate_weight_planes(reset=True)


def _channel_amax(t):
    """[C] largest magnitude per channel over samples and positions -- what the BatchNorm ops report."""
    return t.abs().amax([d for d in range(t.dim()) if d != 1])


def _slice_amax(t, slots=8):
    """[N, slots] largest magnitudes of the `slots` equal slices of every sample -- what the GroupNorm ops report."""
    return t.reshape(t.shape[0], slots, -1).abs().amax(2).contiguous()


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS_X3)
@pytest.mark.parametrize("N,keep,swap", [(5, 5, 0), (8, 4, 2), (33, 11, 11)])
@pytest.mark.parametrize("form", ["per_sample", "global"])
def test_fp16_split_with_operand_bounds(hip, Cs, Cl, hs, N, keep, swap, form):
    """The fp16-split kernels scaled by magnitude bounds from the producer (per sample, as GroupNorm reports them, or
    per channel over the whole tensor, as BatchNorm does) instead of by their own scan of the operands: same 1e-5
    bound against fp64; samples of very different magnitude keep their own accuracy with per-sample bounds; bounds
    that are loose by 2^3 cost nothing."""
    rng = np.random.default_rng(21)
    mags = torch.logspace(-5, 5, N).view(N, 1, 1, 1) if form == "per_sample" else torch.ones(N, 1, 1, 1)
    x, y = rnd(rng, N, Cl, 2 * hs, 2 * hs) * mags, rnd(rng, N, Cs, hs, hs) * mags
    w = rnd(rng, Cs, Cl, 4, 4, scale=0.05)
    xg, yg, wg = gpu(x, y, w)
    if form == "per_sample":
        bx, by = _slice_amax(xg), _slice_amax(yg) * 8.0
    else:
        bx, by = xg.abs().amax((0, 2, 3)).contiguous() * 8.0, yg.abs().amax((0, 2, 3)).contiguous()
    hip.precision = "h2"
    try:
        want = torch.zeros(N, Cs, hs, hs, dtype=torch.float64)
        REF.conv_down(*d64(x, w), want)
        got = torch.full((N, Cs, hs, hs), float("nan"), device=DEV)
        hip.conv_down(xg, wg, got, bound=bx)
        wantu = torch.zeros(N, Cl, 2 * hs, 2 * hs, dtype=torch.float64)
        REF.conv_up(*d64(y, w), wantu)
        gotu = torch.full((N, Cl, 2 * hs, 2 * hs), float("nan"), device=DEV)
        hip.conv_up(yg, wg, gotu, bound=by)
        for n in range(N):
            close(f"conv_down, bounded, sample {n}", got[n], want[n])
            close(f"conv_up, bounded, sample {n}", gotu[n], wantu[n])
        xw, yw = (x / mags), (y / mags)
        wantw = torch.zeros(Cs, Cl, 4, 4, dtype=torch.float64)
        REF.conv_wgrad(*d64(yw, xw), wantw, keep, swap)
        gotw = torch.full((Cs, Cl, 4, 4), float("nan"), device=DEV)
        xwg, ywg = gpu(xw, yw)
        if form == "per_sample":
            bs, bl = _slice_amax(ywg), _slice_amax(xwg) * 8.0
        else:
            bs, bl = ywg.abs().amax((0, 2, 3)).contiguous(), xwg.abs().amax((0, 2, 3)).contiguous()
        hip.conv_wgrad(ywg, xwg, gotw, keep, swap, bound_s=bs, bound_l=bl)
        close("conv_wgrad, bounded", gotw, wantw)
        hip.conv_wgrad(ywg, xwg, gotw, keep, swap, bound_s=bs)          # one operand bounded, the other scanned
        close("conv_wgrad, half bounded", gotw, wantw)
    finally:
        hip.precision = "x3"


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS_X3)
@pytest.mark.parametrize("N,keep,swap", [(5, 5, 0), (8, 4, 2), (12, 0, 6), (33, 11, 11)])
@pytest.mark.parametrize("pipe", ["fp32", "bf16", "h2"])
def test_conv_wgrad(hip, Cs, Cl, hs, N, keep, swap, pipe):
    if hs == 32 and pipe == "fp32":
        pytest.skip("the 128 x 128 stage exists on the bf16 pipe only")
    rng = np.random.default_rng(3)
    ya, x = rnd(rng, N, Cs, hs, hs), rnd(rng, N, Cl, 2 * hs, 2 * hs)
    want = torch.zeros(Cs, Cl, 4, 4, dtype=torch.float64)
    REF.conv_wgrad(*d64(ya, x), want, keep, swap)
    got = torch.full((Cs, Cl, 4, 4), float("nan"), device=DEV)
    hip.wgrad_x3 = pipe != "fp32"
    hip.precision = "h2" if pipe == "h2" else "x3"
    base = rnd(rng, Cs, Cl, 4, 4)
    acc = base.to(DEV)
    try:
        hip.conv_wgrad(*gpu(ya, x), got, keep, swap)
        hip.conv_wgrad(*gpu(ya, x), acc, keep, swap, accumulate=True)
    finally:
        hip.wgrad_x3 = True
        hip.precision = "x3"
    close("conv_wgrad", got, want)
    close("conv_wgrad.accumulate", acc, want + base.double())


CANCEL_REPORT = []


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS)
@pytest.mark.parametrize("kind", ["down", "up", "wgrad"])
@pytest.mark.parametrize("layout", ["far", "adjacent"])
def test_split_arithmetic_on_cancelling_sums(hip, Cs, Cl, hs, kind, layout):
    """Dot products that cancel to a thousandth of the sum of their terms' magnitudes: along the reduction axis one
    operand repeats its values in pairs (u, u) and the other holds (v, -v + eta) with eta ~ 1e-3 |v|, so every output
    is sum(u eta) while sum |x| |w| is a thousand times larger.  layout "adjacent": the two members of a pair are
    neighbours on the reduction axis, so running sums never grow (the best case for a pipe that rounds its accumulator:
    the fp32 matrix pipe); "far": they sit half the axis apart, so the running sum grows like a random walk before it
    cancels.  Errors against fp64 are measured relative to sum |x| |w| -- what a dot product's rounding scales with.
    Stated bounds: the two-plane fp16 split (22-bit operands, products exact, fp32 accumulators) stays below 2e-8 of
    sum |x| |w| (3 orders below the 1e-5 parity bound even at this 1000-fold cancellation) and within 2x of the fp32
    pipe where running sums grow; where they do not, the fp32 pipe's exact 24-bit operands win by 3-4x, which the report
    line shows.  The three-plane bf16 split (24-bit operands) is at least as accurate as the fp32 pipe in every case."""
    rng = np.random.default_rng(5)
    N = 6

    def paired(t, dim):          # (a0, a1, ...) -> pairs along `dim`, adjacent or half an axis apart
        return t.repeat_interleave(2, dim) if layout == "adjacent" else torch.cat([t, t], dim)

    def opposed(v, dim):
        other = -v + 1e-3 * v.abs().mean() * rnd(rng, *v.shape)
        if layout == "far":
            return torch.cat([v, other], dim)
        sh = list(v.shape)
        sh[dim] *= 2
        return torch.stack([v, other], dim + 1).reshape(sh)
    if kind == "wgrad":          # reduction over samples and pixels: samples in equal / opposite pairs
        x = paired(rnd(rng, N // 2, Cl, 2 * hs, 2 * hs), 0)
        ya = opposed(rnd(rng, N // 2, Cs, hs, hs), 0)
        want, mag = torch.zeros(Cs, Cl, 4, 4, dtype=torch.float64), torch.zeros(Cs, Cl, 4, 4, dtype=torch.float64)
        REF.conv_wgrad(*d64(ya, x), want)
        REF.conv_wgrad(*d64(ya.abs(), x.abs()), mag)
        run = lambda o, out: o.conv_wgrad(*gpu(ya, x), out)
    elif kind == "down":         # reduction over the large side's channels (and the taps)
        x = paired(rnd(rng, N, Cl // 2, 2 * hs, 2 * hs), 1)
        w = opposed(rnd(rng, Cs, Cl // 2, 4, 4), 1)
        want = torch.zeros(N, Cs, hs, hs, dtype=torch.float64)
        ref, run = REF.conv_down, (lambda o, out: o.conv_down(*gpu(x, w), out))
    else:                        # reduction over the small side's channels
        x = paired(rnd(rng, N, Cs // 2, hs, hs), 1)
        w = opposed(rnd(rng, Cs // 2, Cl, 4, 4), 0)
        want = torch.zeros(N, Cl, 2 * hs, 2 * hs, dtype=torch.float64)
        ref, run = REF.conv_up, (lambda o, out: o.conv_up(*gpu(x, w), out))
    if kind != "wgrad":
        mag = torch.zeros_like(want)
        ref(*d64(x, w), want)
        ref(*d64(x.abs(), w.abs()), mag)
    assert float(want.abs().mean()) < 5e-3 * float(mag.mean())          # the sums do cancel
    err = {}
    for pipe in ("fp32", "h2", "x3"):
        got = torch.full(tuple(want.shape), float("nan"), device=DEV)
        hip.precision, hip.wgrad_x3 = pipe, pipe != "fp32"
        try:
            run(hip, got)
            torch.cuda.synchronize()
        finally:
            hip.precision, hip.wgrad_x3 = "x3", True
        e = (got.cpu().double() - want) / mag.clamp_min(1e-300)
        err[pipe] = (float(e.pow(2).mean().sqrt()), float(e.abs().max()))
    line = (f"{kind:5s} Cs={Cs:3d} Cl={Cl:3d} hs={hs:2d} {layout:8s}: rms error / sum|x||w|  fp32 pipe {err['fp32'][0]:.2e}  "
            f"fp16 split {err['h2'][0]:.2e} ({err['h2'][0] / err['fp32'][0]:.2f}x)  bf16 split {err['x3'][0]:.2e} "
            f"({err['x3'][0] / err['fp32'][0]:.2f}x)")
    print(line)
    CANCEL_REPORT.append(line)
    assert err["h2"][0] <= 2e-8 and err["h2"][1] <= 2e-7, err
    assert err["x3"][0] <= 1.25 * err["fp32"][0] + 2e-10, err
    if layout == "far":
        assert err["h2"][0] <= 2 * err["fp32"][0] + 1e-9, err


def test_cancelling_sums_report():
    """Prints the table of the cancelling-sum measurements above (one line per convolution kind, layer shape and
    layout), for the record in the test log."""
    assert len(CANCEL_REPORT) == 18
    print("\n".join(CANCEL_REPORT))


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS_X3)
@pytest.mark.parametrize("bm,target", [(64, 64), (64, 1024), (128, 64), (128, 256), (128, 1024)])
def test_conv_wgrad_tiles(hip, Cs, Cl, hs, bm, target):
    """Every tile height of the fp16-split WGRAD kernel at every stage size, from one block per output tile (all chunks
    in one K range: the pipelined loop's long even / odd runs) to one chunk per block."""
    from pgan_hip import set_x3_option
    if bm == 128 and Cs % 128:
        pytest.skip("128-row tiles need Cs % 128 == 0")
    rng = np.random.default_rng(11)
    N, keep, swap = 21, 7, 7
    ya, x = rnd(rng, N, Cs, hs, hs), rnd(rng, N, Cl, 2 * hs, 2 * hs)
    want = torch.zeros(Cs, Cl, 4, 4, dtype=torch.float64)
    REF.conv_wgrad(*d64(ya, x), want, keep, swap)
    got = torch.full((Cs, Cl, 4, 4), float("nan"), device=DEV)
    hip.precision = "h2"
    try:
        set_x3_option("wg_bm", bm)
        set_x3_option("wg_target", target)
        hip.conv_wgrad(*gpu(ya, x), got, keep, swap)
    finally:
        set_x3_option("wg_bm", 0)
        set_x3_option("wg_target", 0)
        hip.precision = "x3"
    close("conv_wgrad", got, want)


@pytest.mark.parametrize("Cs,hs", [(64, 32), (32, 64)])
@pytest.mark.parametrize("N", [1, 3])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv_c1(hip, N, act, Cs, hs):
    rng = np.random.default_rng(4)
    S = 2 * hs
    x, w, y = rnd(rng, N, 1, S, S), rnd(rng, Cs, 1, 4, 4, scale=0.2), rnd(rng, N, Cs, hs, hs)
    want = torch.zeros(N, Cs, hs, hs, dtype=torch.float64)
    REF.conv_down(*d64(x, w), want)
    got = torch.full((N, Cs, hs, hs), float("nan"), device=DEV)
    hip.conv_down(*gpu(x, w), got)
    close("conv_down_c1", got, want)
    want = torch.zeros(N, 1, S, S, dtype=torch.float64)
    REF.conv_up(*d64(y, w * 0.2), want, act)
    got = torch.full((N, 1, S, S), float("nan"), device=DEV)
    hip.conv_up(*gpu(y, w * 0.2), got, act)
    close("conv_up_c1", got, want)
    want = torch.zeros(Cs, 1, 4, 4, dtype=torch.float64)
    REF.conv_wgrad(*d64(y, x), want, 1, 1 if N == 3 else 0)
    acc = torch.ones(Cs, 1, 4, 4, device=DEV)
    hip.conv_wgrad(*gpu(y, x), acc, 1, 1 if N == 3 else 0, accumulate=True)
    close("conv_wgrad_c1.accumulate", acc, want + 1)
    got = torch.full((Cs, 1, 4, 4), float("nan"), device=DEV)
    hip.conv_wgrad(*gpu(y, x), got, 1, 1 if N == 3 else 0)
    close("conv_wgrad_c1", got, want)


@pytest.mark.parametrize("N", [1, 3, 64, 130])
def test_first_stage_conv_with_groupnorm_sums(hip, N):
    """Discriminator stage 1 (src/layers.py:441-443): the convolution that leaves the following GroupNorm's slice sums in
    the stream's norm scratch, then the norm's apply launch alone -- against the fp64 statement of conv -> GroupNorm ->
    LeakyReLU, and the convolution output bit for bit against the plain kernel's."""
    rng = np.random.default_rng(40 + N)
    x, w = rnd(rng, N, 1, 64, 64), rnd(rng, 64, 1, 4, 4, scale=0.2)
    gamma, beta = 1 + 0.2 * rnd(rng, 64), 0.2 * rnd(rng, 64)
    a_w = torch.zeros(N, 64, 32, 32, dtype=torch.float64)
    REF.conv_down(*d64(x, w), a_w)
    h_w, st_w = torch.zeros_like(a_w), torch.zeros(N, 2, dtype=torch.float64)
    REF.gn_fwd(a_w, *d64(gamma, beta), h_w, st_w)
    nan = lambda *sh: torch.full(sh, float("nan"), device=DEV)
    a_g, a_plain, h_g, st_g, am = nan(N, 64, 32, 32), nan(N, 64, 32, 32), nan(N, 64, 32, 32), nan(N, 2), nan(N, 8)
    xg, wg = gpu(x, w)
    assert hip.conv_down_gn_sums(xg, wg, a_g) is True
    hip.gn_fwd(a_g, *gpu(gamma, beta), h_g, st_g, amax=am, presummed=True)
    hip.conv_down(xg, wg, a_plain)
    torch.cuda.synchronize()
    assert torch.equal(a_g, a_plain)
    close("conv_down_gn_sums.a", a_g, a_w)
    close("gn_fwd_presummed.h", h_g, h_w)
    close("gn_fwd_presummed.stats", st_g, st_w)
    assert torch.equal(am.amax(1), h_g.abs().flatten(1).amax(1))
    # shapes the fused kernel does not take fall back to the plain convolution
    x2, w2 = gpu(rnd(rng, 2, 1, 128, 128), rnd(rng, 32, 1, 4, 4, scale=0.2))
    a2 = nan(2, 32, 64, 64)
    assert hip.conv_down_gn_sums(x2, w2, a2) is False and not torch.isnan(a2).any()


GN_SHAPES = [(64, 32), (128, 16), (256, 8), (512, 4), (32, 64)]


@pytest.mark.parametrize("mask_from", ["h", "a"])
@pytest.mark.parametrize("C,hs", GN_SHAPES + [(256, 4)])
def test_groupnorm_family(hip, C, hs, mask_from):
    """mask_from = "a": the backward ops get beta and recompute the LeakyReLU mask from a; they are handed no h at all."""
    rng = np.random.default_rng(5)
    N = 3
    sh = (N, C, hs, hs)
    a = rnd(rng, *sh) * 0.7 + 0.3
    gamma, beta = 1 + 0.2 * rnd(rng, C), 0.2 * rnd(rng, C)
    u, t, hbar = rnd(rng, *sh), rnd(rng, *sh), rnd(rng, *sh)
    # forward
    h_w, st_w = torch.zeros(sh, dtype=torch.float64), torch.zeros(N, 2, dtype=torch.float64)
    REF.gn_fwd(*d64(a, gamma, beta), h_w, st_w)
    h_g, st_g = torch.zeros(sh, device=DEV), torch.zeros(N, 2, device=DEV)
    am = torch.full((N, 8), float("nan"), device=DEV)
    hip.gn_fwd(*gpu(a, gamma, beta), h_g, st_g, amax=am)
    assert torch.equal(am, _slice_amax(h_g))          # the reported magnitudes are those of the output, exactly
    close("gn_fwd.h", h_g, h_w)
    close("gn_fwd.stats", st_g, st_w)
    h32 = h_g.cpu()   # masks come from the GPU's own forward output
    h_in, mk = (h_g, {}) if mask_from == "h" else (None, dict(beta=beta.to(DEV)))
    # first-order backward
    d_w, sb_w = torch.zeros(sh, dtype=torch.float64), torch.zeros(N, 2, dtype=torch.float64)
    pg_w, pb_w = torch.zeros(N, C, dtype=torch.float64), torch.zeros(N, C, dtype=torch.float64)
    REF.gn_bwd(*d64(u, h32, a), st_w, gamma.double(), d_w, sb_w, pg_w, pb_w)
    d_g, sb_g = torch.zeros(sh, device=DEV), torch.zeros(N, 2, device=DEV)
    pg_g, pb_g = torch.zeros(N, C, device=DEV), torch.zeros(N, C, device=DEV)
    hip.gn_bwd(*gpu(u), h_in, a.to(DEV), st_g, gamma.to(DEV), d_g, sb_g, pg_g, pb_g, amax=am, **mk)
    assert torch.equal(am, _slice_amax(d_g))
    close("gn_bwd.d", d_g, d_w)
    close("gn_bwd.pg", pg_g, pg_w)
    close("gn_bwd.pb", pb_g, pb_w)
    # tangent
    hd_w, stt_w = torch.zeros(sh, dtype=torch.float64), torch.zeros(N, 2, dtype=torch.float64)
    REF.gn_jvp(*d64(t, h32, a), st_w, gamma.double(), hd_w, stt_w)
    hd_g, stt_g = torch.zeros(sh, device=DEV), torch.zeros(N, 2, device=DEV)
    hip.gn_jvp(t.to(DEV), h_in, a.to(DEV), st_g, gamma.to(DEV), hd_g, stt_g, amax=am, **mk)
    assert torch.equal(am, _slice_amax(hd_g))
    close("gn_jvp.hd", hd_g, hd_w)
    # second-order backward, with and without an incoming primal adjoint
    for hb in (hbar, None):
        p_w = torch.zeros(sh, dtype=torch.float64)
        REF.gn_bwd2(None if hb is None else hb.double(), *d64(u, t, h32, a), st_w, sb_w, stt_w, gamma.double(), p_w,
                    pg_w, pb_w)
        p_g = torch.zeros(sh, device=DEV)
        hip.gn_bwd2(None if hb is None else hb.to(DEV), u.to(DEV), t.to(DEV), h_in, a.to(DEV), st_g, sb_g, stt_g,
                    gamma.to(DEV), p_g, pg_g, pb_g, amax=am, **mk)
        assert torch.equal(am, _slice_amax(p_g))
        close("gn_bwd2.p", p_g, p_w, rtol=2e-5)
        close("gn_bwd2.pg", pg_g, pg_w, rtol=2e-5)
        if hb is not None:
            close("gn_bwd2.pb", pb_g, pb_w)
        else:
            assert float(pb_g.abs().max()) == 0.0


def test_rows_reduce_multi(hip):
    """gamma | beta partial sums written side by side by gn_bwd, then every stage reduced in one launch."""
    rng = np.random.default_rng(11)
    N = 7
    parts, outs, wants = [], [], []
    for C, hs in GN_SHAPES[:4]:           # one launch takes up to four stages
        sh = (N, C, hs, hs)
        a, u, gamma, beta = rnd(rng, *sh), rnd(rng, *sh), 1 + 0.2 * rnd(rng, C), 0.2 * rnd(rng, C)
        h_g, st_g = torch.zeros(sh, device=DEV), torch.zeros(N, 2, device=DEV)
        hip.gn_fwd(*gpu(a, gamma, beta), h_g, st_g)
        pgb = torch.full((N, 2 * C), float("nan"), device=DEV)
        hip.gn_bwd(u.to(DEV), h_g, a.to(DEV), st_g, gamma.to(DEV), torch.zeros(sh, device=DEV),
                   torch.zeros(N, 2, device=DEV), pgb[:, :C], pgb[:, C:])
        pg_w, pb_w = torch.zeros(N, C, dtype=torch.float64), torch.zeros(N, C, dtype=torch.float64)
        REF.gn_bwd(*d64(u, h_g.cpu(), a), st_g.cpu().double(), gamma.double(), torch.zeros(sh, dtype=torch.float64),
                   torch.zeros(N, 2, dtype=torch.float64), pg_w, pb_w)
        close("pgb", pgb, torch.cat([pg_w, pb_w], 1))
        parts.append(pgb)
        outs.append(torch.full((2 * C,), float("nan"), device=DEV))
        wants.append(torch.cat([pg_w, pb_w], 1))
    for rows in [(0, 2, 5, 7), (3, 7, 0, 0), (0, N, 0, 0)]:
        hip.rows_reduce_multi(parts, outs, *rows)
        for o, w in zip(outs, wants):
            close("rows_reduce_multi", o, w[rows[0]:rows[1]].sum(0) + w[rows[2]:rows[3]].sum(0))
    hip.rows_reduce_multi(parts[:2], outs[:2], 1, 2, 0, 0)
    close("rows_reduce_multi2", outs[1], wants[1][1])


@pytest.mark.parametrize("shape", [(6, 512, 4, 4), (6, 64, 32, 32), (6, 64, 4), (6, 8, 32), (16, 64, 32, 32), (24, 128, 16, 16)])
def test_batchnorm(hip, shape):
    rng = np.random.default_rng(6)
    C = shape[1]
    a, dh = rnd(rng, *shape) * 0.5 + 0.2, rnd(rng, *shape)
    gamma, beta = 1 + 0.2 * rnd(rng, C), 0.2 * rnd(rng, C)
    rm, rv = 0.1 * rnd(rng, C), 1 + 0.1 * rnd(rng, C).abs()
    h_w, st_w = torch.zeros(shape, dtype=torch.float64), torch.zeros(2, C, dtype=torch.float64)
    rm_w, rv_w = rm.double().clone(), rv.double().clone()
    REF.bn_fwd(*d64(a, gamma, beta), rm_w, rv_w, h_w, st_w)
    h_g, st_g = torch.zeros(shape, device=DEV), torch.zeros(2, C, device=DEV)
    rm_g, rv_g = rm.to(DEV), rv.to(DEV)
    am = torch.full((C,), float("nan"), device=DEV)
    hip.bn_fwd(*gpu(a, gamma, beta), rm_g, rv_g, h_g, st_g, amax=am)
    assert torch.equal(am, _channel_amax(h_g))          # largest output magnitude per channel, exactly
    close("bn_fwd.h", h_g, h_w)
    close("bn_fwd.rmean", rm_g, rm_w)
    close("bn_fwd.rvar", rv_g, rv_w)
    da_w = torch.zeros(shape, dtype=torch.float64)
    dg_w, db_w = torch.zeros(C, dtype=torch.float64), torch.zeros(C, dtype=torch.float64)
    REF.bn_bwd(*d64(dh, h_g.cpu(), a), st_w, gamma.double(), da_w, dg_w, db_w)
    da_g, dg_g, db_g = torch.zeros(shape, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    hip.bn_bwd(dh.to(DEV), h_g, a.to(DEV), st_g, gamma.to(DEV), da_g, dg_g, db_g, amax=am)
    assert torch.equal(am, _channel_amax(da_g))
    close("bn_bwd.da", da_g, da_w)
    close("bn_bwd.dgamma", dg_g, dg_w)
    close("bn_bwd.dbeta", db_g, db_w)


@pytest.mark.parametrize("groups,shape", [(3, (8, 512, 4, 4)), (2, (16, 64, 32, 32)), (6, (4, 64, 4)), (3, (8, 128, 16, 16)),
                                          (5, (64, 256, 8, 8)), (5, (64, 8, 32)), (1, (8, 128, 16, 16))])
def test_batchnorm_forward_passes_in_one_call(hip, groups, shape):
    """groups forward passes stacked along the batch axis: each normalised with its own statistics, running
    statistics stepped through them in order -- as that many separate calls (fused, sliced and split entry points)."""
    rng = np.random.default_rng(26)
    B, C = shape[:2]
    full = (groups * B,) + tuple(shape[1:])
    a = rnd(rng, *full) * 0.5 + 0.2 + torch.arange(groups).repeat_interleave(B).view(-1, *([1] * (len(shape) - 1))) * 0.3
    gamma, beta = 1 + 0.2 * rnd(rng, C), 0.2 * rnd(rng, C)
    rm, rv = 0.1 * rnd(rng, C), 1 + 0.1 * rnd(rng, C).abs()
    h_w, st_w = torch.zeros(full, dtype=torch.float64), torch.zeros(groups, 2, C, dtype=torch.float64)
    rm_w, rv_w = rm.double().clone(), rv.double().clone()
    for g in range(groups):
        REF.bn_fwd(*d64(a[g * B:(g + 1) * B], gamma, beta), rm_w, rv_w, h_w[g * B:(g + 1) * B], st_w[g])
    h_g, st_g = torch.zeros(full, device=DEV), torch.zeros(groups, 2, C, device=DEV)
    rm_g, rv_g = rm.to(DEV), rv.to(DEV)
    am = torch.full((C,), float("nan"), device=DEV)
    hip.bn_fwd(*gpu(a, gamma, beta), rm_g, rv_g, h_g, st_g, groups=groups, amax=am)
    assert torch.equal(am, _channel_amax(h_g))
    close("bn_groups.h", h_g, h_w)
    close("bn_groups.stats", st_g, st_w)
    close("bn_groups.rmean", rm_g, rm_w)
    close("bn_groups.rvar", rv_g, rv_w)
    hw = int(np.prod(shape[2:]))
    part = torch.zeros(hip.bn_fwd_partials(B, C, hw, groups), dtype=torch.float64, device=DEV)
    h_s, st_s = torch.zeros(full, device=DEV), torch.zeros(groups, 2, C, device=DEV)
    rm_s, rv_s = rm.to(DEV), rv.to(DEV)
    am.fill_(float("nan"))
    hip.bn_fwd_sums(a.to(DEV), part, groups=groups, amax=am)
    hip.bn_fwd_apply(*gpu(a, gamma, beta), rm_s, rv_s, h_s, st_s, part, 1, groups=groups, amax=am)
    assert torch.equal(am, _channel_amax(h_s))
    close("bn_groups.split.h", h_s, h_w)
    close("bn_groups.split.rmean", rm_s, rm_w)
    close("bn_groups.split.rvar", rv_s, rv_w)
    close("bn_groups.split.stats", st_s, st_w)


@pytest.mark.parametrize("shape", [(8, 512, 4, 4), (32, 64, 32, 32), (8, 64, 4), (16, 128, 16, 16)])
def test_batchnorm_over_two_ranks(hip, shape):
    """Split entry points: two "ranks" of B/2 samples whose partial sums are added equal one BatchNorm over B."""
    rng = np.random.default_rng(16)
    B, C = shape[:2]
    a, dh = rnd(rng, *shape) * 0.5 + 0.2, rnd(rng, *shape)
    gamma, beta = 1 + 0.2 * rnd(rng, C), 0.2 * rnd(rng, C)
    rm, rv = 0.1 * rnd(rng, C), 1 + 0.1 * rnd(rng, C).abs()
    h_w, st_w = torch.zeros(shape, dtype=torch.float64), torch.zeros(2, C, dtype=torch.float64)
    rm_w, rv_w = rm.double().clone(), rv.double().clone()
    REF.bn_fwd(*d64(a, gamma, beta), rm_w, rv_w, h_w, st_w)
    halves = [slice(0, B // 2), slice(B // 2, B)]
    ag, dhg = a.to(DEV), dh.to(DEV)
    hw = int(np.prod(shape[2:]))
    n = max(hip.bn_partials(B // 2, C, hw), hip.bn_fwd_partials(B // 2, C, hw))
    parts = [torch.zeros(n, dtype=torch.float64, device=DEV) for _ in halves]
    for sl, part in zip(halves, parts):
        hip.bn_fwd_sums(ag[sl], part)
    total = parts[0] + parts[1]
    h_g = torch.zeros(shape, device=DEV)
    stats, run = [], []
    for sl in halves:
        st_g, rm_g, rv_g = torch.zeros(2, C, device=DEV), rm.to(DEV), rv.to(DEV)
        hip.bn_fwd_apply(ag[sl], *gpu(gamma, beta), rm_g, rv_g, h_g[sl], st_g, total, 2)
        stats.append(st_g)
        run.append((rm_g, rv_g))
    close("bn2.h", h_g, h_w)
    for rm_g, rv_g in run:
        close("bn2.rmean", rm_g, rm_w)
        close("bn2.rvar", rv_g, rv_w)
    assert torch.equal(stats[0], stats[1])
    da_w = torch.zeros(shape, dtype=torch.float64)
    dg_w, db_w = torch.zeros(C, dtype=torch.float64), torch.zeros(C, dtype=torch.float64)
    REF.bn_bwd(*d64(dh, h_g.cpu(), a), st_w, gamma.double(), da_w, dg_w, db_w)
    ams = [torch.full((C,), float("nan"), device=DEV) for _ in halves]
    for sl, part, am in zip(halves, parts, ams):
        hip.bn_bwd_sums(dhg[sl], h_g[sl], ag[sl], stats[0], part, amax=am)
    total = parts[0] + parts[1]
    da_g = torch.zeros(shape, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    for sl, am in zip(halves, ams):
        dg_g, db_g = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        hip.bn_bwd_apply(dhg[sl], h_g[sl], ag[sl], stats[0], gamma.to(DEV), da_g[sl], dg_g, db_g, total, 2, amax=am)
        assert torch.equal(am, _channel_amax(da_g[sl]))
        dg += dg_g
        db += db_g
    close("bn2.da", da_g, da_w)
    close("bn2.dgamma", dg, dg_w)      # the ranks' values add up to the total
    close("bn2.dbeta", db, db_w)


def test_head_and_penalties(hip):
    rng = np.random.default_rng(7)
    N, K2, K1 = 6, 8192, 3520
    h4, f1 = rnd(rng, N, 512, 4, 4), rnd(rng, N, K1)
    w, b, coef = rnd(rng, 1, K2 + K1, scale=0.01), rnd(rng, 1), rnd(rng, N)
    want = torch.zeros(N, dtype=torch.float64)
    REF.head_fwd(*d64(h4, f1, w, b), want)
    got = torch.zeros(N, device=DEV)
    hip.head_fwd(*gpu(h4, f1, w, b), got)
    close("head_fwd", got, want)
    want = torch.zeros(N, 512, 4, 4, dtype=torch.float64)
    REF.head_seed(*d64(w, coef), want)
    got = torch.zeros(N, 512, 4, 4, device=DEV)
    hip.head_seed(*gpu(w, coef), got)
    close("head_seed", got, want)
    want = torch.zeros(K2, dtype=torch.float64)
    REF.head_wgrad(*d64(h4, coef), want)
    got = torch.zeros(K2, device=DEV)
    hip.head_wgrad(*gpu(h4, coef), got)
    close("head_wgrad", got, want)
    # interpolate / gp / div
    B = 4
    real, fake, t = rnd(rng, B, 1, 64, 64), rnd(rng, B, 1, 64, 64), torch.from_numpy(rng.random(B).astype(np.float32))
    want = torch.zeros(B, 1, 64, 64, dtype=torch.float64)
    REF.interpolate(*d64(t, real, fake), want)
    got = torch.zeros(B, 1, 64, 64, device=DEV)
    hip.interpolate(*gpu(t, real, fake), got)
    close("interpolate", got, want)
    g = rnd(rng, B, 1, 64, 64, scale=0.02)
    v_w, pen_w = torch.zeros(B, 1, 64, 64, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
    REF.gp_penalty(g.double(), 10.0, v_w, pen_w)
    v_g, pen_g = torch.zeros(B, 1, 64, 64, device=DEV), torch.zeros(1, device=DEV)
    hip.gp_penalty(g.to(DEV), 10.0, v_g, pen_g)
    close("gp.v", v_g, v_w)
    close("gp.pen", pen_g, pen_w)
    REF.div_penalty(g.double(), B // 2, v_w, pen_w)
    hip.div_penalty(g.to(DEV), B // 2, v_g, pen_g)
    close("div.v", v_g, v_w)
    close("div.pen", pen_g, pen_w)
    score, pen = rnd(rng, 2 * B), rnd(rng, 1)
    out_w = torch.zeros(4, dtype=torch.float64)
    REF.critic_loss(score.double(), B, pen.double(), out_w)
    out_g = torch.zeros(4, device=DEV)
    hip.critic_loss(score.to(DEV), B, pen.to(DEV), out_g)
    close("critic_loss", out_g, out_w)
    o_w, o_g = torch.zeros(1, dtype=torch.float64), torch.zeros(1, device=DEV)
    REF.neg_mean(score.double(), o_w)
    hip.neg_mean(score.to(DEV), o_g)
    close("neg_mean", o_g, o_w)
    y = torch.sigmoid(rnd(rng, B, 1, 64, 64))
    for act in (1, 2):
        dx_w, dx_g = torch.zeros(B, 1, 64, 64, dtype=torch.float64), torch.zeros(B, 1, 64, 64, device=DEV)
        REF.act_bwd(*d64(real, y), dx_w, act)
        hip.act_bwd(*gpu(real, y), dx_g, act)
        close("act_bwd", dx_g, dx_w)
    a_w, a_g = torch.zeros(B, 1, 64, 64, dtype=torch.float64), torch.zeros(B, 1, 64, 64, device=DEV)
    REF.affine(real.double(), 2.0, -1.0, a_w)
    hip.affine(real.to(DEV), 2.0, -1.0, a_g)
    close("affine", a_g, a_w)


def test_adam(hip):
    rng = np.random.default_rng(8)
    n = 4096 + 64
    p, g, m = rnd(rng, n, scale=0.05), rnd(rng, n, scale=1e-3), rnd(rng, n, scale=1e-3)
    v = rnd(rng, n, scale=1e-3).abs() * 1e-3
    pw, mw, vw = p.double().clone(), m.double().clone(), v.double().clone()
    pg_, mg, vg = p.to(DEV), m.to(DEV), v.to(DEV)
    step_dev = torch.full((1,), 7, dtype=torch.int32, device=DEV)
    for step in (7, 8):
        REF.adam(pw, g.double(), mw, vw, step, 5e-5, 0.5, 0.999, 1e-8)
    hip.adam(pg_, g.to(DEV), mg, vg, 7, 5e-5, 0.5, 0.999, 1e-8)             # step count from the host
    hip.adam(pg_, g.to(DEV), mg, vg, 0, 5e-5, 0.5, 0.999, 1e-8, step_dev)   # device counter 7 -> 8
    close("adam.p", pg_, pw, rtol=1e-6, etol=1e-5)
    close("adam.m", mg, mw)
    close("adam.v", vg, vw)
    assert int(step_dev) == 8


def test_philox(hip):
    state = torch.tensor([1234, 0, 0, 0], dtype=torch.int32, device=DEV)
    out = torch.zeros(1 << 20, device=DEV)
    hip.philox(out, state, True)
    assert abs(float(out.mean())) < 5e-3 and abs(float(out.std()) - 1.0) < 5e-3
    assert int(state[2]) == (1 << 20) // 4
    out2 = torch.zeros(1 << 20, device=DEV)
    hip.philox(out2, state, False)
    assert 0.0 <= float(out2.min()) and float(out2.max()) < 1.0 and abs(float(out2.mean()) - 0.5) < 2e-3
    state.copy_(torch.tensor([1234, 0, 0, 0], dtype=torch.int32))
    out3 = torch.zeros(1 << 20, device=DEV)
    hip.philox(out3, state, True)
    assert torch.equal(out, out3)


@pytest.mark.parametrize("L", [64, 128])
def test_d1d_tower(hip, L):
    rng = np.random.default_rng(9)
    N = 5
    ch, ln = (1, 16, 32, 64), (L, L - 3, L - 6, L - 9)
    F1 = 64 * (L - 9)
    xp = torch.sort(torch.from_numpy(rng.random((N, 1, L)).astype(np.float32)), dim=-1).values
    cw = [rnd(rng, ch[i + 1], ch[i], 4, scale=0.3) for i in range(3)]
    cb = [rnd(rng, ch[i + 1], scale=0.1) for i in range(3)]
    gw = [1 + 0.2 * rnd(rng, ch[i + 1]) for i in range(3)]
    gb = [0.2 * rnd(rng, ch[i + 1]) for i in range(3)]
    coef, w_tail = rnd(rng, N), rnd(rng, F1, scale=0.02)
    a_w = [torch.zeros(N, ch[i + 1], ln[i + 1], dtype=torch.float64) for i in range(3)]
    f_w = torch.zeros(N, F1, dtype=torch.float64)
    REF.d1d_fwd(*d64(xp, cw, cb, gw, gb), a_w, f_w)
    a_g = [torch.zeros(N, ch[i + 1], ln[i + 1], device=DEV) for i in range(3)]
    f_g = torch.zeros(N, F1, device=DEV)
    hip.d1d_fwd(*gpu(xp, cw, cb, gw, gb), a_g, f_g)
    close("d1d_fwd.feat", f_g, f_w)
    for i in range(3):
        close(f"d1d_fwd.a{i}", a_g[i], a_w[i])
    zw = lambda ts: [torch.zeros_like(t, dtype=torch.float64) for t in ts]
    zg = lambda ts: [torch.zeros_like(t, device=DEV) for t in ts]
    dw_w, db_w, dgw_w, dgb_w = zw(cw), zw(cb), zw(gw), zw(gb)
    dx_w = torch.zeros(N, 1, L, dtype=torch.float64)
    REF.d1d_bwd(*d64(xp, cw, cb, gw, gb), a_w, coef.double(), w_tail.double(), dw_w, db_w, dgw_w, dgb_w, dx_w)
    dw_g, db_g, dgw_g, dgb_g = zg(cw), zg(cb), zg(gw), zg(gb)
    dx_g = torch.zeros(N, 1, L, device=DEV)
    hip.d1d_bwd(*gpu(xp, cw, cb, gw, gb), a_g, coef.to(DEV), w_tail.to(DEV), dw_g, db_g, dgw_g, dgb_g, dx_g)
    close("d1d_bwd.dx", dx_g, dx_w, rtol=2e-5)
    for i in range(3):
        close(f"d1d_bwd.cw{i}", dw_g[i], dw_w[i], rtol=2e-5)
        close(f"d1d_bwd.cb{i}", db_g[i], db_w[i], rtol=2e-5)
        close(f"d1d_bwd.gw{i}", dgw_g[i], dgw_w[i], rtol=2e-5)
        close(f"d1d_bwd.gb{i}", dgb_g[i], dgb_w[i], rtol=2e-5)


def test_generator_small_ops(hip):
    rng = np.random.default_rng(10)
    N = 6
    for Cs, Cl, Ls, act in [(64, 32, 4, 0), (16, 8, 16, 0), (8, 1, 32, 1), (8, 1, 32, 2)]:
        y, w, x = rnd(rng, N, Cs, Ls), rnd(rng, Cs, Cl, 4, scale=0.2), rnd(rng, N, Cl, 2 * Ls)
        want = torch.zeros(N, Cl, 2 * Ls, dtype=torch.float64)
        REF.convt1d_up(*d64(y, w), want, act)
        got = torch.zeros(N, Cl, 2 * Ls, device=DEV)
        hip.convt1d_up(*gpu(y, w), got, act)
        close("ct1d_up", got, want)
        want = torch.zeros(N, Cs, Ls, dtype=torch.float64)
        REF.convt1d_down(*d64(x, w), want)
        got = torch.zeros(N, Cs, Ls, device=DEV)
        hip.convt1d_down(*gpu(x, w), got)
        close("ct1d_down", got, want)
        want = torch.zeros(Cs, Cl, 4, dtype=torch.float64)
        REF.convt1d_wgrad(*d64(y, x), want)
        got = torch.zeros(Cs, Cl, 4, device=DEV)
        hip.convt1d_wgrad(*gpu(y, x), got)
        close("ct1d_wgrad", got, want)
    z, w1, da = rnd(rng, N, 128), rnd(rng, 128, 8192, scale=0.1), rnd(rng, N, 8192)
    want, got = torch.zeros(N, 8192, dtype=torch.float64), torch.zeros(N, 8192, device=DEV)
    REF.gemm_nn(*d64(z, w1), want)
    hip.gemm_nn(*gpu(z, w1), got)
    close("gemm_nn", got, want)
    want, got = torch.zeros(128, 8192, dtype=torch.float64), torch.zeros(128, 8192, device=DEV)
    REF.gemm_tn(*d64(z, da), want)
    hip.gemm_tn(*gpu(z, da), got)
    close("gemm_tn", got, want)


def test_step_graph_capture_and_replay(hip):
    """pgan_step_graph_*: a short chain of launches (with a fork onto a second stream and back) recorded once and
    replayed equals the same chain launched eagerly; nothing runs during the recording."""
    import pgan_hip
    rng = np.random.default_rng(21)
    x0 = rnd(rng, 4096).to(DEV)
    x, y, z = x0.clone(), torch.zeros(4096, device=DEV), torch.zeros(4096, device=DEV)
    main, other = torch.cuda.Stream(), torch.cuda.Stream()

    def chain():
        hip.affine(x, 1.5, 0.25, x)          # x <- 1.5 x + 0.25
        other.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(other):
            hip.affine(x, 2.0, 0.0, y)       # y <- 2 x     (second stream)
        hip.affine(x, -1.0, 1.0, z)          # z <- 1 - x
        torch.cuda.current_stream().wait_stream(other)
        hip.affine(y, 1.0, 3.0, y)           # y <- y + 3
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        g = pgan_hip.StepGraph()
        with g:
            chain()
        main.synchronize()
        assert torch.equal(x, x0) and float(y.abs().sum()) == 0.0      # recorded, not run
        for _ in range(3):
            g.launch()
        main.synchronize()
    want = x0.clone()
    for _ in range(3):
        want = 1.5 * want + 0.25
    close("graph.x", x, want.double().cpu())
    close("graph.y", y, (2 * want + 3).double().cpu())
    close("graph.z", z, (1 - want).double().cpu())
    del g


@pytest.mark.gpu
@pytest.mark.parametrize("N", [3, 8, 64])
def test_groupnorm_backward_in_one_launch_equals_the_two_launch_form(N):
    """pgan_gn_bwd_onepass (a block keeps its eighth of a sample in registers between the sums and the apply part; the
    eight blocks of a sample exchange their partial sums through device memory inside the launch) gives the bits of
    pgan_gn_bwd on the four stage shapes, with sample counts that are and are not multiples of eight (two block
    orders), repeatedly on one workspace (every call must leave it as it found it), and never gives up waiting."""
    from pgan_hip import HipOps
    dev = torch.device("cuda:0")
    o = HipOps(dev)
    g = torch.Generator(device="cpu").manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    for C, H in ((64, 32), (128, 16), (256, 8), (512, 4)):
        a, u = rnd(N, C, H, H), rnd(N, C, H, H)
        gamma, beta = 1 + 0.1 * rnd(C), 0.1 * rnd(C)
        h, stats = torch.empty_like(a), torch.empty(N, 2, device=dev)
        o.gn_fwd(a, gamma, beta, h, stats)
        got = {}
        for mode in (0, 2, 2):
            o.gn_onepass = mode
            out = dict(d=torch.full_like(a, float("nan")), sb=torch.full((N, 2), float("nan"), device=dev),
                       pg=torch.full((N, C), float("nan"), device=dev), pb=torch.full((N, C), float("nan"), device=dev),
                       am=torch.full((N, 8), float("nan"), device=dev))
            for use_h in (False, True):      # mask recomputed from a (beta given) and read from h
                o.gn_bwd(u, h if use_h else None, a, stats, gamma, out["d"], out["sb"], out["pg"], out["pb"], amax=out["am"],
                         beta=None if use_h else beta)
                torch.cuda.synchronize()
                key = (mode > 0, use_h)
                if key in got:
                    for k, v in out.items():
                        assert torch.equal(v, got[key][k]), (C, H, k, "second call on the same workspace")
                got[key] = {k: v.clone() for k, v in out.items()}
        for use_h in (False, True):
            for k in got[(False, use_h)]:
                assert torch.isfinite(got[(True, use_h)][k]).all(), (C, H, k)
                assert torch.equal(got[(False, use_h)][k], got[(True, use_h)][k]), (C, H, k, use_h)
    assert not o.gn_onepass_error()


@pytest.mark.parametrize("Cs,Cl,hs", LAYERS + [(64, 1, 32)])
@pytest.mark.parametrize("N", [64, 128, 320])
@pytest.mark.parametrize("precision", ["h2", "x3", "fp32"])
def test_convolutions_at_bench_sizes(Cs, Cl, hs, N, precision):
    """The launch shapes of the benchmark (64- and 128-sample chains, the 320-sample batched generator forward) choose
    other tiles, K splits and block orders than the small cases above.  Checked there through what needs no oracle at
    that size -- DOWN, UP and WGRAD are the three partial derivatives of one trilinear form T(x, w, y) = <conv(x, w), y>,
    so <DOWN(x, w), y> = <x, UP(y, w)> = <WGRAD(y, x), w> -- and against the fp64 statement on the first and last two
    samples of the batch (tile edges)."""
    from pgan_hip import HipOps
    o = HipOps(torch.device(DEV), precision=precision, max_samples=2048)
    g = torch.Generator(device="cpu").manual_seed(100 + hs + N)
    x = torch.randn(N, Cl, 2 * hs, 2 * hs, generator=g).to(DEV)
    y = torch.randn(N, Cs, hs, hs, generator=g).to(DEV)
    w = (torch.randn(Cs, Cl, 4, 4, generator=g) * 0.05).to(DEV)
    dx = torch.full_like(y, float("nan"))
    uy = torch.full_like(x, float("nan"))
    dw = torch.full_like(w, float("nan"))
    o.conv_down(x, w, dx)
    o.conv_up(y, w, uy)
    o.conv_wgrad(y, x, dw)
    torch.cuda.synchronize()
    t_down = float((dx.double() * y.double()).sum())
    t_up = float((uy.double() * x.double()).sum())
    t_wg = float((dw.double() * w.double()).sum())
    scale = float(dx.double().norm() * y.double().norm())
    print(f"{precision} N={N} Cs={Cs} Cl={Cl} hs={hs}: T = {t_down:.6e} / {t_up:.6e} / {t_wg:.6e}, scale {scale:.3e}")
    assert abs(t_down - t_up) <= 1e-5 * scale and abs(t_down - t_wg) <= 1e-5 * scale
    edge = torch.tensor([0, 1, N - 2, N - 1])
    xe, ye = x[edge].cpu().double(), y[edge].cpu().double()
    want_d = torch.nn.functional.conv2d(xe, w.cpu().double(), stride=2, padding=1)
    want_u = torch.nn.functional.conv_transpose2d(ye, w.cpu().double(), stride=2, padding=1)
    close("conv_down, edge samples", dx[edge], want_d)
    close("conv_up, edge samples", uy[edge], want_u)
    # the weight gradient over the whole batch against fp64 on the device
    want_w = torch.nn.grad.conv2d_weight(x.double(), w.shape, y.double(), stride=2, padding=1)
    close("conv_wgrad", dw, want_w.cpu())


@pytest.mark.parametrize("precision", ["h2", "x3"])
def test_conv_down_into_an_output_cleared_beforehand(precision):
    """Launch shapes that split their reduction over two blocks per tile add into the output (conv_down_adds says which);
    with the output cleared beforehand by zero_multi, conv_down(zeroed=True) gives the bits of the plain call, which clears
    it itself; zero_multi clears several buffers of different sizes in one launch."""
    from pgan_hip import HipOps
    o = HipOps(torch.device(DEV), precision=precision)
    g = torch.Generator(device="cpu").manual_seed(21)
    adds = {(64, 512, 256, 4): True, (64, 256, 128, 8): True, (128, 512, 256, 4): True, (256, 512, 256, 4): True,
            (128, 128, 64, 16): False, (320, 128, 64, 16): False}
    outs = []
    for (N, Cs, Cl, hs), want in adds.items():
        assert o.conv_down_adds(N, Cs, Cl, hs) == want, (N, Cs, Cl, hs)
        x = torch.randn(N, Cl, 2 * hs, 2 * hs, generator=g).to(DEV)
        w = (torch.randn(Cs, Cl, 4, 4, generator=g) * 0.05).to(DEV)
        plain = torch.full((N, Cs, hs, hs), float("nan"), device=DEV)
        ahead = torch.full((N, Cs, hs, hs), float("nan"), device=DEV)
        o.conv_down(x, w, plain)
        outs.append((x, w, plain, ahead))
    o.zero_multi([t[3] for t in outs])
    torch.cuda.synchronize()
    assert all(float(t[3].abs().max()) == 0.0 for t in outs)
    for x, w, plain, ahead in outs:
        o.conv_down(x, w, ahead, zeroed=True)
    torch.cuda.synchronize()
    for x, w, plain, ahead in outs:
        assert torch.isfinite(plain).all() and torch.equal(plain, ahead), tuple(x.shape)
