"""pgan_gn_fwd_static, the entry point that allocates nothing (partial sums in a fixed array inside the library),
against pgan_gn_fwd (caller workspace), through the C-ABI: bit-identical outputs on two calls in a row, the second
with a smaller input; sizes beyond the fixed maximum are refused before anything is launched; no device memory is
requested by the calls (the framework allocator's request count and the device's free bytes do not move); the call can
be captured into a hipGraph and replayed."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _call(lib, name, a, gamma, beta, ws=None):
    N, C, HW = a.shape[0], a.shape[1], a.shape[2] * a.shape[3]
    h, st = torch.full_like(a, -7.0), torch.full((N, 2), -7.0, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    if ws is None:
        rc = lib.pgan_gn_fwd_static(a.data_ptr(), gamma.data_ptr(), beta.data_ptr(), h.data_ptr(), st.data_ptr(), N, C,
                                    HW, stream)
    else:
        rc = lib.pgan_gn_fwd(a.data_ptr(), gamma.data_ptr(), beta.data_ptr(), h.data_ptr(), st.data_ptr(),
                             ws.data_ptr(), ws.numel(), N, C, HW, None, stream)
    assert rc == 0, (name, rc)
    torch.cuda.synchronize()
    return h, st


def test_static_scratch_groupnorm_equals_workspace_groupnorm():
    from pgan_hip import _lib
    lib = _lib._real
    nmax = lib.pgan_gn_fwd_static_max_samples()
    assert nmax == 2048
    ws = torch.empty(lib.pgan_gn_workspace_bytes(nmax), dtype=torch.uint8, device=DEV)
    g = torch.Generator().manual_seed(0)
    assert lib.pgan_gn_fwd_static_init() == 0           # the array's address on this device, looked up outside any capture
    # two calls in a row on the same GPU, the second smaller (the array keeps the first call's partial sums)
    for N, C, hs in ((192, 128, 16), (5, 512, 4), (64, 64, 32), (3, 256, 8)):
        a = (torch.randn(N, C, hs, hs, generator=g) * 2 + 0.3).to(DEV)
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
        h1, s1 = _call(lib, "static", a, gamma, beta)
        # the same call again, on outputs that exist: nothing is requested from the framework's allocator, and the
        # device's free memory (which a hipMalloc inside the library would lower) does not move
        hx, sx = torch.full_like(a, -7.0), torch.full((N, 2), -7.0, device=DEV)
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats(DEV).get("allocation.all.allocated", 0)
        free0 = torch.cuda.mem_get_info(DEV)[0]
        rc = lib.pgan_gn_fwd_static(a.data_ptr(), gamma.data_ptr(), beta.data_ptr(), hx.data_ptr(), sx.data_ptr(), N, C,
                                    hs * hs, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0 and torch.cuda.mem_get_info(DEV)[0] == free0
        assert torch.cuda.memory_stats(DEV).get("allocation.all.allocated", 0) == before
        assert torch.equal(hx, h1) and torch.equal(sx, s1)
        h2, s2 = _call(lib, "workspace", a, gamma, beta, ws)
        assert torch.equal(h1, h2) and torch.equal(s1, s2), (N, C, hs)
        assert not bool((h1 == -7.0).all())
        # reference statement of the op in fp64
        x = a.double().view(N, -1)
        mu, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
        y = ((x - mu) / (var + 1e-5).sqrt()).view(N, C, -1) * gamma.double().view(1, C, 1) + beta.double().view(1, C, 1)
        want = torch.where(y > 0, y, 0.1 * y).view_as(a)
        assert float((h1.double() - want).norm() / want.norm()) < 1e-6
    # beyond the maximum: refused (-3), nothing written
    big = torch.zeros(nmax + 1, 8, 4, 4, device=DEV)
    h = torch.full_like(big, -7.0)
    st = torch.full((nmax + 1, 2), -7.0, device=DEV)
    one = torch.ones(8, device=DEV)
    rc = lib.pgan_gn_fwd_static(big.data_ptr(), one.data_ptr(), one.data_ptr(), h.data_ptr(), st.data_ptr(), nmax + 1, 8,
                                16, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == -3 and bool((h == -7.0).all())
    # at the maximum it runs
    rc = lib.pgan_gn_fwd_static(big.data_ptr(), one.data_ptr(), one.data_ptr(), h.data_ptr(), st.data_ptr(), nmax, 8,
                                16, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0 and bool((h[:nmax] == 1.0).all()) and bool((h[nmax:] == -7.0).all())


def test_static_scratch_groupnorm_inside_a_graph_capture():
    """The callers the entry point is for include code under hipGraph capture: captured through the library's own
    begin / end calls and replayed twice on new inputs, it equals pgan_gn_fwd."""
    from pgan_hip import StepGraph, _lib
    lib = _lib._real
    assert lib.pgan_gn_fwd_static_init() == 0
    N, C, hs = 24, 128, 16
    g = torch.Generator().manual_seed(1)
    a = torch.randn(N, C, hs, hs, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    h, st = torch.full_like(a, -7.0), torch.full((N, 2), -7.0, device=DEV)
    ws = torch.empty(lib.pgan_gn_workspace_bytes(N), dtype=torch.uint8, device=DEV)
    cap = torch.cuda.Stream()
    torch.cuda.synchronize()
    graph = StepGraph()
    with torch.cuda.stream(cap):
        with graph:
            rc = lib.pgan_gn_fwd_static(a.data_ptr(), gamma.data_ptr(), beta.data_ptr(), h.data_ptr(), st.data_ptr(), N, C,
                                        hs * hs, cap.cuda_stream)
            assert rc == 0
    torch.cuda.synchronize()
    assert bool((h == -7.0).all())              # captured, not run
    for rep in range(2):
        a.copy_(torch.randn(N, C, hs, hs, generator=g).to(DEV) * (rep + 1))
        torch.cuda.synchronize()
        with torch.cuda.stream(cap):
            graph.launch()
        cap.synchronize()
        h2, s2 = _call(lib, "workspace", a, gamma, beta, ws)
        assert torch.equal(h, h2) and torch.equal(st, s2), rep
