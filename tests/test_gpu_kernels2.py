"""Parity of the weight-gradient GEMM, the first-conv path, BatchNorm/ReLU/Dropout/MaxPool,
the 1x1 head and the small helpers against the CPU oracle (through the C ABI).

bf16 kernels are checked against the oracle evaluated on the same bf16-rounded operands
(see test_gpu_kernels.py for the tolerance rationale); fp32 outputs (statistics, parameter
gradients, recon) to 1e-4 .. 1e-5 relative per tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle as O
from golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from vae_amd import _lib
    _lib.load()
    return _lib


def bf(t):
    return t.to(torch.bfloat16).float()


def nhwc(t, dtype=torch.bfloat16):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def check_bf16(got, ref, what, rtol=3e-3):
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max()
    assert (err <= tol).all(), f"{what}: max err {err.max().item()} (ref max {ref.abs().max().item()})"
    assert rel_l2(got, ref) <= rtol, what


def wgrad(L, P, Q, M, N, K, ldp, ldq, mode, H, W, taps, nsplit=None):
    s = nsplit or L.call("vae_wgrad_ksplit", M, N, taps, K)
    slab = torch.full((s, M, taps, N), float("nan"), device=DEV)
    L.call("vae_wgrad_gemm", L.ptr(P), L.ptr(Q), L.ptr(slab), M, N, K, ldp, ldq, mode, H, W, s, L.current_stream())
    return slab, s


# ------------------------------------------------------------------ first conv (8-channel input)
@pytest.mark.parametrize("n,H,W,Ci", [(2, 8, 8, 3), (3, 32, 32, 3), (1, 6, 10, 1)])
def test_conv_c8_fwd_and_wgrad(L, n, H, W, Ci):
    g = torch.Generator().manual_seed(n + H)
    Co = 64
    x = bf(torch.rand(n, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / 5)
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    xd = x.to(DEV)
    x8 = torch.full((n, H, W, 8), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.call("vae_nchw_to_nhwc8", L.ptr(xd), L.ptr(x8), n, Ci, H, W, L.current_stream())
    exp8 = torch.zeros(n, H, W, 8)
    exp8[..., :Ci] = x.permute(0, 2, 3, 1)
    assert torch.equal(x8.float().cpu(), exp8)
    # pack [Co][16 taps][8] with the pack kernel: ref index (co, ci, tap) -> co*128 + tap*8 + ci
    wp = torch.zeros(Co, 16, 8, device=DEV, dtype=torch.bfloat16)
    wd = w.to(DEV)
    L.call("vae_pack_bf16", L.ptr(wd), L.ptr(wp), L.long4(Co, Ci, 9, 1), L.long4(128, 1, 8, 0), 0, L.current_stream())
    exp = torch.zeros(Co, 16, 8)
    exp[:, :9, :Ci] = w.permute(0, 2, 3, 1).reshape(Co, 9, Ci)
    assert torch.equal(wp.float().cpu(), exp)
    bd = b.to(DEV)
    y = torch.full((n, H, W, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.call("vae_conv3x3_c8_gemm", L.ptr(x8), L.ptr(wp), L.ptr(bd), L.ptr(y), None, None, n, H, W, Co, L.current_stream())
    check_bf16(from_nhwc(y), ref, "conv c8 fwd")
    # weight gradient
    dy = bf(torch.randn(n, Co, H, W, generator=g))
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, padding=1).backward(dy)
    dyd = nhwc(dy)
    slab, s = wgrad(L, dyd, x8, Co, 64, n * H * W, Co, 8, 3, H, W, 2)
    dw = torch.full((Co, Ci, 3, 3), float("nan"), device=DEV)
    L.call("vae_reduce_slabs", L.ptr(slab), s, Co * 128, L.ptr(dw), L.long4(Co, Ci, 9, 1), L.long4(128, 1, 8, 0), 0, 0,
           L.current_stream())
    assert rel_l2(dw.cpu(), wr.grad) <= 1e-4


# ------------------------------------------------------------------ weight gradients
@pytest.mark.parametrize("n,H,W,Ci,Co", [(2, 4, 4, 64, 64), (3, 8, 8, 64, 128), (2, 16, 16, 128, 128),
                                         (5, 4, 4, 256, 256), (2, 32, 32, 64, 64), (1, 6, 10, 128, 64)])
def test_conv3x3_wgrad(L, n, H, W, Ci, Co):
    g = torch.Generator().manual_seed(n + H + Ci)
    x = bf(torch.randn(n, Ci, H, W, generator=g))
    dy = bf(torch.randn(n, Co, H, W, generator=g))
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    xd, dyd = nhwc(x), nhwc(dy)
    slab, s = wgrad(L, dyd, xd, Co, Ci, n * H * W, Co, Ci, 1, H, W, 9)
    dw = torch.full((Co, Ci, 3, 3), float("nan"), device=DEV)
    # ref index (co, ci, tap) -> packed co*(9*Ci) + tap*Ci + ci
    L.call("vae_reduce_slabs", L.ptr(slab), s, 9 * Co * Ci, L.ptr(dw), L.long4(Co, Ci, 9, 1), L.long4(9 * Ci, 1, Ci, 0),
           0, 0, L.current_stream())
    assert rel_l2(dw.cpu(), w.grad) <= 1e-4
    # accumulate flag adds on top
    L.call("vae_reduce_slabs", L.ptr(slab), s, 9 * Co * Ci, L.ptr(dw), L.long4(Co, Ci, 9, 1), L.long4(9 * Ci, 1, Ci, 0),
           0, 1, L.current_stream())
    assert rel_l2(dw.cpu(), 2 * w.grad) <= 1e-4


@pytest.mark.parametrize("n,H,W,Ci,Co,ns", [
    (5, 4, 4, 256, 64, 0), (3, 4, 4, 64, 64, 1), (3, 8, 8, 128, 64, 0), (2, 16, 16, 64, 128, 2), (2, 32, 32, 64, 64, 0),
    (1, 64, 64, 64, 64, 7), (9, 2, 4, 64, 64, 0), (1, 16, 8, 64, 64, 0), (64, 8, 8, 128, 256, 0),
    # one, two and three K tiles per workgroup (the shortest pipelines of the copying / computing form), an odd count
    (1, 8, 8, 64, 64, 1), (2, 8, 8, 64, 64, 1), (3, 8, 8, 64, 64, 1), (7, 16, 16, 64, 64, 4)])
@pytest.mark.parametrize("form", [2, 1])
def test_conv3x3_wgrad_patch(L, n, H, W, Ci, Co, ns, form):
    """nine-taps-per-workgroup kernel == autograd's Conv2d weight gradient (images smaller than, equal to and
    larger than a 64-pixel K tile, ragged last tile, forced and suggested K splits), in both of its forms: every wave
    copies and computes (2), separate copying and computing waves (1)"""
    L.call("vae_set_tuning", b"wgrad_pc", form)
    try:
        _check_conv3x3_wgrad_patch(L, n, H, W, Ci, Co, ns)
    finally:
        L.call("vae_set_tuning", b"wgrad_pc", 0)


def _check_conv3x3_wgrad_patch(L, n, H, W, Ci, Co, ns):
    assert L.call("vae_conv3_wgrad_patch_ok", H, W, Ci, Co) == 1
    g = torch.Generator().manual_seed(n + H + Ci)
    x = bf(torch.randn(n, Ci, H, W, generator=g))
    dy = bf(torch.randn(n, Co, H, W, generator=g))
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    xd, dyd = nhwc(x), nhwc(dy)
    s = ns or L.call("vae_conv3_wgrad_patch_ksplit", n, H, W, Ci, Co)
    assert 1 <= s <= (n * H * W + 63) // 64
    slab = torch.full((s, Co, 9, Ci), float("nan"), device=DEV)
    L.call("vae_conv3_wgrad_patch", L.ptr(dyd), L.ptr(xd), L.ptr(slab), n, H, W, Ci, Co, s, L.current_stream())
    dw = slab.sum(0).permute(0, 2, 1).reshape(Co, Ci, 3, 3)
    assert rel_l2(dw.cpu(), w.grad) <= 1e-4


def _stage_jobs(n, size):
    """(H, Ci, Co) of the conv3x3 layers VAEConv's backward hands to one launch: blocks 3 + 2, then blocks 1 + 0."""
    return ([(size // 8, 256, 256), (size // 8, 256, 256), (size // 4, 256, 256), (size // 4, 128, 256)],
            [(size // 2, 128, 128), (size // 2, 64, 128), (size, 64, 64)])


@pytest.mark.parametrize("n,size,max_wgs", [(16, 32, 256), (3, 32, 256), (8, 64, 256), (64, 32, 64), (5, 32, 8)])
def test_conv3x3_wgrad_patch_multi(L, n, size, max_wgs):
    """several layers' weight gradients from ONE launch (each workgroup finds its layer by its index) == the per-layer
    launch on the same operands with the same K split, bit for bit, and == autograd's gradient; layers of different
    widths share the launch, at budgets above and below the number of output tiles"""
    import ctypes
    g = torch.Generator().manual_seed(n + size)
    for geo in _stage_jobs(n, size):
        jobs = (L.WgradJob * len(geo))()
        keep, want = [], []
        for q, (H, Ci, Co) in zip(jobs, geo):
            x = bf(torch.randn(n, Ci, H, H, generator=g))
            dy = bf(torch.randn(n, Co, H, H, generator=g))
            w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
            F.conv2d(x, w, None, padding=1).backward(dy)
            xd, dyd = nhwc(x), nhwc(dy)
            keep.append((xd, dyd))
            want.append(w.grad)
            q.dy, q.x, q.n_img, q.H, q.W, q.Cin, q.Cout = L.ptr(dyd), L.ptr(xd), n, H, H, Ci, Co
        ns = (ctypes.c_int * len(geo))()
        L.call("vae_conv3_wgrad_multi_plan", jobs, len(geo), max_wgs, ns)
        tiles = [(Ci // 64) * (Co // 64) for _, Ci, Co in geo]
        assert sum(t * s for t, s in zip(tiles, ns)) <= max(max_wgs, sum(tiles))
        slabs = []
        for q, s, (H, Ci, Co) in zip(jobs, ns, geo):
            assert 1 <= s <= (n * H * H + 63) // 64
            slabs.append(torch.full((s, Co, 9, Ci), float("nan"), device=DEV))
            q.slab, q.nsplit = L.ptr(slabs[-1]), s
        L.call("vae_conv3_wgrad_patch_multi", jobs, len(geo), L.current_stream())
        for slab, s, (xd, dyd), ref, (H, Ci, Co) in zip(slabs, ns, keep, want, geo):
            dw = slab.sum(0).permute(0, 2, 1).reshape(Co, Ci, 3, 3)
            assert rel_l2(dw.cpu(), ref) <= 1e-4, (H, Ci, Co)
            one = torch.full((s, Co, 9, Ci), float("nan"), device=DEV)
            L.call("vae_set_tuning", b"wgrad_pc", 1)          # the form of the kernel the shared launch is built from
            try:
                L.call("vae_conv3_wgrad_patch", L.ptr(dyd), L.ptr(xd), L.ptr(one), n, H, H, Ci, Co, s, L.current_stream())
            finally:
                L.call("vae_set_tuning", b"wgrad_pc", 0)
            assert torch.equal(one, slab), (H, Ci, Co)


def test_conv3x3_wgrad_patch_multi_rejects(L):
    jobs = (L.WgradJob * 1)()
    t = torch.zeros(64, device=DEV)
    q = jobs[0]
    q.dy = q.x = q.slab = L.ptr(t)
    q.n_img, q.H, q.W, q.Cin, q.Cout, q.nsplit = 1, 6, 6, 64, 64, 1
    with pytest.raises(RuntimeError, match="not covered"):
        L.call("vae_conv3_wgrad_patch_multi", jobs, 1, L.current_stream())
    with pytest.raises(RuntimeError, match="out of range"):
        L.call("vae_conv3_wgrad_patch_multi", jobs, 0, L.current_stream())
    q.H = q.W = 8
    q.nsplit = 5                  # more splits than the single K tile of one 8x8 image
    with pytest.raises(RuntimeError):
        L.call("vae_conv3_wgrad_patch_multi", jobs, 1, L.current_stream())


def test_conv3x3_wgrad_patch_rejects(L):
    assert L.call("vae_conv3_wgrad_patch_ok", 6, 6, 64, 64) == 0
    assert L.call("vae_conv3_wgrad_patch_ok", 8, 8, 32, 64) == 0
    assert L.call("vae_conv3_wgrad_patch_ok", 128, 128, 64, 64) == 0
    t = torch.zeros(64, device=DEV)
    with pytest.raises(RuntimeError, match="not covered"):
        L.call("vae_conv3_wgrad_patch", L.ptr(t), L.ptr(t), L.ptr(t), 1, 6, 6, 64, 64, 1, L.current_stream())


@pytest.mark.parametrize("n,H,W,Ci,Co", [(2, 4, 4, 256, 128), (3, 8, 8, 128, 64), (2, 16, 16, 64, 64), (1, 3, 5, 64, 64)])
def test_convt_wgrad(L, n, H, W, Ci, Co):
    g = torch.Generator().manual_seed(n + H + Ci)
    x = bf(torch.randn(n, Ci, H, W, generator=g))
    dy = bf(torch.randn(n, Co, 2 * H, 2 * W, generator=g))
    w = torch.zeros(Ci, Co, 2, 2, requires_grad=True)
    F.conv_transpose2d(x, w, None, stride=2).backward(dy)
    xd, dyd = nhwc(x), nhwc(dy)
    slab, s = wgrad(L, dyd, xd, Co, Ci, n * H * W, Co, Ci, 2, H, W, 4)
    dw = torch.full((Ci, Co, 2, 2), float("nan"), device=DEV)
    # ref index (ci, co, tap) -> packed co*(4*Ci) + tap*Ci + ci
    L.call("vae_reduce_slabs", L.ptr(slab), s, 4 * Co * Ci, L.ptr(dw), L.long4(Ci, Co, 4, 1), L.long4(1, 4 * Ci, Ci, 0),
           0, 0, L.current_stream())
    assert rel_l2(dw.cpu(), w.grad) <= 1e-4


@pytest.mark.parametrize("rows,M,N", [(4, 256, 4096), (512, 256, 1024), (130, 4096, 128), (7, 64, 64)])
def test_linear_wgrad(L, rows, M, N):
    g = torch.Generator().manual_seed(rows + M)
    dout = bf(torch.randn(rows, M, generator=g))
    x = bf(torch.randn(rows, N, generator=g))
    ref = dout.t() @ x
    dd, xd = dout.to(torch.bfloat16).to(DEV), x.to(torch.bfloat16).to(DEV)
    slab, s = wgrad(L, dd, xd, M, N, rows, M, N, 0, 1, 1, 1)
    dw = torch.full((M, N), float("nan"), device=DEV)
    L.call("vae_reduce_slabs", L.ptr(slab), s, M * N, L.ptr(dw), L.long4(M, N, 1, 1), L.long4(N, 1, 0, 0), 0, 0,
           L.current_stream())
    assert rel_l2(dw.cpu(), ref) <= 1e-4


# ------------------------------------------------------------------ BatchNorm / ReLU / Dropout / MaxPool
def stats_inputs(y_nchw):
    yn = y_nchw.permute(0, 2, 3, 1).reshape(-1, y_nchw.shape[1])
    return yn.sum(0, keepdim=True).contiguous().to(DEV), (yn ** 2).sum(0, keepdim=True).contiguous().to(DEV)


@pytest.mark.parametrize("n,H,W,C,pool,p", [(4, 8, 8, 64, 0, 0.0), (3, 8, 8, 128, 1, 0.0), (2, 16, 16, 64, 1, 0.1),
                                            (5, 4, 4, 256, 0, 0.1), (2, 6, 10, 128, 1, 0.3)])
@pytest.mark.parametrize("inject", [False, True])
def test_bn_act_fwd_bwd(L, n, H, W, C, pool, p, inject):
    g = torch.Generator().manual_seed(n + H + C + pool)
    y = bf(torch.randn(n, C, H, W, generator=g) * 1.5 + 0.3)
    gamma = 0.5 + torch.rand(C, generator=g)
    beta = 0.4 * torch.rand(C, generator=g) - 0.2
    rm, rv = torch.randn(C, generator=g) * 0.1, 0.5 + torch.rand(C, generator=g)
    seed, step, stream_id, rank = 99, 3, 2, 1
    M = n * H * W
    if p > 0:
        if inject:
            mask_nhwc = (torch.rand(n, H, W, C, generator=g) >= p).to(torch.uint8)
        else:
            mask_nhwc = torch.from_numpy(O.philox_keep_mask(M * C, p, seed, stream_id, step, rank)).view(n, H, W, C)
        mask = mask_nhwc.permute(0, 3, 1, 2)
    else:
        mask_nhwc = mask = None
    # ---- oracle
    yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    stats = {}
    a = O.dropout(torch.relu(O.batch_norm(yr, gr, br, rm, rv, True, stats, "bn")), p, True, mask)
    out_ref = F.max_pool2d(a, 2) if pool else a
    gin = bf(torch.randn(out_ref.shape, generator=g))
    (out_ref * gin).sum().backward()
    # ---- HIP
    ssum, ssq = stats_inputs(y)
    gd, bd, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    sc, sh, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
    L.call("vae_bn_finalize", L.ptr(ssum), L.ptr(ssq), 1, C, 1, float(M), L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd),
           0.1, 1e-5, 0, L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.current_stream())
    assert rel_l2(rmd.cpu(), stats["bn.running_mean"]) <= 1e-5
    assert rel_l2(rvd.cpu(), stats["bn.running_var"]) <= 1e-5
    yd = nhwc(y)
    md = mask_nhwc.contiguous().to(DEV) if (inject and p > 0) else None
    out = torch.full(tuple(nhwc(out_ref).shape), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.call("vae_bn_act_fwd", L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(out), n, H, W, C, pool, p, L.ptr(md), seed, step,
           stream_id, rank, L.current_stream())
    check_bf16(from_nhwc(out), out_ref.detach(), "bn_act fwd")
    gind = nhwc(gin)
    nb = L.call("vae_bn_act_bwd_blocks", n, H, W, C, pool)
    partial = torch.full((nb, 2 * C), float("nan"), device=DEV)
    sums = torch.full((2 * C,), float("nan"), device=DEV)
    dy = torch.full((n, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.call("vae_bn_act_bwd", L.ptr(gind), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.ptr(partial),
           L.ptr(sums), L.ptr(dy), n, H, W, C, pool, p, L.ptr(md), seed, step, stream_id, rank, L.current_stream())
    assert rel_l2(sums[:C].cpu(), gr.grad) <= 2e-5
    assert rel_l2(sums[C:].cpu(), br.grad) <= 2e-5
    check_bf16(from_nhwc(dy), yr.grad, "bn_act bwd", rtol=4e-3)
    if not pool:
        with pytest.raises(RuntimeError, match="route bytes exist only for the pooled form"):
            L.call("vae_bn_act_fwd_route", L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(out), L.ptr(partial), n, H, W, C, pool, p,
                   L.ptr(md), seed, step, stream_id, rank, L.current_stream())
        return
    # ---- the routed pair: the forward records which window element won and whether a gradient flows
    # through it, the backward reads that instead of regenerating the dropout stream
    route = torch.full((n, H // 2, W // 2, C), 255, device=DEV, dtype=torch.uint8)
    out2 = torch.full_like(out, float("nan"))
    L.call("vae_bn_act_fwd_route", L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(out2), L.ptr(route), n, H, W, C, pool, p,
           L.ptr(md), seed, step, stream_id, rank, L.current_stream())
    assert torch.equal(out2, out)
    # expected route from the oracle's activation: first maximum in scan order, live iff it is positive
    win = a.detach().unfold(2, 2, 2).unfold(3, 2, 2).reshape(n, C, H // 2, W // 2, 4)
    live_ref = (win.max(-1).values > 0).permute(0, 2, 3, 1)
    r = route.cpu()
    assert int(r.max()) <= 7
    # ties and bf16-rounded activations can move the winner between equal values: the flag must match exactly,
    # the position must select a maximal element
    assert torch.equal((r >> 2).bool(), live_ref)
    picked = torch.gather(win, 4, (r & 3).permute(0, 3, 1, 2).long().unsqueeze(-1)).squeeze(-1)
    assert torch.allclose(picked, win.max(-1).values, rtol=1e-2, atol=1e-3)
    partial.fill_(float("nan")); sums2 = torch.full_like(sums, float("nan")); dy2 = torch.full_like(dy, float("nan"))
    L.call("vae_bn_act_bwd_route", L.ptr(gind), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.ptr(partial),
           L.ptr(sums2), L.ptr(dy2), L.ptr(route), n, H, W, C, pool, p, L.ptr(md), seed, step, stream_id, rank,
           L.current_stream())
    assert rel_l2(sums2[:C].cpu(), gr.grad) <= 2e-5
    assert rel_l2(sums2[C:].cpu(), br.grad) <= 2e-5
    check_bf16(from_nhwc(dy2), yr.grad, "bn_act bwd (routed)", rtol=4e-3)
    assert rel_l2(dy2.float().cpu(), dy.float().cpu()) <= 3e-3      # same result as the recomputing form, to bf16 rounding
    # ---- the same pair with ymax: the forward also leaves the conv output at each window's maximum, the backward's
    # reduction reads that instead of the four window elements -- the same numbers, so bit-identical results
    ymax = torch.full((n, H // 2, W // 2, C), float("nan"), device=DEV, dtype=torch.bfloat16)
    route3 = torch.full_like(route, 255)
    out3 = torch.full_like(out, float("nan"))
    L.call("vae_bn_act_fwd_route_y", L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(out3), L.ptr(route3), L.ptr(ymax), n, H, W, C, pool,
           p, L.ptr(md), seed, step, stream_id, rank, L.current_stream())
    assert torch.equal(out3, out) and torch.equal(route3, route)
    ywin = yd.view(n, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(n, H // 2, W // 2, C, 4)
    assert torch.equal(ymax, torch.gather(ywin, 4, (route & 3).long().unsqueeze(-1)).squeeze(-1))
    partial.fill_(float("nan")); sums3 = torch.full_like(sums, float("nan")); dy3 = torch.full_like(dy, float("nan"))
    L.call("vae_bn_act_bwd_route_y", L.ptr(gind), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.ptr(partial),
           L.ptr(sums3), L.ptr(dy3), L.ptr(route), L.ptr(ymax), n, H, W, C, pool, p, L.ptr(md), seed, step, stream_id, rank,
           L.current_stream())
    assert torch.equal(sums3, sums2) and torch.equal(dy3, dy2)
    with pytest.raises(RuntimeError, match="ymax goes with the route bytes"):
        L.call("vae_bn_act_bwd_route_y", L.ptr(gind), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.ptr(partial),
               L.ptr(sums3), L.ptr(dy3), None, L.ptr(ymax), n, H, W, C, pool, p, L.ptr(md), seed, step, stream_id, rank,
               L.current_stream())


def test_bn_finalize_eval_and_fold(L):
    C = 64
    g = torch.Generator().manual_seed(5)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    gd, bd, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    sc, sh, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
    L.call("vae_bn_finalize", None, None, 0, C, 1, 0.0, L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd), 0.1, 1e-5, 1,
           L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.current_stream())
    s_ref = gamma / torch.sqrt(rv + 1e-5)
    assert rel_l2(sc.cpu(), s_ref) <= 1e-6 and rel_l2(sh.cpu(), beta - rm * s_ref) <= 1e-6
    assert torch.equal(rmd.cpu(), rm) and torch.equal(rvd.cpu(), rv)      # eval leaves running stats alone
    # fold = 4 (ConvTranspose columns tap*C + c), 37 partial rows
    rows = 37
    ps = torch.randn(rows, 4 * C, generator=g)
    pq = torch.rand(rows, 4 * C, generator=g) * 30 + 40
    psd, pqd = ps.to(DEV), pq.to(DEV)
    count = 5000.0
    L.call("vae_bn_finalize", L.ptr(psd), L.ptr(pqd), rows, C, 4, count, L.ptr(gd), L.ptr(bd), None, None, 0.1, 1e-5, 0,
           L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.current_stream())
    m_ref = ps.double().sum(0).view(4, C).sum(0) / count
    v_ref = pq.double().sum(0).view(4, C).sum(0) / count - m_ref ** 2
    assert rel_l2(mean.cpu(), m_ref) <= 1e-6
    assert rel_l2(rstd.cpu(), 1 / torch.sqrt(v_ref + 1e-5)) <= 1e-6


# ------------------------------------------------------------------ head
@pytest.mark.parametrize("n,H,W,Co", [(2, 8, 8, 3), (3, 32, 32, 3), (1, 5, 7, 1), (2, 4, 4, 4)])
def test_head_fwd_bwd(L, n, H, W, Co):
    g = torch.Generator().manual_seed(n + H + Co)
    C = 64
    y = bf(torch.randn(n, C, H, W, generator=g))
    gamma, beta = 0.5 + torch.rand(C, generator=g), 0.4 * torch.rand(C, generator=g) - 0.2
    w = torch.randn(Co, C, 1, 1, generator=g) * 0.2
    b = torch.randn(Co, generator=g) * 0.1
    rm, rv = torch.zeros(C), torch.ones(C)
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    wr, bbr = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    a = torch.relu(O.batch_norm(yr, gr, br, rm, rv, True))
    recon_ref = torch.sigmoid(F.conv2d(a, wr, bbr))
    dr = torch.randn(recon_ref.shape, generator=g)
    (recon_ref * dr).sum().backward()
    M = n * H * W
    ssum, ssq = stats_inputs(y)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    sc, sh, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
    L.call("vae_bn_finalize", L.ptr(ssum), L.ptr(ssq), 1, C, 1, float(M), L.ptr(gd), L.ptr(bd), None, None, 0.1, 1e-5, 0,
           L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.current_stream())
    yd, wd, bbd = nhwc(y), w.view(Co, C).contiguous().to(DEV), b.to(DEV)
    recon = torch.full((n, Co, H, W), float("nan"), device=DEV)
    L.call("vae_head_fwd", L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(wd), L.ptr(bbd), L.ptr(recon), n, H, W, Co,
           L.current_stream())
    assert rel_l2(recon.cpu(), recon_ref.detach()) <= 1e-5
    nb = L.call("vae_head_bwd_blocks", n, H, W)
    ncols = (3 + Co) * 64
    partial = torch.full((nb, ncols), float("nan"), device=DEV)
    sums = torch.full((128,), float("nan"), device=DEV)
    sums_head = torch.full((Co * 65,), float("nan"), device=DEV)
    dy = torch.full((n, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
    drd = dr.to(DEV)
    L.call("vae_head_bwd", L.ptr(drd), L.ptr(recon), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd),
           L.ptr(wd), L.ptr(partial), L.ptr(sums), L.ptr(sums_head), L.ptr(dy), n, H, W, Co, L.current_stream())
    s, sh_ = sums.cpu(), sums_head.cpu()
    assert rel_l2(s[:64], gr.grad) <= 2e-5
    assert rel_l2(s[64:128], br.grad) <= 2e-5
    assert rel_l2(sh_[:Co * 64], wr.grad.view(-1)) <= 2e-5
    assert rel_l2(sh_[Co * 64:], bbr.grad) <= 2e-5
    check_bf16(from_nhwc(dy), yr.grad, "head bwd", rtol=4e-3)


# ------------------------------------------------------------------ small helpers
def test_tanh_bwd_and_sum_partials(L):
    g = torch.Generator().manual_seed(3)
    rows, cols = 130, 4096
    dx, h = bf(torch.randn(rows, cols, generator=g)), bf(torch.tanh(torch.randn(rows, cols, generator=g)))
    ref = dx * (1 - h * h)
    dxd, hd = dx.to(torch.bfloat16).to(DEV), h.to(torch.bfloat16).to(DEV)
    nr = L.call("vae_tanh_bwd_rows", rows)
    dpre = torch.empty(rows, cols, device=DEV, dtype=torch.bfloat16)
    colp = torch.full((nr, cols), float("nan"), device=DEV)
    L.call("vae_tanh_bwd", L.ptr(dxd), L.ptr(hd), L.ptr(dpre), L.ptr(colp), rows, cols, L.current_stream())
    check_bf16(dpre.float().cpu(), ref, "tanh bwd")
    out = torch.empty(cols, device=DEV)
    L.call("vae_sum_partials", L.ptr(colp), nr, cols, L.ptr(out), 1.0, L.current_stream())
    assert rel_l2(out.cpu(), ref.sum(0)) <= 1e-5


def test_layout_conversion_bumps_the_batch_counters(L):
    """nn.BatchNorm2d adds 1 to num_batches_tracked in every train-mode forward (torch; vae/models/vae.py:26,32): the
    first kernel of a pass can do it for the layers the pass runs (cfg.batches_tracked of the whole-model entry points)."""
    g = torch.Generator().manual_seed(9)
    n, C, H, W = 3, 3, 8, 8
    x = torch.rand(n, C, H, W, generator=g).to(DEV)
    a = torch.full((n, H, W, 8), float("nan"), device=DEV, dtype=torch.bfloat16)
    b = torch.full_like(a, float("nan"))
    counters = torch.arange(11, device=DEV, dtype=torch.int64)
    L.call("vae_nchw_to_nhwc8", L.ptr(x), L.ptr(a), n, C, H, W, L.current_stream())
    L.call("vae_nchw_to_nhwc8_count", L.ptr(x), L.ptr(b), n, C, H, W, L.ptr(counters), 8, L.current_stream())
    assert torch.equal(a, b)
    assert torch.equal(a[..., :3].float().cpu(), x.permute(0, 2, 3, 1).to(torch.bfloat16).float().cpu()) and float(a[..., 3:].abs().max()) == 0
    assert counters.tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 10]
    L.call("vae_nchw_to_nhwc8_count", L.ptr(x), L.ptr(b), n, C, H, W, None, 0, L.current_stream())
    assert counters.tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 10]
