"""GPU parity of the one-pass backward of the first  Conv2d(Cin, 64, 3, padding=1) -> BatchNorm2d -> ReLU
(vae/models/vae.py:24-31 with the image as input, vae.py:34): d gamma, d beta and the conv weight gradient from one
read of the activation gradient and the stored conv output, against an fp64 torch evaluation of the same formulas
(BatchNorm backward, then torch.nn.grad.conv2d_weight) and against the separate kernels it replaces."""
import pytest
import torch

from golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
WGRAD_CONV3_C8 = 3


@pytest.fixture(scope="module")
def L():
    from vae_amd import _lib
    _lib.load()
    return _lib


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def make_case(n, H, W, Cin, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, Cin, H, W, generator=g)                               # an image batch in [0, 1)
    y = (torch.randn(n, 64, H, W, generator=g) * torch.rand(64, generator=g).view(1, -1, 1, 1) * 2 +
         torch.randn(64, generator=g).view(1, -1, 1, 1)).to(torch.bfloat16)  # conv output with per-channel offset / spread
    gin = (torch.randn(n, 64, H, W, generator=g) * 1e-3 + 2e-4).to(torch.bfloat16)
    yf = y.float()
    mean = yf.mean((0, 2, 3))
    var = yf.var((0, 2, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma = torch.rand(64, generator=g) + 0.5
    beta = torch.randn(64, generator=g) * 0.3
    scale = gamma * rstd
    shift = beta - mean * scale
    return x, y, gin, scale, shift, mean, rstd


def reference(x, y, gin, scale, shift, mean, rstd):
    n, Cin, H, W = x.shape
    v = lambda t: t.view(1, -1, 1, 1)
    pos = (y.float() * v(scale) + v(shift)) > 0                              # the kernels' fp32 predicate
    d = torch.where(pos, gin.double(), torch.zeros((), dtype=torch.float64))
    xh = (y.double() - v(mean).double()) * v(rstd).double()
    cnt = n * H * W
    s_dx, s_d = (d * xh).sum((0, 2, 3)), d.sum((0, 2, 3))
    gy = v(scale).double() * (d - v(s_d) / cnt - xh * v(s_dx) / cnt)
    xb = x.to(torch.bfloat16).double()
    dW = torch.nn.grad.conv2d_weight(xb, (64, Cin, 3, 3), gy, padding=1)
    return s_dx, s_d, dW, gy


def run_one_pass(L, x, y, gin, scale, shift, mean, rstd):
    n, Cin, H, W = x.shape
    xd = x.to(DEV)
    x8 = torch.zeros(n, H, W, 8, device=DEV, dtype=torch.bfloat16)
    L.call("vae_nchw_to_nhwc8", L.ptr(xd), L.ptr(x8), n, Cin, H, W, L.current_stream())
    yd, gd = nhwc(y).to(DEV), nhwc(gin).to(DEV)
    co = [t.to(DEV) for t in (scale, shift, mean, rstd)]
    nf = L.call("vae_bn_bwd_conv1_scratch_floats", n, H, W)
    scratch = torch.full((nf,), float("nan"), device=DEV)
    sums = torch.full((128,), float("nan"), device=DEV)
    dW = torch.full((64, Cin, 3, 3), float("nan"), device=DEV)
    L.call("vae_bn_bwd_conv1", L.ptr(gd), L.ptr(yd), L.ptr(x8), *[L.ptr(t) for t in co], L.ptr(scratch), L.ptr(sums), L.ptr(dW),
           n, H, W, Cin, L.current_stream())
    torch.cuda.synchronize()
    return sums.cpu(), dW.cpu(), (x8, yd, gd, co)


@pytest.mark.parametrize("n,H,W,Cin", [(8, 32, 32, 3), (3, 12, 20, 1), (2, 16, 16, 2), (1, 8, 8, 3), (96, 32, 32, 3), (5, 6, 10, 3)])
def test_one_pass_matches_fp64_formulas(L, n, H, W, Cin):
    """sizes: power-of-two and other image shapes, a last tile that is not full (720, 300 pixels), one workgroup walking
    several tiles (98 304 pixels = 1 536 tiles on 768 workgroups), one, two and three input channels"""
    case = make_case(n, H, W, Cin, 1000 * n + H + Cin)
    s_dx, s_d, dW_ref, _ = reference(*case)
    sums, dW, _ = run_one_pass(L, *case)
    assert rel_l2(sums[:64], s_dx.float()) <= 1e-5
    assert rel_l2(sums[64:], s_d.float()) <= 1e-5
    assert rel_l2(dW, dW_ref.float()) <= 2e-5
    sums2, dW2, _ = run_one_pass(L, *case)
    assert torch.equal(sums, sums2) and torch.equal(dW, dW2)                 # fixed fold order: reproducible


def test_one_pass_vs_the_separate_kernels(L):
    """the path it replaces: vae_bn_act_bwd (two passes, g rounded to bf16) + the 8-channel-input weight-gradient GEMM +
    its slab fold.  Both are compared with the fp64 formulas: the one-pass result must be the closer one."""
    n, H, W, Cin = 16, 32, 32, 3
    case = make_case(n, H, W, Cin, 77)
    s_dx, s_d, dW_ref, gy = reference(*case)
    sums, dW, (x8, yd, gd, co) = run_one_pass(L, *case)
    nb = L.call("vae_bn_act_bwd_blocks", n, H, W, 64, 0)
    partial = torch.empty(nb * 128, device=DEV)
    sums_s = torch.empty(128, device=DEV)
    g = torch.empty(n, H, W, 64, device=DEV, dtype=torch.bfloat16)
    L.call("vae_bn_act_bwd", L.ptr(gd), L.ptr(yd), *[L.ptr(t) for t in co], L.ptr(partial), L.ptr(sums_s), L.ptr(g), n, H, W, 64,
           0, 0.0, None, 0, 0, 0, 0, L.current_stream())
    M = n * H * W
    ns = L.call("vae_wgrad_ksplit", 64, 64, 2, M)
    slab = torch.empty(ns, 64, 16, 8, device=DEV)
    L.call("vae_wgrad_gemm", L.ptr(g), L.ptr(x8), L.ptr(slab), 64, 64, M, 64, 8, WGRAD_CONV3_C8, H, W, ns, L.current_stream())
    dW_s = slab.sum(0)[:, :9, :Cin].permute(0, 2, 1).reshape(64, Cin, 3, 3).cpu()
    assert rel_l2(sums_s.cpu(), sums) <= 1e-5
    assert rel_l2(g.float().cpu(), nhwc(gy).float()) <= 4e-3                 # bf16 rounding of g
    e_one, e_sep = rel_l2(dW, dW_ref.float()), rel_l2(dW_s, dW_ref.float())
    print(f"one-pass {e_one:.2e}  separate {e_sep:.2e}")
    assert e_one <= 2e-5 and e_sep <= 2e-2 and e_one <= e_sep


def test_one_pass_rejects_what_it_cannot_hold(L):
    case = make_case(1, 8, 8, 3, 5)
    sums, dW, (x8, yd, gd, co) = run_one_pass(L, *case)
    scratch = torch.empty(L.call("vae_bn_bwd_conv1_scratch_floats", 1, 8, 8), device=DEV)
    out = torch.empty(128, device=DEV)
    dw = torch.empty(64 * 4 * 9, device=DEV)
    with pytest.raises(RuntimeError, match="Cin=4 out of range"):
        L.call("vae_bn_bwd_conv1", L.ptr(gd), L.ptr(yd), L.ptr(x8), *[L.ptr(t) for t in co], L.ptr(scratch), L.ptr(out), L.ptr(dw),
               1, 8, 8, 4, L.current_stream())
    with pytest.raises(RuntimeError, match="null pointer"):
        L.call("vae_bn_bwd_conv1", L.ptr(gd), L.ptr(yd), None, *[L.ptr(t) for t in co], L.ptr(scratch), L.ptr(out), L.ptr(dw),
               1, 8, 8, 3, L.current_stream())
    assert L.call("vae_bn_bwd_conv1_scratch_floats", 0, 8, 8) == -1


@pytest.mark.parametrize("n,H,W,Cin", [(8, 32, 32, 3), (2, 16, 16, 2), (1, 8, 8, 3), (96, 32, 32, 3), (3, 16, 32, 1), (5, 8, 8, 3)])
def test_bits_form_without_the_stored_conv_output(L, n, H, W, Cin):
    """forward in two passes (vae_conv3x3_c8_stats / _bn_act: the conv output is never stored, one ReLU bit per element
    is), backward from the bits with y rebuilt per tile: d gamma, d beta and dW against the stored-output kernel on the
    SAME conv output (1e-4: a rebuilt y may differ in its last bf16 bit) and against the fp64 formulas; reproducible;
    includes a last tile that is not full (320 pixels) and workgroups that walk several tiles"""
    g = torch.Generator().manual_seed(31 * n + H + Cin)
    x = torch.rand(n, Cin, H, W, generator=g)
    w = (torch.randn(64, Cin, 3, 3, generator=g) / 3)
    b = torch.randn(64, generator=g) / 2
    gin = (torch.randn(n, 64, H, W, generator=g) * 1e-3 + 2e-4).to(torch.bfloat16)
    s = L.current_stream()
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    x8 = torch.zeros(n, H, W, 8, device=DEV, dtype=torch.bfloat16)
    L.call("vae_nchw_to_nhwc8", L.ptr(xd), L.ptr(x8), n, Cin, H, W, s)
    wp = torch.zeros(64, 16, 8, device=DEV, dtype=torch.bfloat16)
    L.call("vae_pack_bf16", L.ptr(wd), L.ptr(wp), L.long4(64, Cin, 9, 1), L.long4(128, 1, 8, 0), 0, s)
    rows = L.call("vae_conv3x3_c8_stat_rows", n, H, W, 64)
    ssum, ssq = torch.empty(rows, 64, device=DEV), torch.empty(rows, 64, device=DEV)
    y = torch.empty(n, H, W, 64, device=DEV, dtype=torch.bfloat16)
    L.call("vae_conv3x3_c8_gemm", L.ptr(x8), L.ptr(wp), L.ptr(bd), L.ptr(y), L.ptr(ssum), L.ptr(ssq), n, H, W, 64, s)
    gam, bet = (torch.rand(64, generator=g) + 0.5).to(DEV), (torch.randn(64, generator=g) * 0.3).to(DEV)
    coef = torch.empty(4, 64, device=DEV)
    L.call("vae_bn_finalize", L.ptr(ssum), L.ptr(ssq), rows, 64, 1, float(n * H * W), L.ptr(gam), L.ptr(bet), None, None, 0.1, 1e-5, 0,
           L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(coef[2]), L.ptr(coef[3]), s)
    act = torch.empty_like(y)
    bits = torch.empty(n * H * W, 8, device=DEV, dtype=torch.uint8)
    L.call("vae_conv3x3_c8_bn_act", L.ptr(x8), L.ptr(wp), L.ptr(bd), L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(act), L.ptr(bits), n, H, W, 64, s)
    gd = nhwc(gin).to(DEV)
    nf = L.call("vae_bn_bwd_conv1_scratch_floats", n, H, W)

    def stored():
        scratch, sums, dW = torch.full((nf,), float("nan"), device=DEV), torch.full((128,), float("nan"), device=DEV), torch.full((64, Cin, 3, 3), float("nan"), device=DEV)
        L.call("vae_bn_bwd_conv1", L.ptr(gd), L.ptr(y), L.ptr(x8), L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(coef[2]), L.ptr(coef[3]),
               L.ptr(scratch), L.ptr(sums), L.ptr(dW), n, H, W, Cin, s)
        return sums.cpu(), dW.cpu()

    def from_bits():
        scratch, sums, dW = torch.full((nf,), float("nan"), device=DEV), torch.full((128,), float("nan"), device=DEV), torch.full((64, Cin, 3, 3), float("nan"), device=DEV)
        L.call("vae_bn_bwd_conv1_bits", L.ptr(gd), L.ptr(bits), L.ptr(x8), L.ptr(wd), L.ptr(bd), L.ptr(coef[0]), L.ptr(coef[2]), L.ptr(coef[3]),
               L.ptr(scratch), L.ptr(sums), L.ptr(dW), n, H, W, Cin, s)
        return sums.cpu(), dW.cpu()

    s0, w0 = stored()
    s1, w1 = from_bits()
    assert rel_l2(s1[:64], s0[:64]) <= 1e-4 and rel_l2(w1, w0) <= 1e-4
    assert torch.equal(s1[64:], s0[64:])                                     # sum d: the same d, the same order
    s2, w2 = from_bits()
    assert torch.equal(s1, s2) and torch.equal(w1, w2)
    # and against the fp64 formulas on the stored conv output
    yc = y.float().cpu().permute(0, 3, 1, 2).to(torch.bfloat16)
    c = coef.cpu()
    s_dx, s_d, dW_ref, _ = reference(x, yc, gin, c[0], c[1], c[2], c[3])
    assert rel_l2(s1[:64], s_dx.float()) <= 1e-4 and rel_l2(s1[64:], s_d.float()) <= 1e-5
    assert rel_l2(w1, dW_ref.float()) <= 1e-4
