"""The 8-wave (256 pixels x 128 channels per workgroup) form of the 3x3 convolution against the 4-wave form and
against torch: both kernels run the same MFMA chain per output element, so their bf16 outputs must be equal bit for
bit -- which kernel a launch picks (it depends on the batch size) must never change a result -- and the per-channel
statistics must agree to fp32 rounding."""
import pytest
import torch
import torch.nn.functional as F

from vae_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"


def conv(mode, x, w, b, n, H, Ci, Co, ws64=1):
    """mode: conv3_wide setting; ws64: conv3_ws64 setting (1 = never: the comparisons below are between the 4-wave
    and 8-wave kernels unless a test asks for the weight-stationary one)."""
    L.call("vae_set_tuning", b"conv3_wide", mode)
    L.call("vae_set_tuning", b"conv3_ws64", ws64)
    try:
        rows = L.call("vae_conv3x3_stat_rows", n, H, H, Ci, Co)
        y = torch.empty(n, H, H, Co, device=DEV, dtype=torch.bfloat16)
        ss = torch.zeros(rows + 32, Co, device=DEV)
        sq = torch.zeros(rows + 32, Co, device=DEV)
        L.call("vae_conv3x3_gemm", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), ss.data_ptr(), sq.data_ptr(),
               n, H, H, Ci, Co, L.current_stream())
        torch.cuda.synchronize()
        return y, ss[:rows].sum(0), sq[:rows].sum(0)
    finally:
        L.call("vae_set_tuning", b"conv3_wide", 0)
        L.call("vae_set_tuning", b"conv3_ws64", 0)


# (images, H = W, Cin, Cout): several tiles per workgroup with one and two channel chunks; one tile per workgroup with
# four chunks; two tiles of channels (Cout 256); whole images inside a tile (8x8) and image rows (32x32, 64x64);
# a launch with fewer tiles than workgroups could hold
@pytest.mark.parametrize("n,H,Ci,Co", [(512, 16, 64, 128), (384, 16, 128, 128), (512, 8, 256, 256), (96, 32, 128, 128),
                                       (4, 64, 64, 256), (12, 8, 64, 128)])
def test_wide_kernel_equals_the_four_wave_kernel_and_torch(n, H, Ci, Co):
    L.load()
    g = torch.Generator(device=DEV).manual_seed(H * 1000 + Ci + Co + n)
    x = torch.randn(n, H, H, Ci, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(Co, 9, Ci, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(Co, device=DEV, generator=g)
    y4, s4, q4 = conv(1, x, w, b, n, H, Ci, Co)
    y8, s8, q8 = conv(2, x, w, b, n, H, Ci, Co)
    assert torch.equal(y4, y8)
    assert ((s4 - s8).abs().max() / s4.abs().max()).item() <= 1e-5
    assert ((q4 - q8).abs().max() / q4.abs().max()).item() <= 1e-5
    # and against torch on the same bf16 operands (fp32 accumulation, one rounding of the output)
    wt = w.float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt, b, padding=1).permute(0, 2, 3, 1)
    err = ((y8.float() - ref).norm() / ref.norm()).item()
    assert err <= 3e-3, err                      # bf16 output rounding: 2^-9 relative per element
    assert ((s8 - ref.sum((0, 1, 2))).abs().max() / ref.sum((0, 1, 2)).abs().max()).item() <= 1e-3


@pytest.mark.parametrize("n,H,Ci,Co", [(512, 4, 256, 256), (40, 8, 256, 128), (3, 4, 128, 64), (2, 32, 128, 128)])
def test_split_k_inside_the_workgroup_equals_the_plain_form(n, H, Ci, Co):
    """two wave groups of one workgroup each sum half of the channel chunks (the form picked for launches with at most
    one tile per CU): the same sums in two parts, so equal to the plain form to fp32 rounding of the accumulator"""
    L.load()
    g = torch.Generator(device=DEV).manual_seed(n + H + Ci + Co)
    x = torch.randn(n, H, H, Ci, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(Co, 9, Ci, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(Co, device=DEV, generator=g)
    out = {}
    for ks in (1, 2):
        L.call("vae_set_tuning", b"conv3_ksplit", ks)
        try:
            out[ks] = conv(1, x, w, b, n, H, Ci, Co)
        finally:
            L.call("vae_set_tuning", b"conv3_ksplit", 0)
    (y1, s1, q1), (y2, s2, q2) = out[1], out[2]
    assert ((y1.float() - y2.float()).norm() / y1.float().norm()).item() <= 2e-4      # a bf16 rounding flips on ~1e-4 of the elements
    assert ((s1 - s2).abs().max() / s1.abs().max()).item() <= 1e-5
    assert ((q1 - q2).abs().max() / q1.abs().max()).item() <= 1e-5


# (images, H = W, Cout) at 64 input channels: many tiles per workgroup (the first block's second conv and its input
# gradient at batch 512), two channel tiles (the second block's first conv), 16-wide images (whole images inside a
# tile), fewer tiles than CUs, one tile only
@pytest.mark.parametrize("n,H,Co", [(512, 32, 64), (512, 16, 128), (64, 32, 64), (3, 16, 192), (1, 16, 64), (20, 32, 128)])
def test_weight_stationary_kernel_equals_the_four_wave_kernel_and_torch(n, H, Co):
    """conv3_ws64_kernel (weights resident in LDS, 64 input channels) runs the same MFMA chain per output element as
    the 4-wave kernel: equal bf16 outputs bit for bit, statistics equal to fp32 rounding; with and without the
    statistics / bias arguments (the input-gradient launches pass neither)."""
    L.load()
    Ci = 64
    g = torch.Generator(device=DEV).manual_seed(H * 1000 + Co + n)
    x = torch.randn(n, H, H, Ci, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(Co, 9, Ci, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(Co, device=DEV, generator=g)
    y4, s4, q4 = conv(1, x, w, b, n, H, Ci, Co, ws64=1)
    yw, sw, qw = conv(1, x, w, b, n, H, Ci, Co, ws64=2)
    assert torch.equal(y4, yw)
    if n * H * H * (Co // 64) >= 256 * 256:              # mode 3 = mode 0 (launches that fill the chip) with the pipelined K loop
        yp, sp, qp = conv(1, x, w, b, n, H, Ci, Co, ws64=3)
        assert torch.equal(y4, yp) and torch.equal(sw, sp) and torch.equal(qw, qp)
    assert ((s4 - sw).abs().max() / s4.abs().max()).item() <= 1e-5
    assert ((q4 - qw).abs().max() / q4.abs().max()).item() <= 1e-5
    wt = w.float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt, b, padding=1).permute(0, 2, 3, 1)
    assert ((yw.float() - ref).norm() / ref.norm()).item() <= 3e-3
    # no bias, no statistics
    outs = []
    for ws in (1, 2):
        L.call("vae_set_tuning", b"conv3_wide", 1)
        L.call("vae_set_tuning", b"conv3_ws64", ws)
        try:
            y = torch.empty(n, H, H, Co, device=DEV, dtype=torch.bfloat16)
            L.call("vae_conv3x3_gemm", x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, None, n, H, H, Ci, Co, L.current_stream())
            torch.cuda.synchronize()
            outs.append(y)
        finally:
            L.call("vae_set_tuning", b"conv3_wide", 0)
            L.call("vae_set_tuning", b"conv3_ws64", 0)
    assert torch.equal(outs[0], outs[1])
