"""The allocation-free loss entry point (vae_recon_loss_fwd_bwd_static) against the entry point it is a variant
of (vae_recon_loss_fwd_bwd + vae_elbo_combine with a caller-provided partial buffer), through the C ABI: same
gradient and same sum, bit for bit (same kernels, same grid, same summation order), on two calls in a row, the
second on a smaller input -- the fixed array keeps the first call's partial sums, which must not leak into the
second result."""
import pytest
import torch

from vae_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"


def reference(recon, target, kind, scale):
    n = recon.numel()
    grid = L.call("vae_recon_loss_grid", n)
    partial = torch.empty(grid, device=DEV)
    grad = torch.empty_like(recon)
    out3 = torch.zeros(3, device=DEV)
    L.call("vae_recon_loss_fwd_bwd", recon.data_ptr(), target.data_ptr(), grad.data_ptr(), partial.data_ptr(), n, kind,
           scale, L.current_stream())
    L.call("vae_elbo_combine", partial.data_ptr(), grid, None, 0, 0.0, 1.0, out3.data_ptr(), L.current_stream())
    return grad, out3


@pytest.mark.parametrize("kind", [0, 2])
def test_static_entry_point_equals_the_allocating_one(kind):
    L.load()
    g = torch.Generator(device=DEV).manual_seed(7 + kind)
    # first call: 512 x 3 x 32 x 32 (768 workgroups = 768 partial sums in the fixed array); second: a ragged, much
    # smaller input (8 workgroups), which must not see the first call's leftovers
    for n in (512 * 3 * 32 * 32, 5 * 3 * 32 * 32 + 3):
        recon = torch.rand(n, device=DEV, generator=g) * 0.98 + 0.01
        target = torch.rand(n, device=DEV, generator=g)
        want_grad, want3 = reference(recon, target, kind, 0.125)
        grad = torch.full_like(recon, float("nan"))
        out3 = torch.full((3,), float("nan"), device=DEV)
        # prime call: the first call on a device resolves the address of the array (and may load the code object)
        L.call("vae_recon_loss_fwd_bwd_static", recon.data_ptr(), target.data_ptr(), None, out3.data_ptr(), n, kind,
               0.125, L.current_stream())
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        held0 = torch.cuda.memory_allocated()
        L.call("vae_recon_loss_fwd_bwd_static", recon.data_ptr(), target.data_ptr(), grad.data_ptr(), out3.data_ptr(), n, kind,
               0.125, L.current_stream())
        torch.cuda.synchronize()
        # no device memory was taken by the call: neither from the driver (hipMemGetInfo) nor from torch's allocator
        assert torch.cuda.mem_get_info()[0] == free0 and torch.cuda.memory_allocated() == held0
        assert torch.equal(grad, want_grad), (kind, n)
        assert torch.equal(out3, want3), (kind, n, out3.tolist(), want3.tolist())
        # forward only (grad = NULL): same sum
        out3b = torch.zeros(3, device=DEV)
        L.call("vae_recon_loss_fwd_bwd_static", recon.data_ptr(), target.data_ptr(), None, out3b.data_ptr(), n, kind, 0.125,
               L.current_stream())
        assert torch.equal(out3b, want3)


def test_static_entry_point_rejects_oversize_and_bad_arguments():
    x = torch.zeros(8, device=DEV)
    out3 = torch.zeros(3, device=DEV)
    too_many = L.call("vae_recon_loss_static_max_elems") + 4
    with pytest.raises(RuntimeError, match="exceeds the maximum"):
        L.call("vae_recon_loss_fwd_bwd_static", x.data_ptr(), x.data_ptr(), None, out3.data_ptr(), too_many, 0, 1.0,
               L.current_stream())
    with pytest.raises(RuntimeError, match="n_elems"):
        L.call("vae_recon_loss_fwd_bwd_static", x.data_ptr(), x.data_ptr(), None, out3.data_ptr(), 0, 0, 1.0, L.current_stream())
    with pytest.raises(RuntimeError, match="null"):
        L.call("vae_recon_loss_fwd_bwd_static", x.data_ptr(), x.data_ptr(), None, None, 8, 0, 1.0, L.current_stream())
