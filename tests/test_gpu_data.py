"""GPU tests of the device-side batch feed (through the C ABI) and of the checkpoint / sampling helpers around the model.
The feed must give exactly what torchvision's ToTensor gives on the host: HWC uint8 -> CHW fp32, value / 255."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(images_u8, idx):
    x = torch.from_numpy(images_u8)[idx]                       # [B, H, W, C]
    return x.permute(0, 3, 1, 2).contiguous().float().div(255)


@pytest.mark.parametrize("n,h,w,c,batch", [(64, 32, 32, 3, 16), (10, 5, 7, 1, 10), (33, 8, 8, 4, 5), (9, 64, 64, 3, 9)])
def test_feed_matches_totensor_bit_for_bit(n, h, w, c, batch):
    from vae_amd import _lib as L
    rng = np.random.default_rng(n + h)
    images = rng.integers(0, 256, size=(n, h, w, c), dtype=np.uint8)
    images[0] = 0
    images[-1] = 255
    idx = torch.from_numpy(rng.integers(-n, n, size=batch)).long()
    idx[0], idx[-1] = 0, -1
    d = torch.from_numpy(images).to(DEV)
    out = torch.full((batch, c, h, w), float("nan"), device=DEV)
    L.call("vae_feed_u8", d.data_ptr(), n, idx.to(DEV).data_ptr(), out.data_ptr(), batch, c, h, w, L.current_stream())
    assert torch.equal(out.cpu(), _ref(images, idx))
    # no index: the first `batch` images
    out.fill_(float("nan"))
    L.call("vae_feed_u8", d.data_ptr(), n, None, out.data_ptr(), batch, c, h, w, L.current_stream())
    assert torch.equal(out.cpu(), _ref(images, torch.arange(batch)))
    # an index outside the dataset gives a zero image, the rest is untouched by it
    bad = idx.clone()
    bad[1 % batch] = n + 3
    L.call("vae_feed_u8", d.data_ptr(), n, bad.to(DEV).data_ptr(), out.data_ptr(), batch, c, h, w, L.current_stream())
    ref = _ref(images, idx)
    ref[1 % batch] = 0
    assert torch.equal(out.cpu(), ref)


def test_feed_rejects_bad_arguments():
    from vae_amd import _lib as L
    d = torch.zeros(4, 8, 8, 3, dtype=torch.uint8, device=DEV)
    out = torch.empty(8, 3, 8, 8, device=DEV)
    with pytest.raises(RuntimeError, match="channels=2"):
        L.call("vae_feed_u8", d.data_ptr(), 4, None, out.data_ptr(), 4, 2, 8, 8, L.current_stream())
    with pytest.raises(RuntimeError, match="exceeds"):
        L.call("vae_feed_u8", d.data_ptr(), 4, None, out.data_ptr(), 8, 3, 8, 8, L.current_stream())
    with pytest.raises(RuntimeError, match="null pointer"):
        L.call("vae_feed_u8", None, 4, None, out.data_ptr(), 4, 3, 8, 8, L.current_stream())


def test_device_feed_epochs():
    from vae_amd import data
    rng = np.random.default_rng(3)
    images = rng.integers(0, 256, size=(103, 16, 16, 3), dtype=np.uint8)
    tr, va = data.split_indices(103, 0.9, torch.Generator().manual_seed(1337))
    feed = data.DeviceFeed(images, 32, indices=va, shuffle=False)
    got = torch.cat([b.cpu() for b in feed])
    assert torch.equal(got, _ref(images, va))                          # validation order is the split's order
    g = torch.Generator().manual_seed(5)
    feed = data.DeviceFeed(images, 32, indices=tr, shuffle=True, generator=g)
    assert len(feed) == 3
    e1 = [b.cpu() for b in feed]
    e2 = [b.cpu() for b in feed]
    assert [b.shape[0] for b in e1] == [32, 32, 28]                    # the short last batch is kept
    a, b = torch.cat(e1), torch.cat(e2)
    assert not torch.equal(a, b)                                       # a new permutation per epoch
    # each epoch is a permutation of the training subset
    key = lambda t: sorted(t.flatten(1).sum(1).tolist())
    assert key(a) == key(_ref(images, tr)) == key(b)
    out = torch.empty(4, 3, 16, 16, device=DEV)
    assert feed.batch(torch.tensor([0, 1, 2, -1]), out=out) is out


def test_checkpoint_and_sampling_through_the_model(tmp_path):
    from vae_amd import data
    from vae_amd.vae import VAEConv
    torch.manual_seed(11)
    m = VAEConv(3, 3, 64, 0.1, 16).to(DEV)
    x = torch.rand(8, 3, 16, 16, device=DEV)
    m.train()
    m(x)                                                               # moves the BatchNorm running statistics
    path = str(tmp_path / "best-model.pth")
    data.save_checkpoint(m, path)
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert len(raw) == 85 and "conv1.net.0.weight" in raw and "bn1.num_batches_tracked" in raw   # the reference's 85 entries
    m2 = VAEConv(3, 3, 64, 0.1, 16).to(DEV)
    res = data.load_checkpoint(m2, path)
    assert not res.missing_keys and not res.unexpected_keys
    m.eval(), m2.eval()
    with torch.no_grad():
        (mu1, lv1), (mu2, lv2) = m.encode(x), m2.encode(x)             # running statistics came along
    assert torch.equal(mu1, mu2) and torch.equal(lv1, lv2)
    m.train(), m2.train()
    r1 = data.reconstruct(m, x)                                        # stochastic: eps is drawn in eval mode too (vae.py:108-111)
    assert r1.shape == x.shape and float(r1.min()) > 0 and float(r1.max()) < 1
    assert m.training and m2.training                                  # the helpers restore the mode
    s1 = data.sample(m, 6, torch.Generator(device=DEV).manual_seed(3))
    s2 = data.sample(m2, 6, torch.Generator(device=DEV).manual_seed(3))
    assert s1.shape == (6, 3, 16, 16) and torch.equal(s1, s2)
    assert float(s1.min()) > 0 and float(s1.max()) < 1
    data.save_image_grid(s1, str(tmp_path / "samples.ppm"), nrow=3)


def test_interpolate_latents_and_command_line(tmp_path):
    """interpolate_latents (train.py:115-155): the end points are the decodes of the two images' means and the steps
    walk the straight line between them; `python -m vae_amd` runs sample / reconstruct / interpolate from a
    checkpoint the way the reference's load_and_test_model does (train.py:182-229)."""
    from vae_amd import data
    from vae_amd.__main__ import main
    from vae_amd.vae import VAEConv
    torch.manual_seed(5)
    m = VAEConv(3, 3, 128, 0.1, 32).to(DEV).eval()
    x = torch.rand(2, 3, 32, 32, device=DEV)
    walk = data.interpolate_latents(m, x[0], x[1].unsqueeze(0), num_steps=5)
    assert walk.shape == (5, 3, 32, 32)
    with torch.no_grad():
        mu, _ = m.encode(x)
        ends = m.decode(mu)
        mid = m.decode((0.5 * mu[0:1] + 0.5 * mu[1:2]).contiguous())
    assert torch.allclose(walk[0], ends[0], atol=2e-3) and torch.allclose(walk[4], ends[1], atol=2e-3)
    assert torch.allclose(walk[2], mid[0], atol=2e-3)
    with pytest.raises(ValueError):
        data.interpolate_latents(m, x[0], x[1], num_steps=1)
    ckpt = str(tmp_path / "best-model.pth")
    data.save_checkpoint(m, ckpt)
    imgs = (torch.rand(12, 32, 32, 3) * 255).to(torch.uint8).numpy()
    np.save(str(tmp_path / "val.npy"), imgs)
    for cmd, extra, shape_rows in (("sample", ["--n", "6"], 1), ("reconstruct", ["--n", "4", "--images", str(tmp_path / "val.npy")], 2),
                                   ("interpolate", ["--steps", "7", "--images", str(tmp_path / "val.npy")], 1),
                                   ("interpolate", ["--steps", "4", "--precision", "fp32"], 1)):
        out = str(tmp_path / f"{cmd}.ppm")
        assert main([cmd, "--checkpoint", ckpt, "--config", "cfg2_1gpu", "--out", out] + extra) == 0
        with open(out, "rb") as f:
            assert f.read(2) == b"P6"
    with pytest.raises(SystemExit):
        main(["reconstruct", "--images", str(tmp_path / "val.npy"), "--n", "50", "--out", str(tmp_path / "x.ppm")])


def test_fit_runs_the_reference_epoch_loop(tmp_path):
    """train pass, eval pass, best-model checkpoint on improvement, one record per epoch (train.py:316-373)"""
    from vae_amd.data import DeviceFeed, load_checkpoint, split_indices
    from vae_amd.optim import FlatAdamW
    from vae_amd.train import Config, fit
    from vae_amd.vae import VAEConv
    torch.manual_seed(0)
    imgs = torch.randint(0, 256, (200, 32, 32, 3), dtype=torch.uint8, device="cuda")
    tr, va = split_indices(200, 0.9, torch.Generator().manual_seed(1))
    m = VAEConv(3, 3, 64, 0.1, 32).cuda()
    conf = Config(kl_weight=1.0, percep_weight=0.0, recon_loss="l1", lr=1e-3, n_epochs=3, save_model=True)
    path = str(tmp_path / "best-model.pth")
    seen = []
    hist = fit(m, FlatAdamW(m, lr=1e-3), conf, DeviceFeed(imgs, 32, tr, shuffle=True), DeviceFeed(imgs, 32, va), log=seen.append,
               checkpoint_path=path)
    assert hist == seen and [r["epoch"] for r in hist] == [1, 2, 3]
    assert all(torch.isfinite(torch.tensor([r["train/loss"], r["val/loss"]])).all() for r in hist)
    assert hist[-1]["train/loss"] < hist[0]["train/loss"]
    assert not m.training                                    # the loop leaves the model in eval mode, as the reference does
    m2 = VAEConv(3, 3, 64, 0.1, 32).cuda()
    load_checkpoint(m2, path)                                 # the checkpoint is a plain reference-format state_dict
    assert set(m2.state_dict()) == set(m.state_dict())
    # the autograd path gives the same kind of record.  Its optimizer starts without moment estimates, so the first
    # steps move every parameter by about lr and the first epoch's loss is back near the untrained one (measured: 802-806
    # against 804.5); the second epoch is well below it
    hist2 = fit(m, torch.optim.AdamW(m.parameters(), lr=1e-3), Config(kl_weight=1.0, percep_weight=0.0, recon_loss="l1", lr=1e-3,
                                                                    n_epochs=2, save_model=False),
                DeviceFeed(imgs, 32, tr), DeviceFeed(imgs, 32, va), fused=False)
    assert len(hist2) == 2 and hist2[1]["train/loss"] < hist[0]["train/loss"] - 10.0
