"""CPU tests of the host-side mirror of the reference interface: the yaml config loader
(vae/training/train.py:443-453), loss-kind selection, data-parallel bucket planning and the
shipped benchmark configs.  No GPU, no kernel launches."""
import os

import pytest
import torch
import torch.nn.functional as F

from vae_amd import dp, train

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIG_DIR = os.path.join(ROOT, "config")


def test_load_config_env_overrides_base(tmp_path):
    (tmp_path / "base.yaml").write_text("seed: 1337\nlr: 1e-4\nkl_weight: 1.0\nname: run\nproject: vae-${name}\n")
    (tmp_path / "server.yaml").write_text("kl_weight: 4.0\nbatch_size: 512\n")
    cfg = train.load_config("server", str(tmp_path))
    assert cfg.seed == 1337 and cfg.batch_size == 512
    assert cfg.kl_weight == 4.0                       # env file wins
    assert isinstance(cfg.lr, float) and cfg.lr == 1e-4   # "1e-4" is a float, as OmegaConf reads it
    assert cfg.project == "vae-run"                   # ${key} interpolation
    cfg.device = "cuda"                               # attribute writes, as main() does (train.py:426-427)
    assert cfg["device"] == "cuda"
    with pytest.raises(AttributeError):
        cfg.not_a_key


def test_load_config_missing_env_file_is_base_only(tmp_path):
    (tmp_path / "base.yaml").write_text("seed: 7\n")
    assert dict(train.load_config("nowhere", str(tmp_path))) == {"seed": 7}


@pytest.mark.parametrize("env,latent,size,batch,klw", [
    ("cfg1_cpu", 128, 32, 64, 1.0), ("cfg2_1gpu", 128, 32, 512, 1.0), ("cfg3_8gpu", 128, 32, 512, 1.0),
    ("cfg4_celeba64", 256, 64, 1024, 1.0), ("cfg5_beta", 512, 32, 512, 4.0)])
def test_shipped_benchmark_configs(env, latent, size, batch, klw):
    cfg = train.load_config(env, CONFIG_DIR)
    assert cfg.latent_dims == latent and cfg.image_size == size and cfg.batch_size == batch
    assert float(cfg.kl_weight) == klw
    assert cfg.num_channels == 3 and float(cfg.p_dropout) == 0.1 and float(cfg.lr) == 1e-4
    assert float(cfg.percep_weight) == 0.0
    assert train.recon_kind_from_config(cfg) in train.RECON_KINDS.values()


def test_linear_anneal_kl():
    cfg = train.Config(n_epochs=10, kl_clip=0.5, kl_start=0.0, kl_end=4.0)
    assert train.linear_anneal_kl(cfg, 0) == 0.0
    assert train.linear_anneal_kl(cfg, 2) == pytest.approx(0.8)
    assert train.linear_anneal_kl(cfg, 5) == 2.0 == train.linear_anneal_kl(cfg, 9)     # clipped at kl_clip
    cfg2 = train.Config(n_epochs=4, kl_clip=1.0, kl_start=1.0, kl_end=0.5)
    assert train.linear_anneal_kl(cfg2, 4) == 0.5 and train.linear_anneal_kl(cfg2, 2) == 0.75


def test_loss_kind_selection():
    assert train._kind_of(F.l1_loss) == 0 and train._kind_of(F.mse_loss) == 1
    # the torch callable is the UNCLAMPED form (vae/training/train.py:46); the clamped draft form has its own name
    assert train._kind_of(F.binary_cross_entropy) == 2 and train._kind_of(F.binary_cross_entropy_with_logits) == 4
    assert train._kind_of("bce") == 2 and train._kind_of("bce_logits") == 3 and train._kind_of("bce_logits_raw") == 4
    with pytest.raises(ValueError):
        train._kind_of(F.smooth_l1_loss)


def test_loss_refuses_cpu_tensors_and_lpips():
    cfg = train.Config(kl_weight=1.0, percep_weight=0.0)
    x = torch.rand(2, 3, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        train.l1_kl_percep_loss(x, x, torch.zeros(2, 4), torch.zeros(2, 4), cfg)
    cfg.percep_weight = 0.5
    with pytest.raises(NotImplementedError):
        train.vae_loss(x, x, torch.zeros(2, 4), torch.zeros(2, 4), F.l1_loss, lambda *a, **k: 0, cfg)
    # the perceptual term is never silently dropped: every loss entry point refuses a non-zero weight,
    # with or without an lpips_fn
    with pytest.raises(NotImplementedError):
        train.vae_loss(x, x, torch.zeros(2, 4), torch.zeros(2, 4), F.l1_loss, None, cfg)
    with pytest.raises(NotImplementedError):
        train.l1_kl_percep_loss(x, x, torch.zeros(2, 4), torch.zeros(2, 4), cfg)
    with pytest.raises(NotImplementedError):
        train.train_step(None, None, x, cfg)
    with pytest.raises(NotImplementedError):
        train.val_step(None, x, cfg)
    with pytest.raises(NotImplementedError):
        train.FusedStep(None, None, cfg)


def test_stage_ranges_partition_the_flat_buffer():
    numels = [10 + i for i in range(52)]
    offsets, o = [], 0
    for n in numels:
        offsets.append((o, n))
        o += n
    total = o
    r = dp.stage_ranges(offsets, total)
    # stage 0 = heads + decoder (params 32..51), stage 1 = conv3+conv4, stage 2 = conv1+conv2
    assert r[0] == (offsets[32][0], total) and r[1] == (offsets[16][0], offsets[32][0]) and r[2] == (0, offsets[16][0])
    covered = sorted(r)
    assert covered[0][0] == 0 and covered[-1][1] == total
    assert all(covered[i][1] == covered[i + 1][0] for i in range(2))


@pytest.mark.parametrize("begin,end,size", [(0, 100, 30), (5, 6, 100), (10, 10, 4), (0, 64, 16)])
def test_split_buckets(begin, end, size):
    b = dp.split_buckets(begin, end, size)
    if begin == end:
        assert b == []
        return
    assert b[0][0] == begin and b[-1][1] == end
    assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert all(0 < e - a <= size for a, e in b)


def test_reducer_is_a_no_op_without_a_process_group():
    g = torch.arange(12, dtype=torch.float32)
    red = dp.BucketedReducer(g, [(8, 12), (4, 8), (0, 4)], bucket_bytes=8)
    for s in range(3):
        red.stage_done(s)
    red.wait()
    assert torch.equal(g, torch.arange(12, dtype=torch.float32))
