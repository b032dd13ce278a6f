"""CPU checks of the C-ABI boundary: the shared library loads without a GPU, exports every
function include/vae_hip.h declares, the ctypes table binds exactly that set, and the
host-only entry points (sizes, layouts, thresholds, geometry predicates) answer correctly.
No kernel is launched here."""
import ctypes
import os
import re

import pytest

import vae_amd._lib as L
from oracle import vae_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vae_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    names = re.findall(r"\b(vae_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_functions():
    names = declared_functions()
    assert len(names) >= 50
    for must in ("vae_model_forward", "vae_model_backward", "vae_conv3x3_gemm", "vae_reparam_kl_fwd",
                 "vae_recon_loss_fwd_bwd", "vae_adamw_clip_step", "vae_last_error", "vae_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.LIB_PATH), "build the library first: make -C vae_amd/csrc"
    lib = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/vae_hip.h but not exported: {missing}"


def test_ctypes_table_matches_header():
    declared = set(declared_functions())
    bound = set(L.SIGNATURES)
    assert bound <= declared, f"bound but not declared: {sorted(bound - declared)}"
    assert declared <= bound, f"declared but not bound: {sorted(declared - bound)}"


def test_no_torch_types_in_header():
    src = open(HEADER).read()
    assert 'extern "C"' in src
    for bad in ("at::", "torch::", "c10::", "std::"):
        assert bad not in src


def test_version_and_error_string():
    lib = L.load()
    assert lib.vae_version() == L.ABI_VERSION
    assert isinstance(lib.vae_last_error(), bytes)
    # the header's revision is the one the binding was written for
    with open(os.path.join(ROOT, "include", "vae_hip.h")) as f:
        assert f"#define VAE_HIP_VERSION {L.ABI_VERSION}\n" in f.read()


def test_binding_refuses_a_library_of_another_abi_revision(monkeypatch):
    """a caller built against an older header must not reach the kernels with a shorter struct"""
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "ABI_VERSION", L.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI revision"):
        L.load()


def test_model_cfg_of_another_size_is_rejected():
    lib = L.load()
    cfg = _cfg(8, 128, 32)
    assert cfg.struct_size == ctypes.sizeof(L.ModelCfg) and lib.vae_model_param_count(ctypes.byref(cfg)) > 0
    cfg.struct_size -= 16           # what a binding built before the last two fields were added would pass
    assert lib.vae_model_param_count(ctypes.byref(cfg)) == -1
    assert lib.vae_model_workspace_bytes(ctypes.byref(cfg)) == -1
    assert b"struct_size" in lib.vae_last_error()


def test_tuning_keys_are_range_checked():
    L.call("vae_set_tuning", b"conv3_wide", 0)
    with pytest.raises(RuntimeError, match="out of range"):
        L.call("vae_set_tuning", b"conv3_wide", 7)
    with pytest.raises(RuntimeError, match="unknown key"):
        L.call("vae_set_tuning", b"conv3_cu_budget", 1)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()


def _cfg(batch, latent, size, cin=3, cout=3):
    return L.ModelCfg(batch=batch, in_channels=cin, out_channels=cout, latent=latent, image_size=size, train=1,
                      p_dropout=0.1, seed=1337, step=0, rank=0)


@pytest.mark.parametrize("latent,size,total", [(128, 32, 4086787), (256, 64, 15109379), (512, 32, 8806147)])
def test_flat_parameter_layout_matches_state_dict_order(latent, size, total):
    """the flat fp32 buffer holds the 52 parameters back to back in the reference's state_dict order with the
    reference's shapes (vae/models/vae.py:34-74); totals are the parameter counts of the three benchmark models"""
    cfg = _cfg(8, latent, size)
    shapes = O.state_shapes(3, 3, latent, size)
    params = [(k, s) for k, s in shapes.items()
              if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    assert len(params) == 52
    off = 0
    for i, (name, shape) in enumerate(params):
        assert L.call("vae_model_param_offset", ctypes.byref(cfg), i) == off, name
        n = 1
        for d in shape:
            n *= d
        off += n
    assert off == total
    # the buffer length is the parameter count rounded up to whole float4s (the optimizer streams float4)
    assert L.call("vae_model_param_count", ctypes.byref(cfg)) == (total + 3) // 4 * 4
    # running_mean and running_var of the 11 BatchNorm layers
    assert L.call("vae_model_bn_count", ctypes.byref(cfg)) == sum(
        s[0] for k, s in shapes.items() if k.endswith(("running_mean", "running_var")))


def test_workspace_is_deterministic_and_grows_with_batch():
    a = L.call("vae_model_workspace_bytes", ctypes.byref(_cfg(64, 128, 32)))
    b = L.call("vae_model_workspace_bytes", ctypes.byref(_cfg(512, 128, 32)))
    assert a == L.call("vae_model_workspace_bytes", ctypes.byref(_cfg(64, 128, 32)))
    assert 0 < a < b < (8 << 30)
    assert L.call("vae_model_wpack_bytes", ctypes.byref(_cfg(64, 128, 32))) > 2 * 4086787 * 0.9


def test_precision_field_selects_the_workspace_layout_and_is_checked():
    """vae_model_cfg.precision: 0 = bf16 kernels, 1 = fp32 kernels (own workspace layout, no packed weights);
    anything else is rejected.  The ctypes mirror must match the C struct field for field."""
    lib = L.load()
    bf = _cfg(64, 128, 32)
    assert bf.precision == L.PRECISIONS["bf16"] == 0
    fp = _cfg(64, 128, 32)
    fp.precision = L.PRECISIONS["fp32"]
    a, b = L.call("vae_model_workspace_bytes", ctypes.byref(bf)), L.call("vae_model_workspace_bytes", ctypes.byref(fp))
    assert a > 0 and b > 0 and a != b
    # fp32 activations: at least 4 bytes for every element of the eleven saved pre-BatchNorm tensors
    assert b > 64 * 4 * (2 * 64 * 1024 + 2 * 128 * 256 + 2 * 256 * 64 + 2 * 256 * 16)
    # parameter layout does not depend on the precision
    assert L.call("vae_model_param_count", ctypes.byref(bf)) == L.call("vae_model_param_count", ctypes.byref(fp))
    bad = _cfg(64, 128, 32)
    bad.precision = 7
    assert lib.vae_model_workspace_bytes(ctypes.byref(bad)) < 0
    assert b"precision" in lib.vae_last_error()


def test_bad_model_config_is_an_error_not_a_crash():
    lib = L.load()
    bad = _cfg(8, 128, 20)          # three 2x2 pools need a multiple of 8
    assert lib.vae_model_workspace_bytes(ctypes.byref(bad)) < 0
    assert b"image_size" in lib.vae_last_error()


@pytest.mark.parametrize("p", [0.0, 0.1, 0.25, 0.5, 0.9])
def test_dropout_threshold_matches_oracle(p):
    assert L.call("vae_dropout_threshold", p) == O.dropout_threshold(p)


def test_geometry_helpers():
    assert L.call("vae_gemm_rows_per_block") in (64, 128)
    for M, N, taps, K in [(64, 64, 9, 524288), (256, 256, 9, 8192), (4096, 128, 1, 512), (128, 64, 4, 131072)]:
        s = L.call("vae_wgrad_ksplit", M, N, taps, K)
        chunk = -(-K // s)
        assert s >= 1 and (s - 1) * chunk < K
    assert L.call("vae_conv3_wgrad_patch_ok", 32, 32, 64, 64) == 1
    assert L.call("vae_conv3_wgrad_patch_ok", 4, 4, 256, 256) == 1
    assert L.call("vae_conv3_wgrad_patch_ok", 6, 6, 64, 64) == 0
    assert L.call("vae_conv3_wgrad_patch_ok", 32, 32, 3, 64) == 0
    for n, H, Ci, Co in [(512, 32, 64, 64), (512, 4, 256, 256), (1, 8, 64, 64)]:
        s = L.call("vae_conv3_wgrad_patch_ksplit", n, H, H, Ci, Co)
        assert 1 <= s <= (n * H * H + 63) // 64
    assert L.call("vae_recon_loss_grid", 1) >= 1 and L.call("vae_reparam_grid", 1) >= 1 and L.call("vae_adamw_grid", 1) >= 1


@pytest.mark.parametrize("n,size,max_wgs", [(512, 32, 256), (512, 64, 256), (4, 32, 256), (1, 32, 256), (64, 32, 16)])
def test_batched_weight_gradient_plan(n, size, max_wgs):
    """the K splits one launch over several conv3x3 weight gradients is planned with: every layer gets at least one
    workgroup per output tile, the launch stays within the workgroup budget whenever the tiles alone fit in it, no split
    is empty, and K tiles per workgroup are even across layers (within a factor of two of each other where a layer
    has enough of them to split)"""
    geo = [(size // 8, 256, 256), (size // 8, 256, 256), (size // 4, 256, 256), (size // 4, 128, 256), (size // 2, 128, 128)]
    jobs = (L.WgradJob * len(geo))()
    for q, (H, Ci, Co) in zip(jobs, geo):
        q.n_img, q.H, q.W, q.Cin, q.Cout = n, H, H, Ci, Co
    ns = (ctypes.c_int * len(geo))()
    L.call("vae_conv3_wgrad_multi_plan", jobs, len(geo), max_wgs, ns)
    tiles = [(Ci // 64) * (Co // 64) for _, Ci, Co in geo]
    ktiles = [-(-n * H * H // 64) for H, _, _ in geo]
    assert all(1 <= s <= k for s, k in zip(ns, ktiles))
    assert sum(t * s for t, s in zip(tiles, ns)) <= max(max_wgs, sum(tiles))
    per = [-(-k // s) for k, s in zip(ktiles, ns)]
    assert all((s - 1) * p < k for s, p, k in zip(ns, per, ktiles))          # no empty split
    split = [p for p, s in zip(per, ns) if s > 1]
    if split:
        assert max(split) <= 2 * min(split)
    jobs[0].H = 6
    with pytest.raises(RuntimeError, match="not covered"):
        L.call("vae_conv3_wgrad_multi_plan", jobs, len(geo), max_wgs, ns)
    with pytest.raises(RuntimeError, match="out of range"):
        L.call("vae_conv3_wgrad_multi_plan", jobs, 9, max_wgs, ns)


def test_batched_weight_gradient_plan_over_many_batches():
    """every batch size 1..96 and a spread of larger ones, both image sizes, both backward stages: the plan never has an
    empty K range, never asks for more K ranges than K tiles, and stays within the budget"""
    import random
    rnd = random.Random(5)
    batches = list(range(1, 97)) + [rnd.randrange(97, 5000) for _ in range(60)]
    for size in (32, 64):
        stages = ([(size // 8, 256, 256), (size // 8, 256, 256), (size // 4, 256, 256), (size // 4, 128, 256)],
                  [(size // 2, 128, 128), (size // 2, 64, 128), (size, 64, 64)])
        for n in batches:
            for geo in stages:
                jobs = (L.WgradJob * len(geo))()
                for q, (H, Ci, Co) in zip(jobs, geo):
                    q.n_img, q.H, q.W, q.Cin, q.Cout = n, H, H, Ci, Co
                ns = (ctypes.c_int * len(geo))()
                L.call("vae_conv3_wgrad_multi_plan", jobs, len(geo), 256, ns)
                tiles = [(Ci // 64) * (Co // 64) for _, Ci, Co in geo]
                ktiles = [-(-n * H * H // 64) for H, _, _ in geo]
                assert all(1 <= s <= k for s, k in zip(ns, ktiles)), (size, n, list(ns))
                assert all((s - 1) * -(-k // s) < k for s, k in zip(ns, ktiles)), (size, n, list(ns))
                assert sum(t * s for t, s in zip(tiles, ns)) <= max(256, sum(tiles)), (size, n, list(ns))
