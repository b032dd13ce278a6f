"""Parity of the statistics fold with many partial rows, the coalesced slab reducer, the
one-launch weight packer, the split-K linear GEMM and the fused clip + AdamW step."""
import ctypes

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


@pytest.mark.parametrize("rows,C,fold", [(65, 64, 1), (4096, 64, 1), (300, 64, 4), (1000, 256, 1)])
def test_bn_finalize_many_rows(L, rows, C, fold):
    g = torch.Generator().manual_seed(rows + C)
    ncols = fold * C
    ps = torch.randn(rows, ncols, generator=g)
    pq = torch.rand(rows, ncols, generator=g) * 3 + 4
    count = float(rows * 128 * fold)
    scratch = L.load().vae_gemm_rows_per_block() and 32
    psd = torch.cat([ps, torch.full((scratch, ncols), float("nan"))]).to(DEV)
    pqd = torch.cat([pq, torch.full((scratch, ncols), float("nan"))]).to(DEV)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    gd, bd, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    sc, sh, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
    L.call("vae_bn_finalize", L.ptr(psd), L.ptr(pqd), rows, C, fold, count, L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd),
           0.1, 1e-5, 0, L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.current_stream())
    m_ref = ps.double().sum(0).view(fold, C).sum(0) / count
    v_ref = (pq.double().sum(0).view(fold, C).sum(0) / count - m_ref ** 2).clamp_min(0)
    assert rel_l2(mean.cpu(), m_ref) <= 1e-5
    assert rel_l2(rstd.cpu(), 1 / torch.sqrt(v_ref + 1e-5)) <= 1e-5
    assert rel_l2(rmd.cpu(), 0.9 * rm + 0.1 * m_ref) <= 1e-5
    assert rel_l2(rvd.cpu(), 0.9 * rv + 0.1 * v_ref * count / (count - 1)) <= 1e-5


def test_reduce_slabs_scatter_and_multi_pack(L):
    g = torch.Generator().manual_seed(11)
    Co, Ci, S = 8, 16, 5
    slab = torch.randn(S, Co, 9, Ci, generator=g)
    sd = slab.to(DEV)
    out = torch.full((Co, Ci, 3, 3), float("nan"), device=DEV)
    # slab [co][tap][ci] -> weight [co][ci][tap]
    L.call("vae_reduce_slabs_scatter", L.ptr(sd), S, Co * 9 * Ci, 0, L.ptr(out), L.long4(Co, 9, Ci, 1),
           L.long4(9 * Ci, 1, 9, 0), None, 0, 0, L.current_stream())
    ref = slab.sum(0).permute(0, 2, 1).reshape(Co, Ci, 3, 3)
    assert rel_l2(out.cpu(), ref) <= 1e-6
    # padding limits: only taps < 7 and channels < 10 are written
    out2 = torch.zeros(Co, 10, 7, device=DEV)
    L.call("vae_reduce_slabs_scatter", L.ptr(sd), S, Co * 9 * Ci, 0, L.ptr(out2), L.long4(Co, 9, Ci, 1),
           L.long4(70, 1, 7, 0), L.long4(Co, 7, 10, 1), 0, 0, L.current_stream())
    assert rel_l2(out2.cpu(), slab.sum(0)[:, :7, :10].permute(0, 2, 1)) <= 1e-6

    # multi-pack: two jobs in one launch (bf16 pack with a permutation, fp32 permuting copy)
    w = torch.randn(Co, Ci, 3, 3, generator=g)
    b = torch.randn(12, generator=g)
    wd, bd = w.to(DEV), b.to(DEV)
    wp = torch.zeros(Co, 9, Ci, device=DEV, dtype=torch.bfloat16)
    bp = torch.zeros(12, device=DEV)

    class Job(ctypes.Structure):
        _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dims", ctypes.c_long * 4),
                    ("strides", ctypes.c_long * 4), ("base", ctypes.c_long), ("f32", ctypes.c_int)]
    jobs = (Job * 2)()
    jobs[0] = Job(wd.data_ptr(), wp.data_ptr(), (ctypes.c_long * 4)(Co, Ci, 9, 1), (ctypes.c_long * 4)(9 * Ci, 1, Ci, 0), 0, 0)
    jobs[1] = Job(bd.data_ptr(), bp.data_ptr(), (ctypes.c_long * 4)(3, 4, 1, 1), (ctypes.c_long * 4)(1, 3, 0, 0), 0, 1)
    L.call("vae_multi_pack", jobs, 2, L.current_stream())
    assert torch.equal(wp.float().cpu(), w.permute(0, 2, 3, 1).reshape(Co, 9, Ci).to(torch.bfloat16).float())
    assert torch.equal(bp.cpu(), b.view(3, 4).t().reshape(-1))


def test_multi_reduce_equals_the_single_folds(L):
    """several slab folds in one launch: bit-identical to the one-at-a-time calls (same kernel body, same order)"""
    g = torch.Generator().manual_seed(23)

    class RJob(ctypes.Structure):
        _fields_ = [("slab", ctypes.c_void_p), ("nsplit", ctypes.c_int), ("slab_stride", ctypes.c_long),
                    ("slab_base", ctypes.c_long), ("out", ctypes.c_void_p), ("dims", ctypes.c_long * 4),
                    ("strides", ctypes.c_long * 4), ("limits", ctypes.c_long * 4), ("base", ctypes.c_long),
                    ("accumulate", ctypes.c_int)]

    l4 = lambda *v: (ctypes.c_long * 4)(*v)
    shapes = [(7, 64, 9, 64), (3, 8, 9, 16), (16, 128, 4, 256), (2, 64, 16, 8)]
    slabs = [torch.randn(*s, generator=g).to(DEV) for s in shapes]
    specs = []       # (dims, strides, limits, out shape)
    for (S, Co, T, Ci) in shapes:
        if T == 16:      # padded first-conv layout [co][16][8] -> [co][3][9]: taps < 9, channels < 3
            specs.append((l4(Co, 16, 8, 1), l4(27, 1, 9, 0), l4(Co, 9, 3, 1), (Co, 3, 9)))
        else:            # [co][tap][ci] -> [co][ci][tap]
            specs.append((l4(Co, T, Ci, 1), l4(T * Ci, 1, T, 0), l4(Co, T, Ci, 1), (Co, Ci, T)))
    singles, multis = [], []
    jobs = (RJob * len(shapes))()
    for i, ((S, Co, T, Ci), sl, (dims, strides, limits, oshape)) in enumerate(zip(shapes, slabs, specs)):
        o1 = torch.zeros(oshape, device=DEV)
        L.call("vae_reduce_slabs_scatter", L.ptr(sl), S, Co * T * Ci, 0, L.ptr(o1), dims, strides, limits, 0, 0, L.current_stream())
        singles.append(o1)
        o2 = torch.zeros(oshape, device=DEV)
        multis.append(o2)
        jobs[i] = RJob(sl.data_ptr(), S, Co * T * Ci, 0, o2.data_ptr(), dims, strides, limits, 0, 0)
    L.call("vae_multi_reduce_slabs_scatter", jobs, len(shapes), L.current_stream())
    for (S, Co, T, Ci), sl, a, b in zip(shapes, slabs, singles, multis):
        assert torch.equal(a, b)
        ref = sl.double().sum(0)
        ref = ref[:, :9, :3].permute(0, 2, 1) if T == 16 else ref.permute(0, 2, 1)
        assert rel_l2(b.cpu(), ref.float().cpu()) <= 1e-6
    with pytest.raises(RuntimeError, match="out of range"):
        L.call("vae_multi_reduce_slabs_scatter", jobs, 0, L.current_stream())
    jobs[1].nsplit = 0
    with pytest.raises(RuntimeError, match="bad arguments"):
        L.call("vae_multi_reduce_slabs_scatter", jobs, 2, L.current_stream())


@pytest.mark.parametrize("S,Co,T,Ci", [(1, 8, 9, 16), (2, 256, 9, 256), (5, 64, 4, 128), (8, 64, 9, 64), (2, 1024, 9, 272)])
def test_fold_of_few_slabs_in_both_forms(L, S, Co, T, Ci):
    """slab folds of at most 8 slabs take a form of their own (one thread per 16-byte column, no barrier): same sums
    as the split-group form up to the order of the additions, same scatter, same accumulate; the last case is larger
    than one round of the capped grid"""
    g = torch.Generator().manual_seed(5)
    slab = torch.randn(S, Co, T, Ci, generator=g).to(DEV)
    ref = slab.double().sum(0).permute(0, 2, 1).float()          # [co][ci][tap]
    outs = []
    try:
        for form in (1, 0):
            L.call("vae_set_tuning", b"reduce_direct", form)
            out = torch.ones(Co, Ci, T, device=DEV)
            L.call("vae_reduce_slabs_scatter", L.ptr(slab), S, Co * T * Ci, 0, L.ptr(out), L.long4(Co, T, Ci, 1),
                   L.long4(T * Ci, 1, T, 0), None, 0, 0, L.current_stream())
            assert rel_l2(out.cpu(), ref.cpu()) <= 1e-6
            acc = torch.ones(Co, Ci, T, device=DEV)
            L.call("vae_reduce_slabs_scatter", L.ptr(slab), S, Co * T * Ci, 0, L.ptr(acc), L.long4(Co, T, Ci, 1),
                   L.long4(T * Ci, 1, T, 0), None, 0, 1, L.current_stream())
            assert rel_l2(acc.cpu(), (ref + 1.0).cpu()) <= 1e-6
            outs.append(out)
    finally:
        L.call("vae_set_tuning", b"reduce_direct", 1)
    assert rel_l2(outs[0].cpu(), outs[1].cpu()) <= 1e-6
    if S <= 2:
        assert torch.equal(outs[0], outs[1])        # one addition per element: no order to differ in


@pytest.mark.parametrize("n", [4096 * 997 + 3, 1000])
def test_adamw_clip_step_matches_oracle(L, n):
    g = torch.Generator().manual_seed(n)
    p, grad = torch.randn(n, generator=g), torch.randn(n, generator=g) * (3.0 / n ** 0.5)
    params, grads = {"w": p.clone()}, {"w": grad.clone()}
    m, v = {"w": torch.zeros(n)}, {"w": torch.zeros(n)}
    pd, gd = p.to(DEV), grad.to(DEV)
    md, vd = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    part = torch.empty(L.call("vae_adamw_grid", n), device=DEV)
    norm = torch.zeros(1, device=DEV)
    for step in (1, 2, 3):
        gn = O.clip_grad_norm(grads, 1.0)
        O.adamw_step(params, grads, m, v, step, lr=1e-3)
        L.call("vae_adamw_clip_step", L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(vd), n, L.ptr(part), 1.0, 1.0, 1e-3, 0.9, 0.999,
               1e-8, 0.01, step, L.ptr(norm), L.current_stream())
        assert abs(norm.item() - gn.item()) <= 1e-5 * gn.item()
        assert rel_l2(pd.cpu(), params["w"]) <= 1e-6
        assert rel_l2(md.cpu(), m["w"]) <= 1e-5 and rel_l2(vd.cpu(), v["w"]) <= 1e-5
        grads["w"] = grad.clone()      # the oracle clipped in place; the kernel leaves grads untouched


def test_adamw_clip_step_takes_the_loss_sums_along(L):
    """vae_adamw_clip_step_elbo = vae_elbo_combine (train.py:44-51) + vae_adamw_clip_step (train.py:336-338) with the
    loss sums finished by an extra workgroup of the first launch: the same three loss numbers and the same update, bit
    for bit."""
    g = torch.Generator().manual_seed(5)
    n = 300_001
    p, grad = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.01
    rp, kp = torch.rand(777, generator=g).to(DEV), torch.rand(33, generator=g).to(DEV)
    outs = []
    for fused in (False, True):
        pd, gd = p.to(DEV), grad.to(DEV)
        md, vd = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        part = torch.empty(L.call("vae_adamw_grid", n), device=DEV)
        norm, out3 = torch.zeros(1, device=DEV), torch.full((3,), float("nan"), device=DEV)
        for step in (1, 2):
            if fused:
                L.call("vae_adamw_clip_step_elbo", L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(vd), n, L.ptr(part), 1.0, 1.0, 1e-3,
                       0.9, 0.999, 1e-8, 0.01, step, L.ptr(norm), L.ptr(rp), 777, L.ptr(kp), 33, 0.5, 1.0 / 64, L.ptr(out3),
                       L.current_stream())
            else:
                L.call("vae_elbo_combine", L.ptr(rp), 777, L.ptr(kp), 33, 0.5, 1.0 / 64, L.ptr(out3), L.current_stream())
                L.call("vae_adamw_clip_step", L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(vd), n, L.ptr(part), 1.0, 1.0, 1e-3, 0.9,
                       0.999, 1e-8, 0.01, step, L.ptr(norm), L.current_stream())
        outs.append((pd.cpu(), md.cpu(), vd.cpu(), norm.cpu(), out3.cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    ref = (rp.double().sum() + 0.5 * kp.double().sum()) / 64
    assert abs(outs[1][4][0].item() - ref.item()) <= 1e-6 * ref.item()
    # without out3 the partial-sum pointers are not looked at
    L.call("vae_adamw_clip_step_elbo", L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(vd), n, L.ptr(part), 1.0, 1.0, 1e-3, 0.9, 0.999,
           1e-8, 0.01, 3, L.ptr(norm), None, 0, None, 0, 0.0, 0.0, None, L.current_stream())
    with pytest.raises(RuntimeError, match="null partial sums"):
        L.call("vae_adamw_clip_step_elbo", L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(vd), n, L.ptr(part), 1.0, 1.0, 1e-3, 0.9,
               0.999, 1e-8, 0.01, 4, L.ptr(norm), None, 5, None, 0, 0.0, 0.0, L.ptr(out3), L.current_stream())


@pytest.mark.parametrize("L_,C,hw", [(128, 256, 16), (64, 256, 64), (72, 64, 36), (128, 256, 4)])
def test_multi_pack_transposing_jobs(L, L_, C, hw):
    """the nn.Linear weight jobs of the model's pack (vae/models/vae.py:94 flatten order -> NHWC order, and the
    transposed copies of the input-gradient GEMMs): jobs whose source and destination run along different dims go
    through LDS tiles when both dims are long enough; sizes here include tile edges that are not full (72, 36) and a
    source-fastest dim too short for the tiled form (4: the gathering form).  bf16 of the permuted source, exactly."""
    g = torch.Generator().manual_seed(L_ + C + hw)
    F_ = C * hw
    w_mu = torch.randn(L_, F_, generator=g)          # fc_mu.weight [L][(c, p)]
    w_up = torch.randn(F_, L_, generator=g)          # fc_up.weight [(c, p)][L]
    dmu, dup = w_mu.to(DEV), w_up.to(DEV)
    fc_f = torch.zeros(L_, hw, C, device=DEV, dtype=torch.bfloat16)       # [l][p][c]
    fc_d = torch.zeros(hw, C, 2 * L_, device=DEV, dtype=torch.bfloat16)   # [(p, c)][2L], this job fills columns < L
    up_d = torch.zeros(L_, hw, C, device=DEV, dtype=torch.bfloat16)       # [l][(p, c)]

    class Job(ctypes.Structure):
        _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dims", ctypes.c_long * 4),
                    ("strides", ctypes.c_long * 4), ("base", ctypes.c_long), ("f32", ctypes.c_int)]
    l4 = lambda *v: (ctypes.c_long * 4)(*v)
    jobs = (Job * 3)()
    jobs[0] = Job(dmu.data_ptr(), fc_f.data_ptr(), l4(L_, C, hw, 1), l4(F_, 1, C, 0), 0, 0)
    jobs[1] = Job(dmu.data_ptr(), fc_d.data_ptr(), l4(L_, C, hw, 1), l4(1, 2 * L_, C * 2 * L_, 0), 0, 0)
    jobs[2] = Job(dup.data_ptr(), up_d.data_ptr(), l4(C, hw, L_, 1), l4(1, C, F_, 0), 0, 0)
    L.call("vae_multi_pack", jobs, 3, L.current_stream())
    bf = lambda t: t.to(torch.bfloat16).float()
    w3 = w_mu.view(L_, C, hw)
    assert torch.equal(fc_f.float().cpu(), bf(w3.permute(0, 2, 1)))
    assert torch.equal(fc_d.float().cpu()[:, :, :L_], bf(w3.permute(2, 1, 0)))
    assert float(fc_d.float().abs().cpu()[:, :, L_:].max()) == 0.0
    assert torch.equal(up_d.float().cpu(), bf(w_up.view(C, hw, L_).permute(2, 1, 0)))


@pytest.mark.parametrize("Co,Ci", [(64, 64), (256, 128), (72, 24), (128, 8)])
def test_multi_pack_first_dim_to_fastest_jobs(L, Co, Ci):
    """the conv weights packed for the input-gradient GEMMs ([co][ci][tap] -> [ci][8 - tap][co], the flipped kernel of
    vae/models/vae.py:20's Conv2d backward) and the ConvTranspose weights ([ci][co][tap] -> [tap][co][ci],
    vae/models/vae.py:131): the source's first dim becomes the destination's fastest one and the source's fastest dim
    (9 or 4 taps) is too short for a tile of its own, so the inner dims are tiled as one run.  Sizes include ragged
    tile edges (72 rows, runs of 216 and 96).  bf16 of the permuted source, exactly."""
    g = torch.Generator().manual_seed(Co * 7 + Ci)
    w = torch.randn(Co, Ci, 3, 3, generator=g)
    wt = torch.randn(Co, Ci, 2, 2, generator=g)        # a ConvTranspose weight [in = Co][out = Ci][2][2]
    wd, wtd = w.to(DEV), wt.to(DEV)
    conv_d = torch.zeros(Ci, 9, Co, device=DEV, dtype=torch.bfloat16)
    ct_f = torch.zeros(4, Ci, Co, device=DEV, dtype=torch.bfloat16)

    class Job(ctypes.Structure):
        _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dims", ctypes.c_long * 4),
                    ("strides", ctypes.c_long * 4), ("base", ctypes.c_long), ("f32", ctypes.c_int)]
    l4 = lambda *v: (ctypes.c_long * 4)(*v)
    jobs = (Job * 2)()
    jobs[0] = Job(wd.data_ptr(), conv_d.data_ptr(), l4(Co, Ci, 9, 1), l4(1, 9 * Co, -Co, 0), 8 * Co, 0)
    jobs[1] = Job(wtd.data_ptr(), ct_f.data_ptr(), l4(Co, Ci, 4, 1), l4(1, Co, Ci * Co, 0), 0, 0)
    L.call("vae_multi_pack", jobs, 2, L.current_stream())
    bf = lambda t: t.to(torch.bfloat16).float()
    assert torch.equal(conv_d.float().cpu(), bf(w.view(Co, Ci, 9).flip(2).permute(1, 2, 0)))
    assert torch.equal(ct_f.float().cpu(), bf(wt.view(Co, Ci, 4).permute(2, 1, 0)))
