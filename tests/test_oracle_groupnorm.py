"""The oracle's GroupNorm double-conv block against fixtures made by running the reference's own draft class
(vae/practice/vae_conv.py:20-36) -- forward, pooled output and every gradient (oracle/make_fixtures.py --gn)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle as O
from golden_util import GOLDEN, rel_l2


@pytest.mark.parametrize("name", ["gn_block", "gn_block_dropout"])
def test_double_conv_gn_matches_the_reference_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = float(z["cfg/p"])
    sd = {"blk." + k[3:]: torch.from_numpy(z[k]).requires_grad_(True) for k in z.files if k.startswith("sd/")}
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    shp = tuple(int(v) for v in z["mask/shape"])
    mask = torch.from_numpy(np.unpackbits(z["mask/bits"])[: int(np.prod(shp))].reshape(shp).copy())
    y = O.double_conv_gn(x, sd, "blk", True, p, mask)
    pooled = F.max_pool2d(y, 2)
    assert rel_l2(y.detach(), z["y"]) <= 1e-5 and rel_l2(pooled.detach(), z["pooled"]) <= 1e-5
    (pooled * torch.from_numpy(z["r"])).sum().backward()
    assert rel_l2(x.grad, z["grad/x"]) <= 1e-4
    for k, t in sd.items():
        assert rel_l2(t.grad, z["grad/" + k[4:]]) <= 1e-4, k


def test_group_norm_is_the_published_formula():
    """the restatement against torch's own kernel, ragged group sizes included"""
    g = torch.Generator().manual_seed(0)
    for n, c, h, w, groups in [(2, 16, 5, 3, 8), (1, 64, 4, 4, 8), (3, 24, 2, 7, 4)]:
        x = torch.randn(n, c, h, w, generator=g) * 2 + 1
        gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
        assert rel_l2(O.group_norm(x, gamma, beta, groups), F.group_norm(x, groups, gamma, beta, 1e-5)) <= 1e-6
