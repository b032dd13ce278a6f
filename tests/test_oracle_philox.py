"""Philox4x32-10 restatement against the Random123 known-answer vectors, and the
distributional sanity of the integer / normal transforms built on it."""
import numpy as np

from oracle import vae_oracle as O

KAT = [  # (counter, key, expected) from Random123 kat_vectors, philox4x32 10 rounds
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_kat():
    for ctr, key, exp in KAT:
        out = O.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))
        assert tuple(int(v) for v in out[0]) == exp


def test_stream_layout():
    a = O.philox_bits(10, seed=1337, stream=4, step=0)
    b = O.philox_bits(7, seed=1337, stream=4, step=0)
    assert (a[:7] == b).all()
    assert (O.philox_bits(8, 1337, 4, 1) != a[:8]).any()
    assert (O.philox_bits(8, 1337, 3, 0) != a[:8]).any()
    assert (O.philox_bits(8, 1337, 4, 0, rank=1) != a[:8]).any()


def test_keep_mask_rate():
    m = O.philox_keep_mask(1 << 20, 0.1, 1337, 0, 0)
    assert abs(m.mean() - 0.9) < 2e-3
    assert O.dropout_threshold(0.1) == 6554


def test_normal_moments_and_accuracy():
    z = O.philox_normal(1 << 20, 1337, O.STREAM_EPS, 0).astype(np.float64)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs((z ** 4).mean() - 3) < 0.05
    # the fp32-only log / sincos agree with libm to a few ulp
    bits = O.philox_bits(1 << 16, 7, 0, 0).reshape(-1, 2)
    z0, z1 = O.normal_pair_from_bits(bits[:, 0], bits[:, 1])
    u1 = ((bits[:, 0] >> 8).astype(np.float64) + 1) * 2.0 ** -24
    u2 = (bits[:, 1] >> 8).astype(np.float64) * 2.0 ** -24
    r = np.sqrt(-2 * np.log(u1))
    assert np.abs(z0 - r * np.cos(2 * np.pi * u2)).max() < 5e-6
    assert np.abs(z1 - r * np.sin(2 * np.pi * u2)).max() < 5e-6
