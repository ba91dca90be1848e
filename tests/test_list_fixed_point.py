"""CPU test of the Verlet-list candidate test (csrc/nm_kernels.h, Replica::rebuild / test16): the arithmetic on 16-bit
fixed-point coordinates restated in numpy must accept every pair whose exact minimum-image separation is below rc + skin (the list
is a superset of the exact one; the pair loop's exact cutoff test discards the surplus), for boxes from the minimum-image limit
to the largest BASELINE cell, positions inside and outside the box, and pairs placed right at the list radius."""
import numpy as np
import pytest


def fixed(x, L):
    """u = round(65536 frac(x / L)) mod 2^16 per coordinate, as the kernel's __double2int_rn(fract(x / L) * 65536) & 0xFFFF"""
    s = x * (1.0 / L)
    return (np.rint((s - np.floor(s)) * 65536.0).astype(np.int64) & 0xFFFF).astype(np.uint16)


def accepted(ui, uj, L, rlist):
    """test16: differences wrap modulo 2^16 (v_pk_sub_i16), squares summed in 32 bits (v_dot2_i32_i16), unsigned compare"""
    d = (ui.astype(np.int64) - uj.astype(np.int64) + 32768) % 65536 - 32768      # int16 wrap
    r2 = ((d * d).sum(-1)) & 0xFFFFFFFF
    rt = rlist * (65536.0 / L) + 1.8
    t2 = np.uint64(np.floor(rt * rt)) + np.uint64(1)
    return r2.astype(np.uint64) < t2


@pytest.mark.parametrize('L,rlist', [(5.0, 2.9), (6.03, 2.9), (6.2, 2.9), (9.2, 2.95), (12.2, 3.1), (13.4, 3.1), (16.2, 8.3)])
def test_fixed_point_test_accepts_every_pair_inside_the_list_radius(L, rlist):
    rng = np.random.default_rng(int(L * 100))
    n = 600
    x = rng.uniform(-0.7 * L, 1.7 * L, (n, 3))                    # unwrapped positions, as in the middle of a trajectory
    # plant pairs right at the list radius (inside by 1e-9 .. 1e-4), through every periodic image
    k = 200
    dirs = rng.normal(size=(k, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    eps = 10.0 ** rng.uniform(-9, -4, k)
    x[:k] = x[k:2 * k] + dirs * (rlist - eps)[:, None] + L * rng.integers(-1, 2, (k, 3))
    u = fixed(x, L)
    d = x[:, None, :] - x[None, :, :]
    d -= L * np.rint(d / L)
    r = np.sqrt((d * d).sum(-1))
    inside = r < rlist
    acc = accepted(u[:, None, :], u[None, :, :], L, rlist)
    assert (acc | ~inside).all()                                   # superset
    # and it is not much of a superset: whatever is accepted lies within the radius plus a few fixed-point units
    assert (r[acc] < rlist + 4.0 * L / 65536.0).all()
    assert inside.sum() > 2 * k
