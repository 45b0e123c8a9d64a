"""CPU check of the arithmetic identity behind the bf16x6 kernels (csrc/common.h split_pair_bf16x3, used by csrc/conv_bww_x6.hip and
csrc/conv_x6.hip): an fp32 value is the EXACT sum of three bf16 pieces made by round-to-nearest-even, and the six products the kernels
accumulate differ from the exact product by at most 2^-23 |a b| (the dropped terms m*l, l*m, l*l).  numpy restatement of the device
code's steps (v_cvt_pk_bf16_f32 = RNE to 8 significand bits); the GPU tests (tests/test_gpu_bww_x6.py, tests/test_gpu_fwd_x6.py) check
the kernels themselves."""
import numpy as np


def bf16_rne(x):
    """float32 -> the float32 value of its bfloat16 rounding (round to nearest even), as v_cvt_pk_bf16_f32 does for finite inputs"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return r.astype(np.uint32).view(np.float32)


def split3(a):
    a = np.asarray(a, np.float32)
    h = bf16_rne(a)
    r = (a - h).astype(np.float32)                      # exact
    m = bf16_rne(r)
    l = bf16_rne((r - m).astype(np.float32))            # r - m is exact and has at most 8 significant bits
    return h, m, l


def test_three_bf16_pieces_sum_exactly_to_the_fp32_value():
    rng = np.random.default_rng(5)
    # normal numbers whose pieces stay normal (|a| >= 2^-100): activations, weights and gradients of the nets live there
    a = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.exp(rng.uniform(-60, 60, 200000)).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 1.0000001, 0.99999994, 255.5, 3.4e38 / 4, -7.3e-30], np.float32)])
    h, m, l = split3(a)
    for p in (h, m, l):
        assert np.all((p.view(np.uint32) & np.uint32(0xFFFF)) == 0), "a piece is not a bf16 value"
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), a.astype(np.float64))
    nz = a != 0
    assert np.all(np.abs(m[nz]) <= np.abs(a[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(a[nz]) * 2.0 ** -16)


def test_six_products_match_the_exact_product_to_fp32_rounding():
    rng = np.random.default_rng(6)
    a = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    b = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    ah, am, al = [p.astype(np.float64) for p in split3(a)]
    bh, bm, bl = [p.astype(np.float64) for p in split3(b)]
    six = ah * bh + ah * bm + am * bh + am * bm + ah * bl + al * bh
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.maximum(np.abs(exact), 1e-300)
    assert rel.max() <= 2.0 ** -23, rel.max()           # fp32 rounding of the product itself: 2^-24
