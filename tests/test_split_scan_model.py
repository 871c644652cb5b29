"""The arithmetic of the split scan (csrc/saf_query.hip, query_split_kernel) restated in numpy -- no GPU: every fp32 operand cut into
two fp16 pieces under a power-of-two scale (labels: per label; feature rows: following the row's running maximum, group of 64
features by group, the accumulator rescaled when the scale drops), the dot product as hi.hi + hi.lo + lo.hi accumulated in fp32.
Checks the bound the kernel's header and INTEGRATION.md state -- a score within 3 x 2^-22 of sum |a b| of the exact dot product, for
rows and labels of any magnitude -- on the rows tests/test_split_scan.py feeds the device, and that a FIXED scale would not do."""
import numpy as np
import pytest

CUT = 3.0 * 2.0 ** -22
ACC = 32 * 2.0 ** -24  # one fp32 rounding of the running sum per k-step of 16 (the model accumulates a k-step exactly)


def _exponent(mx):
    """the e with mx * 2^e in [2^13, 2^14) (split_exponent)"""
    return 14 - (int(np.floor(np.log2(mx))) + 1)


def _cut(y):
    hi = y.astype(np.float16)
    lo = (y - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def split_dot(a, b, running=True):
    """a [D] fp32 row, b [D] fp32 label -> the split scan's fp32 score (float64 arithmetic stands in for the exact fp16 products)"""
    d = a.shape[0]
    bmax = float(np.abs(b).max())
    be = _exponent(bmax) if 0 < bmax < np.inf else 0
    bh, bl = _cut(np.ldexp(b, be).astype(np.float32))
    acc, re, seen = np.float32(0.0), 0, False
    for g0 in range(0, d, 64):
        grp = a[g0:g0 + 64]
        gm = float(np.abs(grp).max())
        if running and (np.ldexp(gm, re) >= 32768.0 or (not seen and gm > 0)):
            ne = _exponent(gm) if gm < np.inf else 0
            if seen:
                acc = np.float32(np.ldexp(np.float64(acc), ne - re))
            re, seen = ne, seen or gm > 0
        with np.errstate(over="ignore"):
            ah, al = _cut(np.ldexp(grp, re).astype(np.float32))
        for k0 in range(0, grp.shape[0], 16):
            s = slice(k0, k0 + 16)
            t = slice(g0 + k0, g0 + k0 + 16)
            acc = np.float32(np.float64(acc) + (al[s] * bh[t]).sum() + (ah[s] * bl[t]).sum() + (ah[s] * bh[t]).sum())
    return float(np.ldexp(np.float64(acc), -(re + be)))


def _rows(rng, d=512):
    base = rng.standard_normal((12, d)).astype(np.float32)
    base[0] *= 1e30
    base[1] *= 1e-30
    base[2] *= np.float32(1e-40)
    base[3, 64:] *= 1e5
    base[4] *= np.logspace(-6, 9, d).astype(np.float32)
    base[5, :64] = 0.0
    base[6] = 0.0
    base[6, -1] = 3.0
    base[7] *= np.logspace(9, -6, d).astype(np.float32)
    base[8, 100] = 6.0e4
    base[9] = np.abs(base[9]) * 1e-3 + 1.0
    base[10, 448:] *= 1e12
    base[11] *= 65504.0
    return base


def test_split_scan_arithmetic_holds_its_bound():
    rng = np.random.default_rng(7)
    rows = _rows(rng)
    labels = rng.standard_normal((6, 512)).astype(np.float32)
    labels[0] *= 1e20
    labels[1] *= 1e-20
    labels[2, 5:] = 0.0
    labels[3] *= np.float32(1e-40)
    worst = 0.0
    for a in rows:
        for b in labels:
            exact = float(a.astype(np.float64) @ b.astype(np.float64))
            mag = float(np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64)))
            if not (1e-30 < mag < 1e37):
                continue  # (the score itself leaves fp32's range, or is all denormal)
            err = abs(split_dot(a, b) - exact)
            assert err <= (CUT + ACC) * mag, (err / mag, a[:4], b[:4])
            worst = max(worst, err / mag)
    assert worst > 0.0  # (the model is not the exact product)


def test_a_fixed_scale_would_not_do():
    """the same cut WITHOUT the running row scale (scale 1 for every row): rows far from 1 lose their pieces to fp16's range"""
    rng = np.random.default_rng(8)
    b = rng.standard_normal(512).astype(np.float32)
    small = (rng.standard_normal(512) * 1e-7).astype(np.float32)   # below fp16's normal range: the pieces are denormal or zero
    exact = float(small.astype(np.float64) @ b.astype(np.float64))
    mag = float(np.abs(small.astype(np.float64)) @ np.abs(b.astype(np.float64)))
    assert abs(split_dot(small, b, running=False) - exact) > 100 * CUT * mag
    assert abs(split_dot(small, b) - exact) <= (CUT + ACC) * mag
    big = (rng.standard_normal(512) * 1e6).astype(np.float32)       # above 65504: the pieces overflow
    with np.errstate(over="ignore", invalid="ignore"):
        assert not np.isfinite(split_dot(big, b, running=False))
    exact = float(big.astype(np.float64) @ b.astype(np.float64))
    mag = float(np.abs(big.astype(np.float64)) @ np.abs(b.astype(np.float64)))
    assert abs(split_dot(big, b) - exact) <= (CUT + ACC) * mag
