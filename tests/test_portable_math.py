"""bp_osd_amd/csrc/portable_math.h (the tanh / log the product-sum kernels evaluate) against the platform libm (CPU)."""
import numpy as np

from oracle import portable_math


def _ulps(a, b):
    ia = np.ascontiguousarray(a).view(np.int64).copy()
    ib = np.ascontiguousarray(b).view(np.int64).copy()
    ia[ia < 0] = np.iinfo(np.int64).min - ia[ia < 0]
    ib[ib < 0] = np.iinfo(np.int64).min - ib[ib < 0]
    return np.abs(ia - ib)


def test_accuracy_against_libm():
    rng = np.random.default_rng(0)
    n = 2_000_000
    x = (rng.random(n) * 2 - 1) * 10.0 ** (rng.random(n) * 6 - 4)        # |x| from 1e-4 to 1e2, both signs
    assert _ulps(portable_math("tanh", x), np.tanh(x)).max() <= 4
    y = 10.0 ** (rng.random(n) * 40 - 20)
    assert _ulps(portable_math("log", y), np.log(y)).max() <= 2
    z = (rng.random(n) * 2 - 1) * 50
    assert _ulps(portable_math("expm1", z), np.expm1(z)).max() <= 2
    # the composite of the check update: log((1 + x) / (1 - x)) with x a product of two tanh values
    a, b = portable_math("tanh", (rng.random(n) * 60 - 30) / 2), portable_math("tanh", (rng.random(n) * 60 - 30) / 2)
    r = (1 + a * b) / (1 - a * b)
    with np.errstate(divide="ignore"):
        ref = np.log(r)
    got = portable_math("log", r)
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all() and _ulps(got[fin], ref[fin]).max() <= 2
    assert (got[~fin] == ref[~fin]).all()


def test_special_values():
    with np.errstate(all="ignore"):
        x = np.array([0.0, -0.0, 1e-320, 4.9e-324, 1.0, -1.0, 0.5, 2.0, 21.999, 22.0, 1e300, np.inf, -np.inf, np.nan,
                      1 - 1e-16, 2.2250738585072014e-308])
        for name, f in (("tanh", np.tanh), ("log", np.log)):
            got, ref = portable_math(name, x), f(x)
            assert (np.isnan(got) == np.isnan(ref)).all(), name
            ok = ~np.isnan(ref)
            assert (np.signbit(got[ok]) == np.signbit(ref[ok])).all(), name
            fin = ok & np.isfinite(ref)
            assert (np.isfinite(got[ok]) == np.isfinite(ref[ok])).all(), name
            assert _ulps(got[fin], ref[fin]).max() <= 4, name
    # saturation points of the product-sum update: tanh rounds to exactly 1 from |x| = 19.07 on
    assert portable_math("tanh", np.array([19.1, -30.0, 400.0])).tolist() == [1.0, -1.0, 1.0]
    assert portable_math("tanh", np.array([18.0]))[0] < 1.0


def test_two_division_form_of_the_check_update():
    """Round 4: tanh(x/2) with one division (exp from a polynomial) and log(A/B) with one division (the quotient folded into
    the logarithm's reduction) -- what the product-sum kernels and the oracle's ps_math = 1 evaluate.  Against long-double
    references and glibc: pm_tanh_half within 5 ulp of tanh(x/2) and as often bit-identical to glibc's as pm_tanh is;
    pm_log_quot no less accurate than log(fl(A/B)); special values and saturation."""
    from oracle import portable_tanh_half, portable_log_quot

    L = np.longdouble
    rng = np.random.default_rng(1)
    n = 1_000_000
    x = (rng.random(n) * 2 - 1) * 10.0 ** (rng.random(n) * 7 - 5)
    got = portable_tanh_half(x)
    ref = np.tanh(x.astype(L) / 2)
    assert float((np.abs((got.astype(L) - ref) / ref)).max()) <= 5 * 2.0 ** -53
    same_new = (got == np.tanh(x / 2)).mean()
    same_old = (portable_math("tanh", x / 2) == np.tanh(x / 2)).mean()
    assert same_new >= same_old - 0.03, (same_new, same_old)
    X = np.tanh((rng.random(n) * 60 - 30) / 2) * np.tanh((rng.random(n) * 60 - 30) / 2)
    A, B = 1 + X, 1 - X
    with np.errstate(all="ignore"):
        libm = np.log(A / B)
    got = portable_log_quot(A, B)
    fin = np.isfinite(libm)
    assert (np.isfinite(got) == fin).all() and (got[~fin] == libm[~fin]).all()
    exact = np.log(A[fin].astype(L) / B[fin].astype(L))
    nz = exact != 0
    err = lambda v: np.abs((v[fin][nz].astype(L) - exact[nz]) / exact[nz])
    assert float(err(got).max()) <= float(err(libm).max()) and float(np.median(err(got))) <= 2.0 ** -53
    # special values: signed zero, tiny arguments, saturation (tanh(x/2) rounds to 1 from |x| = 38.2 on), infinities, NaN
    t = portable_tanh_half(np.array([0.0, -0.0, 1e-300, -1e-20, 38.3, -50.0, np.inf, -np.inf, 36.0]))
    assert t[:8].tolist() == [0.0, -0.0, 5e-301, -5e-21, 1.0, -1.0, 1.0, -1.0] and np.signbit(t[:2]).tolist() == [False, True] and t[8] < 1.0
    assert np.isnan(portable_tanh_half(np.array([np.nan]))[0])
    with np.errstate(all="ignore"):
        r = portable_log_quot(np.array([2.0, 0.0, 1.0, 1.5, np.nan, 0.0]), np.array([0.0, 2.0, 1.0, 0.5, 1.0, 0.0]))
    assert r[0] == np.inf and r[1] == -np.inf and r[2] == 0.0 and abs(r[3] - np.log(3.0)) < 1e-15 and np.isnan(r[4]) and np.isnan(r[5])
