"""The dense Fresnel / Snell pin (round 6): the oracle's fresnel() against 11,000 outputs of the reference's OWN python --
scripts/reflections/fresnel.py (fresnel_reflect_dir, fresnel_refract_dir, the render() closure's Reff / Teff) and
scripts/reflections/snell_multi.py (snell_refract_dir) -- imported / run from /root/reference by
tests/golden/gen_pyref_dense.py; the committed .npy files hold the scripts' outputs only, the inputs come from the seeded
generator tests/golden/pyref_cases.py (v1 != 0.3, v2 > v1, the angle_limit branch of radar_algorithms.h:82-88, the two eps
branches of :112-118).

What the comparison can and cannot say.  The scripts work in f64; the C++ (and the oracle, and the kernels) measures both
angles with `acosf` of an f32 dot product (the float overload) -- resolution 6e-8 / sin(angle) -- and uses
eps = 1e-4 where the script has 1e-5.  So:
  * directions agree to 2e-7 + 6e-8 n12^2 / cos(t) (the refraction's square root amplifies the f32 cosine near the limit angle);
  * the transmitted / totally-reflected DECISION agrees everywhere except within 1e-6 rad of asin(n2 / n1);
  * energies agree to 5e-6 + 6e-6 / min(i, t) + 2e-7 / cos^2(t) (transmitted) resp. 5e-6 + 3e-7 / (pi/2 - i) + 2e-7 / i (not
    transmitted: the f32 value of pi/2 puts R up to 5e-4 ABOVE 1 near grazing) from 5e-3 rad on -- 85 % of the random cases
    within 1e-6, 97 % within 1e-5;
  * below 5e-3 rad the C++'s own f32 quantisation decides (acosf(1 - 6e-8) = 3.45e-4 > eps): where both dot products round to
    exactly 1 the `< eps` branch gives ((n1 - n2) / (n1 + n2))^2, which the script confirms to 5e-6; where the refraction's
    x-component rounds to 1 - 6e-8 the C++ formula yields R = 1 (total reflection at normal incidence) and where it rounds
    above 1, NaN -- the reference's behaviour, restated by the oracle, which no f64 script can pin: asserted as such;
  * in the band pi - 1e-4 < i + t <= pi - 1e-5 (v1 == v2 at grazing incidence) script and C++ take different branches BY
    THEIR OWN eps: the oracle must give exactly 1 there (radar_algorithms.h:110,116).
"""
import math
import os
import sys

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import pyref_cases  # noqa: E402

N = (-1.0, 0.0, 0.0)


def _load(fam, what):
    return np.load(os.path.join(GOLDEN, "pyref_dense_%s_%s.npy" % (fam, what)))


def _oracle_all(oracle, th, v1, v2):
    D = pyref_cases.direction(th)
    out = [oracle.fresnel(N, D[i], 1.0, 0.5, float(v1[i]), float(v2[i])) for i in range(len(th))]
    rd = np.array([o[0] for o in out], np.float64); td = np.array([o[2] for o in out], np.float64)
    return D, rd, np.array([o[1] for o in out]), td, np.array([o[3] for o in out])


def _snell_t(ti, v1, v2):
    return math.asin(min(1.0, math.sin(ti) * v2 / v1))        # n1 := v2, n2 := v1: sin t = sin i * n1 / n2


@pytest.mark.parametrize("fam", pyref_cases.FAMILIES)
def test_directions_and_branch_decisions(oracle, fam):
    th, v1, v2 = pyref_cases.cases(fam)
    D, rd, re, td, te = _oracle_all(oracle, th, v1, v2)
    refl, refr, snell = _load(fam, "refl"), _load(fam, "refr"), _load(fam, "snell")
    assert refl.shape == (len(th), 2) and np.abs(rd[:, :2] - refl).max() < 1e-7 and not rd[:, 2].any()
    py_t, or_t = np.any(refr != 0, axis=1), np.any(td != 0, axis=1)
    n_near = 0
    for i in np.nonzero(py_t != or_t)[0]:                     # a flipped decision: only on the limit angle itself
        assert v2[i] > v1[i] and abs(th[i] - math.asin(v1[i] / v2[i])) < 1e-6, (i, th[i], v1[i], v2[i])
        n_near += 1
    both = py_t & or_t
    worst = 0.0
    for i in np.nonzero(both)[0]:
        n12 = v2[i] / v1[i]
        ct = max(math.cos(_snell_t(th[i], v1[i], v2[i])), 1e-4)
        tol = 2e-7 + 6e-8 * n12 * n12 / ct
        e = np.abs(td[i, :2] - refr[i]).max()
        worst = max(worst, e / tol)
        assert e < tol, (i, th[i], v1[i], v2[i], td[i], refr[i])
        if np.isfinite(snell[i]).all():                       # the second script's refraction: the same direction
            assert np.abs(td[i, :2] - snell[i]).max() < tol
    if fam == "random":
        assert both.sum() > 5000 and (~py_t).sum() > 3000 and (v1 != np.float32(0.3)).sum() > 8000 and (v2 > v1).sum() > 3000
    if fam == "limit":
        below = th < np.arcsin(v1 / v2)
        assert 200 < below.sum() < 300 and np.array_equal(py_t, below) and n_near <= 2
        assert (np.abs(th - np.arcsin(v1 / v2)) <= 1e-3 + 1e-7).all()


def test_energies_random_and_limit(oracle):
    for fam in ("random", "limit"):
        th, v1, v2 = pyref_cases.cases(fam)
        D, rd, re, td, te = _oracle_all(oracle, th, v1, v2)
        en, refr = _load(fam, "energy"), _load(fam, "refr")
        errs, n_quirk = [], 0
        for i in range(len(th)):
            if not np.isfinite(en[i, 0]):
                continue
            tr = bool(np.any(refr[i] != 0))
            if tr != bool(np.any(td[i] != 0)):
                continue
            tt = _snell_t(th[i], v1[i], v2[i]) if tr else math.pi / 2
            if min(th[i], tt) < 5e-3 or not np.isfinite(re[i]):
                n_quirk += 1
                continue
            tol = (5e-6 + 6e-6 / min(th[i], tt) + 2e-7 / max(math.cos(tt), 1e-4) ** 2) if tr else (5e-6 + 3e-7 / (math.pi / 2 - th[i]) + 2e-7 / th[i])
            e = max(abs(re[i] - en[i, 0]), abs(te[i] - en[i, 1]))
            assert e < tol, (fam, i, th[i], v1[i], v2[i], re[i], en[i], tol)
            assert abs(re[i] + te[i] - 1.0) < 1e-12
            errs.append(e)
        errs = np.array(errs)
        if fam == "random":
            assert len(errs) > 9500 and n_quirk < 150
            assert (errs < 1e-6).mean() > 0.80 and (errs < 1e-5).mean() > 0.95 and np.median(errs) < 5e-7
        else:
            assert len(errs) > 480 and (errs < 1e-5).mean() > 0.80


def test_eps_branches(oracle):
    """radar_algorithms.h:108-122, eps = 1e-4."""
    # --- incidence + refraction < eps -------------------------------------------------------------------------------
    th, v1, v2 = pyref_cases.cases("eps_small")
    D, rd, re, td, te = _oracle_all(oracle, th, v1, v2)
    en = _load("eps_small", "energy")
    assert (D[:, 0] == 1.0).all()                             # cos(theta) rounds to 1.0f: the C++'s incidence angle is exactly 0
    n_branch = n_one = n_nan = 0
    for i in range(len(th)):
        lim = ((v2[i] - v1[i]) / (v2[i] + v1[i])) ** 2        # rs = rp = (n1 - n2) / (n1 + n2)
        tx = np.float32(td[i, 0])
        if tx == np.float32(1.0):                             # both angles exactly 0: the branch
            assert re[i] == lim or abs(re[i] - lim) < 1e-15, (i, re[i], lim)
            assert abs(en[i, 0] - lim) < 5e-6                 # ... which the script's f64 evaluation confirms
            n_branch += 1
        elif tx < np.float32(1.0):                            # acosf(1 - k 6e-8) >= 3.45e-4 > eps: -sin(-t) / sin(t) = 1 -- the C++'s own f32 quirk
            assert re[i] == 1.0, (i, tx, re[i])
            n_one += 1
        else:                                                 # acosf(> 1) = NaN -> NaN energies: the wave dies (RadarCPU.cpp:288,367 compare false)
            assert np.isnan(re[i]) and np.isnan(te[i])
            n_nan += 1
    assert n_branch > 180 and n_branch + n_one + n_nan == len(th), (n_branch, n_one, n_nan)
    # --- incidence + refraction > pi - eps --------------------------------------------------------------------------
    th, v1, v2 = pyref_cases.cases("eps_grazing")
    D, rd, re, td, te = _oracle_all(oracle, th, v1, v2)
    en, refr = _load("eps_grazing", "energy"), _load("eps_grazing", "refr")
    n_band = n_cmp = 0
    for i in range(len(th)):
        tr = bool(np.any(refr[i] != 0))
        assert tr == bool(np.any(td[i] != 0))
        tt = _snell_t(th[i], v1[i], v2[i]) if tr else math.pi / 2
        gap = math.pi - (th[i] + tt)                          # f64; the C++ adds two f32 angles: decided within ~3e-7 of the edge
        if gap < 1e-4 - 1e-6:
            assert re[i] == 1.0 and te[i] == 0.0, (i, gap, re[i])             # the branch, exactly
            if gap > 1e-5 + 1e-6 and tr:
                n_band += 1                                   # the script (eps = 1e-5) is still on the general formula here: not a comparand
            else:
                n_cmp += int(abs(en[i, 0] - 1.0) < 1e-3)
        elif gap > 1e-4 + 1e-6:
            tol = (5e-6 + 2e-7 / max(math.cos(tt), 1e-4) ** 2) if tr else (5e-6 + 3e-7 / (math.pi / 2 - th[i]))
            assert abs(re[i] - en[i, 0]) < tol, (i, gap, re[i], en[i, 0], tol)
            n_cmp += 1
    assert n_band >= 10 and n_cmp > 100, (n_band, n_cmp)
