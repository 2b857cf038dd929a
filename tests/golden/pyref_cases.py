"""The (incidence angle, v1, v2) cases of the dense Fresnel / Snell pin (round 6) -- a pure-numpy, seeded generator shared by
tests/golden/gen_pyref_dense.py (which feeds them to the reference's own python scripts, imported from /root/reference, and
stores ONLY the scripts' outputs as .npy) and by the tests (which rebuild the same inputs and feed them to the oracle and to the
GPU's fresnel_split).  numpy's legacy RandomState streams are stable across versions by policy.

Families (radar_algorithms.h line numbers; n1 := v2, n2 := v1, :62-63):
  random      10,000 draws: theta in [0, 90) deg, v1 in [0.05, 0.45] (so v1 != 0.3 is the rule), v2 in {0} (5 %), [0.001, 0.6]:
              both v2 < v1 (into the slower medium) and v2 > v1 (into the faster one: total reflection beyond asin(v1 / v2))
  limit         500 draws within +-1e-3 rad of the angle of total reflection asin(n2 / n1) (:82-88; needs v2 > v1)
  eps_small     250 draws with theta in [0, 2e-4) rad: the `incidence + refraction < eps` branch and its edge (:112-115)
  eps_grazing   250 draws with theta within 2e-4 rad of 90 deg -- opaque (v2 = 0), totally reflected (v2 > v1) and v2 == v1
              (the only transmitted case whose angle sum reaches pi): the `> pi - eps` branch and its edge (:116-118)
All values are float32-representable (what the C++ / the kernels receive); theta is handed to the scripts in degrees (f64).
"""
import numpy as np

FAMILIES = ("random", "limit", "eps_small", "eps_grazing")


def cases(family):
    """-> theta_rad (f64, f32-representable), v1, v2 (f64, f32-representable)"""
    rs = np.random.RandomState({"random": 60001, "limit": 60002, "eps_small": 60003, "eps_grazing": 60004}[family])
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)     # noqa: E731
    if family == "random":
        n = 10000
        th = np.radians(rs.uniform(0.0, 90.0, n))
        v1 = rs.uniform(0.05, 0.45, n)
        v2 = rs.uniform(0.001, 0.6, n)
        v2[rs.uniform(0, 1, n) < 0.05] = 0.0
        v1[::7] = 0.3                                              # the hot loop's own v1 (RadarCPU.cpp:107-110) stays well covered
    elif family == "limit":
        n = 500
        v1 = rs.uniform(0.05, 0.4, n)
        v2 = v1 * rs.uniform(1.02, 3.0, n)
        v1, v2 = f32(v1), f32(v2)
        th = np.arcsin(v1 / v2) + rs.uniform(-1e-3, 1e-3, n)
    elif family == "eps_small":
        n = 250
        th = rs.uniform(0.0, 2e-4, n)
        th[:10] = 0.0
        v1 = rs.uniform(0.05, 0.45, n)
        v2 = rs.uniform(0.001, 0.6, n)
    else:
        n = 250
        th = np.pi / 2 - rs.uniform(0.0, 2e-4, n)
        v1 = rs.uniform(0.05, 0.45, n)
        v2 = v1 * rs.uniform(1.0, 2.0, n)
        v2[0::3] = 0.0
        v2[1::3] = v1[1::3]
    return f32(th), f32(v1), f32(v2)


def direction(theta_rad):
    """incidence direction for the surface normal (-1, 0, 0): (cos, sin, 0) -- the convention of tests/test_oracle_kat.py"""
    return np.stack([np.cos(theta_rad), np.sin(theta_rad), np.zeros_like(theta_rad)], -1).astype(np.float32)
