"""The BRDF lobe against the reference's OWN python (round 6): scripts/radarays_snell_fresnel_brdf.py:9-24 keeps
energy_return_function(A, B, C, w) = clamp0(A + B max(cos w, 0) + (1 - A - B) max(cos w, 0)^C) and energy_reflect_function, run
from /root/reference by tests/golden/gen_pyref_brdf.py (the committed pyref_brdf.npy holds inputs and the script's outputs).
The C++ path's back_reflection_shader (radar_algorithms.h:168-187 with material.{ambient, diffuse, specular},
RadarCPU.cpp:310-316) = ambient + diffuse cos(w)^specular is that lobe on the sub-family B = 0, diffuse = 1 - A: 3,624 cases
(random; the corners w = 0 / the last float32 below pi/2, C = 0 / 0.1 / 2000, A = 0 / 1; the presets' neighbourhood).

What the comparison can say: the script works in f64, the C++ (and the oracle, and k_shade) in f32 -- cosf, powf on f32 inputs.
A relative error e of the cosine becomes C e in the lobe, and near grazing incidence the cosine itself is only known to
6e-8 / cos w: the law below is 1.5e-7 + lobe * (1e-7 + 6e-8 C (1 + 1 / cos w)), capped where the lobe has died anyway.  Measured:
92 % of the cases within 1e-7 of the script, 99 % within 1e-6, the worst 1.0e-5 (C = 2000 at w = 1e-4: cosf rounds to 1).
Before this fixture the shader was pinned by five transcribed known answers (tests/golden/survey_kat.json)."""
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    X = np.load(os.path.join(GOLDEN, "pyref_brdf.npy"))
    assert X.shape == (3624, 5) and np.isfinite(X).all()
    return X


def test_fixture_is_the_scripts_subfamily():
    """the two functions of the script agree with each other (reflect = return of the angle difference) and with the closed form
    the docstring states -- a check of the FIXTURE, so that what the oracle is compared with is what the header says"""
    X = _cases()
    A, C, w = X[:, 0], X[:, 1], X[:, 2]
    closed = A + (1.0 - A) * np.cos(w) ** C
    assert np.abs(X[:, 3] - closed).max() < 1e-15
    # the second function saw |(w_in + w) - w_in|: the same angle up to the rounding of that difference (amplified by C tan w)
    assert np.abs(X[:, 4] - X[:, 3]).max() < 1e-9
    assert (np.float32(A) == A).all() and (np.float32(C) == C).all() and (np.float32(w) == w).all()


def test_oracle_shader_against_the_reference_script(oracle):
    X = _cases()
    worst = 0.0; n_1e6 = 0
    for a, c, w, want, _ in X:
        got = oracle.back_reflection_shader(np.float32(w), 1.0, float(a), float(np.float32(1.0) - np.float32(a)), float(c))
        cw = math.cos(w)
        lobe = (1.0 - a) * cw ** c
        tol = 1.5e-7 + lobe * (1e-7 + 6e-8 * c * (1.0 + 1.0 / max(cw, 1e-7)))
        tol = min(tol, 1.5e-7 + lobe)            # (an error of the exponentiated cosine cannot exceed the lobe by much)
        err = abs(got - want)
        assert err <= tol, (a, c, w, got, want, err, tol)
        worst = max(worst, err); n_1e6 += err <= 1e-6
    # the plain statement: nearly all of the cases sit within 1e-6 of the script, all within 1.1e-5
    assert n_1e6 >= 0.98 * len(X) and worst < 1.1e-5, (n_1e6, worst)


def test_oracle_shader_scales_with_energy_and_diffuse(oracle):
    """the two things the sub-family cannot show: the shader is linear in the energy and in the diffuse weight (its own f32
    arithmetic: diffuse * 1 * ... + specular_fac * pow) -- so a free diffuse weight is pinned up to one f32 multiplication"""
    X = _cases()[::37]
    for a, c, w, want, _ in X:
        base = oracle.back_reflection_shader(np.float32(w), 1.0, 0.0, 1.0, float(c))              # cos^C alone
        for d in (0.0, 0.25, 0.8, 1.7):
            for e in (1.0, 0.37, 12.5):
                got = oracle.back_reflection_shader(np.float32(w), e, float(a), d, float(c))
                ref = np.float32(np.float32(np.float32(a) + np.float32(np.float32(d) * np.float32(base))) * np.float32(e))
                assert abs(got - float(ref)) <= 2.5e-7 * max(abs(float(ref)), 1e-30) + 1e-37, (a, c, w, d, e, got, float(ref))
