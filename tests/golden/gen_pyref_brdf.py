"""Generates tests/golden/pyref_brdf.npy by RUNNING the reference's own scripts/radarays_snell_fresnel_brdf.py from /root/reference
(dev container only; what is committed are the script's OUTPUTS beside the inputs they belong to, never the script).

    MPLBACKEND=Agg python tests/golden/gen_pyref_brdf.py

The script keeps its lobe in two importable functions (scripts/radarays_snell_fresnel_brdf.py:9-24):
    energy_return_function(A, B, C, w)            = clamp0(A + B max(cos w, 0) + (1 - A - B) max(cos w, 0)^C)
    energy_reflect_function(A, B, C, w_in, w_ref) = energy_return_function(A, B, C, |w_ref - w_in|)
The C++ path's back_reflection_shader (radar_algorithms.h:168-187, called with material.{ambient, diffuse, specular} at
RadarCPU.cpp:310-316) is  ambient + diffuse cos(w)^specular  -- the script's lobe on the sub-family B = 0 with diffuse = 1 - A
(C = 1 reaches the same sub-family through B).  That sub-family is what this fixture can pin; a Lambert term B cos w with a free
diffuse weight does not exist in the C++.

pyref_brdf.npy  [n][5] f64: A, C, w, energy_return_function(A, 0, C, w), energy_reflect_function(A, 0, C, w_in, w_in + w)
                 (A, C, w are float32 values, so the C++ types see the very same inputs; n = 3,624: 116 KB)
"""
import contextlib
import io
import os
import runpy
import sys

sys.dont_write_bytecode = True   # never write into /root/reference
os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/scripts"
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    g = runpy.run_path(os.path.join(REF, "radarays_snell_fresnel_brdf.py"), run_name="__main__")
ret, refl = g["energy_return_function"], g["energy_reflect_function"]

rng = np.random.RandomState(20261003)
f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)
below_half_pi = float(np.nextafter(np.float32(np.pi / 2), np.float32(0)))     # the largest float32 with a positive cosine
rows = []
# random
n = 3000
rows.append(np.stack([f32(rng.uniform(0, 1, n)), f32(rng.uniform(0, 200, n)), f32(rng.uniform(0, below_half_pi, n))], 1))
# the corners: normal and grazing incidence, exponents 0 / tiny / 1 / large, ambient 0 / 1
A = [0.0, 0.01, 0.3, 0.5, 0.99, 1.0]
Cx = [0.0, 0.1, 1.0, 2.0, 25.0, 200.0, 2000.0]
W = [0.0, 1e-4, 0.5, 1.0, 1.5, 1.57, below_half_pi]
rows.append(f32(np.array([(a, c, w) for a in A for c in Cx for w in W])))
# the presets' neighbourhood: ambient + diffuse = 1 is how config/*.yaml writes most materials
n = 330
rows.append(np.stack([f32(rng.choice([0.2, 0.5, 0.8, 0.9], n)), f32(rng.choice([1.0, 10.0, 50.0, 100.0, 1000.0, 3000.0], n)),
                      f32(rng.uniform(0, below_half_pi, n))], 1))
X = np.concatenate(rows, 0)
out = np.zeros((len(X), 5))
out[:, :3] = X
w_in = rng.uniform(-1.0, 1.0, len(X))
for i, (a, c, w) in enumerate(X):
    out[i, 3] = ret(a, 0.0, c, np.array([w]))[0]
    out[i, 4] = refl(a, 0.0, c, w_in[i], np.array([w_in[i] + w]))[0]
np.save(os.path.join(HERE, "pyref_brdf.npy"), out)
print("pyref_brdf.npy", out.shape, os.path.getsize(os.path.join(HERE, "pyref_brdf.npy")), "bytes; finite:", np.isfinite(out).all())
