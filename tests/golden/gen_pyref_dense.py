"""Generates tests/golden/pyref_dense_*.npy by IMPORTING / RUNNING the reference's own python scripts from /root/reference
(dev container only; what is committed are the scripts' OUTPUTS as plain float64 arrays -- the inputs are rebuilt from the
seeded generator in pyref_cases.py -- never the scripts).

    MPLBACKEND=Agg python tests/golden/gen_pyref_dense.py

Per family F of pyref_cases.FAMILIES (rows in the generator's order):
  pyref_dense_F_refl.npy    [n][2] f64: fresnel_reflect_dir of scripts/reflections/fresnel.py:25-33
  pyref_dense_F_refr.npy    [n][2] f64: fresnel_refract_dir of scripts/reflections/fresnel.py:35-57 (zeros: not transmitted)
  pyref_dense_F_snell.npy   [n][2] f64: snell_refract_dir of scripts/reflections/snell_multi.py:14-20 (NaN where its sqrt goes
                            negative or a velocity is 0)                                  (each file <= 160 KB)
  pyref_dense_F_energy.npy  [n][2] f64: Reff, Teff as the render() closure of scripts/reflections/fresnel.py:99-165 computes them
                            (run under the Agg backend, driven through its slider callbacks; NaN where the artist refused an
                            alpha outside [0, 1])
Conventions: the scripts' surface normal is (0, 1) and the ray (sin a, -cos a) in render(); the importable functions are called
with normal (-1, 0) and ray (cos a, sin a) like gen_pyref.py does.  Python's n1 is the C++'s n1 := v2, n2 := v1.
"""
import contextlib
import io
import os
import runpy
import sys

sys.dont_write_bytecode = True   # never write into /root/reference
os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/scripts"
sys.path.insert(0, os.path.join(REF, "reflections"))
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import fresnel as ref_fresnel  # noqa: E402   scripts/reflections/fresnel.py
import snell_multi as ref_snell  # noqa: E402  scripts/reflections/snell_multi.py
import pyref_cases  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    gf = runpy.run_path(os.path.join(REF, "reflections", "fresnel.py"), run_name="__main__")


def drive(fn, val):
    try:
        gf[fn](val)
        return True
    except ValueError:       # the artist refuses an alpha outside [0, 1] (Reff = 1 + 1e-16): the state is set, the value unusable
        return False


for fam in pyref_cases.FAMILIES:
    th, v1, v2 = pyref_cases.cases(fam)
    n = len(th)
    dirs = np.full((n, 6), np.nan)
    energy = np.full((n, 2), np.nan)
    nrm = np.array([-1.0, 0.0])
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        for i in range(n):
            d = np.array([np.cos(th[i]), np.sin(th[i])])
            dirs[i, 0:2] = ref_fresnel.fresnel_reflect_dir(nrm, d, v2[i], v1[i])[0]
            dirs[i, 2:4] = ref_fresnel.fresnel_refract_dir(nrm, d, v2[i], v1[i])[0]
            if v1[i] > 0.0 and v2[i] > 0.0:
                dirs[i, 4:6] = ref_snell.snell_refract_dir(nrm, d, v2[i], v1[i])
            ok = drive("update_n1", float(v2[i]))
            ok = drive("update_n2", float(v1[i])) and ok
            ok = drive("update_inc_angle", float(np.degrees(th[i])))      # the last render has all three values in force
            if ok:
                energy[i] = (float(gf["line_refl"].get_alpha()), float(gf["line_refr"].get_alpha()))
    for k, name in enumerate(("refl", "refr", "snell")):
        np.save(os.path.join(HERE, "pyref_dense_%s_%s.npy" % (fam, name)), dirs[:, 2 * k:2 * k + 2])
    np.save(os.path.join(HERE, "pyref_dense_%s_energy.npy" % fam), energy)
    print("%-12s %5d cases, %5d with energies, %5d transmitted" % (fam, n, int(np.isfinite(energy[:, 0]).sum()), int(np.any(dirs[:, 2:4] != 0, axis=1).sum())))
