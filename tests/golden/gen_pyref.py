"""Generates tests/golden/pyref_*.json by IMPORTING the reference's own python
scripts from /root/reference (only possible in the dev container; the vectors,
not the scripts, are committed).

    MPLBACKEND=Agg python tests/golden/gen_pyref.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True   # never write into /root/reference
os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/scripts"
sys.path.insert(0, os.path.join(REF, "reflections"))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import fresnel as ref_fresnel  # noqa: E402   scripts/reflections/fresnel.py
import snell_multi as ref_snell  # noqa: E402  scripts/reflections/snell_multi.py
import maxwell_boltzmann as ref_mb  # noqa: E402  scripts/maxwell_boltzmann.py

HERE = os.path.dirname(os.path.abspath(__file__))

cases = []
for theta_deg in [0.0, 5.0, 10.0, 20.0, 30.0, 45.0, 60.0, 75.0, 85.0, 89.0]:
    for v2 in [0.03, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.45]:
        th = np.radians(theta_deg)
        d = np.array([np.cos(th), np.sin(th)])
        n = np.array([-1.0, 0.0])
        v1 = 0.3
        # reference convention (radar_algorithms.h:62-63): n1 := v2, n2 := v1
        refl, _ = ref_fresnel.fresnel_reflect_dir(n, d, v2, v1)
        refr, _ = ref_fresnel.fresnel_refract_dir(n, d, v2, v1)
        cases.append({
            "theta_deg": theta_deg, "v1": v1, "v2": v2,
            "incident_angle": float(ref_fresnel.incident_angle(n, d)),
            "reflect_dir": [float(x) for x in refl],
            "refract_dir": [float(x) for x in refr],
            "snell_multi_reflect": [float(x) for x in ref_snell.snell_reflect_dir(n, d)],
        })
json.dump({"_provenance": "scripts/reflections/fresnel.py:25-57, snell_multi.py:11-20 imported from /root/reference",
           "cases": cases}, open(os.path.join(HERE, "pyref_snell.json"), "w"), indent=1)

mb = []
for width, mode in [(50, 20), (35, 12), (100, 40), (10, 3)]:
    x = np.arange(width, dtype=np.float64)
    y = ref_mb.maxwell_boltzmann_pdf(x, a=ref_mb.maxwell_boltzmann_a_from_mode(mode))
    y = y / y.sum()
    mb.append({"width": width, "mode": mode, "normalized": [float(v) for v in y]})
json.dump({"_provenance": "scripts/maxwell_boltzmann.py:6-10 imported from /root/reference", "cases": mb},
          open(os.path.join(HERE, "pyref_mb.json"), "w"), indent=1)
# Fresnel ENERGIES: scripts/reflections/fresnel.py:99-165 computes rs / rp / Reff / Teff inside the render() closure of its
# __main__ block (interactive tool).  Run it as a script under the Agg backend, then drive it through its own slider
# callbacks (update_n1 / update_n2 / update_inc_angle, module globals after the run); render() stores Reff and Teff as the
# alpha of its two line artists -- read back exactly, no parsing.  Python's n1 is "upper medium (or v2)", n2 "lower (or
# v1)": the C++ convention n1 := v2, n2 := v1 (radar_algorithms.h:62-63).  Cases whose Reff leaves [0, 1] (v2 = 0: the
# artist refuses the alpha) are left to the survey's known answers.
import contextlib  # noqa: E402
import io  # noqa: E402
import runpy  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    gf = runpy.run_path(os.path.join(REF, "reflections", "fresnel.py"), run_name="__main__")
energy = []
with contextlib.redirect_stdout(io.StringIO()):
    def drive(fn, val):          # (a render whose Reff rounds to 1 + 1e-15 makes the artist raise: that case is skipped)
        try:
            gf[fn](val)
            return True
        except ValueError:
            return False
    for v2 in [0.03, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.45]:
        drive("update_inc_angle", 30.0)
        drive("update_n1", v2)
        drive("update_n2", 0.3)
        for theta_deg in [0.0, 2.0, 5.0, 10.0, 20.0, 30.0, 45.0, 60.0, 75.0, 85.0, 89.0]:
            if drive("update_inc_angle", theta_deg):
                energy.append({"theta_deg": theta_deg, "v1": 0.3, "v2": v2,
                               "Reff": float(gf["line_refl"].get_alpha()), "Teff": float(gf["line_refr"].get_alpha())})
json.dump({"_provenance": "scripts/reflections/fresnel.py:99-165 (render(): rs, rp, Reff, Teff) run from /root/reference via runpy under "
                          "MPLBACKEND=Agg and driven through its slider callbacks; surface normal (0, 1), ray (sin a, -cos a); "
                          "f64 throughout (the C++ narrows the angles to f32: compare at 1e-6)", "cases": energy},
          open(os.path.join(HERE, "pyref_fresnel_energy.json"), "w"), indent=1)
print("wrote pyref_fresnel_energy.json (%d cases)" % len(energy))

# scripts/radaray_beams.py keeps its formulas under `if __name__ == '__main__'`: run it as a script (Agg backend:
# plt.show() returns at once) with numpy's global generator seeded and read the arrays it leaves behind.  The uniform
# variates are overwritten by the normal ones (:75), so the draws are repeated here in the script's order (:22,25,75,76).
np.random.seed(20240217)
g = runpy.run_path(os.path.join(REF, "radaray_beams.py"), run_name="__main__")
np.random.seed(20240217)
n = int(g["n_samples"])
u_uniform = np.random.uniform(0.0, 1.0, size=n)
np.random.uniform(-np.pi, np.pi, size=n)
u_normal = np.random.normal(0.0, 1.0, size=n)
assert np.array_equal(u_normal, g["samples"]) and np.array_equal(u_uniform * g["radius"], g["r_approx"])
K = 96
json.dump({"_provenance": "scripts/radaray_beams.py:8-25,75-92 run from /root/reference (runpy, MPLBACKEND=Agg, np.random.seed(20240217)); "
                          "the first %d of its %d samples; width / radius in degrees as in the script" % (K, n),
           "width": float(g["width"]), "p_in_cone": float(g["p_in_cone"]), "radius": float(g["radius"]), "z": float(g["z"]),
           "uniform": [float(x) for x in u_uniform[:K]], "normal": [float(x) for x in u_normal[:K]],
           "D1_r_approx": [float(x) for x in g["r_approx"][:K]], "D2_r": [float(x) for x in g["r"][:K]],
           "D3_r1": [float(x) for x in g["r1"][:K]], "D4_r2": [float(x) for x in g["r2"][:K]]},
          open(os.path.join(HERE, "pyref_beams.json"), "w"), indent=1)
print("wrote pyref_beams.json (%d samples per law)" % K)
# Noise amplitude law: scripts/func_deformer.py:6-21 is an importable function (the module also draws a figure: Agg).  Its
# labels are the mirror of the C++'s (its noise_at_0 belongs to signal_ = 0, i.e. to the STRONGEST signal; the C++'s
# ambient_noise_at_signal_0 to a signal of 0); its constants are 0.01 / 0.4, the signal's minimum must be 0 to match
# RadarCPU.cpp:499 (signal_min = 0).
# (the module imports a third-party `perlin` package further down, :77, which this image lacks: only the part of the FILE
# that defines the function -- its first 22 lines, up to the plotting script -- is executed, straight from /root/reference)
_fd_path = os.path.join(REF, "func_deformer.py")
_fd_src = open(_fd_path).read().split("\nN = 1000")[0]
_fd_ns = {}
exec(compile(_fd_src, _fd_path, "exec"), _fd_ns)


class ref_fd:  # noqa: N801
    noise_amplitude = staticmethod(_fd_ns["noise_amplitude"])


sig = np.concatenate([[0.0], np.linspace(0.0, 3.7, 40) ** 2, [13.69]])
json.dump({"_provenance": "scripts/func_deformer.py:6-21 noise_amplitude() imported from /root/reference; signal minimum 0, maximum 13.69; the "
                          "script's constants: amplitude x 0.4 at signal 0, x 0.01 at the maximum",
           "at_zero_signal": 0.4, "at_max_signal": 0.01, "signal": [float(x) for x in sig],
           "noise_amp": [float(x) for x in ref_fd.noise_amplitude(sig)]},
          open(os.path.join(HERE, "pyref_noise_amp.json"), "w"), indent=1)
print("wrote pyref_noise_amp.json (%d samples)" % len(sig))
print("wrote pyref_snell.json (%d cases), pyref_mb.json (%d cases)" % (len(cases), len(mb)))
