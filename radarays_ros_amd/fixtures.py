"""Inputs every harness of this package shares (bench.py, tools/, tests/): the stored beam-sample
directions and the material table that goes with a synthetic scene.

The beam samples are a FIXTURE because the reference draws them with std::mt19937 +
libstdc++ distributions (radar_algorithms.cpp:263-289), which are not portable across standard
libraries: `data/beam_dirs_seed42_n1000.npy` holds 1000 directions generated once
(seed 42, KAIST preset: beam width 10 deg, sample distribution 2, p_in_cone 0.8)."""
import os

import numpy as np

from . import params

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def golden_beams(n):
    d = np.load(os.path.join(DATA, "beam_dirs_seed42_n1000.npy"))
    assert d.shape == (1000, 3) and d.dtype == np.float32
    if n > len(d):
        raise ValueError("the fixture holds 1000 beam samples")
    return np.ascontiguousarray(d[:n])


def materials_for(scene):
    """KAIST materials (config/mulran_kaist02.yaml:8-20) + the penetrable one of the Snell/Fresnel
    configs when the scene has a second object; config 5's 8-material table for per-triangle scenes."""
    if "_pertri" in scene.get("name", ""):
        return params.config5_materials()
    m = params.kaist_materials()
    if max(scene["object_materials"]) >= 2:
        m = m + [params.PENETRABLE]
    return m
