"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np

from radarays_ros_amd import beams, params, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


from radarays_ros_amd.fixtures import golden_beams, materials_for  # noqa: E402,F401  (shared with bench.py / tools)


def mats_tuple(mats):
    return [m.astuple() for m in mats]


def image_diff(a_f32, b_f32, a_u8, b_u8):
    """SURVEY §8d parity gate: mean(|f32_gpu - f32_cpu|)/255, plus u8 stats."""
    fa = np.nan_to_num(a_f32.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    fb = np.nan_to_num(b_f32.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    d8 = np.abs(a_u8.astype(np.int32) - b_u8.astype(np.int32))
    return {
        "mean_dev": float(np.mean(np.abs(fa - fb)) / 255.0),
        "max_abs_f32": float(np.max(np.abs(fa - fb))),
        "u8_mismatch_frac": float(np.mean(d8 > 0)),
        "u8_gt1_frac": float(np.mean(d8 > 1)),
        "u8_max": int(d8.max()),
    }
