"""Golden images from the CPU oracle (brute-force nearest hit) on tiny scenes.
Regenerate with:  python tests/golden/gen_oracle_images.py
They pin the oracle AND the HIP path across rounds (the reference itself ships no
golden image: SURVEY.md §4)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O  # noqa: E402
from radarays_ros_amd import params, scenes  # noqa: E402
from common import golden_beams, mats_tuple  # noqa: E402


def two_room_scene():
    """box12 room (object 0 -> wall) + an inner penetrable box (object 1)."""
    a = scenes.box12()
    v2, f2 = scenes._box_tris([3.0, -2.0, -1.0], [6.0, 2.5, 2.0], vbase=len(a["verts"]))
    return {"verts": np.concatenate([a["verts"], v2]), "faces": np.concatenate([a["faces"], f2]),
            "face_object_id": np.concatenate([a["face_object_id"], np.ones(12, np.uint32)]),
            "object_materials": [1, 2], "name": "box12_inner"}


def case_config1():
    """BASELINE.json configs[0]: single azimuth, 1 bounce, 100 rays, 12-triangle box."""
    s = scenes.box12()
    cfg = params.kaist_preset(n_reflections=1, ambient_noise=0)
    return s, cfg, params.kaist_materials(), golden_beams(100), scenes.default_pose("box12"), (0, 1), None


def case_multibounce():
    s = two_room_scene()
    cfg = params.kaist_preset(n_reflections=4, ambient_noise=0)
    return s, cfg, params.kaist_materials() + [params.PENETRABLE], golden_beams(64), scenes.default_pose("box12"), (0, 400), None


def case_noise():
    s = two_room_scene()
    cfg = params.kaist_preset(n_reflections=2, ambient_noise=2, scroll_image=37)
    rnd = (np.random.RandomState(7).uniform(0, 1, 400) * 1000.0).astype(np.float32)
    return s, cfg, params.kaist_materials() + [params.PENETRABLE], golden_beams(32), scenes.default_pose("box12"), (0, 64), rnd


CASES = {"config1": case_config1, "multibounce": case_multibounce, "noise": case_noise}


def run(name):
    s, cfg, mats, beams_, pose, (a0, a1), rnd = CASES[name]()
    sc = O.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
    u8, f32, st = O.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, beams_, pose,
                             noise_rnd=rnd, az_begin=a0, az_end=a1)
    return u8, f32, st


if __name__ == "__main__":
    for name in CASES:
        u8, f32, st = run(name)
        cols = np.flatnonzero(u8.any(axis=0))
        np.savez_compressed(os.path.join(HERE, "oracle_%s.npz" % name), u8=u8,
                            wave_passes=st["wave_passes"], signals=st["signals"])
        print(name, "nonzero columns", len(cols), "max", u8.max(), st)
