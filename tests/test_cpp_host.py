"""The C++ host side above the C ABI (include/radarays_ros_amd/RadarHIP.hpp, the mirror of the
reference's Radar / RadarCPU classes) driven by a plain C++ program -- no Python, no torch in
the product path."""
import os
import struct
import subprocess

import numpy as np
import pytest

from common import GOLDEN, golden_beams, image_diff, mats_tuple
from radarays_ros_amd import params, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "radar_hip_demo.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "radar_hip_demo")

import sys  # noqa: E402
sys.path.insert(0, GOLDEN)
import gen_oracle_images as gen  # noqa: E402


def build_demo():
    if (not os.path.exists(EXE)) or os.path.getmtime(EXE) < max(
            os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "radarays_ros_amd", "RadarHIP.hpp")),
            os.path.getmtime(os.path.join(ROOT, "include", "radarays_mi355.h"))):       # the rr_config layout lives there
        subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                        "-L", os.path.join(ROOT, "radarays_ros_amd"), "-lradarays_mi355",
                        "-Wl,-rpath,$ORIGIN/../../radarays_ros_amd"], check=True)
    return EXE


def _blob(a, dt):
    a = np.ascontiguousarray(a, dt).ravel()
    return struct.pack("<Q", a.size) + a.tobytes()


def write_scene(path, s, mats, beams_, pose, cfg):
    with open(path, "wb") as f:
        f.write(_blob(s["verts"], np.float32) + _blob(s["faces"], np.uint32) + _blob(s["face_object_id"], np.uint32))
        f.write(_blob([x for m in mats for x in m.astuple()], np.float32) + _blob(s["object_materials"], np.int32))
        f.write(_blob(beams_, np.float32) + _blob(pose, np.float32))
        f.write(_blob([cfg.n_reflections, cfg.ambient_noise, cfg.scroll_image, cfg.signal_denoising_triangular_width,
                       cfg.energy_max, cfg.signal_max, cfg.resolution], np.float64))


def test_cpp_host_builds_and_fails_loudly_without_gpu(tmp_path, native_lib):
    native_lib.build()
    exe = build_demo()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    s = scenes.box12()
    p = str(tmp_path / "s.bin")
    write_scene(p, s, params.kaist_materials(), golden_beams(4), scenes.default_pose("box12"),
                params.kaist_preset(n_reflections=1, ambient_noise=0))
    r = subprocess.run([exe, p, str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 6 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_host_matches_oracle(tmp_path, oracle):
    exe = build_demo()
    s = gen.two_room_scene()
    mats = params.kaist_materials() + [params.PENETRABLE]
    cfg = params.kaist_preset(n_reflections=3, ambient_noise=0, scroll_image=9)
    pose = scenes.default_pose("box12")
    p, o = str(tmp_path / "s.bin"), str(tmp_path / "o.bin")
    write_scene(p, s, mats, golden_beams(40), pose, cfg)
    # the same map as an OBJ file with one `o` group per object: the C++ side loads it with rr_load_mesh_file
    objp = str(tmp_path / "map.obj")
    with open(objp, "w") as fo:
        for v in s["verts"]:
            fo.write("v %r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])))
        fid = np.asarray(s["face_object_id"])
        assert np.all(np.diff(fid.astype(np.int64)) >= 0) and fid[0] == 0       # objects are contiguous runs 0, 1, ...
        cur = -1
        for f_, ob in zip(s["faces"], fid):
            while cur < int(ob):
                cur += 1
                fo.write("o object%d\n" % cur)
            fo.write("f %d %d %d\n" % (f_[0] + 1, f_[1] + 1, f_[2] + 1))
    r = subprocess.run([exe, p, o, objp], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Couldn't get Transform" in r.stdout           # the null-ImagePtr path was exercised first
    raw = open(o, "rb").read()
    h, w = struct.unpack("<II", raw[:8])
    img = np.frombuffer(raw[8:8 + h * w], np.uint8).reshape(h, w)
    # the backend that drew its own beam (seed 42): the directions are beams.py's, its image the oracle's for them
    off = 8 + h * w
    nb = struct.unpack("<Q", raw[off:off + 8])[0]
    drawn = np.frombuffer(raw[off + 8:off + 8 + 4 * nb], np.float32).reshape(-1, 3)
    img_drawn = np.frombuffer(raw[off + 8 + 4 * nb:off + 8 + 4 * nb + h * w], np.uint8).reshape(h, w)
    from radarays_ros_amd import beams as B
    assert drawn.shape == (40, 3) and np.abs(drawn - B.sample_cone_local(10.0, 40, 2, 0.8, seed=42)).max() < 1e-6
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
    o8, of, ost = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, golden_beams(40), pose)
    d = np.abs(img.astype(int) - o8.astype(int))
    assert (h, w) == (3424, 400) and d.max() <= 1 and (d > 0).mean() < 1e-3
    assert "wave_passes %d" % ost["wave_passes"] in r.stdout
    d8, _, _ = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, drawn, pose)
    dd = np.abs(img_drawn.astype(int) - d8.astype(int))
    assert dd.max() <= 1 and (dd > 0).mean() < 1e-3
