"""Host side of the seam, C versions (no GPU): the beam sampler RadarCPU::simulate re-runs after a dynamic reconfigure
(sample_cone_local, radar_algorithms.cpp:248-294 / RadarCPU.cpp:136-145) and the map loader
(rm::import_embree_map, radar_simulator.cpp:149) -- against the oracle and the Python twins."""
import math

import numpy as np
import pytest

from radarays_ros_amd import beams, meshio, native, scenes


@pytest.mark.parametrize("dist", [0, 1, 2, 3])
def test_rr_cone_dirs_is_bit_equal_to_the_oracle(oracle, dist):
    u, r = beams.variates(500, dist, seed=11 + dist)
    for width in (math.radians(10.0), math.radians(2.5)):
        a = native.cone_dirs(np.float32(width), dist, 0.8, u, r)
        b = oracle.sample_cone_local(np.float32(width), dist, 0.8, u, r)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (dist, width)


@pytest.mark.parametrize("dist", [0, 1, 2, 3])
def test_rr_sample_cone_local_draws_numpys_streams(oracle, dist):
    """Same seed -> the variates of beams.variates (numpy RandomState: MT19937, 53-bit doubles, polar Box-Muller with its
    cached second value) -> the directions of beams.sample_cone_local.  Checked bit for bit through the oracle's geometry
    (numpy's float32 cos / sin are allowed an ulp against libm's, beams.py itself is compared at 1e-6)."""
    for seed, n in ((42, 200), (7, 1001), (0, 3)):
        width = np.float32(8.0 * math.pi / 180.0)
        got = native.sample_cone_local(seed, width, n, dist, 0.8)
        u, r = beams.variates(n, dist, seed)
        want = oracle.sample_cone_local(width, dist, 0.8, u, r)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (dist, seed)
        py = beams.cone_dirs(width, dist, 0.8, u, r)
        assert np.abs(got - py).max() < 1e-6
    assert np.allclose(np.linalg.norm(native.sample_cone_local(3, 0.17, 64, dist, 0.8), axis=1), 1.0, atol=1e-6)


def test_rr_cone_dirs_rejects_bad_arguments():
    with pytest.raises(native.RRError):
        native.cone_dirs(0.1, 4, 0.8, [0.5], [0.5])
    with pytest.raises(native.RRError):
        native.sample_cone_local(1, 0.1, 4, -1, 0.8)


@pytest.mark.parametrize("binary", [True, False])
def test_rr_load_mesh_file_ply_equals_meshio(tmp_path, binary):
    s = scenes.heightfield_room(8, n_buildings=3)
    p = str(tmp_path / "m.ply")
    meshio.save_ply(p, s["verts"], s["faces"], binary=binary)
    c, py = native.load_mesh_file(p), meshio.load_mesh(p)
    assert np.array_equal(c["verts"], py["verts"]) and np.array_equal(c["faces"], py["faces"])
    assert np.array_equal(c["face_object_id"], py["face_object_id"]) and c["n_objects"] == 1
    assert np.array_equal(c["faces"], s["faces"])


def test_rr_load_mesh_file_polygons_big_endian_and_obj(tmp_path):
    v = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 1.5, 0]])
    p = tmp_path / "q.ply"
    hdr = ("ply\nformat binary_big_endian 1.0\ncomment made by hand\nelement vertex 5\nproperty double x\nproperty double y\nproperty double z\n"
           "property uchar red\nelement face 2\nproperty list uchar uint vertex_index\nelement edge 1\nproperty int a\nend_header\n")
    body = b""
    for r in v:
        body += np.array(r, ">f8").tobytes() + b"\x07"
    body += b"\x04" + np.array([0, 1, 2, 3], ">u4").tobytes() + b"\x03" + np.array([3, 2, 4], ">u4").tobytes() + b"\0\0\0\1"
    p.write_bytes(hdr.encode() + body)
    m = native.load_mesh_file(p)
    assert np.array_equal(m["verts"], v) and m["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [3, 2, 4]]
    assert np.array_equal(m["faces"], meshio.load_ply(str(p))["faces"])
    o = tmp_path / "s.obj"
    o.write_text("# two objects\no ground\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n"
                 "o wall\nv 0 0 1\nv 1 0 1\nv 1 1 1\nf 5/1/1 6/2/1 7/3/1\nf -3 -2 -1\nf 5//1 6//1 7//1\n")
    c, py = native.load_mesh_file(o), meshio.load_mesh(str(o))
    assert c["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 5, 6], [4, 5, 6]] and c["n_objects"] == 2
    for k in ("verts", "faces", "face_object_id"):
        assert np.array_equal(c[k], py[k]), k


def test_rr_load_mesh_file_errors(tmp_path):
    p = tmp_path / "x.ply"
    p.write_text("nope\n")
    with pytest.raises(native.RRError, match="not a PLY"):
        native.load_mesh_file(p)
    with pytest.raises(native.RRError, match="unsupported mesh format"):
        native.load_mesh_file("scene.dae")
    with pytest.raises(native.RRError, match="cannot open"):
        native.load_mesh_file(tmp_path / "missing.obj")
    t = tmp_path / "t.ply"
    t.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n")
    with pytest.raises(native.RRError, match="out of range"):
        native.load_mesh_file(t)


def test_host_side_under_asan(tmp_path):
    """csrc/rr_host.cpp under AddressSanitizer + UBSan (CPU; it has no GPU code): the sampler's unit vectors and argument
    checks, the three well-formed files, and 6,000 damaged ones (truncated, bytes flipped, digits inserted into counts and
    indices, bytes removed) -- every load ends in an error code or in a mesh whose indices are in range, never in a crash
    (found on the first run: an inflated vertex count reserved 490 GB; counts are now bounded by the file's size)."""
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ missing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_side_check")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "host_side_check.cpp"),
                    os.path.join(root, "radarays_ros_amd", "csrc", "rr_host.cpp"), "-o", exe], check=True)
    work = tmp_path / "files"
    work.mkdir()
    r = subprocess.run([exe, str(work)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
    assert "all: ok" in r.stdout
