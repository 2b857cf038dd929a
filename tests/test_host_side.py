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


def test_object_order_permutes_the_object_ids_and_nothing_else(tmp_path):
    """The material list of a scene is indexed by object id (m_object_materials[obj_id], RadarCPU.cpp:268;
    config/oru4_test.yaml:37-56); this build numbers objects in depth-first scene order, rmagine's importer may not
    (radar_simulator.cpp:149).  rr_mesh_reorder_objects / the adapter's ~hip_object_order renumber by NAME: only
    face_object_id and the order of the names change."""
    o = tmp_path / "scene.obj"
    o.write_text("o ground\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n"
                 "o wall\nv 0 0 1\nv 1 0 1\nv 1 1 1\nf 5 6 7\n"
                 "o roof\nv 0 0 2\nv 1 0 2\nv 1 1 2\nf 8 9 10\nf 10 9 8\n"
                 "o pole\nv 0 0 3\nv 1 0 3\nv 1 1 3\nf 11 12 13\n")
    base = native.load_mesh_file(o)
    assert base["object_names"] == ["ground", "wall", "roof", "pole"] and base["face_object_id"].tolist() == [0, 0, 1, 2, 2, 3]
    for order in (["roof", "ground", "pole", "wall"], ["pole"], ["wall", "ground"], []):
        m = native.load_mesh_file(o, object_order=order)
        assert np.array_equal(m["verts"], base["verts"]) and np.array_equal(m["faces"], base["faces"]) and m["n_objects"] == 4
        want_names = list(order) + [n for n in base["object_names"] if n not in order]      # unlisted: relative order kept, behind the listed
        assert m["object_names"] == want_names
        new_id = {n: k for k, n in enumerate(want_names)}
        assert m["face_object_id"].tolist() == [new_id[base["object_names"][i]] for i in base["face_object_id"]]
    # a material list written for ANOTHER numbering gives the same per-face materials once the objects are named in its order
    mats_theirs = {"roof": 3, "ground": 1, "pole": 4, "wall": 2}
    theirs = native.load_mesh_file(o, object_order=["roof", "ground", "pole", "wall"])
    per_face_theirs = [[3, 1, 4, 2][i] for i in theirs["face_object_id"]]
    per_face_ours = [[mats_theirs[n] for n in base["object_names"]][i] for i in base["face_object_id"]]
    assert per_face_theirs == per_face_ours
    for bad, msg in ((["ground", "nope"], "no object named 'nope'"), (["wall", "wall"], "listed twice")):
        with pytest.raises(native.RRError, match=msg):
            native.load_mesh_file(o, object_order=bad)
    dup = tmp_path / "dup.obj"
    dup.write_text("o a\nv 0 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3\no a\nv 0 0 1\nv 1 0 1\nv 1 1 1\nf 4 5 6\n")
    if native.load_mesh_file(dup)["object_names"].count("a") == 2:
        with pytest.raises(native.RRError, match="two objects named"):
            native.load_mesh_file(dup, object_order=["a"])
    ply = tmp_path / "t.ply"
    ply.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
                   "property list uchar int vertex_index\nend_header\n0 0 0\n1 0 0\n1 1 0\n3 0 1 2\n")
    with pytest.raises(native.RRError, match="no object names"):
        native.load_mesh_file(ply, object_order=["x"])


def test_rr_load_mesh_file_errors(tmp_path):
    p = tmp_path / "x.ply"
    p.write_text("nope\n")
    with pytest.raises(native.RRError, match="not a PLY"):
        native.load_mesh_file(p)
    with pytest.raises(native.RRError, match="unsupported mesh format"):
        native.load_mesh_file("scene.stl")
    with pytest.raises(native.RRError, match="cannot open"):
        native.load_mesh_file(tmp_path / "missing.obj")
    t = tmp_path / "t.ply"
    t.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n")
    with pytest.raises(native.RRError, match="out of range"):
        native.load_mesh_file(t)
    # advisor, round 4: an element without properties and an absurd count reads nothing per row, so nothing failed at EOF
    # and the row loop spun for 2^64 iterations; every element's count is now checked against the file size
    import time
    head = "ply\nformat ascii 1.0\nelement foo 18446744073709551615\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
    t.write_text(head + "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    t0 = time.time()
    with pytest.raises(native.RRError, match="element 'foo': count exceeds the file"):
        native.load_mesh_file(t)
    assert time.time() - t0 < 5.0
    # ... an empty element with a sane count is simply skipped
    t.write_text(head.replace("18446744073709551615", "7") + "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    assert native.load_mesh_file(t)["faces"].tolist() == [[0, 1, 2]]
    # an index that is not a number in [0, 2^32): NaN / inf (casting them is undefined), negative, or large enough to wrap
    import struct
    for bad in (float("nan"), float("inf"), -1.0, 4294967296.0 + 1.0):
        body = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0) + struct.pack("<B3d", 3, 0.0, 1.0, bad)
        t.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                      b"element face 1\nproperty list uchar double vertex_indices\nend_header\n" + body)
        with pytest.raises(native.RRError, match="face index outside"):
            native.load_mesh_file(t)
    o = tmp_path / "neg.obj"
    o.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -9\n")
    with pytest.raises(native.RRError, match="before the first vertex"):
        native.load_mesh_file(o)


DAE_RICH = """<?xml version="1.0" encoding="utf-8"?>
<!DOCTYPE COLLADA [ <!ENTITY x "y"> ]>
<!-- every primitive kind, two material groups in one geometry, library nodes, a namespace prefix -->
<c:COLLADA xmlns:c="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
  <c:asset><c:unit name="inch" meter="0.0254"/><c:up_axis>Z_UP</c:up_axis></c:asset>
  <c:library_geometries>
    <c:geometry id="G" name="Hall &amp; &quot;Annex&quot;"><c:mesh>
      <c:source id="G-pos"><c:float_array id="G-pos-a" count="28">9 9 9 9  7 0 0 0  7 1 0 0  7 1 1 0  7 0 1 0  7 0.5 1.5e0 0  7 -1 .5 +2</c:float_array>
        <c:technique_common><c:accessor source="#G-pos-a" count="6" stride="4" offset="5"/></c:technique_common></c:source>
      <c:vertices id="G-v"><c:input semantic="POSITION" source="#G-pos"/></c:vertices>
      <c:triangles count="2" material="brick"><c:input semantic="NORMAL" source="#none" offset="0"/>
        <c:input semantic="VERTEX" source="#G-v" offset="1"/><c:p>0 0 0 1 0 2</c:p><c:p> 0 0  0 2  0 3 </c:p></c:triangles>
      <c:polygons count="1" material="glass"><c:input semantic="VERTEX" source="#G-v" offset="0"/><c:p>0 1 2 3 4</c:p></c:polygons>
      <c:lines count="1"><c:input semantic="VERTEX" source="#G-v" offset="0"/><c:p>0 1</c:p></c:lines>
    </c:mesh></c:geometry>
    <c:geometry id="S"><c:mesh>
      <c:source id="S-pos"><c:float_array id="S-pos-a" count="18"><![CDATA[0 0 0 1 0 0 1 1 0 0 1 0 0.5 1.5 0 -1 0.5 2]]></c:float_array>
        <c:technique_common><c:accessor source="#S-pos-a" count="6" stride="3"/></c:technique_common></c:source>
      <c:vertices id="S-v"><c:input semantic="POSITION" source="#S-pos"/></c:vertices>
      <c:tristrips count="1"><c:input semantic="VERTEX" source="#S-v" offset="0"/><c:p>0 1 3 2 4</c:p></c:tristrips>
    </c:mesh></c:geometry>
    <c:geometry id="F"><c:mesh>
      <c:source id="F-pos"><c:float_array id="F-pos-a" count="15">0 0 0 1 0 0 1 1 0 0 1 0 -1 1 3</c:float_array>
        <c:technique_common><c:accessor source="#F-pos-a" count="5" stride="3"/></c:technique_common></c:source>
      <c:vertices id="F-v"><c:input semantic="POSITION" source="#F-pos"/></c:vertices>
      <c:trifans count="2"><c:input semantic="VERTEX" source="#F-v" offset="0"/><c:p>0 1 2 3</c:p><c:p>4 3 2 1 0</c:p></c:trifans>
    </c:mesh></c:geometry>
    <c:geometry id="Spline"><c:spline/></c:geometry>
  </c:library_geometries>
  <c:library_nodes>
    <c:node id="Prop"><c:rotate>1 1 0 30</c:rotate><c:instance_geometry url="#F"/>
      <c:node id="PropChild"><c:translate>0 0 1</c:translate><c:instance_geometry url="#S"/></c:node></c:node>
  </c:library_nodes>
  <c:library_visual_scenes>
    <c:visual_scene id="Unused"><c:node><c:instance_geometry url="#S"/></c:node></c:visual_scene>
    <c:visual_scene id="Main">
      <c:node id="A"><c:translate>1 2 3</c:translate><c:rotate>0 0 1 45</c:rotate><c:scale>1 2 0.5</c:scale>
        <c:instance_geometry url="#G"><c:bind_material/></c:instance_geometry>
        <c:instance_node url="#Prop"/>
        <c:node id="A1"><c:matrix>0 -1 0 4  1 0 0 5  0 0 1 6  0 0 0 1</c:matrix><c:instance_geometry url="#S"/>
          <c:instance_geometry url="#Missing"/></c:node>
        <c:instance_node url="#Prop"/>
      </c:node>
      <c:node id="B"><c:lookat>0 0 0 1 1 1 0 0 1</c:lookat><c:instance_geometry url="#F"/><c:instance_geometry url="#Spline"/></c:node>
    </c:visual_scene>
  </c:library_visual_scenes>
  <c:scene><c:instance_visual_scene url="#Main"/></c:scene>
</c:COLLADA>
"""


def _same_mesh(c, py):
    assert c["object_names"] == py["object_names"] and c["n_objects"] == len(py["object_names"])
    assert np.array_equal(c["faces"], py["faces"]) and np.array_equal(c["face_object_id"], py["face_object_id"])
    # both transform in f64 and round once to f32; numpy's matmul may fuse or reorder the 3-term sums: an ulp of f64
    # before the rounding, i.e. at most one f32 ulp, and only on a rounding boundary
    assert np.allclose(c["verts"], py["verts"], rtol=2e-7, atol=1e-7)
    assert np.mean(c["verts"] == py["verts"]) > 0.99


def test_rr_load_mesh_file_collada_equals_meshio(tmp_path):
    """COLLADA in C (csrc/rr_collada.cpp; the reference's default map is a .dae, launch/mro_husky.launch:4) against
    meshio.load_dae: the hand-written scene of test_meshio (polylist + triangles, nested nodes, unit), a scene with every
    primitive kind, several inputs per corner, an accessor with stride 4 / offset 5, two material groups in one geometry,
    <library_nodes> instantiated twice, a DOCTYPE, a CDATA array, a namespace prefix and entities in a name; and a
    multi-object scene written by save_dae."""
    import sys
    from test_meshio import DAE
    from common import GOLDEN
    p = tmp_path / "s.dae"
    p.write_text(DAE)
    c = native.load_mesh_file(p)
    _same_mesh(c, meshio.load_mesh(str(p)))
    assert c["object_names"] == ["Door", "Wall", "Door"] and c["face_object_id"].tolist() == [0, 1, 1, 2]
    r = tmp_path / "rich.dae"
    r.write_text(DAE_RICH)
    c, py = native.load_mesh_file(r), meshio.load_mesh(str(r))
    _same_mesh(c, py)
    # depth-first: A's own geometry (two groups), then its two instance_nodes (Prop: F, then child S) -- before A1 --, then A1, then B
    assert c["object_names"] == ['Hall & "Annex"[brick]', 'Hall & "Annex"[glass]', "F", "S", "F", "S", "S", "F"]
    assert np.bincount(c["face_object_id"]).tolist() == [2, 3, 5, 3, 5, 3, 3, 5]
    sys.path.insert(0, GOLDEN)
    import gen_oracle_images as gen
    s = gen.two_room_scene()
    q = str(tmp_path / "rooms.dae")
    meshio.save_dae(q, s["verts"], s["faces"], s["face_object_id"])
    c = native.load_mesh_file(q)
    _same_mesh(c, meshio.load_mesh(q))
    order = np.argsort(s["face_object_id"], kind="stable")
    assert np.array_equal(c["verts"][c["faces"]], s["verts"][s["faces"][order]])
    assert np.array_equal(c["face_object_id"], s["face_object_id"][order])


def test_rr_load_mesh_file_collada_errors(tmp_path):
    from test_meshio import DAE

    def load(text, name="x.dae"):
        p = tmp_path / name
        p.write_text(text)
        return native.load_mesh_file(p)
    with pytest.raises(native.RRError, match="not a COLLADA"):
        load("<html/>")
    with pytest.raises(native.RRError, match="not well-formed"):
        load(DAE[:len(DAE) // 2])
    with pytest.raises(native.RRError, match="not well-formed"):
        load(DAE.replace("</mesh></geometry>", "</mesh></geometri>", 1))
    with pytest.raises(native.RRError, match="no visual scene"):
        load(DAE[:DAE.index("<library_visual_scenes>")] + "</COLLADA>")
    with pytest.raises(native.RRError, match="out of range"):
        load(DAE.replace("<p>0 1 2</p>", "<p>0 1 3</p>"))
    with pytest.raises(native.RRError, match="out of range"):
        load(DAE.replace("<p>0 1 2</p>", "<p>0 -1 2</p>"))
    with pytest.raises(native.RRError, match="malformed <translate>"):
        load(DAE.replace("<translate>10 0 0</translate>", "<translate>10 0</translate>"))
    with pytest.raises(native.RRError, match="no triangle geometry"):
        load(DAE.replace('url="#Door-mesh"', 'url="#Nothing"').replace('url="#Wall-mesh"', 'url="#Nothing"'))
    # an instance_node that reaches itself: refused, not followed for ever
    cyc = DAE.replace("<library_visual_scenes>", '<library_nodes><node id="L"><instance_geometry url="#Door-mesh"/>'
                      '<instance_node url="#L"/><instance_node url="#L"/></node></library_nodes><library_visual_scenes>')
    cyc = cyc.replace('<node id="Door" name="Door">', '<node id="Door" name="Door"><instance_node url="#L"/>')
    with pytest.raises(native.RRError, match="too deep|too large|more triangles"):
        load(cyc)
    with pytest.raises(ValueError):
        meshio.load_mesh(str(tmp_path / "x.dae"))


def test_host_side_under_asan(tmp_path):
    """csrc/rr_host.cpp + rr_collada.cpp under AddressSanitizer + UBSan (CPU; no GPU code): the sampler's unit vectors and
    argument checks, the four well-formed files (PLY ascii / binary, OBJ, COLLADA), and 8,000 damaged ones (truncated, bytes flipped, digits inserted into counts and
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
                    os.path.join(root, "radarays_ros_amd", "csrc", "rr_host.cpp"),
                    os.path.join(root, "radarays_ros_amd", "csrc", "rr_collada.cpp"), "-o", exe], check=True)
    work = tmp_path / "files"
    work.mkdir()
    r = subprocess.run([exe, str(work)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
    assert "all: ok" in r.stdout
