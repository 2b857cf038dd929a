"""Host logic: mesh ingest (PLY / OBJ / COLLADA) into the flat arrays rr_set_mesh takes."""
import numpy as np
import pytest

from radarays_ros_amd import meshio, scenes


@pytest.mark.parametrize("binary", [True, False])
def test_ply_round_trip(tmp_path, binary):
    s = scenes.heightfield_room(8, n_buildings=3)
    p = str(tmp_path / "m.ply")
    meshio.save_ply(p, s["verts"], s["faces"], binary=binary)
    m = meshio.load_mesh(p)
    assert np.array_equal(m["faces"], s["faces"])
    assert np.array_equal(m["verts"], s["verts"]) if binary else np.allclose(m["verts"], s["verts"], rtol=0, atol=0)
    assert m["face_object_id"].shape == (len(s["faces"]),) and not m["face_object_id"].any()


def test_ply_polygons_are_fan_triangulated_and_big_endian(tmp_path):
    v = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 1.5, 0]])
    p = tmp_path / "q.ply"
    hdr = ("ply\nformat binary_big_endian 1.0\nelement vertex 5\nproperty double x\nproperty double y\nproperty double z\n"
           "property uchar red\nelement face 2\nproperty list uchar uint vertex_index\nend_header\n")
    body = b""
    for r in v:
        body += np.array(r, ">f8").tobytes() + b"\x07"
    body += b"\x04" + np.array([0, 1, 2, 3], ">u4").tobytes() + b"\x03" + np.array([3, 2, 4], ">u4").tobytes()
    p.write_bytes(hdr.encode() + body)
    m = meshio.load_ply(str(p))
    assert np.array_equal(m["verts"], v)
    assert m["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [3, 2, 4]]


def test_obj_objects_become_object_ids(tmp_path):
    p = tmp_path / "s.obj"
    p.write_text("# two objects\no ground\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n"
                 "o wall\nv 0 0 1\nv 1 0 1\nv 1 1 1\nf 5/1/1 6/2/1 7/3/1\nf -3 -2 -1\n")
    m = meshio.load_mesh(str(p))
    assert m["object_names"] == ["ground", "wall"]
    assert m["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 5, 6]]
    assert m["face_object_id"].tolist() == [0, 0, 1, 1]


def test_bad_files(tmp_path):
    p = tmp_path / "x.ply"
    p.write_text("nope\n")
    with pytest.raises(ValueError):
        meshio.load_mesh(str(p))
    with pytest.raises(ValueError):
        meshio.load_mesh("scene.stl")


DAE = """<?xml version="1.0" encoding="utf-8"?>
<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
  <asset><unit name="centimeter" meter="0.5"/><up_axis>Z_UP</up_axis></asset>
  <library_geometries>
    <geometry id="Wall-mesh" name="Wall"><mesh>
      <source id="Wall-pos"><float_array id="Wall-pos-a" count="12">0 0 0 1 0 0 1 1 0 0 1 0</float_array>
        <technique_common><accessor source="#Wall-pos-a" count="4" stride="3"/></technique_common></source>
      <source id="Wall-nrm"><float_array id="Wall-nrm-a" count="3">0 0 1</float_array>
        <technique_common><accessor source="#Wall-nrm-a" count="1" stride="3"/></technique_common></source>
      <vertices id="Wall-v"><input semantic="POSITION" source="#Wall-pos"/></vertices>
      <polylist count="1" material="m0"><input semantic="VERTEX" source="#Wall-v" offset="0"/>
        <input semantic="NORMAL" source="#Wall-nrm" offset="1"/><vcount>4</vcount><p>0 0 1 0 2 0 3 0</p></polylist>
    </mesh></geometry>
    <geometry id="Door-mesh" name="Door"><mesh>
      <source id="Door-pos"><float_array id="Door-pos-a" count="9">0 0 0 2 0 0 0 2 0</float_array>
        <technique_common><accessor source="#Door-pos-a" count="3" stride="3"/></technique_common></source>
      <vertices id="Door-v"><input semantic="POSITION" source="#Door-pos"/></vertices>
      <triangles count="1"><input semantic="VERTEX" source="#Door-v" offset="0"/><p>0 1 2</p></triangles>
    </mesh></geometry>
  </library_geometries>
  <library_visual_scenes><visual_scene id="Scene">
    <node id="Door" name="Door"><translate>10 0 0</translate><rotate>0 0 1 90</rotate>
      <instance_geometry url="#Door-mesh"/></node>
    <node id="Wall" name="Wall"><matrix>1 0 0 0  0 1 0 5  0 0 1 0  0 0 0 1</matrix>
      <instance_geometry url="#Wall-mesh"/>
      <node id="Door2"><scale>2 2 2</scale><instance_geometry url="#Door-mesh"/></node></node>
  </visual_scene></library_visual_scenes>
  <scene><instance_visual_scene url="#Scene"/></scene>
</COLLADA>
"""


def test_collada_scene_order_transforms_and_units(tmp_path):
    p = tmp_path / "s.dae"
    p.write_text(DAE)
    m = meshio.load_mesh(str(p))
    # depth-first scene order: Door (node 1), Wall, Door again under Wall -> object ids 0, 1, 2
    assert m["object_names"] == ["Door", "Wall", "Door"] and m["unit_meter"] == 0.5 and m["up_axis"] == "Z_UP"
    assert m["faces"].shape == (4, 3) and list(m["face_object_id"]) == [0, 1, 1, 2]
    v = m["verts"].reshape(-1, 3, 3)
    # Door: translate(10,0,0) * rotZ(90) applied to (0,0,0),(2,0,0),(0,2,0), then the 0.5 unit scale
    assert np.allclose(v[0], 0.5 * np.array([[10, 0, 0], [10, 2, 0], [8, 0, 0]]), atol=1e-6)
    # Wall quad fan-triangulated, moved by +5 in y
    assert np.allclose(v[1], 0.5 * np.array([[0, 5, 0], [1, 5, 0], [1, 6, 0]]), atol=1e-6)
    assert np.allclose(v[2], 0.5 * np.array([[0, 5, 0], [1, 6, 0], [0, 6, 0]]), atol=1e-6)
    # nested node: parent matrix * scale(2)
    assert np.allclose(v[3], 0.5 * np.array([[0, 5, 0], [4, 5, 0], [0, 9, 0]]), atol=1e-6)
    m2 = meshio.load_dae(str(p), apply_unit=False, apply_up_axis=True)
    assert np.allclose(m2["verts"].reshape(-1, 3, 3)[1], np.array([[0, 0, -5], [1, 0, -5], [1, 0, -6]]), atol=1e-6)


def test_collada_rejects_garbage(tmp_path):
    p = tmp_path / "x.dae"
    p.write_text("<html/>")
    with pytest.raises(ValueError):
        meshio.load_mesh(str(p))


def test_collada_round_trip_of_a_multi_object_scene(tmp_path):
    import sys
    from common import GOLDEN
    sys.path.insert(0, GOLDEN)
    import gen_oracle_images as gen
    s = gen.two_room_scene()
    p = str(tmp_path / "rooms.dae")
    meshio.save_dae(p, s["verts"], s["faces"], s["face_object_id"])
    m = meshio.load_mesh(p)
    assert len(m["object_names"]) == int(s["face_object_id"].max()) + 1
    # same triangles (corner positions), grouped by object in scene order
    order = np.argsort(s["face_object_id"], kind="stable")
    want = s["verts"][s["faces"][order]]
    got = m["verts"][m["faces"]]
    assert np.array_equal(got, want)
    assert np.array_equal(m["face_object_id"], s["face_object_id"][order])
