"""Host logic: mesh ingest (PLY / OBJ) into the flat arrays rr_set_mesh takes."""
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
        meshio.load_mesh("scene.dae")
