"""Mesh ingest for rr_set_mesh -- the data format on the caller's side of the path.

The reference loads its map with `rm::import_embree_map(map_file)` (assimp behind rmagine,
src/radar_simulator.cpp:149): MulRan maps are `.ply` (launch/mulran_sim.launch:7), the ORU
scenes multi-object `.dae` whose object index drives `object_materials`
(config/oru4_test.yaml:37-56).  Here: PLY (ascii / binary_little_endian / binary_big_endian)
and Wavefront OBJ (objects `o`/`g` -> object ids) into the flat arrays of the C ABI:
verts float32 [nv][3], faces uint32 [nf][3], face_object_id uint32 [nf].  Polygons are
fan-triangulated.  (COLLADA is not read: convert with any mesh tool.)
"""
import numpy as np

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
    "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
    "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


def _fan(idx):
    idx = list(idx)
    return [[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)]


def load_ply(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            t = line.decode("ascii", "replace").split()
            if not t or t[0] == "comment" or t[0] == "obj_info":
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                elements.append({"name": t[1], "count": int(t[2]), "props": []})
            elif t[0] == "property":
                if t[1] == "list":
                    elements[-1]["props"].append(("list", t[2], t[3], t[4]))
                else:
                    elements[-1]["props"].append(("scalar", t[1], t[2]))
            elif t[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError("%s: unsupported PLY format %r" % (path, fmt))
        verts, faces = None, []
        if fmt == "ascii":
            tokens = f.read().split()
            pos = 0
            for el in elements:
                rows = []
                for _ in range(el["count"]):
                    row = {}
                    for p in el["props"]:
                        if p[0] == "scalar":
                            row[p[2]] = float(tokens[pos]); pos += 1
                        else:
                            n = int(tokens[pos]); pos += 1
                            row[p[3]] = [int(x) for x in tokens[pos:pos + n]]; pos += n
                    rows.append(row)
                if el["name"] == "vertex":
                    verts = np.array([[r["x"], r["y"], r["z"]] for r in rows], np.float32).reshape(-1, 3)
                elif el["name"] == "face":
                    key = "vertex_indices" if rows and "vertex_indices" in rows[0] else "vertex_index"
                    for r in rows:
                        faces.extend(_fan(r[key]))
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            for el in elements:
                scalar_only = all(p[0] == "scalar" for p in el["props"])
                if scalar_only:
                    dt = np.dtype([(p[2], end + _PLY_TYPES[p[1]]) for p in el["props"]])
                    arr = np.frombuffer(f.read(dt.itemsize * el["count"]), dt, el["count"])
                    if el["name"] == "vertex":
                        verts = np.stack([arr["x"], arr["y"], arr["z"]], -1).astype(np.float32)
                else:
                    for _ in range(el["count"]):
                        row = {}
                        for p in el["props"]:
                            if p[0] == "scalar":
                                dt = np.dtype(end + _PLY_TYPES[p[1]])
                                row[p[2]] = np.frombuffer(f.read(dt.itemsize), dt, 1)[0]
                            else:
                                ct = np.dtype(end + _PLY_TYPES[p[1]])
                                n = int(np.frombuffer(f.read(ct.itemsize), ct, 1)[0])
                                it = np.dtype(end + _PLY_TYPES[p[2]])
                                row[p[3]] = np.frombuffer(f.read(it.itemsize * n), it, n).astype(np.int64)
                        if el["name"] == "face":
                            key = "vertex_indices" if "vertex_indices" in row else "vertex_index"
                            faces.extend(_fan(row[key]))
    if verts is None:
        raise ValueError("%s: no vertex element" % path)
    faces = np.array(faces, np.uint32).reshape(-1, 3)
    return {"verts": verts, "faces": faces, "face_object_id": np.zeros(len(faces), np.uint32)}


def load_obj(path):
    verts, faces, obj = [], [], []
    cur, names = -1, []
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v" and len(t) >= 4:
                verts.append([float(t[1]), float(t[2]), float(t[3])])
            elif t[0] in ("o", "g"):
                names.append(" ".join(t[1:]))
                cur = len(names) - 1
            elif t[0] == "f" and len(t) >= 4:
                idx = []
                for tok in t[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for tri in _fan(idx):
                    faces.append(tri)
                    obj.append(max(cur, 0))
    return {"verts": np.array(verts, np.float32).reshape(-1, 3), "faces": np.array(faces, np.uint32).reshape(-1, 3),
            "face_object_id": np.array(obj, np.uint32), "object_names": names}


def load_mesh(path):
    p = path.lower()
    if p.endswith(".ply"):
        return load_ply(path)
    if p.endswith(".obj"):
        return load_obj(path)
    raise ValueError("unsupported mesh format: %s (PLY and OBJ are read)" % path)


def save_ply(path, verts, faces, binary=True):
    """Writer used by the tests (round trip) and to export the synthetic scenes."""
    verts = np.asarray(verts, np.float32).reshape(-1, 3)
    faces = np.asarray(faces, np.uint32).reshape(-1, 3)
    hdr = ("ply\nformat %s 1.0\ncomment radarays_ros_amd\nelement vertex %d\nproperty float x\nproperty float y\n"
           "property float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
           % ("binary_little_endian" if binary else "ascii", len(verts), len(faces)))
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        if binary:
            f.write(verts.astype("<f4").tobytes())
            rec = np.zeros(len(faces), np.dtype([("n", "u1"), ("i", "<i4", 3)]))
            rec["n"] = 3
            rec["i"] = faces.astype(np.int32)
            f.write(rec.tobytes())
        else:
            for v in verts:
                f.write(("%r %r %r\n" % (float(v[0]), float(v[1]), float(v[2]))).encode())
            for t in faces:
                f.write(("3 %d %d %d\n" % tuple(int(x) for x in t)).encode())
