"""Mesh ingest for rr_set_mesh -- the data format on the caller's side of the path.

The reference loads its map with `rm::import_embree_map(map_file)` (assimp behind rmagine,
src/radar_simulator.cpp:149): MulRan maps are `.ply` (launch/mulran_sim.launch:7), the ORU
scenes multi-object `.dae` whose object index drives `object_materials`
(config/oru4_test.yaml:37-56).  Here: PLY (ascii / binary_little_endian / binary_big_endian)
and Wavefront OBJ (objects `o`/`g` -> object ids) into the flat arrays of the C ABI:
verts float32 [nv][3], faces uint32 [nf][3], face_object_id uint32 [nf].  Polygons are
fan-triangulated.  COLLADA (`.dae`): geometries instantiated by the visual scene, node
transforms applied, one object id per instantiated primitive group in depth-first scene order --
for a Blender export that is the order of the `<name>-mesh` comments of
config/oru4_test.yaml:37-56.  assimp and rmagine are absent from this image, so the object
numbering of the `.dae` path is NOT pinned against them: check `object_names` against your
`object_materials` list.
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


def _dae_local(tag):
    return tag.rsplit("}", 1)[-1]


def _dae_node_matrix(node):
    """Product of the node's transform elements in document order (COLLADA 1.4 §5: column vectors)."""
    m = np.eye(4)
    for e in node:
        k = _dae_local(e.tag)
        if k not in ("matrix", "translate", "rotate", "scale"):
            continue
        v = [float(x) for x in (e.text or "").split()]
        t = np.eye(4)
        if k == "matrix" and len(v) == 16:
            t = np.array(v, float).reshape(4, 4)            # row-major in the file
        elif k == "translate" and len(v) == 3:
            t[:3, 3] = v
        elif k == "scale" and len(v) == 3:
            t[0, 0], t[1, 1], t[2, 2] = v
        elif k == "rotate" and len(v) == 4:
            ax = np.array(v[:3], float)
            n = np.linalg.norm(ax)
            if n > 0:
                x, y, z = ax / n
                a = np.deg2rad(v[3])
                c, s_, C_ = np.cos(a), np.sin(a), 1 - np.cos(a)
                t[:3, :3] = [[c + x * x * C_, x * y * C_ - z * s_, x * z * C_ + y * s_],
                             [y * x * C_ + z * s_, c + y * y * C_, y * z * C_ - x * s_],
                             [z * x * C_ - y * s_, z * y * C_ + x * s_, c + z * z * C_]]
        else:
            raise ValueError("COLLADA: malformed <%s>" % k)
        m = m @ t
    return m


def load_dae(path, apply_unit=True, apply_up_axis=False):
    """COLLADA 1.4/1.5 triangle geometry.  `apply_unit`: scale by <unit meter=...> (assimp does);
    `apply_up_axis`: rotate X_UP / Z_UP scenes to Y_UP as assimp does by default -- off here because the
    radar works in the map frame the file was modelled in (Blender exports Z_UP)."""
    import xml.etree.ElementTree as ET
    try:
        root = ET.parse(path).getroot()
    except ET.ParseError as e:
        raise ValueError("%s: not well-formed XML (%s)" % (path, e))
    if _dae_local(root.tag) != "COLLADA":
        raise ValueError("%s: not a COLLADA document" % path)

    def kids(e, name):
        return [c for c in e if _dae_local(c.tag) == name]

    def first(e, name):
        r = kids(e, name)
        return r[0] if r else None

    unit, up = 1.0, "Y_UP"
    asset = first(root, "asset")
    if asset is not None:
        u = first(asset, "unit")
        if u is not None and u.get("meter"):
            unit = float(u.get("meter"))
        a = first(asset, "up_axis")
        if a is not None and a.text:
            up = a.text.strip()

    # ---- geometries: id -> (name, [primitive groups: (material, [n][3] float64 positions per corner)])
    geoms = {}
    for lib in kids(root, "library_geometries"):
        for g in kids(lib, "geometry"):
            mesh = first(g, "mesh")
            if mesh is None:
                continue
            sources = {}
            for src in kids(mesh, "source"):
                fa = first(src, "float_array")
                if fa is None:
                    continue
                data = np.array((fa.text or "").split(), float)
                stride, offset = 3, 0
                tc = first(src, "technique_common")
                acc = first(tc, "accessor") if tc is not None else None
                if acc is not None:
                    stride, offset = int(acc.get("stride", "1")), int(acc.get("offset", "0"))
                sources["#" + src.get("id")] = (data, stride, offset)
            vert_pos = {}
            for vs in kids(mesh, "vertices"):
                for inp in kids(vs, "input"):
                    if inp.get("semantic") == "POSITION":
                        vert_pos["#" + vs.get("id")] = inp.get("source")
            groups = []
            for prim in mesh:
                kind = _dae_local(prim.tag)
                if kind not in ("triangles", "polylist", "polygons", "trifans", "tristrips"):
                    continue
                inputs = kids(prim, "input")
                n_off = 1 + max([int(i.get("offset", "0")) for i in inputs] or [0])
                vin = [i for i in inputs if i.get("semantic") == "VERTEX"]
                if not vin:
                    continue
                v_off = int(vin[0].get("offset", "0"))
                src_id = vert_pos.get(vin[0].get("source"))
                if src_id not in sources:
                    raise ValueError("%s: geometry %s has no POSITION source" % (path, g.get("id")))
                data, stride, offset = sources[src_id]
                if stride < 3:
                    raise ValueError("%s: POSITION stride < 3 in %s" % (path, g.get("id")))

                def corner_ids(p_text):
                    a = np.array((p_text or "").split(), np.int64)
                    return a.reshape(-1, n_off)[:, v_off]
                polys = []
                ps = kids(prim, "p")
                if kind == "triangles":
                    ids = np.concatenate([corner_ids(x.text) for x in ps]) if ps else np.zeros(0, np.int64)
                    polys = ids[: len(ids) // 3 * 3].reshape(-1, 3).tolist()
                elif kind == "polylist":
                    vc = first(prim, "vcount")
                    counts = np.array((vc.text or "").split(), np.int64) if vc is not None else np.zeros(0, np.int64)
                    ids = corner_ids(ps[0].text) if ps else np.zeros(0, np.int64)
                    k = 0
                    for c in counts:
                        polys += _fan(ids[k:k + c])
                        k += c
                elif kind == "polygons":
                    for x in ps:
                        polys += _fan(corner_ids(x.text))
                elif kind == "trifans":
                    for x in ps:
                        polys += _fan(corner_ids(x.text))
                elif kind == "tristrips":
                    for x in ps:
                        c = corner_ids(x.text)
                        for k in range(len(c) - 2):
                            polys.append([c[k], c[k + 1], c[k + 2]] if k % 2 == 0 else [c[k + 1], c[k], c[k + 2]])
                if not polys:
                    continue
                tri = np.array(polys, np.int64)
                pos = np.stack([data[offset + stride * tri + a] for a in range(3)], axis=-1)    # [n][3 corners][xyz]
                groups.append((prim.get("material"), pos))
            geoms["#" + g.get("id")] = (g.get("name") or g.get("id"), groups)

    lib_nodes = {}
    for lib in kids(root, "library_nodes"):
        for n in lib.iter():
            if _dae_local(n.tag) == "node" and n.get("id"):
                lib_nodes["#" + n.get("id")] = n

    verts, faces, obj, names = [], [], [], []

    def walk(node, parent, depth=0):
        if depth > 64:
            raise ValueError("%s: node hierarchy too deep (cyclic instance_node?)" % path)
        m = parent @ _dae_node_matrix(node)
        for ig in kids(node, "instance_geometry"):
            name, groups = geoms.get(ig.get("url"), (None, []))
            for mat, pos in groups:
                p = pos.reshape(-1, 3) @ m[:3, :3].T + m[:3, 3]
                base = sum(len(v) for v in verts)
                verts.append(p)
                faces.append(base + np.arange(len(p)).reshape(-1, 3))
                obj.append(np.full(len(p) // 3, len(names), np.uint32))
                names.append(name if len(groups) == 1 else "%s[%s]" % (name, mat))
        for inn in kids(node, "instance_node"):
            tgt = lib_nodes.get(inn.get("url"))
            if tgt is not None:
                walk(tgt, m, depth + 1)
        for c in kids(node, "node"):
            walk(c, m, depth + 1)

    top = np.eye(4)
    if apply_unit:
        top[:3, :3] *= unit
    if apply_up_axis and up == "Z_UP":
        top = top @ np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1.0]])
    elif apply_up_axis and up == "X_UP":
        top = top @ np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    scenes_ = {}
    for lib in kids(root, "library_visual_scenes"):
        for vs in kids(lib, "visual_scene"):
            scenes_["#" + (vs.get("id") or "")] = vs
    chosen = None
    sc = first(root, "scene")
    if sc is not None:
        ivs = first(sc, "instance_visual_scene")
        if ivs is not None:
            chosen = scenes_.get(ivs.get("url"))
    if chosen is None and scenes_:
        chosen = next(iter(scenes_.values()))
    if chosen is None:
        raise ValueError("%s: no visual scene" % path)
    for n in kids(chosen, "node"):
        walk(n, top)
    if not verts:
        raise ValueError("%s: the visual scene instantiates no triangle geometry" % path)
    return {"verts": np.concatenate(verts).astype(np.float32), "faces": np.concatenate(faces).astype(np.uint32),
            "face_object_id": np.concatenate(obj), "object_names": names, "unit_meter": unit, "up_axis": up}


def load_mesh(path):
    p = path.lower()
    if p.endswith(".ply"):
        return load_ply(path)
    if p.endswith(".obj"):
        return load_obj(path)
    if p.endswith(".dae"):
        return load_dae(path)
    raise ValueError("unsupported mesh format: %s (PLY, OBJ and COLLADA are read)" % path)


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


def save_dae(path, verts, faces, face_object_id=None, object_names=None):
    """Minimal COLLADA writer (one geometry + one scene node per object id, Z_UP, metres): exports the
    synthetic multi-object scenes in the format the reference's ORU maps use."""
    verts = np.asarray(verts, np.float32).reshape(-1, 3)
    faces = np.asarray(faces, np.uint32).reshape(-1, 3)
    fo = np.zeros(len(faces), np.uint32) if face_object_id is None else np.asarray(face_object_id, np.uint32)
    ids = sorted(set(int(x) for x in fo))
    if ids != list(range(len(ids))):
        raise ValueError("save_dae: object ids must be 0..n-1 without gaps (they become scene order)")
    names = list(object_names) if object_names else ["Object%d" % i for i in ids]
    geo, nodes = [], []
    for i in ids:
        f = faces[fo == i]
        used, inv = np.unique(f.ravel(), return_inverse=True)
        pos = " ".join(repr(float(x)) for x in verts[used].ravel())
        idx = " ".join(str(int(x)) for x in inv)
        g = "%s-mesh" % names[i]
        geo.append('<geometry id="%s" name="%s"><mesh><source id="%s-pos"><float_array id="%s-pos-a" count="%d">%s</float_array>'
                   '<technique_common><accessor source="#%s-pos-a" count="%d" stride="3"><param name="X" type="float"/>'
                   '<param name="Y" type="float"/><param name="Z" type="float"/></accessor></technique_common></source>'
                   '<vertices id="%s-v"><input semantic="POSITION" source="#%s-pos"/></vertices>'
                   '<triangles count="%d"><input semantic="VERTEX" source="#%s-v" offset="0"/><p>%s</p></triangles></mesh></geometry>'
                   % (g, names[i], g, g, 3 * len(used), pos, g, len(used), g, g, len(f), g, idx))
        nodes.append('<node id="%s" name="%s" type="NODE"><matrix sid="transform">1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1</matrix>'
                     '<instance_geometry url="#%s"/></node>' % (names[i], names[i], g))
    with open(path, "w") as f:
        f.write('<?xml version="1.0" encoding="utf-8"?>\n<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">\n'
                '<asset><unit name="meter" meter="1"/><up_axis>Z_UP</up_axis></asset>\n<library_geometries>\n%s\n</library_geometries>\n'
                '<library_visual_scenes><visual_scene id="Scene" name="Scene">\n%s\n</visual_scene></library_visual_scenes>\n'
                '<scene><instance_visual_scene url="#Scene"/></scene>\n</COLLADA>\n' % ("\n".join(geo), "\n".join(nodes)))
