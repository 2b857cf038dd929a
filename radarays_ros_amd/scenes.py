"""Deterministic synthetic scenes -- the reference ships no mesh
(its launch files point at author-local files, launch/mulran_sim.launch:7).

All scenes are CLOSED (every ray hits), so no azimuth column is empty.
Heights use Ken Perlin's improved noise (the same public permutation table the
reference carries in include/radarays_ros/image_algorithms.h:14-50), evaluated
here in numpy.
"""
import numpy as np

_PERM = np.array([
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30,
    69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62,
    94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136,
    171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122,
    60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161,
    1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86,
    164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126,
    255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253,
    19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193,
    238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31,
    181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93,
    222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180], dtype=np.int64)
_P = np.concatenate([_PERM, _PERM])


def perlin2(x, y):
    """Improved Perlin noise at z = 0, vectorised, float64."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    fx, fy = np.floor(x), np.floor(y)
    X = fx.astype(np.int64) & 255
    Y = fy.astype(np.int64) & 255
    x = x - fx
    y = y - fy
    z = np.zeros_like(x)

    def fade(t):
        return t * t * t * (t * (t * 6 - 15) + 10)

    def lerp(t, a, b):
        return a + t * (b - a)

    def grad(h, x, y, z):
        h = h & 15
        u = np.where(h < 8, x, y)
        v = np.where(h < 4, y, np.where((h == 12) | (h == 14), x, z))
        return np.where((h & 1) == 0, u, -u) + np.where((h & 2) == 0, v, -v)

    u, v = fade(x), fade(y)
    A = _P[X] + Y
    AA = _P[A]
    AB = _P[A + 1]
    B = _P[X + 1] + Y
    BA = _P[B]
    BB = _P[B + 1]
    return lerp(v, lerp(u, grad(_P[AA], x, y, z), grad(_P[BA], x - 1, y, z)),
                lerp(u, grad(_P[AB], x, y - 1, z), grad(_P[BB], x - 1, y - 1, z)))


def _box_tris(lo, hi, vbase=0):
    lo = np.asarray(lo, np.float32)
    hi = np.asarray(hi, np.float32)
    v = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3], [4, 6, 5], [4, 7, 6], [0, 5, 1], [0, 4, 5],
                  [1, 6, 2], [1, 5, 6], [2, 7, 3], [2, 6, 7], [3, 4, 0], [3, 7, 4]], np.uint32) + np.uint32(vbase)
    return v, f


def box12():
    """closed box [-10,10]x[-8,8]x[-1,3], 12 triangles, one object."""
    v, f = _box_tris([-10, -8, -1], [10, 8, 3])
    return {"verts": v, "faces": f, "face_object_id": np.zeros(len(f), np.uint32),
            "object_materials": [1], "name": "box12"}


def ground_height(x, y, amp=3.0, freq=0.02):
    return amp * perlin2(freq * np.asarray(x, np.float64), freq * np.asarray(y, np.float64))


def heightfield_room(n_quads, extent=420.0, z_lo=-6.0, z_hi=24.0, n_buildings=0, seed=3,
                     keep_clear=(1.0, 1.5, 12.0)):
    """n_quads x n_quads terrain (2 n^2 triangles) over extent x extent metres inside a
    closed room (12 triangles) + n_buildings axis-aligned boxes (12 triangles each,
    object 1).  Terrain + room are object 0."""
    half = extent / 2.0
    g = np.linspace(-half, half, n_quads + 1)
    X, Y = np.meshgrid(g, g, indexing="xy")
    Z = ground_height(X, Y)
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    n1 = n_quads + 1
    i, j = np.meshgrid(np.arange(n_quads), np.arange(n_quads), indexing="xy")
    v00 = (j * n1 + i).ravel()
    v10 = v00 + 1
    v01 = v00 + n1
    v11 = v01 + 1
    faces = np.concatenate([np.stack([v00, v10, v11], -1), np.stack([v00, v11, v01], -1)], 0).astype(np.uint32)
    rv, rf = _box_tris([-half, -half, z_lo], [half, half, z_hi], vbase=len(verts))
    verts = np.concatenate([verts, rv], 0)
    faces = np.concatenate([faces, rf], 0)
    obj = np.zeros(len(faces), np.uint32)
    if n_buildings > 0:
        rs = np.random.RandomState(seed)
        bv, bf = [], []
        vb = len(verts)
        made = 0
        while made < n_buildings:
            cx, cy = rs.uniform(-half + 15, half - 15, 2)
            sx, sy = rs.uniform(4.0, 18.0, 2)
            h = rs.uniform(4.0, 18.0)
            if abs(cx - keep_clear[0]) < keep_clear[2] + sx / 2 and abs(cy - keep_clear[1]) < keep_clear[2] + sy / 2:
                continue
            z0 = float(ground_height(cx, cy)) - 1.5
            v, f = _box_tris([cx - sx / 2, cy - sy / 2, z0], [cx + sx / 2, cy + sy / 2, z0 + h + 1.5], vbase=vb)
            bv.append(v)
            bf.append(f)
            vb += 8
            made += 1
        verts = np.concatenate([verts] + bv, 0)
        faces = np.concatenate([faces] + bf, 0)
        obj = np.concatenate([obj, np.ones(12 * n_buildings, np.uint32)])
    return {"verts": verts.astype(np.float32), "faces": faces.astype(np.uint32), "face_object_id": obj,
            "object_materials": [1, 2] if n_buildings > 0 else [1],
            "name": "heightfield%d_b%d" % (n_quads, n_buildings)}


# The 18 objects config/oru4_test.yaml:37-56 lists for the ORU4 scene (the .dae itself is author-local,
# launch/mro_husky.launch:4): their names, in the order the reference's object_materials table indexes them
ORU4_OBJECT_NAMES = ["HallwayGround", "DoorHallway1Glass", "HallwayWall", "DoorT1203Wood", "DoorT1203Glass", "DoorT1210Glass",
                     "DoorT1210Wood", "DoorFikaWood", "DoorFikaGlass", "DoorLabGlass", "Bookshelf2", "Bookshelf", "Locker",
                     "Workbench2", "Locker2", "Workbench", "Trash", "Building"]


def oru4_like_scene():
    """A stand-in for the reference's ORU4 office scene with the SAME 18 objects in the same order, so that
    config/oru4_test.yaml's object_materials table (5 materials; glass v = 0.03 refracts) applies as it is: a 24 x 3 m
    hallway inside a closed building shell, wall segments with door openings, wood doors with glass panes, glass doors,
    shelves, lockers, workbenches, a trash bin.  Every object is a few axis-aligned boxes (the real mesh is not in the
    repository); the sensor stands at (1.0, 1.5, 0.2 + ...) in free space."""
    parts = {n: [] for n in ORU4_OBJECT_NAMES}

    def box(name, lo, hi):
        parts[name].append((lo, hi))
    box("Building", (-14.0, -9.0, -1.0), (14.0, 9.0, 4.0))                 # closed shell: every ray ends somewhere
    box("HallwayGround", (-13.5, -8.5, -0.6), (13.5, 8.5, -0.5))
    # the hallway runs along x at y in [0, 3]; its two walls have door openings
    for x0, x1 in ((-12.0, -7.0), (-5.8, -1.0), (0.2, 4.0), (5.2, 12.0)):
        box("HallwayWall", (x0, 3.0, -0.5), (x1, 3.2, 2.6))
    for x0, x1 in ((-12.0, -4.0), (-2.8, 6.0), (7.2, 12.0)):
        box("HallwayWall", (x0, -0.2, -0.5), (x1, 0.0, 2.6))
    box("HallwayWall", (-12.2, -0.2, -0.5), (-12.0, 3.2, 2.6))
    # doors in the openings of the far wall (y = 3.0 .. 3.2): wood leaf with a glass pane above the handle
    box("DoorT1203Wood", (-7.0, 3.04, -0.5), (-5.8, 3.12, 0.9));  box("DoorT1203Glass", (-7.0, 3.06, 0.9), (-5.8, 3.10, 2.1))
    box("DoorT1210Wood", (-1.0, 3.04, -0.5), (0.2, 3.12, 0.9));   box("DoorT1210Glass", (-1.0, 3.06, 0.9), (0.2, 3.10, 2.1))
    box("DoorFikaWood", (4.0, 3.04, -0.5), (5.2, 3.12, 0.9));     box("DoorFikaGlass", (4.0, 3.06, 0.9), (5.2, 3.10, 2.1))
    # all-glass doors: the near wall's openings and the end of the hallway
    box("DoorLabGlass", (-4.0, -0.13, -0.5), (-2.8, -0.07, 2.1))
    box("DoorHallway1Glass", (12.0, 0.0, -0.5), (12.06, 3.0, 2.3))
    # furniture in the rooms behind the walls and along the hallway
    box("Bookshelf", (-11.5, 6.0, -0.5), (-8.5, 6.4, 1.7));       box("Bookshelf2", (-3.0, 7.2, -0.5), (0.5, 7.6, 1.9))
    box("Locker", (6.4, 0.05, -0.5), (7.0, 0.5, 1.4));            box("Locker2", (8.0, 5.0, -0.5), (8.6, 6.2, 1.4))
    box("Workbench", (2.0, 5.0, -0.5), (4.5, 5.9, 0.4));          box("Workbench2", (-9.0, -6.0, -0.5), (-6.0, -5.0, 0.4))
    box("Trash", (3.1, 0.1, -0.5), (3.5, 0.5, 0.1))
    verts, faces, obj = [], [], []
    vb = 0
    for oid, name in enumerate(ORU4_OBJECT_NAMES):
        assert parts[name], name
        for lo, hi in parts[name]:
            v, f = _box_tris(lo, hi, vbase=vb)
            verts.append(v); faces.append(f); obj.append(np.full(len(f), oid, np.uint32))
            vb += 8
    return {"verts": np.concatenate(verts).astype(np.float32), "faces": np.concatenate(faces).astype(np.uint32),
            "face_object_id": np.concatenate(obj), "object_names": list(ORU4_OBJECT_NAMES), "name": "box12_oru4_like"}


# BASELINE.json configs -> scene recipes
def config_scene(config_id):
    if config_id == 1:
        return box12()
    if config_id == 2:
        return heightfield_room(224)                         # 100,352 + 12 triangles
    if config_id == 3:
        return heightfield_room(708, n_buildings=2000)       # 1,002,528 + 12 + 24,000
    if config_id == 4:
        return heightfield_room(2237, n_buildings=20000)     # 10,008,338 + 12 + 240,000
    if config_id == 5:
        # the config-4 mesh with PER-TRIANGLE material ids (8 materials, seed 5)
        return per_triangle_materials(heightfield_room(2237, n_buildings=20000), 8, seed=5)
    raise ValueError("unknown config %r" % (config_id,))


def per_triangle_materials(scene, n_materials=8, seed=5):
    """Every face becomes its own 'object' class: face_object_id = random id in [0, n_materials),
    object k -> material k + 1 (material 0 stays air).  This is how the reference expresses
    per-triangle materials: object_materials[] is indexed by the id the ray cast returns
    (RadarCPU.cpp:266-271), and the mesh import decides what an object is."""
    rs = np.random.RandomState(seed)
    out = dict(scene)
    out["face_object_id"] = rs.randint(0, n_materials, len(scene["faces"])).astype(np.uint32)
    out["object_materials"] = list(range(1, n_materials + 1))
    out["name"] = scene["name"] + "_pertri%d" % n_materials
    return out


def yaw_pose(x, y, z, yaw):
    """Tsm as (qx, qy, qz, qw, tx, ty, tz)."""
    return np.array([0.0, 0.0, np.sin(yaw / 2.0), np.cos(yaw / 2.0), x, y, z], np.float32)


def default_pose(scene_name="", height=2.0):
    """t = (1.0, 1.5, .), yaw 0.3 (cf. launch/mro_husky.launch:15,23)."""
    if scene_name.startswith("box12"):
        return yaw_pose(1.0, 1.5, 0.2, 0.3)
    return yaw_pose(1.0, 1.5, float(ground_height(1.0, 1.5)) + height, 0.3)


def trajectory(n=16, scene_name="", radius=8.0, height=2.0):
    """n poses on a circle around (1.0, 1.5) for timing runs."""
    out = []
    for k in range(n):
        a = 2.0 * np.pi * k / n
        x, y = 1.0 + radius * np.cos(a), 1.5 + radius * np.sin(a)
        if scene_name.startswith("box12"):
            x, y = 1.0 + 0.4 * radius * np.cos(a) * 0.5, 1.5 + 0.4 * radius * np.sin(a) * 0.5
            z = 0.2
        else:
            z = float(ground_height(x, y)) + height
        out.append(yaw_pose(x, y, z, 0.3 + a))
    return out
