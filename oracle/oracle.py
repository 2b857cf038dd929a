"""ctypes binding of the CPU oracle (oracle/libradarays_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (radarays_ros_amd) never
imports this module.

The config object passed in is duck-typed: any object exposing the
RadarModelConfig field names of cfg/RadarModel.cfg (reference) works, e.g.
radarays_ros_amd.params.RadarModelConfig.
"""
import ctypes as C  # noqa: F401 (re-exported for tests)
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


class OrcMaterial(C.Structure):
    _fields_ = [("velocity", C.c_float), ("ambient", C.c_float),
                ("diffuse", C.c_float), ("specular", C.c_float)]


class OrcConfig(C.Structure):
    _fields_ = [
        ("n_cells", C.c_int32), ("n_angles", C.c_int32), ("n_reflections", C.c_int32),
        ("signal_denoising", C.c_int32),
        ("signal_denoising_triangular_width", C.c_int32),
        ("signal_denoising_gaussian_width", C.c_int32),
        ("signal_denoising_mb_width", C.c_int32),
        ("ambient_noise", C.c_int32), ("scroll_image", C.c_int32),
        ("record_multi_reflection", C.c_int32), ("record_multi_path", C.c_int32),
        ("material_id_air", C.c_int32),
        ("resolution", C.c_double), ("energy_max", C.c_double), ("signal_max", C.c_double),
        ("signal_denoising_triangular_mode", C.c_double),
        ("signal_denoising_gaussian_mode", C.c_double),
        ("signal_denoising_mb_mode", C.c_double),
        ("ambient_noise_at_signal_0", C.c_double), ("ambient_noise_at_signal_1", C.c_double),
        ("ambient_noise_energy_max", C.c_double), ("ambient_noise_energy_min", C.c_double),
        ("ambient_noise_energy_loss", C.c_double), ("multipath_threshold", C.c_double),
        ("wave_energy_threshold", C.c_float), ("theta_min", C.c_float),
        ("theta_inc", C.c_float), ("brdf_model", C.c_int32),
    ]


class OrcStats(C.Structure):
    _fields_ = [("wave_passes", C.c_uint64), ("hits", C.c_uint64), ("signals", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("seconds", C.c_double), ("near_threshold", C.c_uint64)]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libradarays_oracle.so")
    src = os.path.join(_HERE, "radarays_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "libradarays_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/include/radarays_ros/radar_math.h"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("RADARAYS_ORACLE_LIB") or os.path.join(_HERE, "libradarays_oracle.so")   # override: `make asan`
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_intersect.restype = C.c_int
    L.orc_intersect.argtypes = [C.c_void_p, fp, fp, fp, C.POINTER(C.c_uint32), fp]
    L.orc_simulate.restype = C.c_int
    L.orc_simulate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                               C.POINTER(OrcConfig), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(OrcStats)]
    L.orc_simulate_motion.restype = C.c_int
    L.orc_simulate_motion.argtypes = L.orc_simulate.argtypes
    L.orc_fresnel.argtypes = [fp, fp, C.c_double, C.c_double, C.c_double, C.c_double,
                              fp, C.POINTER(C.c_double), fp, C.POINTER(C.c_double)]
    L.orc_back_reflection_shader.restype = C.c_float
    L.orc_ct_lobe.restype = C.c_float
    L.orc_ct_lobe.argtypes = [C.c_float, C.c_float]
    L.orc_back_reflection_shader.argtypes = [C.c_float] * 5
    L.orc_incidence_angle.restype = C.c_double
    L.orc_incidence_angle.argtypes = [fp, fp]
    L.orc_make_denoiser.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, fp]
    L.orc_perlin_noise.restype = C.c_double
    L.orc_perlin_noise.argtypes = [C.c_double] * 3
    L.orc_perlin_noise_hilo.restype = C.c_double
    L.orc_perlin_noise_hilo.argtypes = [C.c_double] * 7
    L.orc_erfinvf.restype = C.c_float
    L.orc_erfinvf.argtypes = [C.c_float]
    L.orc_quantile.restype = C.c_float
    L.orc_quantile.argtypes = [C.c_float]
    L.orc_wave_move.argtypes = [fp, fp, C.POINTER(C.c_double), C.c_double, C.c_double]
    L.orc_sample_cone_local.argtypes = [C.c_float, C.c_int, C.c_int, C.c_float, fp, fp, fp]
    L.orc_noise_amplitude.restype = C.c_float
    L.orc_noise_amplitude.argtypes = [C.c_float, C.c_float, C.c_double, C.c_double]
    L.orc_cone_radius.restype = C.c_float
    L.orc_cone_radius.argtypes = [C.c_float, C.c_int, C.c_float, C.c_float]
    L.orc_saturate_u8.restype = C.c_uint8
    L.orc_saturate_u8.argtypes = [C.c_float]
    _LIB = L
    return L


def ref_lib():
    """oracle/_ref/libradarays_refmath.so (reference radar_math.h compiled as-is) or None."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libradarays_refmath.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        for n in ("ref_erfinvf", "ref_quantile"):
            getattr(R, n).restype = C.c_float
            getattr(R, n).argtypes = [C.c_float]
        _REF = R
    return _REF


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


def fresnel(normal, direction, energy=1.0, polarization=0.5, v1=0.3, v2=0.0):
    L = lib()
    rd, td = (C.c_float * 3)(), (C.c_float * 3)()
    re, te = C.c_double(), C.c_double()
    L.orc_fresnel(_f3(normal), _f3(direction), energy, polarization, v1, v2, rd, C.byref(re), td, C.byref(te))
    return np.array(rd[:], np.float32), re.value, np.array(td[:], np.float32), te.value


def ct_lobe(angle, specular_exp):
    return float(lib().orc_ct_lobe(angle, specular_exp))


def back_reflection_shader(angle, energy, a, b, c):
    return float(lib().orc_back_reflection_shader(angle, energy, a, b, c))


def make_denoiser(kind, width, mode, rescale=False):
    out = (C.c_float * max(width, 1))()
    lib().orc_make_denoiser(kind, width, mode, int(rescale), out)
    return np.array(out[:width], np.float32)


def perlin_noise(x, y, z=0.0):
    return float(lib().orc_perlin_noise(x, y, z))


def perlin_noise_hilo(off_x, off_y, x, y, scale_low, scale_high, p_low):
    return float(lib().orc_perlin_noise_hilo(off_x, off_y, x, y, scale_low, scale_high, p_low))


def erfinvf(a):
    return float(lib().orc_erfinvf(a))


def quantile(p):
    return float(lib().orc_quantile(p))


def sample_cone_local(width_rad, sample_dist, p_in_cone, u_angle, r_variate):
    u = np.ascontiguousarray(u_angle, np.float32)
    r = np.ascontiguousarray(r_variate, np.float32)
    out = np.zeros((len(u), 3), np.float32)
    fp = C.POINTER(C.c_float)
    lib().orc_sample_cone_local(width_rad, len(u), sample_dist, p_in_cone,
                                u.ctypes.data_as(fp), r.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out


def noise_amplitude(signal, max_val, at_signal_0, at_signal_1):
    return float(lib().orc_noise_amplitude(signal, max_val, at_signal_0, at_signal_1))


def cone_radius(width, sample_dist, p_in_cone, variate):
    return float(lib().orc_cone_radius(width, sample_dist, p_in_cone, variate))


def saturate_u8(x):
    return int(lib().orc_saturate_u8(x))


def make_config(cfg, n_angles=400, material_id_air=0, wave_energy_threshold=0.001, brdf_model=0):
    """RadarModelConfig-like object -> OrcConfig (Radar.cpp:22-32 constants)."""
    c = OrcConfig()
    c.n_cells = int(cfg.n_cells)
    c.n_angles = int(n_angles)
    c.n_reflections = int(cfg.n_reflections)
    c.signal_denoising = int(cfg.signal_denoising)
    c.signal_denoising_triangular_width = int(cfg.signal_denoising_triangular_width)
    c.signal_denoising_gaussian_width = int(cfg.signal_denoising_gaussian_width)
    c.signal_denoising_mb_width = int(cfg.signal_denoising_mb_width)
    c.ambient_noise = int(cfg.ambient_noise)
    c.scroll_image = int(cfg.scroll_image)
    c.record_multi_reflection = int(bool(cfg.record_multi_reflection))
    c.record_multi_path = int(bool(cfg.record_multi_path))
    c.material_id_air = int(material_id_air)
    for k in ("resolution", "energy_max", "signal_max", "signal_denoising_triangular_mode",
              "signal_denoising_gaussian_mode", "signal_denoising_mb_mode",
              "ambient_noise_at_signal_0", "ambient_noise_at_signal_1",
              "ambient_noise_energy_max", "ambient_noise_energy_min",
              "ambient_noise_energy_loss", "multipath_threshold"):
        setattr(c, k, float(getattr(cfg, k)))
    c.wave_energy_threshold = float(np.float32(wave_energy_threshold))
    c.theta_min = 0.0
    c.theta_inc = float(np.float32(-(2.0 * np.pi) / n_angles))
    c.brdf_model = int(brdf_model)
    return c


class Scene:
    def __init__(self, verts, faces, face_object_id=None, use_bvh=-1):
        self.verts = np.ascontiguousarray(verts, np.float32)
        self.faces = np.ascontiguousarray(faces, np.uint32)
        self.obj = None if face_object_id is None else np.ascontiguousarray(face_object_id, np.uint32)
        self._h = lib().orc_scene_create(
            self.verts.ctypes.data, len(self.verts), self.faces.ctypes.data, len(self.faces),
            None if self.obj is None else self.obj.ctypes.data, use_bvh)
        if not self._h:
            raise MemoryError("orc_scene_create failed")

    def __del__(self):
        try:
            if self._h:
                lib().orc_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def intersect(self, orig, direction):
        t, tri, ng = C.c_float(), C.c_uint32(), (C.c_float * 3)()
        hit = lib().orc_intersect(self._h, _f3(orig), _f3(direction), C.byref(t), C.byref(tri), ng)
        if not hit:
            return None
        return t.value, tri.value, np.array(ng[:], np.float32)


def simulate(scene, materials, object_materials, cfg, beam_dirs, pose, noise_rnd=None,
             az_begin=0, az_end=None, n_angles=400, material_id_air=0, want_f32=True,
             n_threads=0, brdf_model=0):
    """RadarCPU::simulate on the oracle. materials: [(velocity, ambient, diffuse, specular)].
    Returns (u8 [n_cells][n_angles], f32 or None, stats dict)."""
    L = lib()
    oc = make_config(cfg, n_angles, material_id_air, brdf_model=brdf_model)
    if az_end is None:
        az_end = n_angles
    mats = (OrcMaterial * len(materials))(*[OrcMaterial(*[float(x) for x in m]) for m in materials])
    om = np.ascontiguousarray(object_materials, np.int32)
    bd = np.ascontiguousarray(beam_dirs, np.float32)
    ps = np.ascontiguousarray(pose, np.float32)
    motion = ps.ndim == 2          # include_motion: one pose per azimuth
    assert ps.shape == ((n_angles, 7) if motion else (7,))
    nr = None if noise_rnd is None else np.ascontiguousarray(noise_rnd, np.float32)
    u8 = np.zeros((oc.n_cells, n_angles), np.uint8)
    f32 = np.zeros((oc.n_cells, n_angles), np.float32) if want_f32 else None
    st = OrcStats()
    rc = (L.orc_simulate_motion if motion else L.orc_simulate)(scene._h, mats, len(materials), om.ctypes.data, len(om), C.byref(oc),
                        bd.ctypes.data, len(bd), ps.ctypes.data,
                        None if nr is None else nr.ctypes.data,
                        az_begin, az_end, u8.ctypes.data,
                        None if f32 is None else f32.ctypes.data, n_threads, C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_simulate failed: %d" % rc)
    stats = {k: getattr(st, k) for k, _ in OrcStats._fields_}
    return u8, f32, stats
