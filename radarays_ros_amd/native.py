"""ctypes binding of libradarays_mi355.so (include/radarays_mi355.h).

This is the ONLY compute path of the package: if the library is missing or no
HIP device is usable, everything here raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RADARAYS_MI355_LIB") or os.path.join(_HERE, "libradarays_mi355.so")
_LIB = None

SYMBOLS = [
    "rr_abi_version", "rr_default_config", "rr_create", "rr_destroy", "rr_last_error",
    "rr_set_mesh", "rr_set_mesh_gpu", "rr_copy_mesh", "rr_set_materials", "rr_set_config", "rr_set_beam_samples",
    "rr_set_noise_offsets", "rr_set_motion_poses", "rr_simulate", "rr_simulate_columns_device", "rr_simulate_batch_columns_device",
    "rr_simulate_batch_columns_carry_device",
    "rr_assemble_image_device", "rr_assemble_blocks_device", "rr_assemble_frames_device", "rr_simulate_device",
    "rr_simulate_material_sets_device", "rr_simulate_material_sets", "rr_simulate_batch_device", "rr_synchronize", "rr_get_stats",
    "rr_set_stats_mode", "rr_debug_trace", "rr_debug_fresnel", "rr_debug_brdf", "rr_get_bvh_info", "rr_get_trace_grid", "rr_get_graph_stats", "rr_set_timing_mode",
    "rr_get_kernel_time", "rr_get_kernel_samples", "rr_reserve_timing_events",
    "rr_simulate_batch_host_async", "rr_wait_host", "rr_host_alloc", "rr_host_free", "rr_copy_to_host_async", "rr_deliver_to_host_async", "rr_host_delivery_route", "rr_partition", "rr_multi_plan",
    "rr_create_multi", "rr_destroy_multi", "rr_multi_last_error", "rr_multi_device_count", "rr_multi_rccl_version", "rr_multi_ctx",
    "rr_multi_set_mesh", "rr_multi_set_mesh_gpu", "rr_multi_set_materials", "rr_multi_set_config", "rr_multi_set_beam_samples",
    "rr_multi_set_noise_offsets", "rr_multi_set_motion_poses", "rr_multi_simulate", "rr_multi_simulate_batch",
    "rr_multi_simulate_batch_async", "rr_multi_wait", "rr_peek_error_bits_async", "rr_get_traversal_shape",
    "rr_cone_dirs", "rr_sample_cone_local", "rr_load_mesh_file", "rr_mesh_reorder_objects", "rr_free_mesh",
    "rr_simulate_param_sets_device", "rr_simulate_param_sets", "rr_score_images_device",
]


class RRMaterial(C.Structure):
    _fields_ = [("velocity", C.c_float), ("ambient", C.c_float),
                ("diffuse", C.c_float), ("specular", C.c_float)]


class RRConfig(C.Structure):
    _fields_ = [
        ("n_cells", C.c_int32), ("n_angles", C.c_int32), ("n_reflections", C.c_int32),
        ("signal_denoising", C.c_int32),
        ("signal_denoising_triangular_width", C.c_int32),
        ("signal_denoising_gaussian_width", C.c_int32),
        ("signal_denoising_mb_width", C.c_int32),
        ("ambient_noise", C.c_int32), ("scroll_image", C.c_int32),
        ("record_multi_reflection", C.c_int32), ("record_multi_path", C.c_int32),
        ("max_waves_per_azimuth", C.c_int32), ("brdf_model", C.c_int32), ("reserved_", C.c_int32),
        ("resolution", C.c_double), ("energy_max", C.c_double), ("signal_max", C.c_double),
        ("signal_denoising_triangular_mode", C.c_double),
        ("signal_denoising_gaussian_mode", C.c_double),
        ("signal_denoising_mb_mode", C.c_double),
        ("ambient_noise_at_signal_0", C.c_double), ("ambient_noise_at_signal_1", C.c_double),
        ("ambient_noise_energy_max", C.c_double), ("ambient_noise_energy_min", C.c_double),
        ("ambient_noise_energy_loss", C.c_double), ("multipath_threshold", C.c_double),
        ("wave_energy_threshold", C.c_float), ("theta_min", C.c_float),
        ("theta_inc", C.c_float), ("range_max", C.c_float),
    ]


class RRParamSet(C.Structure):
    _fields_ = [("materials", C.c_void_p), ("beam_dirs", C.c_void_p), ("n_reflections", C.c_int32), ("reserved_", C.c_int32)]


class RRMesh(C.Structure):
    _fields_ = [("verts", C.POINTER(C.c_float)), ("n_verts", C.c_size_t), ("faces", C.POINTER(C.c_uint32)), ("n_faces", C.c_size_t),
                ("face_object_id", C.POINTER(C.c_uint32)), ("n_objects", C.c_size_t), ("object_names", C.POINTER(C.c_char_p))]


class RRStats(C.Structure):
    _fields_ = [("wave_passes", C.c_uint64), ("hits", C.c_uint64), ("signals", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("overflow", C.c_uint32), ("pad_", C.c_uint32)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad_"}


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the in-tree shared library."""
    src = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src, f) for f in os.listdir(src)] + [os.path.join(_HERE, "..", "include", "radarays_mi355.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(f) > os.path.getmtime(LIB_PATH) for f in srcs)
    if force or stale:
        subprocess.run(["make", "-C", src], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "radarays_ros_amd: %s is missing -- build it with __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        # share torch's HIP runtime when torch is in the process: both resolve the
        # SONAME libamdhip64.so.7, the first one loaded wins
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rr_abi_version.restype = C.c_int
    L.rr_default_config.argtypes = [C.POINTER(RRConfig)]
    L.rr_create.restype = vp
    L.rr_create.argtypes = [C.c_int]
    L.rr_destroy.argtypes = [vp]
    L.rr_last_error.restype = C.c_char_p
    L.rr_last_error.argtypes = [vp]
    L.rr_set_mesh.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.rr_set_mesh_gpu.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.rr_copy_mesh.argtypes = [vp, vp]
    L.rr_set_materials.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int32]
    L.rr_set_config.argtypes = [vp, C.POINTER(RRConfig)]
    L.rr_set_beam_samples.argtypes = [vp, vp, C.c_size_t]
    L.rr_set_noise_offsets.argtypes = [vp, vp, C.c_size_t]
    L.rr_set_motion_poses.argtypes = [vp, vp, C.c_size_t]
    L.rr_simulate.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(RRStats)]
    L.rr_simulate_columns_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.rr_simulate_batch_columns_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.rr_simulate_batch_columns_carry_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_size_t]
    L.rr_assemble_image_device.argtypes = [vp, vp, vp, vp]
    L.rr_assemble_blocks_device.argtypes = [vp, vp, C.c_int, C.c_size_t, vp, vp]
    L.rr_assemble_frames_device.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_int, C.c_size_t, vp, vp]
    L.rr_simulate_device.argtypes = [vp, vp, vp, vp]
    L.rr_simulate_material_sets_device.argtypes = [vp, vp, vp, C.c_int, C.c_size_t, vp, vp]
    L.rr_simulate_material_sets.argtypes = [vp, vp, vp, C.c_int, C.c_size_t, vp]
    L.rr_simulate_batch_device.argtypes = [vp, vp, C.c_int, vp, vp]
    L.rr_synchronize.argtypes = [vp, vp]
    L.rr_get_stats.argtypes = [vp, C.POINTER(RRStats)]
    L.rr_set_stats_mode.argtypes = [vp, C.c_int]
    L.rr_set_timing_mode.argtypes = [vp, C.c_int]
    L.rr_get_kernel_time.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
    L.rr_debug_trace.argtypes = [vp, vp, vp, C.c_size_t, vp, vp]
    L.rr_debug_fresnel.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rr_debug_brdf.argtypes = [vp, C.c_size_t, vp, C.c_int, vp]
    L.rr_get_trace_grid.argtypes = [vp, vp, vp, C.POINTER(C.c_uint64)]
    L.rr_get_graph_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rr_get_bvh_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.rr_get_kernel_samples.argtypes = [vp, C.c_char_p, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rr_reserve_timing_events.argtypes = [vp, C.c_size_t]
    L.rr_simulate_batch_host_async.argtypes = [vp, vp, C.c_int, vp, vp]
    L.rr_wait_host.argtypes = [vp, vp]
    L.rr_copy_to_host_async.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.rr_deliver_to_host_async.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.rr_host_delivery_route.argtypes = [vp]
    L.rr_host_alloc.restype = vp
    L.rr_host_alloc.argtypes = [C.c_size_t]
    L.rr_host_free.argtypes = [vp]
    L.rr_host_free.restype = None
    L.rr_partition.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rr_partition.restype = None
    L.rr_multi_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), vp, vp, vp]
    L.rr_create_multi.restype = vp
    L.rr_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.rr_destroy_multi.argtypes = [vp]
    L.rr_destroy_multi.restype = None
    L.rr_multi_last_error.restype = C.c_char_p
    L.rr_multi_last_error.argtypes = [vp]
    L.rr_multi_device_count.argtypes = [vp]
    L.rr_multi_rccl_version.argtypes = [vp]
    L.rr_multi_ctx.restype = vp
    L.rr_multi_ctx.argtypes = [vp, C.c_int]
    L.rr_multi_set_mesh.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.rr_multi_set_mesh_gpu.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.rr_multi_set_materials.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int32]
    L.rr_multi_set_config.argtypes = [vp, C.POINTER(RRConfig)]
    L.rr_multi_set_beam_samples.argtypes = [vp, vp, C.c_size_t]
    L.rr_multi_set_noise_offsets.argtypes = [vp, vp, C.c_size_t]
    L.rr_multi_set_motion_poses.argtypes = [vp, vp, C.c_size_t]
    L.rr_multi_simulate.argtypes = [vp, vp, vp]
    L.rr_multi_simulate_batch.argtypes = [vp, vp, C.c_int, vp]
    L.rr_multi_simulate_batch_async.argtypes = [vp, vp, C.c_int, vp]
    L.rr_multi_wait.argtypes = [vp, vp]
    L.rr_peek_error_bits_async.argtypes = [vp, vp, vp]
    L.rr_get_traversal_shape.argtypes = [vp, vp]
    L.rr_simulate_param_sets_device.argtypes = [vp, vp, C.POINTER(RRParamSet), C.c_int, C.c_size_t, vp, vp]
    L.rr_simulate_param_sets.argtypes = [vp, vp, C.POINTER(RRParamSet), C.c_int, C.c_size_t, vp, vp, vp]
    L.rr_score_images_device.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
    L.rr_cone_dirs.argtypes = [C.c_float, C.c_int, C.c_float, vp, vp, C.c_size_t, vp]
    L.rr_sample_cone_local.argtypes = [C.c_uint32, C.c_float, C.c_size_t, C.c_int, C.c_float, vp]
    L.rr_load_mesh_file.argtypes = [C.c_char_p, C.POINTER(RRMesh), C.c_char_p, C.c_size_t]
    L.rr_free_mesh.argtypes = [C.POINTER(RRMesh)]
    L.rr_mesh_reorder_objects.argtypes = [C.POINTER(RRMesh), C.POINTER(C.c_char_p), C.c_size_t, C.c_char_p, C.c_size_t]
    L.rr_free_mesh.restype = None
    for n in SYMBOLS:
        getattr(L, n)
    _LIB = L
    return L


def make_config(cfg, n_angles=400, max_waves_per_azimuth=0, wave_energy_threshold=0.001,
                ray_range_max=1000.0, brdf_model=0):
    """RadarModelConfig (params.py) -> rr_config."""
    c = RRConfig()
    lib().rr_default_config(C.byref(c))
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
    c.max_waves_per_azimuth = int(max_waves_per_azimuth)
    c.brdf_model = int(brdf_model)
    for k in ("resolution", "energy_max", "signal_max", "signal_denoising_triangular_mode",
              "signal_denoising_gaussian_mode", "signal_denoising_mb_mode",
              "ambient_noise_at_signal_0", "ambient_noise_at_signal_1",
              "ambient_noise_energy_max", "ambient_noise_energy_min",
              "ambient_noise_energy_loss", "multipath_threshold"):
        setattr(c, k, float(getattr(cfg, k)))
    c.wave_energy_threshold = float(np.float32(wave_energy_threshold))
    c.theta_min = 0.0
    c.theta_inc = float(np.float32(-(2.0 * np.pi) / n_angles))   # Radar.cpp:27
    c.range_max = float(ray_range_max)                            # radar_algorithms.cpp:158
    return c


class RRError(RuntimeError):
    pass


class Context:
    """Owns one rr_ctx (one GPU)."""

    def __init__(self, device=0):
        self._L = lib()
        self._h = self._L.rr_create(int(device))
        if not self._h:
            raise RRError(self._L.rr_last_error(None).decode())
        self.device = int(device)
        self.cfg = None
        self.n_angles = 400

    def close(self):
        if getattr(self, "_h", None):
            self._L.rr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RRError("%s (rc=%d)" % (self._L.rr_last_error(self._h).decode(), rc))

    def set_mesh(self, verts, faces, face_object_id=None, builder="host"):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(faces, np.uint32).reshape(-1, 3)
        o = None if face_object_id is None else np.ascontiguousarray(face_object_id, np.uint32)
        if o is not None and len(o) != len(f):
            raise ValueError("face_object_id must have one entry per face")
        fn = {"host": self._L.rr_set_mesh, "gpu": self._L.rr_set_mesh_gpu}[builder]
        self._ck(fn(self._h, v.ctypes.data, len(v), f.ctypes.data, len(f), None if o is None else o.ctypes.data))

    def copy_mesh(self, src):
        """take the finished tree of another context (same or another device): rr_copy_mesh"""
        self._ck(self._L.rr_copy_mesh(self._h, src._h))

    def set_materials(self, materials, object_materials, material_id_air=0):
        m = (RRMaterial * len(materials))(*[RRMaterial(*[float(x) for x in (t.astuple() if hasattr(t, "astuple") else t)])
                                            for t in materials])
        om = np.ascontiguousarray(object_materials, np.int32)
        self._ck(self._L.rr_set_materials(self._h, m, len(materials), om.ctypes.data, len(om), int(material_id_air)))

    def set_config(self, cfg, n_angles=400, **kw):
        self.cfg = cfg
        self.n_angles = n_angles
        self._rrcfg = make_config(cfg, n_angles, **kw)
        self._ck(self._L.rr_set_config(self._h, C.byref(self._rrcfg)))

    def set_beam_samples(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        self._ck(self._L.rr_set_beam_samples(self._h, d.ctypes.data, len(d)))

    def set_noise_offsets(self, rnd):
        """[n_angles] offsets, or [k][n_angles] (flat or 2-D): frame f of a batch uses row f % k."""
        r = np.ascontiguousarray(rnd, np.float32).ravel()
        self._ck(self._L.rr_set_noise_offsets(self._h, r.ctypes.data, r.size))

    def set_motion_poses(self, poses):
        """include_motion: [n_angles][7] per-azimuth poses, or [k][n_angles][7] (frame f of a batch uses table f % k);
        None/empty switches it off."""
        if poses is None or len(poses) == 0:
            self._ck(self._L.rr_set_motion_poses(self._h, None, 0))
            return
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        self._ck(self._L.rr_set_motion_poses(self._h, p.ctypes.data, len(p)))

    def simulate(self, pose, az_begin=0, az_end=None, want_f32=False):
        """Host-buffer path (rr_simulate). Returns (u8 [n_cells][n_angles], f32|None, stats)."""
        p = np.ascontiguousarray(pose, np.float32)
        assert p.shape == (7,)
        if az_end is None:
            az_end = self.n_angles
        n_cells = self.cfg.n_cells if self.cfg is not None else 1   # unconfigured: the library reports it
        u8 = np.zeros((n_cells, self.n_angles), np.uint8)
        f32 = np.zeros((n_cells, self.n_angles), np.float32) if want_f32 else None
        st = RRStats()
        self._ck(self._L.rr_simulate(self._h, p.ctypes.data, az_begin, az_end, u8.ctypes.data,
                                     None if f32 is None else f32.ctypes.data, C.byref(st)))
        return u8, f32, st.asdict()

    def simulate_into(self, pose, out_u8):
        """rr_simulate into a preallocated uint8 [n_cells][n_angles] array, no statistics: the call a latency probe times."""
        p = np.ascontiguousarray(pose, np.float32)
        self._ck(self._L.rr_simulate(self._h, p.ctypes.data, 0, self.n_angles, out_u8.ctypes.data, None, None))

    def traversal_shape(self):
        """stats mode: wave-level loop shape of the last frame (rr_get_traversal_shape)."""
        a = np.zeros(8, np.uint64)
        self._ck(self._L.rr_get_traversal_shape(self._h, a.ctypes.data))
        return dict(zip(("waves", "iterations", "node_path_issues", "leaf_path_issues", "live_quad_steps", "max_iterations",
                         "node_steps", "leaf_steps"), (int(x) for x in a)))

    def simulate_columns_device(self, pose, az_begin, az_end, d_cols_u8_ptr, d_cols_f32_ptr=None, stream=None):
        p = np.ascontiguousarray(pose, np.float32)
        self._ck(self._L.rr_simulate_columns_device(self._h, p.ctypes.data, az_begin, az_end,
                                                    d_cols_u8_ptr, d_cols_f32_ptr, stream))

    def simulate_batch_device(self, poses, d_imgs_ptr, stream=None):
        """Whole frames of up to 64 (RR_MAX_BATCH) poses in one set of launches on `stream`: images [n][n_cells][n_angles] in HBM."""
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        self._ck(self._L.rr_simulate_batch_device(self._h, p.ctypes.data, len(p), d_imgs_ptr, stream))

    def simulate_batch_host_async(self, poses, h_imgs_ptr, stream=None):
        """Whole frames delivered to (page-locked) host memory [n][n_cells][n_angles]; complete after wait_host()."""
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        self._ck(self._L.rr_simulate_batch_host_async(self._h, p.ctypes.data, len(p), h_imgs_ptr, stream))

    def wait_host(self, h_imgs_ptr=None):
        self._ck(self._L.rr_wait_host(self._h, h_imgs_ptr))

    def copy_to_host_async(self, d_src_ptr, h_dst_ptr, nbytes, stream=None):
        """device -> host on `stream` by the library's own copy kernel (page-locked destination) -- rr_copy_to_host_async"""
        self._ck(lib().rr_copy_to_host_async(self._h, C.c_void_p(d_src_ptr), C.c_void_p(h_dst_ptr), C.c_size_t(nbytes), C.c_void_p(stream)))

    def host_delivery_route(self):
        """rr_host_delivery_route: "sdma" (ROCr's SDMA path in use), "sdma (untried)" or "stream copies" (deferred / trickled / copy kernel)"""
        return {2: "sdma", 1: "sdma (untried)", 0: "stream copies"}.get(int(lib().rr_host_delivery_route(self._h)), "?")

    def deliver_to_host_async(self, d_src_ptr, h_dst_ptr, nbytes, stream=None):
        """device -> page-locked host over SDMA once `stream` has got here; complete after wait_host(h_dst_ptr) -- rr_deliver_to_host_async"""
        self._ck(lib().rr_deliver_to_host_async(self._h, C.c_void_p(d_src_ptr), C.c_void_p(h_dst_ptr), C.c_size_t(nbytes), C.c_void_p(stream)))

    def simulate_batch_columns_device(self, poses, az_begin, az_end, d_cols_u8_ptr, stream=None):
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        self._ck(self._L.rr_simulate_batch_columns_device(self._h, p.ctypes.data, len(p), az_begin, az_end,
                                                          d_cols_u8_ptr, stream))

    def simulate_batch_columns_carry_device(self, poses, az_begin, az_end, d_cols_u8_ptr, stream, d_carry_src, h_carry_dst, carry_bytes):
        """rr_simulate_batch_columns_carry_device: the batch's later-pass trace launches carry a device -> host copy"""
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        self._ck(self._L.rr_simulate_batch_columns_carry_device(self._h, p.ctypes.data, len(p), az_begin, az_end, d_cols_u8_ptr, stream,
                                                                d_carry_src, h_carry_dst, int(carry_bytes)))

    def assemble_image_device(self, d_cols_u8_ptr, d_img_ptr, stream=None):
        self._ck(self._L.rr_assemble_image_device(self._h, d_cols_u8_ptr, d_img_ptr, stream))

    def assemble_blocks_device(self, d_cols_u8_ptr, n_loc, block_stride, d_img_ptr, stream=None):
        self._ck(self._L.rr_assemble_blocks_device(self._h, d_cols_u8_ptr, int(n_loc), int(block_stride), d_img_ptr, stream))

    def assemble_frames_device(self, d_cols_u8_ptr, n_loc, block_stride, n_frames, frame_stride, d_imgs_ptr, stream=None):
        self._ck(self._L.rr_assemble_frames_device(self._h, d_cols_u8_ptr, int(n_loc), int(block_stride), int(n_frames),
                                                   int(frame_stride), d_imgs_ptr, stream))

    @staticmethod
    def _material_sets(sets):
        a = np.ascontiguousarray(np.asarray(sets, dtype=np.float32))
        if a.ndim != 3 or a.shape[2] != 4:
            raise ValueError("material sets must have shape [n_sets][n_materials][4] (velocity, ambient, diffuse, specular)")
        return a

    def simulate_material_sets_device(self, pose, sets, d_imgs_ptr, stream=None):
        """n_sets material tables, one pose -> images [n_sets][n_cells][n_angles] in HBM."""
        a = self._material_sets(sets)
        p = np.ascontiguousarray(pose, dtype=np.float32)
        self._ck(self._L.rr_simulate_material_sets_device(self._h, p.ctypes.data, a.ctypes.data, a.shape[0], a.shape[1], d_imgs_ptr, stream))

    def simulate_material_sets(self, pose, sets):
        """Host-buffer variant: returns a uint8 array [n_sets][n_cells][n_angles]."""
        a = self._material_sets(sets)
        p = np.ascontiguousarray(pose, dtype=np.float32)
        n_cells = self.cfg.n_cells if self.cfg is not None else 1
        out = np.zeros((a.shape[0], n_cells, self.n_angles), dtype=np.uint8)
        self._ck(self._L.rr_simulate_material_sets(self._h, p.ctypes.data, a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data))
        return out

    @staticmethod
    def _param_sets(sets):
        """sets: iterable of dicts {"materials": [n_mat][4] or None, "beam_dirs": [n_beam][3] or None, "n_reflections": int or None}
        -> (ctypes array, the numpy arrays it points into, n_materials or None)"""
        keep, n_mat = [], None
        arr = (RRParamSet * len(sets))()
        for k, st in enumerate(sets):
            m = st.get("materials")
            if m is not None:
                m = np.ascontiguousarray(np.asarray([(t.astuple() if hasattr(t, "astuple") else t) for t in m], dtype=np.float32))
                if m.ndim != 2 or m.shape[1] != 4:
                    raise ValueError("materials of a set must have shape [n_materials][4]")
                n_mat = m.shape[0] if n_mat is None else n_mat
                keep.append(m)
                arr[k].materials = m.ctypes.data
            b = st.get("beam_dirs")
            if b is not None:
                b = np.ascontiguousarray(b, np.float32).reshape(-1, 3)
                keep.append(b)
                arr[k].beam_dirs = b.ctypes.data
            nr = st.get("n_reflections")
            arr[k].n_reflections = -1 if nr is None else int(nr)
        return arr, keep, n_mat

    def simulate_param_sets(self, pose, sets, n_materials, ref_u8=None, want_images=True):
        """rr_simulate_param_sets: one pose, n parameter sets (see _param_sets).  Returns (images uint8
        [n][n_cells][n_angles] or None, psnr float64 [n] or None against ref_u8)."""
        arr, keep, n_mat = self._param_sets(sets)
        p = np.ascontiguousarray(pose, dtype=np.float32)
        n_cells = self.cfg.n_cells if self.cfg is not None else 1
        out = np.zeros((len(sets), n_cells, self.n_angles), dtype=np.uint8) if want_images else None
        psnr = ref = None
        if ref_u8 is not None:
            ref = np.ascontiguousarray(ref_u8, np.uint8)
            if ref.shape != (n_cells, self.n_angles):
                raise ValueError("reference image must be [n_cells][n_angles] uint8")
            psnr = np.zeros(len(sets), np.float64)
        self._ck(self._L.rr_simulate_param_sets(self._h, p.ctypes.data, arr, len(sets), int(n_materials),
                                                None if out is None else out.ctypes.data,
                                                None if ref is None else ref.ctypes.data, None if psnr is None else psnr.ctypes.data))
        return out, psnr

    def simulate_param_sets_device(self, pose, sets, n_materials, d_imgs_ptr, stream=None):
        arr, keep, _ = self._param_sets(sets)
        p = np.ascontiguousarray(pose, dtype=np.float32)
        self._ck(self._L.rr_simulate_param_sets_device(self._h, p.ctypes.data, arr, len(sets), int(n_materials), d_imgs_ptr, stream))

    def score_images_device(self, d_imgs_ptr, n_images, d_ref_ptr, stream=None, want_sse=False):
        """rr_score_images_device -> psnr float64 [n] (and the exact sums of squared differences uint64 [n])."""
        psnr = np.zeros(n_images, np.float64); sse = np.zeros(n_images, np.uint64)
        self._ck(self._L.rr_score_images_device(self._h, d_imgs_ptr, int(n_images), d_ref_ptr, psnr.ctypes.data, sse.ctypes.data, stream))
        return (psnr, sse) if want_sse else psnr

    def simulate_device(self, pose, d_img_ptr, stream=None):
        p = np.ascontiguousarray(pose, np.float32)
        self._ck(self._L.rr_simulate_device(self._h, p.ctypes.data, d_img_ptr, stream))

    def synchronize(self, stream=None):
        self._ck(self._L.rr_synchronize(self._h, stream))

    def stats(self):
        st = RRStats()
        self._ck(self._L.rr_get_stats(self._h, C.byref(st)))
        return st.asdict()

    def set_stats_mode(self, on):
        self._ck(self._L.rr_set_stats_mode(self._h, int(bool(on))))

    def set_timing_mode(self, on):
        self._ck(self._L.rr_set_timing_mode(self._h, int(on)))

    def kernel_time(self, name, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        self._ck(self._L.rr_get_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def kernel_samples(self, name):
        """Every launch duration (ms) recorded for `name` since the last reset."""
        n = C.c_size_t()
        self._ck(self._L.rr_get_kernel_samples(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        if n.value:
            self._ck(self._L.rr_get_kernel_samples(self._h, name.encode(), out.ctypes.data, n.value, C.byref(n)))
        return out

    def reserve_timing_events(self, n):
        self._ck(self._L.rr_reserve_timing_events(self._h, int(n)))

    def trace_grid(self):
        """rr_get_trace_grid -> (rows[24] of the last call, hist[24], repaired 16-ray groups)"""
        rows = np.zeros(24, np.uint32); hist = np.zeros(24, np.uint32); rep = C.c_uint64()
        self._ck(self._L.rr_get_trace_grid(self._h, rows.ctypes.data, hist.ctypes.data, C.byref(rep)))
        return rows, hist, int(rep.value)

    def graph_stats(self):
        """rr_get_graph_stats -> (launch chains captured, replayed)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self._L.rr_get_graph_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def bvh_info(self):
        a, b, d, s = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        self._ck(self._L.rr_get_bvh_info(self._h, C.byref(a), C.byref(b), C.byref(d), C.byref(s)))
        return {"n_nodes": a.value, "n_tris": b.value, "depth": d.value, "stack_need": s.value}

    def debug_trace(self, origs, dirs):
        o = np.ascontiguousarray(origs, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        t = np.zeros(len(o), np.float32)
        f = np.zeros(len(o), np.uint32)
        self._ck(self._L.rr_debug_trace(self._h, o.ctypes.data, d.ctypes.data, len(o), t.ctypes.data, f.ctypes.data))
        return t, f

    def debug_fresnel(self, normals, dirs, energy, v1, v2):
        """rr_debug_fresnel: the kernels' fresnel_split on n inputs -> (refl_dir [n][3] f32, refl_energy [n] f64, refr_dir, refr_energy)"""
        nr = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(d)
        e = np.ascontiguousarray(np.broadcast_to(np.asarray(energy, np.float64), (n,)))
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(v1, np.float64), (n,)))
        b = np.ascontiguousarray(np.broadcast_to(np.asarray(v2, np.float32), (n,)))
        rd, td = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        re, te = np.zeros(n, np.float64), np.zeros(n, np.float64)
        self._ck(self._L.rr_debug_fresnel(self._h, n, nr.ctypes.data, d.ctypes.data, e.ctypes.data, a.ctypes.data, b.ctypes.data,
                                          rd.ctypes.data, re.ctypes.data, td.ctypes.data, te.ctypes.data))
        return rd, re, td, te

    def debug_brdf(self, angle, energy, ambient, diffuse, specular, brdf_model=0):
        """rr_debug_brdf: the kernels' back_reflection_shader on n inputs (arrays or scalars, broadcast) -> [n] f32"""
        cols = np.broadcast_arrays(*[np.asarray(x, np.float32) for x in (angle, energy, ambient, diffuse, specular)])
        x = np.ascontiguousarray(np.stack([c.reshape(-1) for c in cols], 1), np.float32)
        out = np.zeros(len(x), np.float32)
        self._ck(self._L.rr_debug_brdf(self._h, len(x), x.ctypes.data, int(brdf_model), out.ctypes.data))
        return out


class HostImages:
    """Page-locked host memory for images (rr_host_alloc), viewed as a numpy array."""

    def __init__(self, shape):
        self._L = lib()
        self.shape = tuple(int(x) for x in shape)
        n = int(np.prod(self.shape))
        self.ptr = self._L.rr_host_alloc(n)
        if not self.ptr:
            raise RRError("rr_host_alloc(%d) failed" % n)
        self.array = np.ctypeslib.as_array((C.c_uint8 * n).from_address(self.ptr)).reshape(self.shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._L.rr_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cone_dirs(width_rad, sample_dist, p_in_cone, u_angle, r_variate):
    """rr_cone_dirs (host only): sample_cone_local's geometry on given variates -> [n][3] float32."""
    u = np.ascontiguousarray(u_angle, np.float32); r = np.ascontiguousarray(r_variate, np.float32)
    out = np.zeros((len(u), 3), np.float32)
    rc = lib().rr_cone_dirs(float(width_rad), int(sample_dist), float(p_in_cone), u.ctypes.data, r.ctypes.data, len(u), out.ctypes.data)
    if rc:
        raise RRError("rr_cone_dirs: rc=%d" % rc)
    return out


def sample_cone_local(seed, width_rad, n, sample_dist=2, p_in_cone=0.8):
    """rr_sample_cone_local (host only): the C twin of beams.sample_cone_local."""
    out = np.zeros((int(n), 3), np.float32)
    rc = lib().rr_sample_cone_local(int(seed), float(width_rad), int(n), int(sample_dist), float(p_in_cone), out.ctypes.data)
    if rc:
        raise RRError("rr_sample_cone_local: rc=%d" % rc)
    return out


def load_mesh_file(path, object_order=None):
    """rr_load_mesh_file (host only): PLY / OBJ / DAE -> {"verts", "faces", "face_object_id", "n_objects", "object_names"}.
    object_order: names in the order their ids should run (rr_mesh_reorder_objects: the file's own numbering is depth-first
    scene order -- a material list written for another numbering, e.g. rmagine's, is matched by naming the objects)."""
    m = RRMesh()
    err = C.create_string_buffer(512)
    rc = lib().rr_load_mesh_file(str(path).encode(), C.byref(m), err, len(err))
    if rc:
        raise RRError("%s (rc=%d)" % (err.value.decode(errors="replace"), rc))
    try:
        if object_order:
            names = (C.c_char_p * len(object_order))(*[str(x).encode() for x in object_order])
            rc = lib().rr_mesh_reorder_objects(C.byref(m), names, len(object_order), err, len(err))
            if rc:
                raise RRError("%s (rc=%d)" % (err.value.decode(errors="replace"), rc))
        out = {"verts": np.ctypeslib.as_array(m.verts, (m.n_verts, 3)).copy() if m.n_verts else np.zeros((0, 3), np.float32),
               "faces": np.ctypeslib.as_array(m.faces, (m.n_faces, 3)).copy() if m.n_faces else np.zeros((0, 3), np.uint32),
               "face_object_id": np.ctypeslib.as_array(m.face_object_id, (m.n_faces,)).copy() if m.n_faces else np.zeros(0, np.uint32),
               "n_objects": int(m.n_objects),
               "object_names": [m.object_names[k].decode(errors="replace") for k in range(m.n_objects)] if m.object_names else []}
    finally:
        lib().rr_free_mesh(C.byref(m))
    return out


def partition(n_angles, world, rank):
    """rr_partition: the azimuth block [begin, end) of `rank` (pure host arithmetic, no GPU needed)."""
    b, e = C.c_int(), C.c_int()
    lib().rr_partition(int(n_angles), int(world), int(rank), C.byref(b), C.byref(e))
    return b.value, e.value


def multi_plan(n_angles, n_cells, n_devices, n_frames):
    """rr_multi_plan -> (equal, bytes_per_device, send_off[r][f], recv_off[r][f], piece_bytes[r][f])"""
    eq, bpd = C.c_int(), C.c_size_t()
    so = np.zeros((n_devices, n_frames), np.uint64); ro = np.zeros_like(so); pb = np.zeros_like(so)
    assert C.sizeof(C.c_size_t) == 8
    rc = lib().rr_multi_plan(n_angles, n_cells, n_devices, n_frames, C.byref(eq), C.byref(bpd), so.ctypes.data, ro.ctypes.data, pb.ctypes.data)
    if rc:
        raise RRError("rr_multi_plan: rc=%d" % rc)
    return bool(eq.value), int(bpd.value), so, ro, pb


class MultiContext:
    """rr_multi: several GPUs of one node behind one object (one process; RCCL inside the library)."""

    def __init__(self, devices):
        self._L = lib()
        d = (C.c_int * len(devices))(*[int(x) for x in devices])
        self._h = self._L.rr_create_multi(d, len(devices))
        if not self._h:
            raise RRError(self._L.rr_multi_last_error(None).decode())
        self.cfg = None
        self.n_angles = 400

    def close(self):
        if getattr(self, "_h", None):
            self._L.rr_destroy_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RRError("%s (rc=%d)" % (self._L.rr_multi_last_error(self._h).decode(), rc))

    def rccl_version(self):
        """NCCL version code of the RCCL library behind the communicator (0: none)"""
        return int(self._L.rr_multi_rccl_version(self._h))

    def device_count(self):
        return self._L.rr_multi_device_count(self._h)

    def set_mesh(self, verts, faces, face_object_id=None, builder="host"):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(faces, np.uint32).reshape(-1, 3)
        o = None if face_object_id is None else np.ascontiguousarray(face_object_id, np.uint32)
        fn = {"host": self._L.rr_multi_set_mesh, "gpu": self._L.rr_multi_set_mesh_gpu}[builder]
        self._ck(fn(self._h, v.ctypes.data, len(v), f.ctypes.data, len(f), None if o is None else o.ctypes.data))

    def set_materials(self, materials, object_materials, material_id_air=0):
        m = (RRMaterial * len(materials))(*[RRMaterial(*[float(x) for x in (t.astuple() if hasattr(t, "astuple") else t)])
                                            for t in materials])
        om = np.ascontiguousarray(object_materials, np.int32)
        self._ck(self._L.rr_multi_set_materials(self._h, m, len(materials), om.ctypes.data, len(om), int(material_id_air)))

    def set_config(self, cfg, n_angles=400, **kw):
        self.cfg, self.n_angles = cfg, n_angles
        self._rrcfg = make_config(cfg, n_angles, **kw)
        self._ck(self._L.rr_multi_set_config(self._h, C.byref(self._rrcfg)))

    def set_beam_samples(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        self._ck(self._L.rr_multi_set_beam_samples(self._h, d.ctypes.data, len(d)))

    def set_noise_offsets(self, rnd):
        r = np.ascontiguousarray(rnd, np.float32).ravel()
        self._ck(self._L.rr_multi_set_noise_offsets(self._h, r.ctypes.data, r.size))

    def set_motion_poses(self, poses):
        """include_motion on every device: [n_angles][7], or [k][n_angles][7] (one table per frame of a batch); None: off."""
        if poses is None or len(poses) == 0:
            self._ck(self._L.rr_multi_set_motion_poses(self._h, None, 0))
            return
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        self._ck(self._L.rr_multi_set_motion_poses(self._h, p.ctypes.data, len(p)))

    def simulate_batch(self, poses):
        """-> uint8 [n][n_cells][n_angles] in host memory."""
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        out = np.zeros((len(p), self.cfg.n_cells, self.n_angles), np.uint8)
        self._ck(self._L.rr_multi_simulate_batch(self._h, p.ctypes.data, len(p), out.ctypes.data))
        return out

    def simulate(self, pose):
        return self.simulate_batch([pose])[0]

    def simulate_batch_async(self, poses, h_imgs_ptr):
        """Pipelined: enqueue the batch, images [n][n_cells][n_angles] arrive at h_imgs_ptr (page-locked: HostImages);
        complete after wait(h_imgs_ptr)."""
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        self._ck(self._L.rr_multi_simulate_batch_async(self._h, p.ctypes.data, len(p), h_imgs_ptr))

    def wait(self, h_imgs_ptr=None):
        self._ck(self._L.rr_multi_wait(self._h, h_imgs_ptr))
