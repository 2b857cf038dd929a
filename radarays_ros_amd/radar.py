"""Host-side mirror of the reference's backend interface for this path.

Reference seam (C++):   class Radar { virtual sensor_msgs::ImagePtr simulate(ros::Time) = 0; }
                        include/radarays_ros/Radar.hpp:34-105, src/radarays_ros/Radar.cpp
Backends there:         RadarCPU (Embree), RadarGPU (OptiX) -- chosen in radar_simulator.cpp:118-176.
`RadarHIP` is the third backend: same member names / same argument meaning /
same error behaviour (simulate() returns None when no transform is known, like
the null ImagePtr of RadarCPU.cpp:129-133), on top of libradarays_mi355.so.
ROS itself is absent here: TF lookup is replaced by updateTsm(pose) and the
returned message is a plain `Image` object with sensor_msgs/Image's fields.
"""
from dataclasses import dataclass, field

import numpy as np

from . import beams, native
from .params import (N_ANGLES, WAVE_ENERGY_THRESHOLD, RadarModelConfig, RadarParams,
                     default_params)


@dataclass
class Header:
    stamp: float = 0.0
    frame_id: str = ""


@dataclass
class Image:
    """sensor_msgs/Image as RadarCPU.cpp:555-561 fills it."""
    header: Header = field(default_factory=Header)
    height: int = 0
    width: int = 0
    encoding: str = "mono8"
    is_bigendian: int = 0
    step: int = 0
    data: np.ndarray = None     # [height][width] uint8


class RadarHIP:
    def __init__(self, verts, faces, face_object_id=None, map_frame="map", sensor_frame="sensor",
                 device=0, beam_seed=42):
        self.m_map_frame = map_frame
        self.m_sensor_frame = sensor_frame
        self.m_params: RadarParams = default_params()          # Radar.cpp:22
        self.m_material_id_air = 0                             # Radar.cpp:23
        self.m_wave_energy_threshold = WAVE_ENERGY_THRESHOLD   # Radar.cpp:24
        self.m_resample = True                                 # Radar.cpp:25
        self.m_object_materials = []
        self.m_cfg = RadarModelConfig()
        self.m_waves_start = None
        self.Tsm_last = None
        self._beam_seed = beam_seed
        self._noise_seed = 7
        self._motion = None
        self._ctx = native.Context(device)
        self._ctx.set_mesh(verts, faces, face_object_id)
        self._dirty_cfg = True
        self._dirty_mat = True
        self.updateDynCfg(self.m_cfg)

    # -- Radar.cpp:220-226
    def loadParams(self, materials, object_materials, material_id_air=0):
        self.m_params.materials = list(materials)
        self.m_object_materials = list(object_materials)
        self.m_material_id_air = int(material_id_air)
        self._dirty_mat = True

    def getParams(self):
        return self.m_params

    def setParams(self, params: RadarParams):
        self.m_params = params
        self._dirty_mat = True
        self._dirty_cfg = True

    # -- Radar.cpp:188-218
    def updateDynCfg(self, config: RadarModelConfig, level=0):
        old = self.m_cfg
        if (config.beam_sample_dist != old.beam_sample_dist
                or abs(config.beam_width - old.beam_width) > 0.001
                or config.n_samples != old.n_samples
                or abs(config.beam_sample_dist_normal_p_in_cone - old.beam_sample_dist_normal_p_in_cone) > 0.001):
            self.m_resample = True
        self.m_params.model.beam_width = config.beam_width * np.pi / 180.0
        self.m_params.model.n_samples = config.n_samples
        self.m_params.model.n_reflections = config.n_reflections
        self.m_cfg = config.copy()
        self._dirty_cfg = True

    # -- Radar.cpp:80-132 (TF lookup replaced by an explicit pose)
    def updateTsm(self, pose_qxyzw_t=None):
        if pose_qxyzw_t is not None:
            p = np.asarray(pose_qxyzw_t, np.float32)
            if p.shape != (7,) or not np.all(np.isfinite(p)):
                return False
            self.Tsm_last = p
        return self.Tsm_last is not None

    def setBeamSamples(self, dirs):
        """Inject m_waves_start (tests use the committed fixture)."""
        self.m_waves_start = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        self.m_resample = False
        self._ctx.set_beam_samples(self.m_waves_start)

    def setMotionPoses(self, poses):
        """include_motion (cfg/RadarModel.cfg:85): the pose TF would return at each azimuth
        (RadarCPU.cpp:190-196), [400][7]; None -> one pose per frame."""
        self._motion = None if poses is None else np.ascontiguousarray(poses, np.float32).reshape(-1, 7)
        self._ctx.set_motion_poses(self._motion)
        if self._motion is not None:
            self.Tsm_last = self._motion[-1].copy()

    def setNoiseOffsets(self, rnd):
        self._ctx.set_noise_offsets(rnd)

    def _push(self):
        if self._dirty_cfg:
            cfg = self.m_cfg.copy(n_reflections=self.m_params.model.n_reflections)
            self._ctx.set_config(cfg, N_ANGLES, wave_energy_threshold=self.m_wave_energy_threshold)
            self._dirty_cfg = False
        if self._dirty_mat:
            self._ctx.set_materials(self.m_params.materials, self.m_object_materials, self.m_material_id_air)
            self._dirty_mat = False
        if self.m_resample:     # RadarCPU.cpp:136-145
            self.m_waves_start = beams.sample_cone_local_rad(     # model.beam_width: radians, what the C++ twin passes
                self.m_params.model.beam_width, self.m_params.model.n_samples, self.m_cfg.beam_sample_dist,
                self.m_cfg.beam_sample_dist_normal_p_in_cone, seed=self._beam_seed)
            self._ctx.set_beam_samples(self.m_waves_start)
            self.m_resample = False

    # -- the seam: Radar.hpp:64 / RadarCPU.cpp:30-564
    def simulate(self, stamp=0.0, want_f32=False):
        if not self.updateTsm():
            print("Couldn't get Transform between sensor and map. Skipping...")   # RadarCPU.cpp:131
            return None
        self._push()
        u8, f32, stats = self._ctx.simulate(self.Tsm_last, 0, N_ANGLES, want_f32=want_f32)
        msg = Image(header=Header(stamp=stamp, frame_id=self.m_sensor_frame),
                    height=u8.shape[0], width=u8.shape[1], encoding="mono8", step=u8.shape[1], data=u8)
        self.last_f32 = f32
        self.last_stats = stats
        return msg

    def _batch(self, poses, sweeps, stamp):
        """Offline generation (the twin of integration/.../RadarHIP.cpp: simulateBatch / simulateSweeps): one image per pose,
        up to 64 poses per set of launches, delivered to page-locked host memory (rr_simulate_batch_host_async)."""
        self._push()
        out = []
        poses = np.ascontiguousarray(poses, np.float32)
        n_total = len(poses)
        for at in range(0, n_total, 64):
            chunk = poses[at:at + 64]
            if sweeps:         # chunk: [n][n_angles][7] -- table k = the per-azimuth poses of frame k (RadarCPU.cpp:190-196)
                self._ctx.set_motion_poses(chunk.reshape(-1, 7))
                first = np.ascontiguousarray(chunk[:, 0, :])
            else:
                self._ctx.set_motion_poses(None)
                first = chunk
            h = native.HostImages((len(chunk), self.m_cfg.n_cells, N_ANGLES))
            try:
                self._ctx.simulate_batch_host_async(first, h.ptr)
                self._ctx.wait_host(h.ptr)
                for k in range(len(chunk)):
                    u8 = h.array[k].copy()
                    out.append(Image(header=Header(stamp=stamp, frame_id=self.m_sensor_frame), height=u8.shape[0], width=u8.shape[1],
                                     encoding="mono8", step=u8.shape[1], data=u8))
            finally:
                h.close()
        self._ctx.set_motion_poses(self._motion if getattr(self, "_motion", None) is not None else None)
        return out

    def simulateBatch(self, poses, stamp=0.0):
        """[n][7] poses -> n Images (one set of launches per 64 poses)."""
        return self._batch(np.asarray(poses, np.float32).reshape(-1, 7), False, stamp)

    def simulateSweeps(self, sweeps, stamp=0.0):
        """include_motion: [n][400][7] per-azimuth pose tables -> n Images, one table per frame of a batch."""
        return self._batch(np.asarray(sweeps, np.float32).reshape(-1, N_ANGLES, 7), True, stamp)

    def simulateMaterialSets(self, sets, stamp=0.0):
        """The gen_radar_image action of the optimisation loop (action/GenRadarImage.action,
        scripts/radaray_opti.py:170-200), batched: `sets` is a list of material lists (each as long
        as loadParams() gave); returns one mono8 Image per set for the current pose, one call."""
        if not self.updateTsm():
            print("Couldn't get Transform between sensor and map. Skipping...")
            return []
        self._push()
        n_mat = len(self.m_params.materials)
        if any(len(x) != n_mat for x in sets):
            raise ValueError("every material set needs %d entries" % n_mat)
        arr = [[m.astuple() for m in x] for x in sets]
        imgs = self._ctx.simulate_material_sets(self.Tsm_last, arr)
        return [Image(header=Header(stamp=stamp, frame_id=self.m_sensor_frame), height=u8.shape[0], width=u8.shape[1],
                      encoding="mono8", step=u8.shape[1], data=u8) for u8 in imgs]

    def simulateParamSets(self, sets, stamp=0.0, real=None, want_images=True):
        """The same action over the optimiser's WHOLE parameter vector (scripts/radaray_opti.py:36-113): `sets` is a list
        of RadarParams (materials + model.beam_width [rad] / n_samples / n_reflections); one rr_simulate_param_sets call.
        Beams are drawn like _push() draws them (same seed: equal widths share pass 0).  Returns (images or None,
        psnr or None) -- with `real` (mono8 [n_cells][400]) the objective values of radaray_opti.py:196."""
        if not self.updateTsm():
            print("Couldn't get Transform between sensor and map. Skipping...")
            return None, None
        self._push()
        n_mat, nb = len(self.m_params.materials), self.m_params.model.n_samples
        ps = []
        for p in sets:
            if len(p.materials) != n_mat or p.model.n_samples != nb:
                raise ValueError("every parameter set needs %d materials and n_samples = %d" % (n_mat, nb))
            dirs = None
            if abs(p.model.beam_width - self.m_params.model.beam_width) > 1e-7:
                dirs = beams.sample_cone_local_rad(p.model.beam_width, nb, self.m_cfg.beam_sample_dist,
                                                   self.m_cfg.beam_sample_dist_normal_p_in_cone, seed=self._beam_seed)
            ps.append({"materials": [m.astuple() for m in p.materials], "beam_dirs": dirs, "n_reflections": int(p.model.n_reflections)})
        imgs, psnr = self._ctx.simulate_param_sets(self.Tsm_last, ps, n_mat, ref_u8=real, want_images=want_images)
        msgs = None if imgs is None else [
            Image(header=Header(stamp=stamp, frame_id=self.m_sensor_frame), height=u8.shape[0], width=u8.shape[1],
                  encoding="mono8", step=u8.shape[1], data=u8) for u8 in imgs]
        return msgs, psnr

    @property
    def context(self):
        return self._ctx
