"""Parameter surface of the reference kept 1:1.

RadarModelConfig mirrors cfg/RadarModel.cfg:11-85 (dynamic_reconfigure fields,
same names, defaults and ranges); RadarMaterial / RadarModel / RadarParams mirror
msg/RadarMaterial.msg, msg/RadarModel.msg, msg/RadarParams.msg; default_params()
mirrors include/radarays_ros/ros_helper.h (beam 8 deg, 200 samples, 2 reflections).
"""
from dataclasses import dataclass, field, asdict
from typing import List
import math


@dataclass
class RadarMaterial:            # msg/RadarMaterial.msg:1-4
    velocity: float = 0.3
    ambient: float = 1.0
    diffuse: float = 0.0
    specular: float = 1.0

    def astuple(self):
        return (self.velocity, self.ambient, self.diffuse, self.specular)


@dataclass
class RadarModel:               # msg/RadarModel.msg:1-3
    beam_width: float = 8.0 * math.pi / 180.0   # radians (Radar.cpp:213)
    n_samples: int = 200
    n_reflections: int = 2


@dataclass
class RadarParams:              # msg/RadarParams.msg
    materials: List[RadarMaterial] = field(default_factory=list)
    model: RadarModel = field(default_factory=RadarModel)


def default_params() -> RadarParams:
    """include/radarays_ros/ros_helper.h:21-35."""
    return RadarParams(materials=[], model=RadarModel())


@dataclass
class RadarModelConfig:         # cfg/RadarModel.cfg
    z_offset: float = 0.0
    range_min: float = 0.0
    range_max: float = 600.0
    beam_width: float = 8.0                    # degrees
    resolution: float = 0.0438
    n_cells: int = 3424
    n_samples: int = 10
    beam_sample_dist: int = 2                  # 0 D1, 1 D2, 2 D3 normal, 3 D4
    beam_sample_dist_normal_p_in_cone: float = 0.8
    n_reflections: int = 4
    energy_min: float = 0.0
    energy_max: float = 0.5
    signal_max: float = 120.0
    signal_denoising: int = 1                  # 0 none 1 triangular 2 gaussian 3 maxwell_boltzmann
    signal_denoising_triangular_width: int = 50
    signal_denoising_triangular_mode: float = 0.35
    signal_denoising_gaussian_width: int = 50
    signal_denoising_gaussian_mode: float = 0.5
    signal_denoising_mb_width: int = 50
    signal_denoising_mb_mode: float = 0.4
    ambient_noise: int = 2                     # 0 none 1 uniform 2 perlin
    ambient_noise_at_signal_0: float = 0.3
    ambient_noise_at_signal_1: float = 0.03
    ambient_noise_energy_max: float = 0.5
    ambient_noise_energy_min: float = 0.1
    ambient_noise_energy_loss: float = 0.05
    ambient_noise_uniform_max: float = 0.15    # unused on the CPU path
    ambient_noise_perlin_scale_low: float = 0.05
    ambient_noise_perlin_scale_high: float = 0.2
    ambient_noise_perlin_p_low: float = 0.9
    scroll_image: int = 0
    multipath_threshold: float = 0.5
    record_multi_reflection: bool = True
    record_multi_path: bool = False
    include_motion: bool = True

    def copy(self, **kw):
        d = asdict(self)
        d.update(kw)
        return RadarModelConfig(**d)


def kaist_preset(**kw) -> RadarModelConfig:
    """cfg/mulran_kaist_dyncfg.yaml (the paper preset)."""
    c = RadarModelConfig(
        ambient_noise=2, ambient_noise_at_signal_0=0.1, ambient_noise_at_signal_1=0.03,
        ambient_noise_energy_loss=0.05, ambient_noise_energy_max=0.1, ambient_noise_energy_min=0.05,
        ambient_noise_uniform_max=0.15, beam_sample_dist=2, beam_sample_dist_normal_p_in_cone=0.8,
        beam_width=10.0, energy_max=0.72, energy_min=0.0, include_motion=False,
        multipath_threshold=0.5, n_cells=3424, n_reflections=4, n_samples=50,
        range_max=600.0, range_min=0.0, record_multi_path=False, record_multi_reflection=True,
        resolution=0.0595238, scroll_image=0, signal_denoising=1,
        signal_denoising_gaussian_mode=0.5, signal_denoising_gaussian_width=50,
        signal_denoising_mb_mode=0.4, signal_denoising_mb_width=50,
        signal_denoising_triangular_mode=0.35, signal_denoising_triangular_width=35,
        signal_max=110.0, z_offset=0.0)
    return c.copy(**kw)


def laserlike_preset(**kw) -> RadarModelConfig:
    """cfg/mulran_kaist_dyncfg_laserlike.yaml: ONE ray per azimuth (beam_width 1e-4 deg, D1), one pass, no smear kernel
    (signal_denoising 0: slice[bin] = max(slice[bin], I), RadarCPU.cpp:438-446), no ambient noise -- the radar as a 2D
    laser scanner."""
    c = kaist_preset(ambient_noise=0, beam_sample_dist=0, beam_sample_dist_normal_p_in_cone=0.999, beam_width=0.0001,
                     energy_min=0.72, n_reflections=1, n_samples=1, signal_denoising=0)
    return c.copy(**kw)


def minimal_preset(**kw) -> RadarModelConfig:
    """cfg/mulran_kaist_dyncfg_minimal.yaml: 10 samples in a 2 degree beam, smear kernel W = 23 with mode 0.1
    (-> int(0.1 * 23) = 2, RadarCPU.cpp:57), weaker noise.  The file holds neither include_motion nor signal_max: what
    `dynparam load` does not set keeps the .cfg default (True, 120)."""
    c = kaist_preset(ambient_noise_at_signal_0=0.05, ambient_noise_at_signal_1=0.01, ambient_noise_energy_max=0.08,
                     beam_sample_dist_normal_p_in_cone=0.99, beam_width=2.0, n_samples=10,
                     signal_denoising_triangular_mode=0.1, signal_denoising_triangular_width=23,
                     include_motion=True, signal_max=120.0)
    return c.copy(**kw)


def kaist_materials() -> List[RadarMaterial]:
    """config/mulran_kaist02.yaml:8-20 (air, wall stone)."""
    return [RadarMaterial(0.3, 1.0, 0.0, 1.0), RadarMaterial(0.0, 1.0, 0.0, 3000.0)]


def oru4_test_materials() -> List[RadarMaterial]:
    """config/oru4_test.yaml:8-33: air, wall stone, shelf wood, window glass (v = 0.03: refracts), metal."""
    return [RadarMaterial(0.3, 1.0, 0.0, 1.0), RadarMaterial(0.0, 1.0, 0.0, 3000.0), RadarMaterial(0.0, 1.0, 0.0, 1.0),
            RadarMaterial(0.03, 1.0, 0.0, 100.0), RadarMaterial(0.0, 1.0, 0.0, 1.0)]


def oru3_legacy_materials() -> List[RadarMaterial]:
    """config/oru3.yaml (legacy structure-of-arrays table, read by src/ray_reflection_test.cpp:156-167): air + 11 geological
    materials that all transmit (water 0.033 .. ice 0.16 m/ns) with fractional BRDF exponents (0.1 .. 0.8) + one opaque
    material with exponent 0 (cos^0 = 1)."""
    v = [0.3, 0.033, 0.01, 0.16, 0.15, 0.06, 0.12, 0.09, 0.07, 0.06, 0.13, 0.13, 0.0]
    a = [0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 1.0, 0.5, 1.0]
    d = [0.1, 0.1, 0.1, 0.3, 0.8, 0.8, 0.8, 0.8, 0.8, 0.8, 0.8, 0.8, 0.0]
    s = [0.1, 0.8, 0.8, 0.6, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.0]
    return [RadarMaterial(*t) for t in zip(v, a, d, s)]


def oru4_legacy_materials() -> List[RadarMaterial]:
    """config/oru4.yaml (legacy table): stone / wood with velocities of 0.001 / 0.002 m/ns (n21 = 300: no total reflection
    limit, a refraction direction almost along the normal), glass 0.05 with ambient 0.01 and exponent 2000, metal, and a
    "test" material with exponent 0."""
    return [RadarMaterial(0.3, 0.5, 0.1, 0.1), RadarMaterial(0.001, 0.6, 0.3, 30.0), RadarMaterial(0.002, 0.6, 0.3, 70.0),
            RadarMaterial(0.05, 0.01, 0.04, 2000.0), RadarMaterial(0.0, 1.0, 1.0, 2000.0), RadarMaterial(0.0, 1.0, 0.0, 0.0)]


# config/oru4_test.yaml:37-56 (the same list closes config/mulran_kaist02.yaml:24-43): material of each of the 18 objects
# of the ORU4 scene, in the order rmagine numbers the geometries of the .dae (ground, door glass, wall, door wood, ...)
ORU4_OBJECT_MATERIALS = [1, 3, 1, 2, 3, 3, 2, 2, 3, 3, 2, 2, 4, 2, 4, 2, 4, 1]


# one penetrable material so that Snell/Fresnel splitting
# actually occurs (the KAIST table alone never refracts: v = 0)
PENETRABLE = RadarMaterial(0.1, 0.6, 0.3, 30.0)

def config5_materials() -> List[RadarMaterial]:
    """Air + the 8 per-triangle materials of SURVEY §8d config 5: four opaque ones (v = 0, like the
    KAIST wall: reflection only) and four penetrable ones, with spread BRDF parameters.  The BRDF is
    the checkout's A + B cos^C (the Cook-Torrance family of the dev/flex branch is not in it)."""
    return [RadarMaterial(0.3, 1.0, 0.0, 1.0),
            RadarMaterial(0.0, 1.0, 0.0, 3000.0), RadarMaterial(0.0, 0.8, 0.2, 100.0),
            RadarMaterial(0.0, 0.5, 0.5, 10.0), RadarMaterial(0.0, 0.2, 0.8, 2.0),
            RadarMaterial(0.1, 0.6, 0.3, 30.0), RadarMaterial(0.15, 0.4, 0.4, 8.0),
            RadarMaterial(0.2, 0.7, 0.2, 300.0), RadarMaterial(0.25, 0.3, 0.6, 1.0)]


N_ANGLES = 400                       # Radar.cpp:29
WAVE_ENERGY_THRESHOLD = 0.001        # Radar.cpp:24
