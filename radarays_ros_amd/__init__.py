"""radarays_ros_amd -- MI355X (gfx950) implementation of radarays_ros' per-azimuth
radar ray loop (RadarCPU::simulate), behind the reference's Radar::simulate seam.

Only the hot path lives here: csrc/ (HIP kernels + the C ABI of
include/radarays_mi355.h), the ctypes binding (native), the host mirror of the
reference backend interface (radar.RadarHIP), the parameter surface (params),
and the synthetic inputs of SURVEY.md §8d (scenes, beams).
"""
from . import params, scenes, beams  # noqa: F401

__all__ = ["params", "scenes", "beams", "meshio", "native", "radar", "dist"]
