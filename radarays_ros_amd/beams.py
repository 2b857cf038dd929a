"""Beam-sample generation: sample_cone_local of the reference
(src/radarays_ros/radar_algorithms.cpp:248-294) with a seeded generator.

The reference seeds std::mt19937 from std::random_device, so its samples are not
reproducible; the C ABI therefore takes the directions as an INPUT
(rr_set_beam_samples) and this module is the convenience that produces them.
Variates come from numpy's RandomState (documented, portable); the geometry
(z = sqrt(2) erfinv(p), radius laws D1..D4, Euler(0, alpha, beta) * e_x) follows
the reference line by line in float32.
"""
import numpy as np


def erfinvf(a):
    """radar_math.h:13-44 (single-precision polynomial), scalar."""
    a = np.float32(a)
    t = np.float32(np.log(np.float32(np.float32(1.0) - a * a)))
    if abs(t) > 6.125:
        c = [3.03697567e-10, 2.93243101e-8, 1.22150334e-6, 2.84108955e-5, 3.93552968e-4,
             3.02698812e-3, 4.83185798e-3, -2.64646143e-1, 8.40016484e-1]
    else:
        c = [5.43877832e-9, 1.43285448e-7, 1.22774793e-6, 1.12963626e-7, -5.61530760e-5,
             -1.47697632e-4, 2.31468678e-3, 1.15392581e-2, -2.32015476e-1, 8.86226892e-1]
    p = np.float32(c[0])
    for k in c[1:]:
        p = np.float32(np.float64(p) * np.float64(t) + np.float64(np.float32(k)))
    return np.float32(a * p)


def _quat_from_euler(roll, pitch, yaw):
    f = np.float32
    cr, sr = np.cos(roll / f(2)), np.sin(roll / f(2))
    cp, sp = np.cos(pitch / f(2)), np.sin(pitch / f(2))
    cy, sy = np.cos(yaw / f(2)), np.sin(yaw / f(2))
    w = cr * cp * cy + sr * sp * sy
    x = sr * cp * cy - cr * sp * sy
    y = cr * sp * cy + sr * cp * sy
    z = cr * cp * sy - sr * sp * cy
    return x.astype(f), y.astype(f), z.astype(f), w.astype(f)


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz)


def variates(n_samples, sample_dist, seed):
    rs = np.random.RandomState(seed)
    u_angle = rs.uniform(0.0, 1.0, n_samples).astype(np.float32)
    if sample_dist in (0, 1):
        r = rs.uniform(0.0, 1.0, n_samples).astype(np.float32)
    else:
        r = rs.standard_normal(n_samples).astype(np.float32)
    return u_angle, r


def cone_radius(width, sample_dist, p_in_cone, r_variate):
    """radar_algorithms.cpp:263-280: the radius law D1..D4 (scripts/radaray_beams.py:23-25,78-92) in float32;
    `width` in any angular unit, the radius comes back in the same."""
    f = np.float32
    r_variate = np.asarray(r_variate, f)
    z = f(np.float64(np.sqrt(2.0)) * np.float64(erfinvf(p_in_cone)))
    radius = f(np.float64(f(width)) / 2.0)
    if sample_dist == 0:
        rr = r_variate * radius
    elif sample_dist == 1:
        rr = np.sqrt(r_variate) * radius
    elif sample_dist == 2:
        rr = (r_variate / z) * radius
    elif sample_dist == 3:
        rr = np.sqrt(np.abs(r_variate) / z) * radius
    else:
        raise ValueError("beam_sample_dist must be 0..3")
    return rr.astype(f)


def cone_dirs(width_rad, sample_dist, p_in_cone, u_angle, r_variate):
    """radar_algorithms.cpp:263-289 on given variates -> [n][3] float32."""
    f = np.float32
    u_angle = np.asarray(u_angle, f)
    ang = (np.float64(u_angle * f(2.0)) * np.pi - np.pi).astype(f)
    rr = cone_radius(width_rad, sample_dist, p_in_cone, r_variate)
    alpha = (rr * np.cos(ang)).astype(f)
    beta = (rr * np.sin(ang)).astype(f)
    q = _quat_from_euler(np.zeros_like(alpha), alpha, beta)
    zero = np.zeros_like(alpha)
    one = np.ones_like(alpha)
    p = (one, zero, zero, zero)
    qi = (-q[0], -q[1], -q[2], q[3])
    r = _qmul(_qmul(q, p), qi)
    return np.stack([r[0], r[1], r[2]], -1).astype(f)


def sample_cone_local(beam_width_deg, n_samples, sample_dist=2, p_in_cone=0.8, seed=42):
    """m_waves_start directions for RadarModelConfig.beam_width (degrees)."""
    return sample_cone_local_rad(np.float32(beam_width_deg * np.pi / 180.0), n_samples, sample_dist, p_in_cone, seed)   # Radar.cpp:213


def sample_cone_local_rad(width_rad, n_samples, sample_dist=2, p_in_cone=0.8, seed=42):
    """The same for RadarModel.beam_width (radians, f32: msg/RadarModel.msg) -- what rr_sample_cone_local takes.  A width that
    went through degrees and back can differ by an ulp from the one the C++ twin passes (advisor, round 4)."""
    u, r = variates(n_samples, sample_dist, seed)
    return cone_dirs(np.float32(width_rad), sample_dist, p_in_cone, u, r)
