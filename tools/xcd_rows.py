#!/usr/bin/env python3
"""How a later-pass trace launch's LIVE 16-ray workgroups fall on the 8 XCDs for a given row length (CPU model of DESIGN.md §3
"Rows are kept odd").  The hardware deals the workgroups of a grid (row, n_seg) out round robin in flat order y * row + x;
workgroup x of segment y is live while x * 16 < count[y].  The per-segment wave counts come from the CPU restatement
(ORC_COUNTS hook of oracle/radarays_oracle.c) on the first 8 poses of a workload's trajectory = one 8-frame batch.
usage: tools/xcd_rows.py [workload = target_10M_400x200_4pass] [row lengths ...]"""
import os
import sys
import tempfile

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402
from radarays_ros_amd import params, scenes  # noqa: E402
from radarays_ros_amd.fixtures import golden_beams, materials_for  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "target_10M_400x200_4pass"
rows = [int(x) for x in sys.argv[2:]] or [63, 64, 65, 72, 73, 80, 81]
scene_id, n_pass, n_rays = bench.WORKLOADS[wl]
scene = scenes.config_scene(scene_id)
cfg = params.kaist_preset(n_reflections=n_pass, n_samples=n_rays, ambient_noise=0)
mats = [m.astuple() for m in materials_for(scene)]
poses = scenes.trajectory(16, scene["name"])
O.build()
sc = O.Scene(scene["verts"], scene["faces"], scene["face_object_id"])
cnt = np.zeros((8, n_pass, params.N_ANGLES), int)
with tempfile.TemporaryDirectory() as tmp:
    for k in range(8):
        os.environ["ORC_COUNTS"] = path = os.path.join(tmp, "c%d.txt" % k)
        O.simulate(sc, mats, scene["object_materials"], cfg, golden_beams(n_rays), poses[k], want_f32=False)
        for line in open(path):
            a, p, n = map(int, line.split())
            cnt[k, p, a] = n
    os.environ.pop("ORC_COUNTS")
for p in range(1, n_pass):
    live = np.ceil(cnt[:, p, :].reshape(-1) / 16).astype(int)         # segment y = frame * 400 + azimuth
    print("pass %d: live 16-ray workgroups per segment: mean %.1f, max %d" % (p, live.mean(), live.max()))
    for row in rows:
        if row < live.max():
            continue
        w = np.zeros(8)
        for y, n in enumerate(live):
            np.add.at(w, (y * row + np.arange(n)) % 8, 1.0)
        print("   row of %3d workgroups: live workgroups per XCD %6d .. %6d, busiest / mean %.3f" % (row, w.min(), w.max(), w.max() / w.mean()))
