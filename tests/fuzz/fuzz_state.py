"""Stateful fuzz of the C ABI: ONE context is reconfigured at random (mesh / builder, materials, config,
beam samples, noise offsets, motion poses), and after every change its frame is compared with the frame of
a FRESH context given the same state.  Catches stale tables, buffer sizing and lane-reuse mistakes.
usage: fuzz_state.py [iterations] [seed]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from common import golden_beams, GOLDEN
sys.path.insert(0, GOLDEN)
import gen_oracle_images as gen


def run(iters=100, seed=0, verbose=True):
    rs = np.random.RandomState(seed)
    meshes = [scenes.box12(), gen.two_room_scene(), scenes.heightfield_room(8, n_buildings=3)]
    for m in meshes:
        m.setdefault("object_materials", [1] * (int(m["face_object_id"].max()) + 1))
    state = dict(mesh=0, builder="host", n_angles=400, beams=16, noise_seed=1, motion=False,
                 mats=params.kaist_materials() + [params.PENETRABLE],
                 cfg=params.kaist_preset(n_reflections=2, ambient_noise=2))

    def apply(c, st, only=None):
        m = meshes[st["mesh"]]
        if only in (None, "mesh"):
            c.set_mesh(m["verts"], m["faces"], m["face_object_id"], builder=st["builder"])
        if only in (None, "mesh", "mats"):
            om = [min(int(x), len(st["mats"]) - 1) for x in m["object_materials"]]
            c.set_materials(st["mats"], om, 0)
        if only in (None, "cfg", "angles"):
            c.set_config(st["cfg"], st["n_angles"])
        if only in (None, "beams"):
            c.set_beam_samples(golden_beams(st["beams"]))
        if only in (None, "noise", "angles"):
            c.set_noise_offsets((np.random.RandomState(st["noise_seed"]).uniform(0, 1, st["n_angles"]) * 1000).astype(np.float32))
        if only in (None, "motion", "angles"):
            if st["motion"]:
                ps = np.tile(scenes.default_pose("box12"), (st["n_angles"], 1)).astype(np.float32)
                ps[:, 4] += np.linspace(0, 0.3, st["n_angles"], dtype=np.float32)
                c.set_motion_poses(ps)
            else:
                c.set_motion_poses(None)

    c = native.Context(0)
    apply(c, state)
    pose = scenes.default_pose("box12")
    bad = 0
    for it in range(iters):
        what = rs.choice(["mesh", "mats", "cfg", "beams", "noise", "motion", "angles", "none"])
        if what == "mesh":
            state["mesh"] = int(rs.randint(0, len(meshes))); state["builder"] = str(rs.choice(["host", "gpu"]))
        elif what == "mats":
            state["mats"] = [params.RadarMaterial(0.3, 1.0, 0.0, 1.0)] + [
                params.RadarMaterial(float(rs.choice([0.0, 0.1, 0.2])), float(rs.uniform(0, 1)), float(rs.uniform(0, 1)),
                                     float(rs.choice([1.0, 30.0, 3000.0]))) for _ in range(int(rs.randint(2, 5)))]
        elif what == "cfg":
            state["cfg"] = params.kaist_preset(
                n_reflections=int(rs.randint(0, 5)), ambient_noise=int(rs.choice([0, 1, 2])), n_cells=int(rs.choice([3424, 500, 1000])),
                signal_denoising=int(rs.choice([0, 1, 3])), signal_denoising_triangular_width=int(rs.randint(1, 80)),
                scroll_image=int(rs.randint(0, state["n_angles"])), record_multi_path=bool(rs.randint(0, 2)))
        elif what == "beams":
            state["beams"] = int(rs.randint(1, 80))
        elif what == "noise":
            state["noise_seed"] = int(rs.randint(0, 1000))
        elif what == "motion":
            state["motion"] = not state["motion"]
        elif what == "angles":
            state["n_angles"] = int(rs.choice([400, 100, 36]))
            state["cfg"] = state["cfg"].copy(scroll_image=int(rs.randint(0, state["n_angles"])))
        if what != "none":
            apply(c, state, what)
        try:
            got = c.simulate(pose, 0, state["n_angles"])[0]
            err = None
        except native.RRError as e:
            got, err = None, str(e)
        f = native.Context(0)
        apply(f, state)
        try:
            want = f.simulate(pose, 0, state["n_angles"])[0]
            werr = None
        except native.RRError as e:
            want, werr = None, str(e)
        f.close()
        ok = (err is None) == (werr is None) and (got is None or np.array_equal(got, want))
        # the asynchronous / batched entry points of the SAME long-lived context, after the same
        # reconfiguration (lane buffers sized for an earlier config: n_cells / capacity growth must
        # never leave a stale pointer behind) -- compared with the fresh context's frame
        dev_bad = []
        if want is not None and err is None:
            import torch
            A, Cc = state["n_angles"], state["cfg"].n_cells
            img = torch.zeros((2, Cc, A), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            sp = torch.cuda.current_stream().cuda_stream
            for lane in range(3):          # rr_simulate_device rotates over its lane streams: visit them all
                img.zero_()
                torch.cuda.synchronize()
                c.simulate_device(pose, img[0].data_ptr(), sp)
                c.synchronize(sp)
                if not np.array_equal(img[0].cpu().numpy(), want): dev_bad.append("simulate_device[%d]" % lane)
            if not state["motion"]:
                for lane in range(4):
                    img.zero_()
                    torch.cuda.synchronize()
                    c.simulate_batch_device(np.stack([pose, pose]), img.data_ptr(), sp)
                    c.synchronize(sp)
                    g2 = img.cpu().numpy()
                    if not (np.array_equal(g2[0], want) and np.array_equal(g2[1], want)): dev_bad.append("simulate_batch_device[%d]" % lane)
            if dev_bad:
                ok = False
                print("device entry points differ:", dev_bad)
        if not ok:
            bad += 1
            print("MISMATCH at iteration", it, "after", what, "state", {k: (v if k not in ("mats", "cfg") else "...") for k, v in state.items()},
                  "errors", err, werr, "diff pixels", None if got is None or want is None else int((got != want).sum()))
    c.close()
    if verbose:
        print("stateful fuzz: %d iterations, %d mismatches" % (iters, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
