"""A/B of a library env switch on the bench's batch shape (8 poses per batch, 4 streams): images/s and image hash.
usage: probe_env.py <config id> VAR=a VAR=b ..."""
import sys, os, time, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
if sys.argv[2] == "child":
    import numpy as np, torch, hashlib
    from radarays_ros_amd import native, params, scenes
    from radarays_ros_amd.fixtures import golden_beams, materials_for
    wl = int(sys.argv[1]); P = 4 if wl != 2 else 1
    s = scenes.config_scene(wl)
    cfg = params.kaist_preset(n_reflections=P, n_samples=200, ambient_noise=2)
    c = native.Context(0)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
    c.set_config(cfg); c.set_beam_samples(golden_beams(200))
    c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 16 * 400) * 1000).astype(np.float32))
    poses = scenes.trajectory(16, s["name"])
    F, NS = 8, 4
    streams = [torch.cuda.Stream() for _ in range(NS)]
    dimgs = [torch.zeros((F, cfg.n_cells, 400), dtype=torch.uint8, device="cuda") for _ in range(NS)]
    def one(k): c.simulate_batch_device([poses[(k * F + f) % 16] for f in range(F)], dimgs[k % NS].data_ptr(), streams[k % NS].cuda_stream)
    for k in range(80): one(k)
    c.synchronize(); torch.cuda.synchronize()
    res = []
    for rep in range(2):
        t0 = time.perf_counter(); K = 200
        for k in range(K): one(k)
        torch.cuda.synchronize(); res.append(round(K * F / (time.perf_counter() - t0), 1))
    c.set_timing_mode(2); c.kernel_time("trace", True)
    for k in range(8): one(k); torch.cuda.synchronize()
    ms, n = c.kernel_time("trace", True)
    one(0); torch.cuda.synchronize()
    print(json.dumps({"env": sys.argv[3], "img_s": res, "trace_us_isolated": round(1e3 * ms / max(n, 1), 1),
                      "md5": hashlib.md5(dimgs[0].cpu().numpy().tobytes()).hexdigest()[:8]}), flush=True)
else:
    for kv in sys.argv[2:]:        # VAR=a or VAR=a,VAR2=b
        env = dict(os.environ)
        for one in kv.split(","):
            k, v = one.split("=")
            env[k] = v
        subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1], "child", kv], env=env)
