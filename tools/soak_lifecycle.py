"""Life-cycle soak of a context with host delivery: create -> deliver a few batches -> (wait for all | wait for some | wait for
none) -> destroy, many times in one process.  Checks every image that was waited for against the synchronous path, that the
delivery route stays what it was, and that threads, file descriptors and resident memory of the process do not grow (the SDMA
route starts two worker threads and two HSA signals per context: rr_sdma.cpp).
usage: soak_lifecycle.py [cycles] [seed]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
s = scenes.config_scene(2)
cfg = params.kaist_preset(n_reflections=2, n_samples=64, ambient_noise=2)
F = 4
noise = (np.random.RandomState(7).uniform(0, 1, (F, 400)) * 1000).astype(np.float32)
poses = scenes.trajectory(16, s["name"])


def make():
    c = native.Context(0)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
    c.set_config(cfg); c.set_beam_samples(golden_beams(64))
    return c


def usage():
    rss = 0
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS:"): rss = int(line.split()[1]) // 1024
    return len(os.listdir("/proc/self/task")), len(os.listdir("/proc/self/fd")), rss


c = make()
ref = {}
for f in range(F):
    c.set_noise_offsets(noise[f])
    for p in range(16):
        ref[(p, f)] = c.simulate(poses[p])[0].copy()
route0 = None
c.close()
streams = [torch.cuda.Stream() for _ in range(3)]
hosts = [native.HostImages((F, cfg.n_cells, 400)) for _ in range(6)]
bad = checked = dropped = 0
marks = {}
t0 = time.time()
for k in range(N):
    c = make(); c.set_noise_offsets(noise)
    nb = int(rng.randint(1, 7))
    sent = []
    for b in range(nb):
        p0 = int(rng.randint(0, 16))
        c.simulate_batch_host_async([poses[(p0 + f) % 16] for f in range(F)], hosts[b].ptr, streams[b % 3].cuda_stream)
        sent.append((b, p0))
    route = c.host_delivery_route()
    route0 = route0 or route
    assert route == route0, "the delivery route changed: %s -> %s (cycle %d)" % (route0, route, k)
    mode = int(rng.randint(0, 4))          # 0, 1: wait for all; 2: wait for some, drop the rest; 3: destroy at once
    if mode <= 1: c.wait_host(None); waited = sent
    elif mode == 2:
        waited = [x for x in sent if rng.rand() < 0.5]
        for (b, _) in waited: c.wait_host(hosts[b].ptr)
    else: waited = []
    for (b, p0) in waited:
        for f in range(F):
            bad += not np.array_equal(hosts[b].array[f], ref[((p0 + f) % 16, f)]); checked += 1
    dropped += len(sent) - len(waited)
    c.close()
    if k in (9, N - 1): marks[k] = usage()
dt = time.time() - t0
th0, fd0, rss0 = marks[9]; th1, fd1, rss1 = marks[N - 1]
leak = th1 > th0 + 2 or fd1 > fd0 + 4 or rss1 > rss0 + 256
print("life-cycle soak: %d contexts in %.1f s (route %s), %d images checked, mismatching: %d, batches dropped at destroy: %d; "
      "threads / fds / RSS MB after 10 cycles %d / %d / %d, after %d: %d / %d / %d%s"
      % (N, dt, route0, checked, bad, dropped, th0, fd0, rss0, N, th1, fd1, rss1, "  LEAK" if leak else ""))
sys.exit(1 if (bad or leak) else 0)
