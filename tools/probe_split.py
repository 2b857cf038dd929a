"""One frame split into K azimuth blocks that run on K streams (frame lanes) at once, against the one-chain frame:
is a single simulate() shorter when its kernels' tails and launch gaps overlap?  Also the strong-scaling proxy
(one 400/N-column block alone on the GPU vs the whole frame) and what the D2H of one image costs.
usage: probe_split.py [config id] [passes]      (RR_LANES=8 for K = 8)"""
import sys, time, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ.setdefault("RR_LANES", "8")
import torch
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = scenes.config_scene(cid)
cfg = params.kaist_preset(n_reflections=npass, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
dev = torch.device("cuda", 0)
A, C = 400, cfg.n_cells
cols = torch.zeros((A, C), dtype=torch.uint8, device=dev)
img = torch.zeros((C, A), dtype=torch.uint8, device=dev)
ref, _, _ = c.simulate(poses[0])


def med(f, n=60, warm=8):
    for k in range(warm): f(k)
    ts = []
    for k in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(k); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    return "median %.3f ms (p10 %.3f, p90 %.3f)" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90))


print("rr_simulate (host image)          :", med(lambda k: c.simulate(poses[k % 16])))
main = torch.cuda.Stream(device=dev)


def dev_frame(k):
    c.simulate_device(poses[k % 16], img.data_ptr(), main.cuda_stream); main.synchronize()


print("rr_simulate_device + sync         :", med(dev_frame))
for K in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    evs = [torch.cuda.Event() for _ in range(K)]
    blocks = [native.partition(A, K, r) for r in range(K)]

    def split(k, K=K, streams=streams, evs=evs, blocks=blocks):
        for r in range(K):
            b, e = blocks[r]
            c.simulate_columns_device(poses[k % 16], b, e, cols.data_ptr() + b * C, None, streams[r].cuda_stream)
            evs[r].record(streams[r])
        for r in range(K): main.wait_event(evs[r])
        c.assemble_image_device(cols.data_ptr(), img.data_ptr(), main.cuda_stream)
        main.synchronize()
    print("K = %d blocks on %d streams + sync  :" % (K, K), med(split))
    split(0)
    assert np.array_equal(img.cpu().numpy(), ref), "split frame differs"
# strong-scaling proxy: one block of 400 / N columns alone on the GPU
one = torch.cuda.Stream(device=dev)
for N in (1, 2, 4, 8):
    b, e = native.partition(A, N, N // 2)

    def blk(k, b=b, e=e):
        c.simulate_columns_device(poses[k % 16], b, e, cols.data_ptr() + b * C, None, one.cuda_stream); one.synchronize()
    print("block of %3d columns alone        :" % (e - b), med(blk))
# the image's way home
pin = native.HostImages((C, A)); pg = np.zeros((C, A), np.uint8)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def d2h_pinned(k):
    hip.hipMemcpyAsync(ctypes.c_void_p(pin.ptr), ctypes.c_void_p(img.data_ptr()), ctypes.c_size_t(A * C), 2, ctypes.c_void_p(main.cuda_stream)); main.synchronize()
def d2h_pageable(k):
    hip.hipMemcpyAsync(ctypes.c_void_p(pg.ctypes.data), ctypes.c_void_p(img.data_ptr()), ctypes.c_size_t(A * C), 2, ctypes.c_void_p(main.cuda_stream)); main.synchronize()
def d2h_pinned_memcpy(k):
    d2h_pinned(k); np.copyto(pg, pin.array)
print("D2H 1.37 MB pinned                :", med(d2h_pinned))
print("D2H 1.37 MB pageable              :", med(d2h_pageable))
print("D2H pinned + host memcpy          :", med(d2h_pinned_memcpy))
c.close()
