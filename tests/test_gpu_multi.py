"""GPU tests of the multi-GPU paths (north_star: "shard azimuth columns across the 8 GPUs of one node with a
single RCCL gather over xGMI"; the fan-out it replaces: `#pragma omp parallel for`, RadarCPU.cpp:155).

  * the pipelined rr_multi calls (rr_multi_simulate_batch_async / rr_multi_wait): one device, and the n-device path in
    loopback on one GPU -- interleaved batches in flight, byte-equal to rr_simulate
  * an error inside a pipelined batch is reported once, the object is drained and healthy afterwards
  * tests that need TWO OR MORE GPUs (skipped on the one-GPU pool, live on an 8-GPU node): rr_create_multi with the
    real RCCL communicator on 2, 3 (ragged) and all devices; a spawned 2-rank nccl group through AzimuthShard, weak
    (all_to_all) and strong (all-gather), byte-equal to the single-GPU frames
"""
import os
import socket
import sys

import numpy as np
import pytest

from common import golden_beams, materials_for
from radarays_ros_amd import params, scenes

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _n_gpus():
    try:
        import torch
        return int(torch.cuda.device_count())       # (counting devices does not initialise the GPU)
    except Exception:
        return 0


need2 = pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs: the real RCCL calls (one-GPU pool: loopback tests cover the plan)")


def _setup(obj, s, cfg, mats, beams, noise=None, **kw):
    obj.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    obj.set_materials(mats, s["object_materials"], 0)
    obj.set_config(cfg, 400, **kw)
    obj.set_beam_samples(beams)
    if noise is not None:
        obj.set_noise_offsets(noise)


@pytest.fixture(scope="module")
def small():
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    cfg = params.kaist_preset(n_reflections=3, n_samples=60, ambient_noise=2)
    noise = (np.random.RandomState(5).uniform(0, 1, (4, 400)) * 1000.0).astype(np.float32)
    return s, cfg, materials_for(s), golden_beams(60), noise, scenes.trajectory(12, s["name"])


def _reference_frames(native_lib, small, poses_batches):
    s, cfg, mats, beams, noise, _ = small
    c = native_lib.Context(0)
    _setup(c, s, cfg, mats, beams, noise[0])
    out = []
    for batch in poses_batches:
        frames = []
        for f, p in enumerate(batch):
            c.set_noise_offsets(noise[f % 4])
            frames.append(c.simulate(p)[0])
        out.append(np.stack(frames))
    c.close()
    return out


@pytest.mark.parametrize("n_dev,threads", [(1, 0), (2, 0), (3, 0), (8, 0), (3, 1), (8, 1)])
def test_multi_async_batches_in_flight(native_lib, small, monkeypatch, n_dev, threads):
    """rr_multi_simulate_batch_async: 7 batches of different sizes issued back to back over a ring of 5 host buffers
    (more batches than slots: a call that finds its slot busy waits for the older batch itself), waited for in a
    scrambled order -- every image equals rr_simulate's.  n_dev = 1: the single-device route (deferred host copy);
    n_dev > 1: the n-device path in loopback (equal blocks 2 / 8, ragged 3); threads = 1: RR_MULTI_THREADS, one enqueue
    thread per device entry issues that entry's launches."""
    s, cfg, mats, beams, noise, poses = small
    if n_dev > 1:
        monkeypatch.setenv("RR_MULTI_LOOPBACK", "1")
    if threads:
        monkeypatch.setenv("RR_MULTI_THREADS", "1")
    m = native_lib.MultiContext([0] * n_dev)
    _setup(m, s, cfg, mats, beams, noise)
    sizes = [3, 1, 4, 2, 4, 1, 3]
    batches, k = [], 0
    for n in sizes:
        batches.append([poses[(k + j) % len(poses)] for j in range(n)]); k += n
    refs = _reference_frames(native_lib, small, batches)
    ring = [native_lib.HostImages((4, cfg.n_cells, 400)) for _ in range(5)]
    got = [None] * len(batches)
    inflight = {}

    def collect(slot):
        b = inflight.pop(slot)
        m.wait(ring[slot].ptr)
        got[b] = ring[slot].array[:len(batches[b])].copy()
    for b, batch in enumerate(batches):
        slot = b % len(ring)
        if slot in inflight:
            collect(slot)
        ring[slot].array[:] = 0xAB
        m.simulate_batch_async(batch, ring[slot].ptr)
        inflight[slot] = b
    for slot in sorted(inflight, key=lambda x: (x * 3) % 5):      # scrambled
        collect(slot)
    for b in range(len(batches)):
        assert np.array_equal(got[b], refs[b]), (n_dev, b)
    # the synchronous call still works in between / afterwards
    assert np.array_equal(m.simulate_batch(batches[2]), refs[2])
    m.wait(None)
    for h in ring:
        h.close()
    m.close()


@pytest.mark.parametrize("n_dev,threads", [(1, 0), (3, 0), (3, 1)])
def test_multi_async_error_is_reported_once_and_drains(native_lib, small, monkeypatch, n_dev, threads):
    """A batch that overflows its wave queue inside the pipeline: rr_multi_wait returns -7 for it; after the error return
    nothing is in flight (a late D2H copy must not hit a freed buffer, sticky bits must not fail the next
    call), and the next, healthy batch renders the right bytes."""
    s, cfg, mats, beams, noise, poses = small
    if n_dev > 1:
        monkeypatch.setenv("RR_MULTI_LOOPBACK", "1")
    if threads:
        monkeypatch.setenv("RR_MULTI_THREADS", "1")
    m = native_lib.MultiContext([0] * n_dev)
    _setup(m, s, cfg, mats, beams, noise)
    good = [poses[0], poses[1]]
    ref = _reference_frames(native_lib, small, [good])[0]
    h = [native_lib.HostImages((2, cfg.n_cells, 400)) for _ in range(3)]
    m.simulate_batch_async(good, h[0].ptr)
    m.set_config(cfg, 400, max_waves_per_azimuth=61)          # 60 beam samples: any split overflows
    m.simulate_batch_async(good, h[1].ptr)
    with pytest.raises(native_lib.RRError, match="capacity exceeded"):
        m.wait(None)
    m.wait(None)                                               # reported once
    m.set_config(cfg, 400)
    m.simulate_batch_async(good, h[2].ptr)
    m.wait(h[2].ptr)
    assert np.array_equal(h[2].array, ref)
    # an error invalidates every batch in flight: the drain reads and clears ALL lanes' error bits, so the
    # batches in flight beside the failing one report the error from the wait for their own buffers -- never a silent success
    m.simulate_batch_async(good, h[0].ptr)
    m.set_config(cfg, 400, max_waves_per_azimuth=61)
    m.simulate_batch_async(good, h[1].ptr)                     # overflows
    m.simulate_batch_async(good, h[2].ptr)                     # overflows too; its bits would be cleared by the drain
    m.wait(h[0].ptr)                                           # the healthy batch, waited for first: fine
    with pytest.raises(native_lib.RRError, match="capacity exceeded|overflow"):
        m.wait(h[1].ptr)
    with pytest.raises(native_lib.RRError, match="invalidates every batch in flight"):
        m.wait(h[2].ptr)
    m.wait(h[2].ptr); m.wait(None)                             # each reported once
    m.set_config(cfg, 400)
    m.simulate_batch_async(good, h[2].ptr)
    m.wait(None)
    assert np.array_equal(h[2].array, ref)
    assert np.array_equal(m.simulate_batch(good), ref)
    for x in h:
        x.close()
    m.close()


@pytest.mark.parametrize("mode", ["1", "2"])
def test_multi_real_rccl_calls_on_one_gpu(native_lib, small, monkeypatch, mode):
    """RR_MULTI_SELF_RCCL: what the loopback leaves out, on a one-GPU box -- librccl loaded at run time, ncclCommInitAll
    (one rank), ONE group of ncclSend / ncclRecv per call on the slot's stream (mode 1: the block as one piece, the equal
    plan; mode 2: one pair per frame, the ragged plan), ordered between the render and the transpose -- the device sends
    its block to itself.  Synchronous and pipelined calls, byte-equal to rr_simulate."""
    s, cfg, mats, beams, noise, poses = small
    monkeypatch.setenv("RR_MULTI_SELF_RCCL", mode)
    m = native_lib.MultiContext([0])
    monkeypatch.delenv("RR_MULTI_SELF_RCCL")
    _setup(m, s, cfg, mats, beams, noise)
    batches = [poses[0:4], poses[4:5], poses[5:8]]
    refs = _reference_frames(native_lib, small, batches)
    for b, batch in enumerate(batches):
        assert np.array_equal(m.simulate_batch(batch), refs[b]), (mode, b)
    ring = [native_lib.HostImages((4, cfg.n_cells, 400)) for _ in range(3)]
    for rep in range(4):                      # 12 pipelined batches over 4 slots
        for b, batch in enumerate(batches):
            m.wait(ring[b].ptr)
            if rep:
                assert np.array_equal(ring[b].array[:len(batch)], refs[b]), (mode, rep, b)
            m.simulate_batch_async(batch, ring[b].ptr)
    m.wait(None)
    for b, batch in enumerate(batches):
        assert np.array_equal(ring[b].array[:len(batch)], refs[b]), (mode, b)
    for h in ring:
        h.close()
    m.close()


# ---------------------------------------------------------------------------------------------------------------
# two or more GPUs: the real RCCL calls
# ---------------------------------------------------------------------------------------------------------------
@need2
@pytest.mark.parametrize("n_dev", [2, 3, 0])
def test_multi_real_rccl_equals_rr_simulate(native_lib, small, n_dev):
    """rr_create_multi over REAL devices: ncclCommInitAll, one group of ncclSend / ncclRecv pairs to the root per call
    (equal blocks: 2 and all devices when their number divides 400; ragged: 3), map built once and copied device to
    device (hipMemcpyPeer).  Every frame byte-equal to rr_simulate on device 0; synchronous and pipelined calls."""
    n = _n_gpus() if n_dev == 0 else n_dev
    if n > _n_gpus():
        pytest.skip("needs %d GPUs" % n)
    s, cfg, mats, beams, noise, poses = small
    m = native_lib.MultiContext(list(range(n)))
    assert m.device_count() == n
    _setup(m, s, cfg, mats, beams, noise)
    batches = [poses[0:4], poses[4:6], poses[6:9]]
    refs = _reference_frames(native_lib, small, batches)
    for b, batch in enumerate(batches):
        assert np.array_equal(m.simulate_batch(batch), refs[b]), (n, b)
    ring = [native_lib.HostImages((4, cfg.n_cells, 400)) for _ in range(3)]
    for rep in range(3):
        for b, batch in enumerate(batches):
            m.simulate_batch_async(batch, ring[b].ptr)
        m.wait(None)
        for b, batch in enumerate(batches):
            assert np.array_equal(ring[b].array[:len(batch)], refs[b]), (n, rep, b)
    for h in ring:
        h.close()
    m.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, strong, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from radarays_ros_amd import native
    from radarays_ros_amd.dist import AzimuthShard
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    cfg = params.kaist_preset(n_reflections=3, n_samples=60, ambient_noise=2)
    noise = (np.random.RandomState(5).uniform(0, 1, (4, 400)) * 1000.0).astype(np.float32)
    poses = scenes.trajectory(12, s["name"])
    c = native.Context(rank)
    _setup(c, s, cfg, materials_for(s), golden_beams(60), noise)
    fpr = 1 if strong else 2
    sh = AzimuthShard(c, cfg.n_cells, 400, rank, world, dev, n_slots=3, strong=strong, frames_per_rank=fpr, host_out=True)
    fps = sh.frames_per_step
    steps = [[poses[(k * fps + f) % len(poses)] for f in range(fps)] for k in range(5)]
    outs, host_ok = [], True
    for k, st in enumerate(steps):
        imgs = sh.step(st, None)
        sh.wait()
        torch.cuda.current_stream().synchronize()
        outs.append(imgs.cpu().numpy().copy())
        # host delivery (bench.py's bracket): strong = at once; weak = carried out by the step n_slots later on the same slot
        back = k if strong else k - 3
        if back >= 0:
            sh.slots[k % 3].stream.synchronize()
            h = sh.host_images(back)
            host_ok = host_ok and h is not None and np.array_equal(h.numpy(), outs[back])
    sh.flush_host()
    for k in range(len(steps) - 3, len(steps)):
        h = sh.host_images(k)
        host_ok = host_ok and h is not None and np.array_equal(h.numpy(), outs[k])
    sh.close()
    # reference: the same frames through the single-GPU synchronous path of THIS rank's context
    ok = True
    for k, st in enumerate(steps):
        mine = range(1) if strong else range(rank * fpr, rank * fpr + fpr)
        for j, f in enumerate(mine):
            c.set_noise_offsets(noise[f % 4] if not strong else noise[0])
            ref, _, _ = c.simulate(st[f])
            ok = ok and np.array_equal(outs[k][j], ref)
    np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array([int(ok and host_ok)]))
    c.close()
    dist.barrier()
    dist.destroy_process_group()


@need2
@pytest.mark.parametrize("strong", [False, True])
def test_two_rank_nccl_azimuth_shard(tmp_path, strong):
    """One process per GPU, backend nccl (= RCCL) over xGMI: the step loop of dist.py with world = 2 -- weak: 4 frames
    per step, ONE all_to_all_single, rank r ends up with frames 2r, 2r + 1; strong: one frame per step, ONE all-gather,
    every rank gets it.  Byte-equal to the single-GPU frames (noise row = index of the frame in its batch)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_shard_worker, args=(2, port, strong, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(os.path.join(str(tmp_path), "ok%d.npy" % r))[0] == 1, (strong, r)
