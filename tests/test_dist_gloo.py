"""N > 1 path on CPU: world_size-2 gloo run of the azimuth sharding contract
(radarays_ros_amd/dist.py): partition -> per-rank column blocks -> ONE all-gather ->
[n_angles][n_cells] in azimuth order -> mono8 image.  Column blocks come from the
oracle here (no GPU); on the GPU box the same functions carry device tensors over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from radarays_ros_amd.dist import gather_columns, partition  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))


def test_partition_covers_all_azimuths():
    for n, w in [(400, 1), (400, 2), (400, 8), (400, 3), (7, 8), (401, 4)]:
        blocks = [partition(n, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        for (b0, e0), (b1, e1) in zip(blocks, blocks[1:]):
            assert e0 == b1 and e0 >= b0
        sizes = [e - b for b, e in blocks]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_angles, scroll, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from radarays_ros_amd import params, scenes
    from common import golden_beams, mats_tuple
    import gen_oracle_images as gen
    s = gen.two_room_scene()
    cfg = params.kaist_preset(n_reflections=2, ambient_noise=0, scroll_image=0, n_cells=512, resolution=0.1)
    sc = O.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
    mats = mats_tuple(params.kaist_materials() + [params.PENETRABLE])
    b, e = partition(n_angles, world, rank)
    # this rank's block, column-major [n_local][n_cells] (what rr_simulate_columns_device produces)
    u8, _, _ = O.simulate(sc, mats, s["object_materials"], cfg, golden_beams(8), scenes.default_pose("box12"),
                          az_begin=b, az_end=e, n_angles=n_angles, n_threads=1)
    block = torch.from_numpy(np.ascontiguousarray(u8[:, b:e].T))
    cols = gather_columns(block, n_angles, world)
    assert cols.shape == (n_angles, cfg.n_cells)
    # assemble like rr_assemble_image_device: img[c][(scroll + a) % A] = cols[a][c]
    img = torch.roll(cols.t().contiguous(), shifts=scroll, dims=1)
    if rank == 0:
        full, _, _ = O.simulate(sc, mats, s["object_materials"], cfg.copy(scroll_image=scroll), golden_beams(8),
                                scenes.default_pose("box12"), n_angles=n_angles, n_threads=1)
        np.save(os.path.join(out_dir, "ok.npy"), np.array([int(np.array_equal(img.numpy(), full))]))
    # every rank holds the same frame
    ref = img.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, img)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_angles,scroll", [(40, 0), (41, 5)])
def test_two_rank_gather_assembles_the_frame(tmp_path, oracle, n_angles, scroll):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_angles, scroll, str(tmp_path)), nprocs=2, join=True)
    assert np.load(os.path.join(str(tmp_path), "ok.npy"))[0] == 1


def _worker_weak(rank, world, port, n_angles, out_dir):
    """Weak-scaling step: `world` frames, my azimuth block of each, ONE all_to_all_single,
    I assemble frame number `rank` (the layout contract of AzimuthShard.step)."""
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from radarays_ros_amd import params, scenes
    from common import golden_beams, mats_tuple
    import gen_oracle_images as gen
    s = gen.two_room_scene()
    cfg = params.kaist_preset(n_reflections=2, ambient_noise=0, n_cells=512, resolution=0.1)
    sc = O.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
    mats = mats_tuple(params.kaist_materials() + [params.PENETRABLE])
    poses = scenes.trajectory(world, "box12")
    b, e = partition(n_angles, world, rank)
    blocks = []
    for f in range(world):          # what rr_simulate_batch_columns_device writes: [frame][n_local][n_cells]
        u8, _, _ = O.simulate(sc, mats, s["object_materials"], cfg, golden_beams(8), poses[f],
                              az_begin=b, az_end=e, n_angles=n_angles, n_threads=1)
        blocks.append(np.ascontiguousarray(u8[:, b:e].T))
    block = torch.from_numpy(np.stack(blocks))
    cols = torch.empty((n_angles, cfg.n_cells), dtype=torch.uint8)
    dist.all_to_all_single(cols.view(-1), block.view(-1))
    img = cols.t().contiguous().numpy()
    full, _, _ = O.simulate(sc, mats, s["object_materials"], cfg, golden_beams(8), poses[rank],
                            n_angles=n_angles, n_threads=1)
    np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array([int(np.array_equal(img, full))]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_weak_step_all_to_all(tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker_weak, args=(2, port, 40, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(os.path.join(str(tmp_path), "ok%d.npy" % r))[0] == 1


class _MockCtx:
    """Stands in for native.Context in the CPU run of AzimuthShard: same entry points, same
    pointer/stride contracts, columns computed by the oracle (test infrastructure)."""

    def __init__(self, O, sc, mats, objm, cfg, beams, n_angles):
        self.O, self.sc, self.mats, self.objm, self.cfg, self.beams, self.A = O, sc, mats, objm, cfg, beams, n_angles
        self.C = cfg.n_cells

    def _view(self, ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr))

    def _cols(self, pose, b, e):
        u8, _, _ = self.O.simulate(self.sc, self.mats, self.objm, self.cfg.copy(scroll_image=0), self.beams, pose,
                                   az_begin=b, az_end=e, n_angles=self.A, n_threads=1)
        return np.ascontiguousarray(u8[:, b:e].T)

    def simulate_columns_device(self, pose, b, e, ptr, f32, sp):
        self._view(ptr, (e - b) * self.C)[:] = self._cols(pose, b, e).ravel()

    def simulate_batch_columns_device(self, poses, b, e, ptr, sp):
        n = (e - b) * self.C
        v = self._view(ptr, len(poses) * n)
        for f, p in enumerate(poses):
            v[f * n:(f + 1) * n] = self._cols(p, b, e).ravel()

    def simulate_batch_columns_carry_device(self, poses, b, e, ptr, sp, src, dst, nbytes):
        self._view(dst, nbytes)[:] = self._view(src, nbytes)       # (the real call trickles it out on its trace launches)
        self.carried = getattr(self, "carried", 0) + 1
        self.simulate_batch_columns_device(poses, b, e, ptr, sp)

    def assemble_blocks_device(self, ptr, n_loc, stride, img_ptr, sp):
        img = self._view(img_ptr, self.C * self.A).reshape(self.C, self.A)
        for a in range(self.A):
            col = self._view(ptr + (a // n_loc) * stride + (a % n_loc) * self.C, self.C)
            img[:, (self.cfg.scroll_image + a) % self.A] = col

    def assemble_image_device(self, ptr, img_ptr, sp):
        self.assemble_blocks_device(ptr, self.A, self.A * self.C, img_ptr, sp)

    def assemble_frames_device(self, ptr, n_loc, stride, n_frames, frame_stride, imgs_ptr, sp):
        for j in range(n_frames):
            self.assemble_blocks_device(ptr + j * frame_stride, n_loc, stride, imgs_ptr + j * self.C * self.A, sp)


def _worker_shard(rank, world, port, out_dir):
    """The REAL AzimuthShard step loop (weak fpr=2 and strong) on 2 gloo ranks with a mock context."""
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from radarays_ros_amd import params, scenes
    from radarays_ros_amd.dist import AzimuthShard
    from common import golden_beams, mats_tuple
    import gen_oracle_images as gen
    A = 40
    s = gen.two_room_scene()
    cfg = params.kaist_preset(n_reflections=2, ambient_noise=0, n_cells=256, resolution=0.2, scroll_image=3)
    sc = O.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
    mats = mats_tuple(params.kaist_materials() + [params.PENETRABLE])
    ctx = _MockCtx(O, sc, mats, s["object_materials"], cfg, golden_beams(6), A)
    poses = scenes.trajectory(8, "box12")
    full = lambda p: O.simulate(sc, mats, s["object_materials"], cfg, golden_beams(6), p, n_angles=A, n_threads=1)[0]
    ok = 1
    dev = torch.device("cpu")
    weak = AzimuthShard(ctx, cfg.n_cells, A, rank, world, dev, n_slots=2, frames_per_rank=2)
    assert weak.frames_per_step == 4
    for k in range(3):                                   # 3 steps over 2 slots: slot reuse included
        step_poses = [poses[(k * 4 + f) % 8] for f in range(4)]
        imgs = weak.step(step_poses)
        for j in range(2):                               # I own frames rank*2 + j of the step
            ok &= int(np.array_equal(imgs[j].numpy(), full(step_poses[rank * 2 + j])))
    # host_out: every frame also reaches host memory -- the images of step k ride out with step k + n_slots on the same slot,
    # flush_host() sends the rest
    hosty = AzimuthShard(ctx, cfg.n_cells, A, rank, world, dev, n_slots=2, frames_per_rank=2, host_out=True)
    kept = {}
    for k in range(5):
        step_poses = [poses[(k * 4 + f) % 8] for f in range(4)]
        kept[k] = [full(step_poses[rank * 2 + j]) for j in range(2)]
        hosty.step(step_poses)
        if k >= 2:                                       # step k - 2 has been carried out by this call
            h = hosty.host_images(k - 2)
            ok &= int(h is not None and all(np.array_equal(h[j].numpy(), kept[k - 2][j]) for j in range(2)))
    ok &= int(ctx.carried == 3 and hosty.host_images(4) is None)
    hosty.flush_host()
    for k in (3, 4):
        h = hosty.host_images(k)
        ok &= int(h is not None and all(np.array_equal(h[j].numpy(), kept[k][j]) for j in range(2)))
    strong = AzimuthShard(ctx, cfg.n_cells, A, rank, world, dev, n_slots=2, strong=True)
    assert strong.frames_per_step == 1
    for k in range(2):
        img = strong.step([poses[k]])
        ok &= int(np.array_equal(img[0].numpy(), full(poses[k])))
    ragged = AzimuthShard(ctx.__class__(O, sc, mats, s["object_materials"], cfg, golden_beams(6), 41), cfg.n_cells, 41,
                          rank, world, dev)            # 41 % 2 != 0 -> falls back to the gather mode
    assert ragged.strong and ragged.frames_per_step == 1
    img = ragged.step([poses[5]])
    want = O.simulate(sc, mats, s["object_materials"], cfg, golden_beams(6), poses[5], n_angles=41, n_threads=1)[0]
    ok &= int(np.array_equal(img[0].numpy(), want))
    np.save(os.path.join(out_dir, "shard%d.npy" % rank), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_azimuth_shard_step_loop(tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker_shard, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(os.path.join(str(tmp_path), "shard%d.npy" % r))[0] == 1


def _worker_rccl_block(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    desc = {"device_index": rank, "name": "AMD Instinct MI355X", "pci_bus_id": "0000:%02x:00" % (5 + 8 * rank), "uuid": None, "pid": os.getpid()}
    blk = bench.rccl_block(rank, world, desc, "all_to_all_single (one per batch: frames d*F.. -> rank d)", 16 * 200 * 3424,
                           4000.0 + 100.0 * (1 - rank))
    if rank == 0:
        import json
        with open(os.path.join(out_dir, "rccl.json"), "w") as f:
            json.dump(blk, f)
    else:
        assert blk is None
    dist.barrier()
    dist.destroy_process_group()


def test_bench_rccl_block_schema_on_two_ranks(tmp_path):
    """The `rccl` object of the N > 1 bench line, built by bench.rccl_block on a world-2 gloo group:
    world size and backend come from torch.distributed, one entry per rank with its device, per-rank rates, slowest rank."""
    import json
    port = _free_port()
    mp.spawn(_worker_rccl_block, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    b = json.load(open(os.path.join(str(tmp_path), "rccl.json")))
    assert set(b) == {"world_size", "backend", "ranks", "distinct_devices", "collective", "bytes_per_collective",
                      "per_rank_images_per_s", "slowest_rank", "what"}
    assert b["world_size"] == 2 and b["backend"] == "gloo" and b["distinct_devices"] == 2
    assert [r["rank"] for r in b["ranks"]] == [0, 1] and [r["device_index"] for r in b["ranks"]] == [0, 1]
    assert all(set(r) == {"rank", "device_index", "name", "pci_bus_id", "uuid", "pid"} for r in b["ranks"])
    assert b["ranks"][0]["pid"] != b["ranks"][1]["pid"]                  # one process per GPU
    assert b["per_rank_images_per_s"] == [4100.0, 4000.0] and b["slowest_rank"] == 1
    assert b["bytes_per_collective"] == 16 * 200 * 3424 and b["collective"].startswith("all_to_all_single")


def test_bench_n1_reference_reads_profiles():
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    r = bench.n1_reference("target_10M_400x200_4pass")
    assert r is not None and r["value"] > 1000 and r["source"].startswith("profiles/")
    assert bench.n1_reference("no_such_workload") is None


def test_bench_keeps_torchs_runtime_unless_asked_and_for_n_above_one(monkeypatch):
    """bench.py's RR_BENCH_SYSTEM_HIP switch (which engine copies): off by default, never for N > 1, never
    once torch is in the process (the runtime that was loaded first serves the process)."""
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    import torch  # noqa: F401  (as in every test process)
    monkeypatch.delenv("RR_BENCH_SYSTEM_HIP", raising=False)
    assert bench.prefer_system_hip_runtime(1).startswith("torch wheel")
    monkeypatch.setenv("RR_BENCH_SYSTEM_HIP", "1")
    assert bench.prefer_system_hip_runtime(8).startswith("torch wheel")
    assert bench.prefer_system_hip_runtime(1).startswith("torch wheel")      # torch is already imported here


class _BenchMockCtx:
    """native.Context's surface as bench.main() uses it, with canned counters and columns that only say who rendered them --
    for the CPU dry run of the N > 1 bench (no oracle: the numbers of a dry run mean nothing, its control flow is the point)."""

    def __init__(self, local_rank):
        self.rank = local_rank
        self.calls = {"batch": 0, "carry": 0, "assemble": 0}
        self.C, self.A = 3424, 400

    def _view(self, ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr))

    def set_mesh(self, *a, **k): pass
    def set_materials(self, *a, **k): pass
    def set_config(self, cfg, n_angles=400, **k): self.C, self.A = cfg.n_cells, n_angles
    def set_beam_samples(self, *a): pass
    def set_noise_offsets(self, *a): pass
    def set_stats_mode(self, on): pass
    def set_timing_mode(self, on): pass
    def reserve_timing_events(self, n): pass
    def close(self): pass
    def stats(self): return {"wave_passes": 80000 * 8, "hits": 1, "signals": 1, "nodes_visited": 10 ** 7, "tris_tested": 10 ** 6, "overflow": 0}
    def traversal_shape(self): return {"waves": 5000, "iterations": 100000, "node_path_issues": 90000, "leaf_path_issues": 30000,
                                       "live_quad_steps": 1200000, "max_iterations": 60, "node_steps": 900000, "leaf_steps": 300000}
    def kernel_time(self, name, reset=False): return (1.0, 4)

    def simulate_batch_columns_device(self, poses, b, e, ptr, sp):
        self.calls["batch"] += 1
        self._view(ptr, len(poses) * (e - b) * self.C)[:] = 10 + self.rank

    def simulate_batch_columns_carry_device(self, poses, b, e, ptr, sp, src, dst, nbytes):
        self.calls["carry"] += 1
        self._view(dst, nbytes)[:] = self._view(src, nbytes)
        self.simulate_batch_columns_device(poses, b, e, ptr, sp)

    def simulate_columns_device(self, pose, b, e, ptr, f32, sp):
        self._view(ptr, (e - b) * self.C)[:] = 10 + self.rank

    def assemble_frames_device(self, ptr, n_loc, stride, n_frames, frame_stride, imgs_ptr, sp):
        self.calls["assemble"] += 1
        self._view(imgs_ptr, n_frames * self.C * self.A)[:] = 1

    def assemble_image_device(self, ptr, img_ptr, sp):
        self._view(img_ptr, self.C * self.A)[:] = 1


def _worker_bench_dryrun(rank, world, port, out_dir, fail_rank):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank), "RR_BENCH_DRYRUN": "1", "RR_BENCH_ERRDIR": os.path.join(out_dir, "err")})
    import bench

    def make(local_rank):
        c = _BenchMockCtx(local_rank)
        if local_rank == fail_rank:
            def boom(*a, **k):
                raise RuntimeError("device %d: wave/signal queue capacity exceeded (injected)" % local_rank)
            c.simulate_batch_columns_device = boom
        return c
    bench.TEST_HOOKS = {"context": make}
    fd = os.open(os.path.join(out_dir, "stdout_rank%d.txt" % rank), os.O_CREAT | os.O_WRONLY | os.O_TRUNC)
    os.dup2(fd, 1)                       # what a launcher would see on the rank's stdout
    try:
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--workload", "config2_100k_400x200_1pass"])
    except BaseException as exc:         # noqa: BLE001  what `python bench.py` does at top level
        bench.report_failure(exc)
        os._exit(7 if rank == fail_rank else 0)       # (the other rank would hang in its collective: a launcher kills it)


def test_bench_two_rank_dry_run_prints_the_full_line(tmp_path):
    """`bench.py --gpus 2` end to end on the CPU: two gloo ranks, a mock context, the REAL main() -- rank / barrier / all_reduce
    flow, the sharded step loop with host delivery, the rccl block, n1_reference, the line's assembly.  ONE line on rank 0's
    stdout, nothing on rank 1's."""
    import json
    port = _free_port()
    mp.spawn(_worker_bench_dryrun, args=(2, port, str(tmp_path), -1), nprocs=2, join=True)
    out0 = open(os.path.join(str(tmp_path), "stdout_rank0.txt")).read().strip().splitlines()
    assert len(out0) == 1 and open(os.path.join(str(tmp_path), "stdout_rank1.txt")).read().strip() == ""
    d = json.loads(out0[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["frames_per_batch"] == 16
    r = d["rccl"]
    assert r["world_size"] == 2 and r["backend"] == "gloo" and len(r["ranks"]) == 2 and len(r["per_rank_images_per_s"]) == 2
    assert all(x > 0 for x in r["per_rank_images_per_s"]) and r["collective"].startswith("all_to_all_single")
    assert r["bytes_per_collective"] == 16 * 200 * 3424
    assert d["n1_reference"] is not None and d["n1_reference"]["source"].startswith("profiles/")
    assert d["roofline"] is not None and "cpu_baseline" not in d and d["hbm_resident"] is None
    assert abs(d["value"] - 3 * 32 / (1e-3 * d["ms_per_step"] * 3)) < 1e-6 * d["value"] + 0.01       # value = all ranks' frames / slowest rank's time


def test_bench_rank_failure_prints_one_error_line(tmp_path):
    """a rank that fails inside its step loop -> ONE JSON line carrying "error" (the failing layer's text)
    and the failing rank, a non-zero exit code; no second line from anybody."""
    import json
    port = _free_port()
    ctx = mp.spawn(_worker_bench_dryrun, args=(2, port, str(tmp_path), 1), nprocs=2, join=False)
    import time
    t0 = time.time()
    while time.time() - t0 < 120 and any(p.is_alive() for p in ctx.processes):
        if not ctx.processes[1].is_alive():              # the launcher's part: the failed rank takes the others down
            for p in ctx.processes:
                if p.is_alive():
                    p.terminate()
        time.sleep(0.2)
    for p in ctx.processes:
        p.join(10)
    assert ctx.processes[1].exitcode == 7
    lines = []
    for r in (0, 1):
        lines += [x for x in open(os.path.join(str(tmp_path), "stdout_rank%d.txt" % r)).read().strip().splitlines() if x]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["value"] is None and d["failing_rank"] == 1 and d["n_gpus"] == 2
    assert "queue capacity exceeded (injected)" in d["error"] and d["error"].startswith("RuntimeError")
    rec = json.load(open(os.path.join(str(tmp_path), "err", "rank_1.json")))
    assert rec["rank"] == 1 and "Traceback" in rec["traceback"]
