#!/usr/bin/env python3
"""bench.py -- polar images/s of the radar hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

Workload (default): the configuration BASELINE.json's north_star target is quoted on --
400 azimuths x 3424 range bins, 200 rays per beam, 4 ray-cast passes (Snell/Fresnel split), 10M-triangle
synthetic mesh, KAIST parameter preset (cfg/mulran_kaist_dyncfg.yaml) including the Perlin ambient-noise stage
with injected per-column offsets (fresh row per frame).  `--workload` selects the other BASELINE configs.

A step = ONE pass over the 16-pose trajectory: every GPU finishes 16 polar images per step, rendered as two
batches of 8 poses (one set of launches each; `--frames-per-rank` poses per batch, `--batches-per-step` batches),
3 batches in flight on 3 streams (`--slots`; 4 where a batch is fewer than 8 frames).  Mesh, BVH, parameters and beam samples are resident in HBM before the timed
region; poses are 7 floats passed as kernel arguments.  Timing bracket = the reference's stopwatch
(RadarCPU.cpp:147-148 -> :550).

`value` (round 5): the bracket of SURVEY §8(d) = the reference's stopwatch -- the reference's simulate() ends with the image in
HOST memory (m_polar_image, RadarCPU.cpp:542,555-561), so the K steps are timed THROUGH the last D2H copy of every image into
page-locked host memory (N = 1: rr_simulate_batch_host_async; N > 1: every rank delivers the frames it assembled,
AzimuthShard(host_out=True); either way the images ride out on the trace launches of the next batch on the same stream, a few
waves with one store in flight each).  Inputs (mesh, tree, parameters, beam) are resident in HBM before the timed region.
The same line carries `hbm_resident` (the same steps with the images left in HBM: `value` of rounds 1-4) and `single_pose`: one pose per
launch set (the latency-oriented shape a live ROS node would use).  Before every timed region the GPU is
pre-warmed by wall time (>= 0.3 s of steps, untimed) so that `--steps 20` reads sustained clocks.

N > 1 (north_star): the 400 azimuth columns of EVERY frame are sharded over the ranks (400/N columns each) and
assembled by ONE RCCL collective over xGMI per batch (default, "scaling": "weak"): a batch renders N*F frames;
rank r simulates its 400/N-column block of all of them in one set of launches and ONE all_to_all_single (the
per-frame gathers fused) hands frames d*F.. to rank d -- per-GPU work per step is constant, value = N*16*K/t.
`--strong`: one frame per batch + one all-gather (latency mode).

`python bench.py --gpus N` with no launcher around it starts its own N ranks (a `torch.distributed.run` child, before this
process touches the GPU) and exits 3 with one line when the box has fewer than N GPUs.  stdout carries the JSON line and
nothing else (RCCL's version banner is sent to stderr).

N > 1 also carries `rccl` -- what a reader needs to believe RCCL saw N ranks: world size and backend as torch.distributed
reports them, every rank's device (index, name, PCI bus id), the collective and its bytes, every rank's own images/s and the
slowest rank -- and `n1_reference`, the N = 1 figure of the same workload from profiles/.

Also on the line: `weak_scaling_proxy` (the per-step work of rank 0 of N -- its 400/N-column block of N x F frames and the
transpose of its F frames -- on this one GPU against the N = 1 step: the per-GPU side of weak scaling), `single_frame_sync` (ONE synchronous rr_simulate per frame: the reference's own call shape,
radar_simulator.cpp:197-212) and `strong_scaling_proxy` (a block of 400/N azimuth columns alone on the GPU against the
whole frame: the bound of what sharding ONE frame over N GPUs can win; the default N > 1 mode is weak scaling).

Extra objects on the JSON line:
  "roofline"     the kernel that takes the most GPU time (picked from the measured per-kernel times, normally the
                 later-pass k_trace) against the bound that really limits it: VALU instruction issue.
                 achieved = wave-instructions/s = SQ_INSTS_VALU per launch (rocprofv3 PMC of this same command,
                 profiles/roofline_counters.json) / the launch duration measured live in THIS run (hipExtLaunchKernel
                 begin/end events on the launch stream); peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64
                 VALU instruction (SIMD-32) = 1.2288 T/s.  HBM traffic (PMC) and the algorithmic byte rate are
                 secondary keys -- the tree is served by L1/L2, HBM runs at a few percent of its peak.
                 `useful_issue_frac`: the share of the issued lane-work that advanced a ray (statistics build).  The
                 counters file carries a hash of the kernel sources: when it does not match, `frac` is withheld.
  "cpu_baseline" (N = 1, rank 0) the CPU oracle = line-faithful port of RadarCPU::simulate, OpenMP over azimuths
                 like RadarCPU.cpp:155, timed on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# MI355X_MICROARCH.md "Wave scheduling": 4 SIMD-32 per CU, a wave64 VALU instruction issues over 2 cycles
VALU_PEAK_WAVE_INSTR_S = 256 * 4 * 2.4e9 / 2.0
PREWARM_S = 0.3


def prefer_system_hip_runtime(world):
    """RR_BENCH_SYSTEM_HIP=1, N = 1: load the image's HIP runtime (/opt/rocm/lib/libamdhip64.so.7, ROCm 7.2 -- what the
    library is built and linked against, and what a C++ caller such as the reference's node runs on) into the global
    symbol scope BEFORE torch, so that it, not the ROCm 7.0.2 runtime bundled inside the torch wheel, serves this
    process.  Why it matters: the bundled runtime carries every hipMemcpyAsync to page-locked host memory as a blit
    KERNEL (`__amd_rocclr_copyBuffer`, competing with the simulation's kernels for the CUs), the system runtime as an SDMA
    transfer (54 GB/s beside a chip-filling kernel, which it slows by 0.5 %: tools/d2h_engine.hip, tools/d2h_py.py).
    N > 1 keeps torch's own runtime: its bundled RCCL has only ever been run with that one.  Returns what is in force."""
    path = "/opt/rocm/lib/libamdhip64.so.7"
    if os.environ.get("RR_BENCH_SYSTEM_HIP", "0") != "1" or world != 1 or "torch" in sys.modules or not os.path.exists(path):
        return "torch wheel (bundled)"
    import ctypes
    try:
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return "torch wheel (bundled)"
    return path


def algorithmic_bytes_per_wave_pass(n_tris):
    """structure-independent floor D*64 + 4*48 + 132 with
    D = ceil(log2(T/4)) BVH levels, 4 triangles tested, 132 B of wave state."""
    d = int(math.ceil(math.log2(max(n_tris, 8) / 4.0)))
    return d * 64 + 4 * 48 + 132


WORKLOADS = {
    # name: (scene config id, n_reflections, rays/beam)
    "config2_100k_400x200_1pass": (2, 1, 200),
    "config3_1M_400x200_4pass": (3, 4, 200),
    "config4_10M_400x1000_4pass": (4, 4, 1000),
    "target_10M_400x200_4pass": (4, 4, 200),
    # configs[4]: per-triangle materials (8), 8 passes, 10M triangles, Cook-Torrance lobe (rr_config.brdf_model = 1:
    # the build's own GGX / Smith specification -- the reference's model lives on its dev/flex branch, not in the
    # checkout, so this one config is parity-unpinned); run with --frames-per-rank 1
    "config5_10M_400x1000_8pass_pertri": (5, 8, 1000),
}
KERNEL_LABEL = {"trace0": "k_trace<FIRST> (pass 0)", "trace": "k_trace (passes 1..P-1)", "shade": "k_shade",
                "scan": "k_scan", "column": "k_column", "assemble": "k_assemble_u8x4"}


def pct(v, q):
    v = np.sort(np.asarray(v, np.float64))
    return float(v[min(len(v) - 1, int(q * len(v)))]) if len(v) else 0.0


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources that decide what the kernels execute: the kernels, the device layout,
    the tree builders.  profiles/make_counters.py stores it beside the PMC counters."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rr_kernels.hip", "rr_device.h", "rr_bvh.h", "rr_bvh.cpp", "rr_lbvh.hip"):
        with open(os.path.join(ROOT, "radarays_ros_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def visible_gpus():
    """Number of GPUs this process could use, WITHOUT initialising the HIP runtime: the KFD topology in sysfs
    (a node with SIMDs is a GPU), narrowed by HIP_/ROCR_VISIBLE_DEVICES; torch.cuda.device_count() -- which does not
    initialise the GPU on this image -- where sysfs is not readable."""
    n = None
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(top):
            with open(os.path.join(top, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        n = None
    if not n:
        import torch
        return int(torch.cuda.device_count())
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


_JSON_OUT = None          # where the ONE line goes once main() has moved file descriptor 1 out of the way


def _errdir():
    """Where the ranks of one launch leave their failure records: given by the self-launching parent, else keyed by the
    launcher's pid (the ranks of a torch.distributed.run share their parent)."""
    import tempfile
    d = os.environ.get("RR_BENCH_ERRDIR") or os.path.join(tempfile.gettempdir(), "rr_bench_err_%d" % os.getppid())
    os.makedirs(d, exist_ok=True)
    return d


def error_line(world, rank, message, extra=None):
    """The ONE line a failed run prints instead of a result (same keys a reader of the result line looks at first)."""
    out = {"metric": "polar images/sec (400 az x 3424 bins)", "value": None, "unit": "images/s", "n_gpus": int(world),
           "error": message, "failing_rank": rank}
    if extra:
        out.update(extra)
    return out


def report_failure(exc):
    """A rank failed: leave a record for the launcher and -- the first rank to get here only -- print the error line.  The
    message carries what the failing layer said: RRError texts are rr_last_error / rr_multi_last_error, torch.distributed
    errors carry RCCL's own string."""
    import traceback
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    msg = "%s: %s" % (type(exc).__name__, exc)
    d = _errdir()
    with open(os.path.join(d, "rank_%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "error": msg, "traceback": traceback.format_exc()[-4000:]}, f)
    try:
        os.close(os.open(os.path.join(d, "printed"), os.O_CREAT | os.O_EXCL | os.O_WRONLY))
    except FileExistsError:
        return
    line = json.dumps(error_line(world, rank, msg)) + "\n"
    if _JSON_OUT is not None:
        _JSON_OUT.write(line); _JSON_OUT.flush()
    else:
        sys.stdout.write(line); sys.stdout.flush()


def self_launch(n_gpus):
    """Runs this very command under `python -m torch.distributed.run` with one rank per GPU (child process; its
    rank 0 prints the JSON line on our stdout) and returns its exit code.  Fewer GPUs than asked for: one clear
    line, non-zero, at once."""
    import socket
    import subprocess
    have = visible_gpus()
    if have < n_gpus:
        print("bench.py: --gpus %d asked for, but this box shows %d GPU%s -- not launched" % (n_gpus, have, "" if have == 1 else "s"),
              file=sys.stderr, flush=True)
        return 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    import shutil
    import tempfile
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    errdir = tempfile.mkdtemp(prefix="rr_bench_err_")
    env["RR_BENCH_ERRDIR"] = errdir
    rc = subprocess.call(cmd, env=env)
    if rc != 0 and not os.path.exists(os.path.join(errdir, "printed")):
        # no rank got as far as reporting (killed by a signal, a crash below python): the launcher's code is all there is
        recs = []
        for f in sorted(os.listdir(errdir)):
            if f.startswith("rank_"):
                try:
                    recs.append(json.load(open(os.path.join(errdir, f))))
                except (OSError, ValueError):
                    pass
        print(json.dumps(error_line(n_gpus, recs[0]["rank"] if recs else None,
                                    recs[0]["error"] if recs else "a rank ended without a python exception (launcher exit code %d)" % rc,
                                    {"launcher_rc": rc})), flush=True)
    shutil.rmtree(errdir, ignore_errors=True)
    return rc


def rccl_block(rank, world, device_desc, collective, bytes_per_collective, own_images_per_s, group=None):
    """The `rccl` object of the N > 1 line (every rank calls it; rank 0 gets the dict, the others None).  Uses only
    torch.distributed object collectives, so the world-2 gloo test on the CPU runs this very function."""
    import torch.distributed as dist
    mine = dict(device_desc)
    mine["rank"] = int(rank)
    mine["images_per_s"] = round(float(own_images_per_s), 2)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    if rank != 0:
        return None
    everyone = sorted(everyone, key=lambda d: d["rank"])
    rates = [d.pop("images_per_s") for d in everyone]
    return {"world_size": int(dist.get_world_size(group)), "backend": str(dist.get_backend(group)),
            "ranks": everyone, "distinct_devices": len({(d.get("pci_bus_id"), d.get("device_index")) for d in everyone}),
            "collective": collective, "bytes_per_collective": int(bytes_per_collective),
            "per_rank_images_per_s": rates, "slowest_rank": int(min(range(world), key=lambda r: rates[r])),
            "what": "world size and backend as torch.distributed reports them; one entry per rank with the device it ran on; "
                    "per_rank_images_per_s = the frames a rank assembled and delivered / its OWN wall time for the timed steps "
                    "(value uses the slowest rank's time)"}


def n1_reference(workload):
    """The N = 1 figure of the same workload recorded under profiles/ (newest round first), for the N > 1 line."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_%s.json" % workload)), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("n_gpus") == 1:
                return {"value": d["value"], "unit": d.get("unit"), "images": d.get("config", {}).get("images"),
                        "source": os.path.relpath(f, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return None


# ---- CPU dry run of the N > 1 control flow (tests/test_dist_gloo.py; never on a GPU box) --------------------------------
# RR_BENCH_DRYRUN=1 + TEST_HOOKS = {"context": factory(local_rank) -> object with native.Context's methods}: main() runs on
# the gloo backend with CPU tensors and this stand-in for the few torch.cuda calls it makes, so that the rank / barrier /
# all_reduce / rccl-block / line-assembly code of an 8-GPU run is EXECUTED before the first such run exists.  The numbers it
# prints mean nothing; the product path is untouched (the flag is refused when a HIP device is visible).
TEST_HOOKS = None


class _CpuCuda:
    class Event:
        def __init__(self, enable_timing=False):
            self.t = None

        def record(self, stream=None):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return 1e3 * (other.t - self.t)

    class Stream:
        cuda_stream = None

        def __init__(self, device=None):
            pass

        def synchronize(self):
            pass

    @staticmethod
    def synchronize():
        pass

    @staticmethod
    def current_stream():
        return _CpuCuda.Stream()

    @staticmethod
    def set_device(i):
        pass

    @staticmethod
    def is_available():
        return True

    @staticmethod
    def get_device_properties(i):
        class P:
            name = "cpu dry run"
        return P()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="target_10M_400x200_4pass", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the hbm_resident / single_pose regions")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget")
    ap.add_argument("--ambient-noise", type=int, default=2)
    ap.add_argument("--strong", action="store_true", help="N>1: one frame per batch + all-gather")
    ap.add_argument("--self-launch", action="store_true",
                    help="go through the torch.distributed.run child even for N = 1 (exercises the launcher path on a one-GPU box)")
    ap.add_argument("--force-slots", action="store_true", help="run the N>1 step loop (with its collective) on one rank (debug)")
    ap.add_argument("--slots", type=int, default=None,
                    help="batches in flight (streams + buffer sets; RR_LANES must be >= slots).  Default: 3 for batches of >= 8 frames per GPU, "
                         "else 4 -- the fewest that hold the rate (round 6, profiles/r06_experiments.txt: the target reads the same images/s "
                         "from 2 to 6; a launch's live duration, and with it roofline.frac, only says how many batches share the chip)")
    ap.add_argument("--frames-per-rank", type=int, default=8,
                    help="frames each GPU finishes per batch (one set of launches); a batch = N x this many frames")
    ap.add_argument("--batches-per-step", type=int, default=2, help="batches per step (default: 2 x 8 = the 16-pose trajectory)")
    args = ap.parse_args(argv)
    if args.slots is None:
        args.slots = 3 if (args.frames_per_rank >= 8 and not args.strong) else 4

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and (args.gpus > 1 or args.self_launch):
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, BEFORE anything in this
        # process touches the GPU (the parent only counts devices and waits for the child)
        raise SystemExit(self_launch(args.gpus))
    # stdout carries ONE line, the JSON line: whatever else writes to file descriptor 1 from here on -- RCCL prints a
    # five-line version banner there when a communicator is made, from C, flushed at exit, i.e. AFTER the line -- goes to stderr
    global _JSON_OUT
    json_out = _JSON_OUT = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    if world != args.gpus:
        args.gpus = world

    runtime = prefer_system_hip_runtime(world)
    import torch
    import torch.distributed as dist
    from radarays_ros_amd import native, params, scenes
    from radarays_ros_amd.dist import AzimuthShard
    from radarays_ros_amd.fixtures import golden_beams, materials_for

    dry = os.environ.get("RR_BENCH_DRYRUN", "0") == "1" and TEST_HOOKS is not None
    if dry and torch.cuda.is_available():
        raise SystemExit("bench.py: RR_BENCH_DRYRUN is a CPU-only test switch")
    tc = _CpuCuda if dry else torch.cuda
    if not tc.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    tc.set_device(local_rank)
    if world > 1 or args.force_slots:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    scene_id, n_pass, n_rays = WORKLOADS[args.workload]
    scene = scenes.config_scene(scene_id)
    cfg = params.kaist_preset(n_reflections=n_pass, n_samples=n_rays, ambient_noise=args.ambient_noise)
    mats = materials_for(scene)
    beams = golden_beams(n_rays)
    # fresh offsets per frame of a batch like the reference draws them (RadarCPU.cpp:461-472): 16 rows, frame f -> row f % 16
    noise = (np.random.RandomState(7).uniform(0, 1, 16 * params.N_ANGLES) * 1000.0).astype(np.float32)
    poses = scenes.trajectory(16, scene["name"])

    ctx = TEST_HOOKS["context"](local_rank) if dry else native.Context(local_rank)
    ctx.set_mesh(scene["verts"], scene["faces"], scene["face_object_id"])
    ctx.set_materials(mats, scene["object_materials"], 0)
    ctx.set_config(cfg, params.N_ANGLES, brdf_model=1 if scene_id == 5 else 0)
    ctx.set_beam_samples(beams)
    ctx.set_noise_offsets(noise)
    n_tris = len(scene["faces"])

    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    # the timed steps end with every image in HOST memory (the reference's stopwatch bracket).  N = 1: the
    # library's own host delivery (rr_simulate_batch_host_async); N > 1: every rank delivers the frames it assembled
    host_mode = world == 1 and not args.force_slots
    shard = AzimuthShard(ctx, cfg.n_cells, params.N_ANGLES, rank, world, dev,
                         force_collective=args.force_slots, strong=args.strong,
                         frames_per_rank=args.frames_per_rank, n_slots=args.slots, host_out=not host_mode)
    fpb = shard.frames_per_step                 # frames of one batch (all ranks)
    bps = max(1, args.batches_per_step)
    fps = fpb * bps                             # frames per step (all ranks)

    def step(k, done=None):
        for b in range(bps):
            shard.step([poses[((k * bps + b) * fpb + f) % len(poses)] for f in range(fpb)], None,
                       done_event=done if b == bps - 1 else None)

    if host_mode:
        F = args.frames_per_rank
        h_streams = [tc.Stream(device=dev) for _ in range(args.slots)]
        # a ring of host buffers twice as deep as the batches in flight: the consumer side (rr_wait_host before a
        # buffer is handed out again) then never stalls the producer
        hosts = [native.HostImages((F, cfg.n_cells, params.N_ANGLES)) for _ in range(2 * args.slots)]
        h_state = {"n": 0}

        def step_main(k, done=None):
            for b in range(bps):
                n = h_state["n"]
                h_state["n"] += 1
                h = hosts[n % len(hosts)]
                ctx.wait_host(h.ptr)                  # the images this buffer received 2 x slots batches ago are complete
                st_ = h_streams[n % args.slots]
                ctx.simulate_batch_host_async([poses[((k * bps + b) * F + f) % len(poses)] for f in range(F)], h.ptr, st_.cuda_stream)
                if done is not None and b == bps - 1:
                    done.record(st_)

        def finish_main():
            ctx.wait_host(None)                       # through the last D2H copy
    else:
        step_main = step

        def finish_main():
            shard.flush_host()

    def prewarm(fn, seconds=PREWARM_S, collective=False):
        """untimed: at least `seconds` of steps so the timed region starts at sustained clocks.  With a collective
        inside the step every rank must run the SAME number of steps: the count is derived from the slowest rank's
        time for the first four."""
        t0, k = time.perf_counter(), 0
        if collective and world > 1:
            for k in range(4):
                fn(k)
            tc.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            more = int(math.ceil(seconds / max(float(t.item()) / 4.0, 1e-6))) - 4
            for k in range(4, 4 + max(more, 0)):
                fn(k)
            tc.synchronize()
            return 4 + max(more, 0)
        while True:
            fn(k); k += 1
            if k % 4 == 0:
                tc.synchronize()
                if time.perf_counter() - t0 >= seconds:
                    return k

    for k in range(args.warmup):
        step(k)
    tc.synchronize()
    # ---- instrumentation, all of it outside (and before) the timed region ---------------------------------
    # one batch: wave-pass count of this workload
    shard.step([poses[f % len(poses)] for f in range(fpb)], None)
    tc.synchronize()
    st = ctx.stats()
    wave_passes_batch_rank = st["wave_passes"]
    assert st["overflow"] == 0, st
    # one batch with the counting build of k_trace: measured node / triangle fetches per wave-pass
    ctx.set_stats_mode(True)
    shard.step([poses[f % len(poses)] for f in range(fpb)], None)
    tc.synchronize()
    st2 = ctx.stats()
    shape = ctx.traversal_shape()
    ctx.set_stats_mode(False)
    # every kernel alone on the GPU (one batch at a time, nothing else in flight): which kernel takes the most
    # time, and its own speed -- as opposed to its duration while several batches share the chip below
    prewarm(step, collective=True)
    ctx.reserve_timing_events(4096)
    ctx.set_timing_mode(1)
    for name in KERNEL_LABEL:
        ctx.kernel_time(name, reset=True)
    for k in range(6):
        shard.step([poses[(k * fpb + f) % len(poses)] for f in range(fpb)], None)
        tc.synchronize()
    iso = {name: ctx.kernel_time(name, reset=True) for name in KERNEL_LABEL}      # (total ms, launches)
    ctx.set_timing_mode(0)
    dominant = max(iso, key=lambda n: iso[n][0])
    # ---- the timed region: exactly `steps` steps between two synchronisation points ------------------------
    # (timing mode 2: pooled hipExtLaunchKernel events around the k_trace launches only, on the launch stream)
    done = [tc.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ctx.reserve_timing_events(4 * n_pass * bps * args.steps + 64)
    prewarm(step_main, collective=True)
    finish_main(); tc.synchronize()
    # (RR_BENCH_LIVE_TIMING=0: a diagnostic -- the timed region without the begin / end events around its k_trace launches;
    # roofline.frac then falls back to the isolated launch time and says so)
    # The events exist to time the DOMINANT kernel live (roofline.achieved); they are taken only when that kernel is a k_trace
    # launch.  Where another kernel dominates (config 2: k_column) they would buy nothing and they are not free there: the
    # chains of a one-pass workload are then issued kernel by kernel instead of being replayed from launch graphs, 34.7k ->
    # 27.8k images/s on config 2 (round 6, same box) -- that was the "regression" of the round-5 line (30.9k against the 39.0k
    # round 4 had measured in an untimed extra region)
    live_timing = os.environ.get("RR_BENCH_LIVE_TIMING", "1") != "0" and dominant in ("trace", "trace0")
    graphs_before = ctx.graph_stats() if hasattr(ctx, "graph_stats") else (0, 0)
    ctx.set_timing_mode(2 if live_timing else 0)
    ctx.kernel_time("trace", reset=True); ctx.kernel_time("trace0", reset=True); ctx.kernel_time("trace_repair", reset=True)
    if world > 1:
        dist.barrier()
    tc.synchronize()
    done[0].record(tc.current_stream())
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_main(k, done[k + 1])
    finish_main()                                     # ... through the last D2H copy of the last image
    tc.synchronize()
    t_own = time.perf_counter() - t0                  # this rank's own time (the line's `value` uses the slowest rank's)
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    live = {n: ctx.kernel_time(n, reset=True) for n in ("trace", "trace0", "trace_repair")}
    ctx.set_timing_mode(0)
    graphs_after = ctx.graph_stats() if hasattr(ctx, "graph_stats") else (0, 0)
    # ---- contention: every kernel's duration while `slots` batches share the chip (outside the timed region: stream events
    # around every launch keep the launch graphs off and cost host time) against its duration alone (`iso` above)
    live_all = None
    if rank == 0 and world == 1 and not args.force_slots:
        ctx.reserve_timing_events(2 * 8 * (n_pass * 3 + 2) * bps + 64)
        for k in range(2):
            step_main(k)
        ctx.set_timing_mode(1)
        for name in KERNEL_LABEL:
            ctx.kernel_time(name, reset=True)
        for k in range(6):
            step_main(k)
        finish_main(); tc.synchronize()
        live_all = {name: ctx.kernel_time(name, reset=True) for name in KERNEL_LABEL}
        ctx.set_timing_mode(0)
    # per-step times: steps overlap and may finish out of order (4 streams), so a single step has no duration of
    # its own; what is defined is the cadence -- completion times in completion order, differenced over windows of
    # `slots` completions (one window = as many steps as can be in flight)
    tdone = sorted(done[0].elapsed_time(done[k]) for k in range(1, args.steps + 1))
    win = max(1, min(args.slots, len(tdone) - 1))
    periods_ms = [(tdone[i + win] - tdone[i]) / win for i in range(len(tdone) - win)]

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        wp = torch.tensor([wave_passes_batch_rank], dtype=torch.int64, device=dev)
        dist.all_reduce(wp, op=dist.ReduceOp.SUM)
        wave_passes_batch = int(wp.item())
    else:
        wave_passes_batch = wave_passes_batch_rank

    # ---- N = 1 extras: the image delivered to host memory; one pose per launch set ---------------------------
    hbm_res = single = sync1 = proxy = wproxy = None
    if world == 1 and not args.no_extras and not args.force_slots:
        npx = cfg.n_cells * params.N_ANGLES
        for h in hosts:
            h.close()
        # the same steps with the images left in HBM (`value` of rounds 1-4)
        prewarm(step)
        tc.synchronize()
        th0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        tc.synchronize()
        th1 = time.perf_counter()
        hbm_res = {"value": round(args.steps * fps / (th1 - th0), 2), "unit": "images/s",
                   "ms_per_step": round(1e3 * (th1 - th0) / args.steps, 4),
                   "what": "same steps, every mono8 image assembled and LEFT IN HBM (no D2H copy in the timed region): the "
                           "`value` of rounds 1-4"}
        sp_slots = max(4, args.slots)               # single-frame launch sets want four in flight (2 / 3 / 4: 2,980 / 3,710 / 4,180 images/s)
        one = AzimuthShard(ctx, cfg.n_cells, params.N_ANGLES, 0, 1, dev, frames_per_rank=1, n_slots=sp_slots)
        n1 = max(args.steps, 16)
        prewarm(lambda k: one.step([poses[k % len(poses)]], None))
        tc.synchronize()
        ts0 = time.perf_counter()
        for k in range(n1):
            one.step([poses[k % len(poses)]], None)
        tc.synchronize()
        ts1 = time.perf_counter()
        single = {"value": round(n1 / (ts1 - ts0), 2), "unit": "images/s", "frames_per_launch_set": 1,
                  "ms_per_image": round(1e3 * (ts1 - ts0) / n1, 4), "images": n1,
                  "what": "one pose per set of launches, %d in flight on %d streams, image left in HBM" % (sp_slots, sp_slots)}
        one.close()
        # the reference's own call shape (radar_simulator.cpp:197-212): ONE synchronous simulate() per frame, image in host memory
        img = np.zeros((cfg.n_cells, params.N_ANGLES), np.uint8)
        for k in range(8):
            ctx.simulate_into(poses[k % len(poses)], img)
        lat = []
        for k in range(60):
            tq = time.perf_counter()
            ctx.simulate_into(poses[k % len(poses)], img)
            lat.append(1e3 * (time.perf_counter() - tq))
        sync1 = {"ms_per_frame": round(pct(lat, 0.5), 4), "p10": round(pct(lat, 0.1), 4), "p90": round(pct(lat, 0.9), 4),
                 "images_per_s": round(1e3 / pct(lat, 0.5), 1), "frames": len(lat),
                 "what": "rr_simulate: one pose, nothing else in flight, wall time of the call including the D2H copy of the mono8 image "
                         "into pageable host memory (the latency a ROS node calling Radar::simulate() per frame sees)"}
        # strong-scaling proxy on ONE GPU: a block of 400/N azimuth columns alone on the chip against the whole frame -- what
        # sharding ONE frame N ways can win at best (the collective and the transpose still to be added)
        blk = torch.zeros((params.N_ANGLES, cfg.n_cells), dtype=torch.uint8, device=dev)
        s1 = tc.Stream(device=dev)

        def block_ms(b, e, n=40):
            ts = []
            for k in range(n + 6):
                tc.synchronize(); tq = time.perf_counter()
                ctx.simulate_columns_device(poses[k % len(poses)], b, e, blk.data_ptr(), None, s1.cuda_stream); s1.synchronize()
                if k >= 6:
                    ts.append(1e3 * (time.perf_counter() - tq))
            return pct(ts, 0.5)
        t_full = block_ms(0, params.N_ANGLES)
        proxy = {"ms_400_columns": round(t_full, 4)}
        for n_sh in (2, 4, 8):
            b, e = native.partition(params.N_ANGLES, n_sh, n_sh // 2)
            t_b = block_ms(b, e)
            proxy["ms_%d_columns" % (e - b)] = round(t_b, 4)
            proxy["speedup_bound_%d_gpus" % n_sh] = round(t_full / t_b, 3)
            proxy["strong_ceiling_%d" % n_sh] = round(t_full / (n_sh * t_b), 4)
        proxy["what"] = ("one frame's kernel chain for a block of 400/N columns alone on the GPU vs all 400: an upper bound of the "
                         "strong-scaling speed-up of ONE frame on N GPUs (efficiency = strong_ceiling_N); the default N > 1 mode of "
                         "this bench is WEAK scaling (N x frames per batch), which keeps every GPU's launches as large as at N = 1")
        # weak-scaling proxy on ONE GPU: the work of rank 0 of N -- its 400/N-column block of N x F frames in one set of
        # launches, then the transpose of the F frames it ends up with -- against the N = 1 step (all 400 columns of F
        # frames).  Same number of segments per launch by construction; what can differ is the cost of many frames'
        # narrow blocks (pass-0 tiles, per-frame tables, the transpose from [frame][50][cells] pieces).  The collective
        # itself (F x 1.37 MB in and out per rank per batch) is NOT in it
        wproxy = None
        Fw = args.frames_per_rank
        if Fw * 8 <= 64:
            wstreams = [tc.Stream(device=dev) for _ in range(args.slots)]
            wproxy = {}
            base_ms = None
            for n_sh in (1, 2, 4, 8):
                b, e = native.partition(params.N_ANGLES, n_sh, 0)
                nl = e - b
                nfr = n_sh * Fw
                blocks = [torch.zeros((nfr, nl, cfg.n_cells), dtype=torch.uint8, device=dev) for _ in range(args.slots)]
                imgs_w = [torch.zeros((Fw, cfg.n_cells, params.N_ANGLES), dtype=torch.uint8, device=dev) for _ in range(args.slots)]
                state_w = {"n": 0}

                def step_w(k):
                    for bb in range(bps):
                        n = state_w["n"]; state_w["n"] += 1
                        st_ = wstreams[n % args.slots]
                        pl = [poses[((k * bps + bb) * nfr + f) % len(poses)] for f in range(nfr)]
                        ctx.simulate_batch_columns_device(pl, b, e, blocks[n % args.slots].data_ptr(), st_.cuda_stream)
                        # the rank's own F frames, as the all_to_all would deliver them: [source rank][F][nl][cells]
                        ctx.assemble_frames_device(blocks[n % args.slots].data_ptr(), nl, Fw * nl * cfg.n_cells, Fw, nl * cfg.n_cells,
                                                   imgs_w[n % args.slots].data_ptr(), st_.cuda_stream)
                prewarm(step_w)
                tc.synchronize()
                tw0 = time.perf_counter()
                n_st = max(args.steps // 2, 10)
                for k in range(n_st):
                    step_w(k)
                tc.synchronize()
                ms = 1e3 * (time.perf_counter() - tw0) / n_st
                if n_sh == 1:
                    base_ms = ms
                wproxy["ms_per_step_rank0_of_%d" % n_sh] = round(ms, 4)
                wproxy["weak_ceiling_%d" % n_sh] = round(base_ms / ms, 4)
                del blocks, imgs_w
            wproxy["what"] = ("per-GPU step time of the weak-scaling shape on ONE GPU: rank 0 of N renders its 400/N-column block of N x %d "
                              "frames per batch and transposes the %d frames it would own, against N = 1 (400 columns of %d frames); "
                              "weak_ceiling_N = what N GPUs could reach of N x the single-GPU rate before the collective (%.1f MB in and "
                              "out per rank per batch over xGMI) is paid; images left in HBM" % (Fw, Fw, Fw, Fw * cfg.n_cells * params.N_ANGLES / 1e6))

    # ---- N > 1: what a reader needs to believe RCCL saw N ranks ------------------------------------------------
    rccl = None
    if world > 1 or args.force_slots:
        props = tc.get_device_properties(local_rank)
        bus = None
        if all(hasattr(props, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            bus = "%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        desc = {"device_index": int(local_rank), "name": props.name, "pci_bus_id": bus,
                "uuid": str(getattr(props, "uuid", "")) or None, "pid": os.getpid()}
        n_loc = shard.n_loc
        if shard.strong:
            coll, nbytes = "all_gather_into_tensor (one per frame)", n_loc * cfg.n_cells
        else:
            coll, nbytes = "all_to_all_single (one per batch: frames d*F.. -> rank d)", fpb * n_loc * cfg.n_cells
        own = args.steps * bps * shard.fpr / t_own        # frames THIS rank assembled and delivered per second of its own time
        rccl = rccl_block(rank, world, desc, coll, nbytes, own)

    out = None
    if rank == 0:
        img_per_s = args.steps * fps / elapsed
        b_wp = algorithmic_bytes_per_wave_pass(n_tris)
        wp2 = max(int(st2["wave_passes"]), 1)
        measured_b_wp = (st2["nodes_visited"] * 128 + st2["tris_tested"] * 48) / wp2 + 132
        # --- roofline of the dominant kernel ------------------------------------------------------------------
        counters = {}
        cf = os.path.join(ROOT, "profiles", "roofline_counters.json")
        if os.path.exists(cf):
            counters = json.load(open(cf)).get(args.workload, {})
        # the counters describe the kernels they were collected from: a library built from other kernel sources makes them
        # stale -- then no fraction is printed rather than a wrong one (profiles/make_counters.py records the hash)
        src_now = kernel_source_hash()
        stale = bool(counters) and counters.get("kernel_source_sha16") not in (None, src_now)
        if stale:
            print("bench.py: profiles/roofline_counters.json was collected from other kernel sources (%s, now %s): roofline.frac "
                  "withheld -- re-run profiles/collect.sh + make_counters.py" % (counters.get("kernel_source_sha16"), src_now), file=sys.stderr)
            counters = {"source": counters.get("source", "") + " (STALE: other kernel sources)", "frames_per_launch": counters.get("frames_per_launch")}
        kc = counters.get("kernels", {}).get(dominant, {})
        n_launch_frames = fpb // world                      # frames one launch of this rank covers
        # the counters were collected with `frames_per_launch` frames per launch (the default batch shape); another
        # shape (--strong, --frames-per-rank) does proportionally more or less work per launch
        scale = n_launch_frames / float(counters.get("frames_per_launch", n_launch_frames) or n_launch_frames)
        insts = kc.get("SQ_INSTS_VALU")                     # wave-instructions per launch (PMC, same command)
        traffic = kc.get("hbm_bytes")
        if insts is not None:
            insts = int(round(insts * scale))
        if traffic is not None:
            traffic = int(round(traffic * scale))
        iso_ms, iso_n = iso[dominant]
        iso_s = 1e-3 * iso_ms / max(iso_n, 1)
        if dominant in live and live[dominant][1] > 0:
            live_s = 1e-3 * live[dominant][0] / live[dominant][1]
            live_n = int(live[dominant][1])
        else:                                               # a kernel other than k_trace dominates (or RR_BENCH_LIVE_TIMING=0): isolated figure only
            live_s, live_n = iso_s, int(iso_n)
        wp_launch = wave_passes_batch_rank / max(n_pass, 1)  # mean wave-passes of one k_trace launch
        achieved = (insts / live_s) if (insts and live_s > 0) else None
        chip = None
        ks = counters.get("kernels", {})
        if ks and all(k in ks for k in ("trace0", "column", "shade")):
            per_batch = {"trace0": 1, "trace": max(n_pass - 1, 0), "shade": n_pass, "scan": max(n_pass - 1, 0), "column": 1, "assemble": 1}
            tot = sum(ks[k]["SQ_INSTS_VALU"] * n * scale for k, n in per_batch.items() if k in ks and n)
            t_batch = elapsed / (args.steps * bps)
            chip = {"wave_instr_per_batch": int(tot), "achieved": round(tot / t_batch / 1e9, 2),
                    "frac": round(tot / t_batch / VALU_PEAK_WAVE_INSTR_S, 4),
                    "what": "all kernels of a batch (SQ_INSTS_VALU per launch x launches per batch) / measured time per batch; "
                            "the dominant kernel's own `frac` is lower because %d batches share the chip" % args.slots}
        # `frac` is an ISSUE rate, not an efficiency: a wave holds 16 rays and runs until its slowest ray is done, and an
        # iteration issues the node path (46 instructions) and / or the leaf path (75) for all 16 quads whichever of them
        # need it.  Useful = the steps live rays actually took, weighted the same way (statistics build, all k_trace
        # launches of one batch)
        useful = None
        if shape["iterations"]:
            C_NODE, C_LEAF = 46.0, 75.0
            issued = 16.0 * (shape["node_path_issues"] * C_NODE + shape["leaf_path_issues"] * C_LEAF)
            used = shape["node_steps"] * C_NODE + shape["leaf_steps"] * C_LEAF
            w = float(max(shape["waves"], 1))
            useful = {"value": round(used / issued, 4),
                      "per_wave": {"iterations": round(shape["iterations"] / w, 2), "node_path_issues": round(shape["node_path_issues"] / w, 2),
                                   "leaf_path_issues": round(shape["leaf_path_issues"] / w, 2),
                                   "live_quad_steps": round(shape["live_quad_steps"] / w, 1), "issued_quad_slots": round(16.0 * shape["iterations"] / w, 1)},
                      "what": "(node steps x 46 + leaf steps x 75 instructions of live rays) / (16 quads x (iterations issuing the node "
                              "path x 46 + the leaf path x 75)): the share of k_trace's issued lane-work that advanced a ray; "
                              "useful rate = frac x this"}
        roof = {"bound": "valu_issue", "kernel": KERNEL_LABEL[dominant],
                "achieved": None if achieved is None else round(achieved / 1e9, 2),
                "peak": round(VALU_PEAK_WAVE_INSTR_S / 1e9, 1), "unit": "G wave-instr/s",
                "frac": None if achieved is None else round(achieved / VALU_PEAK_WAVE_INSTR_S, 4),
                "traffic": traffic,
                "avg_launch_us": round(live_s * 1e6, 2), "launches": live_n, "batches_in_flight": int(args.slots),
                # the k_trace_repair launch behind every tightened later-pass row has its own events: NOT inside avg_launch_us
                "trace_repair_avg_us": (round(1e3 * live["trace_repair"][0] / live["trace_repair"][1], 2) if live.get("trace_repair", (0, 0))[1] else None),
                "wave_instr_per_launch": insts,
                "basis": "SQ_INSTS_VALU per launch (rocprofv3 --pmc of this command: %s) / average launch duration of the "
                         "timed region (hipExtLaunchKernel begin/end events on the launch stream); unweighted: k_trace "
                         "issues no f64 and one transcendental (v_rcp_f32) per leaf step; peak = 1024 SIMD-32 x 2.4 GHz / 2 "
                         "cycles per wave64 instruction" % counters.get("source", "profiles/roofline_counters.json missing"),
                "isolated": {"avg_launch_us": round(iso_s * 1e6, 2),
                             "achieved": None if not insts or iso_s <= 0 else round(insts / iso_s / 1e9, 2),
                             "frac": None if not insts or iso_s <= 0 else round(insts / iso_s / VALU_PEAK_WAVE_INSTR_S, 4),
                             "what": "the same launches with ONE batch on the GPU at a time"},
                # the chip as a whole in the timed region: every kernel's instructions (PMC per launch x launches per
                # batch) over the measured time per batch
                "chip": chip,
                "kernel_time_share_isolated": {KERNEL_LABEL[n]: round(iso[n][0] / max(sum(v[0] for v in iso.values()), 1e-9), 4) for n in iso},
                # secondary views (never `frac`): what crosses the HBM interface, and the SURVEY §8d algorithmic byte rate
                "hbm_measured": (None if not traffic or live_s <= 0 else
                                 {"GBps": round(traffic / live_s / 1e9, 2), "of_peak": round(traffic / live_s / 1e9 / HBM_PEAK_GBS, 5),
                                  "peak_GBps": HBM_PEAK_GBS,
                                  "source": "2 x FETCH_SIZE + WRITE_SIZE per launch, separate --pmc passes (profiles/roofline_counters.json)"}),
                "algorithmic_bytes_per_wave_pass": b_wp,
                "measured_bytes_per_wave_pass": round(measured_b_wp, 1),
                "algorithmic_GBps": (round(wp_launch * b_wp / live_s / 1e9, 2) if dominant in ("trace", "trace0") and live_s > 0 else None),
                "frames_per_launch": n_launch_frames,
                "useful_issue_frac": useful}
        contention = None
        if live_all:
            per_batch = {"trace0": 1, "trace": max(n_pass - 1, 0), "shade": n_pass, "scan": max(n_pass - 1, 0), "column": 1, "assemble": 1}
            avg = lambda d, n: (d[n][0] / d[n][1]) if d[n][1] else 0.0     # noqa: E731  (ms per launch)
            ratio = lambda n: (round(avg(live_all, n) / avg(iso, n), 2) if avg(iso, n) > 0 and avg(live_all, n) > 0 else None)   # noqa: E731
            contention = {"batch_ms_isolated_sum": round(sum(avg(iso, n) * k for n, k in per_batch.items()), 4),
                          "batch_ms_live": round(1e3 * elapsed / (args.steps * bps), 4),
                          "trace_live_over_alone": ratio("trace"), "trace0_live_over_alone": ratio("trace0"),
                          "shade_live_over_alone": ratio("shade"), "scan_live_over_alone": ratio("scan"),
                          "column_live_over_alone": ratio("column"), "assemble_live_over_alone": ratio("assemble"),
                          "live_us": {KERNEL_LABEL[n]: round(1e3 * avg(live_all, n), 1) for n in live_all},
                          "alone_us": {KERNEL_LABEL[n]: round(1e3 * avg(iso, n), 1) for n in iso},
                          "what": "batch_ms_isolated_sum = the kernels of one batch one after the other, each at its duration with ONE batch on "
                                  "the chip; batch_ms_live = the timed region's time per batch with %d batches in flight: what running them "
                                  "concurrently buys is the ratio of the two.  *_live_over_alone = a launch's duration (stream events "
                                  "around it, a short region after the timed one) with %d batches sharing the chip / alone" % (args.slots, args.slots)}
        out = {
            "metric": "polar images/sec (400 az x 3424 bins)",
            "value": round(img_per_s, 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "strong" if (world > 1 and args.strong) else "weak",
            "vs_baseline": None,
            "dtype": "f32 rays / f64 energy+time",
            "data": "synthetic",
            "config": {"workload": args.workload, "triangles": int(n_tris), "azimuths": params.N_ANGLES,
                       "range_bins": int(cfg.n_cells), "rays_per_beam": n_rays, "passes": n_pass,
                       "ambient_noise": int(cfg.ambient_noise),
                       "frames_per_step": fps, "frames_per_batch": fpb, "batches_per_step": bps,
                       "images": "delivered to page-locked host memory; timed through the last D2H copy (SURVEY §8d bracket = the reference's stopwatch, RadarCPU.cpp:147-550)",
                       "d2h_GBps": round(img_per_s * cfg.n_cells * params.N_ANGLES / 1e9, 3),
                       "hip_runtime": runtime,
                       "host_delivery": (ctx.host_delivery_route() if hasattr(ctx, "host_delivery_route") else None),
                       "sharding": ("single GPU" if world == 1 else
                                    "azimuth columns x%d, 1 frame/batch + 1 all-gather" % world if args.strong else
                                    "azimuth columns x%d, %d frames/batch, 1 all_to_all/batch (frame f -> rank f)" % (world, fpb))},
            "rays_per_s": round(wave_passes_batch / fpb * img_per_s, 1),
            "wave_passes_per_frame": int(wave_passes_batch // fpb),
            "images_per_s_per_step": {"median": round(fps / (1e-3 * pct(periods_ms, 0.5)), 2) if periods_ms else None,
                                      "p10": round(fps / (1e-3 * pct(periods_ms, 0.9)), 2) if periods_ms else None,
                                      "p90": round(fps / (1e-3 * pct(periods_ms, 0.1)), 2) if periods_ms else None,
                                      "basis": "cadence of step completions (hip events), differenced over windows of %d completions" % win},
            "prewarm_s": PREWARM_S,
            "rccl": rccl,
            "n1_reference": n1_reference(args.workload) if world > 1 else None,
            "hbm_resident": hbm_res,
            "single_pose": single,
            "single_frame_sync": sync1,
            "strong_scaling_proxy": proxy,
            "weak_scaling_proxy": wproxy,
            "roofline": roof,
            "contention": contention,
            "launch_graphs": {"replays_in_timed_region": int(graphs_after[1] - graphs_before[1]),
                              "captures_in_timed_region": int(graphs_after[0] - graphs_before[0]),
                              "live_timing_events": bool(live_timing),
                              "what": "chains replayed from hipGraphs inside the timed region.  0 where the dominant kernel is a k_trace "
                                      "launch: its begin / end events (roofline.avg_launch_us is measured live) keep the chains kernel by "
                                      "kernel -- GPU time is the same either way, a replay saves host time; `hbm_resident` and "
                                      "`single_pose` replay graphs.  (In the fallback delivery route, RR_HOST_SDMA=0, the multi-pass chains "
                                      "also carry the previous batch's host copy and are never replayed.)"},
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene, cfg, mats, beams, noise, poses, args.cpu_seconds,
                                           brdf_model=1 if scene_id == 5 else 0)

    shard.close()
    ctx.close()
    if world > 1 or args.force_slots:
        dist.destroy_process_group()
    if rank == 0:
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()


def cpu_baseline(scene, cfg, mats, beams, noise, poses, budget_s, brdf_model=0):
    """The oracle (kind "port") on the host cores of this box, bounded sample."""
    from oracle import oracle as O
    O.build()
    ncpu = os.cpu_count() or 1
    # what the process may actually USE: the scheduler affinity and the cgroup's CPU quota (the pool's GPU boxes show 256
    # hardware threads and grant 16 CPUs' worth of time: cpu.max "1600000 100000" -- the oracle scales linearly to 16
    # threads there and loses beyond, tools/cpu_scaling.py)
    usable = ncpu
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
            usable = max(1, min(usable, int(math.ceil(quota))))
    except (OSError, ValueError):
        pass
    sc = O.Scene(scene["verts"], scene["faces"], scene["face_object_id"])
    m = [x.astuple() for x in mats]

    def frame(k, nt):
        _, _, st = O.simulate(sc, m, scene["object_materials"], cfg, beams, poses[k % len(poses)],
                              noise_rnd=noise[:400], want_f32=False, n_threads=nt, brdf_model=brdf_model)
        return st["seconds"]

    # the reference parallelises azimuths with OpenMP (RadarCPU.cpp:155); pick the thread count that is FASTEST
    # among those the process can really run, so the baseline is not handicapped
    cands = sorted({c for c in (4, 8, 16, 32, 64, 128) if c <= usable} | {usable})
    best_nt, best_t = cands[0], float("inf")
    for nt in cands:
        frame(0, nt)
        t = min(frame(1, nt), frame(2, nt))
        if t < best_t:
            best_nt, best_t = nt, t
    cores = best_nt
    secs, frames = [], 0
    t_wall = time.perf_counter()
    k = 0
    while True:
        secs.append(frame(k, cores))
        k += 1
        frames += 1
        if (time.perf_counter() - t_wall > budget_s and frames >= 3) or frames >= 64:
            break
    med = float(np.median(secs))
    return {"value": round(1.0 / med, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "host_threads": ncpu, "usable_cpus": usable, "cgroup_cpu_quota": quota,
            "p10": round(1.0 / pct(secs, 0.9), 3), "p90": round(1.0 / pct(secs, 0.1), 3),
            "embree": "unavailable",        # no Embree (nor rmagine) on this box: the in-repo SAH BVH2 stands in
            "sample": "%d full frames of the same workload (16-pose trajectory), median of the "
                      "RadarCPU.cpp:147-550 stopwatch bracket, OpenMP over azimuths, in-repo SAH BVH2 "
                      "(Embree absent); threads = fastest of %s; the host shows %d hardware threads, this process may use %d "
                      "(affinity / cgroup quota)" % (frames, cands, ncpu, usable)}


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:       # noqa: BLE001  one JSON line with "error" instead of a traceback only, then non-zero
        report_failure(exc)
        raise
