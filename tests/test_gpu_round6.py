"""GPU tests of round 6 (include/radarays_mi355.h, ABI 6):
  * launch graphs replayed back to back with NO host synchronisation between the calls (advisor, round 5: a replay re-sets
    the pose node of an exec whose previous launch may still be queued) -- every batch must show ITS poses
  * host delivery by the library's own copy kernel (rr_copy_to_host_async, k_copy_host): the bytes, whatever its shape
  * the per-pass history reaches the host through a kernel's stores (no hipMemcpyAsync left in a chain)
  * k_trace with the root step peeled (RR_ROOT_PRELOAD builds) is covered by the ordinary parity suite: same code path
"""
import os

import numpy as np
import pytest

from common import golden_beams, materials_for
from radarays_ros_amd import params, scenes

pytestmark = pytest.mark.gpu


def _ctx(native_lib, s, cfg, mats, beams, noise=None):
    c = native_lib.Context(0)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    c.set_materials(mats, s["object_materials"], 0)
    c.set_config(cfg)
    c.set_beam_samples(beams)
    if noise is not None:
        c.set_noise_offsets(noise)
    return c


@pytest.mark.parametrize("lanes", ["1", "2", "4"])
def test_graph_replays_back_to_back_show_their_own_poses(native_lib, monkeypatch, lanes):
    """10 batches of DIFFERENT poses into 10 distinct buffers, issued back to back on one stream without a host wait, twice
    over (so that every shape is captured and then replayed while earlier replays are still queued): each buffer equals the
    kernel-by-kernel render (RR_GRAPHS=0) of its own poses.  RR_LANES=1: every replay meets the exec (pair) of the one lane;
    the batches are heavy enough (8 frames x 400 azimuths x 96 rays x 3 passes) that several are in flight at any time."""
    import torch
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    cfg = params.kaist_preset(n_reflections=3, n_samples=96, ambient_noise=2)
    mats = materials_for(s)
    beams = golden_beams(96)
    noise = (np.random.RandomState(11).uniform(0, 1, (8, 400)) * 1000.0).astype(np.float32)
    P = scenes.trajectory(21, s["name"])                 # 21 poses, windows of 8 at a stride of 2: all batches differ
    batches = [[P[(2 * k + j) % 21] for j in range(8)] for k in range(10)]

    monkeypatch.setenv("RR_LANES", lanes)
    monkeypatch.setenv("RR_GRAPHS", "0")
    c0 = _ctx(native_lib, s, cfg, mats, beams, noise)
    st = torch.cuda.current_stream().cuda_stream
    one = torch.zeros((8, cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
    want = []
    for b in batches:
        c0.simulate_batch_device(b, one.data_ptr(), st)
        c0.synchronize(st)
        want.append(one.cpu().numpy().copy())
    assert c0.graph_stats() == (0, 0)
    c0.close()
    assert not np.array_equal(want[0], want[1])

    monkeypatch.delenv("RR_GRAPHS")
    c = _ctx(native_lib, s, cfg, mats, beams, noise)
    outs = [torch.zeros((8, cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0") for _ in range(10)]
    for rnd in range(3):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for k, b in enumerate(batches):                  # no synchronisation in here
            c.simulate_batch_device(b, outs[k].data_ptr(), st)
        c.synchronize(st)
        for k in range(10):
            assert np.array_equal(outs[k].cpu().numpy(), want[k]), (rnd, k)
    cap, rep = c.graph_stats()
    assert cap >= 1 and rep >= 10, (cap, rep)
    c.close()


def test_copy_to_host_async_moves_the_bytes(native_lib, monkeypatch):
    """rr_copy_to_host_async: page-locked destination -> the copy kernel (any number of workgroups / stores in flight),
    pageable or misaligned -> hipMemcpyAsync; the same bytes either way."""
    import torch
    s = scenes.box12()
    cfg = params.kaist_preset(n_reflections=1, ambient_noise=0)
    rng = np.random.RandomState(3)
    n = 3 * 3424 * 400
    src = torch.from_numpy(rng.randint(0, 256, n + 64).astype(np.uint8)).to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for blocks, inflight in (("1", "1"), ("7", "0"), ("64", "4"), ("1024", "64")):
        monkeypatch.setenv("RR_FLUSH_BLOCKS", blocks)
        monkeypatch.setenv("RR_FLUSH_INFLIGHT", inflight)
        c = _ctx(native_lib, s, cfg, params.kaist_materials(), golden_beams(16))
        h = native_lib.HostImages((n + 64,))
        for off, nb in ((0, n), (16, 4096), (0, 16), (32, n - 32), (16, 1000), (3, 1000), (0, 0)):     # (16, 1000) / (3, 1000): not multiples of 16 / misaligned
            h.array[:] = 0
            c.copy_to_host_async(src.data_ptr() + off, h.ptr + off, nb, st)
            c.synchronize(st)
            assert np.array_equal(h.array[off:off + nb], src[off:off + nb].cpu().numpy()), (blocks, inflight, off, nb)
            assert not h.array[off + nb:].any() and not h.array[:off].any()
        pageable = np.zeros(n, np.uint8)
        c.copy_to_host_async(src.data_ptr(), pageable.ctypes.data, n, st)
        c.synchronize(st)
        assert np.array_equal(pageable, src[:n].cpu().numpy())
        h.close(); c.close()


@pytest.mark.parametrize("passes", [1, 3])
@pytest.mark.parametrize("flush", [("sdma",), ("1", "8", "1"), ("1", "64", "0"), ("0", "32", "4"), ("stream",)])
def test_host_delivery_routes(native_lib, monkeypatch, passes, flush):
    """rr_simulate_batch_host_async with one pass (no later-pass launch a copy could ride on) and with three, 18 batches over
    four streams, under every route the images can take: `sdma` -- the default: ROCr's SDMA path, a worker thread, each copy
    behind its batch's last kernel (csrc/rr_sdma.cpp) -- and, with RR_HOST_SDMA=0, the deferred copies: through k_copy_host
    in several shapes, through hipMemcpyAsync (RR_FLUSH_KERNEL=0), one-pass frames on a dedicated copy stream
    (RR_HOST_COPY_STREAM=1).  The images of rr_simulate, and each buffer complete when ITS wait returns."""
    import torch
    monkeypatch.setenv("RR_HOST_SDMA", "1" if flush[0] == "sdma" else "0")
    monkeypatch.setenv("RR_HOST_SDMA_VERBOSE", "1")
    if flush[0] == "stream":
        monkeypatch.setenv("RR_HOST_COPY_STREAM", "1")
    elif flush[0] != "sdma":
        monkeypatch.setenv("RR_FLUSH_KERNEL", flush[0]); monkeypatch.setenv("RR_FLUSH_BLOCKS", flush[1]); monkeypatch.setenv("RR_FLUSH_INFLIGHT", flush[2])
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    cfg = params.kaist_preset(n_reflections=passes, n_samples=60, ambient_noise=2)
    noise = (np.random.RandomState(5).uniform(0, 1, 400) * 1000.0).astype(np.float32)
    poses = scenes.trajectory(6, s["name"])
    c = _ctx(native_lib, s, cfg, materials_for(s), golden_beams(60), noise)
    ref = [c.simulate(p)[0] for p in poses]
    streams = [torch.cuda.Stream() for _ in range(4)]
    NB = 18
    bufs = [native_lib.HostImages((4, cfg.n_cells, 400)) for _ in range(NB)]
    for b in bufs:
        b.array[:] = 9
    for k in range(NB):
        c.simulate_batch_host_async([poses[(k + j) % 6] for j in range(4)], bufs[k].ptr, streams[k % 4].cuda_stream)
    for k in (NB - 1, 0, 7, 16):
        c.wait_host(bufs[k].ptr)
        for j in range(4):
            assert np.array_equal(bufs[k].array[j], ref[(k + j) % 6]), (k, j)
    c.wait_host(None)
    for k in range(NB):
        for j in range(4):
            assert np.array_equal(bufs[k].array[j], ref[(k + j) % 6]), (k, j)
    assert c.host_delivery_route() == ("sdma" if flush[0] == "sdma" else "stream copies")     # no silent fallback
    c.synchronize()
    for b in bufs:
        b.close()
    c.close()


def test_trace_row_history_reaches_the_host_without_a_copy_call(native_lib, monkeypatch):
    """The per-pass history (GridHint) is stored into its page-locked host copy by the chain's k_column: after a few batches
    the rows a lane launches are tightened (rr_get_trace_grid), exactly as when a 96-byte hipMemcpyAsync carried it."""
    import torch
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    cfg = params.kaist_preset(n_reflections=4, n_samples=64, ambient_noise=0)
    c = _ctx(native_lib, s, cfg, materials_for(s), golden_beams(64))
    poses = scenes.trajectory(8, s["name"])
    imgs = torch.zeros((8, cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(6):
        c.simulate_batch_device(poses, imgs.data_ptr(), st)
        c.synchronize(st)
    rows, hist, repaired = c.trace_grid()
    assert hist[1] > 0 and hist[2] > 0 and hist[3] > 0, hist
    full = [(64 << p) // 16 for p in range(4)]
    assert any(0 < rows[p] < full[p] for p in (1, 2, 3)), (rows, full)      # at least one pass runs tightened rows
    c.close()


def test_gpu_fresnel_split_against_the_oracle_on_the_reference_derived_cases(native_lib, oracle):
    """The kernels' fresnel_split (rr_debug_fresnel) on the 11,000 cases of tests/golden/pyref_cases.py -- the inputs whose
    reference-python outputs pin the oracle in tests/test_oracle_dense_pin.py (v1 != 0.3, v2 > v1, the angle-limit branch,
    both eps branches).  Reflection directions: bit for bit (pure un-fused f32 arithmetic).  The transmitted / totally-
    reflected decision: identical.  Everything else hangs on the incidence ANGLE, an f32 value both sides get from an arc
    cosine -- the host's acosf against the GPU's (float)acos((double)x) -- which agree in all but a few per cent of the inputs
    and then differ by one f32 ulp, i.e. ~1e-7 rad; where they agree the refraction direction is bit-equal and the energies
    agree to 1e-12, where they do not the results move by that ulp times the formulas' sensitivity (the law of the CPU test).
    NaN energies (the reference's acosf(> 1) at near-normal incidence) are NaN on both sides."""
    import math
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import pyref_cases
    c = native_lib.Context(0)
    report = {}
    for fam in pyref_cases.FAMILIES:
        th, v1, v2 = pyref_cases.cases(fam)
        D = pyref_cases.direction(th)
        Nn = np.tile(np.array([[-1.0, 0.0, 0.0]], np.float32), (len(th), 1))
        rd, re, td, te = c.debug_fresnel(Nn, D, 1.0, v1, v2.astype(np.float32))
        o = [oracle.fresnel((-1.0, 0.0, 0.0), D[i], 1.0, 0.5, float(v1[i]), float(v2[i])) for i in range(len(th))]
        ord_, ore, otd, ote = (np.array([x[k] for x in o]) for k in range(4))
        otd32 = otd.astype(np.float32)
        assert np.array_equal(rd.view(np.uint32), ord_.astype(np.float32).view(np.uint32)), fam
        tr = np.any(otd32 != 0, axis=1)
        flip = np.any(td != 0, axis=1) != tr
        for i in np.nonzero(flip)[0]:                          # a flipped decision: only on the limit angle itself (one ulp of the angle)
            assert v2[i] > v1[i] and abs(th[i] - math.asin(v1[i] / v2[i])) < 1e-6, (fam, i)
        same = np.all(td.view(np.uint32) == otd32.view(np.uint32), axis=1) & ~flip
        for i in np.nonzero(~same & ~flip)[0]:
            n12 = v2[i] / v1[i]
            ct = max(math.cos(math.asin(min(1.0, math.sin(th[i]) * n12))), 1e-4)
            assert np.abs(td[i] - otd32[i]).max() < 2e-7 + 2e-7 * n12 * n12 / ct, (fam, i, td[i], otd32[i])
        nan_o, nan_g = np.isnan(ore), np.isnan(re)
        ok = same & ~nan_o & ~nan_g
        tight = np.abs(re[ok] - ore[ok]) < 1e-12
        report[fam] = (len(th), int(flip.sum()), float(same.mean()), float(tight.mean()), int((nan_o != nan_g).sum()),
                       float(np.abs(re[ok] - ore[ok]).max()))
        assert same.mean() > 0.93 and tight.mean() > 0.93, report
        assert (nan_o != nan_g).sum() <= 0.02 * len(th), report          # (a NaN needs the dot product to round above 1: the same f32 arithmetic)
        loose = ok & (np.minimum(th, np.where(tr, np.arcsin(np.minimum(1.0, np.sin(th) * v2 / np.maximum(v1, 1e-9))), np.pi / 2)) > 5e-3)
        assert not loose.any() or np.abs(re[loose] - ore[loose]).max() < 2e-5, report      # one ulp of an angle of >= 5e-3 rad
        # a scaled energy scales both results (radar_algorithms.h:135-136)
        rd2, re2, td2, te2 = c.debug_fresnel(Nn, D, 0.37, v1, v2.astype(np.float32))
        assert np.allclose(re2[ok], 0.37 * re[ok], rtol=1e-14, atol=0) and np.array_equal(td2.view(np.uint32), td.view(np.uint32))
    print("fresnel GPU vs oracle (cases, flips, dir bit-equal, energy < 1e-12, NaN mismatches, max |dE|):", report)
    c.close()


def test_stack_free_traversal_renders_the_same_bytes(native_lib, oracle, monkeypatch):
    """RR_STACKLESS=1: k_trace walks the tree without a stack (traverse_stackless: parent links, children in key order, a node
    re-fetched each time the walk returns to it; no LDS) -- the traversal north_star names, measured in round 6 at 1.7x the
    launch time of the stack walk and therefore not the default.  The nearest hit is the minimum over (t, face id) whatever
    the walk, so frames are byte-identical: a 3-pass frame of a scene with penetrable buildings, counts included, on both
    builders' trees."""
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    cfg = params.kaist_preset(n_reflections=3, n_samples=64, ambient_noise=2)
    noise = (np.random.RandomState(2).uniform(0, 1, 400) * 1000.0).astype(np.float32)
    pose = scenes.trajectory(5, s["name"])[2]
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RR_STACKLESS", mode)
        for builder in ("host", "gpu"):
            c = native_lib.Context(0)
            c.set_mesh(s["verts"], s["faces"], s["face_object_id"], builder=builder)
            c.set_materials(materials_for(s), s["object_materials"], 0)
            c.set_config(cfg)
            c.set_beam_samples(golden_beams(64))
            c.set_noise_offsets(noise)
            out[(mode, builder)] = c.simulate(pose, want_f32=True)
            c.close()
    ref8, reff, refst = out[("0", "host")]
    assert refst["hits"] > 10000 and refst["overflow"] == 0
    for k, (g8, gf, st) in out.items():
        assert np.array_equal(g8, ref8) and np.array_equal(gf.view(np.uint32), reff.view(np.uint32)), k
        assert (st["wave_passes"], st["hits"], st["signals"]) == (refst["wave_passes"], refst["hits"], refst["signals"]), k


def test_launch_graph_capture_while_deliveries_are_in_flight(native_lib, monkeypatch):
    """Found by fuzz_batch in round 6: a launch chain is CAPTURED (second use of a shape on a lane) on a stream whose earlier
    batch still has an SDMA delivery pending -- the worker thread sits in hipEventSynchronize on an event of that very stream,
    which the runtime refuses while the stream captures (the capture was invalidated and the entry point failed).  run_frame
    now lets the deliveries in flight finish before it captures.  Host deliveries and device-path batches interleaved on two
    streams over four lanes, shapes met for the first and second time all along: every image right, no error, and the SDMA
    route still in use at the end (a refused wait would have switched it off)."""
    import torch
    monkeypatch.setenv("RR_HOST_SDMA_VERBOSE", "1")
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    cfg = params.kaist_preset(n_reflections=3, n_samples=96, ambient_noise=2)
    noise = (np.random.RandomState(5).uniform(0, 1, 400) * 1000.0).astype(np.float32)
    c = _ctx(native_lib, s, cfg, materials_for(s), golden_beams(96), noise)
    P = scenes.trajectory(7, s["name"])
    ref = np.stack([c.simulate(p)[0] for p in P])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    rs = np.random.RandomState(1)
    for K in (3, 5, 2):                                   # three batch sizes = three families of shapes to capture
        poses = P[:K]
        hosts = [native_lib.HostImages((K, cfg.n_cells, 400)) for _ in range(9)]
        imgs = torch.zeros((K, cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
        for b in range(9):
            ps = np.roll(np.asarray(poses), -b, axis=0)
            c.simulate_batch_host_async(ps, hosts[b].ptr, streams[b % 2].cuda_stream)
            if rs.randint(0, 2) == 0:
                c.simulate_batch_device(poses, imgs.data_ptr(), streams[b % 2].cuda_stream)
        for b in rs.permutation(9):
            c.wait_host(hosts[int(b)].ptr)
            assert np.array_equal(hosts[int(b)].array, np.roll(ref[:K], -int(b), axis=0)), (K, int(b))
        c.synchronize(streams[0].cuda_stream); c.synchronize(streams[1].cuda_stream)
        assert np.array_equal(imgs.cpu().numpy(), ref[:K])
        for h in hosts:
            h.close()
    assert c.host_delivery_route() == "sdma" and c.graph_stats()[0] >= 3
    c.close()


@pytest.mark.gpu
def test_gpu_brdf_against_the_oracle_on_the_reference_derived_cases(native_lib, oracle):
    """The kernels' back_reflection_shader (rr_debug_brdf = the function k_shade calls) on the 3,624 cases of
    tests/golden/pyref_brdf.npy -- inputs whose expected values came out of the reference's own
    scripts/radarays_snell_fresnel_brdf.py -- against (a) the oracle: both compute cosf / powf in f32, the GPU with its own libm,
    so the results are the same float or a few ulp of the lobe apart (one ulp of the cosine times the exponent); (b) the
    script's values themselves, by the law of tests/test_oracle_brdf_pin.py.  Plus linearity in energy and diffuse weight,
    bit for bit (one f32 multiplication each), and the brdf_model = 1 lobe against its CPU twin."""
    X = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pyref_brdf.npy"))
    a, cx, w, want = X[:, 0], X[:, 1], X[:, 2], X[:, 3]
    d = (np.float32(1.0) - a.astype(np.float32)).astype(np.float32)
    c = native_lib.Context(0)
    got = c.debug_brdf(w, 1.0, a, d, cx).astype(np.float64)
    orc = np.array([oracle.back_reflection_shader(np.float32(w[i]), 1.0, float(a[i]), float(d[i]), float(cx[i])) for i in range(len(X))])
    cw = np.cos(w)
    lobe = (1.0 - a) * cw ** cx
    # (a) GPU vs oracle: ulp-level differences of cosf / powf only
    dev = np.abs(got - orc)
    law_a = 1.5e-7 * np.maximum(orc, 1e-30) + lobe * (2.5e-7 + 1.3e-7 * cx)      # (an ulp of the cosine becomes C ulps of the lobe)
    same = float((got.astype(np.float32) == orc.astype(np.float32)).mean())
    print("brdf GPU vs oracle: bit-equal %.4f, max |d| %.3g" % (same, dev.max()))
    assert same >= 0.90, same
    assert (dev <= law_a + 1e-12).all(), (dev.max(), X[np.argmax(dev - law_a)])
    # (b) GPU vs the script
    tol = np.minimum(1.5e-7 + lobe * (1e-7 + 6e-8 * cx * (1.0 + 1.0 / np.maximum(cw, 1e-7))), 1.5e-7 + lobe)
    assert (np.abs(got - want) <= tol).all()
    # linear in the energy and in the diffuse weight: cos^C alone, then one multiplication each
    base = c.debug_brdf(w, 1.0, 0.0, 1.0, cx)
    for dd, e in ((0.25, 0.37), (1.7, 12.5), (0.0, 1.0)):
        g = c.debug_brdf(w, e, a, dd, cx)
        ref = ((a.astype(np.float32) + np.float32(dd) * base).astype(np.float32) * np.float32(e)).astype(np.float32)
        assert np.array_equal(g, ref), (dd, e)
    # the build's own lobe (brdf_model = 1) against its CPU twin: same operation order, the GPU's sincosf / sqrtf
    import ctypes as C_
    L = oracle.lib()
    L.orc_back_reflection_shader_model.restype = C_.c_float
    L.orc_back_reflection_shader_model.argtypes = [C_.c_float] * 5 + [C_.c_int32]
    g1 = c.debug_brdf(w, 1.0, a, d, cx, brdf_model=1).astype(np.float64)
    o1 = np.array([L.orc_back_reflection_shader_model(np.float32(w[i]), 1.0, float(a[i]), float(d[i]), float(cx[i]), 1) for i in range(len(X))])
    assert np.abs(g1 - o1).max() <= 2e-6, np.abs(g1 - o1).max()
    with pytest.raises(native_lib.RRError, match="brdf_model"):
        c.debug_brdf(w[:4], 1.0, 0.5, 0.5, 10.0, brdf_model=2)
