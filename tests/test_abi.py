"""The C-ABI library loads and exports every symbol include/radarays_mi355.h
declares; struct layouts of the ctypes binding match the header.  No compute."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "radarays_mi355.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_exports_every_declared_symbol(native_lib):
    native_lib.build()
    L = C.CDLL(native_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "missing symbol %s" % n
    assert sorted(native_lib.SYMBOLS) == names


def test_no_torch_types_in_the_abi():
    src = open(HEADER).read()
    assert "torch" not in src.lower() and "at::" not in src and "c10" not in src


def test_struct_layouts_match_header(native_lib, tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "radarays_mi355.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(rr_config), sizeof(rr_material),'
                    ' sizeof(rr_stats), offsetof(rr_config, resolution), offsetof(rr_config, wave_energy_threshold),'
                    ' offsetof(rr_config, range_max));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    cfg, mat, st, off_res, off_thr, off_rm = map(int, out)
    assert C.sizeof(native_lib.RRConfig) == cfg
    assert C.sizeof(native_lib.RRMaterial) == mat == 16
    assert C.sizeof(native_lib.RRStats) == st
    assert native_lib.RRConfig.resolution.offset == off_res
    assert native_lib.RRConfig.wave_energy_threshold.offset == off_thr
    assert native_lib.RRConfig.range_max.offset == off_rm


def test_default_config_matches_reference_cfg(native_lib):
    """rr_default_config == cfg/RadarModel.cfg defaults (params.RadarModelConfig())."""
    from radarays_ros_amd import params
    c = native_lib.RRConfig()
    native_lib.lib().rr_default_config(C.byref(c))
    d = params.RadarModelConfig()
    for k in ("n_cells", "n_reflections", "signal_denoising", "signal_denoising_triangular_width",
              "signal_denoising_gaussian_width", "signal_denoising_mb_width", "ambient_noise", "scroll_image",
              "resolution", "energy_max", "signal_max", "signal_denoising_triangular_mode",
              "signal_denoising_gaussian_mode", "signal_denoising_mb_mode", "ambient_noise_at_signal_0",
              "ambient_noise_at_signal_1", "ambient_noise_energy_max", "ambient_noise_energy_min",
              "ambient_noise_energy_loss", "multipath_threshold"):
        assert getattr(c, k) == getattr(d, k), k
    assert c.n_angles == 400 and abs(c.theta_inc + 2 * 3.141592653589793 / 400) < 1e-8
    assert abs(c.wave_energy_threshold - 0.001) < 1e-9 and c.range_max == 1000.0
    assert c.record_multi_reflection == 1 and c.record_multi_path == 0
    assert native_lib.lib().rr_abi_version() == 6


def test_missing_library_fails_loudly(native_lib, monkeypatch, tmp_path):
    monkeypatch.setattr(native_lib, "_LIB", None)
    monkeypatch.setattr(native_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native_lib.lib()


def test_product_never_touches_the_oracle():
    """The shipped package must not import/link/execute anything under oracle/."""
    pkg = os.path.join(ROOT, "radarays_ros_amd")
    bad = re.compile(r"import\s+oracle|from\s+oracle|oracle/|libradarays_oracle|\borc_[a-z]")
    n = 0
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                n += 1
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert not bad.search(txt), f
    assert n >= 10


def test_partition_matches_the_python_shard_arithmetic(native_lib):
    """rr_partition (the block a device of rr_multi renders) == dist.partition (the block a rank of the
    torch.distributed path renders): contiguous, exhaustive, sizes differ by at most one column.  Pure host code."""
    from radarays_ros_amd.dist import partition
    for n_angles in (1, 7, 50, 399, 400, 401, 1000):
        for world in (1, 2, 3, 4, 7, 8, 16, 64):
            blocks = [native_lib.partition(n_angles, world, r) for r in range(world)]
            assert blocks == [partition(n_angles, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n_angles
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_multi_create_refuses_bad_device_lists(native_lib):
    """No GPU here: rr_create_multi must fail with a message, never crash or fall back."""
    L = native_lib.lib()
    assert not L.rr_create_multi(None, 0)
    assert b"1..64" in L.rr_multi_last_error(None)
    d = (C.c_int * 2)(0, 0)
    assert not L.rr_create_multi(d, 2)
    assert b"twice" in L.rr_multi_last_error(None)


def test_multi_plan_covers_every_frame_exactly_once(native_lib):
    """rr_multi_plan = the send / receive plan of rr_multi_simulate_batch (csrc/rr_multi.hip), pure arithmetic.  Emulated
    here with numpy: every device's block buffer [frame][n_loc_r][n_cells] is filled with (device, frame, column, cell)
    tags, the plan is applied, and the root's [frame][n_angles][n_cells] buffer must hold azimuth a of frame f where the
    assemble kernel reads it -- for ragged blocks (send/recv pieces) and for equal blocks (all-gather layout
    [device][frame][n_loc][n_cells] addressed with block_stride / frame_stride as rr_assemble_frames_device does)."""
    import numpy as np
    for n_angles, n_dev, n_frames, C_ in ((400, 8, 3, 5), (400, 7, 2, 4), (10, 4, 3, 2), (5, 8, 1, 3), (400, 1, 4, 2), (401, 3, 2, 2)):
        eq, bpd, so, ro, pb = native_lib.multi_plan(n_angles, C_, n_dev, n_frames)
        blocks = [native_lib.partition(n_angles, n_dev, r) for r in range(n_dev)]
        assert eq == (len({e - b for b, e in blocks}) == 1)
        bufs = []
        for r, (b, e) in enumerate(blocks):        # tag = frame * 1e6 + azimuth * 1e2 + cell
            blk = np.zeros((n_frames, e - b, C_), np.int64)
            for f in range(n_frames):
                for a in range(b, e):
                    blk[f, a - b] = f * 1_000_000 + a * 100 + np.arange(C_)
            bufs.append(blk.ravel())
        want = np.zeros((n_frames, n_angles, C_), np.int64)
        for f in range(n_frames):
            for a in range(n_angles):
                want[f, a] = f * 1_000_000 + a * 100 + np.arange(C_)
        # ragged plan (valid for equal blocks too): pieces
        root = np.full(n_frames * n_angles * C_, -1, np.int64)
        for r in range(n_dev):
            for f in range(n_frames):
                n = int(pb[r, f])
                root[int(ro[r, f]):int(ro[r, f]) + n] = bufs[r][int(so[r, f]):int(so[r, f]) + n]
        assert np.array_equal(root.reshape(want.shape), want)
        if eq:                                      # all-gather layout + the assemble kernel's block addressing
            n_loc = blocks[0][1] - blocks[0][0]
            assert bpd == n_frames * n_loc * C_
            gathered = np.concatenate(bufs)         # [device][frame][n_loc][cells]
            block_stride, frame_stride = bpd, n_loc * C_
            got = np.zeros_like(want)
            for f in range(n_frames):
                for a in range(n_angles):
                    off = (a // n_loc) * block_stride + f * frame_stride + (a % n_loc) * C_
                    got[f, a] = gathered[off:off + C_]
            assert np.array_equal(got, want)


def test_environment_switches_are_documented():
    """Every RR_* environment switch the library reads is listed in the header's table, and the table lists no switch
    that nothing reads."""
    import glob
    import re
    csrc = os.path.join(ROOT, "radarays_ros_amd", "csrc")
    used = set()
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.h")):
        used |= set(re.findall(r'getenv\("(RR_[A-Z0-9_]+)"\)', open(f).read()))
    text = open(os.path.join(ROOT, "include", "radarays_mi355.h")).read()
    block = text[text.index("---- environment switches"):]
    listed = set(re.findall(r"RR_[A-Z0-9_]+", block))
    # shorthand of the table: "RR_BVH_ALPHA / _BETA / _BUDGET / _WZ"
    for suffix in re.findall(r"/ (_[A-Z]+)", block):
        listed.add("RR_BVH" + suffix)
    assert used - listed == set(), "undocumented: %s" % sorted(used - listed)
    assert listed - used == set(), "documented but unused: %s" % sorted(listed - used)


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md (the reference-side binding a maintainer would add) has a row for every function the header declares
    -- literally, or through the table's shorthands `rr_multi_set_*`, `name[_suffix]`."""
    import re
    header = open(os.path.join(ROOT, "include", "radarays_mi355.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", header))
    named = set(re.findall(r"rr_[a-z0-9_]+", doc))
    for base, opt in re.findall(r"(rr_[a-z0-9_]+)\[(_[a-z0-9_]+)\]", doc):      # rr_simulate_param_sets[_device]
        named.add(base + opt)
    for base, opt, tail in re.findall(r"(rr_[a-z0-9_]+)\[(_[a-z0-9_]+)\](_[a-z0-9_]+)", doc):   # rr_multi_simulate[_batch]_async
        named.add(base + opt + tail); named.add(base + tail)
    missing = sorted(d for d in declared if d not in named and not (d.startswith("rr_multi_set_") and "rr_multi_set_*" in doc))
    assert missing == [], "no row in INTEGRATION.md: %s" % missing


def test_bench_refuses_more_gpus_than_the_box_has_at_once():
    """`python bench.py --gpus 8` (no launcher): the parent counts devices WITHOUT touching the GPU, and where there are
    fewer than asked for it exits non-zero with one clear line instead of starting ranks that would fail one by one."""
    import subprocess
    import sys
    import time
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "--gpus 64 asked for" in r.stderr and "not launched" in r.stderr
    assert r.stdout.strip() == ""
    assert time.time() - t0 < 60


def test_hand_declared_rccl_abi_matches_rccl_h():
    """csrc/rr_multi.hip loads librccl at run time and calls it through prototypes declared BY HAND: the
    parameter lists and the datatype constant are compared with the image's rccl.h here, so that a drift shows on the CPU and
    not inside the first 8-GPU run.  (At run time rr_create_multi also refuses a library outside NCCL [2.7, 3.0) and sends 16
    guarded bytes through the constant it takes for ncclUint8.)"""
    import re
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h in this image")
    h = open(hdr).read()
    src = open(os.path.join(ROOT, "radarays_ros_amd", "csrc", "rr_multi.hip")).read()
    assert re.search(r"ncclUint8\s*=\s*1\b", h) and re.search(r"kNcclUint8\s*=\s*1\b", src)
    major, minor = int(re.search(r"#define NCCL_MAJOR (\d+)", h).group(1)), int(re.search(r"#define NCCL_MINOR (\d+)", h).group(1))
    assert major == 2 and minor >= 7                                  # the range rr_create_multi accepts
    norm = lambda s: re.sub(r"\s+", "", re.sub(r"\b(sendbuff|recvbuff|count|datatype|peer|comm|stream|ndev|devlist|result|version)\b", "", s))   # noqa: E731
    want = {"CommInitAll": "(ncclComm_t*, int, const int*)", "CommDestroy": "(ncclComm_t)",
            "Send": "(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)",
            "Recv": "(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)",
            "GroupStart": "()", "GroupEnd": "()", "GetErrorString": "(ncclResult_t)", "GetVersion": "(int*)"}
    for name, params in want.items():
        m = re.search(r"\bnccl%s\s*\(([^)]*)\)\s*;" % name, h)
        assert m, name
        assert norm("(" + m.group(1) + ")") == norm(params), (name, m.group(1))
        d = re.search(r"\(\*%s\)\s*\(([^)]*)\)" % name, src)                       # the hand-written declaration
        assert d and norm("(" + d.group(1) + ")") == norm(params), (name, d and d.group(1))
    assert re.search(r"const char\*\s*ncclGetErrorString", h) and re.search(r"const char\* \(\*GetErrorString\)", src)
