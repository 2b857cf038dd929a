"""Host-side SAH builder (csrc/rr_bvh.cpp) under AddressSanitizer + UBSan on the CPU (GPU ASan is not
available on the pool): structural invariants of the BVH4 the traversal kernel walks, degenerate and
invalid inputs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bvh_builder_invariants_under_asan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("g++ missing")
    exe = str(tmp_path / "bvh_host_check")
    csrc = os.path.join(ROOT, "radarays_ros_amd", "csrc")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-I", csrc, os.path.join(ROOT, "tests", "cpp", "bvh_host_check.cpp"),
                    os.path.join(csrc, "rr_bvh.cpp"), "-o", exe, "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
    assert r.stdout.count(": ok") == 12


def test_bvh_builder_parallel_paths_under_tsan(tmp_path):
    """ThreadSanitizer over a build that is large enough for the parallel paths (subtree tasks, chunk-parallel loops, the
    per-build block pool, per-thread node / leaf slot blocks): no data race, invariants hold, twice in one process."""
    if shutil.which("g++") is None:
        pytest.skip("g++ missing")
    exe = str(tmp_path / "bvh_host_check_tsan")
    csrc = os.path.join(ROOT, "radarays_ros_amd", "csrc")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-fno-omit-frame-pointer",
                    "-I", csrc, os.path.join(ROOT, "tests", "cpp", "bvh_host_check.cpp"),
                    os.path.join(csrc, "rr_bvh.cpp"), "-o", exe, "-lpthread"], check=True)
    r = subprocess.run([exe, "big"], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr, r.stderr
    assert r.stdout.count(": ok") == 2
