"""One seed of fuzz_trace.py in detail: which ray, which triangles, under which switches.  usage: fuzz_trace_debug.py SEED"""
import sys, os, ctypes as C, subprocess, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle
from fuzz_trace import scene, rays
seed = int(sys.argv[1])
rs = np.random.RandomState(seed)
v, f = scene(rs); o, d = rays(rs, v, f)
brute = oracle.Scene(v, f, None, use_bvh=0)
want = [brute.intersect(o[i], d[i]) for i in range(len(o))]
wt = np.array([w[0] if w else -1.0 for w in want], np.float32); wf = np.array([w[1] if w else 0xFFFFFFFF for w in want], np.uint32)
print("scene: %d triangles, extent %s .. %s" % (len(f), v.min(0), v.max(0)))


def trace(lib_path, builder, env):
    code = r'''
import sys, ctypes as C, numpy as np
L = C.CDLL(sys.argv[1]); L.rr_create.restype = C.c_void_p
v = np.load(sys.argv[3]); f = np.load(sys.argv[4]); o = np.load(sys.argv[5]); d = np.load(sys.argv[6])
c = C.c_void_p(L.rr_create(0))
fn = L.rr_set_mesh if sys.argv[2] == "host" else L.rr_set_mesh_gpu
rc = fn(c, C.c_void_p(v.ctypes.data), C.c_size_t(len(v)), C.c_void_p(f.ctypes.data), C.c_size_t(len(f)), None); assert rc == 0, rc
t = np.zeros(len(o), np.float32); face = np.zeros(len(o), np.uint32)
rc = L.rr_debug_trace(c, C.c_void_p(o.ctypes.data), C.c_void_p(d.ctypes.data), C.c_size_t(len(o)), C.c_void_p(t.ctypes.data), C.c_void_p(face.ctypes.data)); assert rc == 0, rc
np.save(sys.argv[7], t); np.save(sys.argv[8], face)
L.rr_destroy(c)
'''
    tmp = "/tmp/ftd_%d_" % os.getpid()
    for n, a in (("v", v), ("f", f), ("o", o), ("d", d)):
        np.save(tmp + n + ".npy", a)
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, "-c", code, lib_path, builder, tmp + "v.npy", tmp + "f.npy", tmp + "o.npy", tmp + "d.npy", tmp + "t.npy", tmp + "face.npy"],
                   check=True, env=e)
    return np.load(tmp + "t.npy"), np.load(tmp + "face.npy")


libs = {"current": os.path.join(R, "radarays_ros_amd", "libradarays_mi355.so"), "round3": os.path.join(R, "radarays_ros_amd", "libradarays_mi355_r03pure.so")}
for name, path in libs.items():
    if not os.path.exists(path):
        continue
    for builder in ("host", "gpu"):
        for env in ({}, {"RR_CULL_POP": "0"}, {"RR_BVH_CHOOSE": "0"}, {"RR_BVH_ALPHA": "-1"}, {"RR_STACK_LDS": "4"}):
            t, face = trace(path, builder, env)
            hit = wt >= 0
            bad = np.nonzero((t != wt) | ((face != wf) & hit))[0]
            print("%-8s %-5s %-22s mismatching rays: %d %s" % (name, builder, env, len(bad), bad[:5]))
            for i in bad[:2]:
                print("   ray %d o %s d %s: gpu (%r, %d) oracle (%r, %d)" % (i, o[i], d[i], t[i], face[i], wt[i], wf[i]))
                for fc in (int(face[i]), int(wf[i])):
                    if fc < len(f):
                        tri = v[f[fc]]
                        print("      face %d: size %.4g, vertices %s" % (fc, np.abs(tri - tri.mean(0)).max(), tri.tolist()))
