"""Which engine carries D2H copies in a python process (rocprofv3 --kernel-trace --memory-copy-trace --stats around this
script)?  Everything through ctypes: hipMalloc, hipHostMalloc, hipStreamCreateWithFlags, 40 x hipMemcpyAsync of 11 MB.
argv[1]:  plain        no torch in the process (system runtime /opt/rocm/lib/libamdhip64.so.7)
          import       `import torch` first (its bundled runtime is then the one in the process), torch.cuda never touched
          init         import torch and torch.cuda.init()
          tensor       ... and one tensor allocated on the device
          stream       ... and one torch.cuda.Stream created"""
import ctypes as C
import sys

case = sys.argv[1] if len(sys.argv) > 1 else "plain"
if case != "plain":
    import torch
    if case in ("init", "tensor", "stream"):
        torch.cuda.init()
    if case in ("tensor", "stream"):
        t = torch.zeros(16, device="cuda:0")
        torch.cuda.synchronize()
    if case == "stream":
        ts = torch.cuda.Stream(device=torch.device("cuda", 0))
    path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
else:
    path = "/opt/rocm/lib/libamdhip64.so.7"
hip = C.CDLL(path)
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
N = 8 * 3424 * 400
d, h, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(d), N) == 0
assert hip.hipMemset(d, 7, N) == 0
assert hip.hipHostMalloc(C.byref(h), N, 0) == 0
assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
assert hip.hipDeviceSynchronize() == 0
for _ in range(40):
    assert hip.hipMemcpyAsync(h, d, N, 2, s) == 0
assert hip.hipStreamSynchronize(s) == 0
print("case", case, "runtime", path, "byte", C.cast(h, C.POINTER(C.c_ubyte))[12345])
