"""D2H copy rates of this box: torch pinned memory vs rr_host_alloc (hipHostMalloc), idle GPU."""
import sys, os, time, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from radarays_ros_amd import native
L = native.lib()
hip = ctypes.CDLL("libamdhip64.so")
for mb in (1.37, 11.0, 64.0):
    n = int(mb * 1e6)
    d = torch.zeros(n, dtype=torch.uint8, device="cuda")
    hp = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    hr = native.HostImages((n,))
    hpage = torch.empty(n, dtype=torch.uint8)
    s = torch.cuda.Stream()
    for name, ptr in (("torch pinned", hp.data_ptr()), ("rr_host_alloc", hr.ptr), ("pageable", hpage.data_ptr())):
        for _ in range(3):
            hip.hipMemcpyAsync(ctypes.c_void_p(ptr), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), 2, ctypes.c_void_p(s.cuda_stream))
        s.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            hip.hipMemcpyAsync(ctypes.c_void_p(ptr), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), 2, ctypes.c_void_p(s.cuda_stream))
        s.synchronize()
        dt = (time.perf_counter() - t0) / K
        print("%6.2f MB  %-14s %7.1f us  %6.2f GB/s" % (mb, name, dt * 1e6, n / dt / 1e9), flush=True)
    hr.close()
