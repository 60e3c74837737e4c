#!/usr/bin/env python3
"""tools/pcie_pinning.py: what the host entry points pay per 1920x1080 double plane (16.6 MB) -- pageable hipMemcpy, hipHostRegister +
copy + unregister, copy from / to memory that is already pinned -- to decide how ofx_tvl1_multiscale stages its planes."""
import ctypes as C, time, sys
import numpy as np
hip = C.CDLL("libamdhip64.so")
n = 1920 * 1080 * 8
dptr = C.c_void_p()
assert hip.hipMalloc(C.byref(dptr), n) == 0
pin = C.c_void_p()
assert hip.hipHostMalloc(C.byref(pin), n, 0) == 0
a = np.random.rand(1920 * 1080)
b = np.empty_like(a)
H2D, D2H = 1, 2


def t(fn, reps=10):
    fn()
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    hip.hipDeviceSynchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("pageable H2D  %.3f ms" % t(lambda: hip.hipMemcpy(dptr, a.ctypes.data_as(C.c_void_p), n, H2D)))
print("pageable D2H  %.3f ms" % t(lambda: hip.hipMemcpy(b.ctypes.data_as(C.c_void_p), dptr, n, D2H)))
print("pinned   H2D  %.3f ms" % t(lambda: hip.hipMemcpy(dptr, pin, n, H2D)))
print("pinned   D2H  %.3f ms" % t(lambda: hip.hipMemcpy(pin, dptr, n, D2H)))
print("register      %.3f ms" % t(lambda: (hip.hipHostRegister(a.ctypes.data_as(C.c_void_p), n, 0), hip.hipHostUnregister(a.ctypes.data_as(C.c_void_p)))))


def reg_copy():
    p = a.ctypes.data_as(C.c_void_p)
    hip.hipHostRegister(p, n, 0)
    hip.hipMemcpy(dptr, p, n, H2D)
    hip.hipHostUnregister(p)


print("register + H2D + unregister %.3f ms" % t(reg_copy))
print("host memcpy into pinned     %.3f ms" % t(lambda: C.memmove(pin, a.ctypes.data_as(C.c_void_p), n)))
