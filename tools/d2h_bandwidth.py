import ctypes as C, numpy as np, time
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
def chk(r): assert r == 0, r
n = 805306368
host = np.empty(n, dtype=np.uint8); host[:] = 1   # touch pages
dev = C.c_void_p()
chk(hip.hipMalloc(C.byref(dev), C.c_size_t(n)))
chk(hip.hipMemset(dev, 7, C.c_size_t(n)))
chk(hip.hipDeviceSynchronize())
for rep in range(3):
    t0=time.perf_counter(); chk(hip.hipMemcpy(C.c_void_p(host.ctypes.data), dev, C.c_size_t(n), 2)); t1=time.perf_counter()
    print("pageable D2H %.1f ms %.1f GB/s" % ((t1-t0)*1e3, n/(t1-t0)/1e9))
for rep in range(3):
    t0=time.perf_counter(); chk(hip.hipHostRegister(C.c_void_p(host.ctypes.data), C.c_size_t(n), 0)); t1=time.perf_counter()
    chk(hip.hipMemcpy(C.c_void_p(host.ctypes.data), dev, C.c_size_t(n), 2)); t2=time.perf_counter()
    chk(hip.hipHostUnregister(C.c_void_p(host.ctypes.data))); t3=time.perf_counter()
    print("register %.1f ms, pinned D2H %.1f ms (%.1f GB/s), unregister %.1f ms" % ((t1-t0)*1e3,(t2-t1)*1e3,n/(t2-t1)/1e9,(t3-t2)*1e3))
# pinned staging alloc
p = C.c_void_p()
chk(hip.hipHostMalloc(C.byref(p), C.c_size_t(64<<20), 0))
t0=time.perf_counter()
for off in range(0, n, 64<<20):
    sz=min(64<<20, n-off)
    chk(hip.hipMemcpy(p, C.c_void_p(dev.value+off), C.c_size_t(sz), 2))
t1=time.perf_counter(); print("D2H into pinned staging only: %.1f ms %.1f GB/s" % ((t1-t0)*1e3, n/(t1-t0)/1e9))
src = (C.c_uint8 * (64<<20)).from_address(p.value)
t0=time.perf_counter()
for off in range(0, n, 64<<20):
    sz=min(64<<20, n-off)
    C.memmove(host.ctypes.data+off, p, sz)
t1=time.perf_counter(); print("memcpy pinned->pageable 1 thread: %.1f ms %.1f GB/s" % ((t1-t0)*1e3, n/(t1-t0)/1e9))
