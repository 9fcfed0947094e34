#!/usr/bin/env python3
"""How long does pinned host memory take to get?  hipHostMalloc (flags), hipHostRegister of malloc'd / huge-page memory,
several allocations side by side -- the question behind sdf_pool_host (a 182 MB character pool per stage lane)."""
import ctypes as C
import mmap
import sys
import threading
import time

hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
libc = C.CDLL("libc.so.6")
libc.aligned_alloc.restype = C.c_void_p
libc.aligned_alloc.argtypes = [C.c_size_t, C.c_size_t]
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
libc.memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
libc.free.argtypes = [C.c_void_p]
assert hip.hipSetDevice(0) == 0
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), 256 << 20) == 0  # (initialises the runtime)
MB = 1 << 20
N = int(sys.argv[1]) * MB if len(sys.argv) > 1 else 182 * MB


def t(label, f):
    t0 = time.perf_counter()
    r = f()
    print("%-70s %7.1f ms" % (label, 1e3 * (time.perf_counter() - t0)), flush=True)
    return r


def host_malloc(n, flags):
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), n, flags) == 0
    return p


for rep in range(2):
    for flags, name in ((0, "default"), (0x80000000, "hipHostMallocNonCoherent"), (0x2, "hipHostMallocMapped"), (0x1, "Portable")):
        p = t("hipHostMalloc %d MB, %s" % (N // MB, name), lambda: host_malloc(N, flags))
        t("  first upload from it", lambda: hip.hipMemcpy(d, p, N, 1))
        t("  second upload", lambda: hip.hipMemcpy(d, p, N, 1))
        t("  hipHostFree", lambda: hip.hipHostFree(p))
    q = t("aligned_alloc(2 MB) + MADV_HUGEPAGE + memset", lambda: (lambda a: (libc.madvise(a, N, 14), libc.memset(a, 1, N), a)[2])(libc.aligned_alloc(2 * MB, N)))
    t("  hipHostRegister of it", lambda: hip.hipHostRegister(q, N, 0))
    t("  upload from it", lambda: hip.hipMemcpy(d, q, N, 1))
    t("  hipHostUnregister", lambda: hip.hipHostUnregister(q))
    libc.free(q)
    q = t("aligned_alloc(4 KB) + memset (small pages)", lambda: (lambda a: (libc.madvise(a, N, 15), libc.memset(a, 1, N), a)[2])(libc.aligned_alloc(4096, N)))
    t("  hipHostRegister of it", lambda: hip.hipHostRegister(q, N, 0))
    t("  hipHostUnregister", lambda: hip.hipHostUnregister(q))
    t("  pageable upload from it", lambda: hip.hipMemcpy(d, q, N, 1))
    libc.free(q)
    ps = []

    def one(n):
        ps.append(host_malloc(n, 0))
    def side_by_side():
        th = [threading.Thread(target=one, args=(N // 8,)) for _ in range(8)]
        [x.start() for x in th]
        [x.join() for x in th]
    t("8 x hipHostMalloc %d MB on 8 threads" % (N // 8 // MB), side_by_side)
    for p in ps:
        hip.hipHostFree(p)
    p = t("hipHostMalloc 16 MB", lambda: host_malloc(16 * MB, 0))
    hip.hipHostFree(p)

# ---- round 6, second question: registered BEFORE the pages are written (what sdf_pool_host does at set-up), then filled, then an
# asynchronous upload on a stream: how long does the ENQUEUE take, how long the copy?
hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
st = C.c_void_p()
assert hip.hipStreamCreate(C.byref(st)) == 0
for order in ("register, then write", "write, then register"):
    q = libc.aligned_alloc(2 * MB, N)
    libc.madvise(q, N, 14)
    if order.startswith("register"):
        t("[%s] hipHostRegister of untouched huge-page memory" % order, lambda: hip.hipHostRegister(q, N, 0))
        t("  memset", lambda: libc.memset(q, 1, N))
    else:
        t("[%s] memset" % order, lambda: libc.memset(q, 1, N))
        t("  hipHostRegister", lambda: hip.hipHostRegister(q, N, 0))
    for rep in range(3):
        t("  hipMemcpyAsync enqueue", lambda: hip.hipMemcpyAsync(d, q, N, 1, st))
        t("  ... stream synchronize", lambda: hip.hipStreamSynchronize(st))
    hip.hipHostUnregister(q)
    libc.free(q)
p = host_malloc(N, 0)
libc.memset(p, 1, N)
for rep in range(2):
    t("[hipHostMalloc] hipMemcpyAsync enqueue", lambda: hip.hipMemcpyAsync(d, p, N, 1, st))
    t("  ... stream synchronize", lambda: hip.hipStreamSynchronize(st))

# ---- third question: the ENQUEUE of an asynchronous device-to-host copy into hipHostMalloc memory, by size
hm = host_malloc(256 * MB, 0)
libc.memset(hm, 1, 256 * MB)
for mb in (4, 8, 16, 24, 32, 34, 48, 64, 128):
    for rep in range(2):
        t0 = time.perf_counter()
        hip.hipMemcpyAsync(hm, d, mb * MB, 2, st)
        t1 = time.perf_counter()
        hip.hipStreamSynchronize(st)
        t2 = time.perf_counter()
        print("D2H %4d MB into hipHostMalloc memory: enqueue %6.2f ms, then synchronize %6.2f ms" % (mb, 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
for mb in (34, 185):
    for rep in range(2):
        t0 = time.perf_counter()
        hip.hipMemcpyAsync(d, hm, mb * MB, 1, st)
        t1 = time.perf_counter()
        hip.hipStreamSynchronize(st)
        t2 = time.perf_counter()
        print("H2D %4d MB from hipHostMalloc memory: enqueue %6.2f ms, then synchronize %6.2f ms" % (mb, 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
