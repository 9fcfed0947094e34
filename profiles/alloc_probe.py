#!/usr/bin/env python3
"""How long does hipMalloc of a large buffer take in a process that starts right after another one released as much?
   alloc_probe.py            the driver: runs the child back to back for a few sizes, with and without the child touching the buffer
   alloc_probe.py child GiB touch
(The stage's lanes allocate their direction-flag workspace when they are set up: profiles/r04_stage.txt.)"""
import ctypes as C, subprocess, sys, time

def child(gib, touch):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    p = C.c_void_p()
    t0 = time.time(); hip.hipSetDevice(0); hip.hipFree(None); t_init = time.time() - t0
    n = int(gib * (1 << 30))
    t0 = time.time(); rc = hip.hipMalloc(C.byref(p), n); t_alloc = time.time() - t0
    t_touch = 0.0
    if touch:
        t0 = time.time(); hip.hipMemset(p, 1, n); hip.hipDeviceSynchronize(); t_touch = time.time() - t0
    t0 = time.time(); hip.hipFree(p); t_free = time.time() - t0
    print("  %5.1f GiB touch=%d: init %6.1f ms  hipMalloc %7.1f ms (rc %d)  memset %6.1f ms  hipFree %6.1f ms" % (
        gib, touch, t_init * 1e3, t_alloc * 1e3, rc, t_touch * 1e3, t_free * 1e3), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(float(sys.argv[2]), int(sys.argv[3]))
else:
    for gib, touch in ((8, 0), (8, 1), (24, 0), (24, 1), (64, 1), (2, 1)):
        print("%g GiB, touch=%d, six processes back to back:" % (gib, touch), flush=True)
        for _ in range(6):
            subprocess.run([sys.executable, __file__, "child", str(gib), str(touch)])
