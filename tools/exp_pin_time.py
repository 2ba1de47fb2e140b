#!/usr/bin/env python3
"""Diagnostic: how long does page-locking (hipHostRegister, portable) take by size, and unregistering?"""
import ctypes as C, time, mmap
hip = C.CDLL("libamdhip64.so")
hip.hipInit(0)
hip.hipSetDevice(0)
for mb in (64, 550, 2200, 4900):
    n = mb << 20
    buf = mmap.mmap(-1, n)
    buf[0:1] = b"x"
    addr = C.addressof(C.c_char.from_buffer(buf))
    C.memset(addr, 1, n)                      # touch every page first (the Decoder's buffer has been read into)
    t0 = time.perf_counter()
    st = hip.hipHostRegister(C.c_void_p(addr), C.c_size_t(n), C.c_uint(1))
    t1 = time.perf_counter()
    hip.hipHostUnregister(C.c_void_p(addr))
    t2 = time.perf_counter()
    print("%5d MB: register %.1f ms (status %d), unregister %.1f ms" % (mb, (t1 - t0) * 1e3, st, (t2 - t1) * 1e3), flush=True)
