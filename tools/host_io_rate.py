#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point orbx_extract_batch (upload of the frames + download of the results):
pageable vs page-locked host memory (orbx_host_alloc), one chunk vs pipelined chunks; also the synchronous single-frame call.
Reported in DESIGN.md next to the HBM-resident bench value (which is what bench.py's `value` is)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_detailed_comments_amd import ORBextractor, synth, _capi

N, W, H = int(os.environ.get("HOST_IO_FRAMES", "256")), 640, 480
frames = synth.stream(W, H, 64, stream_id=100)
frames = np.concatenate([frames] * (N // 64))
L = _capi.lib()
for mem in ("pageable", "pinned"):
    for mb in (256, 64, 32):
        ex = ORBextractor(1000, max_batch=mb)
        cap = ex.max_keypoints(W, H)
        if mem == "pinned":
            keep = [_capi.PinnedArray((N, H, W)), _capi.PinnedArray((N, cap), _capi.KP_DTYPE), _capi.PinnedArray((N, cap, 32)),
                    _capi.PinnedArray((N,), np.int32)]
            img, kps, desc, cnt = (k.array for k in keep)
            img[...] = frames
        else:
            img, kps, desc, cnt = frames.copy(), np.zeros((N, cap), _capi.KP_DTYPE), np.zeros((N, cap, 32), np.uint8), np.zeros(N, np.int32)
        call = lambda: _capi.check(L.orbx_extract_batch(ex.handle, N, _capi.ptr(img), W, H, W, W * H, _capi.ptr(kps), _capi.ptr(desc), _capi.ptr(cnt), cap))
        call(); call()
        t = time.perf_counter(); n = 6
        for _ in range(n): call()
        dt = (time.perf_counter() - t) / n
        print(f"{mem:8s} host memory, {N} frames per call in chunks of {mb:3d}: {N / dt:8.0f} frames/s ({dt * 1e3:.2f} ms per call), mean keypoints {cnt.mean():.0f}")
        del ex
ex1 = ORBextractor(1000, max_batch=1)
ex1(frames[0])
t = time.perf_counter()
for i in range(200):
    ex1(frames[i % 64])
dt = time.perf_counter() - t
print(f"single-frame synchronous call (batch 1, pageable host buffers, python call overhead included): {200 / dt:.0f} frames/s, {dt / 200 * 1e6:.0f} us/frame")
