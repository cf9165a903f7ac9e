#!/usr/bin/env python3
"""Which side of the host-fed call makes page-locked memory slow?  orbx_extract_batch (256 frames, chunks of 64) with the INPUT
and the OUTPUT buffers page-locked or pageable independently.   python tools/host_io_mixed.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_detailed_comments_amd import ORBextractor, synth, _capi

N, W, H = 256, 640, 480
frames = np.concatenate([synth.stream(W, H, 64, stream_id=100)] * (N // 64))
L = _capi.lib()
for mb in (64, 256):
    for pin_in in (False, True):
        for pin_out in (False, True):
            ex = ORBextractor(1000, max_batch=mb)
            cap = ex.max_keypoints(W, H)
            keep = []
            def buf(shape, dt, pinned):
                if pinned:
                    p = _capi.PinnedArray(shape, dt); keep.append(p); return p.array
                return np.zeros(shape, dt)
            img = buf((N, H, W), np.uint8, pin_in); img[...] = frames
            kps, desc, cnt = buf((N, cap), _capi.KP_DTYPE, pin_out), buf((N, cap, 32), np.uint8, pin_out), buf((N,), np.int32, pin_out)
            call = lambda: _capi.check(L.orbx_extract_batch(ex.handle, N, _capi.ptr(img), W, H, W, W * H, _capi.ptr(kps), _capi.ptr(desc), _capi.ptr(cnt), cap))
            call(); call()
            ts = []
            for _ in range(8):
                t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
            print(f"chunks of {mb:3d}: input {'page-locked' if pin_in else 'pageable   '} output {'page-locked' if pin_out else 'pageable   '}: "
                  f"median {np.median(ts) * 1e3:.2f} ms  min {min(ts) * 1e3:.2f} ms  -> {N / np.median(ts):8.0f} frames/s", flush=True)
            del ex
