#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (orbx_extract_batch: H2D of frames + D2H of results),
reported in DESIGN.md next to the HBM-resident bench value (which is what bench.py's `value` is)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_detailed_comments_amd import ORBextractor, synth
B = 64
frames = synth.stream(640, 480, B, stream_id=100)
ex = ORBextractor(1000, max_batch=B)
ex.extract_batch(frames)
t = time.perf_counter(); n = 10
for _ in range(n):
    ex.extract_batch(frames)
dt = time.perf_counter() - t
print(f"host-buffer path: {B * n / dt:.0f} frames/s ({dt / n * 1e3:.2f} ms per {B}-frame batch, pageable host memory)")
ex1 = ORBextractor(1000, max_batch=1)
ex1(frames[0])
t = time.perf_counter()
for i in range(200):
    ex1(frames[i % B])
dt = time.perf_counter() - t
print(f"single-frame latency path (batch 1, host buffers, python call overhead included): {200 / dt:.0f} frames/s, {dt / 200 * 1e6:.0f} us/frame")
