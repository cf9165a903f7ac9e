#!/usr/bin/env python3
"""Raw host <-> device copy rates of the box (page-locked vs pageable host memory, orbx_host_alloc = hipHostMalloc), next to
tools/host_io_rate.py: what the PCIe-inclusive frame rate can be at best.   python tools/pcie_probe.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_detailed_comments_amd import _capi

dev = torch.device("cuda", 0)
for mb in (4, 20, 80):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    pin = _capi.PinnedArray((n,))
    pinned = torch.from_numpy(pin.array)
    pageable = torch.from_numpy(np.zeros(n, np.uint8))
    tpin = torch.empty(n, dtype=torch.uint8).pin_memory()
    for name, hbuf in (("hipHostMalloc (orbx_host_alloc)", pinned), ("torch pin_memory", tpin), ("pageable", pageable)):
        for direction in ("H2D", "D2H"):
            def run():
                if direction == "H2D": d.copy_(hbuf, non_blocking=True)
                else: hbuf.copy_(d, non_blocking=True)
                torch.cuda.synchronize()
            run(); run()
            t = time.perf_counter()
            for _ in range(10): run()
            dt = (time.perf_counter() - t) / 10
            print(f"{mb:3d} MiB {direction} {name:32s} {n / dt / 1e9:7.1f} GB/s  ({dt * 1e6:8.0f} us)")
