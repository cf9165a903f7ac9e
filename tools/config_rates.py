#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (3: KITTI stereo geometry, 4: EuRoC stereo geometry,
5: 1080p) on one MI355X; numbers quoted in DESIGN.md.  bench.py stays on config 2."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from orb_slam2_detailed_comments_amd import ORBextractor, Frame, synth

def bench(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args, capture_output=True, text=True)
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    return j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["end_to_end_frac"]

for name, a in (("config 5: 1920x1080 / 4000 features, batch 32", ["--width", "1920", "--height", "1080", "--nfeatures", "4000", "--batch", "32", "--steps", "6"]),
                ("config 3 geometry: 1241x376 / 2000 features (one eye), batch 128", ["--width", "1241", "--height", "376", "--nfeatures", "2000", "--batch", "128", "--steps", "6"]),
                ("config 4 geometry: 752x480 / 1200 features (one eye), batch 128", ["--width", "752", "--height", "480", "--nfeatures", "1200", "--batch", "128", "--steps", "6"])):
    v, ms, k, frac = bench(a)
    print(f"{name}: {v:.0f} frames/s, {ms:.3f} ms/step, dominant {k}, end-to-end {100 * frac:.1f} % of HBM peak")

# stereo: extract both eyes + ComputeStereoMatches (host-buffer entry points, synchronous)
for (w, h, nf, mb, mbf, tag) in ((1241, 376, 2000, 0.537, 386.1448, "KITTI 00"), (752, 480, 1200, 0.11, 47.9, "EuRoC MH_01")):
    L, R = synth.stereo_pair(w, h, stream_id=6)
    exL, exR = ORBextractor(nf), ORBextractor(nf)
    kL, dL = exL(L); kR, dR = exR(R)
    FL, FR = Frame(kL, dL, w, h), Frame(kR, dR, w, h)
    FL.ComputeStereoMatches(FR, exL, exR, mb, mbf)
    t = time.perf_counter(); n = 50
    for _ in range(n):
        nm = FL.ComputeStereoMatches(FR, exL, exR, mb, mbf)
    dt = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n):
        exL(L); exR(R)
    de = (time.perf_counter() - t) / n
    print(f"{tag} stereo {w}x{h}/{nf}: ComputeStereoMatches {dt * 1e6:.0f} us per pair ({nm} matches, host buffers, synchronous); "
          f"two synchronous single-frame extractions {de * 1e6:.0f} us")
