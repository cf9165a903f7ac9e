#!/usr/bin/env python3
"""Single-frame latency of the drop-in path (one orbx_extract call per frame, host buffers, synchronous) with and without
the hipGraph replay of the launch sequence; also small batches through orbx_extract_batch."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from orb_slam2_detailed_comments_amd import ORBextractor, synth
    frames = synth.stream(640, 480, 16, stream_id=100)
    for B in (1, 4):
        ex = ORBextractor(1000, max_batch=B)
        for i in range(20): ex.extract_batch(frames[:B])
        t = time.perf_counter(); n = 300
        for i in range(n): ex.extract_batch(frames[(i % 4) * B:(i % 4) * B + B] if B < 4 else frames[:B])
        dt = (time.perf_counter() - t) / n
        print(f"  batch {B}: {dt * 1e6:7.1f} us per call, {B / dt:8.0f} frames/s")
    sys.exit(0)
for label, env in (("graph replay (default)", {}), ("plain launches", {"ORBX_GRAPH_MAX_BATCH": "0"})):
    print(label, flush=True)
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), check=True)
