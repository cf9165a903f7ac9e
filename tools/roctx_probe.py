#!/usr/bin/env python3
"""Two small batched extractions + a match with ORBX_ROCTX=1, for `rocprofv3 --marker-trace --kernel-trace -- python3 tools/roctx_probe.py`:
the marker trace then carries one roctx range per stage (k_pyr_l0, k_pyr_resize, k_fast_rows, ...) around that stage's launches."""
import os, sys
os.environ["ORBX_ROCTX"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, synth
frames = synth.stream(640, 480, 4, stream_id=5)
ex = ORBextractor(1000, max_batch=4)
for _ in range(2):
    res = ex.extract_batch(frames)
m = ORBmatcher(0.9, True, extractor=ex)
m.match_bruteforce(res[1][1], res[0][1])
print("roctx probe done", [len(k) for k, _ in res])
