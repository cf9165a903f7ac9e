#!/usr/bin/env python3
"""30 single-frame orbx_extract calls (for rocprofv3 --kernel-trace: kernel durations and gaps at batch 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_detailed_comments_amd import ORBextractor, synth
frames = synth.stream(640, 480, 4, stream_id=100)
ex = ORBextractor(1000, max_batch=1)
for i in range(30): ex(frames[i % 4])
