#!/usr/bin/env python3
"""FAST candidate diff HIP vs CPU oracle for one frame (diagnostic): python tools/fast_diff.py [W H nfeatures stream_id]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from orb_slam2_detailed_comments_amd import ORBextractor, synth


def main():
    W, H, NF, SID = (int(a) for a in (sys.argv[1:5] + [640, 480, 1000, 0][len(sys.argv) - 1:]))
    frames = synth.stream(W, H, 1, stream_id=SID)
    ex = ORBextractor(NF, 1.2, 8, 20, 7, max_batch=1)
    ex.extract_batch(frames)
    orc = oracle.OracleExtractor(NF, 1.2, 8, 20, 7)
    orc.extract(frames[0])
    for l in range(8):
        oc, gc = orc.level_candidates(l), ex.debug_candidates(l, 0)
        so = set(zip(oc['x'].astype(int), oc['y'].astype(int), oc['response'].astype(int)))
        sg = set(zip(gc['x'].astype(int), gc['y'].astype(int), gc['response'].astype(int)))
        miss, extra = sorted(so - sg), sorted(sg - so)
        print(f"L{l}: oracle {len(so)} gpu {len(sg)} missing {len(miss)} extra {len(extra)}")
        if miss:
            xs = np.array([m[0] for m in miss]); ys = np.array([m[1] for m in miss])
            print("   missing x hist (bins of 31):", np.bincount(xs // 31, minlength=1).tolist())
            print("   missing y hist (bins of 31):", np.bincount(ys // 31, minlength=1).tolist())
            print("   missing x % 3:", np.bincount(xs % 3, minlength=3).tolist(), " first:", miss[:8])
        if extra:
            print("   extra first:", extra[:8])
        if so:
            xs = np.array([m[0] for m in so])
            print("   oracle x hist (bins of 31):", np.bincount(xs // 31, minlength=1).tolist())


if __name__ == "__main__":
    main()
