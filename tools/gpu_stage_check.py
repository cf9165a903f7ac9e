#!/usr/bin/env python3
"""Stage-by-stage parity report HIP vs CPU oracle (diagnostic; the asserting version is tests/test_gpu_parity.py).
Runs on the GPU box: python tools/gpu_stage_check.py [W H nfeatures nframes]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from orb_slam2_detailed_comments_amd import ORBextractor, synth


def main():
    W, H, NF, B = (int(a) for a in (sys.argv[1:5] + [640, 480, 1000, 2][len(sys.argv) - 1:]))
    frames = synth.stream(W, H, B, stream_id=3)
    ex = ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B)
    t = time.time()
    res = ex.extract_batch(frames)
    print(f"GPU extract_batch {B} frames: {time.time() - t:.3f}s, counts {[len(k) for k, _ in res]}")
    orc = oracle.OracleExtractor(NF, 1.2, 8, 20, 7)
    bad_total = 0
    for f in range(B):
        n, okps, odesc = orc.extract(frames[f])
        for l in range(8):
            a, b = ex.pyramid_level(l, f), orc.level_image(l)
            d = int((a != b).sum()) if a.shape == b.shape else -1
            oc = orc.level_candidates(l)
            gc = ex.debug_candidates(l, f)
            so = sorted(zip(oc['x'].astype(int), oc['y'].astype(int), oc['response'].astype(int)))
            sg = sorted(zip(gc['x'].astype(int), gc['y'].astype(int), gc['response'].astype(int)))
            ok_l = orc.level_keypoints(l)
            gk_l = ex.debug_level_keypoints(l, f)
            same_pos = len(ok_l) == len(gk_l) and bool(np.all(ok_l['x'] == gk_l['x']) and np.all(ok_l['y'] == gk_l['y'])
                                                        and np.all(ok_l['response'] == gk_l['response']))
            same_ang = same_pos and bool(np.all(ok_l['angle'].view(np.uint32) == gk_l['angle'].view(np.uint32)))
            ob = orc.level_image(l, blur=True)
            gb = ex.pyramid_level(l, f, blur=True)
            db = int((ob != gb).sum()) if ob is not None and ob.shape == gb.shape else -1
            flag = (d != 0) or (so != sg) or (not same_ang) or (db != 0)
            bad_total += flag
            print(f"f{f} L{l}: pyr diff {d:6d} | cand oracle {len(so):5d} gpu {len(sg):5d} same={so == sg} | "
                  f"kps oracle {len(ok_l):4d} gpu {len(gk_l):4d} pos={same_pos} angle={same_ang} | blur diff {db}")
            if so != sg and len(so) and len(sg):
                so_s, sg_s = set(so), set(sg)
                print("    only oracle:", sorted(so_s - sg_s)[:5], " only gpu:", sorted(sg_s - so_s)[:5])
        gk, gd = res[f]
        same_n = n == len(gk)
        same_k = same_n and gk.tobytes() == okps.tobytes()
        same_d = same_n and gd.tobytes() == odesc.tobytes()
        if same_n and not same_d:
            print("    descriptor rows differing:", int((gd != odesc).any(axis=1).sum()), "of", n)
        if same_n and not same_k:
            for fld in gk.dtype.names:
                print("    kp field", fld, "diff rows:", int((gk[fld] != okps[fld]).sum()))
        bad_total += (not same_k) + (not same_d)
        print(f"f{f} FINAL: n oracle {n} gpu {len(gk)} keypoints_bitexact={same_k} descriptors_bitexact={same_d}")
    print("STAGE CHECK", "ALL OK" if bad_total == 0 else f"{bad_total} stage mismatches")
    return 0 if bad_total == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
