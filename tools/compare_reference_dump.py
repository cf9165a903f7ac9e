#!/usr/bin/env python3
"""Compares a dump of the REFERENCE's ORBextractor (tools/dump_reference_frame.cc, run inside an ORB-SLAM2 tree) with this
library (and its CPU oracle) on the same image: pyramid levels, keypoint count / coordinates / angles / responses, descriptors.
This is the one-shot check that would pin the parity claim against a real ORB-SLAM2 + OpenCV build; nothing
in this repository's tests depends on it, because the build image has neither.

    python tools/compare_reference_dump.py dump.bin [--fp-mode 0|1]
"""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def load(path):
    b = open(path, "rb").read()
    assert b[:8] == b"ORBXREF1", "not a dump_reference_frame file"
    w, h, nf, nl, ini, mn = struct.unpack_from("<6i", b, 8)
    sf, = struct.unpack_from("<f", b, 32)
    n, = struct.unpack_from("<i", b, 36)
    o = 40
    img = np.frombuffer(b, np.uint8, w * h, o).reshape(h, w); o += w * h
    from orb_slam2_detailed_comments_amd._capi import KP_DTYPE
    kps = np.frombuffer(b, KP_DTYPE, n, o); o += 28 * n
    desc = np.frombuffer(b, np.uint8, 32 * n, o).reshape(n, 32); o += 32 * n
    pyr = []
    for _ in range(nl):
        c, r = struct.unpack_from("<2i", b, o); o += 8
        pyr.append(np.frombuffer(b, np.uint8, c * r, o).reshape(r, c)); o += c * r
    return dict(w=w, h=h, nf=nf, nl=nl, ini=ini, mn=mn, sf=sf, img=img, kps=kps, desc=desc, pyr=pyr)


def main():
    d = load(sys.argv[1])
    fp = int(sys.argv[sys.argv.index("--fp-mode") + 1]) if "--fp-mode" in sys.argv else 0
    from orb_slam2_detailed_comments_amd import ORBextractor
    ex = ORBextractor(d["nf"], d["sf"], d["nl"], d["ini"], d["mn"], fp_mode=fp)
    k, de = ex(d["img"])
    ok = True
    for l in range(d["nl"]):
        g = ex.pyramid_level(l)
        same = g.shape == d["pyr"][l].shape and np.array_equal(g, d["pyr"][l])
        ok &= same
        if not same:
            diff = int((g != d["pyr"][l]).sum()) if g.shape == d["pyr"][l].shape else -1
            print(f"pyramid level {l}: DIFFERENT ({diff} pixels; shapes {g.shape} vs {d['pyr'][l].shape})")
    print(f"keypoints: reference {len(d['kps'])}, this library {len(k)}")
    if len(k) == len(d["kps"]):
        for f in k.dtype.names:
            same = np.array_equal(k[f].view(np.uint32), d["kps"][f].view(np.uint32))
            ok &= same
            if not same:
                print(f"  field {f}: {int((k[f].view(np.uint32) != d['kps'][f].view(np.uint32)).sum())} keypoints differ")
        rows = int((de != d["desc"]).any(axis=1).sum())
        ok &= rows == 0
        print(f"  descriptors: {rows} rows differ ({int(np.unpackbits(de ^ d['desc']).sum())} bits)")
    else:
        ok = False
        a = set(zip(k["x"].tolist(), k["y"].tolist(), k["octave"].tolist())); b = set(zip(d["kps"]["x"].tolist(), d["kps"]["y"].tolist(), d["kps"]["octave"].tolist()))
        print(f"  common (x, y, octave): {len(a & b)}; only here {len(a - b)}; only in the reference {len(b - a)}")
    print("BIT-IDENTICAL to the reference dump" if ok else "differences found (see above; DESIGN.md section 2 lists the OpenCV-version-dependent stages)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
