#!/usr/bin/env python3
"""Generates tests/golden/*.npz / golden_index.json from the CPU oracle (authoring container).

The reference holds no golden vectors for this path and cannot be built here (needs OpenCV), so these fixtures
pin the ORACLE (the CPU restatement; parity unpinned at the OpenCV boundary) -- they make sure neither the
oracle nor the HIP path drifts between rounds.  Inputs: (a) small frames stored verbatim in the .npz,
(b) frames re-generated from the seeded synthetic generator, identified by a sha256 of the pixels.
"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from orb_slam2_detailed_comments_amd import synth

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run(img, nf, fp):
    o = oracle.OracleExtractor(nf, 1.2, 8, 20, 7, fp_mode=fp)
    n, k, d = o.extract(img)
    per_level = [int((k["octave"] == l).sum()) for l in range(8)] if n > 0 else [0] * 8
    return n, k, d, per_level


index = {"stored": [], "generated": []}
# (a) stored small inputs
for name, (w, h, nf, sid) in {"s160x120": (160, 120, 300, 11), "s200x96": (200, 96, 200, 12), "s97x131": (97, 131, 150, 13)}.items():
    img = synth.stream(w, h, 1, stream_id=sid)[0]
    rec = {"name": name, "nfeatures": nf, "file": name + ".npz"}
    arrs = {"image": img}
    for fp, tag in ((0, "fma"), (1, "strict")):
        n, k, d, pl = run(img, nf, fp)
        arrs["kps_" + tag] = k.view(np.uint8).reshape(n, 28)
        arrs["desc_" + tag] = d
        rec["n_" + tag] = n
        rec["per_level_" + tag] = pl
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    index["stored"].append(rec)
# (b) generated inputs: hashes + first rows
for (w, h, nf, sid, t) in [(640, 480, 1000, 0, 0), (640, 480, 1000, 0, 5), (640, 480, 1000, 1, 0), (752, 480, 1200, 2, 0),
                           (1241, 376, 2000, 3, 0)]:
    img = synth.Scene(w, h, sid).frame(t)
    rec = {"width": w, "height": h, "nfeatures": nf, "stream_id": sid, "t": t, "image_sha256": sha(img)}
    for fp, tag in ((0, "fma"), (1, "strict")):
        n, k, d, pl = run(img, nf, fp)
        rec["n_" + tag] = n
        rec["per_level_" + tag] = pl
        rec["kps_sha256_" + tag] = sha(k)
        rec["desc_sha256_" + tag] = sha(d)
        rec["first_desc_" + tag] = d[:4].tolist()
        rec["first_kps_" + tag] = [[float(v) for v in (r["x"], r["y"], r["size"], r["angle"], r["response"])] for r in k[:4]]
    index["generated"].append(rec)
index["glibc"] = os.confstr("CS_GNU_LIBC_VERSION")
json.dump(index, open(os.path.join(OUT, "golden_index.json"), "w"), indent=1)
print(json.dumps({k: len(v) if isinstance(v, list) else v for k, v in index.items()}))
