#!/usr/bin/env python3
"""Stereo frame pairs per second on one MI355X, everything resident on the device: extraction of both eyes (two handles,
as src/Frame.cc:158-168 uses two extractor objects) + Frame::ComputeStereoMatches with its median cut.
BASELINE.json configs 3 (KITTI 1241x376 / 2000 features) and 4 (EuRoC 752x480 / 1200 features)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_detailed_comments_amd import ORBextractor, synth, _capi

dev = torch.device("cuda", 0)
L = _capi.lib()
for name, w, h, nf, B, mb, mbf in (("KITTI 1241x376/2000", 1241, 376, 2000, 64, 0.537, 386.1448), ("EuRoC 752x480/1200", 752, 480, 1200, 128, 0.11, 47.9)):
    base = [synth.stereo_pair(w, h, stream_id=70 + i) for i in range(8)]
    imgs = {"L": np.stack([base[i % 8][0] for i in range(B)]), "R": np.stack([base[i % 8][1] for i in range(B)])}
    ex = {k: ORBextractor(nf, max_batch=B) for k in "LR"}
    cap = ex["L"].max_keypoints(w, h)
    b = {k: dict(img=torch.from_numpy(imgs[k]).to(dev), kps=torch.zeros((B, cap * 28), dtype=torch.uint8, device=dev),
                 desc=torch.zeros((B, cap * 32), dtype=torch.uint8, device=dev), cnt=torch.zeros(B, dtype=torch.int32, device=dev),
                 st=torch.zeros(B, dtype=torch.int32, device=dev)) for k in "LR"}
    ur = torch.zeros((B, cap), dtype=torch.float32, device=dev); dep = torch.zeros_like(ur); nm = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    for k in "LR": ex[k].set_stream(stream.cuda_stream)
    def step():
        for k in "LR":
            ex[k].extract_batch_device(b[k]["img"], B, w, h, w, w * h, b[k]["kps"], b[k]["desc"], b[k]["cnt"], b[k]["st"], cap)
        _capi.check(L.orbx_stereo_match_batch_device(ex["L"].handle, ex["R"].handle, B, _capi.ptr(b["L"]["kps"]), _capi.ptr(b["L"]["desc"]),
                                                     _capi.ptr(b["L"]["cnt"]), _capi.ptr(b["R"]["kps"]), _capi.ptr(b["R"]["desc"]),
                                                     _capi.ptr(b["R"]["cnt"]), cap, mb, mbf, _capi.ptr(ur), _capi.ptr(dep), _capi.ptr(nm)))
    for _ in range(3): step()
    torch.cuda.synchronize(dev)
    ex["L"].profile_enable(1 << 7)
    t0 = time.perf_counter(); n = 10
    for _ in range(n): step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / n
    ms = ex["L"].profile_read(reset=True)["k_match"][0] / n
    print(f"{name}: {B / dt:8.0f} stereo pairs/s ({dt * 1e3:.2f} ms per {B} pairs; stereo match + median cut {ms * 1e3:.0f} us of it; "
          f"mean keypoints L {float(b['L']['cnt'].float().mean()):.0f}, mean stereo matches {float(nm.float().mean()):.0f})")
