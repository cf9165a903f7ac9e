#!/usr/bin/env python3
"""Runs a few batched extractions + matches so that rocprofv3 --pmc can attribute counters per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_detailed_comments_amd import ORBextractor, synth, _capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = synth.stream(640, 480, B, stream_id=100)
dev = torch.device('cuda', 0)
d_imgs = torch.from_numpy(frames).to(dev)
ex = ORBextractor(1000, max_batch=B)
cap = ex.max_keypoints(640, 480)
d_kps = torch.zeros((B + 1, cap * 28), dtype=torch.uint8, device=dev); d_desc = torch.zeros((B + 1, cap * 32), dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(B + 1, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
mi = torch.zeros((B, cap), dtype=torch.int32, device=dev); mb = torch.zeros_like(mi); ms = torch.zeros_like(mi)
torch.cuda.synchronize()
L = _capi.lib()
for i in range(4):
    ex.extract_batch_device(d_imgs, B, 640, 480, 640, 640 * 480, d_kps[1:], d_desc[1:], d_cnt[1:], d_st, cap)
    _capi.check(L.orbx_match_bruteforce_device(ex.handle, B, _capi.ptr(d_desc[1:]), _capi.ptr(d_cnt[1:]), cap * 32, _capi.ptr(d_desc),
                                               _capi.ptr(d_cnt), cap * 32, _capi.ptr(mi), _capi.ptr(mb), _capi.ptr(ms), cap))
ex.synchronize()
print("done", d_cnt[1:5].cpu().tolist())
