#!/usr/bin/env python3
"""Throughput of the batched DBoW2 transform (k_bow_transform) on a synthetic ORBvoc-sized tree (k=10, L=6: 1.1 M nodes,
35 MB of node descriptors) over the descriptors of a 256-frame extraction."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_detailed_comments_amd import ORBextractor, ORBVocabulary, synth, _capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k, L = 10, 6
n = sum(k ** l for l in range(L + 1))
rng = np.random.default_rng(0)
begin = np.zeros(n + 1, np.int32)
first_leaf = sum(k ** l for l in range(L))
begin[1:first_leaf + 1] = k * np.arange(1, first_leaf + 1); begin[first_leaf + 1:] = begin[first_leaf]
child_ids = np.arange(1, n, dtype=np.uint32)          # BFS numbering: children of node i are i*k+1 .. i*k+k
desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
word = np.zeros(n, np.uint32); word[first_leaf:] = np.arange(n - first_leaf)
dev = torch.device("cuda", 0)
frames = synth.stream(640, 480, B, stream_id=100)
ex = ORBextractor(1000, max_batch=B)
V = ORBVocabulary(ex, n_nodes=n, k=k, L=L, child_begin=begin, child_ids=child_ids, desc=desc, weight=rng.uniform(0.1, 9, n), word_id=word)
cap = ex.max_keypoints(640, 480)
d_imgs = torch.from_numpy(frames).to(dev)
kps = torch.zeros((B, cap * 28), dtype=torch.uint8, device=dev); dsc = torch.zeros((B, cap * 32), dtype=torch.uint8, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
ex.extract_batch_device(d_imgs, B, 640, 480, 640, 640 * 480, kps, dsc, cnt, st, cap)
leaf = torch.zeros((B, cap), dtype=torch.int32, device=dev); nid = torch.zeros_like(leaf)
torch.cuda.synchronize()
L_ = _capi.lib()
def run():
    _capi.check(L_.orbx_bow_transform_device(ex.handle, V._h, B, _capi.ptr(dsc), _capi.ptr(cnt), cap * 32, cap, 4, _capi.ptr(leaf),
                                             _capi.ptr(nid), cap))
for _ in range(3): run()
ex.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
ex.synchronize()
dt = (time.perf_counter() - t0) / 20
nd = int(cnt.sum().item())
print(f"k_bow_transform: {dt * 1e6:.1f} us per {B} frames ({nd} descriptors, {n} nodes): {B / dt:.0f} frames/s, "
      f"{nd * k * L / dt / 1e9:.2f} G node distances/s")
# spot check against the host-buffer entry point
wid, w, nn = V.transform_features(dsc[0].cpu().numpy().reshape(-1, 32)[:int(cnt[0])], 4)
assert np.array_equal(nn, nid[0, :int(cnt[0])].cpu().numpy().astype(np.uint32))
print("device batch == host entry point for frame 0")
