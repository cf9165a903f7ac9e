#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match on MI355X (BASELINE.json metric), one JSON line on rank 0.

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
  K0..K6 extraction of `--batch` frames (per GPU) + K7 brute-force Hamming match of frame t vs t-1
  (+ for N > 1 one RCCL all-gather of the per-frame keypoint records).
Weak scaling: every rank owns its own batch; no collective on the data path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from orb_slam2_detailed_comments_amd import ORBextractor, synth, _capi
from orb_slam2_detailed_comments_amd import sharding

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def level_pixels(ex, w, h):
    """padded pixels per level (SURVEY 8d 'P'), from the product's own geometry"""
    inv = ex.GetInverseScaleFactors()
    out = []
    for l in range(ex.nlevels):
        sw = int(np.rint(np.float32(w) * inv[l])); sh = int(np.rint(np.float32(h) * inv[l]))
        out.append((sw + 38) * (sh + 38))
    return out


def algorithmic_bytes(ex, w, h, n_kp):
    """per-frame algorithmic bytes of each kernel (SURVEY.md section 8d model, split per kernel)"""
    P = level_pixels(ex, w, h)
    Pt = sum(P)
    return {
        "k_pyr_l0": w * h + P[0],
        "k_pyr_resize": (Pt - P[-1]) + (Pt - P[0]),
        "k_fast_cells": Pt,
        "k_quadtree": 0,
        "k_orient": n_kp * 749,
        "k_blur": 0,                                   # fused into k_describe (the blurred image is never written)
        "k_describe": 2 * Pt + n_kp * 512 + n_kp * 60,  # blur read+write + descriptor taps + output
        "k_match": 2 * n_kp * 32 + n_kp * 12,
    }


def cpu_baseline(frames, nfeatures, budget_s=20.0):
    """single-thread CPU oracle (port of the reference path) on a bounded sample of the same frames"""
    import oracle
    orc = oracle.OracleExtractor(nfeatures, 1.2, 8, 20, 7)
    t0 = time.perf_counter()
    n = 0
    prev = None
    for f in frames:
        _, k, d = orc.extract(f)
        if prev is not None:
            oracle.match_bruteforce(d, prev)
        prev = d
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return n / dt, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", action="store_true", help="also print a per-kernel table to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ORB front-end has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, W, H, NF = args.batch, args.width, args.height, args.nfeatures
    frames = synth.stream(W, H, B, stream_id=100 + rank)
    d_imgs = torch.from_numpy(frames).to(dev)
    ex = ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B, device=local_rank)
    L = _capi.lib()
    cap = ex.max_keypoints(W, H)
    # slot 0 of the result buffers carries the last frame of the previous step (match t vs t-1)
    d_kps = torch.zeros((B + 1, cap * 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B + 1, cap * 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_status = torch.zeros(B, dtype=torch.int32, device=dev)
    d_midx = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_mbest = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_msecond = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    ex.set_stream(stream.cuda_stream)
    gatherer = sharding.RecordGatherer(B, cap, dev) if world > 1 else None

    def step():
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps[1:], d_desc[1:], d_counts[1:], d_status, cap)
        _capi.check(L.orbx_match_bruteforce_device(
            ex.handle, B, _capi.ptr(d_desc[1:]), _capi.ptr(d_counts[1:]), cap * 32, _capi.ptr(d_desc),
            _capi.ptr(d_counts), cap * 32, _capi.ptr(d_midx), _capi.ptr(d_mbest), _capi.ptr(d_msecond), cap))
        if gatherer is not None:
            gatherer.gather(d_counts[1:], d_kps[1:], d_desc[1:], async_op=True)
        # carry the last frame into slot 0 for the next step
        d_desc[0].copy_(d_desc[B]); d_counts[0:1].copy_(d_counts[B:B + 1])

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize(dev)
    status = d_status.cpu().numpy()
    if status.any():
        raise SystemExit(f"extraction reported status {status.tolist()}")
    counts = d_counts[1:].cpu().numpy()
    n_kp = float(counts.mean())

    # calibration pass: which kernel dominates?  (all kernels timed with HIP events on the launch stream)
    ex.profile_enable(0x1ff)
    for _ in range(2):
        step()
    prof = ex.profile_read(reset=True)
    dominant = max((k for k in prof if k != "misc"), key=lambda k: prof[k][0])
    kid = _capi.K_NAMES.index(dominant)
    ex.profile_enable(1 << kid)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gatherer is not None:
        gatherer.wait_all()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    dom_ms, dom_launches = ex.profile_read(reset=True)[dominant]
    ex.profile_enable(0)

    if rank == 0:
        fps = world * B * args.steps / dt
        ab = algorithmic_bytes(ex, W, H, n_kp)
        launches_per_step = dom_launches / max(args.steps, 1)
        bytes_per_launch = ab[dominant] * B / max(launches_per_step, 1e-9)
        avg_launch_s = dom_ms * 1e-3 / max(dom_launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        total_ab = sum(ab.values())
        out = {
            "metric": "frames/sec ORB extract+match (1000 kp, 640x480)",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"synthetic {W}x{H} mono stream, nFeatures={NF}, 8 levels, scale 1.2, FAST 20/7, "
                                   f"extract+match(t vs t-1), batch {B} frames/GPU resident in HBM",
                       "frames_per_gpu_per_step": B, "mean_keypoints": round(n_kp, 1),
                       "parallelism": f"frames sharded x{world}, RCCL all-gather of keypoint records" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "avg_launch_us": round(avg_launch_s * 1e6, 2),
                         "end_to_end_GBs": round(total_ab * fps / world / 1e9, 2),
                         "end_to_end_frac": round(total_ab * fps / world / 1e9 / HBM_PEAK_GBS, 5)},
            "kernel_ms_per_step": {k: round(v[0] / 2, 4) for k, v in prof.items()},
        }
        if not args.no_cpu_baseline:
            cfps, nsample = cpu_baseline(frames, NF)
            out["cpu_baseline"] = {"value": round(cfps, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": f"{nsample} frames of the same synthetic stream, single-thread CPU oracle "
                                             f"(extract + brute-force match), host has {os.cpu_count()} cores"}
        if args.stages:
            for k, v in prof.items():
                print(f"  {k:14s} {v[0] / 2:9.4f} ms/step  ({v[1] // 2} launches/step)", file=sys.stderr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
