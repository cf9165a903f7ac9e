#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match on MI355X (BASELINE.json metric), one JSON line on rank 0.

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
  K0..K6 extraction of `--batch` frames (per GPU) + K7 brute-force Hamming match of frame t vs t-1
  (+ for N > 1 one RCCL gather of the per-frame keypoint records).
By default the steps run one after the other on one pipeline (handle + HIP stream), so that the per-kernel durations behind
`roofline` (HIP events on the launch stream) are the kernels' own and agree with rocprofv3 of the same command.
`--streams N` runs consecutive steps on N pipelines round-robin (the tail of step i overlaps the head of step i+1; every step
still does all of its work, frame 0 of a step is matched against the last frame of the step before it through an event):
+8 % frames/s at N = 3, but a kernel's duration then includes the time it shares the chip with its neighbours' kernels.
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
        "k_fast_rows": Pt,
        "k_quadtree": 0,
        "k_orient": n_kp * 749,
        "k_blur": 0,                                   # fused into k_describe (the blurred image is never written)
        "k_describe": 2 * Pt + n_kp * 512 + n_kp * 60,  # blur read+write + descriptor taps + output
        "k_match": 2 * n_kp * 32 + n_kp * 12,
    }


def measured_traffic(kernel, B, W, H, NF):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 calibration in tools/fetch_calib.hip).
    None when the committed counters were taken on another workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if (t["batch"], t["width"], t["height"], t["nfeatures"]) != (B, W, H, NF):
            return None
        return int(t["kernels"][kernel]["hbm_bytes_per_launch"])
    except Exception:
        return None


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
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--streams", type=int, default=1,
                    help="extract+match pipelines per GPU (handle + HIP stream each), used round-robin over the steps: the "
                         "latency-bound tail of one step (quadtree, small pyramid levels) overlaps the issue-bound kernels of "
                         "the next (+8 %% frames/s at 3, DESIGN.md section 6).  The default 1 keeps per-kernel durations un-overlapped")
    ap.add_argument("--gather", default="gather", choices=("gather", "all_gather"),
                    help="collective for the per-frame keypoint records: to rank 0 (default) or to every rank")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", action="store_true", help="also print a per-kernel table to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ORB front-end has no CPU fallback")
    if os.environ.get("ORBX_BENCH_SHARE_GPU0"):   # rehearsal of the N > 1 code path on a one-GPU box (gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    B, W, H, NF = args.batch, args.width, args.height, args.nfeatures
    NS = max(1, args.streams)
    frames = synth.stream(W, H, B, stream_id=100 + rank)
    d_imgs = torch.from_numpy(frames).to(dev)
    L = _capi.lib()
    # NS independent pipelines (handle + HIP stream + result buffers), used round-robin: while one batch is in its
    # latency-bound tail (quadtree, small pyramid levels) the next batch's streaming kernels fill the chip.
    exs = [ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B, device=local_rank) for _ in range(NS)]
    ex = exs[0]
    cap = ex.max_keypoints(W, H)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    for e, st in zip(exs, streams):
        e.set_stream(st.cuda_stream)
    # slot 0 of the result buffers carries the last frame of the previous step (match t vs t-1)
    bufs = []
    for _ in range(NS):
        bufs.append(dict(
            kps=torch.zeros((B + 1, cap * 28), dtype=torch.uint8, device=dev),
            desc=torch.zeros((B + 1, cap * 32), dtype=torch.uint8, device=dev),
            counts=torch.zeros(B + 1, dtype=torch.int32, device=dev),
            status=torch.zeros(B, dtype=torch.int32, device=dev),
            midx=torch.zeros((B, cap), dtype=torch.int32, device=dev),
            mbest=torch.zeros((B, cap), dtype=torch.int32, device=dev),
            msecond=torch.zeros((B, cap), dtype=torch.int32, device=dev),
            done=torch.cuda.Event()))
    gatherer = sharding.RecordGatherer(B, cap, dev, mode=args.gather) if world > 1 else None
    torch.cuda.synchronize(dev)   # the zero fills above ran on torch's default stream; the pipelines' streams are not ordered with it
    state = {"i": 0}

    def step():
        i = state["i"]; state["i"] = i + 1
        k = i % NS
        e, st, b = exs[k], streams[k], bufs[k]
        prev = bufs[(i - 1) % NS]
        with torch.cuda.stream(st):
            e.extract_batch_device(d_imgs, B, W, H, W, W * H, b["kps"][1:], b["desc"][1:], b["counts"][1:], b["status"], cap)
            if i > 0:
                st.wait_event(prev["done"])        # the previous batch's last frame is the train set of pair 0
            b["desc"][0].copy_(prev["desc"][B]); b["counts"][0:1].copy_(prev["counts"][B:B + 1])
            _capi.check(L.orbx_match_bruteforce_device(
                e.handle, B, _capi.ptr(b["desc"][1:]), _capi.ptr(b["counts"][1:]), cap * 32, _capi.ptr(b["desc"]),
                _capi.ptr(b["counts"]), cap * 32, _capi.ptr(b["midx"]), _capi.ptr(b["mbest"]), _capi.ptr(b["msecond"]), cap))
            if gatherer is not None:
                gatherer.gather(b["counts"][1:], b["kps"][1:], b["desc"][1:], async_op=True)
            b["done"].record(st)

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize(dev)
    status = torch.stack([b["status"] for b in bufs]).cpu().numpy()
    if status.any():
        raise SystemExit(f"extraction reported status {status.tolist()}")
    counts = bufs[0]["counts"][1:].cpu().numpy()
    n_kp = float(counts.mean())

    # calibration pass: which kernel dominates?  (all kernels timed with HIP events on the launch stream)
    for e in exs:
        e.profile_enable(0x1ff)
    for _ in range(2 * NS):
        step()
    prof = {}
    for e in exs:
        for kname, (ms, n) in e.profile_read(reset=True).items():
            a = prof.get(kname, (0.0, 0))
            prof[kname] = (a[0] + ms, a[1] + n)
    prof = {kname: (v[0] / NS, v[1] // NS) for kname, v in prof.items()}   # per 2 steps, like the single-pipeline table
    dominant = max((k for k in prof if k != "misc"), key=lambda k: prof[k][0])
    kid = _capi.K_NAMES.index(dominant)
    for e in exs:
        e.profile_enable(1 << kid)

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
    dom_ms, dom_launches = 0.0, 0
    for e in exs:
        ms_, n_ = e.profile_read(reset=True)[dominant]
        dom_ms += ms_; dom_launches += n_
        e.profile_enable(0)

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
                                   f"extract+match(t vs t-1), batch {B} frames/GPU resident in HBM, {NS} pipelines/GPU",
                       "frames_per_gpu_per_step": B, "mean_keypoints": round(n_kp, 1),
                       "parallelism": f"frames sharded x{world}, RCCL {args.gather} of keypoint records per step" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": measured_traffic(dominant, B, W, H, NF),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "avg_launch_us": round(avg_launch_s * 1e6, 2),
                         "end_to_end_GBs": round(total_ab * fps / world / 1e9, 2),
                         "end_to_end_frac": round(total_ab * fps / world / 1e9 / HBM_PEAK_GBS, 5)},
            "kernel_ms_per_step": {k: round(v[0] / 2, 4) for k, v in prof.items()},
            "kernel_ms_overlapped": NS > 1,   # with several pipelines the per-kernel durations include time shared with other kernels
        }
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only (the other ranks would idle through it)
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
