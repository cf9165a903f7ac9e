#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match on MI355X (BASELINE.json metric), one JSON line on rank 0.

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM.
  mono configs  : K0..K6 extraction of `--batch` frames (per GPU) + K7 brute-force Hamming match of frame t vs t-1
  stereo configs: K0..K6 extraction of both eyes of `--batch` stereo pairs (one extractor batch of 2 x batch images, left
                  images first; `--stereo-batch split`: two extractor handles, as src/Frame.cc:158-168 uses two extractor
                  objects) + Frame::ComputeStereoMatches with its median cut, all on the device
  (+ for N > 1 one RCCL gather of the per-frame keypoint records).
`--config` selects the BASELINE.json configuration (default: the headline one, config 2):
  tum           synthetic 640x480 mono stream, nFeatures 1000      (configs 1/2, Examples/Monocular/TUM1.yaml:30-43)
  kitti_stereo  synthetic 1241x376 stereo pairs, nFeatures 2000    (config 3, Examples/Stereo/KITTI00-02.yaml:18-51)
  euroc_stereo  synthetic 752x480 stereo pairs, nFeatures 1200     (config 4, Examples/Stereo/EuRoC.yaml:18-106)
  hd1080        synthetic 1920x1080 mono stream, nFeatures 4000    (config 5)
By default the steps run one after the other on one pipeline (handle + HIP stream), so that the per-kernel durations behind
`roofline` (HIP events on the launch stream) are the kernels' own and agree with rocprofv3 of the same command.
`--streams N` (mono) runs consecutive steps on N pipelines round-robin; a kernel's duration then includes the time it shares the
chip with its neighbours' kernels.  Weak scaling: every rank owns its own batch; no collective on the data path.
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

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SCLK_PEAK_HZ = 2.4e9    # same guide: max clock 2400 MHz
N_SIMD = 1024           # 256 CUs x 4 SIMD-32
PROFILE_TAG = "r03"     # profiles/<tag>_traffic.json, profiles/<tag>_sq.json

CONFIGS = {   # name: (width, height, nfeatures, frames or pairs per GPU per step, stereo, mb, mbf)
    # batch sizes: large enough that launch boundaries and the drain of each kernel stop showing (tools/batch_sweep.sh: tum
    # 256 / 512 / 1024 frames = 249 / 254 / 261 k frames/s; hd1080 32 / 64 / 128 = 38.1 / 41.0 / 43.7 k; KITTI 64 / 128 / 256
    # pairs = 69.7 / 73.5 / 75.2 k; EuRoC 128 / 256 = 104.0 / 107.9 k); 1024 VGA frames are 1.5 GB of the GPU's 288 GB
    "tum": (640, 480, 1000, 1024, False, 0.0, 0.0),
    "kitti_stereo": (1241, 376, 2000, 256, True, 0.537, 386.1448),     # bf 386.1448, fx 718.856 (KITTI00-02.yaml)
    "euroc_stereo": (752, 480, 1200, 256, True, 0.11, 47.90639384423901),   # EuRoC.yaml: Camera.bf
    "hd1080": (1920, 1080, 4000, 128, False, 0.0, 0.0),
}


def level_pixels(ex, w, h):
    """padded pixels per level (SURVEY 8d 'P'), from the product's own geometry"""
    inv = ex.GetInverseScaleFactors()
    out = []
    for l in range(ex.nlevels):
        sw = int(np.rint(np.float32(w) * inv[l])); sh = int(np.rint(np.float32(h) * inv[l]))
        out.append((sw + 38) * (sh + 38))
    return out


def algorithmic_bytes(ex, w, h, n_kp, stereo):
    """Per-image algorithmic bytes of each kernel.  "model" = SURVEY.md section 8(d), which was written for an
    unfused pipeline (GaussianBlur reads and writes the whole pyramid: 2P); "fused" = what the kernels here must move
    (the Gaussian is evaluated inside k_describe on a 43x43 patch per keypoint, the blurred image is never written)."""
    P = level_pixels(ex, w, h)
    Pt = sum(P)
    model = {
        "k_pyr_l0": w * h + P[0],
        "k_pyr_resize": (Pt - P[-1]) + (Pt - P[0]),
        "k_fast_rows": Pt,
        "k_quadtree": 0,
        "k_orient": n_kp * 749,
        "k_blur": 0,
        "k_describe": 2 * Pt + n_kp * 512 + n_kp * 60,   # 8(d): blur read + write, descriptor taps, output
        "k_match": 2 * n_kp * 32 + n_kp * 12,
    }
    fused = dict(model)
    fused["k_orient"] = 0                                   # the orientation disc lies inside the descriptor's patch
    fused["k_describe"] = n_kp * (43 * 43) + n_kp * 60      # one un-blurred 43x43 patch per keypoint + keypoint / descriptor out
    if stereo:   # ComputeStereoMatches: both descriptor sets + keypoints once, two 11x21 SAD windows per matched keypoint
        model["k_match"] = fused["k_match"] = 2 * n_kp * (32 + 28) // 2 + n_kp * (11 * 11 + 11 * 21) // 2 + n_kp * 8 // 2
    return model, fused


def kernels_hash():
    """identity of the device code the committed PMC counters were taken on"""
    from orb_slam2_detailed_comments_amd import build
    return build.kernels_hash()


def committed_counters(kind, cfg_name, B, W, H, NF):
    """profiles/<tag>_{traffic,sq}.json (separate rocprofv3 --pmc passes, tools/make_traffic_json.py / tools/make_sq_json.py).
    Returns None unless the file was taken on THIS workload with THIS device code (kernel source hash): a stale file reads
    as 'not measured', never as a number."""
    try:
        name = f"{PROFILE_TAG}_{kind}.json" if cfg_name == "tum" else f"{PROFILE_TAG}_{kind}_{cfg_name}.json"
        t = json.load(open(os.path.join(ROOT, "profiles", name)))
        if t.get("kernels_sha256_16") != kernels_hash():
            return None
        if (t["batch"], t["width"], t["height"], t["nfeatures"]) != (B, W, H, NF) or t.get("config", "tum") != cfg_name:
            return None
        return t
    except Exception:
        return None


def cpu_baseline(frames, right, nfeatures, stereo, mb, mbf, budget_s=20.0):
    """single-thread CPU oracle (port of the reference path) on a bounded sample of the same frames, pinned to one core"""
    import oracle
    # the reference's own build flags (/root/reference/CMakeLists.txt:10-11: -O3 -march=native), compiled HERE for this host's
    # cores; byte-identical to the -O2 parity build (tests/test_oracle_native_flags.py).  Falls back to the parity build.
    flags = "-O2 -ffp-contract=off"
    try:
        os.environ["ORB_ORACLE_LIB"] = oracle.orb_oracle.build_native()
        flags = oracle.orb_oracle.NATIVE_FLAGS
    except Exception:
        os.environ.pop("ORB_ORACLE_LIB", None)
    cores = None
    try:
        cores = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(cores)[-1]})   # taskset -c <core>
        pinned = True
    except Exception:
        pinned = False
    orc = oracle.OracleExtractor(nfeatures, 1.2, 8, 20, 7)
    orcR = oracle.OracleExtractor(nfeatures, 1.2, 8, 20, 7) if stereo else None
    tabs = orc.tables()
    t0 = time.perf_counter()
    n = 0
    prev = None
    for i, f in enumerate(frames):
        _, k, d = orc.extract(f)
        if stereo:
            _, kr, dr = orcR.extract(right[i])
            pl = [orc.level_image(l) for l in range(8)]; pr = [orcR.level_image(l) for l in range(8)]
            oracle.stereo_matches(k, d, kr, dr, tabs["scale"], tabs["inv_scale"], pl, pr, mb, mbf)
        elif prev is not None:
            oracle.match_bruteforce(d, prev)
        prev = d
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    if pinned:
        os.sched_setaffinity(0, cores)   # what follows (host_io) is not a one-core measurement
    return n / dt, n, pinned, flags


def host_io_rate(L, frames, W, H, NF, cap, local_rank):
    """secondary figure: the same extraction fed from HOST memory through orbx_extract_batch (page-locked buffers in and out, the
    step's frames in chunks of 64: upload of chunk c+1 / download of chunk c-1 on one copy stream while chunk c computes; keypoints
    and descriptors are written by the GPU straight into the page-locked result buffers), PCIe inclusive.  Never `value`."""
    nh, hchunk = min(len(frames), 1024), 64
    exh = ORBextractor(NF, 1.2, 8, 20, 7, max_batch=hchunk, device=local_rank)
    keep = [_capi.PinnedArray((nh, H, W)), _capi.PinnedArray((nh, cap), _capi.KP_DTYPE), _capi.PinnedArray((nh, cap, 32)),
            _capi.PinnedArray((nh,), np.int32)]
    himg, hk, hd, hc = (k.array for k in keep)
    himg[...] = frames[:nh]
    hcall = lambda: _capi.check(L.orbx_extract_batch(exh.handle, nh, _capi.ptr(himg), W, H, W, W * H, _capi.ptr(hk), _capi.ptr(hd), _capi.ptr(hc), cap))
    hcall(); hcall()
    th = time.perf_counter()
    for _ in range(6):
        hcall()
    th = (time.perf_counter() - th) / 6
    del exh
    return {"value": round(nh / th, 1), "unit": "frames/s", "frames_per_call": nh, "chunk": hchunk, "ms_per_call": round(th * 1e3, 3),
            "memory": "page-locked host buffers in and out (orbx_host_alloc), extraction only",
            "bytes_per_frame": int(W * H + cap * 60 + 4),
            "link_GBs": round(nh / th * (W * H + cap * 60 + 4) / 1e9, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="tum", choices=sorted(CONFIGS), help="BASELINE.json configuration (default: the headline one)")
    ap.add_argument("--batch", type=int, default=0, help="frames (stereo: pairs) per GPU per step; 0 = the config's default")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--nfeatures", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1,
                    help="extract+match pipelines per GPU (handle + HIP stream each), used round-robin over the steps (mono "
                         "configs).  The default 1 keeps per-kernel durations un-overlapped")
    ap.add_argument("--fork-level", type=int, default=0,
                    help="ORBX_FORK_LEVEL for the extractor handles: levels >= N are resized (and their FAST groups run) on the handle's "
                         "side stream next to the FAST kernel of the large levels (+3..5 %% frames/s at 3; DESIGN.md section 6).  0 = off "
                         "(default): every kernel then runs alone and its duration is its own")
    ap.add_argument("--stereo-streams", type=int, default=1, choices=(1, 2),
                    help="stereo configs: 1 = both eyes' extractions on one stream (kernels run alone, durations are their own); "
                         "2 = one stream per eye, as the reference's two extraction threads (overlapped durations)")
    ap.add_argument("--stereo-batch", default="merged", choices=("merged", "split"),
                    help="stereo configs: merged = both eyes of a step through ONE extractor batch of 2 x batch images (left images "
                         "first; orbx_stereo_match_batch_device with hl == hr); split = one handle and one batch per eye, as the "
                         "reference's two ORBextractor objects")
    ap.add_argument("--gather", default="gather", choices=("gather", "all_gather"),
                    help="collective for the per-frame keypoint records: to rank 0 (default) or to every rank")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-io", action="store_true", help="skip the PCIe-inclusive secondary figure (host_io)")
    ap.add_argument("--stages", action="store_true", help="also print a per-kernel table to stderr")
    args = ap.parse_args()

    if args.fork_level > 0:
        os.environ["ORBX_FORK_LEVEL"] = str(args.fork_level)   # read by the library when a handle is configured
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

    cW, cH, cNF, cB, stereo, mb, mbf = CONFIGS[args.config]
    B, W, H, NF = args.batch or cB, args.width or cW, args.height or cH, args.nfeatures or cNF
    NS = 1 if stereo else max(1, args.streams)
    L = _capi.lib()
    right = None
    if stereo:
        # 8 rendered synthetic pairs (right = left scene re-rendered with per-rectangle disparity; 0.3 s each) fill the batch as B
        # DISTINCT images: pair i is base pair i % 8 rolled by (5 i, 11 i) pixels, both eyes alike (the epipolar geometry holds, the
        # wrap-around seam is one more edge).  A batch of 8 repeated pairs would sit in the 256 MB memory-side cache and make the
        # level-0 source reads look cheaper than a real stream's.
        base = [synth.stereo_pair(W, H, stream_id=70 + 8 * rank + i) for i in range(8)]
        roll = lambda im, i: im if i < 8 else np.roll(im, (5 * i % H, 11 * i % W), axis=(0, 1))
        frames = np.stack([roll(base[i % 8][0], i) for i in range(B)]); right = np.stack([roll(base[i % 8][1], i) for i in range(B)])
    else:
        frames = synth.stream(W, H, B, stream_id=100 + rank)
    d_imgs = torch.from_numpy(frames).to(dev)
    d_right = torch.from_numpy(right).to(dev) if stereo else None
    # NS independent pipelines (handle + HIP stream + result buffers), used round-robin: while one batch is in its
    # latency-bound tail (quadtree, small pyramid levels) the next batch's streaming kernels fill the chip.
    merged = stereo and args.stereo_batch == "merged" and args.stereo_streams == 1
    exs = [ORBextractor(NF, 1.2, 8, 20, 7, max_batch=2 * B if merged else B, device=local_rank) for _ in range(NS)]
    exR = ORBextractor(NF, 1.2, 8, 20, 7, max_batch=B, device=local_rank) if stereo and not merged else None
    d_both = torch.cat([d_imgs, d_right]) if merged else None
    ex = exs[0]
    cap = ex.max_keypoints(W, H)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    for e, st in zip(exs, streams):
        e.set_stream(st.cuda_stream)
    stream_r = None
    if stereo and not merged:   # the reference extracts the two eyes on two threads (src/Frame.cc:158-168): two handles; --stereo-streams 2 gives each its own stream
        stream_r = torch.cuda.Stream(device=dev) if args.stereo_streams == 2 else streams[0]
        exR.set_stream(stream_r.cuda_stream)
    # slot 0 of the result buffers carries the last frame of the previous step (match t vs t-1)
    bufs = []
    for _ in range(NS):
        b = dict(
            kps=torch.zeros((B + 1, cap * 28), dtype=torch.uint8, device=dev),
            desc=torch.zeros((B + 1, cap * 32), dtype=torch.uint8, device=dev),
            counts=torch.zeros(B + 1, dtype=torch.int32, device=dev),
            status=torch.zeros(B, dtype=torch.int32, device=dev),
            midx=torch.zeros((B, cap), dtype=torch.int32, device=dev),
            mbest=torch.zeros((B, cap), dtype=torch.int32, device=dev),
            msecond=torch.zeros((B, cap), dtype=torch.int32, device=dev))
        if stereo:
            b.update(kpsR=torch.zeros((B, cap * 28), dtype=torch.uint8, device=dev),
                     descR=torch.zeros((B, cap * 32), dtype=torch.uint8, device=dev),
                     countsR=torch.zeros(B, dtype=torch.int32, device=dev), statusR=torch.zeros(B, dtype=torch.int32, device=dev),
                     uright=torch.zeros((B, cap), dtype=torch.float32, device=dev), depth=torch.zeros((B, cap), dtype=torch.float32, device=dev),
                     nmatch=torch.zeros(B, dtype=torch.int32, device=dev))
        if merged:   # one batch of 2B images: left eyes in [0, B), right eyes in [B, 2B)
            b.update(kpsM=torch.zeros((2 * B, cap * 28), dtype=torch.uint8, device=dev),
                     descM=torch.zeros((2 * B, cap * 32), dtype=torch.uint8, device=dev),
                     countsM=torch.zeros(2 * B, dtype=torch.int32, device=dev), statusM=torch.zeros(2 * B, dtype=torch.int32, device=dev))
        bufs.append(b)
    # slot 0 of a pipeline's buffers = descriptors + count of the last frame of the PREVIOUS step (the train set of pair 0):
    # written by that step, after its own match, straight into the buffers of the pipeline that runs the next step
    carry_ready = torch.cuda.Event()
    gatherer = sharding.RecordGatherer(B, cap, dev, mode=args.gather) if world > 1 else None
    torch.cuda.synchronize(dev)   # the zero fills above ran on torch's default stream; the pipelines' streams are not ordered with it
    state = {"i": 0}

    def step():
        i = state["i"]; state["i"] = i + 1
        k = i % NS
        e, st, b = exs[k], streams[k], bufs[k]
        with torch.cuda.stream(st):
            if merged:       # stereo, both eyes through one extractor batch: left images in [0, B), right images in [B, 2B)
                e.extract_batch_device(d_both, 2 * B, W, H, W, W * H, b["kpsM"], b["descM"], b["countsM"], b["statusM"], cap)
                _capi.check(L.orbx_stereo_match_batch_device(
                    e.handle, e.handle, B, _capi.ptr(b["kpsM"][:B]), _capi.ptr(b["descM"][:B]), _capi.ptr(b["countsM"][:B]),
                    _capi.ptr(b["kpsM"][B:]), _capi.ptr(b["descM"][B:]), _capi.ptr(b["countsM"][B:]), cap, mb, mbf,
                    _capi.ptr(b["uright"]), _capi.ptr(b["depth"]), _capi.ptr(b["nmatch"])))
            elif stereo:     # one extractor handle and batch per eye
                e.extract_batch_device(d_imgs, B, W, H, W, W * H, b["kps"][1:], b["desc"][1:], b["counts"][1:], b["status"], cap)
                with torch.cuda.stream(stream_r):
                    exR.extract_batch_device(d_right, B, W, H, W, W * H, b["kpsR"], b["descR"], b["countsR"], b["statusR"], cap)
                _capi.check(L.orbx_stereo_match_batch_device(
                    e.handle, exR.handle, B, _capi.ptr(b["kps"][1:]), _capi.ptr(b["desc"][1:]), _capi.ptr(b["counts"][1:]),
                    _capi.ptr(b["kpsR"]), _capi.ptr(b["descR"]), _capi.ptr(b["countsR"]), cap, mb, mbf,
                    _capi.ptr(b["uright"]), _capi.ptr(b["depth"]), _capi.ptr(b["nmatch"])))
            else:            # mono: extract, then frame t against frame t-1
                e.extract_batch_device(d_imgs, B, W, H, W, W * H, b["kps"][1:], b["desc"][1:], b["counts"][1:], b["status"], cap)
                if i > 0 and NS > 1:
                    st.wait_event(carry_ready)      # slot 0 was filled at the end of step i-1 on another pipeline's stream
                _capi.check(L.orbx_match_bruteforce_device(
                    e.handle, B, _capi.ptr(b["desc"][1:]), _capi.ptr(b["counts"][1:]), cap * 32, _capi.ptr(b["desc"]),
                    _capi.ptr(b["counts"]), cap * 32, _capi.ptr(b["midx"]), _capi.ptr(b["mbest"]), _capi.ptr(b["msecond"]), cap))
                nxt = bufs[(i + 1) % NS]            # (NS == 1: this pipeline's own slot 0, after the match has read it)
                nxt["desc"][0].copy_(b["desc"][B]); nxt["counts"][0:1].copy_(b["counts"][B:B + 1])
                if NS > 1:
                    carry_ready.record(st)
            if gatherer is not None:
                if merged:
                    gatherer.gather(b["countsM"][:B], b["kpsM"][:B], b["descM"][:B], async_op=True)
                else:
                    gatherer.gather(b["counts"][1:], b["kps"][1:], b["desc"][1:], async_op=True)

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize(dev)
    status = (bufs[0]["statusM"] if merged else
              torch.stack([b["status"] for b in bufs] + ([bufs[0]["statusR"]] if stereo else []))).cpu().numpy()
    counts = (bufs[0]["countsM"] if merged else bufs[0]["counts"][1:]).cpu().numpy()
    n_kp = float(counts.mean())
    # every rank's extraction status and mean keypoint count go to every rank: a failing rank makes ALL ranks exit non-zero
    # together (a lone SystemExit would leave its peers blocked in the next barrier until the driver's timeout), and rank 0
    # prints what each device produced, so that a slow or failing device can be told from a slow collective
    mdev = dev if (world == 1 or args.backend == "nccl") else torch.device("cpu")   # the few-byte bookkeeping exchanges
    smax = float(status.max())
    if os.environ.get("ORBX_BENCH_INJECT_FAILURE") == str(rank):   # test hook: this rank reports a failed extraction (tests/test_sharding_gpu.py)
        smax = 7.0
    mine = torch.tensor([smax, n_kp], dtype=torch.float64, device=mdev)
    per_rank = [mine.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    per_rank_status = [int(t[0].item()) for t in per_rank]
    per_rank_keypoints = [round(float(t[1].item()), 1) for t in per_rank]
    if any(per_rank_status):
        if rank == 0:
            print(json.dumps({"error": "extraction reported a non-zero status", "per_rank_status": per_rank_status,
                              "per_rank_keypoints": per_rank_keypoints}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)
    handles = exs + ([exR] if exR is not None else [])

    # calibration pass: which kernel dominates?  (all kernels timed with HIP events on the launch stream)
    for e in handles:
        e.profile_enable(0x1ff)
    torch.cuda.synchronize(dev)
    tc = time.perf_counter()
    for _ in range(2 * NS):
        step()
    if gatherer is not None:
        gatherer.wait_all()
    torch.cuda.synchronize(dev)
    calib_ms = (time.perf_counter() - tc) / (2 * NS) * 1e3
    prof = {}
    for e in handles:
        for kname, (ms, n) in e.profile_read(reset=True).items():
            a = prof.get(kname, (0.0, 0))
            prof[kname] = (a[0] + ms, a[1] + n)
    prof = {kname: (v[0] / NS, v[1] // NS) for kname, v in prof.items()}   # per 2 steps, like the single-pipeline table
    dominant = max((k for k in prof if k != "misc"), key=lambda k: prof[k][0])
    kid = _capi.K_NAMES.index(dominant)
    for e in handles:
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
    dt_own = time.perf_counter() - t0      # this rank's own K steps (its records handed over), before it waits for the others
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    own = torch.tensor([dt_own], dtype=torch.float64, device=mdev)
    per_rank_dt = [own.clone() for _ in range(world)]
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank_dt, own)
    dt = float(tmax.item())
    per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 4) for t in per_rank_dt]
    # the exchange alone (events around the synchronous collective on the buffers of the last step), MAX over ranks: what a
    # step would cost if nothing but the gather of its records ran -- with the per-rank times above it attributes a
    # sub-linear scaling curve to the collective, to rank 0's ingress, or to one slow device
    gather_ms = None
    if gatherer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gatherer.collective_only()
        torch.cuda.synchronize(dev)
        dist.barrier()
        e0.record()
        for _ in range(4):
            gatherer.collective_only()
        e1.record()
        torch.cuda.synchronize(dev)
        g = torch.tensor([e0.elapsed_time(e1) / 4], dtype=torch.float64, device=dev)
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
        gather_ms = round(float(g.item()), 4)
    dom_ms, dom_launches = 0.0, 0
    for e in handles:
        ms_, n_ = e.profile_read(reset=True)[dominant]
        dom_ms += ms_; dom_launches += n_
        e.profile_enable(0)

    if rank == 0:
        fps = world * B * args.steps / dt                      # mono: frames/s; stereo: stereo frames (left + right image)/s
        imgs_per_unit = 2 if stereo else 1
        model, fused = algorithmic_bytes(ex, W, H, n_kp, stereo)
        launches_per_step = dom_launches / max(args.steps, 1)
        # bytes one launch of the dominant kernel processes (both eyes' launches counted as launches)
        bytes_per_launch = model[dominant] * B * imgs_per_unit / max(launches_per_step, 1e-9)
        avg_launch_s = dom_ms * 1e-3 / max(dom_launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        total_model = sum(model.values()) * imgs_per_unit
        total_fused = sum(fused.values()) * imgs_per_unit
        traffic = committed_counters("traffic", args.config, B, W, H, NF)
        sq = committed_counters("sq", args.config, B, W, H, NF)
        # per-port busy fractions of the dominant kernel over its LIVE launch duration, at the clock the chip held during that
        # kernel (GRBM_GUI_ACTIVE / 8 / duration of the counter pass).  The ports issue independently (a vector, a scalar and an
        # LDS instruction of different waves can leave a SIMD / CU in the same cycle), so the fractions do not add:
        #   valu = SQ_INSTS_VALU x issue cycles per instruction of the kernel's rate-class mix (static, tools/valu_mix.py:
        #          2 full-rate, 3 shifts, 4 half-rate classes) / (1024 SIMDs x cycles)
        #   salu = SQ_INSTS_SALU / (256 CUs x cycles)      (one scalar instruction per cycle per CU)
        #   lds  = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles)  (LDS-array cycles, bank conflicts included)
        ports = None
        if sq is not None and dominant in sq["kernels"] and avg_launch_s > 0:
            ks = sq["kernels"][dominant]
            clk = ks.get("clock_ghz")
            cyc = avg_launch_s * (clk if clk else SCLK_PEAK_HZ / 1e9) * 1e9
            cpv = ks.get("valu_mix_static", {}).get("cycles_per_valu")
            ports = {"clock_ghz": clk, "clock_source": "GRBM_GUI_ACTIVE/8/duration (PMC pass)" if clk else "peak clock assumed",
                     "valu_frac": round(ks["SQ_INSTS_VALU"] * cpv / (N_SIMD * cyc), 4) if cpv else None,
                     "valu_cycles_per_inst_static": cpv,
                     "valu_frac_if_all_full_rate": round(ks["SQ_INSTS_VALU"] * 2 / (N_SIMD * cyc), 4),
                     "salu_frac": round(ks["SQ_INSTS_SALU"] / (N_SIMD / 4 * cyc), 4),
                     "lds_frac": round(ks["SQ_LDS_IDX_ACTIVE"] / (N_SIMD / 4 * cyc), 4) if "SQ_LDS_IDX_ACTIVE" in ks else None,
                     "lds_bank_conflict_share": round(ks["SQ_LDS_BANK_CONFLICT"] / ks["SQ_LDS_IDX_ACTIVE"], 4) if ks.get("SQ_LDS_IDX_ACTIVE") else None,
                     "valu_lane_inst_per_pixel": round(ks["SQ_INSTS_VALU"] * 64 / (model[dominant] * B * imgs_per_unit / max(launches_per_step, 1e-9)), 2)
                                                 if dominant == "k_fast_rows" else None}
        metric = {"tum": "frames/sec ORB extract+match (1000 kp, 640x480)"}.get(
            args.config if (W, H, NF) == (640, 480, 1000) else "", f"frames/sec ORB extract+match ({NF} kp, {W}x{H}{', stereo' if stereo else ''})")
        kind = (f"synthetic {W}x{H} stereo stream (stereo frame = left + right image, both extracted), nFeatures={NF}, 8 levels, "
                f"scale 1.2, FAST 20/7, extract x2 + ComputeStereoMatches, batch {B} stereo frames/GPU resident in HBM"
                + (f", both eyes through one extractor batch of {2 * B} images" if merged else ", one extractor batch per eye")
                if stereo else
                f"synthetic {W}x{H} mono stream, nFeatures={NF}, 8 levels, scale 1.2, FAST 20/7, "
                f"extract+match(t vs t-1), batch {B} frames/GPU resident in HBM, {NS} pipelines/GPU")
        out = {
            "metric": metric,
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": kind + (f", fork level {os.environ['ORBX_FORK_LEVEL']}" if int(os.environ.get("ORBX_FORK_LEVEL", "0")) > 0 else ""), "name": args.config, "frames_per_gpu_per_step": B, "mean_keypoints": round(n_kp, 1),
                       "images_per_s": round(fps * imgs_per_unit, 1),
                       "parallelism": f"frames sharded x{world}, RCCL {args.gather} of keypoint records per step" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": int(traffic["kernels"][dominant]["hbm_bytes_per_launch"]) if traffic and dominant in traffic["kernels"] else None,
                         "ports": ports,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "avg_launch_us": round(avg_launch_s * 1e6, 2),
                         # SURVEY 8(d) model (unfused pipeline: includes 2P for a blurred image that is never written here) ...
                         "end_to_end_GBs": round(total_model * fps / world / 1e9, 2),
                         "end_to_end_frac": round(total_model * fps / world / 1e9 / HBM_PEAK_GBS, 5),
                         # ... and the bytes the fused kernels really have to move
                         "end_to_end_fused_GBs": round(total_fused * fps / world / 1e9, 2),
                         "end_to_end_fused_frac": round(total_fused * fps / world / 1e9 / HBM_PEAK_GBS, 5)},
            # HIP events of the CALIBRATION pass (every kernel slot timed, 2 steps per pipeline), which runs slower than the timed
            # region (one slot timed): its own wall time per step is printed beside the sum so that the two can be compared
            "kernel_ms_per_step": {k: round(v[0] / 2, 4) for k, v in prof.items()},
            "kernel_ms_sum": round(sum(v[0] / 2 for v in prof.values()), 4),
            "calibration_ms_per_step": round(calib_ms, 4),
            # with several pipelines, or the forked launch sequence, per-kernel durations include time shared with other kernels
            "kernel_ms_overlapped": NS > 1 or int(os.environ.get("ORBX_FORK_LEVEL", "0")) > 0 or (stereo and args.stereo_streams == 2),
        }
        if world > 1:
            out.update({"per_rank_status": per_rank_status, "per_rank_keypoints": per_rank_keypoints,
                        "per_rank_ms_per_step": per_rank_ms, "gather_ms_per_step": gather_ms,
                        "gather_bytes_per_rank_per_step": int(B * sharding.record_bytes(cap))})
        if not stereo and prof["k_match"][0] > 0 and os.environ.get("ORBX_MATCH_KERNEL") != "valu":
            # the one GEMM-shaped kernel of the path: brute-force Hamming on the matrix pipe (dist = |q| + |t| - 2 q.t), by default
            # with FP4 operands (v_mfma_scale_f32_32x32x64_f8f6f4, dense peak 10 Pop/s), with ORBX_MATCH_KERNEL=i8 as
            # v_mfma_i32_32x32x32_i8 (5 Pop/s); 2 operations per (query bit, train bit) pair.
            # Duration: HIP events of the calibration pass (every kernel alone on its stream).
            m_us = prof["k_match"][0] / 2 * 1e3          # per step: one launch (plus the merge kernel for small batches)
            m_ops = 2.0 * B * n_kp * n_kp * 256
            m_i8 = os.environ.get("ORBX_MATCH_KERNEL") == "i8"
            m_peak = 5000.0 if m_i8 else 10000.0
            out["roofline_mfma"] = {"bound": "mfma", "kernel": "k_match", "operands": "int8" if m_i8 else "fp4 (e2m1)",
                                    "achieved": round(m_ops / (m_us * 1e-6) / 1e12, 1),
                                    "peak": m_peak, "unit": "TOP/s", "frac": round(m_ops / (m_us * 1e-6) / 1e12 / m_peak, 4),
                                    "avg_launch_us": round(m_us, 2)}
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only (the other ranks would idle through it)
            cfps, nsample, pinned, cflags = cpu_baseline(frames, right, NF, stereo, mb, mbf)
            out["cpu_baseline"] = {"value": round(cfps, 2), "unit": "frames/s", "cores": 1, "kind": "port", "flags": cflags,
                                   "sample": f"{nsample} {'stereo frames' if stereo else 'frames'} of the same synthetic stream, "
                                             f"single-thread CPU oracle (extract{' x2 + ComputeStereoMatches' if stereo else ' + brute-force match'}), "
                                             f"{'pinned to one core' if pinned else 'not pinned'}, host has {os.cpu_count()} cores"}
        if not stereo and not args.no_host_io and world == 1:
            out["host_io"] = host_io_rate(L, frames, W, H, NF, cap, local_rank)
        if args.stages:
            for k, v in prof.items():
                print(f"  {k:14s} {v[0] / 2:9.4f} ms/step  ({v[1] // 2} launches/step)", file=sys.stderr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
