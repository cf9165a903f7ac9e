#!/usr/bin/env python3
"""Timeline of ONE orbx_extract_batch call from host memory (256 frames in chunks of 64), for
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/host_io_trace.py [pinned|pageable]
and, with a DIR argument instead, the summary of that trace: per copy / per chunk's kernels, start and end relative to the first
event of the LAST call (us)."""
import os, sys, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(mem, N=256):
    import numpy as np
    from orb_slam2_detailed_comments_amd import ORBextractor, synth, _capi
    W, H, mb = 640, 480, 64
    frames = np.concatenate([synth.stream(W, H, 64, stream_id=100)] * (N // 64))
    L = _capi.lib()
    ex = ORBextractor(1000, max_batch=mb)
    cap = ex.max_keypoints(W, H)
    if mem == "pinned":
        keep = [_capi.PinnedArray((N, H, W)), _capi.PinnedArray((N, cap), _capi.KP_DTYPE), _capi.PinnedArray((N, cap, 32)), _capi.PinnedArray((N,), np.int32)]
        img, kps, desc, cnt = (k.array for k in keep)
        img[...] = frames
    else:
        img, kps, desc, cnt = frames.copy(), np.zeros((N, cap), _capi.KP_DTYPE), np.zeros((N, cap, 32), np.uint8), np.zeros(N, np.int32)
    for _ in range(4):
        _capi.check(L.orbx_extract_batch(ex.handle, N, _capi.ptr(img), W, H, W, W * H, _capi.ptr(kps), _capi.ptr(desc), _capi.ptr(cnt), cap))
    print("done", mem, int(cnt.mean()))


def summarise(d):
    ev = []
    for fn in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?")), 0))
    for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:22], 1))
    ev.sort()
    # the last call = the events after the last gap > 1 ms... take the last quarter of the level-0 launches (4 calls x 4 chunks)
    l0 = [e for e in ev if e[2].startswith("k_pyr_l0")]
    nch = int(sys.argv[2]) if len(sys.argv) > 2 else 4   # chunks per call
    t0 = l0[-nch][0] - 600000 if len(l0) >= nch else ev[0][0]
    ev = [e for e in ev if e[0] >= t0]
    base = ev[0][0]
    for s, e, name, is_k in ev:
        if not is_k or name.startswith(("k_pyr_l0", "void k_describe")):
            print(f"{(s - base) / 1e3:9.1f} .. {(e - base) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  {name}")


if __name__ == "__main__":
    a = sys.argv[1] if len(sys.argv) > 1 else "pinned"
    summarise(a) if os.path.isdir(a) else run(a, int(sys.argv[2]) if len(sys.argv) > 2 else 256)
