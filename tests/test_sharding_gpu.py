"""Sharded == unsharded on real hardware (SURVEY.md section 7.4 "same batch through 1/2/4/8 devices => identical concatenated
output", section 8e).  Two fresh child ranks share GPU 0 (gloo rendezvous on 127.0.0.1; an 8-GPU node is the driver's): each
takes its contiguous block of 8 EuRoC-size stereo pairs (BASELINE.json config 4: 752x480, 1200 features) from
sharding.shard_range, extracts both eyes, runs ComputeStereoMatches on the device and hands its fixed-stride records to
sharding.RecordGatherer; rank 0 compares the gathered concatenation byte for byte with the same 8 pairs run through ONE process.
The fact it rests on: the extractor keeps no cross-frame state (reference include/ORBextractor.h:30-35)."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
W, H, NF, NPAIRS, MB, MBF = 752, 480, 1200, 8, 0.11, 47.90639384423901


def _run_block(begin, end):
    """extract both eyes of pairs [begin, end) + stereo match on cuda:0; returns CPU tensors"""
    from orb_slam2_detailed_comments_amd import ORBextractor, synth, _capi
    dev = torch.device("cuda", 0)
    n = end - begin
    pairs = [synth.stereo_pair(W, H, stream_id=300 + i) for i in range(begin, end)]
    L = _capi.lib()
    ex = {k: ORBextractor(NF, max_batch=n) for k in "LR"}
    cap = ex["L"].max_keypoints(W, H)
    out = {}
    for k, idx in (("L", 0), ("R", 1)):
        img = torch.from_numpy(np.stack([p[idx] for p in pairs])).to(dev)
        kps = torch.zeros((n, cap * 28), dtype=torch.uint8, device=dev); desc = torch.zeros((n, cap * 32), dtype=torch.uint8, device=dev)
        cnt = torch.zeros(n, dtype=torch.int32, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ex[k].extract_batch_device(img, n, W, H, W, W * H, kps, desc, cnt, st, cap)
        out[k] = (cnt, kps, desc, st)
    ex["R"].synchronize()
    ur = torch.zeros((n, cap), dtype=torch.float32, device=dev); dep = torch.zeros_like(ur); nm = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    _capi.check(L.orbx_stereo_match_batch_device(ex["L"].handle, ex["R"].handle, n, _capi.ptr(out["L"][1]), _capi.ptr(out["L"][2]),
                                                 _capi.ptr(out["L"][0]), _capi.ptr(out["R"][1]), _capi.ptr(out["R"][2]),
                                                 _capi.ptr(out["R"][0]), cap, MB, MBF, _capi.ptr(ur), _capi.ptr(dep), _capi.ptr(nm)))
    ex["L"].synchronize()
    assert not out["L"][3].any() and not out["R"][3].any()
    # entries beyond a frame's count are whatever the buffers held: compare the meaningful part only
    def clean(cnt, kps, desc):
        cnt, kps, desc = cnt.cpu(), kps.cpu().clone(), desc.cpu().clone()
        for f in range(n):
            kps[f, int(cnt[f]) * 28:] = 0; desc[f, int(cnt[f]) * 32:] = 0
        return cnt, kps, desc
    cl, cr = clean(*out["L"][:3]), clean(*out["R"][:3])
    ur, dep = ur.cpu().clone(), dep.cpu().clone()
    for f in range(n):
        ur[f, int(cl[0][f]):] = 0; dep[f, int(cl[0][f]):] = 0
    return cap, cl, cr, torch.cat([ur, dep, nm.cpu().float().view(n, 1)], dim=1)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from orb_slam2_detailed_comments_amd import sharding
        b, e = sharding.shard_range(NPAIRS, world, rank)
        cap, cl, cr, extra = _run_block(b, e)
        cpu = torch.device("cpu")
        gl, gr = (sharding.RecordGatherer(e - b, cap, cpu, mode="gather") for _ in range(2))
        bufl = gl.gather(*cl, async_op=True); bufr = gr.gather(*cr, async_op=True)
        gl.wait_all(); gr.wait_all()
        parts = [torch.empty_like(extra) for _ in range(world)] if rank == 0 else None
        dist.gather(extra, parts, dst=0)
        if rank != 0:
            q.put((rank, True, ""))
            return
        # the same pairs through one process
        cap1, fl, fr, fextra = _run_block(0, NPAIRS)
        ok, why = cap1 == cap, ""
        for name, buf, full in (("left", bufl, fl), ("right", bufr, fr)):
            c, k, d = sharding.unpack_records(buf, cap)
            for what, a, bb in (("counts", c, full[0]), ("keypoints", k, full[1]), ("descriptors", d, full[2])):
                if not torch.equal(a.contiguous(), bb):
                    ok = False; why += f"{name} {what} differ; "
        if not torch.equal(torch.cat(parts), fextra):
            ok = False; why += "mvuRight / mvDepth / match counts differ; "
        ok = ok and int(fl[0].min()) > 500 and float(fextra[:, -1].min()) > 50    # the comparison is not vacuous
        q.put((rank, ok, why))
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharing_gpu0_equal_one_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)], res


def test_bench_two_rank_line_carries_per_rank_fields():
    """the N > 1 path of bench.py, rehearsed with two ranks sharing GPU 0 (gloo; the 8-GPU run is the driver's): ONE json line from
    rank 0 with every rank's status / keypoint count / own step time and the collective's own time, so that the first scaling
    curve the driver takes can be attributed"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ORBX_BENCH_SHARE_GPU0="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "8",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["frames_per_gpu_per_step"] == 8
    assert j["per_rank_status"] == [0, 0] and len(j["per_rank_keypoints"]) == 2 and min(j["per_rank_keypoints"]) > 500
    assert len(j["per_rank_ms_per_step"]) == 2 and all(0 < t <= j["ms_per_step"] * 1.001 for t in j["per_rank_ms_per_step"])
    assert j["gather_ms_per_step"] > 0 and j["gather_bytes_per_rank_per_step"] > 8 * 1000 * 60
    assert j["value"] == pytest.approx(2 * 8 * 2 / (j["ms_per_step"] * 2 / 1e3), rel=1e-2)


def test_bench_failing_rank_takes_every_rank_down():
    """a rank whose extraction reports a non-zero status must not leave its peers in a barrier until the driver's timeout: every rank
    exits with code 3 after rank 0 has printed the per-rank status; the failure is injected on rank 1"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ORBX_BENCH_SHARE_GPU0="1", ORBX_BENCH_INJECT_FAILURE="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "8",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert p.returncode != 0
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    j = json.loads(line[0])
    assert j["per_rank_status"] == [0, 7] and "error" in j and len(j["per_rank_keypoints"]) == 2
    assert "value" not in j          # no throughput line from a failed run
