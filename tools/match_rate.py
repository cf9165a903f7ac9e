"""k_match alone: orbx_match_bruteforce_device on random descriptors, time per launch (wall clock over a run of launches).
   python tools/match_rate.py [npairs nq nt]      (ORBX_LIB selects a variant library)"""
import sys, time, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from orb_slam2_detailed_comments_amd import _capi
from orb_slam2_detailed_comments_amd.extractor import ORBextractor

npairs, nq, nt = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 1000, 1000)
L = _capi.lib()
ex = ORBextractor(300)
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randint(0, 256, (npairs, nq, 32), dtype=torch.uint8, device="cuda", generator=g)
t = torch.randint(0, 256, (npairs, nt, 32), dtype=torch.uint8, device="cuda", generator=g)
dnq = torch.full((npairs,), nq, dtype=torch.int32, device="cuda"); dnt = torch.full((npairs,), nt, dtype=torch.int32, device="cuda")
out = [torch.zeros((npairs, nq), dtype=torch.int32, device="cuda") for _ in range(3)]
torch.cuda.synchronize()


def run(n):
    for _ in range(n):
        _capi.check(L.orbx_match_bruteforce_device(ex.handle, npairs, _capi.ptr(q), _capi.ptr(dnq), nq * 32, _capi.ptr(t), _capi.ptr(dnt),
                                                   nt * 32, _capi.ptr(out[0]), _capi.ptr(out[1]), _capi.ptr(out[2]), nq))
    ex.synchronize()


run(20)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); run(50); best = min(best, (time.perf_counter() - t0) / 50)
ops = 2.0 * npairs * nq * nt * 256
print(f"{npairs} x {nq} x {nt}: {best * 1e6:.1f} us per launch, {ops / best / 1e12:.1f} Tbit-op/s (dense matrix-pipe peak: int8 ~5000, fp4 ~10000), checksum {int(out[1].sum())}")
