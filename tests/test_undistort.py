"""Frame glue: Frame::UndistortKeyPoints / ComputeImageBounds, src/Frame.cc:770-865 =
cv::undistortPoints(K, D, R = I, P = K).  OpenCV 3.2 arithmetic restated (parity unpinned).
CPU: properties of the oracle -- identity for zero distortion, inverse of the forward distortion model to float precision,
TUM1.yaml intrinsics.  GPU: k_undistort (host entry, batched device entry, image bounds) bit-identical to the oracle."""
import numpy as np
import pytest
import oracle

TUM1 = dict(K=(517.306408, 516.469215, 318.643040, 255.313989), D=(0.262383, -0.953104, -0.005358, 0.002628, 1.163314))   # Examples/Monocular/TUM1.yaml
KITTI = dict(K=(718.856, 718.856, 607.1928, 185.2157), D=(0.0, 0.0, 0.0, 0.0))


def rand_keys(rng, n, w=640, h=480):
    k = np.zeros(n, oracle.KP_DTYPE)
    k["x"] = rng.uniform(0, w, n).astype(np.float32); k["y"] = rng.uniform(0, h, n).astype(np.float32)
    k["size"] = 31; k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["response"] = rng.integers(7, 255, n)
    k["octave"] = rng.integers(0, 8, n); k["class_id"] = -1
    return k


def distort(xu, yu, K, D):
    """forward model of OpenCV: undistorted pixel -> distorted pixel (double)"""
    fx, fy, cx, cy = K; k1, k2, p1, p2 = D[:4]; k3 = D[4] if len(D) > 4 else 0.0
    x, y = (xu - cx) / fx, (yu - cy) / fy
    r2 = x * x + y * y
    cd = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return xd * fx + cx, yd * fy + cy


def test_oracle_identity_without_distortion():
    k = rand_keys(np.random.default_rng(0), 500, 1241, 376)
    assert oracle.undistort_keypoints(k, KITTI["K"], KITTI["D"]).tobytes() == k.tobytes()
    assert oracle.image_bounds(1241, 376, KITTI["K"], KITTI["D"]).tolist() == [0.0, 1241.0, 0.0, 376.0]


def test_oracle_inverts_the_distortion_model():
    rng = np.random.default_rng(1)
    xu = rng.uniform(40, 600, 2000); yu = rng.uniform(40, 440, 2000)
    xd, yd = distort(xu, yu, TUM1["K"], TUM1["D"])
    k = np.zeros(2000, oracle.KP_DTYPE); k["x"] = xd.astype(np.float32); k["y"] = yd.astype(np.float32)
    un = oracle.undistort_keypoints(k, TUM1["K"], TUM1["D"])
    err = np.hypot(un["x"] - xu, un["y"] - yu)
    assert np.median(err) < 1e-4 and err.max() < 5e-3     # five fixed-point iterations converge to float precision here
    b = oracle.image_bounds(640, 480, TUM1["K"], TUM1["D"])
    # TUM1's k1 > 0: the undistorted corners move inwards (the well-known 10.8 .. 626.0 x 14.7 .. 473.3 window of TUM1 runs)
    assert 5 < b[0] < 20 and 620 < b[1] < 635 and 10 < b[2] < 20 and 465 < b[3] < 478


@pytest.mark.gpu
def test_gpu_undistort_equals_oracle():
    import torch
    from orb_slam2_detailed_comments_amd import ORBextractor, Frame, synth, _capi
    ex = ORBextractor(1000, max_batch=3)
    frames = synth.stream(640, 480, 3, stream_id=81)
    res = ex.extract_batch(frames)
    for cam in (TUM1, KITTI, dict(K=TUM1["K"], D=TUM1["D"][:4])):
        for k, d in res:
            F = Frame(k, d, 640, 480)
            un = F.UndistortKeyPoints(ex, cam["K"], cam["D"])
            assert un.tobytes() == oracle.undistort_keypoints(k, cam["K"], cam["D"]).tobytes()
            assert np.array_equal(np.array(F.bounds, np.float32).view(np.uint32), oracle.image_bounds(640, 480, cam["K"], cam["D"]).view(np.uint32))
            F.AssignFeaturesToGrid()
            got = F.GetFeaturesInArea(320.0, 240.0, 60.0)
            assert np.array_equal(got, oracle.grid_query(un, F.bounds, 320.0, 240.0, 60.0))
    # batched device entry on the buffers of extract_batch_device
    dev = torch.device("cuda", 0); cap = ex.max_keypoints(640, 480); B = 3
    kps = torch.zeros((B, cap * 28), dtype=torch.uint8, device=dev); desc = torch.zeros((B, cap * 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch_device(torch.from_numpy(frames).to(dev), B, 640, 480, 640, 640 * 480, kps, desc, cnt, st, cap)
    un = torch.zeros_like(kps)
    torch.cuda.synchronize()   # fills ran on torch's stream; the handle's stream is not ordered with it
    k4 = np.array(TUM1["K"], np.float32); dd = np.array(TUM1["D"], np.float32)
    _capi.check(_capi.lib().orbx_undistort_keypoints_device(ex.handle, B, _capi.ptr(kps), _capi.ptr(cnt), cap, _capi.ptr(k4), _capi.ptr(dd), 5, _capi.ptr(un)))
    ex.synchronize()
    for f in range(B):
        n = int(cnt[f])
        got = np.frombuffer(un[f].cpu().numpy().tobytes(), _capi.KP_DTYPE)[:n]
        assert got.tobytes() == oracle.undistort_keypoints(res[f][0], TUM1["K"], TUM1["D"]).tobytes()
