// compat/ORBmatcher.h -- drop-in replacement of the reference's include/ORBmatcher.h + src/ORBmatcher.cc.
//
// Same namespace, class name, constructor, TH_LOW / TH_HIGH / HISTO_LENGTH and the eleven search-method signatures of
// reference include/ORBmatcher.h:54-225, so Tracking.cc, LocalMapping.cc and LoopClosing.cc compile unchanged
// (callers: src/Tracking.cc:950-962, 1281, 1430, 1445, 1953, 2115, 2419, 2448; src/LocalMapping.cc:349, 430, 750-807;
// src/LoopClosing.cc:422-636, 986-1004).  Every method flattens the Frame / KeyFrame / MapPoint fields the policy reads into
// the orbx_*_view structs of include/orbx.h, calls ONE entry point of liborbx.so, and writes the result back into the
// reference's containers in the order the reference would have.  What stays here is only what is pointer chasing or the
// reference's own OpenCV pose algebra in front of a loop (projection of a MapPoint, distance / viewing-angle gates,
// MapPoint::PredictScale); candidate gating (GetFeaturesInArea), Hamming distances, best / second bookkeeping, the
// rotation histogram and every order-dependent skip rule run behind the C ABI, bit-identical to the CPU path
// (tests/test_gpu_policies.py, test_bow_policies.py, test_projection_policies.py).
//
// Needs the ORB-SLAM2 tree (Frame.h, KeyFrame.h, MapPoint.h) and OpenCV: compiled by the maintainer,
// not by this repository's tests -- this build image has no OpenCV.  Remove src/ORBmatcher.cc from the library's sources.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <climits>
#include <cmath>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"
#include "orbx.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // include/ORBmatcher.h:71 -- a single pair stays the inline popcount it is (src/ORBmatcher.cc:2073-2093); batches go
    // through orbx_match_bruteforce / the policies below
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) {
        const int32_t *pa = a.ptr<int32_t>(), *pb = b.ptr<int32_t>();
        int dist = 0;
        for (int i = 0; i < 8; ++i) dist += __builtin_popcount((unsigned)(pa[i] ^ pb[i]));
        return dist;
    }

    // ---- include/ORBmatcher.h:83 -- src/ORBmatcher.cc:69-184, caller Tracking::SearchLocalPoints (src/Tracking.cc:1953)
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3) {
        const int nmp = (int)vpMapPoints.size();
        std::vector<uint8_t> in_view(nmp), desc((size_t)nmp * 32);
        std::vector<float> proj((size_t)nmp * 3), view_cos(nmp);
        std::vector<int32_t> level(nmp), obs(nmp), frame_obs(F.N, -1), assigned(std::max(F.N, 1), -1);
        for (int i = 0; i < nmp; ++i) {
            MapPoint *p = vpMapPoints[i];
            in_view[i] = p->mbTrackInView && !p->isBad();
            proj[3 * i] = p->mTrackProjX; proj[3 * i + 1] = p->mTrackProjY; proj[3 * i + 2] = p->mTrackProjXR;
            level[i] = p->mnTrackScaleLevel; view_cos[i] = p->mTrackViewCos; obs[i] = p->Observations();
            CopyDescriptor(p->GetDescriptor(), &desc[(size_t)i * 32]);
        }
        for (int i = 0; i < F.N; ++i)
            if (F.mvpMapPoints[i]) frame_obs[i] = F.mvpMapPoints[i]->Observations();
        orbx_frame_view fv = FrameView(F);
        orbx_mappoint_view mv = {nmp, in_view.data(), proj.data(), level.data(), view_cos.data(), desc.data(), obs.data()};
        int n = 0;
        Check(orbx_search_by_projection_mappoints(Handle(), &fv, frame_obs.data(), &mv, th, mfNNratio, assigned.data(), &n));
        for (int i = 0; i < F.N; ++i)
            if (assigned[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[assigned[i]];
        return n;
    }

    // ---- include/ORBmatcher.h:95 -- src/ORBmatcher.cc:1702-1871, caller Tracking::TrackWithMotionModel (src/Tracking.cc:1430,1445)
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) {
        const int nl = LastFrame.N;
        std::vector<uint8_t> has(nl, 0), desc((size_t)nl * 32, 0);
        std::vector<float> pos((size_t)nl * 3, 0.f);
        std::vector<int32_t> obs(nl, 0), matched(std::max(CurrentFrame.N, 1), -1);
        for (int i = 0; i < nl; ++i) {
            MapPoint *p = LastFrame.mvpMapPoints[i];
            if (!p || LastFrame.mvbOutlier[i]) continue;
            has[i] = 1;
            const cv::Mat x = p->GetWorldPos();
            for (int k = 0; k < 3; ++k) pos[3 * i + k] = x.at<float>(k);
            CopyDescriptor(p->GetDescriptor(), &desc[(size_t)i * 32]);
            obs[i] = p->Observations();
        }
        orbx_frame_view cv_ = FrameView(CurrentFrame);
        orbx_last_frame_view lv = {};
        lv.keys_un = Keys(LastFrame.mvKeysUn); lv.n = nl; lv.has_map_point = has.data(); lv.world_pos = pos.data();
        lv.mp_desc = desc.data(); lv.observations = obs.data();
        CopyPose(LastFrame.mTcw, lv.Tcw);
        // the reference tests "mvpMapPoints[i2] && Observations() > 0" on the current frame (:1799-1801); TrackWithMotionModel
        // clears the current frame's MapPoints before the call (src/Tracking.cc:1420), which is the entry point's contract
        int n = 0;
        Check(orbx_search_by_projection_frame(Handle(), &cv_, &lv, th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, matched.data(), &n));
        for (int i2 = 0; i2 < CurrentFrame.N; ++i2)
            if (matched[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[matched[i2]];
        return n;
    }

    // ---- include/ORBmatcher.h:107 -- src/ORBmatcher.cc:1873-2020, caller Tracking::Relocalization (src/Tracking.cc:2419,2448)
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                           const int ORBdist) {
        const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3), tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
        const cv::Mat Ow = -Rcw.t() * tcw;
        const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
        Points pts((int)vpMPs.size());
        for (int i = 0; i < pts.n; ++i) {
            MapPoint *p = vpMPs[i];
            if (!p || p->isBad() || sAlreadyFound.count(p)) continue;
            const cv::Mat xw = p->GetWorldPos(), xc = Rcw * xw + tcw;
            const float invz = 1.0 / xc.at<float>(2);
            const float u = CurrentFrame.fx * xc.at<float>(0) * invz + CurrentFrame.cx;
            const float v = CurrentFrame.fy * xc.at<float>(1) * invz + CurrentFrame.cy;
            if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX || v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
            const float d = cv::norm(xw - Ow);
            if (d < p->GetMinDistanceInvariance() || d > p->GetMaxDistanceInvariance()) continue;
            pts.Set(i, u, v, 0.f, p->PredictScale(d, &CurrentFrame), p->GetDescriptor(), pKF->mvKeysUn[i].angle);
        }
        std::vector<uint8_t> has(std::max(CurrentFrame.N, 1), 0);
        for (int i = 0; i < CurrentFrame.N; ++i) has[i] = CurrentFrame.mvpMapPoints[i] != NULL;
        std::vector<int32_t> matched(std::max(CurrentFrame.N, 1), -1);
        orbx_target_view tv = TargetView(CurrentFrame);
        orbx_projected_points pv = pts.View();
        int n = 0;
        Check(orbx_search_by_projection_keyframe(Handle(), &tv, &pv, th, ORBdist, mbCheckOrientation ? 1 : 0, has.data(),
                                                 matched.data(), &n));
        for (int i2 = 0; i2 < CurrentFrame.N; ++i2)
            if (matched[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[matched[i2]];
        return n;
    }

    // ---- include/ORBmatcher.h:119 -- src/ORBmatcher.cc:415-560, caller LoopClosing::ComputeSim3
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints,
                           std::vector<MapPoint *> &vpMatched, int th) {
        cv::Mat Rcw, tcw, Ow;
        SplitSim3(Scw, Rcw, tcw, Ow);
        std::set<MapPoint *> found(vpMatched.begin(), vpMatched.end());
        found.erase(static_cast<MapPoint *>(NULL));
        Points pts((int)vpPoints.size());
        for (int i = 0; i < pts.n; ++i) {
            MapPoint *p = vpPoints[i];
            if (p->isBad() || found.count(p)) continue;
            float u, v, invz, d;
            if (!ProjectInto(pKF, p, Rcw, tcw, Ow, u, v, invz, d, /*normal gate*/ true)) continue;
            pts.Set(i, u, v, 0.f, p->PredictScale(d, pKF), p->GetDescriptor(), 0.f);
        }
        std::vector<uint8_t> matched(std::max((int)vpMatched.size(), 1), 0);
        for (size_t i = 0; i < vpMatched.size(); ++i) matched[i] = vpMatched[i] != NULL;
        std::vector<int32_t> best(std::max(pts.n, 1), -1);
        orbx_target_view tv = TargetView(pKF);
        orbx_projected_points pv = pts.View();
        int n = 0;
        Check(orbx_search_by_projection_sim3(Handle(), &tv, &pv, th, matched.data(), best.data(), &n));
        for (int i = 0; i < pts.n; ++i)           // point order: a later point may overwrite an earlier one's slot, as :543-547 does
            if (best[i] >= 0) vpMatched[best[i]] = vpPoints[i];
        return n;
    }

    // ---- include/ORBmatcher.h:138 -- src/ORBmatcher.cc:248-410, callers src/Tracking.cc:1281 (TrackReferenceKeyFrame), :2115
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) {
        const std::vector<MapPoint *> vpKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));
        KeyFrameArrays a(pKF, vpKF, /*bad points count as absent*/ true);
        FeatVec ff(F.mFeatVec);
        orbx_keyframe_view kv = a.View();
        orbx_featvec_view fv = ff.View();
        std::vector<int32_t> matched(std::max(F.N, 1), -1);
        int n = 0;
        Check(orbx_search_by_bow_keyframe_frame(Handle(), &kv, Keys(F.mvKeys), F.mDescriptors.ptr<uint8_t>(), F.N, &fv, mfNNratio,
                                                mbCheckOrientation ? 1 : 0, matched.data(), &n));
        for (int i = 0; i < F.N; ++i)
            if (matched[i] >= 0) vpMapPointMatches[i] = vpKF[matched[i]];
        return n;
    }

    // ---- include/ORBmatcher.h:139 -- src/ORBmatcher.cc:722-866, caller LoopClosing::ComputeSim3
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12) {
        const std::vector<MapPoint *> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
        vpMatches12 = std::vector<MapPoint *>(vp1.size(), static_cast<MapPoint *>(NULL));
        KeyFrameArrays a1(pKF1, vp1, true), a2(pKF2, vp2, true);
        orbx_keyframe_view v1 = a1.View(), v2 = a2.View();
        std::vector<int32_t> m12(std::max((int)vp1.size(), 1), -1);
        int n = 0;
        Check(orbx_search_by_bow_keyframes(Handle(), &v1, &v2, mfNNratio, mbCheckOrientation ? 1 : 0, m12.data(), &n));
        for (size_t i = 0; i < vp1.size(); ++i)
            if (m12[i] >= 0) vpMatches12[i] = vp2[m12[i]];
        return n;
    }

    // ---- include/ORBmatcher.h:156 -- src/ORBmatcher.cc:570-712, caller Tracking::MonocularInitialization (src/Tracking.cc:950-962)
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                int windowSize = 10) {
        static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f must be two packed floats");
        vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1);
        const float bounds[4] = {Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY};
        std::vector<int32_t> m12(std::max(F1.N, 1), -1);
        int n = 0;
        Check(orbx_search_for_initialization(Handle(), Keys(F1.mvKeysUn), F1.mDescriptors.ptr<uint8_t>(), F1.N, Keys(F2.mvKeysUn),
                                             F2.mDescriptors.ptr<uint8_t>(), F2.N, bounds,
                                             reinterpret_cast<float *>(vbPrevMatched.data()), windowSize, mfNNratio,
                                             mbCheckOrientation ? 1 : 0, m12.data(), &n));
        for (int i = 0; i < F1.N; ++i) vnMatches12[i] = m12[i];
        return n;
    }

    // ---- include/ORBmatcher.h:168 -- src/ORBmatcher.cc:879-1087, caller LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:349)
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                               const bool bOnlyStereo) {
        // epipole of KF1's centre in KF2 (:892-898)
        const cv::Mat C2 = pKF2->GetRotation() * pKF1->GetCameraCenter() + pKF2->GetTranslation();
        const float invz = 1.0f / C2.at<float>(2);
        const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx, ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
        const std::vector<MapPoint *> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
        KeyFrameArrays a1(pKF1, vp1, /*any non-NULL point counts (":929-932, 960-963")*/ false), a2(pKF2, vp2, false);
        orbx_keyframe_view v1 = a1.View(), v2 = a2.View();
        float F[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) F[3 * r + c] = F12.at<float>(r, c);
        std::vector<int32_t> m12(std::max(pKF1->N, 1), -1);
        int n = 0;
        Check(orbx_search_for_triangulation(Handle(), &v1, &v2, F, ex, ey, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                            m12.data(), &n));
        vMatchedPairs.clear();
        vMatchedPairs.reserve(n);
        for (int i = 0; i < pKF1->N; ++i)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return n;
    }

    // ---- the loop around the call above as ONE device round trip: LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:375-430)
    // searches the current keyframe against each covisible neighbour and turns the matches into MapPoints before the next search.
    //     ORBmatcher::TriangulationBatch batch(pKF1, vpNeighKFs);                  // in front of the loop: all Hamming distances
    //     ... in the loop, in place of matcher.SearchForTriangulation(pKF1, pKF2, F12, vMatchedIndices, false):
    //     batch.Search(matcher, i, F12, vMatchedIndices, false);                    // neighbour i = vpNeighKFs[i]
    // Search reads the MapPoint assignments of both keyframes when it is called, as the reference does (:929-932, 960-963), so
    // the points the loop has created since take part exactly as in the unbatched loop; a neighbour the loop skips is never selected.
    class TriangulationBatch {
    public:
        TriangulationBatch(KeyFrame *pKF1, const std::vector<KeyFrame *> &vpNeighKFs) : kf1_(pKF1), kfs_(vpNeighKFs), b_(NULL) {
            const std::vector<MapPoint *> vp1 = pKF1->GetMapPointMatches();
            KeyFrameArrays a1(pKF1, vp1, false);
            std::vector<KeyFrameArrays> a2;
            a2.reserve(kfs_.size());
            for (size_t k = 0; k < kfs_.size(); ++k) a2.push_back(KeyFrameArrays(kfs_[k], kfs_[k]->GetMapPointMatches(), false));
            std::vector<orbx_keyframe_view> v2(kfs_.size());
            std::vector<const orbx_keyframe_view *> p2(kfs_.size());
            for (size_t k = 0; k < kfs_.size(); ++k) { v2[k] = a2[k].View(); p2[k] = &v2[k]; }
            const orbx_keyframe_view v1 = a1.View();
            Check(orbx_triangulation_batch_create(Handle(), &v1, (int)kfs_.size(), p2.data(), &b_));
        }
        ~TriangulationBatch() { orbx_triangulation_batch_destroy(b_); }
        int Search(ORBmatcher &m, int i, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo) {
            KeyFrame *pKF2 = kfs_[i];
            const cv::Mat C2 = pKF2->GetRotation() * kf1_->GetCameraCenter() + pKF2->GetTranslation();   // epipole (:892-898)
            const float invz = 1.0f / C2.at<float>(2);
            const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx, ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
            const std::vector<MapPoint *> vp1 = kf1_->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();   // the assignments NOW
            KeyFrameArrays a1(kf1_, vp1, false), a2(pKF2, vp2, false);
            orbx_keyframe_view v1 = a1.View(), v2 = a2.View();
            float F[9];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) F[3 * r + c] = F12.at<float>(r, c);
            std::vector<int32_t> m12(std::max(kf1_->N, 1), -1);
            int n = 0;
            Check(orbx_triangulation_batch_select(b_, i, &v1, &v2, F, ex, ey, bOnlyStereo ? 1 : 0, m.mbCheckOrientation ? 1 : 0, m12.data(), &n));
            vMatchedPairs.clear();
            vMatchedPairs.reserve(n);
            for (int j = 0; j < kf1_->N; ++j)
                if (m12[j] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)j, (size_t)m12[j]));
            return n;
        }
    private:
        TriangulationBatch(const TriangulationBatch &);
        TriangulationBatch &operator=(const TriangulationBatch &);
        KeyFrame *kf1_;
        std::vector<KeyFrame *> kfs_;
        orbx_triangulation_batch *b_;
    };

    // ---- include/ORBmatcher.h:184 -- src/ORBmatcher.cc:1433-1690, caller LoopClosing::ComputeSim3
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                     const cv::Mat &t12, const float th) {
        const cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
        const cv::Mat sR12 = s12 * R12, sR21 = (1.0 / s12) * R12.t(), t21 = -sR21 * t12;
        const std::vector<MapPoint *> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vp1.size(), N2 = (int)vp2.size();
        std::vector<bool> done1(N1, false), done2(N2, false);
        for (int i = 0; i < N1; ++i) {
            MapPoint *p = vpMatches12[i];
            if (!p) continue;
            done1[i] = true;
            const int idx2 = p->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) done2[idx2] = true;
        }
        auto project = [&](const std::vector<MapPoint *> &vp, const std::vector<bool> &done, const cv::Mat &Rw, const cv::Mat &tw,
                           const cv::Mat &sR, const cv::Mat &t, KeyFrame *into, Points &pts) {
            for (int i = 0; i < pts.n; ++i) {
                MapPoint *p = vp[i];
                if (!p || done[i] || p->isBad()) continue;
                const cv::Mat c = sR * (Rw * p->GetWorldPos() + tw) + t;   // point in the other keyframe's camera
                if (c.at<float>(2) < 0.0) continue;
                const float iz = 1.0 / c.at<float>(2);
                const float u = into->fx * (c.at<float>(0) * iz) + into->cx, v = into->fy * (c.at<float>(1) * iz) + into->cy;   // :1491-1496
                if (!into->IsInImage(u, v)) continue;
                const float d = cv::norm(c);
                if (d < p->GetMinDistanceInvariance() || d > p->GetMaxDistanceInvariance()) continue;
                pts.Set(i, u, v, 0.f, p->PredictScale(d, into), p->GetDescriptor(), 0.f);
            }
        };
        Points p12(N1), p21(N2);
        project(vp1, done1, R1w, t1w, sR21, t21, pKF2, p12);
        project(vp2, done2, R2w, t2w, sR12, t12, pKF1, p21);
        orbx_target_view tv1 = TargetView(pKF1), tv2 = TargetView(pKF2);
        orbx_projected_points v12 = p12.View(), v21 = p21.View();
        std::vector<int32_t> m12(std::max(N1, 1), -1);
        int n = 0;
        Check(orbx_search_by_sim3(Handle(), &tv1, &tv2, &v12, &v21, th, m12.data(), &n));
        for (int i = 0; i < N1; ++i)
            if (m12[i] >= 0) vpMatches12[i] = vp2[m12[i]];
        return n;
    }

    // ---- include/ORBmatcher.h:197 -- src/ORBmatcher.cc:1100-1280, caller LocalMapping::SearchInNeighbors (src/LocalMapping.cc:750-807)
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0) {
        const cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
        Points pts((int)vpMapPoints.size());
        for (int i = 0; i < pts.n; ++i) {
            MapPoint *p = vpMapPoints[i];
            if (!p || p->isBad() || p->IsInKeyFrame(pKF)) continue;
            float u, v, invz, d;
            if (!ProjectInto(pKF, p, Rcw, tcw, Ow, u, v, invz, d, true)) continue;
            pts.Set(i, u, v, u - pKF->mbf * invz, p->PredictScale(d, pKF), p->GetDescriptor(), 0.f);
        }
        std::vector<int32_t> best(std::max(pts.n, 1), -1);
        orbx_target_view tv = TargetView(pKF);
        orbx_projected_points pv = pts.View();
        int n = 0;
        Check(orbx_fuse(Handle(), &tv, &pv, th, best.data(), &n));
        for (int i = 0; i < pts.n; ++i) {          // the map update of :1248-1275, in point order
            if (best[i] < 0) continue;
            MapPoint *p = vpMapPoints[i], *inKF = pKF->GetMapPoint(best[i]);
            if (inKF) {
                if (!inKF->isBad()) { if (inKF->Observations() > p->Observations()) p->Replace(inKF); else inKF->Replace(p); }
            } else {
                p->AddObservation(pKF, best[i]);
                pKF->AddMapPoint(p, best[i]);
            }
        }
        return n;
    }

    // ---- the loop around the call above, as ONE device round trip: LocalMapping::SearchInNeighbors (src/LocalMapping.cc:750-768)
    //        for (pKFi : vpTargetKFs) matcher.Fuse(pKFi, vpMapPointMatches);      ->      matcher.FuseBatch(vpTargetKFs, vpMapPointMatches);
    // (10-20 neighbour keyframes + their second neighbours: every synchronous orbx_fuse call pays ~60 us of upload / launch /
    // download before it has done any work; orbx_fuse_batch pays it once.)  Same map as the loop, step for step: the selection of
    // a keyframe depends on earlier iterations only through (a) pMP->isBad() / IsInKeyFrame(pKF), tested again here at the moment
    // the reference would test them (:1118-1121), and (b) the descriptor of a point that survived `pMPinKF->Replace(pMP)` --
    // MapPoint::Replace ends in ComputeDistinctiveDescriptors() -- so points whose 32 bytes changed are re-submitted (alone,
    // orbx_fuse) for the keyframes still to come.
    int FuseBatch(const std::vector<KeyFrame *> &vpTargetKFs, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0) {
        const int K = (int)vpTargetKFs.size(), N = (int)vpMapPoints.size();
        if (K == 0 || N == 0) return 0;
        std::vector<Points> pts;
        std::vector<std::vector<int32_t> > best((size_t)K, std::vector<int32_t>((size_t)N, -1));
        std::vector<orbx_target_view> tv((size_t)K);
        std::vector<orbx_projected_points> pv((size_t)K);
        std::vector<const orbx_target_view *> tp((size_t)K);
        std::vector<const orbx_projected_points *> pp((size_t)K);
        std::vector<int32_t *> bp((size_t)K);
        std::vector<int> nf((size_t)K, 0);
        pts.reserve((size_t)K);
        for (int k = 0; k < K; ++k) {
            KeyFrame *pKF = vpTargetKFs[k];
            const cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
            pts.push_back(Points(N));
            for (int i = 0; i < N; ++i) {
                MapPoint *p = vpMapPoints[i];
                if (!p || p->isBad() || p->IsInKeyFrame(pKF)) continue;
                float u, v, invz, d;
                if (!ProjectInto(pKF, p, Rcw, tcw, Ow, u, v, invz, d, true)) continue;
                pts[k].Set(i, u, v, u - pKF->mbf * invz, p->PredictScale(d, pKF), p->GetDescriptor(), 0.f);
            }
        }
        // one descriptor block for all keyframes (the same MapPoints are projected into each): orbx_fuse_batch uploads it once
        std::vector<uint8_t> all_desc((size_t)N * 32, 0);
        for (int i = 0; i < N; ++i)
            if (vpMapPoints[i] && !vpMapPoints[i]->isBad()) CopyDescriptor(vpMapPoints[i]->GetDescriptor(), &all_desc[(size_t)i * 32]);
        for (int k = 0; k < K; ++k) {   // (views after the vector stopped growing: they point into it)
            tv[k] = TargetView(vpTargetKFs[k]); pv[k] = pts[k].View(); pv[k].desc = all_desc.data();
            tp[k] = &tv[k]; pp[k] = &pv[k]; bp[k] = best[k].data();
        }
        Check(orbx_fuse_batch(Handle(), K, tp.data(), pp.data(), th, bp.data(), nf.data()));
        std::vector<uint8_t> dirty((size_t)N, 0);   // descriptor changed since the batch was computed
        bool any_dirty = false;
        int total = 0;
        for (int k = 0; k < K; ++k) {
            KeyFrame *pKF = vpTargetKFs[k];
            if (any_dirty) {   // (b): this keyframe's answers for the points whose descriptor changed, from their new descriptor
                Points sub(N);
                bool some = false;
                for (int i = 0; i < N; ++i) {
                    if (!dirty[i] || !pts[k].valid[i]) continue;
                    sub.Set(i, pts[k].uv[2 * i], pts[k].uv[2 * i + 1], pts[k].ur[i], pts[k].level[i], vpMapPoints[i]->GetDescriptor(), 0.f);
                    some = true;
                }
                if (some) {
                    std::vector<int32_t> b2((size_t)N, -1);
                    orbx_projected_points sv = sub.View();
                    int n2 = 0;
                    Check(orbx_fuse(Handle(), &tv[k], &sv, th, b2.data(), &n2));
                    for (int i = 0; i < N; ++i) if (sub.valid[i]) best[k][i] = b2[i];
                }
            }
            for (int i = 0; i < N; ++i) {              // the map update of :1248-1275, in point order
                if (best[k][i] < 0) continue;
                MapPoint *p = vpMapPoints[i];
                if (p->isBad() || p->IsInKeyFrame(pKF)) continue;   // (a): what an earlier keyframe's fusion did to this point
                MapPoint *inKF = pKF->GetMapPoint(best[k][i]);
                if (inKF) {
                    if (!inKF->isBad()) {
                        if (inKF->Observations() > p->Observations()) p->Replace(inKF);
                        else {
                            uint8_t before[32], after[32];
                            CopyDescriptor(p->GetDescriptor(), before);
                            inKF->Replace(p);
                            CopyDescriptor(p->GetDescriptor(), after);
                            for (int b = 0; b < 32; ++b) if (before[b] != after[b]) { dirty[i] = 1; any_dirty = true; break; }
                        }
                    }
                } else {
                    p->AddObservation(pKF, best[k][i]);
                    pKF->AddMapPoint(p, best[k][i]);
                }
                ++total;
            }
        }
        return total;
    }

    // ---- include/ORBmatcher.h:209 -- src/ORBmatcher.cc:1282-1430, caller LoopClosing::SearchAndFuse (src/LoopClosing.cc:986-1004)
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint) {
        cv::Mat Rcw, tcw, Ow;
        SplitSim3(Scw, Rcw, tcw, Ow);
        const std::set<MapPoint *> found = pKF->GetMapPoints();
        Points pts((int)vpPoints.size());
        for (int i = 0; i < pts.n; ++i) {
            MapPoint *p = vpPoints[i];
            if (p->isBad() || found.count(p)) continue;
            float u, v, invz, d;
            if (!ProjectInto(pKF, p, Rcw, tcw, Ow, u, v, invz, d, true)) continue;
            pts.Set(i, u, v, 0.f, p->PredictScale(d, pKF), p->GetDescriptor(), 0.f);
        }
        std::vector<int32_t> best(std::max(pts.n, 1), -1);
        orbx_target_view tv = TargetView(pKF);
        orbx_projected_points pv = pts.View();
        int n = 0;
        Check(orbx_fuse_sim3(Handle(), &tv, &pv, th, best.data(), &n));
        for (int i = 0; i < pts.n; ++i) {          // :1402-1417
            if (best[i] < 0) continue;
            MapPoint *inKF = pKF->GetMapPoint(best[i]);
            if (inKF) { if (!inKF->isBad()) vpReplacePoint[i] = inKF; }
            else { vpPoints[i]->AddObservation(pKF, best[i]); pKF->AddMapPoint(vpPoints[i], best[i]); }
        }
        return n;
    }

public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:49-51

protected:
    float mfNNratio;
    bool mbCheckOrientation;

    // ---------------------------------------------------------------------------------------------------------------
    // marshalling helpers (nothing below computes a result)
    static void Check(orbx_status st) { if (st != ORBX_OK) throw std::runtime_error(orbx_last_error()); }

    // one handle per calling thread: the reference uses matcher objects from the tracking, local-mapping and loop-closing
    // threads concurrently, and an orbx handle (one HIP stream + scratch arena) is not re-entrant
    static orbx_handle *Handle() {
        struct Holder { orbx_handle *h = NULL; ~Holder() { orbx_destroy(h); } };
        thread_local Holder t;
        if (!t.h) {
            orbx_params p;
            orbx_default_params(&p);          // fp_mode GCC_FMA: the reference as its own CMake flags build it
            Check(orbx_create(&p, &t.h));
        }
        return t.h;
    }
    static const orbx_keypoint *Keys(const std::vector<cv::KeyPoint> &v) {
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be the 28-byte POD");
        return reinterpret_cast<const orbx_keypoint *>(v.data());
    }
    static void CopyDescriptor(const cv::Mat &d, uint8_t *dst) { const uint8_t *s = d.ptr<uint8_t>(); for (int k = 0; k < 32; ++k) dst[k] = s[k]; }
    static void CopyPose(const cv::Mat &T, float *dst) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) dst[4 * r + c] = T.at<float>(r, c); }
    static orbx_frame_view FrameView(const Frame &F) {
        orbx_frame_view v = {};
        v.keys_un = Keys(F.mvKeysUn); v.desc = F.mDescriptors.ptr<uint8_t>(); v.u_right = F.mvuRight.data(); v.n = F.N;
        if (!F.mTcw.empty()) CopyPose(F.mTcw, v.Tcw);
        v.fx = F.fx; v.fy = F.fy; v.cx = F.cx; v.cy = F.cy;
        v.min_x = Frame::mnMinX; v.max_x = Frame::mnMaxX; v.min_y = Frame::mnMinY; v.max_y = Frame::mnMaxY;
        v.mb = F.mb; v.mbf = F.mbf;
        return v;
    }
    static orbx_target_view TargetView(const Frame &F) {
        orbx_target_view v = {};
        v.keys_un = Keys(F.mvKeysUn); v.desc = F.mDescriptors.ptr<uint8_t>(); v.u_right = F.mvuRight.data(); v.n = F.N;
        v.min_x = Frame::mnMinX; v.max_x = Frame::mnMaxX; v.min_y = Frame::mnMinY; v.max_y = Frame::mnMaxY;
        v.scale_factors = F.mvScaleFactors.data(); v.inv_level_sigma2 = F.mvInvLevelSigma2.data();
        return v;
    }
    static orbx_target_view TargetView(const KeyFrame *K) {
        orbx_target_view v = {};
        v.keys_un = Keys(K->mvKeysUn); v.desc = K->mDescriptors.ptr<uint8_t>(); v.u_right = K->mvuRight.data(); v.n = K->N;
        v.min_x = (float)K->mnMinX; v.max_x = (float)K->mnMaxX; v.min_y = (float)K->mnMinY; v.max_y = (float)K->mnMaxY;
        v.scale_factors = K->mvScaleFactors.data(); v.inv_level_sigma2 = K->mvInvLevelSigma2.data();
        return v;
    }
    // DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened in map order
    struct FeatVec {
        std::vector<uint32_t> node, index;
        std::vector<int32_t> begin;
        explicit FeatVec(const DBoW2::FeatureVector &fv) : begin(1, 0) {
            for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
                node.push_back(it->first);
                index.insert(index.end(), it->second.begin(), it->second.end());
                begin.push_back((int32_t)index.size());
            }
        }
        orbx_featvec_view View() const { orbx_featvec_view v = {(int32_t)node.size(), node.data(), begin.data(), index.data()}; return v; }
    };
    struct KeyFrameArrays {
        const KeyFrame *K;
        std::vector<uint8_t> has;
        FeatVec fv;
        KeyFrameArrays(const KeyFrame *k, const std::vector<MapPoint *> &vp, bool bad_is_absent) : K(k), has(std::max(k->N, 1), 0), fv(k->mFeatVec) {
            for (int i = 0; i < k->N && i < (int)vp.size(); ++i) has[i] = vp[i] != NULL && !(bad_is_absent && vp[i]->isBad());
        }
        orbx_keyframe_view View() const {
            orbx_keyframe_view v = {};
            v.keys_un = Keys(K->mvKeysUn); v.desc = K->mDescriptors.ptr<uint8_t>(); v.n = K->N; v.has_map_point = has.data();
            v.u_right = K->mvuRight.data(); v.feat_vec = fv.View();
            v.scale_factors = K->mvScaleFactors.data(); v.level_sigma2 = K->mvLevelSigma2.data();
            return v;
        }
    };
    // per-MapPoint inputs of the projection-guided entry points (orbx_projected_points)
    struct Points {
        int n;
        std::vector<uint8_t> valid, desc;
        std::vector<float> uv, ur, angle;
        std::vector<int32_t> level;
        explicit Points(int n_) : n(n_), valid(std::max(n_, 1), 0), desc((size_t)std::max(n_, 1) * 32, 0), uv((size_t)std::max(n_, 1) * 2, 0.f),
                                  ur(std::max(n_, 1), 0.f), angle(std::max(n_, 1), 0.f), level(std::max(n_, 1), 0) {}
        void Set(int i, float u, float v, float u_right, int lvl, const cv::Mat &d, float ang) {
            valid[i] = 1; uv[2 * i] = u; uv[2 * i + 1] = v; ur[i] = u_right; level[i] = lvl; angle[i] = ang;
            CopyDescriptor(d, &desc[(size_t)i * 32]);
        }
        orbx_projected_points View() const {
            orbx_projected_points v = {n, valid.data(), uv.data(), ur.data(), level.data(), desc.data(), angle.data()};
            return v;
        }
    };
    // Scw = s [R | t]: rotation, translation and camera centre with the scale divided out (:1292-1301, :424-433)
    static void SplitSim3(const cv::Mat &Scw, cv::Mat &Rcw, cv::Mat &tcw, cv::Mat &Ow) {
        const cv::Mat sR = Scw.rowRange(0, 3).colRange(0, 3);
        const float s = std::sqrt(sR.row(0).dot(sR.row(0)));
        Rcw = sR / s;
        tcw = Scw.rowRange(0, 3).col(3) / s;
        Ow = -Rcw.t() * tcw;
    }
    // the gates in front of "const float radius = th * ..." shared by Fuse (:1121-1166), Fuse(Scw) (:1316-1352) and
    // SearchByProjection(KF, Scw) (:445-492): positive depth, inside the image, distance range, viewing angle < 60 degrees
    static bool ProjectInto(KeyFrame *pKF, MapPoint *p, const cv::Mat &Rcw, const cv::Mat &tcw, const cv::Mat &Ow, float &u, float &v,
                            float &invz, float &dist, bool normal_gate) {
        const cv::Mat xw = p->GetWorldPos(), xc = Rcw * xw + tcw;
        if (xc.at<float>(2) < 0.0f) return false;
        invz = 1 / xc.at<float>(2);
        u = pKF->fx * (xc.at<float>(0) * invz) + pKF->cx;
        v = pKF->fy * (xc.at<float>(1) * invz) + pKF->cy;
        if (!pKF->IsInImage(u, v)) return false;
        const cv::Mat PO = xw - Ow;
        dist = cv::norm(PO);
        if (dist < p->GetMinDistanceInvariance() || dist > p->GetMaxDistanceInvariance()) return false;
        if (normal_gate && PO.dot(p->GetNormal()) < 0.5 * dist) return false;
        return true;
    }
};

}  // namespace ORB_SLAM2
#endif  // ORBMATCHER_H
