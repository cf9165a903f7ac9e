// tests/compat_stubs: translation unit of the typo guard (g++ -fsyntax-only): every method of the matcher shim is instantiated
#include "ORBmatcher.h"
using namespace ORB_SLAM2;
int use(Frame &F, Frame &F2, KeyFrame *k1, KeyFrame *k2, std::vector<MapPoint *> &mps, std::set<MapPoint *> &found, cv::Mat S,
        cv::Mat R, cv::Mat t, cv::Mat F12, std::vector<cv::Point2f> &prev, std::vector<int> &m12,
        std::vector<std::pair<size_t, size_t> > &pairs) {
    ORBmatcher m(0.9f, true);
    int n = ORBmatcher::DescriptorDistance(F.mDescriptors, F2.mDescriptors);
    n += m.SearchByProjection(F, mps, 3.f);
    n += m.SearchByProjection(F, F2, 15.f, true);
    n += m.SearchByProjection(F, k1, found, 10.f, 100);
    n += m.SearchByProjection(k1, S, mps, mps, 10);
    n += m.SearchByBoW(k1, F, mps);
    n += m.SearchByBoW(k1, k2, mps);
    n += m.SearchForInitialization(F, F2, prev, m12, 100);
    n += m.SearchForTriangulation(k1, k2, F12, pairs, false);
    n += m.SearchBySim3(k1, k2, mps, 1.f, R, t, 7.5f);
    n += m.Fuse(k1, mps, 3.f);
    n += m.Fuse(k1, S, mps, 4.f, mps);
    std::vector<KeyFrame *> kfs(2, k1);
    n += m.FuseBatch(kfs, mps, 3.f);
    ORBmatcher::TriangulationBatch tb(k1, kfs);
    n += tb.Search(m, 1, F12, pairs, false);
    return n + ORBmatcher::TH_LOW + ORBmatcher::TH_HIGH + ORBmatcher::HISTO_LENGTH;
}
