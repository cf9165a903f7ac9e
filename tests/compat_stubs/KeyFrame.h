// tests/compat_stubs: the KeyFrame members compat/ORBmatcher.h touches, as /root/reference/include/KeyFrame.h declares them
// (declarations only; see README.md in this directory)
#pragma once
#include <set>
#include <vector>
#include <opencv2/core/core.hpp>
#include "MapPoint.h"
#include "Frame.h"
namespace ORB_SLAM2 {
class KeyFrame {
public:
    cv::Mat GetPose(); cv::Mat GetRotation(); cv::Mat GetTranslation(); cv::Mat GetCameraCenter();
    std::vector<MapPoint *> GetMapPointMatches(); std::set<MapPoint *> GetMapPoints(); MapPoint *GetMapPoint(const size_t &idx);
    void AddMapPoint(MapPoint *pMP, const size_t &idx); bool IsInImage(const float &x, const float &y) const;
    const float fx, fy, cx, cy, invfx, invfy, mbf, mb;
    const int N;
    const std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    const std::vector<float> mvuRight, mvDepth;
    const cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    const int mnScaleLevels; const float mfScaleFactor, mfLogScaleFactor;
    const std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    const int mnMinX, mnMinY, mnMaxX, mnMaxY;
};
}  // namespace ORB_SLAM2
