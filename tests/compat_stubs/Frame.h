// tests/compat_stubs: the Frame members compat/ORBmatcher.h and compat/Frame_stereo.inl touch, as /root/reference/include/
// Frame.h:310-446 declares them (declarations only; see README.md in this directory)
#pragma once
#include <map>
#include <vector>
#include <opencv2/core/core.hpp>
#include "ORBextractor.h"
#include "MapPoint.h"
namespace DBoW2 { class FeatureVector : public std::map<unsigned int, std::vector<unsigned int> > {}; }
namespace ORB_SLAM2 {
class Frame {
public:
    void ComputeStereoMatches(); void UndistortKeyPoints();
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    cv::Mat mK, mDistCoef; float fx, fy, cx, cy, invfx, invfy, mbf, mb;
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint *> mvpMapPoints; std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    int mnScaleLevels; float mfScaleFactor, mfLogScaleFactor;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
}  // namespace ORB_SLAM2
