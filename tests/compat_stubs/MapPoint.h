// tests/compat_stubs: the MapPoint members compat/ORBmatcher.h touches, as /root/reference/include/MapPoint.h declares them
// (declarations only; see README.md in this directory)
#pragma once
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class KeyFrame;
class Frame;
class MapPoint {
public:
    cv::Mat GetWorldPos(); cv::Mat GetNormal(); cv::Mat GetDescriptor();
    int Observations(); bool isBad(); bool IsInKeyFrame(KeyFrame *pKF); int GetIndexInKeyFrame(KeyFrame *pKF);
    void AddObservation(KeyFrame *pKF, size_t idx); void Replace(MapPoint *pMP);
    float GetMinDistanceInvariance(); float GetMaxDistanceInvariance();
    int PredictScale(const float &currentDist, KeyFrame *pKF); int PredictScale(const float &currentDist, Frame *pF);
    float mTrackProjX, mTrackProjY, mTrackProjXR; bool mbTrackInView; int mnTrackScaleLevel; float mTrackViewCos;
    long unsigned int mnLastFrameSeen;
};
}  // namespace ORB_SLAM2
