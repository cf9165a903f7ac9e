// compat/ORBextractor.h -- drop-in replacement of the reference's include/ORBextractor.h for Tracking.cc / Frame.cc.
//
// Same namespace, class name, constructor, operator(), getters and the public mvImagePyramid member as
// reference include/ORBextractor.h:82-185; every call forwards to the C ABI of liborbx.so (include/orbx.h).
// Needs OpenCV headers (cv::Mat / cv::KeyPoint), which this build image does not have: the file is compiled
// by the maintainer inside the ORB-SLAM2 tree, not by this repository's tests.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H
#include <vector>
#include <stdexcept>
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "orbx.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
        orbx_params p;
        orbx_default_params(&p);
        p.nfeatures = nfeatures; p.scale_factor = scaleFactor; p.nlevels = nlevels;
        p.ini_th_fast = iniThFAST; p.min_th_fast = minThFAST;
        if (orbx_create(&p, &h_) != ORBX_OK) throw std::runtime_error(orbx_last_error());
        nlevels_ = nlevels;
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        orbx_get_scale_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                              mvInvLevelSigma2.data());
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() { orbx_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // mask is ignored, exactly as in the reference (include/ORBextractor.h:112)
    void operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors) {
        if (image.empty()) return;                                   // src/ORBextractor.cc:1966-1967
        cv::Mat im = image.getMat();
        CV_Assert(im.type() == CV_8UC1);                             // :1972
        const int cap = orbx_max_keypoints(h_, im.cols, im.rows);
        if (cap < 0) throw std::runtime_error(orbx_last_error());
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be the 28-byte POD");
        std::vector<orbx_keypoint> kps(cap);
        cv::Mat desc(cap, 32, CV_8U);
        int n = 0;
        const orbx_status st = orbx_extract(h_, im.data, im.cols, im.rows, (int)im.step, kps.data(), desc.data, cap, &n);
        if (st != ORBX_OK) throw std::runtime_error(orbx_last_error());
        keypoints.clear();
        keypoints.reserve(n);
        if (n == 0) { descriptors.release(); }                       // :1999-2002
        else {
            descriptors.create(n, 32, CV_8U);                        // :2006-2012
            desc.rowRange(0, n).copyTo(descriptors.getMat());
            const cv::KeyPoint *src = reinterpret_cast<const cv::KeyPoint *>(kps.data());
            keypoints.insert(keypoints.end(), src, src + n);
        }
        // mvImagePyramid is public and Frame::ComputeStereoMatches reads it right after ExtractORB (src/Frame.cc:910,1040,
        // 1072,1079), so by default the eight levels are copied back after every call: swapping in this header ALONE stays a
        // drop-in.  An integration that also applies compat/Frame_stereo.inl (orbx_stereo_match: the pyramids stay on the
        // device) turns the copy off with SetEagerPyramid(false) -- the .inl does it on first use -- and saves 1.16 MB of
        // device-to-host traffic per 640x480 frame; FetchPyramid() then fills the member on demand.
        pyramid_stale_ = true;
        if (eager_pyramid_) FetchPyramid();
    }

    // Fills mvImagePyramid with the padded levels of the last frame (fork semantics, src/ORBextractor.cc:2165-2166).  For a
    // caller that keeps the reference's own ComputeStereoMatches nothing needs doing (eager copy is the default); after
    // SetEagerPyramid(false) call it on both extractors before anything reads mvImagePyramid.
    void FetchPyramid() {
        if (!pyramid_stale_) return;
        for (int l = 0; l < nlevels_; ++l) {
            int w, hgt, pitch;
            if (orbx_pyramid_level_info(h_, l, &w, &hgt, &pitch) != ORBX_OK) return;   // nothing extracted yet
            mvImagePyramid[l].create(hgt, w, CV_8U);
            orbx_pyramid_level_copy(h_, 0, l, mvImagePyramid[l].data, (int)mvImagePyramid[l].step);
        }
        pyramid_stale_ = false;
    }
    void SetEagerPyramid(bool on) { eager_pyramid_ = on; }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return orbx_get_scale_factor(h_); }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<cv::Mat> mvImagePyramid;

    orbx_handle *handle() { return h_; }   // for compat/Frame_stereo.inl and compat/ORBmatcher.h

protected:
    orbx_handle *h_ = nullptr;
    int nlevels_ = 0;
    bool pyramid_stale_ = true, eager_pyramid_ = true;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM2
#endif
