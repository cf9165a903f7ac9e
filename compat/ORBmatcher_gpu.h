// compat/ORBmatcher_gpu.h -- GPU Hamming primitives for ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:54-225).
//
// ORBmatcher's search policies keep their pointer-chasing and order-dependent bookkeeping on the host
// (SURVEY.md Appendix E); what moves to the GPU is DescriptorDistance over candidate sets.  This header gives the
// matcher two batched calls on cv::Mat descriptor blocks; ORBmatcher::DescriptorDistance itself (a single pair,
// src/ORBmatcher.cc:2073-2093) stays the inline SWAR popcount it is.  Needs OpenCV; compiled by the maintainer.
#ifndef ORBMATCHER_GPU_H
#define ORBMATCHER_GPU_H
#include <vector>
#include <stdexcept>
#include <opencv2/core/core.hpp>
#include "orbx.h"

namespace ORB_SLAM2 {

// best / second-best Hamming neighbour of every row of `query` among the rows of `train`
// (the bookkeeping of the search loops, e.g. src/ORBmatcher.cc:627-640: strict '<', first minimum wins)
inline void HammingBestTwo(orbx_handle *h, const cv::Mat &query, const cv::Mat &train, std::vector<int> &bestIdx,
                           std::vector<int> &bestDist, std::vector<int> &secondDist) {
    CV_Assert(query.type() == CV_8U && train.type() == CV_8U && query.cols == 32 && train.cols == 32);
    CV_Assert(query.isContinuous() && train.isContinuous());
    bestIdx.assign(query.rows, -1); bestDist.assign(query.rows, INT_MAX); secondDist.assign(query.rows, INT_MAX);
    if (orbx_match_bruteforce(h, query.data, query.rows, train.data, train.rows, bestIdx.data(), bestDist.data(),
                              secondDist.data()) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
}

// full distance matrix (rows = query, cols = train, CV_16U) for policies that gate candidates on the host
inline cv::Mat HammingMatrix(orbx_handle *h, const cv::Mat &query, const cv::Mat &train) {
    CV_Assert(query.isContinuous() && train.isContinuous());
    cv::Mat D(query.rows, train.rows, CV_16U);
    if (orbx_hamming_matrix(h, query.data, query.rows, train.data, train.rows, D.ptr<uint16_t>()) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    return D;
}

}  // namespace ORB_SLAM2
#endif
