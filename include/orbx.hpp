// orbx.hpp -- header-only C++17 convenience layer over the C ABI of liborbx.so (include/orbx.h).
// No OpenCV, no HIP headers: this is what host code that is not ORB-SLAM2 itself (tools, tests, other front-ends) uses;
// the OpenCV-typed drop-in classes with the reference's names live in compat/ (they need OpenCV headers).
// Every method forwards to the C entry point named in its comment; errors become std::runtime_error(orbx_last_error()).
#ifndef ORBX_HPP
#define ORBX_HPP
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "orbx.h"

namespace orbx {

inline void check(orbx_status st) {
    if (st != ORBX_OK) throw std::runtime_error(std::string(orbx_status_string(st)) + ": " + orbx_last_error());
}

struct Features {
    std::vector<orbx_keypoint> keypoints;   // bit-compatible with cv::KeyPoint
    std::vector<uint8_t> descriptors;       // keypoints.size() x 32
};

// ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:82-185) without the OpenCV types
class Extractor {
public:
    Extractor(int nfeatures = 1000, float scale_factor = 1.2f, int nlevels = 8, int ini_th = 20, int min_th = 7,
              int max_batch = 1, int device = -1) {
        orbx_params p;
        orbx_default_params(&p);
        p.nfeatures = nfeatures; p.scale_factor = scale_factor; p.nlevels = nlevels;
        p.ini_th_fast = ini_th; p.min_th_fast = min_th; p.max_batch = max_batch; p.device = device;
        check(orbx_create(&p, &h_));                                                     // orbx_create
    }
    ~Extractor() { orbx_destroy(h_); }
    Extractor(const Extractor &) = delete;
    Extractor &operator=(const Extractor &) = delete;
    Extractor(Extractor &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}

    orbx_handle *handle() const { return h_; }
    int levels() const { return orbx_get_levels(h_); }                                   // GetLevels
    float scale_factor() const { return orbx_get_scale_factor(h_); }                     // GetScaleFactor
    int max_keypoints(int width, int height) const {                                     // orbx_max_keypoints
        const int n = orbx_max_keypoints(h_, width, height);
        if (n < 0) throw std::runtime_error(orbx_last_error());
        return n;
    }
    // operator()(image, mask, keypoints, descriptors): empty image -> empty result, like :1966-1967
    Features operator()(const uint8_t *gray, int width, int height, int stride) const {
        Features f;
        if (!gray || width <= 0 || height <= 0) return f;
        const int cap = max_keypoints(width, height);
        f.keypoints.resize(cap); f.descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(orbx_extract(h_, gray, width, height, stride, f.keypoints.data(), f.descriptors.data(), cap, &n));   // orbx_extract
        f.keypoints.resize(n); f.descriptors.resize((size_t)n * 32);
        return f;
    }
    // batch of frames in host memory (frame_stride bytes apart): orbx_extract_batch
    std::vector<Features> extract_batch(const uint8_t *imgs, int nframes, int width, int height, int stride, int64_t frame_stride) const {
        const int cap = max_keypoints(width, height);
        std::vector<orbx_keypoint> k((size_t)nframes * cap);
        std::vector<uint8_t> d((size_t)nframes * cap * 32);
        std::vector<int32_t> n(nframes);
        check(orbx_extract_batch(h_, nframes, imgs, width, height, stride, frame_stride, k.data(), d.data(), n.data(), cap));
        std::vector<Features> out(nframes);
        for (int i = 0; i < nframes; ++i) {
            out[i].keypoints.assign(k.begin() + (size_t)i * cap, k.begin() + (size_t)i * cap + n[i]);
            out[i].descriptors.assign(d.begin() + (size_t)i * cap * 32, d.begin() + ((size_t)i * cap + n[i]) * 32);
        }
        return out;
    }
    // mvImagePyramid[level] of frame `frame` of the last call: orbx_pyramid_level_info + orbx_pyramid_level_copy
    std::vector<uint8_t> pyramid_level(int level, int &width, int &height, int frame = 0) const {
        int pitch = 0;
        check(orbx_pyramid_level_info(h_, level, &width, &height, &pitch));
        std::vector<uint8_t> img((size_t)width * height);
        check(orbx_pyramid_level_copy(h_, frame, level, img.data(), width));
        return img;
    }

private:
    orbx_handle *h_ = nullptr;
};

struct Match { int32_t index, distance, second; };

// ORBmatcher::DescriptorDistance over all pairs with best / second-best bookkeeping: orbx_match_bruteforce
inline std::vector<Match> match_bruteforce(const Extractor &ex, const Features &query, const Features &train) {
    const int nq = (int)query.keypoints.size(), nt = (int)train.keypoints.size();
    std::vector<int32_t> idx(nq, -1), best(nq, 0x7fffffff), second(nq, 0x7fffffff);
    if (nq > 0)
        check(orbx_match_bruteforce(ex.handle(), query.descriptors.data(), nq, train.descriptors.data(), nt, idx.data(), best.data(), second.data()));
    std::vector<Match> m(nq);
    for (int i = 0; i < nq; ++i) m[i] = Match{idx[i], best[i], second[i]};
    return m;
}

}  // namespace orbx
#endif
