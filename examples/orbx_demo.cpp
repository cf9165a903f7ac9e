// examples/orbx_demo.cpp -- the C ABI used from plain C++ (no OpenCV, no HIP headers, no Python):
//   g++ -std=c++17 -O2 -Iinclude examples/orbx_demo.cpp -Lorb_slam2_detailed_comments_amd/lib -lorbx -Wl,-rpath,$PWD/orb_slam2_detailed_comments_amd/lib -o orbx_demo
// Renders two synthetic frames (the second shifted by 3 px), extracts ORB features from both through orbx::Extractor and
// matches them with the ratio test of ORBmatcher (best <= TH_LOW and best < 0.9 * second); prints a summary line.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "orbx.hpp"

static std::vector<uint8_t> render(int w, int h, int shift) {
    std::vector<uint8_t> img((size_t)w * h, 110);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int r = 0; r < 300; ++r) {                       // random rectangles: plenty of corners
        const int x = (int)(rnd() % (uint32_t)w) + shift, y = (int)(rnd() % (uint32_t)h), sx = 6 + (int)(rnd() % 50), sy = 6 + (int)(rnd() % 50);
        const uint8_t v = (uint8_t)(20 + rnd() % 215);
        for (int yy = y; yy < y + sy && yy < h; ++yy)
            for (int xx = x < 0 ? 0 : x; xx < x + sx && xx < w; ++xx) img[(size_t)yy * w + xx] = v;
    }
    return img;
}

int main() {
    try {
        const int W = 640, H = 480;
        orbx::Extractor ex(1000, 1.2f, 8, 20, 7);
        const auto a = render(W, H, 0), b = render(W, H, 3);
        const orbx::Features fa = ex(a.data(), W, H, W), fb = ex(b.data(), W, H, W);
        const auto m = orbx::match_bruteforce(ex, fb, fa);
        int good = 0;
        for (const auto &x : m)
            if (x.index >= 0 && x.distance <= 50 && (float)x.distance < 0.9f * (float)x.second) ++good;
        int w0 = 0, h0 = 0;
        ex.pyramid_level(0, w0, h0);
        std::printf("orbx_demo: %zu / %zu keypoints, %d ratio-test matches, level 0 is %dx%d (padded), abi %d\n", fa.keypoints.size(),
                    fb.keypoints.size(), good, w0, h0, orbx_abi_version());
        return (fa.keypoints.size() > 300 && good > 100) ? 0 : 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "orbx_demo: %s\n", e.what());
        return 1;
    }
}
