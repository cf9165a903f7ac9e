// tools/dump_reference_frame.cc -- run INSIDE an ORB-SLAM2 tree (the reference, with its own OpenCV): extracts ORB features of one
// image with the REFERENCE's ORBextractor and writes everything a parity check needs into one binary file.  This repository
// cannot build it (no OpenCV in its image); it exists so that a maintainer can pin the bit-exactness claim against a real
// build once ("parity unpinned").
//
//   g++ -std=c++14 -O3 -march=native -I<ORB_SLAM2>/include -I<ORB_SLAM2> tools/dump_reference_frame.cc \
//       <ORB_SLAM2>/src/ORBextractor.cc `pkg-config --cflags --libs opencv` -o dump_reference_frame
//   ./dump_reference_frame image.png dump.bin [nfeatures=1000 scale=1.2 levels=8 iniTh=20 minTh=7]
//   python tools/compare_reference_dump.py dump.bin            (on the MI355X box, in this repository)
//
// File layout (little endian): "ORBXREF1" | int32 w, h, nfeatures, nlevels, iniTh, minTh | float scale | int32 n |
//   uint8 image[h][w] | cv::KeyPoint n x 28 bytes | uint8 desc[n][32] | per level: int32 cols, rows, uint8 pixels[rows][cols]
//   (the fork's padded mvImagePyramid, src/ORBextractor.cc:2165-2166)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <opencv2/core/core.hpp>
#include <opencv2/highgui/highgui.hpp>
#include "ORBextractor.h"

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s image out.bin [nfeatures scale levels iniTh minTh]\n", argv[0]); return 2; }
    const int nf = argc > 3 ? std::atoi(argv[3]) : 1000, nl = argc > 5 ? std::atoi(argv[5]) : 8;
    const float sf = argc > 4 ? (float)std::atof(argv[4]) : 1.2f;
    const int ini = argc > 6 ? std::atoi(argv[6]) : 20, mn = argc > 7 ? std::atoi(argv[7]) : 7;
    cv::Mat im = cv::imread(argv[1], 0 /* grayscale */);
    if (im.empty() || im.type() != CV_8UC1) { std::fprintf(stderr, "cannot read %s as 8-bit gray\n", argv[1]); return 1; }
    if (!im.isContinuous()) im = im.clone();
    ORB_SLAM2::ORBextractor ex(nf, sf, nl, ini, mn);
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    ex(im, cv::Mat(), kps, desc);
    static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint is expected to be the 28-byte POD");
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    const int32_t hdr[6] = {im.cols, im.rows, nf, nl, ini, mn};
    const int32_t n = (int32_t)kps.size();
    std::fwrite("ORBXREF1", 1, 8, f);
    std::fwrite(hdr, sizeof(hdr), 1, f);
    std::fwrite(&sf, 4, 1, f);
    std::fwrite(&n, 4, 1, f);
    std::fwrite(im.data, 1, (size_t)im.cols * im.rows, f);
    if (n) { std::fwrite(kps.data(), 28, (size_t)n, f); std::fwrite(desc.data, 32, (size_t)n, f); }
    for (int l = 0; l < nl; ++l) {
        cv::Mat lv = ex.mvImagePyramid[l].isContinuous() ? ex.mvImagePyramid[l] : ex.mvImagePyramid[l].clone();
        const int32_t d[2] = {lv.cols, lv.rows};
        std::fwrite(d, sizeof(d), 1, f);
        std::fwrite(lv.data, 1, (size_t)lv.cols * lv.rows, f);
    }
    std::fclose(f);
    std::printf("%d keypoints of %dx%d written to %s (OpenCV %s)\n", n, im.cols, im.rows, argv[2], CV_VERSION);
    return 0;
}
