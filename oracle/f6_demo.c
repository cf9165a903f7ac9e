/* F6 demonstration (TEST INFRASTRUCTURE ONLY): SURVEY.md finding F6.
 *
 * In fork mode (mvImagePyramid holds the PADDED images, src/ORBextractor.cc:2165-2166) Frame::ComputeStereoMatches sizes
 * vRowIndices by mvImagePyramid[0].rows (src/Frame.cc:910) but indexes it with rows kpY +- 2*scale[octave] of keypoints
 * whose coordinates were scaled up from the padded coarse levels (src/Frame.cc:918,934-941; ORBextractor.cc:2075): a
 * level-7 keypoint near the bottom of a 480-row image sits at row ~ (134 + 18) * 3.58 = 545 > 518 = rows of level 0.
 * Built with -fsanitize=address: with -DORC_F6_UNCLAMPED (the reference's unchecked index) AddressSanitizer reports a
 * heap-buffer-overflow; without it (the restatement's clamp, which the HIP path shares) the run is clean.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orb_oracle.h"

int main(void) {
    enum { NL = 8 };
    const int W = 640, H = 480;
    float scale[NL], inv[NL];
    int pw[NL], ph[NL];
    uint8_t *pyrL[NL], *pyrR[NL];
    scale[0] = 1.f;
    for (int l = 1; l < NL; ++l) scale[l] = (float)((double)scale[l - 1] * (double)1.2f);
    for (int l = 0; l < NL; ++l) {
        inv[l] = 1.0f / scale[l];
        pw[l] = (int)((float)W * inv[l] + 0.5f) + 38;
        ph[l] = (int)((float)H * inv[l] + 0.5f) + 38;
        pyrL[l] = (uint8_t *)calloc((size_t)pw[l] * ph[l], 1);
        pyrR[l] = (uint8_t *)calloc((size_t)pw[l] * ph[l], 1);
    }
    /* one right keypoint of level 7 at the bottom of the padded level-7 image, in level-0 (scaled) coordinates */
    orc_keypoint kR, kL;
    memset(&kR, 0, sizeof(kR));
    kR.x = 100.f * scale[7];
    kR.y = (float)(ph[7] - 20) * scale[7];   /* 152 * 3.583 = 544.6: beyond ph[0] = 518 */
    kR.octave = 7; kR.class_id = -1;
    kL = kR;
    uint8_t dL[32] = {0}, dR[32] = {0};
    float uRight[1], depth[1];
    printf("rows of level 0 = %d, right keypoint row = %.1f (+- %.1f)\n", ph[0], kR.y, 2.0f * scale[7]);
    fflush(stdout);
    const int n = orc_stereo_matches(&kL, dL, 1, &kR, dR, 1, NL, scale, inv, (const uint8_t *const *)pyrL,
                                     (const uint8_t *const *)pyrR, pw, ph, 0.537f, 386.1448f, uRight, depth);
    printf("matches = %d (clean run)\n", n);
    for (int l = 0; l < NL; ++l) { free(pyrL[l]); free(pyrR[l]); }
    return 0;
}
