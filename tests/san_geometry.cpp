// Sanitizer driver for the HIP-free host translation unit csrc/orbx_geometry.cpp (tests/test_sanitizers.py builds it with
// g++ -fsanitize=address,undefined): tables + geometry for BASELINE.json's sizes and a sweep of odd sizes / scale factors,
// with the invariants the kernels rely on checked on the way (cells inside their level, taps inside the source level,
// slot ranges disjoint and inside the level's candidate region, FAST groups <= 64 + ORBX_FAST_XCOLS interior columns).
#include <cstdio>
#include <cstdlib>
#include "../orb_slam2_detailed_comments_amd/csrc/orbx_internal.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s (w=%d h=%d nf=%d sf=%.2f nl=%d)\n", #c, w, h, p.nfeatures, p.scale_factor, p.nlevels); ++fails; } } while (0)

static void one(int w, int h, int nf, float sf, int nl) {
    orbx_params p;
    p.nfeatures = nf; p.scale_factor = sf; p.nlevels = nl; p.ini_th_fast = 20; p.min_th_fast = 7;
    p.pyramid_mode = ORBX_PYRAMID_FORK_PADDED; p.fp_mode = ORBX_FP_GCC_FMA; p.device = -2; p.max_batch = 1; p.max_cand_per_cell = 0;
    OrbxTables t;
    orbx_build_tables(p, t);
    OrbxGeom g;
    const char *why = "";
    const orbx_status st = orbx_build_geometry(p, t, w, h, g, &why);
    if (st != ORBX_OK) return;   // BAD_ASPECT / UNSUPPORTED are legitimate answers
    long long sum = 0;
    for (int l = 0; l < nl; ++l) sum += t.nfeat[l];
    CHECK(sum >= nf || t.nfeat[nl - 1] == 0);
    for (int l = 0; l < nl; ++l) {
        const OrbxLevelGeom &L = g.lv[l];
        CHECK(L.pw == L.sw + 38 && L.ph == L.sh + 38 && L.pitch >= L.pw && L.pitch % 64 == 0);
        CHECK(L.off >= 0 && L.off + (long long)L.pitch * L.ph <= g.pyr_bytes);
        long long slots = 0;
        for (int c = L.cell_begin; c < L.cell_begin + L.cell_count; ++c) {
            const OrbxCell &C = g.cells[(size_t)c];
            CHECK(C.level == l && C.x0 >= 0 && C.y0 >= 0 && C.x0 + C.cw <= L.pw && C.y0 + C.ch <= L.ph && C.cw >= 7 && C.ch >= 7);
            CHECK(C.slot_begin == slots && C.slot_cap >= 1);
            slots += C.slot_cap;
        }
        CHECK(slots <= L.cand_cap);
        if (l > 0) {
            const OrbxLevelGeom &S = g.lv[l - 1];
            for (int x = 0; x < L.pw; ++x) { const OrbxTap &T = g.taps[(size_t)L.tapx_begin + x]; CHECK(T.s0 >= 0 && T.s1 < S.pw && T.s0 <= T.s1 && T.a0 + T.a1 == 2048); }
            for (int y = 0; y < L.ph; ++y) { const OrbxTap &T = g.taps[(size_t)L.tapy_begin + y]; CHECK(T.s0 >= 0 && T.s1 < S.ph && T.s0 <= T.s1 && T.a0 + T.a1 == 2048); }
        }
    }
    for (const OrbxFastGroup &G : g.fast_groups) {
        CHECK(G.ncell == 1 || G.ncell == 2);
        const OrbxCell &a = g.cells[(size_t)G.cell0], &b = g.cells[(size_t)G.cell0 + G.ncell - 1];
        CHECK(a.level == b.level && a.y0 == b.y0 && b.x0 + b.cw - a.x0 - 6 <= 64 + ORBX_FAST_XCOLS);
    }
}

int main() {
    one(640, 480, 1000, 1.2f, 8); one(1241, 376, 2000, 1.2f, 8); one(752, 480, 1200, 1.2f, 8); one(1920, 1080, 4000, 1.2f, 8);
    unsigned s = 12345;
    for (int i = 0; i < 400; ++i) {
        s = s * 1664525u + 1013904223u; const int w = 40 + (int)((s >> 8) % 1400);
        s = s * 1664525u + 1013904223u; const int h = 40 + (int)((s >> 8) % 900);
        s = s * 1664525u + 1013904223u; const int nf = 20 + (int)((s >> 8) % 5000);
        s = s * 1664525u + 1013904223u; const float sf = 1.05f + (float)((s >> 8) % 100) * 0.01f;
        s = s * 1664525u + 1013904223u; const int nl = 1 + (int)((s >> 8) % 12);
        one(w, h, nf, sf, nl);
    }
    std::printf("geometry sanitizer sweep: %d failures\n", fails);
    return fails != 0;
}
