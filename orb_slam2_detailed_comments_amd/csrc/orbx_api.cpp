// orbx_api.cpp -- the C ABI of liborbx.so (include/orbx.h): handle lifecycle, workspace management,
// stream / event plumbing and the launch sequence of one batched extraction.
//
// Launch sequence per batch (all on one HIP stream, no host synchronisation in the device entry point):
//   clear counters -> K0 border L0 -> K1 resize x (nlevels-1) -> K2 FAST cells -> K3 quadtree ->
//   K4 orientation -> K5 blur -> K6 descriptors + assembly.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "orbx_internal.h"
#include "orbx_launch.h"
#include "orbx_gate.h"

static thread_local std::string g_last_error;
static orbx_status fail(orbx_status s, const std::string &msg) {
    g_last_error = msg;
    return s;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(ORBX_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

struct ProfPair { hipEvent_t a, b; int kid; };

struct orbx_handle {
    orbx_params p;
    OrbxTables tab;
    OrbxGeom geom;
    DGeom dg;
    bool host_only = false, configured = false;
    int dev = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // side stream of the batched extraction (high priority): the small pyramid levels are a chain of short, latency-bound
    // launches; they run here, next to the issue-bound FAST kernel of the large levels on the main stream
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_stereo = nullptr;   // orders the batched stereo match with the OTHER eye's stream (two extractors, two streams)
    int fork_level = 0;       // first level whose resize + FAST run on the side stream (0 = no fork)
    int fork_group = 0;       // first FAST group of that level
    // geometry-dependent device state
    uint8_t *d_pyr = nullptr, *d_blur = nullptr;
    OrbxCell *d_cells = nullptr;
    OrbxFastGroup *d_groups = nullptr;
    bool resize_legacy = false;   // ORBX_RESIZE_IMPL=legacy: k_pyr_resize for every level (A/B runs)
    int match_kernel = 0;         // ORBX_MATCH_KERNEL, read ONCE when the handle is created: 0 = matrix pipe with FP4 operands (default), 1 = "valu" (vector pipe), 2 = "i8" (matrix pipe, int8 operands) -- A/B runs, parity tests
    int fast_stop = 0;      // ORBX_FAST_STOP: only read in -DORBX_TIMING_KNOBS builds
    int fast_lcap = 640;    // LDS work-list entries of k_fast_rows (ORBX_FAST_LCAP; tests shrink it to force the flush paths)
    OrbxTap *d_taps = nullptr;
    uint2 *d_dense = nullptr;   // dense per-(frame, level) key arrays, appended to by k_fast_rows (fill counts in d_cand_count)
    int *d_cand_count = nullptr, *d_lvl_count = nullptr, *d_status = nullptr;
    uint32_t *d_lvl_kp = nullptr;
    float *d_lvl_angle = nullptr;
    uint16_t *d_knode = nullptr;
    int max_cw = 0, max_ch = 0, ncap = 0, lds_keys = 0, lds_keys_few = 0;
    // staging for the host entry points
    uint8_t *st_in[2] = {nullptr, nullptr}; size_t d_in_bytes = 0;
    orbx_keypoint *st_kps[2] = {nullptr, nullptr}; uint8_t *st_desc[2] = {nullptr, nullptr}; int *st_cnt[2] = {nullptr, nullptr};
    int out_cap = 0, stage_chunk = 0;
    int *pin_stat = nullptr; size_t pin_stat_ints = 0;   // page-locked landing place of the per-frame counts | status words of a call (grow-only)
    hipStream_t s_in = nullptr;   // the ONE copy stream of the pipelined host-buffer call (uploads and downloads in turn)
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    int last_batch = 0;
    int mk_w = 0, mk_h = 0, mk_total = 0;   // orbx_max_keypoints cache
    void *d_match_ws = nullptr; size_t match_ws_bytes = 0;   // partial (best, second) keys of k_match
    uint32_t *d_gate_items = nullptr; size_t gate_items_cap = 0;   // candidate lists of k_gate / distance blocks of k_block_dist (grow-only)
    uint8_t *pin = nullptr; size_t pin_bytes = 0;                  // page-locked staging of the host-buffer policy entry points (grow-only)
    size_t gate_guess = 4096;                                      // entries the speculative download of the candidate lists covers
    // grow-only scratch arena for the host-buffer convenience entry points (match / matrix / stereo): no hipMalloc
    // on the steady-state path and nothing to leak on an error return
    uint8_t *d_scratch = nullptr; size_t scratch_bytes = 0, scratch_used = 0;
    uint2 *d_rect = nullptr; int rect_w = 0, rect_h = 0;   // pre-digested rectification maps (orbx_set_rectification)
    int input_format = ORBX_FMT_GRAY8;        // pixel format of the frames handed to the extract entry points
    bool blur_valid = false;                // d_blur holds the blurred pyramid of the last batch
    // profiling
    uint32_t prof_mask = 0;
    std::vector<ProfPair> pending;
    std::vector<hipEvent_t> pool;
    float ms[ORBX_K_COUNT] = {0};
    int launches[ORBX_K_COUNT] = {0};
};

// shared with orbx_policies.cpp
orbx_status orbx_fail(orbx_status s, const std::string &msg) { return fail(s, msg); }
int orbx_handle_fp_mode(const orbx_handle *h) { return h->p.fp_mode; }

static const char *k_names[ORBX_K_COUNT] = {"k_pyr_l0", "k_pyr_resize", "k_fast_rows", "k_quadtree", "k_orient",
                                            "k_blur",   "k_describe",   "k_match",      "misc"};

extern "C" const char *orbx_kernel_name(int k) { return (k >= 0 && k < ORBX_K_COUNT) ? k_names[k] : "?"; }
extern "C" const char *orbx_last_error(void) { return g_last_error.c_str(); }
extern "C" int orbx_abi_version(void) { return ORBX_ABI_VERSION; }
extern "C" const char *orbx_status_string(orbx_status s) {
    switch (s) {
        case ORBX_OK: return "ok";
        case ORBX_EMPTY_IMAGE: return "empty image";
        case ORBX_BAD_ARGUMENT: return "bad argument";
        case ORBX_BAD_ASPECT: return "bad aspect ratio (nIni == 0)";
        case ORBX_CAPACITY: return "capacity exceeded";
        case ORBX_HIP_ERROR: return "HIP error";
        case ORBX_NO_DEVICE: return "no HIP device";
        case ORBX_UNSUPPORTED: return "unsupported geometry";
    }
    return "?";
}

extern "C" void orbx_default_params(orbx_params *p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->nfeatures = 1000; p->scale_factor = 1.2f; p->nlevels = 8; p->ini_th_fast = 20; p->min_th_fast = 7;
    p->pyramid_mode = ORBX_PYRAMID_FORK_PADDED; p->fp_mode = ORBX_FP_GCC_FMA;
    p->device = -1; p->max_batch = 1; p->max_cand_per_cell = 0;
}

// ---------------------------------------------------------------- profiling helpers
// roctx ranges around every stage's launches: ORBX_ROCTX=1 loads the roctx library at the first use and
// brackets each ProfScope with roctxRangePushA(slot name) / roctxRangePop, so `rocprofv3 --marker-trace` groups the
// kernels of a call by stage.  Not linked: without the variable the library has no dependency on the profiler.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = getenv("ORBX_ROCTX");
        if (!e || !atoi(e)) return;
        void *lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);   // the one rocprofv3 --marker-trace records
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);                 // roctracer's (older tools)
        if (!lib) return;
        push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
        pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
static const Roctx &roctx() { static Roctx r; return r; }

struct ProfScope {
    orbx_handle *h; int kid; bool on; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool marked = false;
    ProfScope(orbx_handle *h_, int kid_, hipStream_t st_ = nullptr)
        : h(h_), kid(kid_), on(((h_->prof_mask >> kid_) & 1u) != 0), st(st_ ? st_ : h_->stream) {
        if (roctx().push) { roctx().push(orbx_kernel_name(kid)); marked = true; }
        if (!on) return;
        a = grab(); b = grab();
        hipEventRecord(a, st);   // events are recorded on the stream the kernel is launched on
    }
    hipEvent_t grab() {
        if (!h->pool.empty()) { hipEvent_t e = h->pool.back(); h->pool.pop_back(); return e; }
        hipEvent_t e; hipEventCreate(&e); return e;
    }
    ~ProfScope() {
        if (marked) roctx().pop();
        if (!on) return;
        hipEventRecord(b, st);
        h->pending.push_back({a, b, kid});
    }
};

static void prof_drain(orbx_handle *h) {
    if (h->pending.empty()) return;
    hipStreamSynchronize(h->stream);
    for (auto &pp : h->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pp.a, pp.b) == hipSuccess) { h->ms[pp.kid] += ms; h->launches[pp.kid]++; }
        h->pool.push_back(pp.a); h->pool.push_back(pp.b);
    }
    h->pending.clear();
}

static orbx_status pin_reserve(orbx_handle *h, size_t bytes);   // page-locked host staging (defined with the policy plumbing)

// ---------------------------------------------------------------- workspace
static void free_geometry_buffers(orbx_handle *h) {
    hipFree(h->d_groups); h->d_groups = nullptr;
    hipFree(h->d_pyr); hipFree(h->d_blur); hipFree(h->d_cells); hipFree(h->d_taps);
    hipFree(h->d_dense); h->d_dense = nullptr;
    hipFree(h->d_cand_count); hipFree(h->d_lvl_count); hipFree(h->d_status); hipFree(h->d_lvl_kp);
    hipFree(h->d_lvl_angle); hipFree(h->d_knode);
    h->d_pyr = h->d_blur = nullptr; h->d_cells = nullptr; h->d_taps = nullptr;
    h->d_cand_count = h->d_lvl_count = h->d_status = nullptr; h->d_lvl_kp = nullptr; h->d_lvl_angle = nullptr;
    h->d_knode = nullptr;
    h->configured = false;
}

static orbx_status configure(orbx_handle *h, int width, int height) {
    if (h->configured && h->geom.width == width && h->geom.height == height) return ORBX_OK;
    if (h->host_only) return fail(ORBX_NO_DEVICE, "host-only handle (device = -2) cannot extract");
    HIPCHK(hipSetDevice(h->dev));
    if (h->configured) {   // nothing queued on either of the handle's streams may still read the buffers that are about to go
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->side_stream) HIPCHK(hipStreamSynchronize(h->side_stream));
        free_geometry_buffers(h);
    }
    const char *why = "";
    OrbxGeom g;
    orbx_status st = orbx_build_geometry(h->p, h->tab, width, height, g, &why);
    if (st != ORBX_OK) return fail(st, why);
    h->geom = g;
    const int B = h->p.max_batch, NL = h->p.nlevels;
    // device geometry block
    DGeom &d = h->dg;
    memset(&d, 0, sizeof(d));
    d.nlevels = NL; d.ncells = (int)g.cells.size(); d.kp_total = g.kp_total; d.fp_mode = h->p.fp_mode;
    d.ini_th = std::min(std::max(h->p.ini_th_fast, 0), 255);
    d.min_th = std::min(std::max(h->p.min_th_fast, 0), 255);
    d.pyr_bytes = g.pyr_bytes; d.cand_total = g.cand_total;
    for (int i = 0; i < 16; ++i) d.umax[i] = h->tab.umax[i];
    int tiles = 0;
    h->max_cw = h->max_ch = 7;
    for (const auto &c : g.cells) { h->max_cw = std::max<int>(h->max_cw, c.cw); h->max_ch = std::max<int>(h->max_ch, c.ch); }
    for (int l = 0; l < NL; ++l) {
        const OrbxLevelGeom &L = g.lv[l];
        DLevel &D = d.lv[l];
        D.pw = L.pw; D.ph = L.ph; D.pitch = L.pitch; D.sw = L.sw; D.sh = L.sh;
        D.cell_begin = L.cell_begin; D.cell_count = L.cell_count;
        D.qt_w = L.qt_w; D.qt_h = L.qt_h; D.nini = L.nini; D.hx = L.hx;
        D.nfeat = L.nfeat; D.kp_cap = L.kp_cap; D.kp_begin = L.kp_begin; D.cand_cap = L.cand_cap;
        D.tapx = L.tapx_begin; D.tapy = L.tapy_begin; D.scale = L.scale; D.size = L.size;
        D.off = L.off; D.cand_begin = L.cand_begin;
        D.blur_tx = (L.pw + 127) / 128;  // k_blur tile = 128 x 32
        D.blur_tile_begin = tiles;
        tiles += D.blur_tx * ((L.ph + 31) / 32);
    }
    d.blur_tiles = tiles;
    // quadtree LDS plan: node tables always in LDS, key->node map in LDS when the level's candidates fit
    h->ncap = g.node_cap;
    // Keys of a level (position + node index, 6 bytes each) live in LDS when they fit `lds_keys` slots; denser levels fall
    // back to the global scratch map (same code path through a generic pointer).  The kernel is latency-bound (a chain of
    // barrier-separated stages), so what counts is how many workgroups a CU holds: take the largest residency (4 = the wave
    // limit of 512-thread workgroups, then 3, 2, 1) whose key slots still cover ~6 candidates per kept keypoint of the
    // densest level (level 0 of the 640x480 bench frames has 1230 candidates for 217 keypoints), capped at 4096.
    const size_t node_part = orbx_quadtree_smem(h->ncap, 0);
    if (node_part > 120 * 1024) return fail(ORBX_UNSUPPORTED, "nfeatures too large for the LDS quadtree node table");
    {
        const size_t want = std::min<size_t>({(size_t)4096, (size_t)g.max_cand_cap, (size_t)6 * (size_t)std::max(g.lv[0].nfeat, 1)});
        size_t keys = 0;
        auto fit_at = [&](int wg) -> size_t {
            const size_t budget = (size_t)(160 * 1024) / wg - 512;   // allocation granularity margin
            return budget <= node_part ? 0 : std::min<size_t>({(budget - node_part) / 6, (size_t)4096, (size_t)g.max_cand_cap});
        };
        // ... and no more than that: key slots beyond `want` only lower the number of workgroups a CU can hold (1024-frame
        // batches, 640x480 / 1000 features: 2794 slots = 4 workgroups per CU 250 us, 1400 slots 204 us, 600 slots 208 us).
        // (Launching the small levels on their own with a node table and key slots sized for them -- more workgroups per CU
        // still -- measured 215-220 us against 200, and 277 against 259 us at 1920x1080 / 4000 features where the common plan
        // admits one workgroup per CU: the large levels set the time, and the second launch boundary is pure cost.)
        // Small launches (a few workgroups per CU at most: the single-frame call) have nothing to gain from residency and take
        // the plan that fills the budget, so that the densest level keeps its keys in LDS too (lds_keys_few).
        size_t keys_few = 0;
        for (int wg = 4; wg >= 2 && keys == 0; --wg)
            if (fit_at(wg) >= want) { keys = std::max<size_t>(want, 1024); keys_few = fit_at(wg); }
        h->lds_keys_few = (int)keys_few;
        // large nfeatures (node tables of tens of KB): two workgroups per CU with the dense levels on the global key map beat
        // one workgroup with every level in LDS -- those levels exceed any LDS budget anyway
        if (keys == 0 && fit_at(2) >= 1024) keys = fit_at(2);
        if (keys == 0) keys = fit_at(1);
        if (const char *e = getenv("ORBX_QT_LDS_KEYS")) keys = std::min<size_t>((size_t)std::max(atoi(e), 0), std::min<size_t>(8192, (160 * 1024 - node_part) / 6));
        h->lds_keys = (int)keys;
        if (getenv("ORBX_QT_LDS_KEYS") || h->lds_keys_few < h->lds_keys) h->lds_keys_few = h->lds_keys;
    }
    HIPCHK(orbx_quadtree_prepare(orbx_quadtree_smem(h->ncap, std::max(h->lds_keys, h->lds_keys_few))));
    // buffers.  Every fill / upload below is issued on the handle's stream: the kernels that read them are launched on
    // the same stream, so the order holds by construction (hipMemset on the NULL stream is asynchronous and NOT ordered
    // with a non-blocking stream -- the round-1 race -- and a device-wide barrier would stall every other handle).
    // Source vectors live in h->geom (they outlive the copies).
    if (const char *e = getenv("ORBX_RESIZE_IMPL")) h->resize_legacy = strcmp(e, "legacy") == 0;
#ifdef ORBX_TIMING_KNOBS   // phase-timing builds only (tools/build_variant.sh): stops k_fast_rows early, results are wrong
    if (const char *e = getenv("ORBX_FAST_STOP")) h->fast_stop = atoi(e);
#endif
    if (const char *e = getenv("ORBX_FAST_LCAP")) h->fast_lcap = std::max(64, atoi(e));
    const OrbxGeom &hg = h->geom;
    hipStream_t s = h->stream;
    // Optional fork of the batched launch sequence (run_chunk): ORBX_FORK_LEVEL = l > 0 resizes levels >= l and runs their
    // FAST groups on the side stream next to the FAST kernel of the large levels (+1..3 % frames/s at l = 3 or 4 for 256-frame
    // batches).  Off by default: the gain is small and kernels that share the chip have durations that no longer describe
    // the kernel alone.
    h->fork_level = 0;
    if (const char *e = getenv("ORBX_FORK_LEVEL")) h->fork_level = std::min(std::max(atoi(e), 0), NL - 1);
    h->fork_group = 0;
    if (h->fork_level > 0) {
        while (h->fork_group < (int)hg.fast_groups.size() && hg.cells[(size_t)hg.fast_groups[(size_t)h->fork_group].cell0].level < h->fork_level)
            ++h->fork_group;
        if (h->fork_group == 0 || h->fork_group >= (int)hg.fast_groups.size()) h->fork_level = 0;
    }
    auto setup = [&]() -> hipError_t {
        hipError_t e;
#define ORBX_TRY(expr) do { e = (expr); if (e != hipSuccess) return e; } while (0)
        ORBX_TRY(hipMalloc(&h->d_pyr, (size_t)B * hg.pyr_bytes + 256));   // +256: kernels read whole aligned dwords
        ORBX_TRY(hipMalloc(&h->d_cells, std::max<size_t>(1, hg.cells.size()) * sizeof(OrbxCell)));
        ORBX_TRY(hipMalloc(&h->d_taps, std::max<size_t>(1, hg.taps.size()) * sizeof(OrbxTap)));
        ORBX_TRY(hipMalloc(&h->d_dense, (size_t)B * hg.cand_total * sizeof(uint2)));
        ORBX_TRY(hipMalloc(&h->d_knode, (size_t)B * hg.cand_total * sizeof(uint16_t)));
        ORBX_TRY(hipMalloc(&h->d_cand_count, (size_t)B * NL * sizeof(int)));
        ORBX_TRY(hipMalloc(&h->d_lvl_count, (size_t)B * NL * sizeof(int)));
        ORBX_TRY(hipMalloc(&h->d_status, (size_t)B * sizeof(int)));
        ORBX_TRY(hipMalloc(&h->d_lvl_kp, (size_t)B * hg.kp_total * sizeof(uint32_t)));
        ORBX_TRY(hipMalloc(&h->d_lvl_angle, (size_t)B * hg.kp_total * sizeof(float)));
        ORBX_TRY(hipMalloc(&h->d_groups, std::max<size_t>(1, hg.fast_groups.size()) * sizeof(OrbxFastGroup)));
        if (!hg.cells.empty()) {
            ORBX_TRY(hipMemcpyAsync(h->d_cells, hg.cells.data(), hg.cells.size() * sizeof(OrbxCell), hipMemcpyHostToDevice, s));
            ORBX_TRY(hipMemcpyAsync(h->d_groups, hg.fast_groups.data(), hg.fast_groups.size() * sizeof(OrbxFastGroup),
                                    hipMemcpyHostToDevice, s));
        }
        if (!hg.taps.empty())
            ORBX_TRY(hipMemcpyAsync(h->d_taps, hg.taps.data(), hg.taps.size() * sizeof(OrbxTap), hipMemcpyHostToDevice, s));
        ORBX_TRY(hipMemsetAsync(h->d_pyr, 0, (size_t)B * hg.pyr_bytes + 256, s));
        ORBX_TRY(hipMemsetAsync(h->d_lvl_count, 0, (size_t)B * NL * sizeof(int), s));
        ORBX_TRY(hipMemsetAsync(h->d_cand_count, 0, (size_t)B * NL * sizeof(int), s));
#undef ORBX_TRY
        return hipSuccess;
    };
    const hipError_t se = setup();
    if (se != hipSuccess) {   // nothing half-allocated survives a failed configure()
        hipStreamSynchronize(s);
        free_geometry_buffers(h);
        return fail(ORBX_HIP_ERROR, std::string("configure: ") + hipGetErrorString(se));
    }
    h->configured = true;
    h->last_batch = 0;
    return ORBX_OK;
}

extern "C" orbx_status orbx_create(const orbx_params *params, orbx_handle **out) {
    if (!params || !out) return fail(ORBX_BAD_ARGUMENT, "null argument");
    *out = nullptr;
    if (params->nlevels < 1 || params->nlevels > ORBX_MAX_LEVELS) return fail(ORBX_BAD_ARGUMENT, "nlevels out of range [1,16]");
    if (params->nfeatures < 1 || params->nfeatures > 16000) return fail(ORBX_BAD_ARGUMENT, "nfeatures out of range [1,16000]");
    if (!(params->scale_factor > 1.0f)) return fail(ORBX_BAD_ARGUMENT, "scale_factor must be > 1");
    if (params->pyramid_mode != ORBX_PYRAMID_FORK_PADDED) return fail(ORBX_UNSUPPORTED, "only ORBX_PYRAMID_FORK_PADDED is implemented");
    if (params->fp_mode != ORBX_FP_GCC_FMA && params->fp_mode != ORBX_FP_STRICT) return fail(ORBX_BAD_ARGUMENT, "fp_mode");
    orbx_handle *h = new orbx_handle();
    h->p = *params;
    if (h->p.max_batch < 1) h->p.max_batch = 1;
    if (const char *e = getenv("ORBX_MATCH_KERNEL")) h->match_kernel = strcmp(e, "valu") == 0 ? 1 : strcmp(e, "i8") == 0 ? 2 : 0;
    orbx_build_tables(h->p, h->tab);
    if (params->device == -2) {  // host-only handle: tables and getters, no device work
        h->host_only = true;
        *out = h;
        return ORBX_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        delete h;
        return fail(ORBX_NO_DEVICE, "no HIP device visible: the ORB front-end has no CPU fallback");
    }
    int dev = params->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete h; return fail(ORBX_BAD_ARGUMENT, "device ordinal out of range"); }
    h->dev = dev;
    hipError_t e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = orbx_upload_pattern();
    if (e == hipSuccess) {
        int lo = 0, hi = 0;
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically lowest = highest priority
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, hi);
        // device-scope release: the two streams exchange device memory only (a system-scope fence per record costs ~10 us)
        unsigned evflags = hipEventDisableTiming | hipEventReleaseToDevice;
        if (const char *ef = getenv("ORBX_EVENT_FLAGS")) evflags = (unsigned)strtoul(ef, nullptr, 0);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, evflags);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, evflags);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_stereo, evflags);
    }
    if (e != hipSuccess) { orbx_destroy(h); return fail(ORBX_HIP_ERROR, hipGetErrorString(e)); }
    h->stream = h->own_stream;
    *out = h;
    return ORBX_OK;
}

extern "C" void orbx_destroy(orbx_handle *h) {
    if (!h) return;
    if (!h->host_only) {
        hipSetDevice(h->dev);
        hipStreamSynchronize(h->stream);
        prof_drain(h);
        for (auto e : h->pool) hipEventDestroy(e);
        free_geometry_buffers(h);
        hipFree(h->d_match_ws); hipFree(h->d_scratch); hipFree(h->d_rect); hipFree(h->d_gate_items);
        if (h->pin) hipHostFree(h->pin);
        if (h->pin_stat) hipHostFree(h->pin_stat);
        for (int s = 0; s < 2; ++s) {
            hipFree(h->st_in[s]); hipFree(h->st_kps[s]);   // st_desc / st_cnt live inside the st_kps allocation
            if (h->ev_in[s]) hipEventDestroy(h->ev_in[s]);
            if (h->ev_done[s]) hipEventDestroy(h->ev_done[s]);
            if (h->ev_out[s]) hipEventDestroy(h->ev_out[s]);
        }
        if (h->s_in) hipStreamDestroy(h->s_in);
        if (h->side_stream) { hipStreamSynchronize(h->side_stream); hipStreamDestroy(h->side_stream); }
        if (h->ev_fork) hipEventDestroy(h->ev_fork);
        if (h->ev_join) hipEventDestroy(h->ev_join);
        if (h->ev_stereo) hipEventDestroy(h->ev_stereo);
        if (h->own_stream) hipStreamDestroy(h->own_stream);
    }
    delete h;
}

// ---------------------------------------------------------------- getters
extern "C" int orbx_get_levels(const orbx_handle *h) { return h ? h->p.nlevels : 0; }
extern "C" float orbx_get_scale_factor(const orbx_handle *h) { return h ? (float)(double)h->p.scale_factor : 0.f; }
extern "C" orbx_status orbx_get_scale_tables(const orbx_handle *h, float *scale, float *inv_scale, float *sigma2,
                                             float *inv_sigma2) {
    if (!h) return fail(ORBX_BAD_ARGUMENT, "null handle");
    const int n = h->p.nlevels;
    if (scale) memcpy(scale, h->tab.scale, n * sizeof(float));
    if (inv_scale) memcpy(inv_scale, h->tab.inv_scale, n * sizeof(float));
    if (sigma2) memcpy(sigma2, h->tab.sigma2, n * sizeof(float));
    if (inv_sigma2) memcpy(inv_sigma2, h->tab.inv_sigma2, n * sizeof(float));
    return ORBX_OK;
}
extern "C" orbx_status orbx_get_features_per_level(const orbx_handle *h, int32_t *n) {
    if (!h || !n) return fail(ORBX_BAD_ARGUMENT, "null argument");
    memcpy(n, h->tab.nfeat, h->p.nlevels * sizeof(int32_t));
    return ORBX_OK;
}
extern "C" orbx_status orbx_get_umax(const orbx_handle *h, int32_t *umax16) {
    if (!h || !umax16) return fail(ORBX_BAD_ARGUMENT, "null argument");
    memcpy(umax16, h->tab.umax, 16 * sizeof(int32_t));
    return ORBX_OK;
}
extern "C" int orbx_max_keypoints(orbx_handle *h, int width, int height) {
    if (!h || width <= 0 || height <= 0) return -(int)ORBX_BAD_ARGUMENT;
    if (h->configured && h->geom.width == width && h->geom.height == height) return h->geom.kp_total;
    if (h->mk_w == width && h->mk_h == height) return h->mk_total;
    OrbxGeom g; const char *why = "";
    const orbx_status st = orbx_build_geometry(h->p, h->tab, width, height, g, &why);
    if (st != ORBX_OK) { g_last_error = why; return -(int)st; }
    h->mk_w = width; h->mk_h = height; h->mk_total = g.kp_total;
    return g.kp_total;
}

// ---------------------------------------------------------------- extraction
static inline int orbx_fmt_channels(int fmt) { return fmt == ORBX_FMT_GRAY8 ? 1 : (fmt == ORBX_FMT_RGB8 || fmt == ORBX_FMT_BGR8) ? 3 : 4; }
// cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) of Examples/Stereo/stereo_euroc.cc:183-194 in front of the extractor:
// the float maps are digested once (cvRound(map * 32), split into integer part and 5-bit fractions, as remap() does per
// block) and level 0 samples the raw image through them.  NULL maps switch the rectification off.
extern "C" orbx_status orbx_set_rectification(orbx_handle *h, const float *map_x, const float *map_y, int width, int height) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(h->d_rect); h->d_rect = nullptr; h->rect_w = h->rect_h = 0;
    if (!map_x && !map_y) return ORBX_OK;
    if (!map_x || !map_y || width <= 0 || height <= 0) return fail(ORBX_BAD_ARGUMENT, "bad rectification maps");
    if (h->input_format != ORBX_FMT_GRAY8) return fail(ORBX_BAD_ARGUMENT, "rectification needs 8-bit gray input");
    std::vector<uint2> t((size_t)width * height);
    for (size_t i = 0; i < t.size(); ++i) {
        const int sx = (int)lrintf(map_x[i] * 32.f), sy = (int)lrintf(map_y[i] * 32.f);   // cvRound: half to even
        const int ix = std::min(std::max(sx >> 5, -32768), 32767), iy = std::min(std::max(sy >> 5, -32768), 32767);
        t[i].x = (uint32_t)(uint16_t)(int16_t)ix | ((uint32_t)(uint16_t)(int16_t)iy << 16);
        t[i].y = (uint32_t)(((sy & 31) << 5) | (sx & 31));
    }
    HIPCHK(hipMalloc(&h->d_rect, t.size() * sizeof(uint2)));
    // on the handle's stream (ordered with the kernels that read the maps); `t` is a local: wait for the copy, on this stream only
    HIPCHK(hipMemcpyAsync(h->d_rect, t.data(), t.size() * sizeof(uint2), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->rect_w = width; h->rect_h = height;
    return ORBX_OK;
}
extern "C" orbx_status orbx_set_input_format(orbx_handle *h, int pixel_format) {
    if (!h) return fail(ORBX_BAD_ARGUMENT, "null handle");
    if (pixel_format < ORBX_FMT_GRAY8 || pixel_format > ORBX_FMT_BGRA8) return fail(ORBX_BAD_ARGUMENT, "unknown pixel format");
    if (h->d_rect && pixel_format != ORBX_FMT_GRAY8) return fail(ORBX_BAD_ARGUMENT, "rectification needs 8-bit gray input");
    h->input_format = pixel_format;
    return ORBX_OK;
}
static orbx_status run_chunk(orbx_handle *h, int B, const uint8_t *d_imgs, int W, int H, int stride,
                             int64_t frame_stride, orbx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts,
                             int32_t *d_status, int cap) {
    const DGeom &g = h->dg;
    hipStream_t s = h->stream;
    const int NL = g.nlevels;
    if (h->d_rect && (h->rect_w != W || h->rect_h != H))
        return fail(ORBX_BAD_ARGUMENT, "rectification maps were set for another image size (raw and rectified size must agree)");
    // (no clearing launch: the per-frame status word is reset by the level-0 kernel, the per-level counters are written
    // unconditionally by k_quadtree)
    // (A two-stream level pipeline -- FAST of level l on a low-priority stream while the main stream resizes level
    // l+1 -- was measured and rejected: 81 k frames/s against 116 k for this single in-order sequence; the cross-stream
    // event waits and the 8 small FAST launches cost more than the overlap recovers.)
    { ProfScope ps(h, ORBX_K_PYR_L0);
      if (h->d_rect) {   // cv::remap of the EuRoC rectification fused into level 0
          orbx_launch_pyr_l0_remap(s, g, B, d_imgs, W, H, stride, frame_stride, h->d_pyr, h->d_rect, d_status, h->d_cand_count);
      } else if (h->input_format == ORBX_FMT_GRAY8) {
          orbx_launch_pyr_l0(s, g, B, d_imgs, W, H, stride, frame_stride, h->d_pyr, d_status, h->d_cand_count);
      } else {   // cvtColor of Tracking::GrabImage* fused into level 0
          const int nch = (h->input_format == ORBX_FMT_RGB8 || h->input_format == ORBX_FMT_BGR8) ? 3 : 4;
          const bool rgb = h->input_format == ORBX_FMT_RGB8 || h->input_format == ORBX_FMT_RGBA8;
          orbx_launch_pyr_l0_color(s, g, B, d_imgs, W, H, stride, frame_stride, h->d_pyr, nch, rgb ? 0 : 2, rgb ? 2 : 0, d_status, h->d_cand_count);
      } }
    // Large batches fork after level fork_level - 1: the remaining (small) levels are resized on the high-priority side stream
    // -- seven dependent launches of which the last four have few waves and are pure latency -- followed by their FAST
    // groups, while the main stream runs the issue-bound FAST kernel of the large levels; joined before the quadtree.
    const int ngroups = (int)h->geom.fast_groups.size();
    const bool fork = h->fork_level > 0 && (long long)B * ngroups >= 16384;
    const int l_main_end = fork ? h->fork_level : NL;
    for (int l = 1; l < l_main_end; ++l) {
        ProfScope ps(h, ORBX_K_PYR_RESIZE);
        orbx_launch_pyr_resize(s, g, B, l, h->d_taps, h->d_pyr, h->geom.lv[l].narrow_taps && !h->resize_legacy);
    }
    if (fork) {
        hipStream_t s2 = h->side_stream;
        if (hipEventRecord(h->ev_fork, s) != hipSuccess || hipStreamWaitEvent(s2, h->ev_fork, 0) != hipSuccess)
            return fail(ORBX_HIP_ERROR, "fork event");
        for (int l = h->fork_level; l < NL; ++l) {
            ProfScope ps(h, ORBX_K_PYR_RESIZE, s2);
            orbx_launch_pyr_resize(s2, g, B, l, h->d_taps, h->d_pyr, h->geom.lv[l].narrow_taps && !h->resize_legacy);
        }
        { ProfScope ps(h, ORBX_K_FAST, s2);
          orbx_launch_fast_rows(s2, g, B, h->d_cells, h->d_groups + h->fork_group, ngroups - h->fork_group, h->d_pyr, h->d_dense,
                                h->d_cand_count, d_status, h->max_ch, h->fast_lcap, h->fast_stop); }
        // (The quadtree of the small levels was tried on the side stream behind its FAST groups: it needs CU residency the
        // FAST kernel of the large levels does not give up -- 201 us for 1024 workgroups that take 57 us alone -- and delays the
        // join.  It runs after the join.)
        { ProfScope ps(h, ORBX_K_FAST);
          orbx_launch_fast_rows(s, g, B, h->d_cells, h->d_groups, h->fork_group, h->d_pyr, h->d_dense,
                                h->d_cand_count, d_status, h->max_ch, h->fast_lcap, h->fast_stop); }
        if (hipEventRecord(h->ev_join, s2) != hipSuccess || hipStreamWaitEvent(s, h->ev_join, 0) != hipSuccess)
            { hipStreamSynchronize(s2); return fail(ORBX_HIP_ERROR, "join event"); }   // the side stream's work reads the handle's buffers: never leave it unjoined
    } else {
        ProfScope ps(h, ORBX_K_FAST);
        orbx_launch_fast_rows(s, g, B, h->d_cells, h->d_groups, ngroups, h->d_pyr, h->d_dense,
                              h->d_cand_count, d_status, h->max_ch, h->fast_lcap, h->fast_stop);
    }
    { ProfScope ps(h, ORBX_K_QUADTREE);
      orbx_launch_quadtree(s, g, B, h->d_dense, h->d_cand_count, h->d_lvl_kp,
                           h->d_lvl_count, d_status, h->d_knode,
                           h->ncap, (long long)B * NL >= 1024 ? h->lds_keys : h->lds_keys_few, 0, NL); }
    // orientation (IC_Angle) is computed inside k_describe from the same LDS patch the descriptor uses
    h->blur_valid = false;  // the Gaussian is fused into k_describe; the full blurred image is only built on request
    { ProfScope ps(h, ORBX_K_DESC);
      orbx_launch_describe(s, g, B, h->d_pyr, h->d_lvl_kp, h->d_lvl_count, h->d_lvl_angle, d_kps, d_desc, d_counts,
                           d_status, cap); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, std::string("kernel launch: ") + hipGetErrorString(e));
    h->last_batch = B;
    return ORBX_OK;
}

extern "C" orbx_status orbx_extract_batch_device(orbx_handle *h, int nframes, const uint8_t *d_imgs, int width,
                                                 int height, int stride, int64_t frame_stride, orbx_keypoint *d_kps,
                                                 uint8_t *d_desc, int32_t *d_counts, int32_t *d_status, int cap) {
    if (!h) return fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!d_imgs || width <= 0 || height <= 0 || nframes <= 0) return fail(ORBX_EMPTY_IMAGE, "empty image");
    if (!d_kps || !d_desc || !d_counts || cap <= 0 || stride < width * orbx_fmt_channels(h->input_format)) return fail(ORBX_BAD_ARGUMENT, "bad output buffers / stride");
    orbx_status st = configure(h, width, height);
    if (st != ORBX_OK) return st;
    HIPCHK(hipSetDevice(h->dev));
    const int MB = h->p.max_batch;
    for (int f0 = 0; f0 < nframes; f0 += MB) {
        const int B = std::min(MB, nframes - f0);
        int32_t *stp = d_status ? d_status + f0 : h->d_status;
        st = run_chunk(h, B, d_imgs + (int64_t)f0 * frame_stride, width, height, stride, frame_stride,
                       d_kps + (int64_t)f0 * cap, d_desc + (int64_t)f0 * cap * 32, d_counts + f0, stp, cap);
        if (st != ORBX_OK) return st;
    }
    return ORBX_OK;
}

// Host-buffer entry point: two sets of device staging buffers and two copy streams, so that the upload of chunk c+1 and
// the download of chunk c-1 overlap the kernels of chunk c.  With pageable host memory hipMemcpyAsync stages through the
// runtime's pinned buffers and blocks the calling thread while it does (the GPU keeps computing meanwhile); with pinned
// memory (orbx_host_alloc, or any hipHostMalloc / hipHostRegister memory) every copy is a plain asynchronous DMA.
static orbx_status ensure_staging(orbx_handle *h, size_t in_bytes, int cap, int chunk) {
    if (in_bytes > h->d_in_bytes || cap > h->out_cap || chunk > h->stage_chunk) {
        HIPCHK(hipStreamSynchronize(h->stream));
        const size_t inb = std::max(in_bytes, h->d_in_bytes);
        const int c = std::max(cap, h->out_cap), ch = std::max(chunk, h->stage_chunk);
        for (int s = 0; s < 2; ++s) {
            hipFree(h->st_in[s]); hipFree(h->st_kps[s]);   // st_desc / st_cnt live inside the st_kps allocation
            h->st_in[s] = nullptr; h->st_kps[s] = nullptr; h->st_desc[s] = nullptr; h->st_cnt[s] = nullptr;
        }
        h->d_in_bytes = 0; h->out_cap = 0; h->stage_chunk = 0;
        for (int s = 0; s < 2; ++s) {
            HIPCHK(hipMalloc(&h->st_in[s], inb));
            // one block per set: keypoints | descriptors | counts | status -- a call that fills the block exactly (the
            // single-frame drop-in call) downloads it with ONE copy
            const size_t kb = (size_t)ch * c * sizeof(orbx_keypoint), db = (size_t)ch * c * 32;
            uint8_t *blk = nullptr;
            HIPCHK(hipMalloc(&blk, kb + db + (size_t)2 * ch * sizeof(int)));
            h->st_kps[s] = (orbx_keypoint *)blk; h->st_desc[s] = blk + kb; h->st_cnt[s] = (int *)(blk + kb + db);
        }
        h->d_in_bytes = inb; h->out_cap = c; h->stage_chunk = ch;
    }
    if (!h->s_in) {
        // the copy stream gets the high priority level: the runtime maps streams of one priority onto a small shared pool of
        // hardware queues (4 by default), and a copy stream that lands on the compute stream's queue serialises uploads with
        // the kernels (measured: 95 k instead of 153 k frames/s in a process that had created a few streams before the handle)
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&h->s_in, hipStreamNonBlocking, hi));
        for (int s = 0; s < 2; ++s) {
            HIPCHK(hipEventCreateWithFlags(&h->ev_in[s], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_done[s], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_out[s], hipEventDisableTiming));
        }
    }
    return ORBX_OK;
}

extern "C" void *orbx_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void orbx_host_free(void *p) { if (p) hipHostFree(p); }

extern "C" orbx_status orbx_extract_batch(orbx_handle *h, int nframes, const uint8_t *imgs, int width, int height,
                                          int stride, int64_t frame_stride, orbx_keypoint *kps, uint8_t *desc,
                                          int32_t *counts, int cap) {
    if (!h) return fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!imgs || width <= 0 || height <= 0 || nframes <= 0) return fail(ORBX_EMPTY_IMAGE, "empty image");
    if (!kps || !desc || !counts || cap <= 0 || stride < width * orbx_fmt_channels(h->input_format)) return fail(ORBX_BAD_ARGUMENT, "bad output buffers / stride");
    orbx_status st = configure(h, width, height);
    if (st != ORBX_OK) return st;
    HIPCHK(hipSetDevice(h->dev));
    const int MB = h->p.max_batch;
    // chunks of max_batch frames (all frames of a call that fits one chunk stay resident for the pyramid accessors);
    // longer calls are pipelined chunk by chunk
    const int chunk = std::min(MB, nframes);
    const size_t fbytes = (size_t)stride * height;
    st = ensure_staging(h, (size_t)chunk * fbytes, cap, chunk);
    if (st != ORBX_OK) return st;
    const int nchunks = (nframes + chunk - 1) / chunk;
    // counts | status of every chunk of the call land here (page-locked: a pageable landing place would make each copy
    // synchronous), one slot per chunk so that nothing is reused before the call's single final wait
    if ((size_t)2 * nchunks * chunk > h->pin_stat_ints) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->pin_stat) { hipHostFree(h->pin_stat); h->pin_stat = nullptr; h->pin_stat_ints = 0; }
        HIPCHK(hipHostMalloc((void **)&h->pin_stat, (size_t)2 * nchunks * chunk * sizeof(int), hipHostMallocDefault));
        h->pin_stat_ints = (size_t)2 * nchunks * chunk;
    }
    int *hstat = h->pin_stat;
    orbx_status worst = ORBX_OK;
    // a single chunk needs no second stream: copies and kernels in order on the handle's stream (the latency path)
    const bool piped = nchunks > 1;
    // Pipelined calls use ONE copy stream for both directions: upload(c+1) and download(c-1) take turns on it while chunk c
    // computes.  (Round 2 had a second stream for the downloads; with uploads, downloads and kernels all in flight at once the
    // kernels of chunks >= 1 ran 1.4-5x slower from page-locked caller memory -- profiles/r03_host_io_trace.txt -- and
    // page-locked memory lost to pageable memory: 88 k against 103 k frames/s.  With one copy stream, the download queued in
    // front of the next upload and every dependency a stream-side event wait, the copy stream is busy back to back.)
    hipStream_t sin = piped ? h->s_in : h->stream, sout = sin;
    auto upload = [&](int c) -> hipError_t {
        const int s = c & 1, f0 = c * chunk, B = std::min(chunk, nframes - f0);
        if (c >= 2) {   // the kernels of chunk c-2 read this input set
            const hipError_t e = hipStreamWaitEvent(sin, h->ev_done[s], 0);
            if (e != hipSuccess) return e;
        }
        if (frame_stride == (int64_t)fbytes) {   // frames back to back: one copy per chunk
            const hipError_t e = hipMemcpyAsync(h->st_in[s], imgs + (int64_t)f0 * frame_stride, (size_t)B * fbytes, hipMemcpyHostToDevice, sin);
            if (e != hipSuccess) return e;
        } else {
            for (int i = 0; i < B; ++i) {
                const hipError_t e = hipMemcpyAsync(h->st_in[s] + (size_t)i * fbytes, imgs + (int64_t)(f0 + i) * frame_stride, fbytes,
                                                    hipMemcpyHostToDevice, sin);
                if (e != hipSuccess) return e;
            }
        }
        return piped ? hipEventRecord(h->ev_in[s], sin) : hipSuccess;
    };
    // Page-locked, device-mapped output buffers (orbx_host_alloc, hipHostMalloc, hipHostRegister with the mapped flag) are
    // written by k_describe ITSELF over the link: the results leave while the chunk computes and while the next chunk's frames
    // come in on the link's other direction, and the copy stream carries uploads only (134 k against 120 k frames/s at chunks
    // of 64, 256 frames per call).  Only the per-frame counts | status stay in device memory (several kernels update the
    // status with atomics) and come back with one small copy per chunk.  The caller reads its buffers after the call returns
    // (the call ends with a wait on both streams), as with the copies.
    orbx_keypoint *zk = nullptr; uint8_t *zd = nullptr;
    {
        void *pk = nullptr, *pd = nullptr;
        if (hipHostGetDevicePointer(&pk, kps, 0) == hipSuccess && hipHostGetDevicePointer(&pd, desc, 0) == hipSuccess && pk && pd) {
            zk = (orbx_keypoint *)pk; zd = (uint8_t *)pd;
        } else {
            (void)hipGetLastError();   // pageable (or unmapped) buffers: results are staged in device memory and copied
        }
    }
    // download of chunk c: queued behind the chunk's kernels; its event frees the output set for chunk c+2's kernels
    auto download_enqueue = [&](int c) -> hipError_t {
        const int s = c & 1, f0 = c * chunk, B = std::min(chunk, nframes - f0);
        hipError_t e = piped ? hipStreamWaitEvent(sout, h->ev_done[s], 0) : hipSuccess;
        if (e == hipSuccess && !zk) e = hipMemcpyAsync(kps + (int64_t)f0 * cap, h->st_kps[s], (size_t)B * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, sout);
        if (e == hipSuccess && !zk) e = hipMemcpyAsync(desc + (int64_t)f0 * cap * 32, h->st_desc[s], (size_t)B * cap * 32, hipMemcpyDeviceToHost, sout);
        // counts | status are one device block (st_cnt[s][0 .. 2 chunk)): ONE small copy (every copy costs ~15 us of link
        // turn-around whatever its size); the counts go on to the caller's array from the landing buffer
        if (e == hipSuccess) e = hipMemcpyAsync(hstat + (size_t)c * 2 * chunk, h->st_cnt[s], (size_t)2 * chunk * sizeof(int), hipMemcpyDeviceToHost, sout);
        if (e == hipSuccess && piped) e = hipEventRecord(h->ev_out[s], sout);
        return e;
    };
    auto collect = [&]() {   // after the final wait: counts to the caller, worst status of the call
        for (int c = 0; c < nchunks; ++c) {
            const int f0 = c * chunk, B = std::min(chunk, nframes - f0);
            for (int i = 0; i < B; ++i) {
                counts[f0 + i] = hstat[(size_t)c * 2 * chunk + i];
                if (hstat[(size_t)c * 2 * chunk + chunk + i] != ORBX_OK) worst = (orbx_status)hstat[(size_t)c * 2 * chunk + chunk + i];
            }
        }
    };
    // Latency path (one chunk that fills its staging block exactly, a few MB at most -- the drop-in call of Tracking.cc): the
    // frames go through page-locked staging (one true DMA instead of a runtime-staged pageable copy) and the results come
    // back with ONE copy of the whole output block instead of four.
    const size_t out_bytes = (size_t)chunk * cap * (sizeof(orbx_keypoint) + 32) + (size_t)2 * chunk * sizeof(int);
    const size_t in_bytes_q = (size_t)nframes * fbytes;
    const bool stage_in = in_bytes_q <= ((size_t)1 << 20);   // beyond ~1 MB the host-side memcpy costs more than the runtime's own staging
    const bool quick = !piped && nframes == h->stage_chunk && cap == h->out_cap && out_bytes <= ((size_t)4 << 20) &&
                       pin_reserve(h, (stage_in ? in_bytes_q : 0) + out_bytes) == ORBX_OK;
    if (quick) {
        uint8_t *pin_in = h->pin, *pin_out = h->pin + (stage_in ? in_bytes_q : 0);
        if (stage_in) {
            for (int i = 0; i < nframes; ++i) memcpy(pin_in + (size_t)i * fbytes, imgs + (int64_t)i * frame_stride, fbytes);
            HIPCHK(hipMemcpyAsync(h->st_in[0], pin_in, in_bytes_q, hipMemcpyHostToDevice, h->stream));
        } else {
            HIPCHK(upload(0));
        }
        st = run_chunk(h, nframes, h->st_in[0], width, height, stride, (int64_t)fbytes, h->st_kps[0], h->st_desc[0], h->st_cnt[0],
                       h->st_cnt[0] + chunk, cap);
        if (st != ORBX_OK) { hipStreamSynchronize(h->stream); return st; }
        HIPCHK(hipMemcpyAsync(pin_out, h->st_kps[0], out_bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        const size_t kb = (size_t)nframes * cap * sizeof(orbx_keypoint), db = (size_t)nframes * cap * 32;
        const int *cnt = (const int *)(pin_out + kb + db);
        // only the meaningful part of every frame's records leaves the staging buffer
        for (int i = 0; i < nframes; ++i) {
            const int n = std::min(std::max(cnt[i], 0), cap);
            memcpy(kps + (int64_t)i * cap, pin_out + (size_t)i * cap * sizeof(orbx_keypoint), (size_t)n * sizeof(orbx_keypoint));
            memcpy(desc + (int64_t)i * cap * 32, pin_out + kb + (size_t)i * cap * 32, (size_t)n * 32);
            counts[i] = cnt[i];
            if (cnt[nframes + i] != ORBX_OK) worst = (orbx_status)cnt[nframes + i];
        }
        if (worst != ORBX_OK) return fail(worst, "a frame exceeded the keypoint / candidate capacity");
        return ORBX_OK;
    }
    HIPCHK(upload(0));
    for (int c = 0; c < nchunks; ++c) {
        const int s = c & 1, f0 = c * chunk, B = std::min(chunk, nframes - f0);
        if (piped) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_in[s], 0));             // the chunk's frames have arrived
        st = run_chunk(h, B, h->st_in[s], width, height, stride, (int64_t)fbytes, zk ? zk + (int64_t)f0 * cap : h->st_kps[s],
                       zk ? zd + (int64_t)f0 * cap * 32 : h->st_desc[s], h->st_cnt[s], h->st_cnt[s] + chunk, cap);
        if (st != ORBX_OK) { hipStreamSynchronize(h->stream); hipStreamSynchronize(sin); return st; }
        if (piped) HIPCHK(hipEventRecord(h->ev_done[s], h->stream));
        // copy stream: results of chunk c-1 first (its kernels are done or nearly so), then the frames of chunk c+1 (its input
        // set was read by chunk c-1: the download in front of it already waited for that chunk)
        if (c >= 1) HIPCHK(download_enqueue(c - 1));
        if (c + 1 < nchunks) HIPCHK(upload(c + 1));
        // the calling thread stays one chunk ahead of the device, not more: it waits here until chunk c-1's results have left
        // their output set, which chunk c+1's kernels (enqueued next) write.  (Letting it run ahead of the whole call with
        // stream-side waits instead was measured: 82 k against 111 k frames/s at chunks of 32, 84 k against 95 k at 16 chunks
        // of 64 -- many queued cross-stream waits cost more than the wake-ups they save.)
        if (piped && c >= 1) HIPCHK(hipEventSynchronize(h->ev_out[(c - 1) & 1]));
    }
    HIPCHK(download_enqueue(nchunks - 1));
    HIPCHK(hipStreamSynchronize(sout));
    if (piped) HIPCHK(hipStreamSynchronize(h->stream));
    collect();
    if (worst != ORBX_OK) return fail(worst, "a frame exceeded the keypoint / candidate capacity");
    return ORBX_OK;
}

extern "C" orbx_status orbx_extract(orbx_handle *h, const uint8_t *img, int width, int height, int stride,
                                    orbx_keypoint *kps, uint8_t *desc, int cap, int *n) {
    if (!n) return fail(ORBX_BAD_ARGUMENT, "null count pointer");
    int32_t cnt = 0;
    orbx_status st = orbx_extract_batch(h, 1, img, width, height, stride, (int64_t)stride * height, kps, desc, &cnt, cap);
    if (st == ORBX_OK || st == ORBX_CAPACITY) *n = cnt;
    return st;
}

// ---------------------------------------------------------------- pyramid access
static orbx_status check_level(orbx_handle *h, int frame, int level) {
    if (!h) return fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!h->configured || h->last_batch == 0) return fail(ORBX_BAD_ARGUMENT, "no frame extracted yet");
    if (level < 0 || level >= h->p.nlevels || frame < 0 || frame >= h->last_batch) return fail(ORBX_BAD_ARGUMENT, "frame/level out of range");
    return ORBX_OK;
}
extern "C" orbx_status orbx_pyramid_level_info(orbx_handle *h, int level, int *width, int *height, int *pitch) {
    orbx_status st = check_level(h, 0, level);
    if (st != ORBX_OK) return st;
    if (width) *width = h->geom.lv[level].pw;
    if (height) *height = h->geom.lv[level].ph;
    if (pitch) *pitch = h->geom.lv[level].pitch;
    return ORBX_OK;
}
extern "C" orbx_status orbx_pyramid_level_device(orbx_handle *h, int frame, int level, const uint8_t **d_ptr) {
    orbx_status st = check_level(h, frame, level);
    if (st != ORBX_OK) return st;
    *d_ptr = h->d_pyr + (size_t)frame * h->geom.pyr_bytes + h->geom.lv[level].off;
    return ORBX_OK;
}
static orbx_status copy_level(orbx_handle *h, const uint8_t *slab, int frame, int level, uint8_t *dst, int dst_stride) {
    orbx_status st = check_level(h, frame, level);
    if (st != ORBX_OK) return st;
    const OrbxLevelGeom &L = h->geom.lv[level];
    if (!dst || dst_stride < L.pw) return fail(ORBX_BAD_ARGUMENT, "dst / dst_stride");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy2D(dst, dst_stride, slab + (size_t)frame * h->geom.pyr_bytes + L.off, L.pitch, L.pw, L.ph,
                       hipMemcpyDeviceToHost));
    return ORBX_OK;
}
extern "C" orbx_status orbx_pyramid_level_copy(orbx_handle *h, int frame, int level, uint8_t *dst, int dst_stride) {
    return copy_level(h, h ? h->d_pyr : nullptr, frame, level, dst, dst_stride);
}
extern "C" orbx_status orbx_debug_blur_copy(orbx_handle *h, int frame, int level, uint8_t *dst, int dst_stride) {
    orbx_status st = check_level(h, frame, level);
    if (st != ORBX_OK) return st;
    if (!h->blur_valid) {  // stand-alone k_blur over the resident pyramid (same arithmetic as the fused path)
        HIPCHK(hipSetDevice(h->dev));
        if (!h->d_blur) {      // the blurred slab only exists for inspection: allocated on first request
            HIPCHK(hipMalloc(&h->d_blur, (size_t)h->p.max_batch * h->geom.pyr_bytes + 256));
            HIPCHK(hipMemsetAsync(h->d_blur, 0, (size_t)h->p.max_batch * h->geom.pyr_bytes, h->stream));   // ordered with k_blur below
        }
        { ProfScope ps(h, ORBX_K_BLUR);
          orbx_launch_blur(h->stream, h->dg, h->last_batch, h->d_pyr, h->d_blur); }
        h->blur_valid = true;
    }
    return copy_level(h, h->d_blur, frame, level, dst, dst_stride);
}

// ---------------------------------------------------------------- per-stage inspection
extern "C" orbx_status orbx_debug_candidates(orbx_handle *h, int frame, int level, orbx_keypoint *out, int cap, int *n) {
    orbx_status st = check_level(h, frame, level);
    if (st != ORBX_OK) return st;
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipStreamSynchronize(h->stream));
    const OrbxLevelGeom &L = h->geom.lv[level];
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, h->d_cand_count + frame * h->p.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    if (n) *n = cnt;
    const int m = std::min(cnt, L.cand_cap);
    std::vector<uint2> rec(std::max(m, 1));
    if (m > 0)
        HIPCHK(hipMemcpy(rec.data(), h->d_dense + (size_t)frame * h->geom.cand_total + L.cand_begin, (size_t)m * sizeof(uint2),
                         hipMemcpyDeviceToHost));
    for (int i = 0; i < m && i < cap; ++i) {
        orbx_keypoint k;
        k.x = (float)(rec[i].x & 0xfff); k.y = (float)((rec[i].x >> 12) & 0xfff);
        k.size = 7.f; k.angle = -1.f; k.response = (float)(rec[i].x >> 24);
        k.octave = 0; k.class_id = (int32_t)(rec[i].y & 0xffffff);  // emission-order key (debug only)
        out[i] = k;
    }
    if (cnt > L.cand_cap || m > cap) return fail(ORBX_CAPACITY, "candidate capacity");
    return ORBX_OK;
}

extern "C" orbx_status orbx_debug_level_keypoints(orbx_handle *h, int frame, int level, orbx_keypoint *out, int cap, int *n) {
    orbx_status st = check_level(h, frame, level);
    if (st != ORBX_OK) return st;
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipStreamSynchronize(h->stream));
    const OrbxLevelGeom &L = h->geom.lv[level];
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, h->d_lvl_count + frame * h->p.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    if (n) *n = cnt;
    std::vector<uint32_t> pos(std::max(cnt, 1));
    std::vector<float> ang(std::max(cnt, 1));
    if (cnt > 0) {
        HIPCHK(hipMemcpy(pos.data(), h->d_lvl_kp + (size_t)frame * h->geom.kp_total + L.kp_begin, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ang.data(), h->d_lvl_angle + (size_t)frame * h->geom.kp_total + L.kp_begin, cnt * sizeof(float), hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < cnt && i < cap; ++i) {
        orbx_keypoint k;
        k.x = (float)((pos[i] & 0xfff) + ORBX_EDGE - 3); k.y = (float)(((pos[i] >> 12) & 0xfff) + ORBX_EDGE - 3);
        k.size = L.size; k.angle = ang[i]; k.response = (float)(pos[i] >> 24); k.octave = level; k.class_id = -1;
        out[i] = k;
    }
    if (cnt > cap) return fail(ORBX_CAPACITY, "output capacity");
    return ORBX_OK;
}

// ---------------------------------------------------------------- scratch arena
static orbx_status scratch_reserve(orbx_handle *h, size_t bytes) {
    h->scratch_used = 0;
    if (bytes <= h->scratch_bytes) return ORBX_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(h->d_scratch); h->d_scratch = nullptr; h->scratch_bytes = 0;
    const size_t want = std::max(bytes, (size_t)1 << 20);
    HIPCHK(hipMalloc(&h->d_scratch, want));
    h->scratch_bytes = want;
    return ORBX_OK;
}
template <typename T> static T *scratch_take(orbx_handle *h, size_t count) {
    T *p = (T *)(h->d_scratch + h->scratch_used);
    h->scratch_used += (count * sizeof(T) + 255) & ~(size_t)255;
    return p;
}
static inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---------------------------------------------------------------- device grid + gated candidate lists (orbx_gate.h)
static orbx_status gate_items_reserve(orbx_handle *h, size_t words) {
    if (words <= h->gate_items_cap) return ORBX_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(h->d_gate_items); h->d_gate_items = nullptr; h->gate_items_cap = 0;
    const size_t want = std::max(words + words / 2, (size_t)1 << 16);
    HIPCHK(hipMalloc(&h->d_gate_items, want * sizeof(uint32_t)));
    h->gate_items_cap = want;
    return ORBX_OK;
}
static bool grid_params(float min_x, float max_x, float min_y, float max_y, DGrid &gp) {
    if (!(max_x > min_x) || !(max_y > min_y)) return false;
    gp.minx = min_x; gp.miny = min_y;
    gp.winv = 64.0f / (max_x - min_x);     // mfGridElementWidthInv (src/Frame.cc:96-99): FRAME_GRID_COLS / (mnMaxX - mnMinX)
    gp.hinv = 48.0f / (max_y - min_y);
    return true;
}

// Frame::AssignFeaturesToGrid on the device, for the buffers orbx_extract_batch_device (or orbx_undistort_keypoints_device)
// filled: the keypoints never leave the GPU between extraction and a gated match.
extern "C" orbx_status orbx_grid_build_device(orbx_handle *h, int nframes, const orbx_keypoint *d_kps, const int32_t *d_counts,
                                              int cap, const float *bounds4, int32_t *d_cell_begin, uint16_t *d_items) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (nframes <= 0 || !d_kps || !d_counts || cap <= 0 || cap > 65535 || !bounds4 || !d_cell_begin || !d_items)
        return fail(ORBX_BAD_ARGUMENT, "bad argument (cap <= 65535: bucket entries are 16-bit feature indices)");
    DGrid gp;
    if (!grid_params(bounds4[0], bounds4[1], bounds4[2], bounds4[3], gp)) return fail(ORBX_BAD_ARGUMENT, "bad image bounds");
    HIPCHK(hipSetDevice(h->dev));
    { ProfScope ps(h, ORBX_K_MISC);
      orbx_launch_grid_build(h->stream, gp, nframes, d_kps, d_counts, 0, cap, d_cell_begin, d_items); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    return ORBX_OK;
}

static orbx_status pin_reserve(orbx_handle *h, size_t bytes) {
    if (bytes <= h->pin_bytes) return ORBX_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->pin) hipHostFree(h->pin);
    h->pin = nullptr; h->pin_bytes = 0;
    const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 20);
    HIPCHK(hipHostMalloc((void **)&h->pin, want, hipHostMallocDefault));
    h->pin_bytes = want;
    return ORBX_OK;
}

// One call = one upload (everything packed into page-locked staging), two kernels (k_grid_build, k_gate) and one download that
// is waited for: the (offset, count) spans, the fill count, and speculatively as many candidate entries as the last call
// produced (+50 %) -- only a call that produces more than that pays a second copy.
// BATCHED: K targets (keyframes) share the call -- their keypoints / descriptors are packed `fstride` records apart, k_grid_build
// builds the K grids in one launch (one workgroup per target), every query names its target (DGateQuery::frame) and k_gate
// serves all of them in one launch.  The ~60 us floor of a synchronous call (upload, two launches, waited download) is paid once
// per batch instead of once per (keyframe, point set) pair.  All targets share the image bounds (Frame's static mnMinX .. mnMaxY).
orbx_status orbx_gate_lists_batch(orbx_handle *h, const OrbxGateTarget *tg, int K, float min_x, float max_x, float min_y, float max_y,
                                  const DGateQuery *q, const uint8_t *qdesc, int nq, OrbxGateLists &out, int nqdesc) {
    if (nqdesc < 0) nqdesc = nq;
    out.span.assign((size_t)std::max(nq, 0), make_uint2(0u, 0u));
    out.items.clear();
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    int fstride = 0;
    for (int k = 0; k < K; ++k) fstride = std::max(fstride, tg[k].n);
    if (nq <= 0 || K <= 0 || fstride <= 0) return ORBX_OK;
    if (fstride > 65535) return fail(ORBX_UNSUPPORTED, "more than 65535 target features (candidate entries carry 16-bit indices)");
    DGrid gp;
    if (!grid_params(min_x, max_x, min_y, max_y, gp)) return fail(ORBX_BAD_ARGUMENT, "bad image bounds");
    HIPCHK(hipSetDevice(h->dev));
    const size_t nrec = (size_t)K * fstride;
    // input block (uploaded in one copy): cursor | counts | keys | descriptors | queries | query descriptors
    const size_t o_cur = 0, o_cnt = 256, o_keys = o_cnt + pad256((size_t)K * sizeof(int)), o_desc = o_keys + pad256(nrec * sizeof(orbx_keypoint)),
                 o_q = o_desc + pad256(nrec * 32), o_qd = o_q + pad256((size_t)nq * sizeof(DGateQuery)), in_bytes = o_qd + pad256((size_t)nqdesc * 32);
    // device-only: bucket offsets | bucket items | spans
    const size_t o_cb = in_bytes, o_it = o_cb + pad256((size_t)K * (64 * 48 + 1) * sizeof(int)), o_span = o_it + pad256(nrec * sizeof(uint16_t)),
                 dev_bytes = o_span + pad256((size_t)nq * sizeof(uint2));
    orbx_status st = scratch_reserve(h, dev_bytes + 256);
    if (st != ORBX_OK) return st;
    st = gate_items_reserve(h, std::max(h->gate_guess, (size_t)1 << 16));
    if (st != ORBX_OK) return st;
    const size_t span_bytes = (size_t)nq * sizeof(uint2);
    size_t guess = std::min(h->gate_guess, h->gate_items_cap);
    st = pin_reserve(h, in_bytes + span_bytes + 256 + h->gate_items_cap * 4);
    if (st != ORBX_OK) return st;
    uint8_t *pin_in = h->pin, *pin_out = h->pin + in_bytes;   // pin_out: spans | cursor (256) | items
    memset(pin_in + o_cur, 0, 256);
    for (int k = 0; k < K; ++k) {
        ((int *)(pin_in + o_cnt))[k] = tg[k].n;
        if (tg[k].n <= 0) continue;
        memcpy(pin_in + o_keys + (size_t)k * fstride * sizeof(orbx_keypoint), tg[k].keys, (size_t)tg[k].n * sizeof(orbx_keypoint));
        memcpy(pin_in + o_desc + (size_t)k * fstride * 32, tg[k].desc, (size_t)tg[k].n * 32);
    }
    memcpy(pin_in + o_q, q, (size_t)nq * sizeof(DGateQuery));
    memcpy(pin_in + o_qd, qdesc, (size_t)nqdesc * 32);
    uint8_t *d = scratch_take<uint8_t>(h, dev_bytes);
    uint32_t *dcur = (uint32_t *)(d + o_cur);
    const orbx_keypoint *dk = (const orbx_keypoint *)(d + o_keys);
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(d, pin_in, in_bytes, hipMemcpyHostToDevice, s));
    uint32_t total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        { ProfScope ps(h, ORBX_K_MATCH);
          if (attempt == 0) orbx_launch_grid_build(s, gp, K, dk, (const int *)(d + o_cnt), 0, fstride, (int *)(d + o_cb), (uint16_t *)(d + o_it));
          orbx_launch_gate(s, gp, dk, d + o_desc, (const int *)(d + o_cb), (const uint16_t *)(d + o_it), (const DGateQuery *)(d + o_q),
                           d + o_qd, nq, (uint2 *)(d + o_span), dcur, h->d_gate_items, (uint32_t)std::min<size_t>(h->gate_items_cap, 0xffffffffu),
                           fstride); }
        HIPCHK(hipMemcpyAsync(pin_out, d + o_span, span_bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(pin_out + span_bytes, dcur, 4, hipMemcpyDeviceToHost, s));
        if (guess > 0) HIPCHK(hipMemcpyAsync(pin_out + span_bytes + 256, h->d_gate_items, guess * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        memcpy(&total, pin_out + span_bytes, 4);
        if (total <= h->gate_items_cap) break;
        // the lists did not fit the buffer: nothing beyond the spans was trusted; grow, reset the cursor, run k_gate again
        if (attempt == 1) return fail(ORBX_CAPACITY, "candidate lists");
        st = gate_items_reserve(h, total);
        if (st == ORBX_OK) st = pin_reserve(h, in_bytes + span_bytes + 256 + h->gate_items_cap * 4);
        if (st != ORBX_OK) return st;
        pin_in = h->pin; pin_out = h->pin + in_bytes;
        HIPCHK(hipMemsetAsync(dcur, 0, 4, s));
        guess = total;
    }
    if (total > guess) {   // more entries than the speculative copy covered
        HIPCHK(hipMemcpyAsync(pin_out + span_bytes + 256 + guess * 4, h->d_gate_items + guess, (size_t)(total - guess) * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    h->gate_guess = std::max<size_t>(4096, (size_t)total + total / 2);
    memcpy(out.span.data(), pin_out, span_bytes);
    out.items.resize(total);
    if (total) memcpy(out.items.data(), pin_out + span_bytes + 256, (size_t)total * 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    return ORBX_OK;
}
orbx_status orbx_gate_lists(orbx_handle *h, const orbx_keypoint *tkeys, const uint8_t *tdesc, int nt, float min_x, float max_x,
                            float min_y, float max_y, const DGateQuery *q, const uint8_t *qdesc, int nq, OrbxGateLists &out) {
    const OrbxGateTarget tg = {tkeys, tdesc, nt};
    return orbx_gate_lists_batch(h, &tg, 1, min_x, max_x, min_y, max_y, q, qdesc, nq, out);
}

// host-buffer form of the primitive itself (tests, tools/policy_rates.py): xyr = (x, y, r) per query, levels = (min, max)
extern "C" orbx_status orbx_gated_candidates(orbx_handle *h, const orbx_keypoint *tkeys, const uint8_t *tdesc, int nt,
                                             const float *bounds4, const float *xyr, const int32_t *levels, const uint8_t *qdesc,
                                             int nq, uint32_t *begin, uint32_t *items, int items_cap, int *total) {
    if (!h || !bounds4 || nq < 0 || nt < 0 || !begin || !total || (nq > 0 && (!xyr || !levels || !qdesc)) || (nt > 0 && (!tkeys || !tdesc)))
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    std::vector<DGateQuery> q((size_t)nq);
    for (int i = 0; i < nq; ++i) { q[i].x = xyr[3 * i]; q[i].y = xyr[3 * i + 1]; q[i].r = xyr[3 * i + 2]; q[i].min_level = levels[2 * i]; q[i].max_level = levels[2 * i + 1]; }
    OrbxGateLists L;
    orbx_status st = orbx_gate_lists(h, tkeys, tdesc, nt, bounds4[0], bounds4[1], bounds4[2], bounds4[3], q.data(), qdesc, nq, L);
    if (st != ORBX_OK) return st;
    begin[0] = 0;
    for (int i = 0; i < nq; ++i) begin[i + 1] = begin[i] + (uint32_t)L.count(i);
    *total = (int)begin[nq];
    if ((int)begin[nq] > items_cap) return fail(ORBX_CAPACITY, "candidate list capacity");
    for (int i = 0; i < nq && items; ++i)
        if (L.count(i)) memcpy(items + begin[i], L.list(i), (size_t)L.count(i) * 4);
    return ORBX_OK;
}

orbx_status orbx_block_distances(orbx_handle *h, const uint8_t *d1, int n1, const uint8_t *d2, int n2,
                                 const std::vector<DDistRow> &rows, const std::vector<uint32_t> &col_idx, size_t total,
                                 std::vector<uint16_t> &out) {
    out.assign(total, 0);
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (rows.empty() || total == 0) return ORBX_OK;
    HIPCHK(hipSetDevice(h->dev));
    const size_t o_a = 0, o_b = o_a + pad256((size_t)n1 * 32), o_r = o_b + pad256((size_t)n2 * 32),
                 o_c = o_r + pad256(rows.size() * sizeof(DDistRow)), in_bytes = o_c + pad256(col_idx.size() * 4);
    orbx_status st = scratch_reserve(h, in_bytes + 256);
    if (st == ORBX_OK) st = gate_items_reserve(h, (total + 1) / 2);
    if (st == ORBX_OK) st = pin_reserve(h, in_bytes + pad256(total * sizeof(uint16_t)));
    if (st != ORBX_OK) return st;
    uint8_t *pin_in = h->pin, *pin_out = h->pin + in_bytes;
    memcpy(pin_in + o_a, d1, (size_t)n1 * 32);
    memcpy(pin_in + o_b, d2, (size_t)n2 * 32);
    memcpy(pin_in + o_r, rows.data(), rows.size() * sizeof(DDistRow));
    memcpy(pin_in + o_c, col_idx.data(), col_idx.size() * 4);
    uint8_t *d = scratch_take<uint8_t>(h, in_bytes);
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(d, pin_in, in_bytes, hipMemcpyHostToDevice, s));
    { ProfScope ps(h, ORBX_K_MATCH);
      orbx_launch_block_dist(s, d + o_a, d + o_b, (const DDistRow *)(d + o_r), (const uint32_t *)(d + o_c), (int)rows.size(),
                             (uint16_t *)h->d_gate_items); }
    HIPCHK(hipMemcpyAsync(pin_out, h->d_gate_items, total * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    memcpy(out.data(), pin_out, total * sizeof(uint16_t));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    return ORBX_OK;
}

// ---------------------------------------------------------------- matching
extern "C" orbx_status orbx_match_bruteforce_device(orbx_handle *h, int npairs, const uint8_t *d_q, const int32_t *d_nq,
                                                    int64_t q_stride, const uint8_t *d_t, const int32_t *d_nt,
                                                    int64_t t_stride, int32_t *d_best_idx, int32_t *d_best_dist,
                                                    int32_t *d_second_dist, int out_stride) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (npairs <= 0 || !d_q || !d_t || !d_nq || !d_nt || !d_best_idx || !d_best_dist || !d_second_dist || out_stride <= 0)
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    HIPCHK(hipSetDevice(h->dev));
    const size_t need = orbx_match_workspace_bytes(npairs, out_stride);
    if (need > h->match_ws_bytes) {   // grows only when a larger problem than ever before arrives
        HIPCHK(hipStreamSynchronize(h->stream));
        hipFree(h->d_match_ws); h->d_match_ws = nullptr; h->match_ws_bytes = 0;
        HIPCHK(hipMalloc(&h->d_match_ws, need));
        h->match_ws_bytes = need;
    }
    { ProfScope ps(h, ORBX_K_MATCH);
      orbx_launch_match(h->stream, npairs, out_stride, d_q, d_nq, q_stride, d_t, d_nt, t_stride, d_best_idx, d_best_dist,
                        d_second_dist, out_stride, h->d_match_ws, h->match_kernel); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    return ORBX_OK;
}

extern "C" orbx_status orbx_match_bruteforce(orbx_handle *h, const uint8_t *q, int nq, const uint8_t *t, int nt,
                                             int32_t *best_idx, int32_t *best_dist, int32_t *second_dist) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (nq < 0 || nt < 0 || (nq > 0 && (!q || !best_idx || !best_dist || !second_dist)) || (nt > 0 && !t))
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (nq == 0) return ORBX_OK;
    HIPCHK(hipSetDevice(h->dev));
    orbx_status st = scratch_reserve(h, pad256((size_t)nq * 32) + pad256((size_t)std::max(nt, 1) * 32) + 256 +
                                            pad256((size_t)3 * nq * sizeof(int)));
    if (st != ORBX_OK) return st;
    uint8_t *dq = scratch_take<uint8_t>(h, (size_t)nq * 32), *dt = scratch_take<uint8_t>(h, (size_t)std::max(nt, 1) * 32);
    int *dn = scratch_take<int>(h, 2), *dout = scratch_take<int>(h, (size_t)3 * nq);
    int hn[2] = {nq, nt};
    HIPCHK(hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, h->stream));
    if (nt > 0) HIPCHK(hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dn, hn, sizeof(hn), hipMemcpyHostToDevice, h->stream));
    st = orbx_match_bruteforce_device(h, 1, dq, dn, 0, dt, dn + 1, 0, dout, dout + nq, dout + 2 * nq, nq);
    if (st == ORBX_OK) {
        HIPCHK(hipMemcpyAsync(best_idx, dout, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(best_dist, dout + nq, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(second_dist, dout + 2 * nq, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));   // also keeps hn[] alive until the H2D copy has been consumed
    }
    return st;
}

extern "C" orbx_status orbx_hamming_matrix(orbx_handle *h, const uint8_t *q, int nq, const uint8_t *t, int nt,
                                           uint16_t *dist) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (nq <= 0 || nt <= 0) return ORBX_OK;
    if (!q || !t || !dist) return fail(ORBX_BAD_ARGUMENT, "null argument");
    HIPCHK(hipSetDevice(h->dev));
    orbx_status st = scratch_reserve(h, pad256((size_t)nq * 32) + pad256((size_t)nt * 32) + pad256((size_t)nq * nt * sizeof(uint16_t)));
    if (st != ORBX_OK) return st;
    uint8_t *dq = scratch_take<uint8_t>(h, (size_t)nq * 32), *dt = scratch_take<uint8_t>(h, (size_t)nt * 32);
    uint16_t *dd = scratch_take<uint16_t>(h, (size_t)nq * nt);
    HIPCHK(hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, h->stream));
    { ProfScope ps(h, ORBX_K_MATCH);
      orbx_launch_hamming_matrix(h->stream, dq, nq, dt, nt, dd); }
    hipError_t e = hipMemcpyAsync(dist, dd, (size_t)nq * nt * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    return ORBX_OK;
}

// ---------------------------------------------------------------- stream / timing
extern "C" void *orbx_get_stream(orbx_handle *h) { return h ? (void *)h->stream : nullptr; }
extern "C" orbx_status orbx_set_stream(orbx_handle *h, void *s) {
    if (!h || h->host_only) return fail(ORBX_BAD_ARGUMENT, "no device handle");
    hipSetDevice(h->dev);
    hipStreamSynchronize(h->stream);
    prof_drain(h);
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return ORBX_OK;
}
extern "C" orbx_status orbx_synchronize(orbx_handle *h) {
    if (!h || h->host_only) return fail(ORBX_BAD_ARGUMENT, "no device handle");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ORBX_OK;
}
extern "C" orbx_status orbx_profile_enable(orbx_handle *h, uint32_t mask) {
    if (!h || h->host_only) return fail(ORBX_BAD_ARGUMENT, "no device handle");
    prof_drain(h);
    h->prof_mask = mask;
    return ORBX_OK;
}
extern "C" orbx_status orbx_profile_read(orbx_handle *h, float *ms, int32_t *launches, int reset) {
    if (!h || h->host_only) return fail(ORBX_BAD_ARGUMENT, "no device handle");
    hipSetDevice(h->dev);
    prof_drain(h);
    if (ms) memcpy(ms, h->ms, sizeof(h->ms));
    if (launches) memcpy(launches, h->launches, sizeof(h->launches));
    if (reset) { memset(h->ms, 0, sizeof(h->ms)); memset(h->launches, 0, sizeof(h->launches)); }
    return ORBX_OK;
}

// ================================================================ matcher policies on the path (a13, a16, a17)
// The Hamming work runs on the GPU; the order-dependent bookkeeping of each policy is host code, as the
// reference's own structure dictates.
#include <cmath>
#include <climits>

// ---------------------------------------------------------------- a17: Frame grid (src/Frame.cc:432-460, 633-745)
struct orbx_grid {
    static const int COLS = 64, ROWS = 48;     // FRAME_GRID_COLS / FRAME_GRID_ROWS (include/Frame.h:54,59)
    float minx, miny, winv, hinv;
    const orbx_keypoint *kps;
    std::vector<int> begin, items;
};

extern "C" orbx_grid *orbx_grid_create(const orbx_keypoint *kps, int n, float min_x, float max_x, float min_y, float max_y) {
    if (n < 0 || (n > 0 && !kps) || !(max_x > min_x) || !(max_y > min_y)) { g_last_error = "bad grid arguments"; return nullptr; }
    orbx_grid *g = new orbx_grid();
    g->minx = min_x; g->miny = min_y; g->kps = kps;
    g->winv = (float)orbx_grid::COLS / (max_x - min_x);
    g->hinv = (float)orbx_grid::ROWS / (max_y - min_y);
    const int nc = orbx_grid::COLS * orbx_grid::ROWS;
    std::vector<int> cell(n);
    g->begin.assign(nc + 1, 0);
    for (int i = 0; i < n; ++i) {                     // PosInGrid: C round(), then the range test
        const int px = (int)roundf((kps[i].x - min_x) * g->winv), py = (int)roundf((kps[i].y - min_y) * g->hinv);
        cell[i] = (px < 0 || px >= orbx_grid::COLS || py < 0 || py >= orbx_grid::ROWS) ? -1 : px * orbx_grid::ROWS + py;
        if (cell[i] >= 0) g->begin[cell[i] + 1]++;
    }
    for (int c = 0; c < nc; ++c) g->begin[c + 1] += g->begin[c];
    g->items.resize(n);
    std::vector<int> fill(nc, 0);
    for (int i = 0; i < n; ++i)
        if (cell[i] >= 0) g->items[g->begin[cell[i]] + fill[cell[i]]++] = i;   // push_back order = feature order
    return g;
}
extern "C" void orbx_grid_destroy(orbx_grid *g) { delete g; }

// GetFeaturesInArea: same cell walk (x outer, y inner), same level filter quirk (`minLevel > 0 || maxLevel >= 0`)
extern "C" int orbx_grid_query(const orbx_grid *g, float x, float y, float r, int min_level, int max_level, int32_t *out,
                               int cap) {
    if (!g) return -1;
    int x0 = std::max(0, (int)floorf((x - g->minx - r) * g->winv));
    if (x0 >= orbx_grid::COLS) return 0;
    int x1 = std::min(orbx_grid::COLS - 1, (int)ceilf((x - g->minx + r) * g->winv));
    if (x1 < 0) return 0;
    int y0 = std::max(0, (int)floorf((y - g->miny - r) * g->hinv));
    if (y0 >= orbx_grid::ROWS) return 0;
    int y1 = std::min(orbx_grid::ROWS - 1, (int)ceilf((y - g->miny + r) * g->hinv));
    if (y1 < 0) return 0;
    const bool check = (min_level > 0) || (max_level >= 0);
    int n = 0;
    for (int ix = x0; ix <= x1; ++ix)
        for (int iy = y0; iy <= y1; ++iy) {
            const int c = ix * orbx_grid::ROWS + iy;
            for (int j = g->begin[c]; j < g->begin[c + 1]; ++j) {
                const orbx_keypoint &kp = g->kps[g->items[j]];
                if (check) {
                    if (kp.octave < min_level) continue;
                    if (max_level >= 0 && kp.octave > max_level) continue;
                }
                if (fabsf(kp.x - x) < r && fabsf(kp.y - y) < r) {
                    if (n < cap && out) out[n] = g->items[j];
                    ++n;
                }
            }
        }
    return n;
}

// ---------------------------------------------------------------- a15: ComputeThreeMaxima (src/ORBmatcher.cc:2026-2068)
extern "C" void orbx_three_maxima(const int32_t *sizes, int L, int *ind1, int *ind2, int *ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; ++i) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

// ---------------------------------------------------------------- a13: SearchForInitialization (src/ORBmatcher.cc:570-712)
// GPU: the grid of F2 and, per level-0 feature of F1, its window's candidates with their Hamming distances in the
// reference's visiting order (orbx_gate_lists).  Host: the sequential selection pass over those lists.
extern "C" orbx_status orbx_search_for_initialization(orbx_handle *h, const orbx_keypoint *k1, const uint8_t *d1, int n1,
                                                      const orbx_keypoint *k2, const uint8_t *d2, int n2,
                                                      const float *bounds4, float *prev_matched, int window,
                                                      float nnratio, int check_orientation, int32_t *matches12,
                                                      int *nmatches_out) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (n1 < 0 || n2 < 0 || !bounds4 || !matches12 || !nmatches_out || (n1 > 0 && (!k1 || !d1 || !prev_matched)) ||
        (n2 > 0 && (!k2 || !d2)))
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    const int HISTO = 30, TH_LOW_ = 50;
    *nmatches_out = 0;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    // GPU: GetFeaturesInArea(vbPrevMatched[i1], windowSize, level1, level1) + DescriptorDistance for every level-0 feature
    // of F1, as ordered candidate lists (k_grid_build + k_gate); features of other levels never reach the loop (:603-605)
    std::vector<DGateQuery> gq((size_t)n1);
    for (int i1 = 0; i1 < n1; ++i1) {
        const int level1 = k1[i1].octave;
        gq[i1].x = prev_matched[2 * i1]; gq[i1].y = prev_matched[2 * i1 + 1];
        gq[i1].r = level1 > 0 ? -1.0f : (float)window;
        gq[i1].min_level = level1; gq[i1].max_level = level1;
    }
    OrbxGateLists gl;
    { const orbx_status st = orbx_gate_lists(h, k2, d2, n2, bounds4[0], bounds4[1], bounds4[2], bounds4[3], gq.data(), d1, n1, gl);
      if (st != ORBX_OK) return st; }
    int nmatches = 0;
    std::vector<std::vector<int>> rotHist(HISTO);
    const float factor = HISTO / 360.0f;               // fork value (src/ORBmatcher.cc:583)
    std::vector<int> matchedDist(n2, INT_MAX), matches21(n2, -1);
    for (int i1 = 0; i1 < n1; ++i1) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        const int nc = gl.count(i1);
        if (nc == 0) continue;
        const uint32_t *cl = gl.list(i1);
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; ++c) {
            const int i2 = OrbxGateLists::idx(cl[c]), dist = OrbxGateLists::dist(cl[c]);
            if (matchedDist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW_ && (float)bestDist < (float)bestDist2 * nnratio) {
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matchedDist[bestIdx2] = bestDist;
            nmatches++;
            if (check_orientation) {
                float rot = k1[i1].angle - k2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO) bin = 0;
                if (bin >= 0 && bin < HISTO) rotHist[bin].push_back(i1);
            }
        }
    }
    if (check_orientation) {
        int32_t sizes[30]; int i1, i2, i3;
        for (int i = 0; i < HISTO; ++i) sizes[i] = (int)rotHist[i].size();
        orbx_three_maxima(sizes, HISTO, &i1, &i2, &i3);
        for (int i = 0; i < HISTO; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx1 : rotHist[i])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < n1; ++i)
        if (matches12[i] >= 0) { prev_matched[2 * i] = k2[matches12[i]].x; prev_matched[2 * i + 1] = k2[matches12[i]].y; }
    *nmatches_out = nmatches;
    return ORBX_OK;
}

// ---------------------------------------------------------------- a16: ComputeStereoMatches (src/Frame.cc:880-1176)
extern "C" orbx_status orbx_stereo_match(orbx_handle *hl, orbx_handle *hr, int frame_left, int frame_right,
                                         const orbx_keypoint *kl, const uint8_t *dl, int nl, const orbx_keypoint *kr,
                                         const uint8_t *dr, int nr, float mb, float mbf, float *u_right, float *depth,
                                         int *nmatches_out) {
    if (!hl || !hr || hl->host_only || hr->host_only) return fail(ORBX_BAD_ARGUMENT, "two device handles are required");
    if (nl < 0 || nr < 0 || nr > 65535 || (nl > 0 && (!kl || !dl || !u_right || !depth)) || (nr > 0 && (!kr || !dr)) ||
        !(mb > 0.f))
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    orbx_status st = check_level(hl, frame_left, 0);
    if (st != ORBX_OK) return st;
    st = check_level(hr, frame_right, 0);
    if (st != ORBX_OK) return st;
    if (hl->dev != hr->dev || hl->geom.width != hr->geom.width || hl->geom.height != hr->geom.height ||
        hl->p.nlevels != hr->p.nlevels || hl->p.scale_factor != hr->p.scale_factor)
        return fail(ORBX_BAD_ARGUMENT, "left and right extractor must share device, image size and pyramid parameters");
    if (nmatches_out) *nmatches_out = 0;
    if (nl == 0) return ORBX_OK;
    HIPCHK(hipSetDevice(hl->dev));
    HIPCHK(hipStreamSynchronize(hr->stream));   // the right pyramid is read from the left handle's stream
    OrbxStereoGeom sg;
    memset(&sg, 0, sizeof(sg));
    sg.nlevels = hl->p.nlevels; sg.nrows0 = hl->geom.lv[0].ph; sg.mb = mb; sg.mbf = mbf;
    for (int l = 0; l < sg.nlevels; ++l) {
        sg.scale[l] = hl->tab.scale[l]; sg.inv_scale[l] = hl->tab.inv_scale[l];
        sg.pw[l] = hl->geom.lv[l].pw; sg.ph[l] = hl->geom.lv[l].ph; sg.pitch[l] = hl->geom.lv[l].pitch;
        sg.off[l] = hl->geom.lv[l].off;
    }
    st = scratch_reserve(hl, pad256((size_t)nl * sizeof(orbx_keypoint)) + pad256((size_t)nl * 32) +
                                 pad256((size_t)std::max(nr, 1) * sizeof(orbx_keypoint)) + pad256((size_t)std::max(nr, 1) * 32) +
                                 3 * pad256((size_t)nl * sizeof(float)) + pad256((size_t)(sg.nrows0 + 1) * sizeof(int)) +
                                 pad256((size_t)orbx_stereo_items_per_pair(sg, std::max(nr, 1)) * sizeof(uint2)));
    if (st != ORBX_OK) return st;
    orbx_keypoint *dkl = scratch_take<orbx_keypoint>(hl, nl);
    uint8_t *ddl = scratch_take<uint8_t>(hl, (size_t)nl * 32);
    orbx_keypoint *dkr = scratch_take<orbx_keypoint>(hl, std::max(nr, 1));
    uint8_t *ddr = scratch_take<uint8_t>(hl, (size_t)std::max(nr, 1) * 32);
    float *du = scratch_take<float>(hl, nl), *dz = scratch_take<float>(hl, nl);
    int *dsad = scratch_take<int>(hl, nl);
    int *drow = scratch_take<int>(hl, (size_t)sg.nrows0 + 1);
    uint2 *ditems = scratch_take<uint2>(hl, (size_t)orbx_stereo_items_per_pair(sg, std::max(nr, 1)));
    HIPCHK(hipMemcpyAsync(dkl, kl, (size_t)nl * sizeof(orbx_keypoint), hipMemcpyHostToDevice, hl->stream));
    HIPCHK(hipMemcpyAsync(ddl, dl, (size_t)nl * 32, hipMemcpyHostToDevice, hl->stream));
    if (nr > 0) {
        HIPCHK(hipMemcpyAsync(dkr, kr, (size_t)nr * sizeof(orbx_keypoint), hipMemcpyHostToDevice, hl->stream));
        HIPCHK(hipMemcpyAsync(ddr, dr, (size_t)nr * 32, hipMemcpyHostToDevice, hl->stream));
    }
    { ProfScope ps(hl, ORBX_K_MATCH);
      orbx_launch_stereo(hl->stream, sg, dkl, ddl, nl, dkr, ddr, nr, hl->d_pyr + (size_t)frame_left * hl->geom.pyr_bytes,
                         hr->d_pyr + (size_t)frame_right * hr->geom.pyr_bytes, du, dz, dsad, drow, ditems); }
    std::vector<int> sad(nl);
    HIPCHK(hipMemcpyAsync(u_right, du, (size_t)nl * sizeof(float), hipMemcpyDeviceToHost, hl->stream));
    HIPCHK(hipMemcpyAsync(depth, dz, (size_t)nl * sizeof(float), hipMemcpyDeviceToHost, hl->stream));
    HIPCHK(hipMemcpyAsync(sad.data(), dsad, (size_t)nl * sizeof(int), hipMemcpyDeviceToHost, hl->stream));
    hipError_t e = hipStreamSynchronize(hl->stream);
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    // median cut (:1160-1175): sort (SAD, index), drop everything at or above 1.5 * 1.4 * median
    std::vector<std::pair<int, int>> v;
    for (int i = 0; i < nl; ++i) if (sad[i] >= 0) v.push_back({sad[i], i});
    int kept = (int)v.size();
    if (!v.empty()) {
        std::sort(v.begin(), v.end());
        const float median = (float)v[v.size() / 2].first;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = (int)v.size() - 1; i >= 0; --i) {
            if ((float)v[i].first < thDist) break;
            u_right[v[i].second] = -1; depth[v[i].second] = -1; --kept;
        }
    }
    if (nmatches_out) *nmatches_out = kept;
    return ORBX_OK;
}

// Batched, device-resident Frame::ComputeStereoMatches: pair p = frame p of the last extraction of `hl` (left) and of
// `hr` (right); keypoints / descriptors / counts are the device buffers orbx_extract_batch_device filled (stride `cap`
// records per frame).  The median cut (:1160-1175) also runs on the device (k_stereo_cut), nothing returns to the host.
extern "C" orbx_status orbx_stereo_match_batch_device(orbx_handle *hl, orbx_handle *hr, int npairs, const orbx_keypoint *d_kl,
                                                      const uint8_t *d_dl, const int32_t *d_nl, const orbx_keypoint *d_kr,
                                                      const uint8_t *d_dr, const int32_t *d_nr, int cap, float mb, float mbf,
                                                      float *d_u_right, float *d_depth, int32_t *d_nmatches) {
    if (!hl || !hr || hl->host_only || hr->host_only) return fail(ORBX_BAD_ARGUMENT, "two device handles are required");
    if (npairs <= 0 || cap <= 0 || cap > 65535 || !d_kl || !d_dl || !d_nl || !d_kr || !d_dr || !d_nr || !d_u_right || !d_depth ||
        !d_nmatches || !(mb > 0.f))
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    // hl == hr: both eyes went through ONE batch of 2 * npairs images (left images first): the right pyramids are frames
    // npairs .. 2 npairs - 1 of that handle (half as many launches per stereo frame as two handles need)
    const bool one_batch = hl == hr;
    orbx_status st = check_level(hl, (one_batch ? 2 * npairs : npairs) - 1, 0);
    if (st != ORBX_OK) return st;
    st = check_level(hr, npairs - 1, 0);
    if (st != ORBX_OK) return st;
    if (hl->dev != hr->dev || hl->geom.width != hr->geom.width || hl->geom.height != hr->geom.height ||
        hl->p.nlevels != hr->p.nlevels || hl->p.scale_factor != hr->p.scale_factor)
        return fail(ORBX_BAD_ARGUMENT, "left and right extractor must share device, image size and pyramid parameters");
    HIPCHK(hipSetDevice(hl->dev));
    // The reference extracts the two eyes on two threads (src/Frame.cc:158-168); here that is two handles on two streams.  The
    // match runs on the left stream: it waits for the right stream's work so far (an event, no host synchronisation) ...
    const bool two_streams = hr->stream != hl->stream;
    if (two_streams) {
        HIPCHK(hipEventRecord(hr->ev_stereo, hr->stream));
        HIPCHK(hipStreamWaitEvent(hl->stream, hr->ev_stereo, 0));
    }
    OrbxStereoGeom sg;
    memset(&sg, 0, sizeof(sg));
    sg.nlevels = hl->p.nlevels; sg.nrows0 = hl->geom.lv[0].ph; sg.mb = mb; sg.mbf = mbf;
    for (int l = 0; l < sg.nlevels; ++l) {
        sg.scale[l] = hl->tab.scale[l]; sg.inv_scale[l] = hl->tab.inv_scale[l];
        sg.pw[l] = hl->geom.lv[l].pw; sg.ph[l] = hl->geom.lv[l].ph; sg.pitch[l] = hl->geom.lv[l].pitch;
        sg.off[l] = hl->geom.lv[l].off;
    }
    const size_t ipp = (size_t)orbx_stereo_items_per_pair(sg, cap);
    st = scratch_reserve(hl, pad256((size_t)npairs * cap * sizeof(int)) + pad256((size_t)npairs * (sg.nrows0 + 1) * sizeof(int)) +
                                 pad256((size_t)npairs * ipp * sizeof(uint2)));
    if (st != ORBX_OK) return st;
    int *dsad = scratch_take<int>(hl, (size_t)npairs * cap);
    int *drow = scratch_take<int>(hl, (size_t)npairs * (sg.nrows0 + 1));
    uint2 *ditems = scratch_take<uint2>(hl, (size_t)npairs * ipp);
    { ProfScope ps(hl, ORBX_K_MATCH);
      orbx_launch_stereo_batch(hl->stream, sg, npairs, cap, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, hl->d_pyr,
                               hr->d_pyr + (one_batch ? (size_t)npairs * (size_t)hl->geom.pyr_bytes : 0),
                               (long long)hl->geom.pyr_bytes, d_u_right, d_depth, dsad, d_nmatches, drow, ditems); }
    if (two_streams) {   // ... and whatever the right stream does next (the next pair's pyramids) waits for the match that reads this one's
        HIPCHK(hipEventRecord(hl->ev_stereo, hl->stream));
        HIPCHK(hipStreamWaitEvent(hr->stream, hl->ev_stereo, 0));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    return ORBX_OK;
}

// ---------------------------------------------------------------- a14: SearchByProjection(Frame&, const Frame&, th, bMono)
// (src/ORBmatcher.cc:1702-1871; caller TrackWithMotionModel src/Tracking.cc:1430,1445).  Host: projection (fp32, same
// operation order as the reference incl. the contraction selected by fp_mode).  GPU: grid of the current frame, the
// windows' candidates and their Hamming distances to the MapPoints' representative descriptors (orbx_gate_lists).
// Host: the occupancy rule and the rotation histogram over those lists -- order-dependent.
static inline float orbx_gemm3(const float *a, const float *b, float c) {  // cv::gemm 3x3 * 3x1 float special case
    const float t = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    return (float)((double)t * 1.0 + (double)c * 1.0);
}

extern "C" orbx_status orbx_search_by_projection_frame(orbx_handle *h, const orbx_frame_view *cur,
                                                       const orbx_last_frame_view *last, float th, int mono,
                                                       int check_orientation, int32_t *matched_last, int *nmatches_out) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (!cur || !last || !matched_last || !nmatches_out || cur->n < 0 || last->n < 0)
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    const int nc = cur->n, nl = last->n, HISTO = 30, TH_HIGH_ = 100;
    *nmatches_out = 0;
    for (int i = 0; i < nc; ++i) matched_last[i] = -1;
    if (nc == 0 || nl == 0) return ORBX_OK;
    if (!cur->keys_un || !cur->desc || !cur->u_right || !last->keys_un || !last->has_map_point || !last->world_pos ||
        !last->mp_desc || !last->observations)
        return fail(ORBX_BAD_ARGUMENT, "null frame field");
    const bool fma_mode = h->p.fp_mode == ORBX_FP_GCC_FMA;
    float Rcw[9], tcw[3], Rlw[9], tlw[3], twc[3], tlc[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { Rcw[3 * r + c] = cur->Tcw[4 * r + c]; Rlw[3 * r + c] = last->Tcw[4 * r + c]; }
        tcw[r] = cur->Tcw[4 * r + 3]; tlw[r] = last->Tcw[4 * r + 3];
    }
    for (int r = 0; r < 3; ++r) {
        const float t = Rcw[0 + r] * tcw[0] + Rcw[3 + r] * tcw[1] + Rcw[6 + r] * tcw[2];
        twc[r] = (float)((double)t * -1.0);
    }
    for (int r = 0; r < 3; ++r) tlc[r] = orbx_gemm3(&Rlw[3 * r], twc, tlw[r]);
    const bool bForward = tlc[2] > cur->mb && !mono, bBackward = -tlc[2] > cur->mb && !mono;
    // pass 1 (host, independent of the selection state): project every MapPoint of the last frame, derive its window and
    // octave band (:1742-1783).  pass 2 (GPU): the windows' candidates + Hamming distances.  pass 3 (host): the selection.
    std::vector<DGateQuery> gq((size_t)nl);
    std::vector<float> q_invz((size_t)nl, 0.f), q_u((size_t)nl, 0.f);
    for (int i = 0; i < nl; ++i) {
        gq[i].r = -1.0f; gq[i].x = gq[i].y = 0.f; gq[i].min_level = gq[i].max_level = -1;
        if (!last->has_map_point[i]) continue;
        float pc[3];
        for (int r = 0; r < 3; ++r) pc[r] = orbx_gemm3(&Rcw[3 * r], last->world_pos + 3 * (size_t)i, tcw[r]);
        const float invzc = (float)(1.0 / pc[2]);
        if (invzc < 0) continue;
        float u, v;
        if (fma_mode) { u = std::fmaf(cur->fx * pc[0], invzc, cur->cx); v = std::fmaf(cur->fy * pc[1], invzc, cur->cy); }
        else { u = cur->fx * pc[0] * invzc + cur->cx; v = cur->fy * pc[1] * invzc + cur->cy; }
        if (u < cur->min_x || u > cur->max_x || v < cur->min_y || v > cur->max_y) continue;
        const int oct = last->keys_un[i].octave;
        if (oct < 0 || oct >= h->p.nlevels) continue;
        gq[i].x = u; gq[i].y = v; gq[i].r = th * h->tab.scale[oct];
        if (bForward) { gq[i].min_level = oct; gq[i].max_level = -1; }
        else if (bBackward) { gq[i].min_level = 0; gq[i].max_level = oct; }
        else { gq[i].min_level = oct - 1; gq[i].max_level = oct + 1; }
        q_invz[i] = invzc; q_u[i] = u;
    }
    OrbxGateLists gl;
    { const orbx_status st = orbx_gate_lists(h, cur->keys_un, cur->desc, nc, cur->min_x, cur->max_x, cur->min_y, cur->max_y,
                                             gq.data(), last->mp_desc, nl, gl);
      if (st != ORBX_OK) return st; }
    std::vector<std::vector<int>> rotHist(HISTO);
    const float factor = HISTO / 360.0f;  // fork value (src/ORBmatcher.cc:1713)
    int nmatches = 0;
    for (int i = 0; i < nl; ++i) {
        if (gq[i].r < 0.f) continue;
        const float invzc = q_invz[i], u = q_u[i], radius = gq[i].r;
        const int ncand = gl.count(i);
        if (ncand == 0) continue;
        const uint32_t *cl = gl.list(i);
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < ncand; ++c) {
            const int i2 = OrbxGateLists::idx(cl[c]);
            if (matched_last[i2] >= 0 && last->observations[matched_last[i2]] > 0) continue;
            if (cur->u_right[i2] > 0) {
                const float ur = fma_mode ? std::fmaf(-cur->mbf, invzc, u) : u - cur->mbf * invzc;
                if (fabsf(ur - cur->u_right[i2]) > radius) continue;
            }
            const int dist = OrbxGateLists::dist(cl[c]);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH_) {
            matched_last[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = last->keys_un[i].angle - cur->keys_un[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO) bin = 0;
                if (bin >= 0 && bin < HISTO) rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int32_t sizes[30]; int i1, i2, i3;
        for (int i = 0; i < HISTO; ++i) sizes[i] = (int)rotHist[i].size();
        orbx_three_maxima(sizes, HISTO, &i1, &i2, &i3);
        for (int i = 0; i < HISTO; ++i)
            if (i != i1 && i != i2 && i != i3)
                for (int idx2 : rotHist[i]) { matched_last[idx2] = -1; nmatches--; }
    }
    *nmatches_out = nmatches;
    return ORBX_OK;
}

// ---------------------------------------------------------------- (f)1: SearchByProjection(Frame&, vector<MapPoint*>&, th)
// (src/ORBmatcher.cc:69-184, RadiusByViewingCos :187-194; caller Tracking::SearchLocalPoints src/Tracking.cc:1953) --
// the largest per-frame matcher load in steady state.  GPU: the frame's grid, every in-view MapPoint's window candidates
// and their Hamming distances (orbx_gate_lists).  Host: occupancy rule, level-aware ratio test (order-dependent).
extern "C" orbx_status orbx_search_by_projection_mappoints(orbx_handle *h, const orbx_frame_view *frame,
                                                           const int32_t *frame_observations,
                                                           const orbx_mappoint_view *mps, float th, float nnratio,
                                                           int32_t *assigned, int *nmatches_out) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (!frame || !mps || !assigned || !nmatches_out || frame->n < 0 || mps->n < 0) return fail(ORBX_BAD_ARGUMENT, "bad argument");
    const int nf = frame->n, nmp = mps->n, TH_HIGH_ = 100;
    *nmatches_out = 0;
    for (int i = 0; i < nf; ++i) assigned[i] = -1;
    if (nf == 0 || nmp == 0) return ORBX_OK;
    if (!frame->keys_un || !frame->desc || !frame->u_right || !frame_observations || !mps->in_view || !mps->proj ||
        !mps->level || !mps->view_cos || !mps->desc || !mps->observations)
        return fail(ORBX_BAD_ARGUMENT, "null field");
    std::vector<DGateQuery> gq((size_t)nmp);
    bool any = false;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < nmp; ++iMP) {
        gq[iMP].x = mps->proj[3 * iMP]; gq[iMP].y = mps->proj[3 * iMP + 1]; gq[iMP].r = -1.0f;
        gq[iMP].min_level = gq[iMP].max_level = -1;
        if (!mps->in_view[iMP]) continue;
        const int lvl = mps->level[iMP];
        if (lvl < 0 || lvl >= h->p.nlevels) continue;
        float r = mps->view_cos[iMP] > 0.998 ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        gq[iMP].r = r * h->tab.scale[lvl];
        gq[iMP].min_level = lvl - 1; gq[iMP].max_level = lvl;
        any = true;
    }
    if (!any) return ORBX_OK;
    OrbxGateLists gl;
    { const orbx_status st = orbx_gate_lists(h, frame->keys_un, frame->desc, nf, frame->min_x, frame->max_x, frame->min_y,
                                             frame->max_y, gq.data(), mps->desc, nmp, gl);
      if (st != ORBX_OK) return st; }
    std::vector<int> occ(frame_observations, frame_observations + nf);
    int nmatches = 0;
    for (int iMP = 0; iMP < nmp; ++iMP) {
        if (gq[iMP].r < 0.f) continue;
        const float radius = gq[iMP].r;
        const int nc = gl.count(iMP);
        if (nc == 0) continue;
        const uint32_t *cl = gl.list(iMP);
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = OrbxGateLists::idx(cl[c]);
            if (occ[idx] > 0) continue;
            if (frame->u_right[idx] > 0 && fabsf(mps->proj[3 * iMP + 2] - frame->u_right[idx]) > radius) continue;
            const int dist = OrbxGateLists::dist(cl[c]);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = frame->keys_un[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = frame->keys_un[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH_) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            assigned[bestIdx] = iMP;
            occ[bestIdx] = mps->observations[iMP];
            nmatches++;
        }
    }
    *nmatches_out = nmatches;
    return ORBX_OK;
}

// ---------------------------------------------------------------- (f)3: DBoW2 transform (Frame::ComputeBoW, src/Frame.cc:750-765)
// TemplatedVocabulary<FORB>::transform(features, BowVector&, FeatureVector&, levelsup)
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1216): the tree descent of every descriptor runs on the GPU
// (k_bow_transform); word id / weight look-ups and the two std::map accumulations (double precision, feature order) on the host.
struct orbx_vocabulary {
    int dev = 0;
    int n_nodes = 0, k = 0, L = 0, weighting = 0, scoring = 0;
    std::vector<int32_t> child_begin;
    std::vector<uint32_t> child_ids, word_id;
    std::vector<double> weight;
    int *d_child_begin = nullptr;
    uint32_t *d_child_ids = nullptr;
    uint8_t *d_desc = nullptr;
};

extern "C" void orbx_vocabulary_destroy(orbx_vocabulary *v) {
    if (!v) return;
    hipSetDevice(v->dev);
    hipFree(v->d_child_begin); hipFree(v->d_child_ids); hipFree(v->d_desc);
    delete v;
}

extern "C" orbx_status orbx_vocabulary_create(orbx_handle *h, const orbx_vocabulary_view *view, orbx_vocabulary **out) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (!view || !out || view->n_nodes < 1 || !view->child_begin || !view->desc || !view->weight || !view->word_id ||
        view->L < 0 || view->weighting < 0 || view->weighting > 3 || view->scoring < 0 || view->scoring > 5)
        return fail(ORBX_BAD_ARGUMENT, "bad vocabulary view");
    const int n = view->n_nodes;
    if (view->child_begin[0] != 0) return fail(ORBX_BAD_ARGUMENT, "child_begin[0] != 0");
    for (int i = 0; i < n; ++i) {
        if (view->child_begin[i + 1] < view->child_begin[i]) return fail(ORBX_BAD_ARGUMENT, "child_begin not monotonic");
        for (int c = view->child_begin[i]; c < view->child_begin[i + 1]; ++c) {
            if (!view->child_ids) return fail(ORBX_BAD_ARGUMENT, "null child_ids");
            // DBoW2 appends children after their parent (create / load*): ids grow down the tree, so the descent terminates
            if (view->child_ids[c] <= (uint32_t)i || view->child_ids[c] >= (uint32_t)n)
                return fail(ORBX_BAD_ARGUMENT, "child id out of range or not greater than its parent");
        }
        if (view->child_begin[i + 1] - view->child_begin[i] > 65535) return fail(ORBX_BAD_ARGUMENT, "too many children");
    }
    HIPCHK(hipSetDevice(h->dev));
    orbx_vocabulary *v = new orbx_vocabulary();
    v->dev = h->dev; v->n_nodes = n; v->k = view->k; v->L = view->L; v->weighting = view->weighting; v->scoring = view->scoring;
    const int nchild = view->child_begin[n];
    v->child_begin.assign(view->child_begin, view->child_begin + n + 1);
    v->child_ids.assign(view->child_ids, view->child_ids + nchild);
    v->word_id.assign(view->word_id, view->word_id + n);
    v->weight.assign(view->weight, view->weight + n);
    hipError_t e = hipMalloc(&v->d_child_begin, (size_t)(n + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&v->d_child_ids, std::max<size_t>(1, nchild) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&v->d_desc, (size_t)n * 32);
    // the vocabulary is shared by every handle of the device: upload on the creating handle's stream and wait for THAT stream
    // (the tables are complete before any other stream can be handed the object; no device-wide barrier)
    if (e == hipSuccess) e = hipMemcpyAsync(v->d_child_begin, view->child_begin, (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && nchild > 0) e = hipMemcpyAsync(v->d_child_ids, view->child_ids, (size_t)nchild * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(v->d_desc, view->desc, (size_t)n * 32, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { orbx_vocabulary_destroy(v); return fail(ORBX_HIP_ERROR, hipGetErrorString(e)); }
    *out = v;
    return ORBX_OK;
}

extern "C" orbx_status orbx_bow_transform_device(orbx_handle *h, const orbx_vocabulary *voc, int nframes, const uint8_t *d_desc,
                                                 const int32_t *d_counts, int64_t desc_frame_stride, int max_n, int levelsup,
                                                 uint32_t *d_leaf_node, uint32_t *d_node_id, int out_stride) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (!voc || nframes <= 0 || !d_desc || !d_counts || max_n <= 0 || !d_leaf_node || !d_node_id || out_stride < max_n)
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (voc->dev != h->dev) return fail(ORBX_BAD_ARGUMENT, "vocabulary lives on another device");
    HIPCHK(hipSetDevice(h->dev));
    { ProfScope ps(h, ORBX_K_MISC);
      orbx_launch_bow_transform(h->stream, nframes, max_n, voc->d_child_begin, voc->d_child_ids, voc->d_desc, voc->n_nodes, voc->L,
                                d_desc, d_counts, desc_frame_stride, levelsup, d_leaf_node, d_node_id, out_stride); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    return ORBX_OK;
}

extern "C" orbx_status orbx_bow_transform(orbx_handle *h, const orbx_vocabulary *voc, const uint8_t *desc, int n, int levelsup,
                                          uint32_t *word_id, double *weight, uint32_t *node_id) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (!voc || n < 0 || (n > 0 && (!desc || !word_id || !weight || !node_id))) return fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (n == 0) return ORBX_OK;
    HIPCHK(hipSetDevice(h->dev));
    orbx_status st = scratch_reserve(h, pad256((size_t)n * 32) + pad256(sizeof(int)) + 2 * pad256((size_t)n * sizeof(uint32_t)));
    if (st != ORBX_OK) return st;
    uint8_t *dd = scratch_take<uint8_t>(h, (size_t)n * 32);
    int *dn = scratch_take<int>(h, 1);
    uint32_t *dleaf = scratch_take<uint32_t>(h, (size_t)n), *dnid = scratch_take<uint32_t>(h, (size_t)n);
    HIPCHK(hipMemcpyAsync(dd, desc, (size_t)n * 32, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dn, &n, sizeof(int), hipMemcpyHostToDevice, h->stream));
    st = orbx_bow_transform_device(h, voc, 1, dd, dn, (int64_t)n * 32, n, levelsup, dleaf, dnid, n);
    if (st != ORBX_OK) return st;
    std::vector<uint32_t> leaf((size_t)n);
    HIPCHK(hipMemcpyAsync(leaf.data(), dleaf, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(node_id, dnid, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; ++i) {
        if (leaf[i] >= (uint32_t)voc->n_nodes) return fail(ORBX_HIP_ERROR, "transform returned an invalid node");
        word_id[i] = voc->word_id[leaf[i]];
        weight[i] = voc->weight[leaf[i]];
    }
    return ORBX_OK;
}

// BowVector / FeatureVector exactly as TemplatedVocabulary::transform fills them (:1150-1216, BowVector.cpp:40-95):
// addWeight / addIfNotExist in feature order, division by the vector size when the scoring does not normalise, L1 / L2
// normalisation in ascending word order.  Outputs are the maps flattened in key order.
extern "C" orbx_status orbx_bow_vectors(const orbx_vocabulary *voc, const uint32_t *word_id, const double *weight,
                                        const uint32_t *node_id, int n, uint32_t *bow_word, double *bow_value, int *n_bow,
                                        uint32_t *fv_node, int32_t *fv_begin, uint32_t *fv_index, int *n_fv_nodes) {
    if (!voc || n < 0 || !n_bow || !n_fv_nodes || !fv_begin || (n > 0 && (!word_id || !weight || !node_id || !bow_word || !bow_value ||
                                                                           !fv_node || !fv_index)))
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    // mustNormalize (ScoringObject.cpp): L1_NORM, CHI_SQUARE, KL, BHATTACHARYYA -> L1; L2_NORM -> L2; DOT_PRODUCT -> none
    const int norm = voc->scoring == 1 ? 2 : voc->scoring == 5 ? 0 : 1;
    const bool tf = voc->weighting == 0 || voc->weighting == 1;   // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    std::vector<int> order;
    order.reserve((size_t)n);
    for (int i = 0; i < n; ++i) if (weight[i] > 0) order.push_back(i);          // w > 0: not a stopped word
    std::vector<int> byword(order), bynode(order);
    std::stable_sort(byword.begin(), byword.end(), [&](int a, int b) { return word_id[a] < word_id[b]; });
    std::stable_sort(bynode.begin(), bynode.end(), [&](int a, int b) { return node_id[a] < node_id[b]; });
    int nb = 0;
    for (size_t i = 0; i < byword.size();) {
        size_t j = i;
        double acc = weight[byword[i]];
        for (j = i + 1; j < byword.size() && word_id[byword[j]] == word_id[byword[i]]; ++j)
            if (tf) acc += weight[byword[j]];                                      // feature order inside one word (stable sort)
        bow_word[nb] = word_id[byword[i]]; bow_value[nb] = acc; ++nb;
        i = j;
    }
    if (tf && nb > 0 && norm == 0) { const double nd = (double)nb; for (int i = 0; i < nb; ++i) bow_value[i] /= nd; }
    if (norm != 0) {
        double s = 0.0;
        if (norm == 1) for (int i = 0; i < nb; ++i) s += fabs(bow_value[i]);
        else { for (int i = 0; i < nb; ++i) s += bow_value[i] * bow_value[i]; s = sqrt(s); }
        if (s > 0.0) for (int i = 0; i < nb; ++i) bow_value[i] /= s;
    }
    *n_bow = nb;
    int nn = 0, pos = 0;
    fv_begin[0] = 0;
    for (size_t i = 0; i < bynode.size();) {
        size_t j = i;
        for (; j < bynode.size() && node_id[bynode[j]] == node_id[bynode[i]]; ++j) fv_index[pos++] = (uint32_t)bynode[j];
        fv_node[nn] = node_id[bynode[i]]; ++nn; fv_begin[nn] = pos;
        i = j;
    }
    *n_fv_nodes = nn;
    return ORBX_OK;
}

// ---------------------------------------------------------------- (f)2: Frame::UndistortKeyPoints / ComputeImageBounds
// (src/Frame.cc:770-865): cv::undistortPoints(K, D, R = I, P = K) on the device (k_undistort).  camera4 = fx, fy, cx, cy
// (mK), dist = mDistCoef (k1, k2, p1, p2[, k3 ...], up to 14).  dist[0] == 0 copies the keypoints (:772-776).
static void undistort_args(const float *camera4, const float *dist, int ndist, double *K4, double *k14, int *identity) {
    for (int i = 0; i < 4; ++i) K4[i] = (double)camera4[i];
    for (int i = 0; i < 14; ++i) k14[i] = i < ndist ? (double)dist[i] : 0.0;
    *identity = (ndist < 1 || dist[0] == 0.0f) ? 1 : 0;
}
extern "C" orbx_status orbx_undistort_keypoints_device(orbx_handle *h, int nframes, const orbx_keypoint *d_kps,
                                                       const int32_t *d_counts, int cap, const float *camera4, const float *dist,
                                                       int ndist, orbx_keypoint *d_kps_un) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (nframes <= 0 || cap <= 0 || !d_kps || !d_counts || !d_kps_un || !camera4 || ndist < 0 || ndist > 14 || (ndist > 0 && !dist))
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    HIPCHK(hipSetDevice(h->dev));
    double K4[4], k14[14]; int identity;
    undistort_args(camera4, dist, ndist, K4, k14, &identity);
    { ProfScope ps(h, ORBX_K_MISC);
      orbx_launch_undistort(h->stream, nframes, cap, cap, K4, k14, identity, d_kps, d_counts, d_kps_un); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ORBX_HIP_ERROR, hipGetErrorString(e));
    return ORBX_OK;
}
extern "C" orbx_status orbx_undistort_keypoints(orbx_handle *h, const orbx_keypoint *kps, int n, const float *camera4,
                                                const float *dist, int ndist, orbx_keypoint *kps_un) {
    if (!h || h->host_only) return fail(h ? ORBX_NO_DEVICE : ORBX_BAD_ARGUMENT, "no device handle");
    if (n < 0 || (n > 0 && (!kps || !kps_un)) || !camera4 || ndist < 0 || ndist > 14 || (ndist > 0 && !dist))
        return fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (n == 0) return ORBX_OK;
    HIPCHK(hipSetDevice(h->dev));
    orbx_status st = scratch_reserve(h, 2 * pad256((size_t)n * sizeof(orbx_keypoint)));
    if (st != ORBX_OK) return st;
    orbx_keypoint *din = scratch_take<orbx_keypoint>(h, n), *dout = scratch_take<orbx_keypoint>(h, n);
    double K4[4], k14[14]; int identity;
    undistort_args(camera4, dist, ndist, K4, k14, &identity);
    HIPCHK(hipMemcpyAsync(din, kps, (size_t)n * sizeof(orbx_keypoint), hipMemcpyHostToDevice, h->stream));
    orbx_launch_undistort(h->stream, 1, n, n, K4, k14, identity, din, nullptr, dout);
    HIPCHK(hipMemcpyAsync(kps_un, dout, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ORBX_OK;
}
// Frame::ComputeImageBounds (:830-865): the four image corners through the same kernel
extern "C" orbx_status orbx_image_bounds(orbx_handle *h, int cols, int rows, const float *camera4, const float *dist, int ndist,
                                         float *bounds4) {
    if (!bounds4 || cols <= 0 || rows <= 0) return fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (ndist < 1 || !dist || dist[0] == 0.0f) {
        bounds4[0] = 0.0f; bounds4[1] = (float)cols; bounds4[2] = 0.0f; bounds4[3] = (float)rows;
        return ORBX_OK;
    }
    orbx_keypoint c[4], u[4];
    memset(c, 0, sizeof(c));
    c[1].x = (float)cols; c[2].y = (float)rows; c[3].x = (float)cols; c[3].y = (float)rows;
    const orbx_status st = orbx_undistort_keypoints(h, c, 4, camera4, dist, ndist, u);
    if (st != ORBX_OK) return st;
    bounds4[0] = std::min(u[0].x, u[2].x); bounds4[1] = std::max(u[1].x, u[3].x);
    bounds4[2] = std::min(u[0].y, u[1].y); bounds4[3] = std::max(u[2].y, u[3].y);
    return ORBX_OK;
}
