// rtc_device.hip -- the MI355X (gfx950 / CDNA4) render path of librtc_amd.so.
//
// This translation unit holds the HOST side of the device path: scene validation and flattening,
// the persistent context, kernel launches, the batched test entry points and the scene-specialising
// JIT.  The device code itself lives in rtc_kernel_core.h (shared with the hiprtc compile).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>

#include <algorithm>
#include <array>
#include <functional>
#include <queue>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <utility>
#include <set>
#include <vector>

#include "rtc_internal.h"
#include "rtc_kernel_core.h"
#include "rtc_wavefront.h"

namespace rtc {

// Host copies of the same tables (rtc_powf_host, used by CPU tests to pin the
// restatement against the C library without a GPU).
static const PowLog2Entry h_pow_log2_tab[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
static const uint64_t h_exp2f_tab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
static const SinCosTab h_sincosf_tab[2] = RTC_SINCOSF_TAB_INIT;
static const uint32_t h_inv_pio4[24] = RTC_INV_PIO4_INIT;

// ============================================================================
//  Host side of the device path
// ============================================================================
#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(RTC_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static int usable_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static rtc_status select_device(int32_t device) {
    int nd = usable_devices();
    if (nd <= 0) return fail(RTC_ERR_NO_DEVICE, "no HIP device visible; librtc_amd has no CPU fallback");
    if (device < 0 || device >= nd) return fail(RTC_ERR_INVALID_ARG, "device %d out of range (have %d)", device, nd);
    HIP_TRY(hipSetDevice(device));
    return RTC_OK;
}

static uint32_t padded_count(uint32_t n) { return n ? (n + 7u) & ~7u : 8u; }

static rtc_status check_tuple(const float v[4], float w, const char* what) {
    if (v[3] != w) return fail(RTC_ERR_INVALID_ARG, "%s: w component must be %g (got %g)", what, (double)w, (double)v[3]);
    return RTC_OK;
}

// The four geometry records of one object (see SceneSoA).
static void pack_geometry(const rtc_object& o, float4 g[4]) {
    uint32_t bits = (uint32_t)o.kind | (o.casts_shadow ? SHAPE_CASTS : 0u) | (o.closed ? SHAPE_CLOSED : 0u);
    if (o.inv[1] == 0.0f && o.inv[2] == 0.0f && o.inv[4] == 0.0f && o.inv[6] == 0.0f && o.inv[8] == 0.0f &&
        o.inv[9] == 0.0f) {
        bits |= SHAPE_DIAG;
        if (o.inv[0] == o.inv[5] && o.inv[5] == o.inv[10]) bits |= SHAPE_UNIFORM;  // a uniform scale (shadow_fast)
    }
    float bits_f;
    std::memcpy(&bits_f, &bits, 4);
    g[0] = make_float4(o.inv[0], o.inv[5], o.inv[10], bits_f);
    g[1] = make_float4(o.inv[1], o.inv[2], o.inv[3], o.min_y);
    g[2] = make_float4(o.inv[4], o.inv[6], o.inv[7], o.max_y);
    g[3] = make_float4(o.inv[8], o.inv[9], o.inv[11], 0.0f);
}
// The three triangle records of one object (see SceneSoA::tri); e1, e2, normal as Triangle::new derives them.
static void pack_triangle(const rtc_object& o, float4 rec[3]) {
    float e1[3], e2[3], nrm[3];
    rtc_triangle_fields(o.p1, o.p2, o.p3, e1, e2, nrm);
    rec[0] = make_float4(o.p1[0], o.p1[1], o.p1[2], nrm[0]);
    rec[1] = make_float4(e1[0], e1[1], e1[2], nrm[1]);
    rec[2] = make_float4(e2[0], e2[1], e2[2], nrm[2]);
}
static rtc_status pack_uv_pattern(const rtc_uv_pattern& u, std::vector<float4>* uvrec, std::vector<float>* texels,
                                  std::vector<std::pair<const float*, size_t>>* seen);

// ---- triangle pre-culling (kernel side and error analysis: rtc_kernel_core.h tri_precull) ------------------------
// world = F * object + f for an object whose (affine) inverse is `inv`; false if the 3x3 part is singular
static bool forward_affine(const float inv[16], double F[9], double f[3]) {
    const double a[9] = {inv[0], inv[1], inv[2], inv[4], inv[5], inv[6], inv[8], inv[9], inv[10]};
    const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    F[0] = (a[4] * a[8] - a[5] * a[7]) / det, F[1] = (a[2] * a[7] - a[1] * a[8]) / det, F[2] = (a[1] * a[5] - a[2] * a[4]) / det;
    F[3] = (a[5] * a[6] - a[3] * a[8]) / det, F[4] = (a[0] * a[8] - a[2] * a[6]) / det, F[5] = (a[2] * a[3] - a[0] * a[5]) / det;
    F[6] = (a[3] * a[7] - a[4] * a[6]) / det, F[7] = (a[1] * a[6] - a[0] * a[7]) / det, F[8] = (a[0] * a[4] - a[1] * a[3]) / det;
    const double t[3] = {inv[3], inv[7], inv[11]};
    for (int r = 0; r < 3; r++) f[r] = -(F[3 * r] * t[0] + F[3 * r + 1] * t[1] + F[3 * r + 2] * t[2]);
    for (int k = 0; k < 9; k++)
        if (!std::isfinite(F[k])) return false;
    return std::isfinite(f[0]) && std::isfinite(f[1]) && std::isfinite(f[2]);
}
// largest and smallest singular value of a 3x3 matrix (cyclic Jacobi on A^T A)
static void singular_range(const double a[9], double* s_max, double* s_min) {
    double m[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) m[3 * r + c] = a[r] * a[c] + a[3 + r] * a[3 + c] + a[6 + r] * a[6 + c];
    for (int sweep = 0; sweep < 12; sweep++)
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (std::fabs(m[3 * p + q]) < 1e-300) continue;
                const double th = 0.5 * std::atan2(2.0 * m[3 * p + q], m[3 * q + q] - m[3 * p + p]), c = std::cos(th), sn = std::sin(th);
                double r[9];
                for (int k = 0; k < 9; k++) r[k] = m[k];
                for (int k = 0; k < 3; k++) {  // columns p, q
                    r[3 * k + p] = c * m[3 * k + p] - sn * m[3 * k + q];
                    r[3 * k + q] = sn * m[3 * k + p] + c * m[3 * k + q];
                }
                for (int k = 0; k < 9; k++) m[k] = r[k];
                for (int k = 0; k < 3; k++) {  // rows p, q
                    r[3 * p + k] = c * m[3 * p + k] - sn * m[3 * q + k];
                    r[3 * q + k] = sn * m[3 * p + k] + c * m[3 * q + k];
                }
                for (int k = 0; k < 9; k++) m[k] = r[k];
            }
    const double e0 = std::fmax(m[0], 0.0), e1 = std::fmax(m[4], 0.0), e2 = std::fmax(m[8], 0.0);
    *s_max = std::sqrt(std::fmax(e0, std::fmax(e1, e2)));
    *s_min = std::sqrt(std::fmin(e0, std::fmin(e1, e2)));
}
// World-space extent of a bounded object; false for unbounded / unsupported ones.  `grow_y` (object units): a cylinder's
// y range widened by that much at either end (scene box: ERROR_BUDGET.md B8).
static bool world_extent(const rtc_object& o, double lo[3], double hi[3], double grow_y = 0.0) {
    double F[9], f[3];
    if (!forward_affine(o.inv, F, f)) return false;
    double pts[8][3];
    int n = 0;
    if (o.kind == RTC_TRIANGLE) {
        for (const float* p : {o.p1, o.p2, o.p3}) pts[n][0] = p[0], pts[n][1] = p[1], pts[n][2] = p[2], n++;
    } else {
        double y0 = -1.0, y1 = 1.0;
        if (o.kind == RTC_CYLINDER || o.kind == RTC_CONE) {
            if (!std::isfinite(o.min_y) || !std::isfinite(o.max_y)) return false;
            y0 = o.min_y - grow_y, y1 = o.max_y + grow_y;
        } else if (o.kind != RTC_SPHERE && o.kind != RTC_CUBE) {
            return false;  // planes
        }
        const double rxz = o.kind == RTC_CONE ? std::fmax(std::fabs(y0), std::fabs(y1)) : 1.0;
        for (int k = 0; k < 8; k++) pts[n][0] = (k & 1) ? rxz : -rxz, pts[n][1] = (k & 2) ? y1 : y0, pts[n][2] = (k & 4) ? rxz : -rxz, n++;
    }
    for (int a = 0; a < 3; a++) lo[a] = INFINITY, hi[a] = -INFINITY;
    for (int k = 0; k < n; k++)
        for (int a = 0; a < 3; a++) {
            const double w = F[3 * a] * pts[k][0] + F[3 * a + 1] * pts[k][1] + F[3 * a + 2] * pts[k][2] + f[a];
            if (!std::isfinite(w)) return false;
            lo[a] = std::fmin(lo[a], w), hi[a] = std::fmax(hi[a], w);
        }
    return true;
}
// The pre-culling box of one triangle (see tri_precull for the derivation of P); `d_world`: bound on the distance
// between a pre-culling ray's origin and the triangle.  Leaves rec[0].w = 0 when no safe box exists.
struct TransformFacts {  // of the last object's inverse transform: the triangles of a mesh share theirs
    float inv[16];
    bool valid = false, usable = false;
    double F[9], f[3], s_max, s_min;  // object = A * world: object lengths are within [s_min, s_max] times world lengths
};
static void triangle_box(const rtc_object& o, const float4 tri[3], double d_world, double guard, double pad_scale, TransformFacts* tf,
                         float4 rec[3]) {
    rec[0] = rec[1] = rec[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!tf->valid || std::memcmp(tf->inv, o.inv, sizeof(tf->inv)) != 0) {
        std::memcpy(tf->inv, o.inv, sizeof(tf->inv));
        tf->valid = true;
        tf->usable = forward_affine(o.inv, tf->F, tf->f);
        if (tf->usable) {
            const double A[9] = {o.inv[0], o.inv[1], o.inv[2], o.inv[4], o.inv[5], o.inv[6], o.inv[8], o.inv[9], o.inv[10]};
            singular_range(A, &tf->s_max, &tf->s_min);
            tf->usable = tf->s_min > 0.0 && std::isfinite(tf->s_max);
        }
    }
    if (!tf->usable) return;
    const double *F = tf->F, *f = tf->f, s_max = tf->s_max, s_min = tf->s_min;
    const double kappa = s_max / s_min;
    // the triangle the kernel intersects: p1, p1 + e1, p1 + e2 with the stored (f32) edges
    const double p[3][3] = {{tri[0].x, tri[0].y, tri[0].z},
                            {(double)tri[0].x + tri[1].x, (double)tri[0].y + tri[1].y, (double)tri[0].z + tri[1].z},
                            {(double)tri[0].x + tri[2].x, (double)tri[0].y + tri[2].y, (double)tri[0].z + tri[2].z}};
    auto len = [](const double v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    const double e1[3] = {tri[1].x, tri[1].y, tri[1].z}, e2[3] = {tri[2].x, tri[2].y, tri[2].z};
    const double e3[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
    const double l1 = len(e1), l2 = len(e2), l3 = len(e3);
    if (!(l1 > 0.0 && l2 > 0.0 && l3 > 0.0)) return;
    auto angle_sin_half = [](double a, double b, double c) {  // sin(angle/2) at the vertex between sides a, b opposite c
        const double cosv = std::fmax(-1.0, std::fmin(1.0, (a * a + b * b - c * c) / (2.0 * a * b)));
        return std::sqrt(std::fmax(0.0, (1.0 - cosv) / 2.0));
    };
    const double sh = std::fmin(angle_sin_half(l1, l2, l3), std::fmin(angle_sin_half(l1, l3, l2), angle_sin_half(l2, l3, l1)));
    if (!(sh > 1e-3)) return;  // a sliver: its rejections are not robust at any useful padding
    const double rho = std::fmax(1.0, (l1 + l2) / l3), s_obj = std::fmax(l1, std::fmax(l2, l3));
    const double eps = 5.9604644775390625e-08;  // 2^-24
    const double d_obj = s_max * d_world;
    const double pad_obj = 4.0 * eps * rho * kappa * (8.0 * d_obj + 10.0 * s_obj) / (guard * sh);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, w[3][3];
    for (int k = 0; k < 3; k++)
        for (int a = 0; a < 3; a++) {
            w[k][a] = F[3 * a] * p[k][0] + F[3 * a + 1] * p[k][1] + F[3 * a + 2] * p[k][2] + f[a];
            lo[a] = std::fmin(lo[a], w[k][a]), hi[a] = std::fmax(hi[a], w[k][a]);
        }
    const double u[3] = {w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2]}, v[3] = {w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2]};
    double nrm[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    const double nl = len(nrm);
    if (!(nl > 0.0) || !std::isfinite(nl)) return;
    double big = 0.0;
    for (int a = 0; a < 3; a++) big = std::fmax(big, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
    // object distance >= s_min * world distance; 1e-5 of the coordinates covers the slab test's own rounding in f32
    const double pad = pad_scale * (pad_obj / s_min + 1e-5 * (big + d_world));
    if (!std::isfinite(pad)) return;
    rec[0] = make_float4((float)(lo[0] - pad), (float)(lo[1] - pad), (float)(lo[2] - pad), 1.0f);
    rec[1] = make_float4((float)(hi[0] + pad), (float)(hi[1] + pad), (float)(hi[2] + pad), 0.0f);
    rec[2] = make_float4((float)(nrm[0] / nl), (float)(nrm[1] / nl), (float)(nrm[2] / nl), 0.0f);
}
// ---- environment switches --------------------------------------------------------------------------------------------
// Every switch the library takes from the environment, read ONCE -- when a context is created (rtc_ctx_create; the
// stateless batched entry points read them per call) -- and kept with the context: a switch changed later does not reach
// a context that exists, and nothing on the launch path calls getenv.  Two classes:
//   * policy switches (always compiled in): on / off of a shortcut or a choice the library makes by itself.  None of them
//     can change an image -- every one has a whole-frame on / off test -- only how fast it is produced;
//   * development switches (RTC_DEV_ENV: compiled in only with -DRTC_DEV_SWITCHES, i.e. into librtc_amd_dev.so, which the
//     tools and a few tests load): substitute kernel source or compiler flags, pin tuning constants, or -- RTC_AMD_TRI_NAIVE --
//     deliberately break a guarantee so that a test can show it notices.  The shipped library does not even hold their names.
#ifdef RTC_DEV_SWITCHES
#define RTC_DEV_ENV(name) std::getenv(name)
#else
#define RTC_DEV_ENV(name) ((const char*)nullptr)
#endif
struct Policy {
    int specialise = 2;  // RTC_AMD_SPECIALIZE: 0 never, 1 always (a failed compile is an error), 2 by frame size
    bool light_cull = true, dark = true, fast_shadow = true, cell_cull = true;  // RTC_AMD_LIGHT_CULL / _DARK / _FAST_SHADOW / _CELL_CULL (SceneHdr::cull_flags)
    bool bvh = true, scene_box = true, gates = true, tri_precull = true, block_list = true, quiet = false;
    bool scene_tiles = true;  // RTC_AMD_SCENE_TILES: a sparse bounded scene's frames as zero-fill + its own tiles (rtc_ctx::scene_tile_mask)
    bool prune = true;    // RTC_AMD_PRUNE: groups / nodes a ray enters beyond what it still wants are left closed (for_each_object, ERROR_BUDGET.md B6)
    int clusters = -1;    // RTC_AMD_CLUSTERS: nodes over long triangle runs -- 0 never, 1 always, -1 by frame size
    int share_log2 = -1;  // RTC_AMD_SHARE_LOG2 = 0..3: lanes per pixel (log2) pinned for every frame; -1: by frame size
    int scene_rect = 1;   // RTC_AMD_SCENE_RECT: 0 never launch the scene's rectangle only, 2 whenever there is one, 1 under half the frame
    bool swizzle = true;         // RTC_AMD_SWIZZLE: a regular grid's blocks permuted within four rows (RenderArgs::swizzle)
    bool grid_feedback = true;   // RTC_AMD_GRID_FEEDBACK: ... and so do the frames of a regular grid: their blocks start longest first
    bool block_feedback = true;  // RTC_AMD_BLOCK_FEEDBACK: a block list's second and later frames go by the first one's wave times (refine_block_list)
    int wavefront = 0;    // RTC_AMD_WAVEFRONT=1: tree worlds are rendered by the level-by-level renderer (rtc_wavefront.h); default: never
    std::string jit_cache;  // RTC_AMD_JIT_CACHE=<dir>; "0" / "off": compiled kernels stay in memory; empty: <library dir>/jit_cache
    // development
    std::string jit_source, jit_flags;  // RTC_AMD_JIT_SOURCE=<path of rtc_kernel_core.h>, RTC_AMD_JIT_FLAGS="-D... -m..."
    bool jit_print = false, cluster_stats = false, tri_naive = false, block_order = true;
    int tree_waves = 0, reg_levels = 0, blocks_y = 0, block_s = -1, block_s_top = -1;  // (0 / -1: the library's own choice)
    uint32_t fill_wgs = 0u, tile_fill_wgs = 0u, cluster_min_run = 0u, cluster_leaf = 0u, area_share_waves = 0u, feedback_pct = 85u, feedback_down_pct = 40u, feedback_passes = 2u, feedback_max_s = 4u;
    double cluster_gmax = -1.0;

    static Policy from_env() {
        Policy p;
        auto flag = [](const char* e, bool dflt) { return (e && *e) ? e[0] != '0' : dflt; };
        auto digit = [](const char* e, int lo, int hi, int dflt) { return (e && e[0] >= '0' + lo && e[0] <= '0' + hi && !e[1]) ? e[0] - '0' : dflt; };
        if (const char* e = std::getenv("RTC_AMD_SPECIALIZE")) p.specialise = !*e ? 2 : e[0] == '0' ? 0 : e[0] == '1' ? 1 : 2;
        p.light_cull = flag(std::getenv("RTC_AMD_LIGHT_CULL"), true);
        p.dark = flag(std::getenv("RTC_AMD_DARK"), true);
        p.fast_shadow = flag(std::getenv("RTC_AMD_FAST_SHADOW"), true);
        p.cell_cull = flag(std::getenv("RTC_AMD_CELL_CULL"), true);
        p.bvh = flag(std::getenv("RTC_AMD_BVH"), true);
        p.scene_box = flag(std::getenv("RTC_AMD_SCENE_BOX"), true);
        p.gates = flag(std::getenv("RTC_AMD_GATES"), true);
        p.tri_precull = flag(std::getenv("RTC_AMD_TRI_PRECULL"), true);
        p.prune = flag(std::getenv("RTC_AMD_PRUNE"), true);
        p.scene_tiles = flag(std::getenv("RTC_AMD_SCENE_TILES"), true);
        p.block_list = flag(std::getenv("RTC_AMD_BLOCK_LIST"), true);
        p.block_feedback = flag(std::getenv("RTC_AMD_BLOCK_FEEDBACK"), true);
        p.grid_feedback = flag(std::getenv("RTC_AMD_GRID_FEEDBACK"), true);
        p.swizzle = flag(std::getenv("RTC_AMD_SWIZZLE"), true);
        p.quiet = flag(std::getenv("RTC_AMD_QUIET"), false);
        if (const char* e = std::getenv("RTC_AMD_CLUSTERS")) p.clusters = *e ? (e[0] != '0' ? 1 : 0) : -1;
        p.share_log2 = digit(std::getenv("RTC_AMD_SHARE_LOG2"), 0, 3, -1);
        if (const char* e = std::getenv("RTC_AMD_SCENE_RECT")) p.scene_rect = e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1;
        if (const char* e = std::getenv("RTC_AMD_WAVEFRONT")) p.wavefront = (*e && e[0] != '0') ? 1 : 0;
        if (const char* e = std::getenv("RTC_AMD_JIT_CACHE")) p.jit_cache = e;
        if (const char* e = RTC_DEV_ENV("RTC_AMD_JIT_SOURCE")) p.jit_source = e;
        if (const char* e = RTC_DEV_ENV("RTC_AMD_JIT_FLAGS")) p.jit_flags = e;
        p.jit_print = flag(RTC_DEV_ENV("RTC_AMD_JIT_PRINT"), false);
        p.cluster_stats = flag(RTC_DEV_ENV("RTC_AMD_CLUSTER_STATS"), false);
        p.tri_naive = flag(RTC_DEV_ENV("RTC_AMD_TRI_NAIVE"), false);
        p.block_order = flag(RTC_DEV_ENV("RTC_AMD_BLOCK_ORDER"), true);
        p.tree_waves = digit(RTC_DEV_ENV("RTC_AMD_TREE_WAVES"), 1, 8, 0);  // (0: by the scene and the frame, rtc_ctx_set_scene)
        p.reg_levels = digit(RTC_DEV_ENV("RTC_AMD_REG_LEVELS"), 0, 8, 0);
        p.blocks_y = digit(RTC_DEV_ENV("RTC_AMD_BLOCKS_Y"), 1, 8, 0);
        p.block_s = digit(RTC_DEV_ENV("RTC_AMD_BLOCK_S"), 0, 3, -1);
        p.block_s_top = digit(RTC_DEV_ENV("RTC_AMD_BLOCK_S_TOP"), 0, 3, -1);
        if (const char* e = RTC_DEV_ENV("RTC_AMD_FEEDBACK_PCT")) p.feedback_pct = std::max(1u, (uint32_t)std::atoi(e));
        if (const char* e = RTC_DEV_ENV("RTC_AMD_FEEDBACK_DOWN_PCT")) p.feedback_down_pct = (uint32_t)std::atoi(e);
        if (const char* e = RTC_DEV_ENV("RTC_AMD_FEEDBACK_PASSES")) p.feedback_passes = std::max(1u, (uint32_t)std::atoi(e));
        if (const char* e = RTC_DEV_ENV("RTC_AMD_AREA_SHARE_WAVES")) p.area_share_waves = (uint32_t)std::atoi(e);
        if (const char* e = RTC_DEV_ENV("RTC_AMD_FEEDBACK_MAX_S")) p.feedback_max_s = std::min(4u, (uint32_t)std::atoi(e));
        if (const char* e = RTC_DEV_ENV("RTC_AMD_FILL_WGS")) p.fill_wgs = std::max(1u, (uint32_t)std::atoi(e));
        if (const char* e = RTC_DEV_ENV("RTC_AMD_TILE_FILL_WGS")) p.tile_fill_wgs = std::max(1u, (uint32_t)std::atoi(e));
        if (const char* e = RTC_DEV_ENV("RTC_AMD_CLUSTER_MIN_RUN")) p.cluster_min_run = std::max(3u, (uint32_t)std::atoi(e));
        if (const char* e = RTC_DEV_ENV("RTC_AMD_CLUSTER_LEAF")) p.cluster_leaf = std::min(64u, std::max(2u, (uint32_t)std::atoi(e)));
        if (const char* e = RTC_DEV_ENV("RTC_AMD_CLUSTER_GMAX")) p.cluster_gmax = std::atof(e);
        return p;
    }
    // The share of the frame below which a scene's rectangle is launched instead of the whole grid (rtc_ctx_render).
    float scene_rect_threshold() const { return scene_rect == 0 ? 0.0f : scene_rect == 2 ? 1.01f : 0.5f; }
};

// Lanes per pixel (RenderArgs::share_log2).  Two kinds of work can be shared between the lanes of a pixel: an area light's
// cells (intensity_at), while the frame would otherwise be fewer than ~4 waves per SIMD; and, in a tree walk, the long
// runs of leaves a divided mesh leaves at every level (for_each_leaf_shared) -- there a frame's time is that of its
// slowest wave, whatever the frame's size.  RTC_AMD_SHARE_LOG2=0..3 overrides.
// (beyond 200 k waves a first frame is better off with one lane everywhere -- mesh 4096^2, 262 k: 8.1 ms with two lanes in the mesh
// tiles, 6.3 with one; here_be_dragons 4000 x 1600, 100 k: 2.9 / 3.8 -- and the feedback finds the few tiles that want more)
static uint32_t choose_share_log2_runs(uint64_t waves) { return waves <= 12000u ? 3u : waves <= 40000u ? 2u : waves <= 200000u ? 1u : 0u; }
static bool has_leaf_runs(const SceneHdr& hdr) {  // a tree walk with the long runs of leaves a divided mesh leaves: lanes can split them
    return hdr.n_trav != 0u && hdr.max_leaf_run >= 16u && !(hdr.light_kind == RTC_LIGHT_RECT && hdr.u_steps * hdr.v_steps >= 8);
}
static uint32_t choose_share_log2(const SceneHdr& hdr, uint32_t rows, const Policy& P, bool lists = true) {
    const bool area = hdr.light_kind == RTC_LIGHT_RECT && hdr.u_steps * hdr.v_steps >= 8;
    const bool runs = has_leaf_runs(hdr);
    if (!area && !runs) return 0u;
    if (P.share_log2 >= 0) return (uint32_t)P.share_log2;
    const uint64_t waves = ((uint64_t)hdr.width * rows + 63) / 64;
    // measured (tools/ab_env.py over RTC_AMD_BLOCK_S, round 3), ms with 2 / 4 / 8 lanes per pixel in the mesh tiles: here_be_dragons 1000 x 400 (6 k
    // waves) 1.33 / 0.93 / 0.77, 2000 x 800 (25 k) 1.61 / 1.41 / 1.50, 4000 x 1600 (100 k) 3.25 / 3.57 / 4.80; mesh 512 x 384
    // (3 k) 3.81 / 2.54 / 2.05, 1024^2 (16 k) 3.12 / 2.58 / 3.13, 2048^2 (65 k) 3.65 / 4.37 / 5.86, 4096^2 (262 k) 8.5 / 11.1 / 16.8
    if (runs) return choose_share_log2_runs(waves);
    // area lights, measured on soft_shadows (ms with 1 / 2 / 4 / 8 lanes per pixel): 512^2 (4 k waves) - / - / 0.119 / 0.084;
    // 1000 x 400 (6 k) - / 0.193 / 0.134 / 0.145; 700^2 (8 k) 0.396 / 0.242 / 0.158 / 0.130; 1024^2 (16 k) 0.317 / 0.204 / 0.150 / -;
    // 1536^2 (37 k) 0.346 / 0.248 / 0.308 / 0.49; 2048^2 (66 k) 0.366 / 0.38 / 0.50 / 0.81: a frame's time is its throughput or
    // its longest wave, whichever is longer, and a wave of 64 pixels x 100 samples is long
    // (`lists`: the frame's lane count is only where the feedback starts from -- rtc_device.hip refine_block_list cuts it per tile
    // from the second frame on, which pays up to larger frames: 2048^2 0.288 -> 0.250 ms, 3072^2 0.511 -> 0.525)
    const uint32_t one_lane_from = P.area_share_waves ? P.area_share_waves : (lists && P.block_feedback) ? 100000u : 50000u;
    return waves < 6144u ? 3u : waves < 24000u ? 2u : waves < one_lane_from ? 1u : 0u;
}

// An internal bounding-volume hierarchy for FLAT worlds (World.objects without GroupShapes) of many bounded objects:
// the object list is left as it is, and a traversal stream with groups of the library's own making is laid over it,
// walked by the same packet kernel as real GroupShapes.  Unlike a GroupShape's box -- which is part of the
// reference's semantics -- these boxes must never change an answer.  They cannot: every box is the hull of its
// objects' world-space bounds INFLATED by 10 % (plus an absolute epsilon), so a ray that misses a box passes at
// least 0.1 radius away from every object inside, where the exact intersection test reports a miss with a margin far
// above its rounding error -- the same argument, and the same proviso, as for light-cone culling: the
// quadratic's cancellation error grows with the ray origin's distance in radii, so the hierarchy is only built when
// no ray can start more than 100 radii from any object (origins are the camera or points on the objects).  Ties in
// hit distance go by object index in the tree kernels, so the visiting order does not matter either.
// Eligible: scale+translate-only spheres and cubes, all of them (a plane's bounds are infinite).
static bool build_flat_bvh(const rtc_scene* scene, const float cam_origin[4], std::vector<float4>* trav) {
    const uint32_t n = scene->n_objects;
    struct Box {
        float lo[3], hi[3];
    };
    std::vector<Box> box(n);
    float all_lo[3] = {cam_origin[0], cam_origin[1], cam_origin[2]}, all_hi[3] = {cam_origin[0], cam_origin[1], cam_origin[2]};
    float r_min = INFINITY;
    for (uint32_t i = 0; i < n; i++) {
        const rtc_object& o = scene->objects[i];
        if (o.kind != RTC_SPHERE && o.kind != RTC_CUBE) return false;
        const float* m = o.inv;
        if (!(m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f && m[6] == 0.0f && m[8] == 0.0f && m[9] == 0.0f)) return false;
        for (int a = 0; a < 3; a++) {
            const float g = m[5 * a], t = m[4 * a + 3];  // x_obj = g * x_world + t, |x_obj| <= 1 (sphere and cube alike)
            if (!(std::fabs(g) > 1e-20f) || !std::isfinite(g) || !std::isfinite(t)) return false;
            const float c = -t / g, h = 1.0f / std::fabs(g);  // world centre and half extent along this axis
            const float pad = 0.1f * h + 1e-4f * (std::fabs(c) + h);
            box[i].lo[a] = c - h - pad;
            box[i].hi[a] = c + h + pad;
            all_lo[a] = std::fmin(all_lo[a], c - h);
            all_hi[a] = std::fmax(all_hi[a], c + h);
            r_min = std::fmin(r_min, h);
        }
    }
    float diag2 = 0.0f;
    for (int a = 0; a < 3; a++) diag2 += (all_hi[a] - all_lo[a]) * (all_hi[a] - all_lo[a]);
    // ERROR_BUDGET.md B9: a ray is turned away from a box when it misses the box padded by 10 % of the object.  That is safe
    // while (E2) no ray starts more than 100 radii -- of the object's SMALLEST axis: object-space units -- from an object, and
    // (E1) the object-space origin M p + t is good to a hundredth of the padding: 3 u (|p| + |centre|) / r <= 1e-3.
    if (!(std::sqrt(diag2) < 100.0f * r_min)) return false;
    float far_coord = 0.0f;
    for (int a = 0; a < 3; a++) far_coord = std::fmax(far_coord, std::fmax(std::fabs(all_lo[a]), std::fabs(all_hi[a])));
    if (!(far_coord <= 2.5e3f * r_min)) return false;  // 3 u * 2 * 2.5e3 = 9e-4
    auto as_f = [](uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    // median split of the centres along the widest axis, down to two objects per group
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    struct Rec {
        static void go(std::vector<uint32_t>& ord, size_t b, size_t e, const std::vector<Box>& box, std::vector<float4>* out,
                       float (*as_f)(uint32_t)) {
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (size_t k = b; k < e; k++)
                for (int a = 0; a < 3; a++) {
                    lo[a] = std::fmin(lo[a], box[ord[k]].lo[a]);
                    hi[a] = std::fmax(hi[a], box[ord[k]].hi[a]);
                }
            const size_t head = out->size();
            float big = 0.0f;
            for (int a = 0; a < 3; a++) big = std::fmax(big, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
            out->push_back(make_float4(lo[0], lo[1], lo[2], 0.0f));
            out->push_back(make_float4(hi[0], hi[1], hi[2], 4e-3f * big));  // pruning slack, ERROR_BUDGET.md B6
            out->push_back(make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            if (e - b <= 2) {
                for (size_t k = b; k < e; k++) {
                    out->push_back(make_float4(0.0f, 0.0f, 0.0f, as_f(ord[k])));
                    out->push_back(make_float4(0.0f, 0.0f, 0.0f, TRAV_LEAF_TAG));
                    out->push_back(make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                }
            } else {
                int axis = 0;
                for (int a = 1; a < 3; a++)
                    if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
                const size_t mid = b + (e - b) / 2;
                std::nth_element(ord.begin() + b, ord.begin() + mid, ord.begin() + e, [&](uint32_t x, uint32_t y) {
                    return box[x].lo[axis] + box[x].hi[axis] < box[y].lo[axis] + box[y].hi[axis];
                });
                go(ord, b, mid, box, out, as_f);
                go(ord, mid, e, box, out, as_f);
            }
            (*out)[head].w = as_f((uint32_t)(out->size() / TRAV_STRIDE));  // skip: the entry after this subtree
        }
    };
    Rec::go(order, 0, n, box, trav, +as_f);
    return true;
}

// A hierarchy of the library's own over every long run of boxed triangle leaves (the rings divide() leaves behind, group.rs:
// 46-73: 168 / 48 / 63 / 27 ... direct children per level of a 3 k-triangle mesh, thousands for a scanned one), written into
// the entry list as NODES -- group-like entries the kernel tells from GroupShapes by e2.w > 0 and tests with node_precull.  A
// node never changes an answer: it is passed by only when tri_precull would have skipped each triangle under it, i.e. the
// ray's line misses the hull of their padded boxes AND the ray is at more than asin(TRI_GUARD) from every one of their
// planes, which the node knows through a cone around their normals (axis a, half-angle phi: |d.a| >= sin(phi +
// asin(TRI_GUARD)) |d| implies |d.n| >= TRI_GUARD |d| for every n within phi of +-a).  The run's leaves are re-ordered
// (median splits of their boxes' centres): the tree kernels resolve equal distances by object index, not by position.
// `trav` must carry the run lengths of mark_leaf_runs; the caller marks the new list again.
static void cluster_leaf_runs(std::vector<float4>* trav, double tri_guard, const Policy& P) {
    // (RTC_AMD_CLUSTER_MIN_RUN, _LEAF, _GMAX: development and tests)
    const uint32_t MIN_RUN = P.cluster_min_run ? P.cluster_min_run : 24u;
    const uint32_t LEAF = P.cluster_leaf ? P.cluster_leaf : 8u;  // triangles under a node of the lowest level, at most
    const double g_max = P.cluster_gmax >= 0.0 ? P.cluster_gmax : 0.85;  // a node whose cone lets fewer than ~15 % of all directions pass is not worth its test
    const size_t ne = trav->size() / TRAV_STRIDE;
    const std::vector<float4>& in = *trav;
    std::vector<float4> out;
    out.reserve(in.size() + in.size() / 4);
    std::vector<uint32_t> new_index(ne + 1, 0);
    std::vector<size_t> groups;  // new positions of the copied group entries (their skip indices are mapped at the end)
    auto as_f = [](uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    struct Build {
        const std::vector<float4>& in;
        std::vector<float4>& out;
        double tri_guard, g_max;
        uint32_t LEAF;
        float (*as_f)(uint32_t);
        void go(std::vector<size_t>& ord, size_t b, size_t e) {
            if (e - b <= LEAF) {
                for (size_t k = b; k < e; k++)
                    for (uint32_t r = 0; r < TRAV_STRIDE; r++) out.push_back(in[TRAV_STRIDE * ord[k] + r]);
                return;
            }
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            double axis[3] = {0.0, 0.0, 0.0};
            for (size_t k = b; k < e; k++) {
                const float4 &mn = in[TRAV_STRIDE * ord[k]], &mx = in[TRAV_STRIDE * ord[k] + 1], &nr = in[TRAV_STRIDE * ord[k] + 2];
                const double a[3] = {mn.x, mn.y, mn.z}, c[3] = {mx.x, mx.y, mx.z}, n[3] = {nr.x, nr.y, nr.z};
                for (int j = 0; j < 3; j++) {
                    lo[j] = std::fmin(lo[j], a[j]), hi[j] = std::fmax(hi[j], c[j]);
                    clo[j] = std::fmin(clo[j], a[j] + c[j]), chi[j] = std::fmax(chi[j], a[j] + c[j]);
                }
                const double sgn = (n[0] * axis[0] + n[1] * axis[1] + n[2] * axis[2]) < 0.0 ? -1.0 : 1.0;  // n and -n are one plane
                for (int j = 0; j < 3; j++) axis[j] += sgn * n[j];
            }
            const double al = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
            double g = 2.0;  // sin(phi + asin(tri_guard)), or "no node"
            if (al > 1e-6) {
                double cmin = 1.0;  // cosine of the cone's half-angle
                for (size_t k = b; k < e; k++) {
                    const float4& nr = in[TRAV_STRIDE * ord[k] + 2];
                    const double nl = std::sqrt((double)nr.x * nr.x + (double)nr.y * nr.y + (double)nr.z * nr.z);
                    const double c = std::fabs(nr.x * axis[0] + nr.y * axis[1] + nr.z * axis[2]) / (al * nl);
                    cmin = std::fmin(cmin, nl > 0.5 ? c : 0.0);
                }
                const double phi = std::acos(std::fmin(1.0, cmin)) + 1e-3;  // 1e-3 rad: the records' normals are f32, so is the kernel's dot product
                const double lim = phi + std::asin(std::fmin(1.0, tri_guard));
                if (lim < 1.5) g = std::sin(lim);
            }
            const bool node = g <= g_max && std::isfinite(lo[0] + lo[1] + lo[2] + hi[0] + hi[1] + hi[2]);
            const size_t head = out.size();
            if (node) {
                float big = 0.0f;
                for (int j = 0; j < 3; j++) big = std::fmax(big, (float)std::fmax(std::fabs(lo[j]), std::fabs(hi[j])));
                // hull of boxes that are f32 already: exact.  The third record: |d.a'| >= tri_guard |d| <=> |d.a| >= g |d|
                const double scale = (tri_guard > 0.0 ? tri_guard : 1.0) / (g * al);
                out.push_back(make_float4((float)lo[0], (float)lo[1], (float)lo[2], 0.0f));
                out.push_back(make_float4((float)hi[0], (float)hi[1], (float)hi[2], 1e-3f * big));
                out.push_back(make_float4((float)(axis[0] * scale), (float)(axis[1] * scale), (float)(axis[2] * scale), 1.0f));
            }
            int ax = 0;
            for (int j = 1; j < 3; j++)
                if (chi[j] - clo[j] > chi[ax] - clo[ax]) ax = j;
            const size_t mid = b + (e - b) / 2;
            std::nth_element(ord.begin() + b, ord.begin() + mid, ord.begin() + e, [&](size_t x, size_t y) {
                const float4 &xa = in[TRAV_STRIDE * x], &xb = in[TRAV_STRIDE * x + 1], &ya = in[TRAV_STRIDE * y], &yb = in[TRAV_STRIDE * y + 1];
                const float cx = ax == 0 ? xa.x + xb.x : ax == 1 ? xa.y + xb.y : xa.z + xb.z;
                const float cy = ax == 0 ? ya.x + yb.x : ax == 1 ? ya.y + yb.y : ya.z + yb.z;
                return cx < cy || (cx == cy && x < y);
            });
            go(ord, b, mid);
            go(ord, mid, e);
            if (node) out[head].w = as_f((uint32_t)(out.size() / TRAV_STRIDE));
        }
    };
    Build build = {in, out, tri_guard, g_max, LEAF, +as_f};
    for (size_t e = 0; e < ne;) {
        new_index[e] = (uint32_t)(out.size() / TRAV_STRIDE);
        if (!(in[TRAV_STRIDE * e + 1].w < 0.0f)) {  // a group
            groups.push_back(out.size() / TRAV_STRIDE);
            for (uint32_t r = 0; r < TRAV_STRIDE; r++) out.push_back(in[TRAV_STRIDE * e + r]);
            e++;
            continue;
        }
        const uint32_t w = (uint32_t)in[TRAV_STRIDE * e + 2].w, run = std::max(1u, w >> 3);
        const size_t end = std::min(ne, e + run);
        bool all_boxed = (w & 4u) != 0u;  // a mesh run: boxed triangles under one transform
        for (size_t k = e; k < end && all_boxed; k++) all_boxed = in[TRAV_STRIDE * k + 1].w == TRAV_BOXED_LEAF_TAG;
        if (all_boxed && end - e >= MIN_RUN) {
            std::vector<size_t> ord(end - e);
            for (size_t k = e; k < end; k++) ord[k - e] = k;
            build.go(ord, 0, ord.size());
        } else {
            for (size_t k = e; k < end; k++)
                for (uint32_t r = 0; r < TRAV_STRIDE; r++) out.push_back(in[TRAV_STRIDE * k + r]);
        }
        for (size_t k = e + 1; k < end; k++) new_index[k] = new_index[e];  // (nothing points into a run)
        e = end;
    }
    new_index[ne] = (uint32_t)(out.size() / TRAV_STRIDE);
    if (P.cluster_stats) {  // development
        size_t leaves = 0, nodes = 0, in_nodes = 0;
        double gsum = 0.0;
        for (size_t e = 0; e < out.size() / TRAV_STRIDE; e++) {
            if (out[TRAV_STRIDE * e + 1].w < 0.0f) leaves++;
            else if (out[TRAV_STRIDE * e + 2].w > 0.0f) {
                nodes++;
                const float4 a = out[TRAV_STRIDE * e + 2];
                gsum += tri_guard / std::sqrt((double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z);
                uint32_t skip;
                std::memcpy(&skip, &out[TRAV_STRIDE * e].w, 4);
                in_nodes += skip - e - 1;
            }
        }
        std::fprintf(stderr, "cluster_leaf_runs: %zu entries -> %zu; %zu groups, %zu leaves, %zu nodes (mean g %.3f, entries under nodes incl. nested %zu)\n",
                     ne, out.size() / TRAV_STRIDE, groups.size(), leaves, nodes, nodes ? gsum / nodes : 0.0, in_nodes);
    }
    for (size_t gpos : groups) {
        uint32_t skip;
        std::memcpy(&skip, &out[TRAV_STRIDE * gpos].w, 4);
        out[TRAV_STRIDE * gpos].w = as_f(new_index[std::min<size_t>(skip, ne)]);
    }
    trav->swap(out);
}

// Every leaf entry learns how many consecutive leaf entries OF THE SAME GROUP start with it, and whether those are a MESH
// run -- boxed triangle leaves that all share one inverse transform and one kind / flags
// word, which is what the children of a parsed, transformed, divided mesh are (group.rs:39-44 bakes the group's transform
// into every child) -- see the kernel's trav_run / trav_mesh_run / trav_more, which share e2.w (it holds `more`, 0..3,
// on entry).  A run ends where a group ends: the leaves after a nested group's last child belong to rays that may not
// have entered that group at all.  Returns the longest run.
static uint32_t mark_leaf_runs(std::vector<float4>* trav, const rtc_scene* scene) {
    const size_t ne = trav->size() / TRAV_STRIDE;
    std::vector<char> ends_subtree(ne + 1, 0);  // [e]: some group's subtree ends right before entry e
    for (size_t e = 0; e < ne; e++)
        if (!((*trav)[TRAV_STRIDE * e + 1].w < 0.0f)) {
            uint32_t skip;
            std::memcpy(&skip, &(*trav)[TRAV_STRIDE * e].w, 4);
            if (skip <= ne) ends_subtree[skip] = 1;
        }
    auto object_of = [&](size_t e) {
        uint32_t idx;
        std::memcpy(&idx, &(*trav)[TRAV_STRIDE * e].w, 4);
        return idx;
    };
    auto same_mesh = [&](size_t a, size_t b) {  // entry b continues the mesh run of entry a
        const uint32_t ia = object_of(a), ib = object_of(b);
        if (ia >= scene->n_objects || ib >= scene->n_objects) return false;
        const rtc_object &oa = scene->objects[ia], &ob = scene->objects[ib];
        return oa.kind == ob.kind && (oa.casts_shadow != 0) == (ob.casts_shadow != 0) && std::memcmp(oa.inv, ob.inv, sizeof(oa.inv)) == 0;
    };
    uint32_t longest = 0, run = 0;
    bool mesh = false;
    for (size_t e = ne; e-- > 0;) {
        if (!((*trav)[TRAV_STRIDE * e + 1].w < 0.0f)) {  // a group
            run = 0;
            continue;
        }
        if (ends_subtree[e + 1]) run = 0;
        const bool boxed_tri = (*trav)[TRAV_STRIDE * e + 1].w == TRAV_BOXED_LEAF_TAG && object_of(e) < scene->n_objects &&
                               scene->objects[object_of(e)].kind == RTC_TRIANGLE;
        mesh = boxed_tri && (run == 0 || (mesh && same_mesh(e, e + 1)));
        run = std::min(run + 1u, 1u << 20);
        longest = std::max(longest, run);
        float& w = (*trav)[TRAV_STRIDE * e + 2].w;
        w = (float)(8u * run + (mesh ? 4u : 0u) + ((uint32_t)w & 3u));
    }
    return longest;
}

// TextureMap / CubicMap: the pattern's second record points at its UV patterns, which are appended to `uvrec`.
static rtc_status pack_texture_map(const rtc_pattern& pt, float4 rec[5], std::vector<float4>* uvrec, std::vector<float>* texels,
                                   std::vector<std::pair<const float*, size_t>>* seen) {
    const uint32_t want = pt.kind == RTC_PATTERN_CUBE_MAP ? 6u : 1u;
    if (pt.n_uv != want || !pt.uv) return fail(RTC_ERR_INVALID_ARG, "pattern kind %d needs %u UV pattern(s), got %u", pt.kind, want, pt.n_uv);
    if (pt.kind == RTC_PATTERN_TEXTURE_MAP && (pt.uv_mapping < RTC_MAP_SPHERICAL || pt.uv_mapping > RTC_MAP_CYLINDRICAL))
        return fail(RTC_ERR_UNSUPPORTED, "UV mapping %d is not on the device path", pt.uv_mapping);
    uint32_t mapping = (uint32_t)pt.uv_mapping, first = (uint32_t)(uvrec->size() / 6);
    float mf, ff;
    std::memcpy(&mf, &mapping, 4);
    std::memcpy(&ff, &first, 4);
    rec[1] = make_float4(mf, ff, 0.0f, 0.0f);
    for (uint32_t k = 0; k < want; k++) {
        rtc_status st = pack_uv_pattern(pt.uv[k], uvrec, texels, seen);
        if (st != RTC_OK) return st;
    }
    return RTC_OK;
}
// The five pattern records of one material (see SceneSoA::pat).
static void pack_pattern(const rtc_pattern& pt, float4 rec[5]) {
    uint32_t kind = (uint32_t)pt.kind;
    float kind_f;
    std::memcpy(&kind_f, &kind, 4);
    rec[0] = make_float4(pt.a[0], pt.a[1], pt.a[2], kind_f);
    // Gradient::new / Sine2D::new keep distance = b - a (gradient.rs:17, sine_2d.rs:17)
    const bool dist = pt.kind == RTC_PATTERN_GRADIENT || pt.kind == RTC_PATTERN_SINE2D;
    rec[1] = dist ? make_float4(pt.b[0] - pt.a[0], pt.b[1] - pt.a[1], pt.b[2] - pt.a[2], 0.0f)
                  : make_float4(pt.b[0], pt.b[1], pt.b[2], 0.0f);
    for (int r = 0; r < 3; r++)
        rec[2 + r] = make_float4(pt.inv[4 * r], pt.inv[4 * r + 1], pt.inv[4 * r + 2], pt.inv[4 * r + 3]);
}

// Validates and flattens rtc_scene + rtc_camera into the kernel's header and
// SoA records (host staging buffer: 7 float4 arrays of np entries each, then 5 pattern records per object).
// Appends the six `uvrec` records of one UV pattern; UVImage canvases go to `texels` (each distinct host image once).
static rtc_status pack_uv_pattern(const rtc_uv_pattern& u, std::vector<float4>* uvrec, std::vector<float>* texels,
                                  std::vector<std::pair<const float*, size_t>>* seen) {
    auto as_f = [](uint32_t v) {
        float f;
        std::memcpy(&f, &v, 4);
        return f;
    };
    if (u.kind < RTC_UV_CHECKERS || u.kind > RTC_UV_IMAGE) return fail(RTC_ERR_UNSUPPORTED, "UV pattern kind %d is not on the device path", u.kind);
    size_t first_texel = 0;
    uint32_t iw = 0, ih = 0;
    if (u.kind == RTC_UV_IMAGE) {
        if (!u.image_rgb || u.image_width == 0 || u.image_height == 0) return fail(RTC_ERR_INVALID_ARG, "UVImage without a canvas");
        iw = u.image_width;
        ih = u.image_height;
        bool found = false;
        for (auto& e : *seen)
            if (e.first == u.image_rgb) {
                first_texel = e.second;
                found = true;
            }
        if (!found) {
            first_texel = texels->size() / 3;
            texels->insert(texels->end(), u.image_rgb, u.image_rgb + (size_t)iw * ih * 3);
            seen->push_back({u.image_rgb, first_texel});
        }
        if (first_texel + (size_t)iw * ih > 0xffffffffull) return fail(RTC_ERR_UNSUPPORTED, "more than 2^32 texels");
    }
    const float (*c)[3] = u.colors;
    uvrec->push_back(make_float4(as_f((uint32_t)u.kind), u.width, u.height, as_f((uint32_t)first_texel)));
    uvrec->push_back(make_float4(as_f(iw), as_f(ih), 0.0f, 0.0f));
    uvrec->push_back(make_float4(c[0][0], c[0][1], c[0][2], c[1][0]));
    uvrec->push_back(make_float4(c[1][1], c[1][2], c[2][0], c[2][1]));
    uvrec->push_back(make_float4(c[2][2], c[3][0], c[3][1], c[3][2]));
    uvrec->push_back(make_float4(c[4][0], c[4][1], c[4][2], 0.0f));
    return RTC_OK;
}

// `heavy_boxes` (optional): world-space boxes, 7 floats each (min, max, rank), of the top-level GroupShapes that hold long runs of leaves
// (divided meshes) -- where a frame's slow waves are (rtc_ctx_render: block list).
// What a primary ray can see at all, for the scene rectangle (rtc_ctx_set_scene): known when every top-level entry is
// bounded -- their padded union is `box` -- or a plane, seen only by rays that point towards it.
// RTC_AMD_PRUNE=0: every group / node entry gets an infinite slack, with which the walks' distance test never closes one
static void no_distance_pruning(std::vector<float4>* trav) {
    for (size_t e = 0, ne = trav->size() / TRAV_STRIDE; e < ne; e++)
        if (!((*trav)[TRAV_STRIDE * e + 1].w < 0.0f)) (*trav)[TRAV_STRIDE * e + 1].w = INFINITY;
}
struct SceneRegion {
    bool known = false, has_box = false;
    double box[6] = {0, 0, 0, 0, 0, 0};
    std::vector<float> entry_boxes;  // the padded boxes of the bounded top-level entries one by one (HEAVY_BOX_FLOATS each; `box` is their union)
    std::vector<std::array<double, 4>> planes;  // the plane's object-space y of a world point p: r[0] p.x + r[1] p.y + r[2] p.z + r[3]
};
static rtc_status flatten(const Policy& P, const rtc_scene* scene, const rtc_camera* cam, SceneHdr* hdr, std::vector<float4>* soa,
                          std::vector<float>* texels, std::vector<float>* heavy_boxes = nullptr, SceneRegion* region = nullptr,
                          bool allow_sequence = false) {
    std::vector<float4> uvrec;
    std::vector<std::pair<const float*, size_t>> seen_images;
    if (!scene) return fail(RTC_ERR_INVALID_ARG, "scene is NULL");
    if (!scene->light) return fail(RTC_ERR_NO_LIGHT, "World light should be set");  // world.rs:66
    if (scene->n_objects && !scene->objects) return fail(RTC_ERR_INVALID_ARG, "scene.objects is NULL");
    std::memset(hdr, 0, sizeof(*hdr));
    const uint32_t n = scene->n_objects;
    hdr->n_objects = n;
    const uint32_t np = padded_count(n);  // stride of each SoA array
    soa->assign((size_t)20 * np, make_float4(0, 0, 0, 0));
    struct Extent {  // world_extent() of every object, computed once (a mesh leaf is asked for it by every group around it)
        double lo[3], hi[3];
        bool ok;
    };
    std::vector<Extent> extent(n);
    {
        uint32_t none = SHAPE_NONE;
        float none_f;
        std::memcpy(&none_f, &none, 4);
        for (uint32_t i = n; i < np; i++) (*soa)[0 * np + i] = make_float4(0.0f, 0.0f, 0.0f, none_f);
    }
    for (uint32_t i = 0; i < n; i++) {
        const rtc_object& o = scene->objects[i];
        if (o.kind < RTC_SPHERE || o.kind > RTC_TRIANGLE)
            return fail(RTC_ERR_UNSUPPORTED, "object %u: shape kind %d is not on the device path", i, o.kind);
        if (o.kind == RTC_TRIANGLE) pack_triangle(o, &(*soa)[12 * (size_t)np + 3 * (size_t)i]);
        if (!is_affine(o.inv))
            return fail(RTC_ERR_UNSUPPORTED,
                        "object %u: inverse transform's last row is not exactly [0,0,0,1] (projective transforms "
                        "are not supported)", i);
        float4 g[4];
        pack_geometry(o, g);
        for (int k = 0; k < 4; k++) (*soa)[(size_t)k * np + i] = g[k];
        (*soa)[18 * (size_t)np + i] = make_float4(o.inv[3], o.inv[7], o.inv[11], 0.0f);
        {
            double lo[3], hi[3];  // bounding sphere: around the world-space box of the shape's own bounds
            float4 bs = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
            // (a cone's near-parallel branch, cone.rs:99-107, reports roots off the bounded cone: no sphere holds its hits)
            Extent& ex = extent[i];
            ex.ok = world_extent(o, ex.lo, ex.hi);
            for (int a = 0; a < 3; a++) lo[a] = ex.lo[a], hi[a] = ex.hi[a];
            if (o.kind != RTC_CONE && ex.ok) {
                double r2 = 0.0;
                for (int a = 0; a < 3; a++) r2 += 0.25 * (hi[a] - lo[a]) * (hi[a] - lo[a]);
                bs = make_float4((float)(0.5 * (lo[0] + hi[0])), (float)(0.5 * (lo[1] + hi[1])), (float)(0.5 * (lo[2] + hi[2])),
                                 (float)(1.001 * std::sqrt(r2)));
                if (!std::isfinite(bs.x) || !std::isfinite(bs.y) || !std::isfinite(bs.z) || !std::isfinite(bs.w))
                    bs = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
            }
            (*soa)[19 * (size_t)np + i] = bs;
        }
        const rtc_material& m = o.material;
        (*soa)[4 * np + i] = make_float4(m.color[0], m.color[1], m.color[2], m.ambient);
        (*soa)[5 * np + i] = make_float4(m.diffuse, m.specular, m.shininess, m.reflective);
        (*soa)[6 * np + i] = make_float4(m.transparency, m.refractive_index, 0.0f, 0.0f);
        const rtc_pattern& pt = m.pattern;
        if (pt.kind != RTC_PATTERN_NONE) {
            if (pt.kind < RTC_PATTERN_STRIPES || pt.kind > RTC_PATTERN_CUBE_MAP)
                return fail(RTC_ERR_UNSUPPORTED, "object %u: pattern kind %d is not on the device path", i, pt.kind);
            if (!is_affine(pt.inv))
                return fail(RTC_ERR_UNSUPPORTED, "object %u: pattern inverse transform is not affine", i);
            hdr->has_patterns = 1;
            float4* rec = &(*soa)[7 * (size_t)np + 5 * (size_t)i];
            pack_pattern(pt, rec);
            if (pt.kind >= RTC_PATTERN_TEXTURE_MAP) {
                rtc_status ust = pack_texture_map(pt, rec, &uvrec, texels, &seen_images);
                if (ust != RTC_OK) return ust;
            }
        }
    }
    // Triangles met by tree walks get a pre-culling box (tri_precull): the ball around everything bounded (and the camera)
    // bounds the distance between a ray's origin and a triangle; rays that start outside it do not pre-cull
    std::vector<float4> tbox;  // 3 records per object: { box.min, usable }, { box.max, 0 }, { unit normal, 0 }, world space
    if (scene->n_groups && P.tri_precull) {
        bool any_triangle = false;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < n; i++) {
            any_triangle = any_triangle || scene->objects[i].kind == RTC_TRIANGLE;
            double olo[3], ohi[3];
            for (int a = 0; a < 3; a++) olo[a] = extent[i].lo[a], ohi[a] = extent[i].hi[a];
            if (extent[i].ok)
                for (int a = 0; a < 3; a++) lo[a] = std::fmin(lo[a], olo[a]), hi[a] = std::fmax(hi[a], ohi[a]);
        }
        if (cam) {
            float org[4];
            const float zero[4] = {0.0f, 0.0f, 0.0f, 1.0f};
            mat_vec4(cam->inv, zero, org);
            for (int a = 0; a < 3; a++) lo[a] = std::fmin(lo[a], (double)org[a]), hi[a] = std::fmax(hi[a], (double)org[a]);
        }
        double r2 = 0.0;
        for (int a = 0; a < 3; a++) r2 += 0.25 * (hi[a] - lo[a]) * (hi[a] - lo[a]);
        if (any_triangle && std::isfinite(r2) && r2 > 0.0) {
            const double radius = 1.05 * std::sqrt(r2);  // a little room: hit points are computed, not exact
            // RTC_AMD_TRI_NAIVE=1 (tests only): no angle guard, no padding -- what tests/test_tri_precull.py must catch
            const bool naive = P.tri_naive;
            hdr->tri_guard = naive ? 0.0f : TRI_GUARD;
            hdr->has_tbox = 1;
            for (int a = 0; a < 3; a++) hdr->cull_c[a] = (float)(0.5 * (lo[a] + hi[a]));
            hdr->cull_r2 = (float)(radius * radius);
            tbox.assign(3 * (size_t)n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            TransformFacts tf;
            // ERROR_BUDGET.md B7: the padding is derived from D, the distance between a ray's origin and a triangle -- and the ray's
            // own transformation into object space is good to 3 u (|M| |origin| + |t|) (E1), which is relative to the
            // COORDINATES: a mesh a thousand of its own sizes from the world's origin moves by that much more.  D stands for both.
            double far_coord = 0.0;
            for (int a = 0; a < 3; a++) far_coord = std::fmax(far_coord, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
            const double d_bound = std::fmax(2.0 * radius, far_coord + radius);
            for (uint32_t i = 0; i < n; i++)
                if (scene->objects[i].kind == RTC_TRIANGLE)
                    triangle_box(scene->objects[i], &(*soa)[12 * (size_t)np + 3 * (size_t)i], d_bound, TRI_GUARD, naive ? 0.0 : 1.0,
                                 &tf, &tbox[3 * (size_t)i]);
        }
    }
    // A flat world of many bounded objects gets a bounding-volume hierarchy of the library's own (see build_flat_bvh)
    if (!scene->n_groups && cam && n >= 16 && P.bvh) {
        std::vector<float4> trav;
        float cam_origin[4];
        const float zero[4] = {0.0f, 0.0f, 0.0f, 1.0f};
        mat_vec4(cam->inv, zero, cam_origin);
        if (build_flat_bvh(scene, cam_origin, &trav)) {
            hdr->max_leaf_run = mark_leaf_runs(&trav, scene);
            hdr->internal_boxes = 1;
            hdr->n_trav = (uint32_t)(trav.size() / TRAV_STRIDE);
            if (!P.prune) no_distance_pruning(&trav);
            soa->insert(soa->end(), trav.begin(), trav.end());
        }
    }
    // GroupShapes: write the depth-first traversal out as an entry list (see SceneSoA::trav)
    if (scene->n_groups) {
        if (!scene->groups) return fail(RTC_ERR_INVALID_ARG, "scene.groups is NULL");
        // SHAPE_LOOSE: leaves that some group around them does not (safely) contain -- see the flag.  Such a group must
        // not be pruned by distance either (for_each_object assumes a group's hits lie inside its box): loose_group.
        std::vector<char> loose_group(scene->n_groups, 0);
        for (uint32_t g = 0; g < scene->n_groups; g++) {
            const rtc_group& grp = scene->groups[g];
            if ((uint64_t)grp.first_object + grp.n_objects > n) continue;  // reported below
            for (uint32_t i = grp.first_object; i < grp.first_object + grp.n_objects; i++) {
                double lo[3], hi[3];
                bool inside = extent[i].ok;
                for (int a = 0; a < 3; a++) lo[a] = extent[i].lo[a], hi[a] = extent[i].hi[a];
                for (int a = 0; a < 3 && inside; a++) {
                    const double tol = 1e-5 * (std::fabs(lo[a]) + std::fabs(hi[a]) + 1.0);
                    inside = lo[a] >= (double)grp.bounds_min[a] - tol && hi[a] <= (double)grp.bounds_max[a] + tol;
                }
                // (a cone's near-parallel branch, cone.rs:99-107, reports roots of the UNBOUNDED double cone: hits outside
                // the cone's own bounds, hence outside any group box built from them)
                if (scene->objects[i].kind == RTC_CONE) loose_group[g] = 1;
                if (!inside) {
                    loose_group[g] = 1;
                    uint32_t bits;
                    std::memcpy(&bits, &(*soa)[i].w, 4);
                    bits |= SHAPE_LOOSE;
                    std::memcpy(&(*soa)[i].w, &bits, 4);
                }
            }
        }
        struct Open {
            uint32_t end;
            size_t entry;
        };
        std::vector<float4> trav;
        std::vector<Open> open;
        std::vector<std::pair<size_t, uint32_t>> top_level;  // (entry, group) of the groups directly under the world
        uint32_t gi = 0;
        bool any = false;
        auto as_f = [](uint32_t u) {
            float f;
            std::memcpy(&f, &u, 4);
            return f;
        };
        for (uint32_t p = 0; p <= n; p++) {
            while (!open.empty() && open.back().end == p) {  // close: skip index = next entry
                trav[TRAV_STRIDE * open.back().entry].w = as_f((uint32_t)(trav.size() / TRAV_STRIDE));
                open.pop_back();
            }
            for (;;) {
                while (gi < scene->n_groups && scene->groups[gi].n_objects == 0) gi++;  // empty groups never hit
                if (gi >= scene->n_groups || scene->groups[gi].first_object != p) break;
                const rtc_group& g = scene->groups[gi];
                const uint64_t end = (uint64_t)g.first_object + g.n_objects;
                if (end > n || (!open.empty() && end > open.back().end))
                    return fail(RTC_ERR_INVALID_ARG, "group %u: objects [%u, %u) do not nest inside the enclosing group / the world",
                                gi, g.first_object, (unsigned)end);
                if (open.empty()) top_level.push_back({trav.size() / TRAV_STRIDE, gi});
                open.push_back({(uint32_t)end, trav.size() / TRAV_STRIDE});
                float big = 0.0f;  // pruning slack: 4e-3 of the largest |coordinate| (NaN-propagating on purpose; ERROR_BUDGET.md B6)
                for (int a = 0; a < 3; a++) {
                    const float lo = std::fabs(g.bounds_min[a]), hi = std::fabs(g.bounds_max[a]);
                    big = (lo != lo || hi != hi) ? NAN : std::fmax(big, std::fmax(lo, hi));
                }
                // q of the pruning margin (kernel: for_each_object): 1e-6 over the smallest size of a sphere, cylinder or cone below
                // this group -- the unit shape under its transform is at least 1 / |inverse 3x3| (Frobenius) across
                float q = 0.0f;
                for (uint32_t i = g.first_object; i < g.first_object + g.n_objects; i++) {
                    const rtc_object& o = scene->objects[i];
                    if (o.kind != RTC_SPHERE && o.kind != RTC_CYLINDER && o.kind != RTC_CONE) continue;
                    double f2 = 0.0;
                    for (int r = 0; r < 3; r++)
                        for (int cidx = 0; cidx < 3; cidx++) f2 += (double)o.inv[4 * r + cidx] * o.inv[4 * r + cidx];
                    const float qi = (float)(1e-6 * std::sqrt(f2));
                    q = (qi != qi) ? q : std::fmax(q, qi);
                }
                trav.push_back(make_float4(g.bounds_min[0], g.bounds_min[1], g.bounds_min[2], 0.0f));
                trav.push_back(make_float4(g.bounds_max[0], g.bounds_max[1], g.bounds_max[2], loose_group[gi] ? INFINITY : 4e-3f * big));
                trav.push_back(make_float4(q, 0.0f, 0.0f, 0.0f));
                any = true;
                gi++;
            }
            if (p < n) {
                // a leaf; a triangle with a usable pre-culling box carries it along (tri_precull)
                const float4* tb = tbox.empty() ? nullptr : &tbox[3 * (size_t)p];
                const bool boxed = tb && tb[0].w > 0.0f;
                trav.push_back(boxed ? make_float4(tb[0].x, tb[0].y, tb[0].z, as_f(p)) : make_float4(0.0f, 0.0f, 0.0f, as_f(p)));
                trav.push_back(boxed ? make_float4(tb[1].x, tb[1].y, tb[1].z, TRAV_BOXED_LEAF_TAG) : make_float4(0.0f, 0.0f, 0.0f, TRAV_LEAF_TAG));
                trav.push_back(boxed ? tb[2] : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            }
        }
        if (gi != scene->n_groups)
            return fail(RTC_ERR_INVALID_ARG, "group %u: groups must be listed in pre-order with first_object inside [0, n_objects)", gi);
        // boxed triangle leaves that follow one another are pre-culled two at a time (for_each_object): mark the first of a pair
        auto mark_pairs = [&trav]() {
            for (size_t e = 0, ne = trav.size() / TRAV_STRIDE; e < ne; e++) {
                if (trav[TRAV_STRIDE * e + 1].w != TRAV_BOXED_LEAF_TAG) continue;
                int more = 0;  // boxed leaves right after this one, up to 3
                while (more < 3 && e + more + 1 < ne && trav[TRAV_STRIDE * (e + more + 1) + 1].w == TRAV_BOXED_LEAF_TAG) more++;
                trav[TRAV_STRIDE * e + 2].w = (float)more;
            }
        };
        mark_pairs();
        hdr->max_leaf_run = mark_leaf_runs(&trav, scene);
        if (heavy_boxes)
            for (const auto& tl : top_level) {
                uint32_t skip, longest = 0;
                std::memcpy(&skip, &trav[TRAV_STRIDE * tl.first].w, 4);
                for (size_t e = tl.first + 1; e < skip && e < trav.size() / TRAV_STRIDE; e++)
                    if (trav[TRAV_STRIDE * e + 1].w < 0.0f) longest = std::max(longest, (uint32_t)trav[TRAV_STRIDE * e + 2].w >> 3);
                if (longest >= 16u) {
                    const rtc_group& g = scene->groups[tl.second];
                    for (int a = 0; a < 3; a++) heavy_boxes->push_back(g.bounds_min[a]);
                    for (int a = 0; a < 3; a++) heavy_boxes->push_back(g.bounds_max[a]);
                    // how dear a pixel on this group is, as a rank: a surface that both reflects and refracts doubles its rays
                    // at every level of the recursion, one that does either keeps them going
                    bool refl = false, refr = false;
                    for (uint32_t i = g.first_object; i < g.first_object + g.n_objects && i < n; i++) {
                        refl |= scene->objects[i].material.reflective > 0.0f;
                        refr |= scene->objects[i].material.transparency > 0.0f;
                    }
                    heavy_boxes->push_back(refl && refr ? 3.0f : (refl || refr) ? 2.0f : 1.0f);
                }
            }
        // Long runs of boxed triangles get a hierarchy of the library's own (RTC_AMD_CLUSTERS=0 / 1: never / always) -- in
        // frames large enough to be traced by one or two lanes per pixel.  Where eight lanes split every run (small frames,
        // whose time is that of their slowest wave) the nodes cut the runs into pieces of a lane's share and every piece
        // ends in a round of shuffles: here_be_dragons 1000 x 400 0.79 -> 0.98 ms, 2000 x 800 1.41 -> 1.49; 4000 x 1600 3.25 -> 2.92.
        const bool clusters_pay = cam && choose_share_log2_runs(((uint64_t)cam->width * cam->height + 63) / 64) <= 1u;
        if (hdr->has_tbox && (hdr->max_leaf_run >= 24u || P.cluster_min_run != 0u) && (P.clusters < 0 ? clusters_pay : P.clusters != 0)) {
            cluster_leaf_runs(&trav, (double)hdr->tri_guard, P);
            hdr->has_tbox = 2;  // ... and the walks look for nodes among the group entries (spec_has_nodes)
            mark_pairs();
            (void)mark_leaf_runs(&trav, scene);  // (max_leaf_run keeps the length of the reference's runs: what the launch policy goes by)
        }
        // A small tree (<= 8 leaves under <= 8 groups) keeps the unrolled flat kernels: every group becomes a GATE -- its box,
        // tested once per ray with the reference's own aabb test -- and a leaf is intersected only if the ray opens all the
        // groups around it, which is all the recursive walk does (group.rs:115-133).  Same leaves in the same order.
        uint32_t n_gates = 0;
        for (uint32_t g = 0; g < scene->n_groups; g++) n_gates += scene->groups[g].n_objects != 0;
        // (render path only: the batched entry points run the any-count loop for flat worlds, which has no gates)
        if (cam && any && n <= 8 && n_gates <= RTC_MAX_GATES && P.gates) {
            uint32_t k = 0;
            for (uint32_t g = 0; g < scene->n_groups; g++) {
                const rtc_group& grp = scene->groups[g];
                if (grp.n_objects == 0) continue;
                for (int a = 0; a < 3; a++) hdr->gate_box[k][a] = grp.bounds_min[a], hdr->gate_box[k][3 + a] = grp.bounds_max[a];
                for (uint32_t i = grp.first_object; i < grp.first_object + grp.n_objects; i++) hdr->gate_mask[i] |= 1u << k;
                k++;
            }
            hdr->n_gates = n_gates;
        } else if (any) {
            hdr->n_trav = (uint32_t)(trav.size() / TRAV_STRIDE);
            if (!P.prune) no_distance_pruning(&trav);
            soa->insert(soa->end(), trav.begin(), trav.end());
        }
    }
    // scene_box (render_body's early-out for primary rays): the union of the top-level entries when all of them are
    // bounded -- groups by their boxes (a ray that misses a group's box is turned away whatever is inside), leaves of the
    // kinds whose hits lie within their bounds (not cones: stray roots; not triangles: ill-conditioned near their plane).
    // Padding: a group's box only needs what the approximate test's rounding needs (the reference's own test of that box
    // is exact about it): 1e-4 of the coordinates.  A leaf is padded by 10 % of its own half extent: ERROR_BUDGET.md B8 --
    // a leaf only counts as bounded when the camera is within ~100 of its own units, where the reference's quadratic reports
    // nothing beyond 1 % of the radius (E2) -- the guards below.  (Up to round 3 the padding also carried 0.6 % of the camera's
    // distance to the scene's far corner, an E2 allowance in world units that the object-space guard makes redundant, and
    // that made the boxes of C5's spheres half as large again as the spheres.)
    if (cam && n > 0) {
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};  // union of the padded entries
        double raw_lo[3] = {INFINITY, INFINITY, INFINITY}, raw_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        struct Entry {
            double lo[3], hi[3];
            bool group;
        };
        std::vector<Entry> entries;  // the bounded ones
        std::vector<std::array<double, 4>> planes;
        bool known = true;  // every entry is bounded or a plane
        uint32_t i = 0, g = 0;
        while (i < n && known) {
            while (g < scene->n_groups && (scene->groups[g].n_objects == 0 || scene->groups[g].first_object < i)) g++;  // nested / empty
            if (g < scene->n_groups && scene->groups[g].first_object == i) {
                const rtc_group& grp = scene->groups[g];
                Entry e;
                e.group = true;
                for (int a = 0; a < 3; a++) {
                    known = known && std::isfinite(grp.bounds_min[a]) && std::isfinite(grp.bounds_max[a]);
                    e.lo[a] = grp.bounds_min[a], e.hi[a] = grp.bounds_max[a];
                }
                entries.push_back(e);
                i = grp.first_object + grp.n_objects;
            } else {
                const rtc_object& o = scene->objects[i];
                if (o.kind == RTC_PLANE) {
                    planes.push_back({(double)o.inv[4], (double)o.inv[5], (double)o.inv[6], (double)o.inv[7]});
                    known = known && std::isfinite(o.inv[4]) && std::isfinite(o.inv[5]) && std::isfinite(o.inv[6]) && std::isfinite(o.inv[7]);
                } else {
                    // ERROR_BUDGET.md B8.  The box asserts "a primary ray that misses it hits nothing", and the reference's f32
                    // quadratic reports hits for lines that pass a sphere / cylinder wall at up to sqrt(1 + 16 u oo) of its radius
                    // (E2; oo: squared distance of the ray's origin -- here always the camera -- in the OBJECT's space: a disc
                    // scaled 1e-3 across, seen from ten world units, is ten thousand of its own units away and "grows" phantom
                    // hits seven radii out: wide seeds 177, 217, 249).  So a leaf only counts as bounded when the camera is within
                    // ~100 of its own units (32 u oo <= 0.02: the phantom rim is 1 % of the radius, the padding below 10 %), and
                    // when the camera's object-space position itself is good to 1e-3 (E1).  A cylinder also reports a wall hit's
                    // height wrongly by up to 2e-3 of the height difference to the camera (E2, relative error of t) -- its box
                    // grows by four times that -- and from within two radii of its axis by more than any padding covers.
                    Entry e;
                    e.group = false;
                    float orgf[4];
                    const float zero4[4] = {0.0f, 0.0f, 0.0f, 1.0f};
                    mat_vec4(cam->inv, zero4, orgf);
                    double oc[3], e1 = 0.0;
                    for (int r = 0; r < 3; r++) {
                        oc[r] = (double)o.inv[4 * r] * orgf[0] + (double)o.inv[4 * r + 1] * orgf[1] + (double)o.inv[4 * r + 2] * orgf[2] + (double)o.inv[4 * r + 3];
                        e1 = std::fmax(e1, std::fabs((double)o.inv[4 * r] * orgf[0]) + std::fabs((double)o.inv[4 * r + 1] * orgf[1]) +
                                               std::fabs((double)o.inv[4 * r + 2] * orgf[2]) + std::fabs((double)o.inv[4 * r + 3]));
                    }
                    const double U = 5.9604644775390625e-8;  // 2^-24
                    const double oo = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2], oo_xz = oc[0] * oc[0] + oc[2] * oc[2];
                    bool well = 3.0 * U * e1 <= 1e-3;
                    double grow_y = 0.0;
                    if (o.kind == RTC_SPHERE) well = well && 32.0 * U * oo <= 0.02;
                    else if (o.kind == RTC_CYLINDER) {
                        well = well && 32.0 * U * oo_xz <= 0.02 && oo_xz >= 4.0;
                        grow_y = 8e-3 * (std::fabs(oc[1]) + std::fmax(std::fabs((double)o.min_y), std::fabs((double)o.max_y)));
                    }
                    known = known && well && (o.kind == RTC_SPHERE || o.kind == RTC_CUBE || o.kind == RTC_CYLINDER) && world_extent(o, e.lo, e.hi, grow_y);
                    entries.push_back(e);
                }
                i++;
            }
        }
        const bool bounded = known && planes.empty();
        if (known && !entries.empty()) {
            float org[4];
            const float zero[4] = {0.0f, 0.0f, 0.0f, 1.0f};
            mat_vec4(cam->inv, zero, org);
            for (const Entry& e : entries)
                for (int a = 0; a < 3; a++) raw_lo[a] = std::fmin(raw_lo[a], e.lo[a]), raw_hi[a] = std::fmax(raw_hi[a], e.hi[a]);
            double far2 = 0.0;
            for (int a = 0; a < 3; a++) {
                const double d = std::fmax(std::fabs(raw_lo[a] - org[a]), std::fabs(raw_hi[a] - org[a]));
                far2 += d * d;
            }
            const double far = std::sqrt(far2);
            std::vector<float> padded;
            for (const Entry& e : entries) {
                float pb[6];
                for (int a = 0; a < 3; a++) {
                    const double pad = (e.group ? 0.0 : 0.05 * (e.hi[a] - e.lo[a])) + 1e-4 * (std::fabs(e.lo[a]) + std::fabs(e.hi[a]) + far);
                    lo[a] = std::fmin(lo[a], e.lo[a] - pad), hi[a] = std::fmax(hi[a], e.hi[a] + pad);
                    pb[a] = (float)std::nextafter((float)(e.lo[a] - pad), -INFINITY), pb[3 + a] = (float)std::nextafter((float)(e.hi[a] + pad), INFINITY);
                }
                padded.insert(padded.end(), pb, pb + 6);
                padded.push_back(1.0f);
            }
            bool ok = std::isfinite(far);
            float box[6];
            for (int a = 0; a < 3 && ok; a++) {
                box[a] = (float)lo[a];
                box[3 + a] = (float)hi[a];
                ok = std::isfinite(box[a]) && std::isfinite(box[3 + a]);
            }
            if (ok && bounded && P.scene_box) {
                for (int a = 0; a < 6; a++) hdr->scene_box[a] = box[a];
                hdr->has_scene_box = 1u;
            }
            if (region && ok) {
                region->has_box = true;
                for (int a = 0; a < 6; a++) region->box[a] = box[a];
                region->entry_boxes = padded;
            }
            known = known && ok;
        }
        if (region) {
            region->known = known;
            region->planes = planes;
        }
    }
    const rtc_light& l = *scene->light;
    hdr->light_kind = l.kind;
    for (int k = 0; k < 3; k++) {
        hdr->li[k] = l.intensity[k];
        hdr->lpos[k] = l.position[k];
        hdr->corner[k] = l.corner[k];
        hdr->uvec[k] = l.u_vec[k];
        hdr->vvec[k] = l.v_vec[k];
    }
    rtc_status st;
    if ((st = check_tuple(l.position, 1.0f, "light.position")) != RTC_OK) return st;
    hdr->u_steps = hdr->v_steps = 1;
    hdr->cells_f = 1.0f;
    hdr->jitter_mode = RTC_JITTER_CONSTANT;
    if (l.kind == RTC_LIGHT_RECT) {
        if ((st = check_tuple(l.corner, 1.0f, "light.corner")) != RTC_OK) return st;
        if ((st = check_tuple(l.u_vec, 0.0f, "light.u_vec")) != RTC_OK) return st;
        if ((st = check_tuple(l.v_vec, 0.0f, "light.v_vec")) != RTC_OK) return st;
        if (l.u_steps <= 0 || l.v_steps <= 0) return fail(RTC_ERR_INVALID_ARG, "light steps must be positive");
        if (l.jitter_mode != RTC_JITTER_CONSTANT && l.jitter_mode != RTC_JITTER_HASHED && l.jitter_mode != RTC_JITTER_SEQUENCE)
            return fail(RTC_ERR_UNSUPPORTED, "jitter mode %d cannot run on the device (closures are host-only)", l.jitter_mode);
        if (l.jitter_mode == RTC_JITTER_SEQUENCE) {
            // test/utils.rs:19-24: the cycle is state carried across every question a light is asked -- across pixels, in the
            // reference's serial loop.  Only one call on a fresh light is defined without that order: the batched
            // rtc_intensity_at / rtc_point_on_light (cam == nullptr and allow_sequence).
            if (cam || !allow_sequence)
                return fail(RTC_ERR_UNSUPPORTED, "sequence jitter (hardcoded_jitter) is serial across pixels and rays: only rtc_intensity_at and "
                                                 "rtc_point_on_light accept it");
            if (l.jitter_seq_len < 1 || l.jitter_seq_len > RTC_JITTER_SEQUENCE_MAX)
                return fail(RTC_ERR_INVALID_ARG, "sequence jitter: %u values (1 .. %d)", l.jitter_seq_len, RTC_JITTER_SEQUENCE_MAX);
        }
        hdr->u_steps = l.u_steps;
        hdr->v_steps = l.v_steps;
        hdr->cells_f = (float)(l.u_steps * l.v_steps);
        hdr->jitter_mode = l.jitter_mode;
        hdr->jitter_const = l.jitter_const;
        hdr->jitter_seed = l.jitter_seed;
        if (l.jitter_mode == RTC_JITTER_SEQUENCE) {
            hdr->jitter_seq_len = l.jitter_seq_len;
            bool unit = true;  // light-cone culling needs every sample inside the parallelogram: all values in [0, 1]
            for (uint32_t k = 0; k < RTC_JITTER_SEQUENCE_MAX; k++) {
                hdr->jitter_seq[k] = l.jitter_seq[k % l.jitter_seq_len];
                unit = unit && l.jitter_seq[k % l.jitter_seq_len] >= 0.0f && l.jitter_seq[k % l.jitter_seq_len] <= 1.0f;
            }
            hdr->jitter_const = unit ? 0.5f : 2.0f;  // (what light_cull_mask looks at for a source that is not the hash)
        }
        // light-cone culling inputs: the parallelogram's corners in every object's space, and its y range
        const float su = (float)l.u_steps, sv = (float)l.v_steps;
        float cw[4][3];
        float y_lo = INFINITY, y_hi = -INFINITY, y_abs = 0.0f;
        for (int k = 0; k < 4; k++) {
            const float fu = (k == 1 || k == 2) ? su : 0.0f, fv = (k >= 2) ? sv : 0.0f;
            for (int a = 0; a < 3; a++) cw[k][a] = l.corner[a] + l.u_vec[a] * fu + l.v_vec[a] * fv;
            y_lo = fminf(y_lo, cw[k][1]);
            y_hi = fmaxf(y_hi, cw[k][1]);
            y_abs += fabsf(cw[k][1]);
        }
        {   // classify_cells (ERROR_BUDGET.md B10): half a cell's diagonal -- the longer one -- padded
            double d1 = 0.0, d2 = 0.0, lmax = 0.0;
            for (int a = 0; a < 3; a++) {
                d1 += ((double)l.u_vec[a] + l.v_vec[a]) * ((double)l.u_vec[a] + l.v_vec[a]);
                d2 += ((double)l.u_vec[a] - l.v_vec[a]) * ((double)l.u_vec[a] - l.v_vec[a]);
                for (int k = 0; k < 4; k++) lmax = std::fmax(lmax, std::fabs((double)cw[k][a]));
            }
            const double hd = 0.5 * std::sqrt(std::fmax(d1, d2)) * 1.01 + 8.0 * 5.9604644775390625e-8 * lmax;
            const bool unit_jitter = l.jitter_mode == RTC_JITTER_HASHED || (hdr->jitter_const >= 0.0f && hdr->jitter_const <= 1.0f);
            hdr->cell_hd = (unit_jitter && std::isfinite(hd) && hd > 0.0) ? (float)hd : 0.0f;
        }
        const float ym = 1e-5f * y_abs + 1e-30f;  // far above the sample points' rounding error
        hdr->light_y_lo = y_lo - ym;
        hdr->light_y_hi = y_hi + ym;
        for (uint32_t i = 0; i < n; i++) {
            const float* m = scene->objects[i].inv;
            float c[4][3];
            for (int k = 0; k < 4; k++)
                for (int r = 0; r < 3; r++)
                    c[k][r] = m[4 * r] * cw[k][0] + m[4 * r + 1] * cw[k][1] + m[4 * r + 2] * cw[k][2] + m[4 * r + 3];
            float4* rec = &(*soa)[15 * (size_t)np + 3 * (size_t)i];
            rec[0] = make_float4(c[0][0], c[0][1], c[0][2], c[1][0]);
            rec[1] = make_float4(c[1][1], c[1][2], c[2][0], c[2][1]);
            rec[2] = make_float4(c[2][2], c[3][0], c[3][1], c[3][2]);
            // ERROR_BUDGET.md E1 for light_cull_mask: E bounds, in the object's own units, how far the pyramid the cull reasons
            // about (apex: the computed object-space shade point o; base: these computed corners) can sit from the rays the
            // exact test traces (same apex, direction M (sample - p)): |delta corner| + |delta o| <= 3 u (|M| (|L| + |p|) + 2 |t|),
            // written with 4 u.  |M| |p| is bounded through o itself, which the cull only trusts within 100 radii:
            // componentwise |g_k p_k| <= |o_k| + |t_k| for a scale+translate object, |M| |p| <= |M|_inf |F|_inf (|o|_inf + |t|_inf)
            // otherwise (F: the forward transform).  A plane's rule looks at o.y alone and has no distance limit: the constant
            // holds what does not depend on p, and for a plane that is not scale+translate only the kernel adds errB * |p|_inf.
            const rtc_object& ob = scene->objects[i];
            const double U4 = 4.0 * 5.9604644775390625e-8;
            double Lmax = 0.0, tinf = 0.0, minf = 0.0;
            for (int k = 0; k < 4; k++)
                for (int a = 0; a < 3; a++) Lmax = std::fmax(Lmax, std::fabs((double)cw[k][a]));
            double rows[3];
            for (int r = 0; r < 3; r++) {
                rows[r] = std::fabs((double)m[4 * r]) + std::fabs((double)m[4 * r + 1]) + std::fabs((double)m[4 * r + 2]);
                minf = std::fmax(minf, rows[r]);
                tinf = std::fmax(tinf, std::fabs((double)m[4 * r + 3]));
            }
            const bool diag = m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f && m[6] == 0.0f && m[8] == 0.0f && m[9] == 0.0f;
            double E, errB = 0.0;
            if (ob.kind == RTC_PLANE) {
                E = U4 * (2.0 * std::fabs((double)m[7]) + rows[1] * Lmax);
                if (!diag) errB = U4 * rows[1];
            } else {
                double hy = 1.0;
                if (ob.kind == RTC_CYLINDER) hy = std::fmax(std::fabs((double)ob.min_y), std::fabs((double)ob.max_y));
                const double rad = 103.0 * std::sqrt(ob.kind == RTC_CUBE ? 3.0 : 1.0 + (ob.kind == RTC_CYLINDER ? hy * hy : 0.0));  // |o| where the cull still decides
                if (diag) {
                    E = U4 * (rad + 2.0 * tinf + minf * Lmax);
                } else {
                    double F[9], f[3], finf = INFINITY;
                    if (forward_affine(m, F, f)) {
                        finf = 0.0;
                        for (int r = 0; r < 3; r++) finf = std::fmax(finf, std::fabs(F[3 * r]) + std::fabs(F[3 * r + 1]) + std::fabs(F[3 * r + 2]));
                    }
                    E = U4 * (minf * finf * (rad + tinf) + tinf + minf * Lmax);
                }
            }
            (*soa)[18 * (size_t)np + i].w = std::isfinite(E) ? (float)E : INFINITY;  // trn.w
            (*soa)[3 * (size_t)np + i].w = (float)errB;                             // off2.w
            if (ob.kind == RTC_SPHERE) {
                // light_cull_mask's cone pre-test: half the parallelogram's longer diagonal in this sphere's space, 0.1 % up
                double d02 = 0.0, d13 = 0.0;
                for (int r = 0; r < 3; r++) d02 += ((double)c[0][r] - c[2][r]) * ((double)c[0][r] - c[2][r]), d13 += ((double)c[1][r] - c[3][r]) * ((double)c[1][r] - c[3][r]);
                const double hdl = 0.5 * std::sqrt(std::fmax(d02, d13)) * 1.001;
                (*soa)[3 * (size_t)np + i].w = std::isfinite(hdl) && hdl > 0.0 ? (float)hdl : 0.0f;
            }
        }
    } else if (l.kind != RTC_LIGHT_POINT) {
        return fail(RTC_ERR_UNSUPPORTED, "light kind %d", l.kind);
    }
    hdr->all_cast = 1;
    for (uint32_t i = 0; i < n; i++)
        if (!scene->objects[i].casts_shadow) hdr->all_cast = 0;
    hdr->cull_flags = (P.light_cull ? CULL_ENABLED : 0u) | (P.dark ? CULL_DARK : 0u) | (P.fast_shadow ? CULL_FAST_SHADOW : 0u) | (P.cell_cull ? CULL_CELLS : 0u);
    hdr->uvrec_off = (uint32_t)soa->size();
    soa->insert(soa->end(), uvrec.begin(), uvrec.end());
    if (cam) {
        if (cam->width == 0 || cam->height == 0) return fail(RTC_ERR_INVALID_ARG, "empty canvas");
        if (!is_affine(cam->inv)) return fail(RTC_ERR_UNSUPPORTED, "camera inverse transform is not affine");
        hdr->width = cam->width;
        hdr->height = cam->height;
        hdr->half_w = cam->half_width;
        hdr->half_h = cam->half_height;
        hdr->pixel_size = cam->pixel_size;
        std::memcpy(hdr->cam, cam->inv, sizeof(float) * 12);
        // camera.rs:70: origin = transform_inverse * point(0,0,0) -- pixel-invariant, computed once here
        const float zero[4] = {0.0f, 0.0f, 0.0f, 1.0f};
        float org[4];
        mat_vec4(cam->inv, zero, org);
        for (int k = 0; k < 3; k++) hdr->cam_origin[k] = org[k];
    }
    return RTC_OK;
}

}  // namespace rtc

using namespace rtc;

struct rtc_ctx_tiles {
    const uint8_t* bits;
    uint32_t w, h;
};
struct BlockList {  // RenderArgs::tiles of one partition (build_block_list), resident on the device
    uint32_t* d = nullptr;
    size_t n = 0, n_listed = 0;  // (n_listed: a grid's list leaves the padding out)
    // feedback (refine_block_list): the list as built, where its first launch leaves its waves' running times, and how far it is
    std::vector<uint32_t> host;
    size_t d_cap = 0, ticks_cap = 0;  // bytes behind d / d_ticks: grown, never shrunk (hipMalloc / hipFree cost the animation's frames milliseconds)
    uint32_t* d_ticks = nullptr;  // [4 n] wave times -- or, for kernels that do not time their waves, [4 n] uint4 work counts (a copy of block_counts)
    bool counts = false, swizzled = false, listed = false;  // (listed: a grid whose frames run from `d`, its blocks in order)
    // IDLE (regular grids): a scene's first frame -- nothing is measured before a second frame of the SAME scene shows that frames repeat
    enum { FRESH, TIMED, REFINED, IDLE } state = FRESH;
    uint32_t passes = 0;  // refinements so far
    // Scenes that change (restart_block_lists): once a scene has changed under a refined list, every frame from it leaves its waves'
    // times and is bracketed by the list's own events; what the frames since the last re-cut took beyond the best of them adds up
    // (loss_ms), and when that reaches what a re-cut costs the host (recut_host_ms, measured) the next frame re-cuts the list from
    // the times of the frame before it -- at the latest after sixteen scenes.
    uint32_t scene_changes = 0;
    bool animated = false, ev_recorded = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double best_ms = 0.0, loss_ms = 0.0, recut_host_ms = 0.5;
};
struct rtc_ctx {
    int device = 0;
    Policy policy;  // the environment's switches as they were when the context was created
    SceneHdr hdr;
    bool has_scene = false;
    float4* d_soa = nullptr;
    size_t soa_cap = 0;  // float4 entries
    float* d_texels = nullptr;  // UVImage canvases (RGB f32), grow-only
    size_t texel_cap = 0;       // floats
    uint32_t n_objects = 0;
    bool simple = false;  // every object scale+translate-only, no cylinder / cone, no patterns
    // workspace of rtc_ctx_to_ppm (grow-only)
    unsigned long long* d_ppm_rows = nullptr;  // per-row length, then offset; [h] is the total
    uint32_t* d_ppm_bits = nullptr;
    size_t ppm_rows_cap = 0, ppm_bits_cap = 0;
    int n_cus = 0;                    // compute_units()
    bool spec_shares = false;         // spec_fn was compiled with -DRTC_SPEC_SHARE=1
    bool spec_blocks_y = false;       // ... with -DRTC_SPEC_BLOCKS_Y=1 (several blocks per workgroup)
    bool spec_rect = false;           // ... with -DRTC_SPEC_RECT=1 (scene rectangle launches: block offsets, zero-filling workgroups)
    hipFunction_t spec_fn = nullptr;  // scene-specialised kernel (hiprtc), or null: ahead-of-time kernels
    // Recursion deeper than RTC_AOT_MAX_DEPTH (ctx_render_slot): the scene's kernel compiled once more with a longer frame
    // stack (-DRTC_SPEC_MAX_DEPTH=16 / 32 / ...), on first use; spec_defs: the options of this scene's kernel (empty when
    // the policy left the scene to the ahead-of-time kernels -- deep_defs() then writes the options from scratch)
    std::vector<std::string> spec_defs;
    std::string spec_name;
    std::map<int, hipFunction_t> deep_fn;
    std::string kernel_name;          // what rtc_ctx_render launches, for rtc_ctx_kernel_name()
    // Block list of the current scene (RenderArgs::tiles): which 16 x 16 pixel tiles of the image a mesh projects to
    // (row-major bitmap, empty: no block list), and the list last built -- for the partition it was built for
    int tree_waves = 6;  // waves per SIMD the scene's tree kernel was compiled for (rtc_ctx_set_scene)
    bool lazy_jit = false;      // a context of the one-call seam (rtc::ctx_mark_one_shot): see jit_get
    bool jit_deferred = false;  // the resident scene's kernel was left uncompiled by its first sighting: the next render of this scene compiles it
    std::vector<uint8_t> heavy_tiles;
    uint32_t heavy_w = 0, heavy_h = 0;
    // key: band_rows, n_parts, part, lanes per pixel (log2; ~0: a regular grid's order), depth (what a block costs depends on it)
    std::map<std::array<uint32_t, 5>, BlockList> block_lists;
    float scene_box_coverage = 1.0f;  // share of the image the scene's box projects to (1: unknown / all of it)
    uint32_t scene_rect[4] = {0u, 0u, 0u, 0u};  // the 16 x 16 tiles outside which no primary ray sees anything: [x0, x1) x [y0, y1); empty: unknown
    // Scene tiles: of a bounded world whose entries project to a small part of the frame (C5: 64 spheres, 7 % of 8192^2), the 16 x 16
    // tiles some entry's padded box projects to.  Frames of such a scene are a zero-fill of the canvas and one workgroup per
    // listed tile (ctx_render_slot), instead of the bounding rectangle of them all.  Empty: not known / not worth it.
    std::vector<uint8_t> scene_tile_mask;
    uint32_t scene_tiles_w = 0u, scene_tiles_h = 0u;
    struct SceneTileList {
        uint32_t* d = nullptr;
        size_t n = 0;
        unsigned long long traced_pixels = 0ull;  // traced pixels (x < w - 1, y < h - 1) inside the listed tiles
        uint2* d_fill = nullptr;  // the runs of tiles that are NOT listed, at most 64 tiles each: {x0 | n << 16, local row} (fill_tiles_kernel)
        size_t n_fill = 0;
    };
    hipStream_t fill_stream = nullptr;  // the zero-fill of a tile launch runs beside the render kernel (fork / join by events)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::map<std::array<uint32_t, 3>, SceneTileList> scene_tile_lists;  // per partition {band_rows, n_parts, part}
    float scene_rect_coverage = 1.0f;           // ... and its share of the frame
    std::string kernel_id;            // rtc_ctx_kernel_id(): names the code object (source + options + compiler), not the scene
    std::string jit_note;             // why spec_fn is null although the policy wanted one (rtc_ctx_jit_status)
    // the scene as last uploaded: an identical one (rtc_render_ex called again for the next frame) is not uploaded twice
    std::vector<float4> soa_host;
    std::set<std::pair<const void*, const void*>> warmed;  // (kernel, stream) pairs that have been launched once (ctx_render_slot)
    std::vector<float> texels_host;
    uint4* d_block_counts = nullptr;
    size_t block_cap = 0;
    // pinned host memory through which the feedback reads wave times back and sends lists out (grow-only: through pageable vectors
    // the two copies of a 2048^2 frame's re-cut took longer than the frame)
    void* h_feedback = nullptr;
    size_t h_feedback_cap = 0;
    // the level-by-level renderer (rtc_wavefront.h): ray lists (two levels x reflection / refraction), the node pool, counters
    WfRay* d_wf_rays[4] = {nullptr, nullptr, nullptr, nullptr};
    WfNode* d_wf_nodes = nullptr;
    uint32_t* d_wf_ctr = nullptr;
    size_t wf_cap_rays = 0, wf_cap_nodes = 0;
    bool wf_pays = false;      // this scene: a tree world with long leaf runs whose materials both reflect and transmit
    bool wf_disabled = false;  // ... but a frame overflowed the pools: per-pixel kernels from then on
    bool wf_last = false;      // the last frame was rendered level by level (rtc_ctx_kernel_name says so)
    std::string wf_name;
    uint32_t* d_progress = nullptr;  // RenderArgs::progress counters (rtc_render_ex), grow-only
    size_t progress_cap = 0;         // dwords
    // {rays, shaded hits, culled shadow rays} per counter slot: slot 0 = the last rtc_ctx_render launch; rtc_render_ex
    // renders a frame in several launches (row chunks in flight while earlier ones travel) and gives each its own
    unsigned long long* d_total = nullptr;
    // HIP-event pairs around the render kernel, one per launch since the last rtc_ctx_stats
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    bool rendered = false;
    uint32_t last_rows = 0;
    uint64_t last_pixels = 0;
};

// A new scene of the same frame size (an animation: the camera or an object has moved): what a tile cost in the frame before is
// still the best guess for what it costs now, and any list is a valid tiling for any scene.  The lists that cut tiles into lanes
// (divided meshes, lane-sharing area lights) stay, and from then on every frame from a refined list leaves its waves' times and is
// bracketed by the list's own events: what the frames since the last re-cut took beyond the best of them adds up, and when that
// reaches what a re-cut costs the host (a read-back, a sort, an upload: 0.2 - 0.5 ms, measured) the next frame starts with a re-cut
// from the times of the frame before it -- rent until the rent equals the price.  Lists age at their own pace (mesh 2048^2, a
// quarter of a degree per frame: 0.1 - 0.3 ms per frame, re-cut every 2 - 9 scenes, kernel 3.85 -> 2.97 ms; soft_shadows 2048^2:
// hardly, every sixteenth -- the cap); a fixed period was wrong for one or the other (every eighth: mesh 3.57 with frames of 5).  A list whose timed frame belonged to the old scene is re-cut from that: it is the frame before.  A regular grid's
// ORDER does not survive: a stale order was slightly worse than the permuted image order (reflect_refract 0.839 -> 0.860 ms).
static void restart_block_lists(rtc_ctx* c) {
    for (auto it = c->block_lists.begin(); it != c->block_lists.end();) {
        BlockList& bl = it->second;
        if (it->first[3] == 0xffffffffu) {  // (its buffers stay for the next scene that repeats)
            bl.state = BlockList::IDLE;
            bl.listed = false;
            ++it;
            continue;
        }
        if (bl.state == BlockList::REFINED) {
            bl.animated = true;
            bool recut = ++bl.scene_changes >= 16u, timed = false;
            if (bl.ev_recorded) {  // the frame before ran from this list (rtc_ctx_set_scene has waited for it)
                bl.ev_recorded = false;
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, bl.ev0, bl.ev1) == hipSuccess && ms > 0.0f) {
                    timed = true;
                    if (bl.best_ms == 0.0 || ms < bl.best_ms) bl.best_ms = ms;
                    bl.loss_ms += ms - bl.best_ms;
                    if (bl.loss_ms >= bl.recut_host_ms) recut = true;
                }
            }
            if (recut) {
                bl.scene_changes = 0u;
                bl.best_ms = bl.loss_ms = 0.0;
                // (timed: that frame's times are on the device -- the next frame starts with the re-cut; else it is timed first)
                bl.state = timed ? BlockList::TIMED : BlockList::FRESH;
                bl.passes = c->policy.feedback_passes ? c->policy.feedback_passes - 1u : 0u;
            }
        }
        ++it;
    }
}
// every block list of the context, with what its feedback holds (the caller knows that no launch is reading them)
static void drop_scene_tile_lists(rtc_ctx* c) {
    for (auto& tl : c->scene_tile_lists) {
        if (tl.second.d) (void)hipFree(tl.second.d);
        if (tl.second.d_fill) (void)hipFree(tl.second.d_fill);
    }
    c->scene_tile_lists.clear();
}
static void drop_block_lists(rtc_ctx* c) {
    for (auto& bl : c->block_lists) {
        if (bl.second.d) (void)hipFree(bl.second.d);
        if (bl.second.d_ticks) (void)hipFree(bl.second.d_ticks);
        if (bl.second.ev0) (void)hipEventDestroy(bl.second.ev0);
        if (bl.second.ev1) (void)hipEventDestroy(bl.second.ev1);
    }
    c->block_lists.clear();
}

// ============================================================================
//  Scene-specialised kernels (hiprtc)
//
//  The generic kernels decide each object's shape kind and flags with wave-uniform branches inside the
//  unrolled object loops.  For a given scene those words are constants, so rtc_ctx_set_scene compiles
//  rtc_kernel_core.h once more with them baked in (-DRTC_SPEC_LIST=...): no kind switches, dead
//  shape code removed, tighter scheduling -- C3 4.06 -> 3.13 ms with bit-identical output.  Only the
//  scene's *shape* (object count, kinds, flags, light kind, jitter mode) is specialised; all values
//  stay run-time data.  Compiled code is cached in-process and on disk (<lib dir>/jit_cache/).
//  Policy: RTC_AMD_SPECIALIZE=0 never, =1 always; default: scenes of <= 8 objects rendered at >= 2^18
//  pixels (a 0.7 s compile is not worth it for thumbnails; the AOT kernels produce the same bits).
// ============================================================================
namespace {

struct CacheHeader {
    char magic[8];
    uint64_t size, checksum;
};
struct JitModule {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    std::string id;  // "spec_<hash of source, options, compiler version>.<checksum of the code object>": names the code that runs (rtc_ctx_kernel_id)
};
std::mutex g_jit_mutex;
std::map<std::string, JitModule> g_jit_cache;  // key: "<device>|<defines>"

std::string lib_dir() {
    Dl_info info;
    if (dladdr((const void*)&rtc_abi_version, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t k = p.find_last_of('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char ch : s) {
        h ^= ch;
        h *= 1099511628211ull;
    }
    return h;
}

bool read_file(const std::string& path, std::string* out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    *out = ss.str();
    return true;
}

// The kernel source travels inside the library: rtc_kernel_core_embed.inc is rtc_kernel_core.h as a string literal,
// written by ray_tracer_challenge_amd/build.py before every compile (under hiprtc the header needs no other file).  A
// deployment is librtc_amd.so alone -- no csrc/ or include/ beside it.  RTC_AMD_JIT_SOURCE=<path> (development) reads
// the header from disk instead, so a kernel experiment needs no rebuild of the library.
const char k_core_src[] =
#include "rtc_kernel_core_embed.inc"
    ;

std::string jit_cache_dir(const Policy& P) {  // RTC_AMD_JIT_CACHE=<dir>, or 0 / off to keep compiled kernels in memory only; default <lib dir>/jit_cache
    if (!P.jit_cache.empty()) return (P.jit_cache == "0" || P.jit_cache == "off") ? std::string() : P.jit_cache;
    return lib_dir() + "/jit_cache";
}

// Compiles (or fetches) the specialised kernel for `defines` on the current device.
// `lazy` (the one-call seam, rtc_render_ex: the reference renders ONE frame per process): a kernel that is neither in this process's
// memory nor in the disk cache is not compiled the first time its scene is seen -- *out stays null and the frame is rendered by the
// ahead-of-time kernel: a compile is 0.5 - 2 s, the frame it speeds up a few milliseconds (tools/first_call.py: C3's first call 693 ms
// with the compile, 121 with a filled cache, 123 ahead-of-time).  The second time the process asks for the same kernel -- frames
// repeat -- it is compiled, and cached on disk for every process after it.
std::set<std::string> g_jit_seen;
rtc_status jit_get(const Policy& P, int device, const std::vector<std::string>& defines, hipFunction_t* out, std::string* id, bool lazy = false) {
    std::string key = std::to_string(device) + "|";
    for (const auto& d : defines) key += d + " ";
    key += "|" + P.jit_source + "|" + P.jit_flags;  // (development builds: another source or other flags are another kernel)
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    auto it = g_jit_cache.find(key);
    if (it != g_jit_cache.end()) {
        *out = it->second.fn;
        *id = it->second.id;
        return RTC_OK;
    }
    std::string core_file;
    const char* core = k_core_src;
    if (!P.jit_source.empty()) {  // development builds only (Policy)
        if (!read_file(P.jit_source, &core_file)) return fail(RTC_ERR_DEVICE, "scene specialisation: cannot read the kernel source %s", P.jit_source.c_str());
        core = core_file.c_str();
    }
    std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};
    for (const auto& d : defines) opts.push_back(d);
    // occupancy target of the specialised kernel: measured 4 -> 3.39, 5 -> 3.42, 6 -> 3.24, 7 -> 3.17, 8 -> 3.17 ms (C3) before
    // light-cone culling; with it (more state per shade point) 5 -> 0.98, 6 -> 0.97, 7 -> 0.95, 8 -> 1.02 ms (tools/ab_env.py over -DRTC_WAVES_PER_SIMD)
    bool waves_given = false;
    for (const auto& d : defines) waves_given = waves_given || d.rfind("-DRTC_WAVES_PER_SIMD=", 0) == 0;
    if (!waves_given) opts.push_back("-DRTC_WAVES_PER_SIMD=7");
    if (!P.jit_flags.empty()) {  // development builds only: extra -D / -m flags, space separated
        std::istringstream ss(P.jit_flags);
        std::string tok;
        while (ss >> tok) opts.push_back(tok);
    }
    // disk cache keyed by the source text, every option, the compiler's version and the ABI the argument block follows
    std::string opt_text;
    for (const auto& o : opts) opt_text += o + "\n";
    if (P.jit_print) {  // development: the options, one line, as tools/spec_asm.sh takes them
        std::string line;
        for (size_t i = 5; i < opts.size(); i++) line += " " + opts[i];
        std::fprintf(stderr, "librtc_amd: scene kernel options:%s\n", line.c_str());
    }
    int rtc_major = 0, rtc_minor = 0;
    (void)hiprtcVersion(&rtc_major, &rtc_minor);
    opt_text += "hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) + " abi " + std::to_string(RTC_ABI_VERSION) +
                " args " + std::to_string(sizeof(RenderArgs)) + "\n";
    // (Not in the key: WHICH libhiprtc this process holds.  A Python process that imported torch first compiles with the wheel's
    // bundled compiler, the same script under rocprofv3 -- which puts /opt/rocm/lib first in LD_LIBRARY_PATH -- with the system's;
    // both report one hiprtcVersion and emit different, equally valid code for these kernels (same images, same speed:
    // profiles/r04_ab_compilers.txt).  Sharing the entry is what lets a profiled run measure the very code object a plain run
    // compiled -- profiles/run_profile.sh compiles first, plainly -- and the id below says which binary it was.)
    char name[64];
    snprintf(name, sizeof(name), "spec_%016llx.hsaco", (unsigned long long)fnv1a(opt_text, fnv1a(core)));
    const std::string cache_dir = jit_cache_dir(P), cache_path = cache_dir + "/" + name;
    auto compile = [&](std::string* code) -> rtc_status {
        hiprtcProgram prog;
        const char* src = "#include \"rtc_kernel_core.h\"\n";
        const char* headers[] = {core};
        const char* header_names[] = {"rtc_kernel_core.h"};
        if (hiprtcCreateProgram(&prog, src, "rtc_scene_spec.hip", 1, headers, header_names) != HIPRTC_SUCCESS)
            return fail(RTC_ERR_DEVICE, "hiprtcCreateProgram failed");
        std::vector<const char*> copts;
        for (const auto& o : opts) copts.push_back(o.c_str());
        hiprtcResult r = hiprtcCompileProgram(prog, (int)copts.size(), copts.data());
        if (r != HIPRTC_SUCCESS) {
            size_t n = 0;
            hiprtcGetProgramLogSize(prog, &n);
            std::string log(n, '\0');
            if (n) hiprtcGetProgramLog(prog, &log[0]);
            hiprtcDestroyProgram(&prog);
            return fail(RTC_ERR_DEVICE, "scene specialisation failed to compile: %s\n%.1500s", hiprtcGetErrorString(r), log.c_str());
        }
        size_t n = 0;
        hiprtcGetCodeSize(prog, &n);
        code->resize(n);
        hiprtcGetCode(prog, &(*code)[0]);
        hiprtcDestroyProgram(&prog);
        // best effort: a read-only tree just means every process compiles for itself.  Only a completely written file
        // is published (a short write -- disk full, quota -- would otherwise poison the cache for every later run).
        if (!cache_dir.empty() && (::mkdir(cache_dir.c_str(), 0777) == 0 || errno == EEXIST)) {
            const std::string tmp = cache_path + "." + std::to_string((long)getpid());
            bool ok = false;
            {
                std::ofstream f(tmp, std::ios::binary);
                if (f) {
                    const CacheHeader h = {{'R', 'T', 'C', 'J', 'I', 'T', '1', 0}, (uint64_t)code->size(), fnv1a(*code)};
                    f.write((const char*)&h, sizeof(h));
                    f.write(code->data(), (std::streamsize)code->size());
                    f.close();
                    ok = f.good();
                }
            }
            if (!ok || std::rename(tmp.c_str(), cache_path.c_str()) != 0) (void)::unlink(tmp.c_str());
        }
        return RTC_OK;
    };
    // a cache entry is its header {magic, size, checksum} + the code object: a truncated or foreign file is ignored (the
    // HIP runtime does not survive a damaged code object), and overwritten by the fresh compile
    std::string code;
    bool cached = false;
    if (!cache_dir.empty() && read_file(cache_path, &code) && code.size() > sizeof(CacheHeader)) {
        CacheHeader h;
        std::memcpy(&h, code.data(), sizeof(h));
        code.erase(0, sizeof(h));
        cached = std::memcmp(h.magic, "RTCJIT1", 8) == 0 && h.size == code.size() && h.checksum == fnv1a(code);
    }
    if (!cached && lazy && g_jit_seen.insert(key).second) {  // (first sighting: see above)
        *out = nullptr;
        return RTC_OK;
    }
    if (!cached) {
        rtc_status st = compile(&code);
        if (st != RTC_OK) return st;
    }
    JitModule m;
    hipError_t le = hipModuleLoadData(&m.mod, code.data());
    if (le == hipSuccess) le = hipModuleGetFunction(&m.fn, m.mod, "render_kernel_spec");
    if (le != hipSuccess && cached) {
        // a cached code object that does not load (truncated, or from another toolchain): drop it and compile once
        (void)hipGetLastError();
        (void)::unlink(cache_path.c_str());
        rtc_status st = compile(&code);
        if (st != RTC_OK) return st;
        le = hipModuleLoadData(&m.mod, code.data());
        if (le == hipSuccess) le = hipModuleGetFunction(&m.fn, m.mod, "render_kernel_spec");
    }
    if (le != hipSuccess) return fail(RTC_ERR_DEVICE, "scene specialisation: the compiled kernel does not load: %s", hipGetErrorString(le));
    // The id names the code object itself: "spec_<hash of source, options, compiler>.<checksum of the compiled code>".  The second
    // half is there because one source does not always give one binary: a hiprtc compile inside a process started under rocprofv3
    // came out different from the same compile in a plain process (LABNOTES "Round 4": 436 against 484 B of scratch per lane, 421 M
    // against 387 M VALU instructions a C3 frame), and a profile must never be quoted for a binary it did not measure.
    char sum[16];
    snprintf(sum, sizeof(sum), ".%08x", (unsigned)(fnv1a(code) & 0xffffffffu));
    m.id = std::string(name, std::strlen(name) - 6) + sum;  // without ".hsaco"
    g_jit_cache[key] = m;
    *out = m.fn;
    *id = m.id;
    return RTC_OK;
}

// identifies the ahead-of-time kernels of this build: the source they were compiled from
std::string aot_kernel_id() {
    char b[40];
    snprintf(b, sizeof(b), "aot_%016llx", (unsigned long long)fnv1a(k_core_src));
    return b;
}

}  // namespace

static SceneSoA soa_view(const float4* base, const SceneHdr& hdr, const float* d_texels) {
    uint32_t m = rtc::padded_count(hdr.n_objects);
    SceneSoA s;
    s.uvrec = base + hdr.uvrec_off;
    s.texels = d_texels;
    s.geo = base + 0 * (size_t)m;
    s.off0 = base + 1 * (size_t)m;
    s.off1 = base + 2 * (size_t)m;
    s.off2 = base + 3 * (size_t)m;
    s.mat_a = base + 4 * (size_t)m;
    s.mat_b = base + 5 * (size_t)m;
    s.mat_c = base + 6 * (size_t)m;
    s.pat = base + 7 * (size_t)m;
    s.tri = base + 12 * (size_t)m;
    s.lcorn = base + 15 * (size_t)m;
    s.trn = base + 18 * (size_t)m;
    s.bsph = base + 19 * (size_t)m;
    s.trav = base + 20 * (size_t)m;
    return s;
}

extern "C" {

int32_t rtc_device_count(void) { return usable_devices(); }

// Diagnostic (not in rtc.h): the level-by-level renderer's counters after the context's last frame (rtc_wavefront.h: nodes,
// overflow, then per level {reflection rays, refraction rays, nodes so far}); returns the number of words written.
uint32_t rtc_ctx_wavefront_counters(rtc_ctx* c, uint32_t* out, uint32_t cap) {
    if (!c || !out || !c->d_wf_ctr) return 0u;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    const uint32_t n = std::min<uint32_t>(cap, WF_CTR_WORDS);
    if (hipMemcpy(out, c->d_wf_ctr, n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return 0u;
    return n;
}

// Diagnostic (not in rtc.h): was this library built with the development switches (Policy, -DRTC_DEV_SWITCHES)?
int32_t rtc_dev_switches(void) {
#ifdef RTC_DEV_SWITCHES
    return 1;
#else
    return 0;
#endif
}

rtc_status rtc_scene_validate(const rtc_scene* scene, const rtc_camera* camera) {
    SceneHdr hdr;
    std::vector<float4> soa;
    std::vector<float> texels;
    return flatten(Policy::from_env(), scene, camera, &hdr, &soa, &texels);
}

rtc_status rtc_ctx_create(int32_t device, rtc_ctx** out) {
    if (!out) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_create: out is NULL");
    int n = usable_devices();
    if (n <= 0) return fail(RTC_ERR_NO_DEVICE, "no HIP device visible; librtc_amd has no CPU fallback");
    if (device < 0 || device >= n) return fail(RTC_ERR_INVALID_ARG, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    rtc_ctx* c = new rtc_ctx();
    c->device = device;
    c->policy = Policy::from_env();  // the one place a context looks at the environment
    HIP_TRY(hipMalloc(&c->d_total, 3 * CTX_TOTAL_SLOTS * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->d_total, 0, 3 * CTX_TOTAL_SLOTS * sizeof(unsigned long long)));
    *out = c;
    return RTC_OK;
}

void rtc_ctx_destroy(rtc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_soa) (void)hipFree(c->d_soa);
    if (c->d_texels) (void)hipFree(c->d_texels);
    if (c->d_block_counts) (void)hipFree(c->d_block_counts);
    if (c->d_progress) (void)hipFree(c->d_progress);
    for (WfRay* r : c->d_wf_rays)
        if (r) (void)hipFree(r);
    if (c->d_wf_nodes) (void)hipFree(c->d_wf_nodes);
    if (c->d_wf_ctr) (void)hipFree(c->d_wf_ctr);
    if (c->d_total) (void)hipFree(c->d_total);
    if (c->d_ppm_rows) (void)hipFree(c->d_ppm_rows);
    if (c->d_ppm_bits) (void)hipFree(c->d_ppm_bits);
    drop_block_lists(c);
    drop_scene_tile_lists(c);
    if (c->fill_stream) (void)hipStreamDestroy(c->fill_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->h_feedback) (void)hipHostFree(c->h_feedback);
    for (auto& e : c->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete c;
}

// Which 16 x 16 pixel tiles do the boxes of the mesh-holding groups project to?  ray_for_pixel (camera.rs:60-74) sends
// pixel (px, py) through the camera-space point (half_width - (px + 0.5) s, half_height - (py + 0.5) s, -1); a world
// point maps to camera space through the inverse of Camera.transform_inverse.  Performance only -- which blocks start
// first and with how many lanes per pixel -- so generous padding and "everything" when a box reaches behind the camera.
constexpr size_t HEAVY_BOX_FLOATS = 7;  // min, max, rank (1 .. 3: project_heavy_boxes keeps a tile's highest)
static void project_heavy_boxes(const Policy& P, const std::vector<float>& boxes, const rtc_camera* cam, std::vector<uint8_t>* tiles, uint32_t* tw, uint32_t* th) {
    tiles->clear();
    *tw = *th = 0;
    if (boxes.empty() || !cam || !P.block_list) return;
    float view[16];
    inverse4(cam->inv, view);
    const uint32_t w = (cam->width + 15u) / 16u, h = (cam->height + 15u) / 16u;
    tiles->assign((size_t)w * h, 0);
    *tw = w;
    *th = h;
    for (size_t b = 0; b + HEAVY_BOX_FLOATS - 1 < boxes.size(); b += HEAVY_BOX_FLOATS) {
        double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        bool everything = false;
        for (int k = 0; k < 8 && !everything; k++) {
            const float p[4] = {boxes[b + ((k & 1) ? 3 : 0)], boxes[b + ((k & 2) ? 4 : 1)], boxes[b + ((k & 4) ? 5 : 2)], 1.0f};
            float q[4];
            mat_vec4(view, p, q);
            if (!(q[2] < -1e-4f) || !std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2])) {
                everything = true;
                break;
            }
            const double px = ((double)cam->half_width - (double)q[0] / -(double)q[2]) / cam->pixel_size - 0.5;
            const double py = ((double)cam->half_height - (double)q[1] / -(double)q[2]) / cam->pixel_size - 0.5;
            x0 = std::fmin(x0, px), x1 = std::fmax(x1, px), y0 = std::fmin(y0, py), y1 = std::fmax(y1, py);
        }
        if (everything) x0 = y0 = -1e9, x1 = y1 = 1e9;
        const long tx0 = std::max(0L, (long)std::floor((x0 - 8.0) / 16.0)), tx1 = std::min((long)w - 1, (long)std::floor((x1 + 8.0) / 16.0));
        const long ty0 = std::max(0L, (long)std::floor((y0 - 8.0) / 16.0)), ty1 = std::min((long)h - 1, (long)std::floor((y1 + 8.0) / 16.0));
        for (long ty = ty0; ty <= ty1; ty++)
            for (long tx = tx0; tx <= tx1; tx++) (*tiles)[(size_t)ty * w + tx] = std::max((*tiles)[(size_t)ty * w + tx], (uint8_t)boxes[b + 6]);
    }
}

// A top-level plane (plane.rs:45-56) is hit by a primary ray only if the ray points towards it: with y_o the plane's
// object-space height of the camera and y_d that of the direction, t = -y_o / y_d >= 0 needs y_d of the other sign.  The
// direction through pixel (px, py) is linear in (px, py) (ray_for_pixel, camera.rs:60-74), hence so is y_d: the pixels that can
// see the plane are one side of a straight line -- the horizon -- taken here with 8 pixels to spare, and marked as the 16 x 16
// tiles of their bounding rectangle (the launch is a rectangle anyway).  A camera in the plane, or not finite: everything.
static void mark_plane_side(const std::array<double, 4>& row, const rtc_camera* cam, std::vector<uint8_t>* tiles, uint32_t tw, uint32_t th) {
    const float* m = cam->inv;
    const double org[3] = {m[3], m[7], m[11]};
    const double y_o = row[0] * org[0] + row[1] * org[1] + row[2] * org[2] + row[3];
    // y_d(px, py) = row . M3 (half_w - (px + 0.5) s, half_h - (py + 0.5) s, -1) = a px + b py + c0
    double col[3];  // row . (columns of the camera matrix's linear part)
    for (int k = 0; k < 3; k++) col[k] = row[0] * m[k] + row[1] * m[4 + k] + row[2] * m[8 + k];
    const double sz = cam->pixel_size, a = -col[0] * sz, b = -col[1] * sz;
    const double c0 = col[0] * (cam->half_width - 0.5 * sz) + col[1] * (cam->half_height - 0.5 * sz) - col[2];
    auto all = [&]() { std::fill(tiles->begin(), tiles->end(), (uint8_t)1); };
    if (!std::isfinite(y_o) || !std::isfinite(a) || !std::isfinite(b) || !std::isfinite(c0) || y_o == 0.0) return all();
    // visible where sign(y_o) * y_d < 0; keep everything with g(px, py) = sign(y_o) * y_d - margin < 0
    const double sgn = y_o > 0.0 ? 1.0 : -1.0, margin = 8.0 * (std::fabs(a) + std::fabs(b));
    auto g = [&](double px, double py) { return sgn * (a * px + b * py + c0) - margin; };
    const double W = cam->width, Hh = cam->height;
    const double cx[4] = {0.0, W, W, 0.0}, cy[4] = {0.0, 0.0, Hh, Hh};
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int k = 0; k < 4; k++) {
        const int n = (k + 1) & 3;
        const double gk = g(cx[k], cy[k]), gn = g(cx[n], cy[n]);
        if (gk < 0.0) x0 = std::fmin(x0, cx[k]), x1 = std::fmax(x1, cx[k]), y0 = std::fmin(y0, cy[k]), y1 = std::fmax(y1, cy[k]);
        if ((gk < 0.0) != (gn < 0.0)) {  // the line crosses this edge of the image
            const double t = gk / (gk - gn), ex = cx[k] + t * (cx[n] - cx[k]), ey = cy[k] + t * (cy[n] - cy[k]);
            x0 = std::fmin(x0, ex), x1 = std::fmax(x1, ex), y0 = std::fmin(y0, ey), y1 = std::fmax(y1, ey);
        }
    }
    if (!(x0 <= x1) || !(y0 <= y1)) return;  // the plane is behind every pixel's ray
    const long tx0 = std::max(0L, (long)std::floor(x0 / 16.0) - 1), tx1 = std::min((long)tw - 1, (long)std::floor(x1 / 16.0) + 1);
    const long ty0 = std::max(0L, (long)std::floor(y0 / 16.0) - 1), ty1 = std::min((long)th - 1, (long)std::floor(y1 / 16.0) + 1);
    for (long ty = ty0; ty <= ty1; ty++)
        for (long tx = tx0; tx <= tx1; tx++) (*tiles)[(size_t)ty * tw + tx] = 1;
}

// RenderArgs::tiles' words: lanes per pixel 2^s (s = 0 .. 4), pixel origin (multiples of 4; local rows below 2^17)
static inline uint32_t tile_word(uint32_t s, uint32_t x0, uint32_t y0) { return (s & 3u) << 30 | (x0 / 4u) << 16 | (s >> 2) << 15 | (y0 / 4u); }
static inline uint32_t tile_s(uint32_t t) { return (t >> 30) | ((t >> 13) & 4u); }
static inline uint32_t tile_x0(uint32_t t) { return ((t >> 16) & 0x3fffu) << 2; }
static inline uint32_t tile_y0(uint32_t t) { return (t & 0x7fffu) << 2; }
// The block list of one partition (RenderArgs::tiles): the 16 x 16 tiles of the partition's compact rows, those a mesh
// projects to first and cut into blocks of 2^mesh_share_log2 lanes per pixel (8 x 8 or 8 x 4 pixels), the others after
// them, whole, one lane per pixel.
static void build_block_list(const Policy& P, const rtc_ctx_tiles& T, uint32_t width, uint32_t mesh_share_log2, uint32_t rows, const Partition& q, std::vector<uint32_t>* out) {
    out->clear();
    std::vector<uint32_t> light;
    uint32_t hs = mesh_share_log2;  // lanes per pixel (log2) in the mesh tiles; RTC_AMD_BLOCK_S=0..3: development
    // the dearest tiles first (rank 3: glass that also reflects), or the frame ends waiting for a few waves that started
    // late (RTC_AMD_BLOCK_ORDER=0: image order; RTC_AMD_BLOCK_S_TOP=0..3: lanes per pixel of rank 3 alone -- development)
    const bool ordered = P.block_order;
    uint32_t hs_top = hs;
    // large frames at two lanes: only the glass keeps them, the other meshes' tiles take one (first frames of here_be_dragons
    // 4000 x 1600 2.87 -> 2.68 ms, mesh 2048^2 3.47 -> 3.24; at 1024^2 and below the other way round: 2.38 -> 2.70,
    // profiles/r03_ab_first_frame_lanes.txt)
    if (hs == 1u && ((uint64_t)width * rows + 63u) / 64u > 60000u) hs = 0u;
    if (P.block_s >= 0) hs = hs_top = (uint32_t)P.block_s;
    if (P.block_s_top >= 0) hs_top = (uint32_t)P.block_s_top;
    for (uint32_t rank = 3u; rank >= 1u; rank--) {
        const uint32_t s = rank == 3u ? hs_top : hs;
        const uint32_t hbw = 16u >> (s >> 1), hbh = 16u >> ((s + 1u) >> 1);
        for (uint32_t yl0 = 0; yl0 < rows; yl0 += 16u) {
            const uint32_t band = yl0 / q.band_rows;
            const uint32_t y = (band * q.n_parts + q.part) * q.band_rows + (yl0 - band * q.band_rows);  // global row of the tile's first row
            for (uint32_t x0 = 0; x0 < width; x0 += 16u) {
                const uint32_t ty = std::min(y / 16u, T.h - 1u), tx = std::min(x0 / 16u, T.w - 1u);
                const uint32_t r = T.bits[(size_t)ty * T.w + tx];
                if (r != 0u && (ordered ? r == rank : rank == 1u)) {
                    for (uint32_t dy = 0; dy < 16u && yl0 + dy < rows; dy += hbh)
                        for (uint32_t dx = 0; dx < 16u && x0 + dx < width; dx += hbw)
                            out->push_back(tile_word(s, x0 + dx, yl0 + dy));
                } else if (rank == 1u && r == 0u) {
                    light.push_back(tile_word(0u, x0, yl0));
                }
            }
        }
    }
    out->insert(out->end(), light.begin(), light.end());
}

// Feedback for block lists.  The list a scene starts with knows three kinds of tile (build_block_list) and nothing of what a tile
// costs; the frame it schedules ends with a tail -- here_be_dragons 4000 x 1600: waves of 2.3 ms that started at 0.8 ms of a 3.1 ms
// frame; mesh 2048^2: the machine runs out of waves at 2.2 ms, the longest (two lanes per pixel, the centre of the glass mesh)
// run to 3.4.  The first launch of a list therefore times its waves (RenderArgs::wave_ticks), and the list of every later frame
// of this scene and partition is made from those times: a 16 x 16 tile whose longest wave ran more than half of the frame's
// throughput time (the sum of all waves' times over the wave slots of the device) gets more lanes per pixel, each doubling
// taken to shorten its waves to 0.7 (measured: tools/ab_env.py over RTC_AMD_BLOCK_S), and the tiles start in the order of their predicted
// longest wave.  Which lanes trace a pixel and when changes nothing about its value (tests/test_gpu_fullsize.py compares first
// and later frames with the oracle).
static void refine_block_list(const std::vector<uint32_t>& list, const uint32_t* ticks /* [4 list.size()] */, uint32_t width, uint32_t rows, double wave_slots,
                              double threshold, double down, std::vector<uint32_t>* out, uint32_t max_s = 4u, double* throughput_ticks = nullptr) {
    struct Tile {
        uint32_t x0, y0, s;
        uint64_t longest = 0;
        double predicted = 0.0;
    };
    const uint32_t tw = (width + 15u) / 16u;
    std::vector<Tile> tiles;
    std::vector<int32_t> index((size_t)tw * ((rows + 15u) / 16u), -1);
    uint64_t total = 0u;  // (an integer: a chain of double additions, four per block, was most of this loop's time)
    for (size_t b = 0; b < list.size(); b++) {
        const uint32_t t = list[b], x0 = tile_x0(t), y0 = tile_y0(t);
        if (x0 >= width || y0 >= rows) continue;  // (a padded grid's blocks outside the image)
        int32_t& slot = index[(size_t)(y0 / 16u) * tw + x0 / 16u];
        if (slot < 0) {
            slot = (int32_t)tiles.size();
            Tile n;
            n.x0 = x0 & ~15u, n.y0 = y0 & ~15u, n.s = tile_s(t);
            tiles.push_back(n);
        }
        Tile& tile = tiles[(size_t)slot];
        const uint32_t* d = ticks + 4u * b;  // (the block's four waves)
        tile.longest = std::max<uint64_t>(tile.longest, std::max(std::max(d[0], d[1]), std::max(d[2], d[3])));
        total += (uint64_t)d[0] + d[1] + d[2] + d[3];
    }
    const double throughput = (double)total / std::max(1.0, wave_slots);  // ticks the frame takes if the work were spread evenly
    if (throughput_ticks) *throughput_ticks = throughput;
    for (Tile& t : tiles) {
        t.predicted = (double)t.longest;
        while (t.s < max_s && t.predicted > threshold * throughput) t.s++, t.predicted *= 0.7;  // (up to sixteen lanes per pixel)
        while (t.s > 0u && t.predicted / 0.7 < down * throughput) t.s--, t.predicted /= 0.7;
    }
    // the tiles by predicted longest wave, longest first, equal ones in list order: a radix sort of the (non-negative) doubles' bit
    // patterns, 16 bits a pass, passes whose digit is the same everywhere skipped -- a comparison sort of 16 384 tiles cost the host
    // 2 ms in front of the frame that waits for the list, this a tenth of that
    std::vector<uint32_t> order(tiles.size()), other(tiles.size());
    {
        std::vector<uint64_t> key(tiles.size());
        uint64_t all_or = 0u, all_and = ~(uint64_t)0u;
        for (uint32_t i = 0; i < order.size(); i++) {
            const double pr = tiles[i].predicted > 0.0 ? tiles[i].predicted : 0.0;
            uint64_t k;
            std::memcpy(&k, &pr, sizeof(k));
            key[i] = ~k;  // (ascending in ~k = descending in the prediction)
            all_or |= key[i], all_and &= key[i];
            order[i] = i;
        }
        std::vector<uint32_t> count(65537u);
        for (uint32_t shift = 0u; shift < 64u; shift += 16u) {
            if ((((all_or ^ all_and) >> shift) & 0xffffu) == 0u) continue;
            std::fill(count.begin(), count.end(), 0u);
            for (uint32_t i : order) count[((key[i] >> shift) & 0xffffu) + 1u]++;
            for (uint32_t d = 0u; d < 65536u; d++) count[d + 1u] += count[d];
            for (uint32_t i : order) other[count[(key[i] >> shift) & 0xffffu]++] = i;
            order.swap(other);
        }
    }
    out->clear();
    for (uint32_t i : order) {
        const Tile& t = tiles[i];
        const uint32_t hbw = 16u >> (t.s >> 1), hbh = 16u >> ((t.s + 1u) >> 1);
        for (uint32_t dy = 0; dy < 16u && t.y0 + dy < rows; dy += hbh)
            for (uint32_t dx = 0; dx < 16u && t.x0 + dx < width; dx += hbw) out->push_back(tile_word(t.s, t.x0 + dx, t.y0 + dy));
    }
}

// How long a frame takes whose blocks start in the given order: every block goes to the workgroup slot that is free first and
// keeps it for as long as its longest wave ran (what the dispatcher does, with costs in whatever unit `cost` is in).
static double simulate_dispatch(const std::vector<uint32_t>& cost, size_t slots) {
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    for (size_t i = 0; i < std::max<size_t>(1, slots); i++) free_at.push(0.0);
    double end = 0.0;
    for (uint32_t c : cost) {
        const double t = free_at.top() + (double)c;
        free_at.pop();
        free_at.push(t);
        end = std::max(end, t);
    }
    return end;
}

// a device buffer of at least `bytes` (nothing may be in flight that reads the old one)
static hipError_t grow(uint32_t** p, size_t* cap, size_t bytes) {
    if (*p != nullptr && *cap >= bytes) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr, *cap = 0;
    const size_t want = std::max<size_t>(256, bytes + bytes / 2);
    hipError_t e = hipMalloc(p, want);
    if (e == hipSuccess) *cap = want;
    return e;
}
static hipError_t feedback_staging(rtc_ctx* c, size_t bytes, void** p) {  // (the caller has synchronised the device: nothing is using the old one)
    if (c->h_feedback == nullptr || c->h_feedback_cap < bytes) {
        if (c->h_feedback) (void)hipHostFree(c->h_feedback);
        c->h_feedback = nullptr, c->h_feedback_cap = 0;
        const size_t want = std::max<size_t>(1u << 16, bytes + bytes / 2);
        hipError_t e = hipHostMalloc(&c->h_feedback, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        c->h_feedback_cap = want;
    }
    *p = c->h_feedback;
    return hipSuccess;
}
static int compute_units(rtc_ctx* c) {  // of the context's device (asked once: the query takes a fraction of a millisecond)
    if (c->n_cus == 0) {
        hipDeviceProp_t prop;
        c->n_cus = hipGetDeviceProperties(&prop, c->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return c->n_cus;
}

// A block list's first launch has left its waves' times (BlockList::TIMED): the list of this and every later frame is made from
// them (refine_block_list); feedback_passes times over.
static rtc_status recut_block_list(rtc_ctx* c, BlockList& bl, uint32_t rows) {
    const Policy& P = c->policy;
    HIP_TRY(hipDeviceSynchronize());  // (once per scene, partition and pass; the launch may be on any stream)
    const auto host_t0 = std::chrono::steady_clock::now();  // (after the wait for the frame, which the caller's next step would have had anyway)
    std::vector<uint32_t> refined;
    double throughput_ticks = 0.0;
    void* staging = nullptr;
    HIP_TRY(feedback_staging(c, 4u * bl.n * sizeof(uint32_t), &staging));
    const uint32_t* ticks = (const uint32_t*)staging;
    HIP_TRY(hipMemcpy(staging, bl.d_ticks, 4u * bl.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    refine_block_list(bl.host, ticks, c->hdr.width, rows, 0.85 * 4.0 * compute_units(c) * c->tree_waves, 0.01 * P.feedback_pct, 0.01 * P.feedback_down_pct,
                      &refined, P.feedback_max_s, &throughput_ticks);
    if (P.jit_print) {
        size_t by_s[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};  // blocks by lanes per pixel (log2), before and after
        for (uint32_t t : bl.host.empty() ? refined : bl.host) by_s[0][tile_s(t)]++;
        for (uint32_t t : refined) by_s[1][tile_s(t)]++;
        std::fprintf(stderr, "librtc_amd: block list made from the frame before's wave times: %zu blocks; pixels by lanes per pixel 1/2/4/8/16: "
                             "%zu/%zu/%zu/%zu/%zu -> %zu/%zu/%zu/%zu/%zu k\n", bl.n, by_s[0][0] * 256 / 1000, by_s[0][1] * 128 / 1000, by_s[0][2] * 64 / 1000,
                     by_s[0][3] * 32 / 1000, by_s[0][4] * 16 / 1000, by_s[1][0] * 256 / 1000, by_s[1][1] * 128 / 1000, by_s[1][2] * 64 / 1000, by_s[1][3] * 32 / 1000,
                     by_s[1][4] * 16 / 1000);
    }
    HIP_TRY(grow(&bl.d, &bl.d_cap, refined.size() * sizeof(uint32_t)));  // (nothing is in flight: the synchronisation above)
    HIP_TRY(feedback_staging(c, refined.size() * sizeof(uint32_t), &staging));  // (the times have been used)
    std::memcpy(staging, refined.data(), refined.size() * sizeof(uint32_t));
    HIP_TRY(hipMemcpy(bl.d, staging, refined.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    bl.n = refined.size();
    bl.passes++;
    bl.host = refined;  // (kept: a later scene of this size starts from it, restart_block_lists)
    bl.state = bl.passes < P.feedback_passes ? BlockList::FRESH /* time this list's first launch as well */ : BlockList::REFINED;
    // How often scenes that change (restart_block_lists) may have the list re-cut: what this re-cut cost the host may be a
    // sixteenth of the frames between -- their time taken from below: the waves' times over the device's wave slots (the timer runs
    // at 100 MHz).
    const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    const double frame_ms = std::max(1e-3, throughput_ticks * 1e-5);
    bl.recut_host_ms = std::max(0.05, host_ms);
    if (P.jit_print) std::fprintf(stderr, "librtc_amd: the re-cut took the host %.3f ms (a frame takes %.3f ms or more)\n", host_ms, frame_ms);
    return RTC_OK;
}

// A regular grid's first frame has left what its waves cost (BlockList::TIMED): their times, or -- kernels that do not time
// their waves -- their work counts.  Is the order worth a list?  Both orders go through a model of the dispatcher with the blocks'
// costs: the list is taken when it ends the frame at least 3 % earlier (reflect_refract 9 %, first_textures 20 %, hexagons 8 %:
// yes; the frames of short, even waves -- C3 1 %, first_scene 0 % -- no: there a list's scalar load and the lost neighbourhood of
// the blocks in flight cost more than the order gives).  gx x gy: the launched (padded) grid.
static rtc_status order_grid(rtc_ctx* c, BlockList& bl, uint32_t gx, uint32_t gy, uint32_t rows) {
    const Policy& P = c->policy;
    HIP_TRY(hipDeviceSynchronize());  // (once per scene and partition)
    const size_t nt = bl.n;  // the blocks of the timed launch: the (padded) grid's
    std::vector<uint32_t> launched(nt), ordered;
    void* staging = nullptr;
    HIP_TRY(feedback_staging(c, 4u * nt * sizeof(uint4), &staging));
    uint32_t* ticks = (uint32_t*)staging;
    if (bl.counts) {
        // what a wave cost, from what it counted: rays that met objects, and shade points (each a light-cone cull, a Phong
        // evaluation, a push or pop of the recursion) at sixteen rays apiece
        const uint4* counts = (const uint4*)staging;
        HIP_TRY(hipMemcpy(staging, bl.d_ticks, 4u * nt * sizeof(uint4), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < 4u * nt; i++) {  // (in place: ticks[i] overwrites a word of counts[i / 4], which has been read)
            const uint4 n = counts[i];
            ticks[i] = (n.x - std::min(n.x, n.z)) + 16u * n.y + n.z / 8u;
        }
    } else {
        HIP_TRY(hipMemcpy(staging, bl.d_ticks, 4u * nt * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    for (uint32_t by = 0; by < gy; by++)
        for (uint32_t bx = 0; bx < gx; bx++) {  // the block workgroup (bx, by) rendered: the kernel's permutation
            uint32_t x = bx, y = by;
            if (bl.swizzled) {
                const uint32_t j = (by & 3u) * gx + bx, r = j & 7u;
                y = (by & ~3u) + (r >> 1);
                x = 2u * (j >> 3) + (r & 1u);
            }
            launched[(size_t)by * gx + bx] = tile_word(0u, 16u * x, 16u * y);
        }
    const double wave_slots = 0.85 * 4.0 * compute_units(c) * 6.0;
    bl.state = BlockList::REFINED;
    bl.listed = false;
    {
        std::vector<uint32_t> block_cost(nt), sorted_cost;
        for (size_t b = 0; b < nt; b++) block_cost[b] = std::max(std::max(ticks[4 * b], ticks[4 * b + 1]), std::max(ticks[4 * b + 2], ticks[4 * b + 3]));
        sorted_cost = block_cost;
        std::sort(sorted_cost.begin(), sorted_cost.end(), std::greater<uint32_t>());
        const size_t wg_slots = (size_t)(wave_slots / 4.0);
        const double in_order = simulate_dispatch(block_cost, wg_slots), longest_first = simulate_dispatch(sorted_cost, wg_slots);
        if (P.jit_print) std::fprintf(stderr, "librtc_amd: grid of %zu blocks: modelled frame %.4g in image order, %.4g longest first\n", nt, in_order, longest_first);
        if (!(longest_first < 0.97 * in_order)) return RTC_OK;  // (the grid stays)
    }
    refine_block_list(launched, ticks, c->hdr.width, rows, wave_slots, INFINITY, 0.0, &ordered);
    if (ordered.empty() || ordered.size() > bl.n) return RTC_OK;
    HIP_TRY(grow(&bl.d, &bl.d_cap, ordered.size() * sizeof(uint32_t)));  // (nothing is in flight: the synchronisation above)
    std::memcpy(staging, ordered.data(), ordered.size() * sizeof(uint32_t));  // (the times have been used; ordered.size() <= nt)
    HIP_TRY(hipMemcpy(bl.d, staging, ordered.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    bl.n_listed = ordered.size();
    bl.listed = true;
    return RTC_OK;
}

// The policy wanted a scene-compiled kernel and hiprtc did not deliver one.  RTC_AMD_SPECIALIZE=1: an error.  Default
// policy: the ahead-of-time kernel renders the same image -- several times slower on area-light scenes -- so say so:
// rtc_ctx_jit_status(), rtc_stats.flags, one line on stderr per process.
static rtc_status jit_failed(rtc_ctx* c, int policy, rtc_status jst) {
    c->spec_fn = nullptr;
    c->spec_shares = false;
    c->kernel_id = aot_kernel_id();
    c->jit_note = rtc_last_error();
    if (policy == 1) {
        c->has_scene = false;
        c->soa_host.clear();
        return jst;
    }
    static bool warned = false;
    if (!warned && !c->policy.quiet) {
        warned = true;
        std::fprintf(stderr, "librtc_amd: scene specialisation unavailable, rendering with the slower ahead-of-time kernel %s: %.300s\n",
                     c->kernel_name.c_str(), c->jit_note.c_str());
    }
    return RTC_OK;
}

rtc_status rtc_ctx_set_scene(rtc_ctx* c, const rtc_scene* scene, const rtc_camera* camera) {
    if (!c) return fail(RTC_ERR_INVALID_ARG, "ctx is NULL");
    SceneHdr hdr;
    std::vector<float4> soa;
    std::vector<float> texels;
    std::vector<float> heavy_boxes;
    SceneRegion region;
    const Policy& P = c->policy;
    rtc_status st = flatten(P, scene, camera, &hdr, &soa, &texels, &heavy_boxes, &region);
    if (st != RTC_OK) return st;
    HIP_TRY(hipSetDevice(c->device));
    if (c->has_scene && std::memcmp(&hdr, &c->hdr, sizeof(hdr)) == 0 && soa.size() == c->soa_host.size() &&
        texels.size() == c->texels_host.size() && std::memcmp(soa.data(), c->soa_host.data(), soa.size() * sizeof(float4)) == 0 &&
        (texels.empty() || std::memcmp(texels.data(), c->texels_host.data(), texels.size() * sizeof(float)) == 0))
    {   // the very scene that is resident (records, camera, light; the switches are the context's for life): nothing to replace --
        // unless its kernel was left uncompiled the first time (jit_get, lazy): frames repeat, so now it pays
        if (c->jit_deferred) {
            c->jit_deferred = false;
            HIP_TRY(hipDeviceSynchronize());  // (nothing of this context may be in flight while its kernel and lists change)
            hipFunction_t fn = nullptr;
            std::string id;
            const rtc_status jst = jit_get(P, c->device, c->spec_defs, &fn, &id);
            if (jst != RTC_OK) {
                const rtc_status fst = jit_failed(c, P.specialise, jst);
                if (fst != RTC_OK) return fst;
            } else {
                c->spec_fn = fn, c->kernel_id = id, c->kernel_name = c->spec_name;
                drop_block_lists(c);  // (lists of the ahead-of-time launches: the scene's kernel takes other ones)
                drop_scene_tile_lists(c);
                c->deep_fn.clear();
            }
        }
        return RTC_OK;
    }
    // Renders are asynchronous on caller streams (torch's are non-blocking: the null-stream copies below do not order
    // against them), and a render still in flight reads the records and the counters this call replaces.  Wait for
    // everything the context has launched before touching them.  (rtc.h: one stream at a time per context.)
    HIP_TRY(hipDeviceSynchronize());
    const bool same_frame = c->has_scene && c->hdr.width == hdr.width && c->hdr.height == hdr.height;  // (block lists: below)
    // until the new scene is fully resident the context has none: a failed allocation below must not leave a stale
    // capacity beside a null pointer, nor a render path that believes the old scene is still there
    // (only the camera has moved -- an animation's usual frame: the records and texels that are resident stay)
    const bool same_records = c->has_scene && soa.size() == c->soa_host.size() && texels.size() == c->texels_host.size() &&
                              std::memcmp(soa.data(), c->soa_host.data(), soa.size() * sizeof(float4)) == 0 &&
                              (texels.empty() || std::memcmp(texels.data(), c->texels_host.data(), texels.size() * sizeof(float)) == 0);
    c->has_scene = false;
    c->spec_fn = nullptr;
    c->jit_note.clear();
    if (!same_records) {
        c->soa_host.clear();
        c->texels_host.clear();
        if (texels.size() > c->texel_cap) {
            if (c->d_texels) (void)hipFree(c->d_texels);
            c->d_texels = nullptr;
            c->texel_cap = 0;
            HIP_TRY(hipMalloc(&c->d_texels, texels.size() * sizeof(float)));
            c->texel_cap = texels.size();
        }
        if (!texels.empty()) HIP_TRY(hipMemcpy(c->d_texels, texels.data(), texels.size() * sizeof(float), hipMemcpyHostToDevice));
        if (soa.size() > c->soa_cap) {
            if (c->d_soa) (void)hipFree(c->d_soa);
            c->d_soa = nullptr;
            c->soa_cap = 0;
            HIP_TRY(hipMalloc(&c->d_soa, soa.size() * sizeof(float4)));
            c->soa_cap = soa.size();
        }
        HIP_TRY(hipMemcpy(c->d_soa, soa.data(), soa.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    c->hdr = hdr;
    c->n_objects = hdr.n_objects;
    c->simple = !hdr.has_patterns;
    for (uint32_t i = 0; i < hdr.n_objects; i++) {
        uint32_t bits;
        std::memcpy(&bits, &soa[i].w, 4);  // geo[i].w
        const uint32_t kind = bits & SHAPE_KIND_MASK;
        if (!(bits & SHAPE_DIAG) || kind == RTC_CYLINDER || kind == RTC_CONE || kind == RTC_TRIANGLE) c->simple = false;
    }
    c->has_scene = true;
    if (!same_records) {
        c->soa_host = soa;
        c->texels_host = texels;
    }
    if (same_frame && P.block_feedback) restart_block_lists(c);
    else drop_block_lists(c);  // (nothing is in flight any more: the synchronisation above)
    c->deep_fn.clear();
    c->spec_defs.clear();
    project_heavy_boxes(P, heavy_boxes, camera, &c->heavy_tiles, &c->heavy_w, &c->heavy_h);

    // which kernel will render this scene
    c->spec_fn = nullptr;
    // sample-parallel rendering (render_body): compiled in when this frame is small enough to want it
    c->spec_shares = choose_share_log2(hdr, hdr.height, P) != 0u || has_leaf_runs(hdr);  // (kernels of mesh scenes always: their block lists)
    // several blocks per workgroup (render_body) where most workgroups see nothing but the sky: the scene's box projects to
    // less than a quarter of the image
    c->scene_box_coverage = 1.0f;
    c->scene_rect[0] = c->scene_rect[1] = c->scene_rect[2] = c->scene_rect[3] = 0u;
    c->scene_tile_mask.clear();
    drop_scene_tile_lists(c);  // (nothing is in flight: the synchronisation above)
    if (hdr.has_scene_box && camera && P.block_list) {
        std::vector<uint8_t> covered;
        uint32_t tw = 0, th = 0;
        std::vector<float> box(hdr.scene_box, hdr.scene_box + 6);
        box.push_back(1.0f);
        project_heavy_boxes(P, box, camera, &covered, &tw, &th);
        size_t n_cov = 0;
        for (uint8_t b : covered) n_cov += b;
        if (!covered.empty()) c->scene_box_coverage = (float)n_cov / (float)covered.size();
    }
    // The scene rectangle (rtc_ctx_render): the 16 x 16 tiles in which a primary ray can see anything at all -- the padded
    // box of the bounded top-level entries, projected, and for every top-level plane the side of its horizon on which rays
    // point towards it.
    c->scene_rect_coverage = 1.0f;
    if (region.known && camera && P.block_list) {
        std::vector<uint8_t> covered;
        uint32_t tw = (camera->width + 15u) / 16u, th = (camera->height + 15u) / 16u;
        if (region.has_box) {
            std::vector<float> box(region.box, region.box + 6);
            box.push_back(1.0f);
            project_heavy_boxes(P, box, camera, &covered, &tw, &th);
        }
        if (covered.empty()) covered.assign((size_t)tw * th, 0);
        for (const auto& pl : region.planes) mark_plane_side(pl, camera, &covered, tw, th);
        size_t n_cov = 0;
        uint32_t x0 = tw, x1 = 0, y0 = th, y1 = 0;
        for (uint32_t ty = 0; ty < th; ty++)
            for (uint32_t tx = 0; tx < tw; tx++)
                if (covered[(size_t)ty * tw + tx])
                    n_cov++, x0 = std::min(x0, tx), x1 = std::max(x1, tx + 1u), y0 = std::min(y0, ty), y1 = std::max(y1, ty + 1u);
        if (x0 < x1 && y0 < y1) {
            c->scene_rect[0] = x0, c->scene_rect[1] = x1, c->scene_rect[2] = y0, c->scene_rect[3] = y1;
            c->scene_rect_coverage = (float)((double)(x1 - x0) * (y1 - y0) / ((double)tw * th));
        }
        if (P.jit_print)
            std::fprintf(stderr, "librtc_amd: scene rectangle tiles [%u, %u) x [%u, %u) of %u x %u: %.3f of the frame\n", x0, x1, y0, y1, tw, th,
                         c->scene_rect_coverage);
        // ... and entry by entry (ERROR_BUDGET.md B8 holds for each padded box as it does for their union): where the entries
        // together cover under a third of the frame and under two thirds of their bounding rectangle, frames are drawn tile by tile
        c->scene_tile_mask.clear();
        if (P.scene_tiles && region.planes.empty() && !region.entry_boxes.empty() && x0 < x1 && y0 < y1) {
            std::vector<uint8_t> each;
            uint32_t ew = 0, eh = 0;
            project_heavy_boxes(P, region.entry_boxes, camera, &each, &ew, &eh);
            size_t n_each = 0;
            for (uint8_t b : each) n_each += b ? 1u : 0u;
            if (ew == tw && eh == th && n_each > 0 && 3u * n_each < (size_t)tw * th && 3u * n_each < 2u * (size_t)(x1 - x0) * (y1 - y0)) {
                c->scene_tile_mask = each;
                c->scene_tiles_w = tw, c->scene_tiles_h = th;
            }
            if (P.jit_print) std::fprintf(stderr, "librtc_amd: scene tiles: %zu of %u x %u%s\n", n_each, tw, th, c->scene_tile_mask.empty() ? " (not used)" : "");
        }
    }
    c->spec_blocks_y = c->scene_box_coverage < 0.25f || P.blocks_y != 0;
    const std::string share_def = std::string("-DRTC_SPEC_SHARE=") + (c->spec_shares ? "1" : "0");
    // Scene rectangle launches need a few more argument loads and operations in front of every wave, which cost frames of
    // short waves 6 - 10 % (first_plane, first_patterns; C4 0.610 -> 0.648 ms, more than the 3 % its sky rows are worth): only
    // where the rectangle is under half the frame (C5, single_sphere) is the scene's kernel compiled with them.
    c->spec_rect = c->scene_rect[0] < c->scene_rect[1] && c->scene_rect_coverage < P.scene_rect_threshold();
    const std::string blocks_def = std::string("-DRTC_SPEC_BLOCKS_Y=") + (c->spec_blocks_y ? "1" : "0");
    const std::string rect_def = std::string("-DRTC_SPEC_RECT=") + (c->spec_rect ? "1" : "0");
    // Material facts (rtc_kernel_core.h): does any material reflect / transmit at all (a scene without either carries no
    // recursion code), does any need powf for a highlight, and how many levels of the recursion stack the kernel keeps in
    // registers (FrameStack).  Register levels were built to take the 2.5 GB of frame traffic out of the glass-and-mirror
    // scene and do (scratch 448 -> 184 B per lane), but the frame gets only 2.5 % faster at 4 waves per SIMD and every
    // other scene slower (tools/ab_env.py): the traffic was not what the waves wait for.  Default 0;
    // RTC_AMD_REG_LEVELS=1..8 keeps the experiment reproducible.
    bool any_refl = false, any_refr = false, any_specular = false;
    for (uint32_t i = 0; i < hdr.n_objects; i++) {
        const rtc_material& m = scene->objects[i].material;
        any_refl = any_refl || !(m.reflective == 0.0f);
        any_refr = any_refr || !(m.transparency == 0.0f);
        any_specular = any_specular || !(m.specular == 0.0f && m.shininess >= 0.0f && m.shininess <= 1e6f);  // phong: needs powf
    }
    // Level-by-level rendering pays where a pixel's ray tree is what makes a frame long: tree worlds with the long leaf runs
    // of divided meshes whose materials both reflect and transmit (every such hit doubles the rays below it)
    c->wf_pays = hdr.n_trav != 0u && hdr.max_leaf_run >= 16u && any_refl && any_refr;
    c->wf_disabled = false;
    int reg_levels = P.reg_levels;
    if (!any_refl && !any_refr) reg_levels = 0;
    // which components of the area light's cell vectors are exact zeros (kernel: LIGHT_ZEROS / point_on_light); only when
    // the factors they would be multiplied with are finite -- hashed jitter is in (0, 1], a constant is the caller's
    uint32_t light_zeros = 0u;
    if (hdr.light_kind == RTC_LIGHT_RECT && (hdr.jitter_mode == RTC_JITTER_HASHED || std::isfinite(hdr.jitter_const)))
        for (int k = 0; k < 3; k++) light_zeros |= (hdr.uvec[k] == 0.0f ? 1u << k : 0u) | (hdr.vvec[k] == 0.0f ? 8u << k : 0u);
    std::vector<std::string> recursion_defs = {"-DRTC_SPEC_LIGHT_ZEROS=" + std::to_string(light_zeros),
                                               std::string("-DRTC_SPEC_ANY_REFL=") + (any_refl ? "1" : "0"),
                                               std::string("-DRTC_SPEC_ANY_REFR=") + (any_refr ? "1" : "0"),
                                               "-DRTC_SPEC_REG_LEVELS=" + std::to_string(reg_levels),
                                               std::string("-DRTC_SPEC_ANY_SPECULAR=") + (any_specular ? "1" : "0")};
    // register levels need the registers: 13 dwords per level on top of the ~70 the kernel works in
    const char* reg_waves = reg_levels == 0 ? nullptr : reg_levels <= 3 ? "-DRTC_WAVES_PER_SIMD=4" : "-DRTC_WAVES_PER_SIMD=3";
    const uint32_t n = hdr.n_objects;
    char nm[96];
    snprintf(nm, sizeof(nm), "render_kernel<%d,%s>", n <= 4 ? 4 : n <= 8 ? 8 : 0, (n <= 8 && c->simple) ? "simple" : "general");
    c->kernel_name = nm;
    c->kernel_id = aot_kernel_id();
    // The options of this scene's kernel are written down whatever the policy says (deep_kernel may need them later);
    // `compile_now`: does the policy want the scene-compiled kernel for ordinary depths.
    std::vector<std::string> defs;
    std::string spec_name;
    bool compile_now = false;
    const int policy = P.specialise;
    const uint64_t pixels = (uint64_t)hdr.width * hdr.height;
    // do all objects share one kind / flags word?  (SHAPE_UNIFORM aside: only the unrolled kernels' fast shadow decision
    // reads it, and a cloud of spheres must not lose its like-objects kernel because some are squashed)
    auto uniform_bits = [&](uint32_t* first) {
        std::memcpy(first, &soa[0].w, 4);
        *first &= ~(uint32_t)SHAPE_UNIFORM;
        for (uint32_t i = 1; i < n; i++) {
            uint32_t bits;
            std::memcpy(&bits, &soa[i].w, 4);
            if ((bits & ~(uint32_t)SHAPE_UNIFORM) != *first) return false;
        }
        return n > 0;
    };
    if (hdr.n_trav) {  // a traversal stream (GroupShapes, or the library's own hierarchy): packet walk, compiled per scene like the flat kernels
        const std::string how = scene->n_groups ? "tree" : "tree,bvh";
        c->kernel_name = "render_kernel<" + how + ">";
        // (worlds with divided meshes are compiled whatever the frame's size: the ahead-of-time walk has neither the
        // triangle pre-culling specialisation nor the leaf-sharing lanes -- mesh 512 x 384: 7.9 ms)
        compile_now = policy == 1 || (policy == 2 && (pixels >= (1ull << 18) || hdr.max_leaf_run >= 16u));
        // the traversal kernel compiled for this scene's light kind / jitter mode / pattern use and, when every object
        // shares one kind / flags word (a triangle mesh, a grid of spheres), for that word as well
        uint32_t first = 0u;
        const bool uniform = uniform_bits(&first);
        char b[16];
        snprintf(b, sizeof(b), "0x%x", first);
        // Waves per SIMD, i.e. registers per lane (80 at six, 96 at five).  At six the walk's state does not fit and the hot loops spill:
        // hexagons 4096 x 2048 moves 940 MB of HBM-side traffic for its 101 MB canvas at six and 557 MB at five, in the same 0.446 ms;
        // here_be_dragons 1000 x 400 265 -> 151 MB and 0.565 -> 0.533 ms; mesh 1024^2 1.56 -> 1.46 ms, 512 x 384 1.48 -> 1.39; hexagons
        // 1000 x 500, C5, grouped_grid: even.  Only the large frames of divided meshes, whose time is wave slots rather than their
        // longest wave, want the sixth wave: mesh 2048^2 2.03 ms at six / 2.12 at five, here_be_dragons 4000 x 1600 1.71 / 1.83 (and
        // 2000 x 800 0.91 / 0.87 the other way).  profiles/r04_tree_waves.txt.
        c->tree_waves = P.tree_waves ? P.tree_waves : (!c->heavy_tiles.empty() && pixels >= 3000000ull) ? 6 : 5;
        defs = {std::string("-DRTC_SPEC_LIST=") + b,
                uniform ? std::string("-DRTC_SPEC_UNIFORM_BITS=") + b : std::string("-DRTC_SPEC_RUNTIME_BITS=1"),
                "-DRTC_SPEC_NOBJ=-1", "-DRTC_SPEC_SIMPLE=0",
                reg_waves ? std::string(reg_waves) : "-DRTC_WAVES_PER_SIMD=" + std::to_string(c->tree_waves),
                std::string("-DRTC_SPEC_TBOX=") + std::to_string(hdr.has_tbox),
                "-DRTC_SPEC_LIGHT_KIND=" + std::to_string(hdr.light_kind),
                "-DRTC_SPEC_JITTER=" + std::to_string(hdr.jitter_mode),
                std::string("-DRTC_SPEC_PATTERNS=") + (hdr.has_patterns ? "1" : "0")};
        defs.push_back(share_def);
        defs.push_back(blocks_def);
        defs.push_back(rect_def);
        defs.insert(defs.end(), recursion_defs.begin(), recursion_defs.end());
        spec_name = std::string("render_kernel_spec[") + how + (uniform ? std::string(";all ") + b : std::string()) + (hdr.has_patterns ? ";patterns" : "") + "]";
    } else if (n >= 1 && n <= 8) {
        compile_now = policy == 1 || (policy == 2 && pixels >= (1ull << 18));
        std::string list = "-DRTC_SPEC_LIST=";
        for (uint32_t i = 0; i < n; i++) {
            uint32_t bits;
            std::memcpy(&bits, &soa[i].w, 4);
            char b[16];
            snprintf(b, sizeof(b), "%s0x%x", i ? "," : "", bits);
            list += b;
        }
        defs.push_back(list);
        defs.push_back("-DRTC_SPEC_NOBJ=" + std::to_string(n));
        defs.push_back(std::string("-DRTC_SPEC_SIMPLE=") + (c->simple ? "1" : "0"));
        defs.push_back("-DRTC_SPEC_LIGHT_KIND=" + std::to_string(hdr.light_kind));
        defs.push_back("-DRTC_SPEC_JITTER=" + std::to_string(hdr.jitter_mode));
        defs.push_back(std::string("-DRTC_SPEC_PATTERNS=") + (hdr.has_patterns ? "1" : "0"));
        defs.push_back(std::string("-DRTC_SPEC_GATES=") + (hdr.n_gates ? "1" : "0"));
        // an area light's geometry as per-lane values in the sample loop (kernel: RTC_LIGHT_VGPRS): the loop's multiplies
        // leave the half-rate class an SGPR operand puts them in, and the compiler no longer re-loads a light vector from
        // the argument block INSIDE the loop when it runs out of scalar registers (C3: 0.988 ms with that load, 0.847
        // without; soft_shadows 2048^2 0.382 -> 0.344; neutral on first_textures and the 1000 x 400 demo frame)
        if (hdr.light_kind == RTC_LIGHT_RECT) defs.push_back("-DRTC_LIGHT_VGPRS");
        defs.push_back(share_def);
        defs.push_back(blocks_def);
        defs.push_back(rect_def);
        defs.insert(defs.end(), recursion_defs.begin(), recursion_defs.end());
        if (reg_waves) defs.push_back(reg_waves);
        // A point light has no sample loop to keep registers free for: cold state stays in VGPRs instead of being parked
        // in LDS around intensity_at (C4 0.86 -> 0.77 ms), and in kernels of a few scale+translate objects the hit
        // object's records are selected from the scalar loads the loops hold anyway instead of being gathered per lane
        // (C4 0.77 -> 0.68 ms; C2 13.9 -> 12.4 us).  Both cost registers that an area light's loop needs (C3 +6 %), and the
        // pattern / rotated-object kernels spill without the parking (reflect_refract 1.21 -> 1.49 ms): left as they were.
        if (hdr.light_kind == RTC_LIGHT_POINT && c->simple && !reg_waves) {
            defs.push_back("-DRTC_SPEC_STASH=0");
            // ... and the LDS this frees holds the reflection halves of the recursion frames (five levels, 30 KB per
            // workgroup): C4 0.71 -> 0.63 ms on one box, and its mirror floor no longer writes its recursion to memory
            if (any_refl || any_refr) defs.push_back("-DRTC_SPEC_LDS_FRAMES=5");
            if (n <= 2) defs.push_back("-DRTC_SPEC_SELECT=1");  // (4 - 6 objects: the selects cost more than the gathers, +13 ... +30 %)
            defs.push_back("-DRTC_WAVES_PER_SIMD=6");
        } else if (hdr.light_kind == RTC_LIGHT_POINT && !reg_waves) {
            // ... and the kernels of rotated objects, cylinders, cones and patterns under a point light (round 3): at seven
            // waves per SIMD they spill without the parking (reflect_refract 1.21 -> 1.49 ms, round 2) -- at FIVE (102 VGPRs) they
            // do not, and the LDS holds five levels of reflection halves instead: reflect_refract 4096 x 2048 1.204 -> 0.984 ms
            // (its counters: 64 % of the wave-cycles waiting on memory, 2.4 GB of scratch traffic for a 0.1 GB frame), skybox
            // 0.198 -> 0.188, first_plane / first_patterns -2 %, first_scene +1 % (profiles/r03_ab_point_light_policy.txt)
            defs.push_back("-DRTC_SPEC_STASH=0");
            if (any_refl || any_refr) defs.push_back("-DRTC_SPEC_LDS_FRAMES=5");
            // ... and FOUR where the world both reflects and transmits and has five objects or more (round 4): reflect_refract's
            // kernel spills 20 registers at five waves (96 VGPRs) and none at four (128) -- 0.874 -> 0.775 ms; skybox (two objects)
            // and the scenes without glass lose up to 12 % at four (profiles/r04_ab_point_light_waves.txt)
            defs.push_back((any_refl && any_refr && n >= 5u) ? "-DRTC_WAVES_PER_SIMD=4" : "-DRTC_WAVES_PER_SIMD=5");
        } else if (hdr.light_kind == RTC_LIGHT_RECT && !c->simple && !reg_waves) {
            // An area light's kernel takes seven waves per SIMD (jit_get's default: C3 0.97 / 0.95 / 1.02 ms at 6 / 7 / 8).  With rotated
            // objects, cylinders, patterns or gates the sample loop's state no longer fits 72 registers and spills: first_textures
            // 4096 x 2048 moves 337 MB HBM-side for its 101 MB canvas at seven and 185 MB at six (84 registers) in the same 0.70 ms; 1024 x
            // 512 and patterns_medley: even, first frames within 2 % either way (profiles/r04_ab_area_light_waves.txt).
            defs.push_back("-DRTC_WAVES_PER_SIMD=6");
        }
        spec_name = "render_kernel_spec[" + list.substr(16) + (c->simple ? ";simple" : "") + (hdr.has_patterns ? ";patterns" : "") + (hdr.n_gates ? ";gates" : "") + "]";
    } else if (n > 8) {
        // many objects: the any-count loop.  When they all share one kind / flags word (C5: 64 scale+translate spheres) that
        // word is a compile-time constant -- no per-object kind switch, two 16-byte records per object -- and the policy
        // compiles the scene's kernel; a mixed list is left to the ahead-of-time loop (and compiled with run-time words
        // only when the recursion is deeper than that kernel's stack)
        uint32_t first = 0u;
        const bool uniform = uniform_bits(&first);
        compile_now = uniform && (policy == 1 || (policy == 2 && pixels >= (1ull << 18)));
        char b[16];
        snprintf(b, sizeof(b), "0x%x", first);
        defs = {std::string("-DRTC_SPEC_LIST=") + b, uniform ? std::string("-DRTC_SPEC_UNIFORM_BITS=") + b : std::string("-DRTC_SPEC_RUNTIME_BITS=1"),
                "-DRTC_SPEC_NOBJ=0", "-DRTC_SPEC_SIMPLE=0",
                "-DRTC_SPEC_LIGHT_KIND=" + std::to_string(hdr.light_kind),
                "-DRTC_SPEC_JITTER=" + std::to_string(hdr.jitter_mode),
                std::string("-DRTC_SPEC_PATTERNS=") + (hdr.has_patterns ? "1" : "0")};
        defs.push_back(share_def);
        defs.push_back(blocks_def);
        defs.push_back(rect_def);
        if (reg_waves) defs.push_back(reg_waves);
        defs.insert(defs.end(), recursion_defs.begin(), recursion_defs.end());
        spec_name = std::string("render_kernel_spec[") + (uniform ? std::string("all ") + b : std::string("any")) + (hdr.has_patterns ? ";patterns" : "") + "]";
    }
    c->spec_defs = defs;
    c->spec_name = spec_name;
    c->jit_deferred = false;
    if (compile_now && !defs.empty()) {
        rtc_status jst = jit_get(P, c->device, defs, &c->spec_fn, &c->kernel_id, c->lazy_jit && policy == 2);
        if (jst != RTC_OK) {
            if ((jst = jit_failed(c, policy, jst)) != RTC_OK) return jst;
        } else if (c->spec_fn != nullptr) {
            c->kernel_name = spec_name;
        } else {
            c->kernel_id = aot_kernel_id();  // (lazy: this frame by the ahead-of-time kernel, whose name kernel_name already holds)
            c->jit_deferred = true;
        }
    }
    return RTC_OK;
}

}  // extern "C"

// The scene's kernel with a frame stack of at least `depth` levels (ctx_render_slot).  The options are the scene's own
// (rtc_ctx_set_scene wrote them down) plus -DRTC_SPEC_MAX_DEPTH; the recursion frames of such a kernel all live in per-lane
// scratch -- the LDS placement of the first levels (a tuning of the depth-5 glass-and-mirror frame) is not carried over.
// Always a scene-compiled kernel, whatever RTC_AMD_SPECIALIZE says: the ahead-of-time kernels stop at RTC_STACK_DEPTH_BASE.
static rtc_status deep_kernel(rtc_ctx* c, int32_t depth, hipFunction_t* out) {
    int cap = 2 * RTC_STACK_DEPTH_BASE;
    while (cap < depth) cap *= 2;
    if (cap > RTC_MAX_DEPTH) cap = RTC_MAX_DEPTH;
    auto it = c->deep_fn.find(cap);
    if (it != c->deep_fn.end()) {
        *out = it->second;
        return RTC_OK;
    }
    if (c->spec_defs.empty()) return fail(RTC_ERR_INVALID_ARG, "depth %d: no kernel options recorded for this scene", depth);
    std::vector<std::string> defs;
    for (const auto& d : c->spec_defs)
        if (d.rfind("-DRTC_SPEC_LDS_FRAMES=", 0) != 0) defs.push_back(d);
    defs.push_back("-DRTC_SPEC_MAX_DEPTH=" + std::to_string(cap));
    hipFunction_t fn = nullptr;
    std::string id;
    rtc_status st = jit_get(c->policy, c->device, defs, &fn, &id);
    if (st != RTC_OK) return st;  // (the message is hiprtc's)
    c->deep_fn[cap] = fn;
    *out = fn;
    return RTC_OK;
}

// The frame level by level (rtc_wavefront.h): one lane per ray, suspended shade_hits as nodes in HBM.  *used: false when the
// frame was not rendered this way (pools could not be allocated, or ran full: the caller renders it with the per-pixel kernel).
static rtc_status render_wavefront(rtc_ctx* c, int32_t depth, const Partition& q, uint32_t rows, void* d_out, bool out_u8, hipStream_t stream,
                                   uint32_t slot, bool* used) {
    *used = false;
    const size_t pixels = (size_t)rows * c->hdr.width;
    // pools: a level's two ray lists hold up to two rays per pixel each, the node pool six suspended hits per pixel (a glass
    // mesh filling a quarter of the frame at depth 5 needs about one and three); a frame that needs more is rendered again
    const size_t cap_rays = std::max<size_t>(1024, 2 * pixels), cap_nodes = std::max<size_t>(1024, 6 * pixels);
    if (cap_nodes > 0x7fffffffull) return RTC_OK;
    if (cap_rays > c->wf_cap_rays || cap_nodes > c->wf_cap_nodes || !c->d_wf_ctr) {
        HIP_TRY(hipDeviceSynchronize());  // (the pools may be in use by a frame in flight)
        for (WfRay*& r : c->d_wf_rays) {
            if (r) (void)hipFree(r);
            r = nullptr;
        }
        if (c->d_wf_nodes) (void)hipFree(c->d_wf_nodes);
        c->d_wf_nodes = nullptr;
        c->wf_cap_rays = c->wf_cap_nodes = 0;
        bool ok = true;
        for (WfRay*& r : c->d_wf_rays) ok = ok && hipMalloc((void**)&r, cap_rays * sizeof(WfRay)) == hipSuccess;
        ok = ok && hipMalloc((void**)&c->d_wf_nodes, cap_nodes * sizeof(WfNode)) == hipSuccess;
        if (ok && !c->d_wf_ctr) ok = hipMalloc((void**)&c->d_wf_ctr, WF_CTR_WORDS * sizeof(uint32_t)) == hipSuccess;
        if (!ok) {  // not enough memory for the pools: the per-pixel kernel needs none
            (void)hipGetLastError();
            for (WfRay*& r : c->d_wf_rays) {
                if (r) (void)hipFree(r);
                r = nullptr;
            }
            if (c->d_wf_nodes) (void)hipFree(c->d_wf_nodes);
            c->d_wf_nodes = nullptr;
            return RTC_OK;
        }
        c->wf_cap_rays = cap_rays, c->wf_cap_nodes = cap_nodes;
    }
    WfArgs a;
    a.hdr = c->hdr;
    a.soa = soa_view(c->d_soa, c->hdr, c->d_texels);
    a.out = out_u8 ? nullptr : (float*)d_out;
    a.out_u8 = out_u8 ? (uint8_t*)d_out : nullptr;
    a.rows = rows, a.band_rows = q.band_rows, a.n_parts = q.n_parts, a.part = q.part;
    a.depth = depth;
    a.nodes = c->d_wf_nodes;
    a.ctr = c->d_wf_ctr;
    a.cap_rays = (uint32_t)c->wf_cap_rays, a.cap_nodes = (uint32_t)c->wf_cap_nodes;
    // (levels after the first and the combining passes draw their work from counters: a launch that fills the chip once)
    const uint32_t ray_wgs = 256u * 6u, node_wgs = 256u * 4u;
    const dim3 primary_grid((c->hdr.width + 15u) / 16u, (rows + 15u) / 16u);
    const size_t n_counts = ((size_t)primary_grid.x * primary_grid.y + (size_t)depth * ray_wgs) * 4;  // one partial per wave of every tracing launch
    if (n_counts > c->block_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->d_block_counts) HIP_TRY(hipFree(c->d_block_counts));
        c->d_block_counts = nullptr;
        c->block_cap = 0;
        HIP_TRY(hipMalloc(&c->d_block_counts, n_counts * sizeof(uint4)));
        c->block_cap = n_counts;
    }
    a.wave_counts = c->d_block_counts;
    unsigned long long* total = c->d_total + 3 * (size_t)slot;
    HIP_TRY(hipMemsetAsync(c->d_wf_ctr, 0, WF_CTR_WORDS * sizeof(uint32_t), stream));
    HIP_TRY(hipMemsetAsync(total, 0, 3 * sizeof(unsigned long long), stream));
    if (c->events_used == c->events.size()) {
        if (c->events.size() >= 4096) {
            c->events_used = 0;
        } else {
            std::pair<hipEvent_t, hipEvent_t> e;
            HIP_TRY(hipEventCreate(&e.first));
            HIP_TRY(hipEventCreate(&e.second));
            c->events.push_back(e);
        }
    }
    auto& ev = c->events[c->events_used++];
    HIP_TRY(hipEventRecord(ev.first, stream));
    for (int32_t level = 0; level <= depth; level++) {
        a.level = (uint32_t)level;
        a.in_refl = c->d_wf_rays[2 * ((level + 1) & 1)], a.in_refr = c->d_wf_rays[2 * ((level + 1) & 1) + 1];
        a.out_refl = c->d_wf_rays[2 * (level & 1)], a.out_refr = c->d_wf_rays[2 * (level & 1) + 1];
        a.count_base = level == 0 ? 0u : (uint32_t)(((size_t)primary_grid.x * primary_grid.y + (size_t)(level - 1) * ray_wgs) * 4);
        if (level == 0)
            hipLaunchKernelGGL(wf_trace_kernel<true>, primary_grid, dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL(wf_trace_kernel<false>, dim3(ray_wgs), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(wf_snapshot_kernel, dim3(1), dim3(1), 0, stream, c->d_wf_ctr, (uint32_t)level, a.cap_nodes);
    }
    for (int32_t level = depth; level >= 0; level--) {
        a.level = (uint32_t)level;
        hipLaunchKernelGGL(wf_combine_kernel, dim3(node_wgs), dim3(256), 0, stream, a);
    }
    HIP_TRY(hipEventRecord(ev.second, stream));
    hipLaunchKernelGGL(sum_counts_kernel, dim3((uint32_t)((n_counts + SUM_COUNTS_SLICE - 1) / SUM_COUNTS_SLICE)), dim3(1024), 0, stream, c->d_block_counts,
                       (uint32_t)n_counts, total, 0ull);
    HIP_TRY(hipGetLastError());
    // did everything fit?  (One small read-back: the call returns when the frame is done -- these frames take milliseconds.)
    uint32_t overflow = 0u;
    HIP_TRY(hipMemcpyAsync(&overflow, c->d_wf_ctr + WF_CTR_OVERFLOW, sizeof(overflow), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (overflow) {
        c->wf_disabled = true;  // this scene needs more than the pools hold: per-pixel kernels from now on
        c->events_used--;
        return RTC_OK;
    }
    c->rendered = true;
    c->wf_last = true;
    *used = true;
    return RTC_OK;
}

// rtc_ctx_render with a counter slot of the caller's choosing (rtc_internal.h)
rtc_status rtc::ctx_render_slot(rtc_ctx* c, int32_t depth, const rtc_partition* part, void* d_out_rgb, void* stream_, uint32_t slot,
                                ProgressPlan* plan, bool out_u8) {
    if (plan) plan->n_chunks = 0u, plan->chunk_rows = 0u;
    if (!c) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_render: null argument");
    if (slot >= CTX_TOTAL_SLOTS) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_render: counter slot %u", slot);
    if (!c->has_scene || c->hdr.width == 0) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_render: no scene/camera set");
    // a partition that owns no band (height < band_rows * n_parts) has nothing to write and may pass a null buffer
    if (!d_out_rgb && partition_rows(c->hdr.height, part) != 0) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_render: null output buffer");
    if (depth < 0 || depth > RTC_MAX_DEPTH)
        return fail(RTC_ERR_INVALID_ARG, "depth %d outside [0, %d]", depth, RTC_MAX_DEPTH);
    Partition q = resolve(part);
    if (q.part >= q.n_parts) return fail(RTC_ERR_INVALID_ARG, "partition %u of %u", q.part, q.n_parts);
    const uint32_t rows = partition_rows(c->hdr.height, part);
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    const Policy& P = c->policy;
    // camera.rs:76 takes any depth; the kernels' frame stack (one suspended shade_hit per level, world.rs:62-86) holds
    // RTC_STACK_DEPTH_BASE levels.  Deeper than that, the scene's kernel is compiled once more with a longer stack -- 16, 32,
    // ... RTC_MAX_DEPTH levels of per-lane scratch -- on first use, and kept with the context.
    hipFunction_t spec_fn = c->spec_fn;
    // (a world in which nothing reflects or transmits -- an empty world among them, which the reference renders black at any
    // depth -- never suspends a shade_hit: whatever the depth asked for, the base kernels trace it as they trace depth 8)
    bool recurses = false;
    for (uint32_t i = 0; i < c->hdr.n_objects && !recurses; i++) {
        const float4 mb = c->soa_host[5 * (size_t)padded_count(c->hdr.n_objects) + i], mc = c->soa_host[6 * (size_t)padded_count(c->hdr.n_objects) + i];
        recurses = !(mb.w == 0.0f) || !(mc.x == 0.0f);  // reflective, transparency (NaN: recurses)
    }
    if (depth > RTC_STACK_DEPTH_BASE && !recurses) depth = RTC_STACK_DEPTH_BASE;
    if (depth > RTC_STACK_DEPTH_BASE && rows > 0u) {
        rtc_status dst = deep_kernel(c, depth, &spec_fn);
        if (dst != RTC_OK) return dst;
    }
    const uint32_t share_log2 = (spec_fn && c->spec_shares) ? choose_share_log2(c->hdr, rows, P, plan == nullptr) : 0u;  // only kernels compiled for it share lanes
    const uint32_t bw = 16u >> (share_log2 >> 1), bh = 16u >> ((share_log2 + 1u) >> 1);  // pixels per workgroup (2x2 wave tiles)
    dim3 grid((c->hdr.width + bw - 1) / bw, (rows + bh - 1) / bh), block(256);
    // Frames of very many very short waves: several blocks per workgroup (RTC_AMD_BLOCKS_Y=1..8 overrides)
    uint32_t blocks_y = 1u;
    if (!(spec_fn && c->spec_blocks_y)) {
        // (only kernels compiled for it loop over blocks)
    } else if (P.blocks_y != 0) {
        blocks_y = (uint32_t)P.blocks_y;
    } else if ((uint64_t)grid.x * grid.y >= (1u << 15)) {
        // most workgroups see nothing but the sky: C5 8192^2 0.51 -> 0.43 ms, single_sphere 4096^2 0.088 -> 0.061 ms.  (Where the
        // waves have work -- hexagons, grouped_grid, whose boxes fill the frame -- four blocks per workgroup cost 8 ... 17 %.)
        blocks_y = 4u;
    }
    grid.y = (grid.y + blocks_y - 1) / blocks_y;
    // Tree worlds with meshes: a block list instead of the regular grid -- the tiles a mesh projects to first, eight
    // lanes per pixel there and one elsewhere (build_block_list).  Not when RTC_AMD_SHARE_LOG2 pins one value for all.
    const uint32_t* d_tiles = nullptr;
    uint32_t* d_ticks = nullptr;
    BlockList* timed_list = nullptr;  // a list whose own events bracket this launch
    void* copy_counts_to = nullptr;
    const bool mesh_list = spec_fn && c->spec_shares && !c->heavy_tiles.empty() && c->hdr.light_kind == RTC_LIGHT_POINT;
    // ... and frames that share an area light's cells between a pixel's lanes (small frames: choose_share_log2), when there is a
    // frame before to go by: the list starts with the frame's one lane count everywhere, and the feedback gives the tiles in the
    // penumbra more lanes, the lit and the empty ones fewer (refine_block_list).  Not for rtc_render_ex, whose rows leave in order.
    const bool area_list = spec_fn && c->spec_shares && c->hdr.light_kind == RTC_LIGHT_RECT && share_log2 != 0u && P.block_feedback && plan == nullptr;
    if ((mesh_list || area_list) && P.share_log2 < 0 && c->hdr.width <= 65532u && rows <= 131068u && rows > 0u) {
        // one list per partition, built on first use and kept until the scene changes: rtc_render_ex renders a frame as
        // several partitions of one context, frame after frame (a single cached list meant a device synchronisation, a
        // rebuild and a blocking copy per chunk launch)
        const std::array<uint32_t, 5> key = {q.band_rows, q.n_parts, q.part, share_log2, P.block_feedback ? (uint32_t)depth : 0u};
        auto it = c->block_lists.find(key);
        if (it == c->block_lists.end()) {
            if (c->block_lists.size() >= 256u) {  // a caller cycling through partitions without end: start over (nothing may be in flight)
                HIP_TRY(hipDeviceSynchronize());
                drop_block_lists(c);
            }
            const rtc_ctx_tiles T = {c->heavy_tiles.data(), c->heavy_w, c->heavy_h};
            std::vector<uint32_t> host;
            if (mesh_list) {
                build_block_list(P, T, c->hdr.width, share_log2, rows, q, &host);
            } else {  // every tile with the frame's lane count, in image order
                const uint32_t hbw = 16u >> (share_log2 >> 1), hbh = 16u >> ((share_log2 + 1u) >> 1);
                for (uint32_t y0 = 0; y0 < rows; y0 += 16u)
                    for (uint32_t x0 = 0; x0 < c->hdr.width; x0 += 16u)
                        for (uint32_t dy = 0; dy < 16u && y0 + dy < rows; dy += hbh)
                            for (uint32_t dx = 0; dx < 16u && x0 + dx < c->hdr.width; dx += hbw)
                                host.push_back(tile_word(share_log2, x0 + dx, y0 + dy));
            }
            BlockList bl;
            bl.n = host.size();
            if (P.block_feedback) bl.host = host;
            HIP_TRY(grow(&bl.d, &bl.d_cap, host.size() * sizeof(uint32_t)));
            // (a new buffer: no launch in flight can be reading it; the copy is complete when the call returns)
            hipError_t ce = hipMemcpy(bl.d, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (ce != hipSuccess) {
                (void)hipFree(bl.d);
                return fail(RTC_ERR_DEVICE, "block list upload failed: %s", hipGetErrorString(ce));
            }
            it = c->block_lists.emplace(key, bl).first;
        }
        BlockList& bl = it->second;
        if (P.block_feedback && bl.state == BlockList::TIMED) {
            const rtc_status st = recut_block_list(c, bl, rows);
            if (st != RTC_OK) return st;
        }
        // (a refined list under scenes that change: every frame is timed -- restart_block_lists decides from them when to re-cut)
        if (P.block_feedback && (bl.state == BlockList::FRESH || (bl.state == BlockList::REFINED && bl.animated)) && bl.n != 0) {
            HIP_TRY(grow(&bl.d_ticks, &bl.ticks_cap, 4u * bl.n * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(bl.d_ticks, 0, 4u * bl.n * sizeof(uint32_t), stream));
            d_ticks = bl.d_ticks;
            if (bl.state == BlockList::FRESH) {
                bl.state = BlockList::TIMED;
            } else {
                if (bl.ev0 == nullptr) {
                    HIP_TRY(hipEventCreate(&bl.ev0));
                    HIP_TRY(hipEventCreate(&bl.ev1));
                }
                timed_list = &bl;
            }
        }
        d_tiles = it->second.d;
        grid = dim3((uint32_t)it->second.n, 1);
    }
    // traced pixels among this partition's rows: x < w-1, y < h-1
    uint64_t traced_rows = 0;
    {
        uint32_t n_bands = (c->hdr.height + q.band_rows - 1) / q.band_rows;
        for (uint32_t b = q.part; b < n_bands; b += q.n_parts) {
            uint32_t y0 = b * q.band_rows;
            uint32_t y1 = y0 + q.band_rows < c->hdr.height ? y0 + q.band_rows : c->hdr.height;
            uint32_t lim = c->hdr.height - 1;
            traced_rows += (y1 < lim ? y1 : lim) - (y0 < lim ? y0 : lim);
        }
    }
    c->last_rows = rows;
    c->last_pixels = traced_rows * (uint64_t)(c->hdr.width - 1);
    // Level by level instead of pixel by pixel (rtc_wavefront.h): only on request (RTC_AMD_WAVEFRONT=1).  Built in round 3 for
    // the frames whose time is their longest wave (glass meshes), bit-identical -- and measured slower everywhere: mesh 2048^2
    // 8.0 ms against 3.8, here_be_dragons 4000 x 1600 9.6 against 3.0 (profiles/r03_wavefront_ab.txt).  A level is a launch, a
    // launch ends with ITS longest wave -- one packet walk over a divided mesh is hundreds of microseconds -- and a frame of
    // depth 5 pays six of those tails where the per-pixel kernel pays one; its walks are also the generic ones (no lanes
    // splitting leaf runs, no per-scene compile).  What the finding asks for is a single persistent launch with a queue of rays
    // and continuation frames, not level-synchronous passes.  Kept as a verified alternative, not a default.
    if (c->hdr.n_trav != 0u && rows > 0u && depth >= 1 && depth <= RTC_STACK_DEPTH_BASE && (int)depth < (int)WF_MAX_LEVELS && !c->wf_disabled &&
        (size_t)rows * c->hdr.width <= (16u << 20) && P.wavefront == 1) {
        bool used = false;
        rtc_status wst = render_wavefront(c, depth, q, rows, d_out_rgb, out_u8, stream, slot, &used);
        if (wst != RTC_OK) return wst;
        if (used) return RTC_OK;
    }
    c->wf_last = false;
    // Scene rectangle: every primary ray outside the rectangle the scene's box projects to (project_heavy_boxes: exact
    // camera arithmetic in double, 8 pixels of padding, "everything" if the box reaches behind the camera) sees nothing --
    // black, one ray.  Where that rectangle is under half the frame (C5: a grid of spheres in the middle of 8192^2) the
    // kernel is launched over the rectangle's blocks only, preceded by workgroups that zero-fill the rest at memory speed
    // while the others render; the rays of the pixels outside are added to the count (sum_counts_kernel).
    // RTC_AMD_SCENE_RECT=0: the whole grid, as before.
    uint32_t block_x0 = 0u, block_y0 = 0u;
    unsigned long long extra_rays = 0ull;
    uint32_t fill_wg_rows = 0u, fill_rows = 0u, fill_period = 1u, fill_rect[4] = {0u, 0u, 0u, 0u};
    bool rect_launch = false;
    // Scene tiles (rtc_ctx::scene_tile_mask): the canvas is zero-filled at memory speed (805 MB of an 8192^2 frame: 0.12 ms) and
    // only the tiles some entry of the world projects to get a workgroup -- C5: 18 k of 262 k, where the bounding rectangle of
    // them all has 60 k, and a frame of such short waves costs what starting them costs (0.78 waves per ns).  The pixels of
    // the other tiles are one ray each that sees nothing (sum_counts_kernel's extra_rays).
    bool tile_launch = false;
    const uint2* fill_jobs = nullptr;
    size_t n_fill_jobs = 0;
    if (d_tiles == nullptr && share_log2 == 0u && rows > 0u && !c->scene_tile_mask.empty() && (q.band_rows & 15u) == 0u &&
        c->hdr.width <= 65532u && rows <= 131068u) {
        const std::array<uint32_t, 3> key = {q.band_rows, q.n_parts, q.part};
        auto it = c->scene_tile_lists.find(key);
        if (it == c->scene_tile_lists.end()) {
            std::vector<uint32_t> list;
            std::vector<uint2> fill;
            rtc_ctx::SceneTileList tl;
            const uint32_t n_bands = (c->hdr.height + q.band_rows - 1) / q.band_rows;
            uint32_t cursor = 0u;
            for (uint32_t b = q.part; b < n_bands; b += q.n_parts) {
                const uint32_t y0 = b * q.band_rows, y1 = std::min(c->hdr.height, y0 + q.band_rows);
                for (uint32_t ty = y0 / 16u; ty * 16u < y1; ty++) {
                    const uint32_t yl = cursor + (ty * 16u - y0);
                    uint32_t run0 = 0u, run = 0u;  // the current run of unlisted tiles: [run0, run0 + run)
                    auto close_run = [&]() {
                        for (uint32_t k = 0; k < run; k += 64u) fill.push_back(make_uint2((run0 + k) | (std::min(64u, run - k) << 16), yl));
                        run = 0u;
                    };
                    for (uint32_t tx = 0; tx < c->scene_tiles_w; tx++) {
                        if (!c->scene_tile_mask[(size_t)ty * c->scene_tiles_w + tx]) {
                            if (run == 0u) run0 = tx;
                            run++;
                            continue;
                        }
                        close_run();
                        list.push_back(tile_word(0u, tx * 16u, yl));
                        const uint32_t px1 = std::min(c->hdr.width - 1u, tx * 16u + 16u), py1 = std::min(std::min(c->hdr.height - 1u, y1), ty * 16u + 16u);
                        if (px1 > tx * 16u && py1 > ty * 16u) tl.traced_pixels += (unsigned long long)(px1 - tx * 16u) * (py1 - ty * 16u);
                    }
                    close_run();
                }
                cursor += y1 - y0;
            }
            tl.n = list.size();
            tl.n_fill = fill.size();
            if (tl.n) {
                HIP_TRY(hipMalloc((void**)&tl.d, tl.n * sizeof(uint32_t)));
                HIP_TRY(hipMemcpyAsync(tl.d, list.data(), tl.n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                if (tl.n_fill) {
                    HIP_TRY(hipMalloc((void**)&tl.d_fill, tl.n_fill * sizeof(uint2)));
                    HIP_TRY(hipMemcpyAsync(tl.d_fill, fill.data(), tl.n_fill * sizeof(uint2), hipMemcpyHostToDevice, stream));
                }
                HIP_TRY(hipStreamSynchronize(stream));  // (the host vectors go out of scope)
            }
            it = c->scene_tile_lists.emplace(key, tl).first;
        }
        const rtc_ctx::SceneTileList& tl = it->second;
        if (tl.n) {
            if (!c->fill_stream) {
                HIP_TRY(hipStreamCreateWithFlags(&c->fill_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            fill_jobs = tl.d_fill, n_fill_jobs = tl.n_fill;
            d_tiles = tl.d;
            grid = dim3((uint32_t)tl.n, 1);
            blocks_y = 1u;
            extra_rays = c->last_pixels - tl.traced_pixels;
            tile_launch = true;
        }
    }
    if (!tile_launch && d_tiles == nullptr && share_log2 == 0u && rows > 0u && c->scene_rect[0] < c->scene_rect[1] && c->scene_rect_coverage < P.scene_rect_threshold() &&
        (spec_fn == nullptr || c->spec_rect)) {
        // local rows of this partition whose global row lies in the rectangle's rows, and the traced ones among them
        const uint32_t gy0 = c->scene_rect[2] * 16u, gy1 = std::min(c->hdr.height, c->scene_rect[3] * 16u);
        uint32_t yl0 = rows, yl1 = 0u, cursor = 0u;
        const uint32_t n_bands = (c->hdr.height + q.band_rows - 1) / q.band_rows;
        for (uint32_t b = q.part; b < n_bands; b += q.n_parts) {
            const uint32_t y0 = b * q.band_rows, y1 = std::min(c->hdr.height, y0 + q.band_rows);
            const uint32_t lo = std::max(y0, gy0), hi = std::min(y1, gy1);
            if (lo < hi) yl0 = std::min(yl0, cursor + (lo - y0)), yl1 = std::max(yl1, cursor + (hi - y0));
            cursor += y1 - y0;
        }
        rect_launch = true;
        if (yl0 < yl1) {
            block_x0 = c->scene_rect[0];
            block_y0 = yl0 / 16u;
            // every workgroup of this launch has work: one block each unless told otherwise (C5 0.395 / 0.407 / 0.427 / 0.46 ms
            // with 1 / 2 / 4 / 8 blocks per workgroup)
            const uint32_t block_rows = (yl1 - block_y0 * 16u + 15u) / 16u;
            if (P.blocks_y == 0) blocks_y = 1u;
            grid = dim3(c->scene_rect[1] - c->scene_rect[0], (block_rows + blocks_y - 1u) / blocks_y);
        } else {
            grid = dim3(1, 1);  // none of this partition's rows: one block of the rectangle's columns, for the launch's bookkeeping
            block_x0 = c->scene_rect[0];
            block_y0 = 0u;
            blocks_y = 1u;
        }
        // traced pixels (x < w - 1, y < h - 1) inside the launched blocks
        const uint32_t lx0 = block_x0 * 16u, lx1 = std::min(c->hdr.width - 1u, (block_x0 + grid.x) * 16u);
        const uint32_t ly0 = block_y0 * 16u, ly1 = std::min(rows, (block_y0 + grid.y * blocks_y) * 16u);
        uint64_t launched_rows = 0;
        cursor = 0u;
        for (uint32_t b = q.part; b < n_bands; b += q.n_parts) {
            const uint32_t y0 = b * q.band_rows, y1 = std::min(c->hdr.height, y0 + q.band_rows);
            // local rows [cursor, cursor + y1 - y0) of this band that are launched, and traced (global row < h - 1)
            const uint32_t a = std::max(cursor, ly0), e = std::min(cursor + (y1 - y0), ly1);
            if (a < e) {
                const uint32_t ga = y0 + (a - cursor), ge = y0 + (e - cursor), lim = c->hdr.height - 1u;
                launched_rows += std::min(ge, lim) - std::min(ga, lim);
            }
            cursor += y1 - y0;
        }
        extra_rays = c->last_pixels - launched_rows * (uint64_t)(lx1 > lx0 ? lx1 - lx0 : 0u);
        // what the launched blocks do not cover is zero-filled by the launch's first workgroups (the kernel's fill_outside):
        // about a thousand of them, a share of the rows each
        fill_rect[0] = block_x0 * 16u, fill_rect[1] = std::min(c->hdr.width, (block_x0 + grid.x) * 16u);
        fill_rect[2] = ly0, fill_rect[3] = ly1;
        // about 160 KB of zeros per workgroup -- C5 (805 MB): 0.350 / 0.329 / 0.313 / 0.329 ms with 256 / 2048 / 4096 / 16384 of
        // them; a frame of 4096 blocks must not get as many again (RTC_AMD_FILL_WGS: development)
        const uint64_t frame_bytes = (uint64_t)rows * c->hdr.width * 12u;
        uint32_t fill_wgs = (uint32_t)std::min<uint64_t>(4096u, std::max<uint64_t>(16u, frame_bytes / (160u << 10)));
        if (P.fill_wgs != 0u) fill_wgs = P.fill_wgs;
        fill_wg_rows = std::max(1u, (fill_wgs + grid.x - 1u) / grid.x);
        fill_rows = (rows + fill_wg_rows * grid.x - 1u) / (fill_wg_rows * grid.x);
        fill_period = std::max(1u, (grid.y + fill_wg_rows) / fill_wg_rows);  // spread among the rendering rows: the fill shares the memory system with them
        if (out_u8) {
            // a frame of bytes: the zeros outside the rectangle are one asynchronous memset in front of the launch (the
            // kernel's filling workgroups write f32 rows)
            HIP_TRY(hipMemsetAsync(d_out_rgb, 0, (size_t)rows * c->hdr.width * 3u, stream));
            fill_wg_rows = 0u, fill_rows = 0u, fill_period = 1u;
        }
        grid.y += fill_wg_rows;
    }
    // the plain regular grid: blocks permuted within four rows (RenderArgs::swizzle), the grid padded to what that needs --
    // workgroups of the padding find no pixel of theirs inside the image
    const bool swizzle = P.swizzle && d_tiles == nullptr && rows > 0u && !rect_launch && blocks_y == 1u;
    if (swizzle) grid = dim3((grid.x + 1u) & ~1u, (grid.y + 3u) & ~3u);
    // A regular grid's frames after the first: the same 16 x 16 blocks, started in the order of their longest wave in the frame
    // before (refine_block_list: a list of one lane per pixel throughout).  Where the grid is the plain one -- one block per
    // workgroup, no scene rectangle, nobody waiting for rows in image order (rtc_render_ex's progress words) -- and the frame
    // has a tail worth the list: its longest wave is a tenth of its throughput time or more.
    if (d_tiles == nullptr && P.block_feedback && P.grid_feedback && plan == nullptr && rows > 0u && !rect_launch && blocks_y == 1u && share_log2 == 0u &&
        c->hdr.width <= 65532u && rows <= 131068u && !(c->hdr.n_trav != 0u && c->policy.wavefront)) {
        const std::array<uint32_t, 5> key = {q.band_rows, q.n_parts, q.part, 0xffffffffu, (uint32_t)depth};
        auto it = c->block_lists.find(key);
        if (it == c->block_lists.end()) {
            if (c->block_lists.size() >= 256u) {
                HIP_TRY(hipDeviceSynchronize());
                drop_block_lists(c);
            }
            BlockList bl;
            bl.n = (size_t)grid.x * grid.y;
            bl.state = BlockList::IDLE;
            it = c->block_lists.emplace(key, bl).first;
        }
        BlockList& bl = it->second;
        if (bl.state == BlockList::TIMED) {
            const rtc_status st = order_grid(c, bl, grid.x, grid.y, rows);
            if (st != RTC_OK) return st;
        }
        if (bl.state == BlockList::IDLE) {
            bl.state = BlockList::FRESH;  // (the next frame of this scene, if there is one, is measured)
        } else if (bl.state == BlockList::FRESH && bl.n == (size_t)grid.x * grid.y) {
            bl.counts = !(spec_fn && c->spec_shares);
            bl.swizzled = swizzle;
            const size_t nt = bl.n;
            HIP_TRY(grow(&bl.d_ticks, &bl.ticks_cap, 4u * nt * (bl.counts ? sizeof(uint4) : sizeof(uint32_t))));
            if (bl.counts) {
                copy_counts_to = bl.d_ticks;  // (after the launch, on its stream)
            } else {
                HIP_TRY(hipMemsetAsync(bl.d_ticks, 0, 4u * nt * sizeof(uint32_t), stream));
                d_ticks = bl.d_ticks;
            }
            bl.state = BlockList::TIMED;
        } else if (bl.state == BlockList::REFINED && bl.listed) {
            d_tiles = bl.d;
            grid = dim3((uint32_t)bl.n_listed, 1);
        }
    }
    const size_t n_blocks = (size_t)grid.x * grid.y * 4;  // partial counts: one per wave
    if (n_blocks > c->block_cap) {  // grow-only workspace (first call / larger image only)
        if (c->d_block_counts) HIP_TRY(hipFree(c->d_block_counts));
        c->d_block_counts = nullptr;
        HIP_TRY(hipMalloc(&c->d_block_counts, n_blocks * sizeof(uint4)));
        c->block_cap = n_blocks;
    }
    // progress reporting: a regular grid only (one workgroup per block, every block of the partition launched)
    uint32_t chunk_block_rows = 1u, n_chunks = 0u;
    if (plan && plan->d_done && rows > 0u && d_tiles == nullptr && !rect_launch && blocks_y == 1u) {
        const uint32_t want = std::max(1u, std::min(plan->want_chunks, PROGRESS_MAX_CHUNKS));
        chunk_block_rows = (grid.y + want - 1u) / want;
        if (swizzle) chunk_block_rows = (chunk_block_rows + 3u) & ~3u;  // (block rows finish four at a time)
        n_chunks = (grid.y + chunk_block_rows - 1u) / chunk_block_rows;
        const size_t words = ((size_t)grid.y + n_chunks) * PROGRESS_STRIDE;
        if (words > c->progress_cap) {
            if (c->d_progress) HIP_TRY(hipFree(c->d_progress));
            c->d_progress = nullptr;
            c->progress_cap = 0;
            HIP_TRY(hipMalloc(&c->d_progress, words * sizeof(uint32_t)));
            c->progress_cap = words;
        }
        HIP_TRY(hipMemsetAsync(c->d_progress, 0, words * sizeof(uint32_t), stream));
        plan->n_chunks = n_chunks;
        plan->chunk_rows = chunk_block_rows * bh;
    }
    if (rows == 0) {  // nothing to launch: the slot's counters read zero
        HIP_TRY(hipMemsetAsync(c->d_total + 3 * (size_t)slot, 0, 3 * sizeof(unsigned long long), stream));
        if (slot == 0) c->rendered = false;
        return RTC_OK;
    }
    RenderArgs a;
    a.hdr = c->hdr;
    a.soa = soa_view(c->d_soa, c->hdr, c->d_texels);
    a.out = out_u8 ? nullptr : (float*)d_out_rgb;
    a.out_u8 = out_u8 ? (uint8_t*)d_out_rgb : nullptr;
    a.progress = n_chunks ? c->d_progress : nullptr;
    a.done = n_chunks ? plan->d_done : nullptr;
    a.chunk_block_rows = chunk_block_rows;
    a.epoch = plan ? plan->epoch : 0u;
    a.block_counts = c->d_block_counts;
    a.total = c->d_total + 3 * (size_t)slot;
    a.rows = rows;
    a.band_rows = q.band_rows;
    a.n_parts = q.n_parts;
    a.part = q.part;
    a.depth = depth;
    a.share_log2 = share_log2;
    a.tiles = d_tiles;
    a.wave_ticks = d_ticks;
    a.blocks_y = blocks_y;
    a.swizzle = (swizzle && d_tiles == nullptr) ? 1u : 0u;
    a.block_x0 = block_x0;
    a.block_y0 = block_y0;
    a.fill_wg_rows = fill_wg_rows, a.fill_rows = fill_rows, a.fill_period = fill_period;
    a.fill_x0 = fill_rect[0], a.fill_x1 = fill_rect[1], a.fill_y0 = fill_rect[2], a.fill_y1 = fill_rect[3];
    if (c->events_used == c->events.size()) {
        if (c->events.size() >= 4096) {
            c->events_used = 0;  // nobody is reading the timings: recycle
        } else {
            std::pair<hipEvent_t, hipEvent_t> e;
            HIP_TRY(hipEventCreate(&e.first));
            HIP_TRY(hipEventCreate(&e.second));
            c->events.push_back(e);
        }
    }
    auto& ev = c->events[c->events_used++];
    // The FIRST launch of a code object on a queue pays what is not the kernel's: the runtime moves the code to the device and
    // sizes the queue's scratch for it -- 6 to 11 ms in a fresh process for these kernels (tools/first_frame_probe.py: C3's first
    // frame 0.95 ms cold, 0.71 once any context of the process has launched the same code; mesh 11.2 / 3.4).  It is paid once per
    // process and queue, here: one workgroup of the same kernel over zero rows (it finds no pixel of its own and writes only
    // the counters the real launch overwrites), in front of the events that time the frame.
    {
        const void* fn_key = spec_fn ? (const void*)spec_fn : (const void*)(uintptr_t)(0x1000u + (c->hdr.n_trav ? 1u : c->n_objects <= 4 ? 2u : c->n_objects <= 8 ? 4u : 6u) + (c->simple ? 1u : 0u));
        const auto key = std::make_pair(fn_key, (const void*)stream);
        if (!c->warmed.count(key)) {
            c->warmed.insert(key);
            RenderArgs w = a;
            w.rows = 0u, w.tiles = nullptr, w.wave_ticks = nullptr, w.progress = nullptr, w.done = nullptr, w.fill_wg_rows = 0u, w.swizzle = 0u, w.blocks_y = 1u;
            const dim3 one(1, 1);
            if (spec_fn) {
                void* wp[] = {&w};
                HIP_TRY(hipModuleLaunchKernel(spec_fn, 1, 1, 1, block.x, 1, 1, 0, stream, wp, nullptr));
            } else if (c->hdr.n_trav) hipLaunchKernelGGL((render_kernel<-1, false>), one, block, 0, stream, w);
            else if (c->n_objects <= 4 && c->simple) hipLaunchKernelGGL((render_kernel<4, true>), one, block, 0, stream, w);
            else if (c->n_objects <= 4) hipLaunchKernelGGL((render_kernel<4, false>), one, block, 0, stream, w);
            else if (c->n_objects <= 8 && c->simple) hipLaunchKernelGGL((render_kernel<8, true>), one, block, 0, stream, w);
            else if (c->n_objects <= 8) hipLaunchKernelGGL((render_kernel<8, false>), one, block, 0, stream, w);
            else hipLaunchKernelGGL((render_kernel<0, false>), one, block, 0, stream, w);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipEventRecord(ev.first, stream));
    if (tile_launch && n_fill_jobs) {  // the zero-fill of the unlisted tiles runs beside the render kernel: memory-bound work under arithmetic
        HIP_TRY(hipEventRecord(c->ev_fork, stream));
        HIP_TRY(hipStreamWaitEvent(c->fill_stream, c->ev_fork, 0));
        // 96 workgroups share the jobs: enough to move 690 MB in the time C5's tiles take to render, few enough to leave the chip's wave
        // slots to the render kernel (C5 8192^2, whole frame: 0.84 / 0.46 / 0.33 / 0.294 / 0.30 / 0.36 / 0.38 ms with 16 / 32 / 64 / 96 /
        // 128 / 512 / 4096 of them; the bounding rectangle with its interleaved fill: 0.335; profiles/r04_c5_tile_fill.txt)
        const uint32_t fill_wgs = P.tile_fill_wgs ? P.tile_fill_wgs : 96u;
        hipLaunchKernelGGL(fill_tiles_kernel, dim3((uint32_t)std::min<size_t>(n_fill_jobs, fill_wgs)), dim3(256), 0, c->fill_stream, fill_jobs,
                           (uint32_t)n_fill_jobs, (uint8_t*)d_out_rgb, c->hdr.width, rows, out_u8 ? 3u : 12u);
        HIP_TRY(hipEventRecord(c->ev_join, c->fill_stream));
    }
    if (timed_list) HIP_TRY(hipEventRecord(timed_list->ev0, stream));
    // instantiation: <= 4 / <= 8 objects get fully unrolled object loops (SIMPLE: all of them
    // scale+translate-only, no cylinder); anything larger takes the generic loop
    if (spec_fn) {
        void* params[] = {&a};
        HIP_TRY(hipModuleLaunchKernel(spec_fn, grid.x, grid.y, 1, block.x, 1, 1, 0, stream, params, nullptr));
    } else if (c->hdr.n_trav) hipLaunchKernelGGL((render_kernel<-1, false>), grid, block, 0, stream, a);
    else if (c->n_objects <= 4 && c->simple) hipLaunchKernelGGL((render_kernel<4, true>), grid, block, 0, stream, a);
    else if (c->n_objects <= 4) hipLaunchKernelGGL((render_kernel<4, false>), grid, block, 0, stream, a);
    else if (c->n_objects <= 8 && c->simple) hipLaunchKernelGGL((render_kernel<8, true>), grid, block, 0, stream, a);
    else if (c->n_objects <= 8) hipLaunchKernelGGL((render_kernel<8, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((render_kernel<0, false>), grid, block, 0, stream, a);
    if (tile_launch && n_fill_jobs) HIP_TRY(hipStreamWaitEvent(stream, c->ev_join, 0));
    HIP_TRY(hipEventRecord(ev.second, stream));
    if (timed_list) {
        HIP_TRY(hipEventRecord(timed_list->ev1, stream));
        timed_list->ev_recorded = true;
    }
    if (copy_counts_to) HIP_TRY(hipMemcpyAsync(copy_counts_to, c->d_block_counts, n_blocks * sizeof(uint4), hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(sum_counts_kernel, dim3((uint32_t)((n_blocks + SUM_COUNTS_SLICE - 1) / SUM_COUNTS_SLICE)), dim3(1024), 0, stream,
                       c->d_block_counts, (uint32_t)n_blocks, c->d_total + 3 * (size_t)slot, extra_rays);
    HIP_TRY(hipGetLastError());
    c->rendered = true;
    return RTC_OK;
}

// After the caller has synchronised with every launch: counters summed over slots [0, n_slots), kernel_ms = the SUM
// of the launches' HIP-event times since the last stats call (the launches of one frame run back to back).
void rtc::ctx_mark_one_shot(rtc_ctx* c) {
    if (c) c->lazy_jit = true;
}

rtc_status rtc::ctx_collect(rtc_ctx* c, uint32_t n_slots, rtc_stats* out) {
    std::memset(out, 0, sizeof(*out));
    if (n_slots > CTX_TOTAL_SLOTS) n_slots = CTX_TOTAL_SLOTS;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<unsigned long long> total(3 * (size_t)n_slots, 0ull);
    if (n_slots) HIP_TRY(hipMemcpy(total.data(), c->d_total, total.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (uint32_t s = 0; s < n_slots; s++) {
        out->rays += total[3 * s];
        out->shaded_hits += total[3 * s + 1];
        out->culled_shadow_rays += total[3 * s + 2];
    }
    double sum_ms = 0.0;
    for (size_t i = 0; i < c->events_used; i++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->events[i].first, c->events[i].second));
        sum_ms += ms;
    }
    out->kernel_ms = (float)sum_ms;
    out->launches = (uint32_t)c->events_used;
    out->flags = c->jit_note.empty() ? 0u : RTC_STATS_JIT_FALLBACK;
    c->events_used = 0;
    return RTC_OK;
}

extern "C" {

// Diagnostics (not in rtc.h; tests/test_block_lists.py, no device needed): refine_block_list and simulate_dispatch as
// rtc_ctx_render uses them.  -> the number of blocks of the new list (its first min(that, cap) are written to `out`).
uint32_t rtc_diag_refine_block_list(const uint32_t* list, const uint32_t* ticks, uint32_t n, uint32_t width, uint32_t rows, double wave_slots,
                                    double threshold, double down, uint32_t* out, uint32_t cap) {
    std::vector<uint32_t> l(list, list + n), refined;
    refine_block_list(l, ticks, width, rows, wave_slots, threshold, down, &refined);
    for (size_t i = 0; i < refined.size() && i < cap; i++) out[i] = refined[i];
    return (uint32_t)refined.size();
}
double rtc_diag_simulate_dispatch(const uint32_t* cost, uint32_t n, uint32_t slots) {
    return simulate_dispatch(std::vector<uint32_t>(cost, cost + n), slots);
}

rtc_status rtc_ctx_render(rtc_ctx* c, int32_t depth, const rtc_partition* part, void* d_out_rgb, void* stream) {
    return rtc::ctx_render_slot(c, depth, part, d_out_rgb, stream, 0u);
}

rtc_status rtc_ctx_stats(rtc_ctx* c, rtc_stats* out) {
    if (!c || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_stats: null argument");
    std::memset(out, 0, sizeof(*out));
    out->rows = c->last_rows;
    out->pixels = c->last_pixels;
    if (!c->rendered) return RTC_OK;
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long total[3] = {0, 0, 0};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(total, c->d_total, sizeof(total), hipMemcpyDeviceToHost));
    double sum_ms = 0.0;
    for (size_t i = 0; i < c->events_used; i++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->events[i].first, c->events[i].second));
        sum_ms += ms;
    }
    out->rays = total[0];
    out->shaded_hits = total[1];
    out->culled_shadow_rays = total[2];
    out->launches = (uint32_t)c->events_used;
    out->kernel_ms = c->events_used ? (float)(sum_ms / (double)c->events_used) : 0.0f;
    out->flags = c->jit_note.empty() ? 0u : RTC_STATS_JIT_FALLBACK;
    c->events_used = 0;
    return RTC_OK;
}

const char* rtc_ctx_kernel_name(rtc_ctx* c) {
    if (!c) return "";
    if (c->wf_last) {  // the frame before this call was rendered by rtc_wavefront.h's kernels, not by the scene's per-pixel kernel
        c->wf_name = "wavefront[tree walk of " + c->kernel_name + "]";
        return c->wf_name.c_str();
    }
    return c->kernel_name.c_str();
}
const char* rtc_ctx_jit_status(rtc_ctx* c) { return c ? c->jit_note.c_str() : ""; }
const char* rtc_ctx_kernel_id(rtc_ctx* c) { return c ? c->kernel_id.c_str() : ""; }

rtc_status rtc_ctx_quantize(rtc_ctx* c, const void* d_rgb, uint64_t n, void* d_out_u8, void* stream_) {
    if (!c || !d_rgb || !d_out_u8) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_quantize: null argument");
    if (n == 0) return RTC_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(quantize_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream_, (const float*)d_rgb, n,
                       (uint8_t*)d_out_u8);
    HIP_TRY(hipGetLastError());
    return RTC_OK;
}

uint64_t rtc_ppm_max_bytes(uint32_t width, uint32_t height) {
    // header "P3\n<w> <h>\n255\n" (<= 32 bytes) + at most 4 bytes per colour channel
    return 32ull + (uint64_t)width * height * 12ull;
}

rtc_status rtc_ctx_to_ppm(rtc_ctx* c, const void* d_rgb, uint32_t width, uint32_t height, void* d_text, uint64_t cap,
                          uint64_t* out_len, void* stream_) {
    if (!c || !d_rgb || !d_text || !out_len) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_to_ppm: null argument");
    if (width == 0 || height == 0) return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_to_ppm: empty canvas");
    if (cap < rtc_ppm_max_bytes(width, height))
        return fail(RTC_ERR_INVALID_ARG, "rtc_ctx_to_ppm: text buffer must hold rtc_ppm_max_bytes() = %llu bytes",
                    (unsigned long long)rtc_ppm_max_bytes(width, height));
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t words = (3u * width + 31u) / 32u;
    if ((size_t)height + 1 > c->ppm_rows_cap) {
        if (c->d_ppm_rows) HIP_TRY(hipFree(c->d_ppm_rows));
        c->d_ppm_rows = nullptr;
        HIP_TRY(hipMalloc(&c->d_ppm_rows, ((size_t)height + 1) * sizeof(unsigned long long)));
        c->ppm_rows_cap = (size_t)height + 1;
    }
    if ((size_t)height * words > c->ppm_bits_cap) {
        if (c->d_ppm_bits) HIP_TRY(hipFree(c->d_ppm_bits));
        c->d_ppm_bits = nullptr;
        HIP_TRY(hipMalloc(&c->d_ppm_bits, (size_t)height * words * sizeof(uint32_t)));
        c->ppm_bits_cap = (size_t)height * words;
    }
    char head[40];
    const int head_len = snprintf(head, sizeof(head), "P3\n%u %u\n255\n", width, height);  // canvas.rs:60-63
    HIP_TRY(hipMemcpyAsync(d_text, head, (size_t)head_len, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(ppm_row_scan_kernel, dim3((height + 63) / 64), dim3(64), 0, stream, (const float*)d_rgb, width, height,
                       c->d_ppm_rows, c->d_ppm_bits, words);
    hipLaunchKernelGGL(ppm_row_offsets_kernel, dim3(1), dim3(1024), 0, stream, c->d_ppm_rows, height,
                       (unsigned long long)head_len, c->d_ppm_rows + height);
    hipLaunchKernelGGL(ppm_emit_kernel, dim3((height + 3) / 4), dim3(256), 0, stream, (const float*)d_rgb, width, height,
                       c->d_ppm_rows, c->d_ppm_bits, words, (char*)d_text);
    HIP_TRY(hipGetLastError());
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, c->d_ppm_rows + height, sizeof(total), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));  // `head` and `total` are host stack memory
    *out_len = total;
    return RTC_OK;
}

// ---- batched test/utility entry points (host buffers) ---------------------------
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
rtc_status begin_batch(const rtc_scene* scene, int32_t device, SceneHdr* hdr, DevBuf* soa_buf, DevBuf* tex_buf, bool allow_sequence = false) {
    int n = usable_devices();
    if (n <= 0) return fail(RTC_ERR_NO_DEVICE, "no HIP device visible; librtc_amd has no CPU fallback");
    if (device < 0 || device >= n) return fail(RTC_ERR_INVALID_ARG, "device %d out of range (have %d)", device, n);
    std::vector<float4> soa;
    std::vector<float> texels;
    rtc_status st = flatten(Policy::from_env(), scene, nullptr, hdr, &soa, &texels, nullptr, nullptr, allow_sequence);
    if (st != RTC_OK) return st;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(soa_buf->alloc(soa.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(soa_buf->p, soa.data(), soa.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(tex_buf->alloc(texels.size() * sizeof(float)));
    if (!texels.empty()) HIP_TRY(hipMemcpy(tex_buf->p, texels.data(), texels.size() * sizeof(float), hipMemcpyHostToDevice));
    return RTC_OK;
}
}  // namespace

rtc_status rtc_color_at(const rtc_scene* scene, const float* origins, const float* directions, uint32_t n,
                        int32_t depth, int32_t device, float* out_rgb) {
    if (!origins || !directions || !out_rgb) return fail(RTC_ERR_INVALID_ARG, "rtc_color_at: null argument");
    if (depth < 0 || depth > RTC_STACK_DEPTH_BASE) return fail(RTC_ERR_INVALID_ARG, "rtc_color_at: depth %d outside [0, %d]", depth, RTC_STACK_DEPTH_BASE);
    if (n == 0) return RTC_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (origins[i * 4 + 3] != 1.0f || directions[i * 4 + 3] != 0.0f)
            return fail(RTC_ERR_INVALID_ARG, "ray %u: origin.w must be 1 and direction.w 0", i);
    }
    SceneHdr hdr;
    DevBuf soa, tex, d_o, d_d, d_out;
    rtc_status st = begin_batch(scene, device, &hdr, &soa, &tex);
    if (st != RTC_OK) return st;
    HIP_TRY(d_o.alloc((size_t)n * 16));
    HIP_TRY(d_d.alloc((size_t)n * 16));
    HIP_TRY(d_out.alloc((size_t)n * 12));
    HIP_TRY(hipMemcpy(d_o.p, origins, (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d.p, directions, (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(color_at_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, hdr,
                       soa_view((const float4*)soa.p, hdr, (const float*)tex.p), (const float4*)d_o.p, (const float4*)d_d.p, n,
                       depth, (float*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out_rgb, d_out.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    return RTC_OK;
}

rtc_status rtc_intensity_at(const rtc_scene* scene, const float* points, uint32_t n, int32_t device, float* out) {
    if (!points || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_intensity_at: null argument");
    if (n == 0) return RTC_OK;
    SceneHdr hdr;
    DevBuf soa, tex, d_p, d_out;
    rtc_status st = begin_batch(scene, device, &hdr, &soa, &tex, /*allow_sequence=*/true);
    if (st != RTC_OK) return st;
    HIP_TRY(d_p.alloc((size_t)n * 16));
    HIP_TRY(d_out.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(d_p.p, points, (size_t)n * 16, hipMemcpyHostToDevice));
    bool simple = !hdr.has_patterns && hdr.n_objects <= 4 && !hdr.n_trav;  // (as rtc_ctx_set_scene decides it for the render kernels)
    for (uint32_t i = 0; simple && i < hdr.n_objects; i++) {
        const uint32_t kind = (uint32_t)scene->objects[i].kind;
        const float* m = scene->objects[i].inv;
        const bool diag = m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f && m[6] == 0.0f && m[8] == 0.0f && m[9] == 0.0f;
        if (!diag || kind == RTC_CYLINDER || kind == RTC_CONE || kind == RTC_TRIANGLE) simple = false;
    }
    if (simple)
        hipLaunchKernelGGL(intensity_at_kernel_simple, dim3((n + 63) / 64), dim3(64), 0, nullptr, hdr,
                           soa_view((const float4*)soa.p, hdr, (const float*)tex.p), (const float4*)d_p.p, n, (float*)d_out.p);
    else if (hdr.n_objects <= 4 && !hdr.n_trav)
        hipLaunchKernelGGL(intensity_at_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, hdr,
                           soa_view((const float4*)soa.p, hdr, (const float*)tex.p), (const float4*)d_p.p, n, (float*)d_out.p);
    else
        hipLaunchKernelGGL(intensity_at_kernel_generic, dim3((n + 63) / 64), dim3(64), 0, nullptr, hdr,
                           soa_view((const float4*)soa.p, hdr, (const float*)tex.p), (const float4*)d_p.p, n, (float*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RTC_OK;
}

rtc_status rtc_point_on_light(const rtc_light* light, const int32_t* cells_uv, uint32_t n, int32_t device, float* out) {
    if (!light || !cells_uv || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_point_on_light: null argument");
    if (light->kind != RTC_LIGHT_RECT) return fail(RTC_ERR_INVALID_ARG, "rtc_point_on_light: not a rectangle light");
    if (n == 0) return RTC_OK;
    for (uint32_t i = 0; i < n; i++)
        if (cells_uv[2 * i] < 0 || cells_uv[2 * i] >= light->u_steps || cells_uv[2 * i + 1] < 0 || cells_uv[2 * i + 1] >= light->v_steps)
            return fail(RTC_ERR_INVALID_ARG, "rtc_point_on_light: cell %u (%d, %d) outside the light's %d x %d", i, cells_uv[2 * i], cells_uv[2 * i + 1],
                        light->u_steps, light->v_steps);
    rtc_scene sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.light = light;
    SceneHdr hdr;
    DevBuf soa, tex, d_c, d_out;
    rtc_status st = begin_batch(&sc, device, &hdr, &soa, &tex, /*allow_sequence=*/true);
    if (st != RTC_OK) return st;
    HIP_TRY(d_c.alloc((size_t)n * 8));
    HIP_TRY(d_out.alloc((size_t)n * 16));
    HIP_TRY(hipMemcpy(d_c.p, cells_uv, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(point_on_light_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, hdr, (const int2*)d_c.p, n, (float4*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    return RTC_OK;
}

rtc_status rtc_is_shadowed(const rtc_scene* scene, const float* light_positions, const float* points, uint32_t n,
                           int32_t device, int32_t* out) {
    if (!light_positions || !points || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_is_shadowed: null argument");
    if (n == 0) return RTC_OK;
    SceneHdr hdr;
    DevBuf soa, tex, d_l, d_p, d_out;
    rtc_status st = begin_batch(scene, device, &hdr, &soa, &tex);
    if (st != RTC_OK) return st;
    HIP_TRY(d_l.alloc((size_t)n * 16));
    HIP_TRY(d_p.alloc((size_t)n * 16));
    HIP_TRY(d_out.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(d_l.p, light_positions, (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_p.p, points, (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(is_shadowed_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, hdr,
                       soa_view((const float4*)soa.p, hdr, (const float*)tex.p), (const float4*)d_l.p, (const float4*)d_p.p, n,
                       (int32_t*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RTC_OK;
}

rtc_status rtc_powf(const float* x, const float* y, uint32_t n, int32_t device, float* out) {
    if (!x || !y || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_powf: null argument");
    if (n == 0) return RTC_OK;
    int nd = usable_devices();
    if (nd <= 0) return fail(RTC_ERR_NO_DEVICE, "no HIP device visible; librtc_amd has no CPU fallback");
    if (device < 0 || device >= nd) return fail(RTC_ERR_INVALID_ARG, "device %d out of range (have %d)", device, nd);
    HIP_TRY(hipSetDevice(device));
    DevBuf d_x, d_y, d_out;
    HIP_TRY(d_x.alloc((size_t)n * 4));
    HIP_TRY(d_y.alloc((size_t)n * 4));
    HIP_TRY(d_out.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(d_x.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_y.p, y, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(powf_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)d_x.p, (const float*)d_y.p, n,
                       (float*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RTC_OK;
}

rtc_status rtc_cosf(const float* x, uint32_t n, int32_t device, float* out) {
    if (!x || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_cosf: null argument");
    if (n == 0) return RTC_OK;
    rtc_status st = select_device(device);
    if (st != RTC_OK) return st;
    DevBuf d_x, d_out;
    HIP_TRY(d_x.alloc((size_t)n * 4));
    HIP_TRY(d_out.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(d_x.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cosf_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)d_x.p, n, (float*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RTC_OK;
}

rtc_status rtc_atan2f(const float* y, const float* x, uint32_t n, int32_t device, float* out) {
    if (!x || !y || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_atan2f: null argument");
    if (n == 0) return RTC_OK;
    rtc_status st = select_device(device);
    if (st != RTC_OK) return st;
    DevBuf d_x, d_y, d_out;
    HIP_TRY(d_x.alloc((size_t)n * 4));
    HIP_TRY(d_y.alloc((size_t)n * 4));
    HIP_TRY(d_out.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(d_x.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_y.p, y, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(atan2f_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)d_y.p, (const float*)d_x.p, n,
                       (float*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RTC_OK;
}
rtc_status rtc_acosf(const float* x, uint32_t n, int32_t device, float* out) {
    if (!x || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_acosf: null argument");
    if (n == 0) return RTC_OK;
    rtc_status st = select_device(device);
    if (st != RTC_OK) return st;
    DevBuf d_x, d_out;
    HIP_TRY(d_x.alloc((size_t)n * 4));
    HIP_TRY(d_out.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(d_x.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(acosf_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)d_x.p, n, (float*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RTC_OK;
}

// One object's geometry as a kernel argument for the batched shape / pattern entry points.
static rtc_status object_arg(const rtc_object* object, const char* who, Obj* ob, DevBuf* d_tri) {
    rtc_object unit;
    if (!object) {  // an untransformed unit sphere
        std::memset(&unit, 0, sizeof(unit));
        unit.kind = RTC_SPHERE;
        unit.casts_shadow = 1;
        unit.min_y = -INFINITY;
        unit.max_y = INFINITY;
        for (int i = 0; i < 4; i++) unit.inv[i * 5] = 1.0f;
        object = &unit;
    }
    if (object->kind < RTC_SPHERE || object->kind > RTC_TRIANGLE)
        return fail(RTC_ERR_UNSUPPORTED, "%s: shape kind %d is not on the device path", who, object->kind);
    if (!is_affine(object->inv)) return fail(RTC_ERR_UNSUPPORTED, "%s: inverse transform is not affine", who);
    float4 g[4];
    pack_geometry(*object, g);
    ob->geo = g[0];
    ob->off0 = g[1];
    ob->off1 = g[2];
    ob->off2 = g[3];
    ob->trn = make_float4(object->inv[3], object->inv[7], object->inv[11], 0.0f);
    std::memcpy(&ob->bits, &g[0].w, 4);
    float4 tri[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (object->kind == RTC_TRIANGLE) pack_triangle(*object, tri);
    HIP_TRY(d_tri->alloc(sizeof(tri)));
    HIP_TRY(hipMemcpy(d_tri->p, tri, sizeof(tri), hipMemcpyHostToDevice));
    return RTC_OK;
}

rtc_status rtc_local_intersect(const rtc_object* object, const float* origins, const float* directions, uint32_t n,
                               int32_t device, float* out_t, int32_t* out_count) {
    if (!object || !origins || !directions || !out_t || !out_count)
        return fail(RTC_ERR_INVALID_ARG, "rtc_local_intersect: null argument");
    Obj ob;
    DevBuf d_tri;
    rtc_status st = select_device(device);
    if (st != RTC_OK) return st;
    if ((st = object_arg(object, "rtc_local_intersect", &ob, &d_tri)) != RTC_OK) return st;
    if (n == 0) return RTC_OK;
    DevBuf d_o, d_d, d_t, d_c;
    HIP_TRY(d_o.alloc((size_t)n * 16));
    HIP_TRY(d_d.alloc((size_t)n * 16));
    HIP_TRY(d_t.alloc((size_t)n * 16));
    HIP_TRY(d_c.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(d_o.p, origins, (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d.p, directions, (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(local_intersect_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, ob, (const float4*)d_tri.p, (const float4*)d_o.p,
                       (const float4*)d_d.p, n, (float4*)d_t.p, (int32_t*)d_c.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out_t, d_t.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_count, d_c.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RTC_OK;
}

rtc_status rtc_normal_at(const rtc_object* object, const float* world_points, uint32_t n, int32_t device, float* out) {
    if (!object || !world_points || !out) return fail(RTC_ERR_INVALID_ARG, "rtc_normal_at: null argument");
    Obj ob;
    DevBuf d_tri;
    rtc_status st = select_device(device);
    if (st != RTC_OK) return st;
    if ((st = object_arg(object, "rtc_normal_at", &ob, &d_tri)) != RTC_OK) return st;
    if (n == 0) return RTC_OK;
    DevBuf d_p, d_out;
    HIP_TRY(d_p.alloc((size_t)n * 16));
    HIP_TRY(d_out.alloc((size_t)n * 16));
    HIP_TRY(hipMemcpy(d_p.p, world_points, (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(normal_at_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, ob, (const float4*)d_tri.p, (const float4*)d_p.p, n,
                       (float4*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    return RTC_OK;
}

rtc_status rtc_pattern_color_at(const rtc_pattern* pattern, const rtc_object* object, const float* world_points,
                                uint32_t n, int32_t device, float* out_rgb) {
    if (!pattern || !world_points || !out_rgb) return fail(RTC_ERR_INVALID_ARG, "rtc_pattern_color_at: null argument");
    if (pattern->kind < RTC_PATTERN_STRIPES || pattern->kind > RTC_PATTERN_CUBE_MAP)
        return fail(RTC_ERR_UNSUPPORTED, "rtc_pattern_color_at: pattern kind %d is not on the device path", pattern->kind);
    if (!is_affine(pattern->inv)) return fail(RTC_ERR_UNSUPPORTED, "rtc_pattern_color_at: pattern inverse transform is not affine");
    Obj ob;
    DevBuf d_tri;
    rtc_status st = select_device(device);
    if (st != RTC_OK) return st;
    if ((st = object_arg(object, "rtc_pattern_color_at", &ob, &d_tri)) != RTC_OK) return st;
    if (n == 0) return RTC_OK;
    float4 rec[5];
    pack_pattern(*pattern, rec);
    std::vector<float4> uvrec;
    std::vector<float> texels;
    std::vector<std::pair<const float*, size_t>> seen;
    if (pattern->kind >= RTC_PATTERN_TEXTURE_MAP && (st = pack_texture_map(*pattern, rec, &uvrec, &texels, &seen)) != RTC_OK) return st;
    DevBuf d_pat, d_p, d_out, d_uv, d_tex;
    HIP_TRY(d_uv.alloc(uvrec.size() * sizeof(float4)));
    HIP_TRY(d_tex.alloc(texels.size() * sizeof(float)));
    if (!uvrec.empty()) HIP_TRY(hipMemcpy(d_uv.p, uvrec.data(), uvrec.size() * sizeof(float4), hipMemcpyHostToDevice));
    if (!texels.empty()) HIP_TRY(hipMemcpy(d_tex.p, texels.data(), texels.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(d_pat.alloc(sizeof(rec)));
    HIP_TRY(d_p.alloc((size_t)n * 16));
    HIP_TRY(d_out.alloc((size_t)n * 12));
    HIP_TRY(hipMemcpy(d_pat.p, rec, sizeof(rec), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_p.p, world_points, (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pattern_color_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, ob, (const float4*)d_pat.p,
                       (const float4*)d_uv.p, (const float*)d_tex.p, (const float4*)d_p.p, n, (float*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out_rgb, d_out.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    return RTC_OK;
}

// Host compile of the same powf restatement (diagnostic: lets the CPU test
// suite pin the algorithm against the C library without a GPU).  Not used by
// any render path.
void rtc_powf_host(const float* x, const float* y, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = powf_glibc(x[i], y[i], h_pow_log2_tab, h_exp2f_tab);
}

// Same for the cosf restatement (pattern/sine_2d.rs:40).
void rtc_cosf_host(const float* x, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = cosf_glibc(x[i], h_sincosf_tab, h_inv_pio4);
}

// Same for atan2f / acosf (pattern/uv.rs:108,115).
void rtc_atan2f_host(const float* y, const float* x, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = atan2f_glibc(y[i], x[i]);
}
void rtc_acosf_host(const float* x, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; i++) out[i] = acosf_glibc(x[i]);
}

// Diagnostic (not in rtc.h): runs fastmath_selftest_kernel on n host vectors (n*3 f32);
// counts[0] = vectors inside the core range, counts[1] = mismatches against sqrtf and '/'.
rtc_status rtc_selftest_fastmath(const float* vectors, uint32_t n, int32_t device, uint32_t counts[2]) {
    if (!vectors || !counts) return fail(RTC_ERR_INVALID_ARG, "rtc_selftest_fastmath: null argument");
    int nd = usable_devices();
    if (nd <= 0) return fail(RTC_ERR_NO_DEVICE, "no HIP device visible; librtc_amd has no CPU fallback");
    if (device < 0 || device >= nd) return fail(RTC_ERR_INVALID_ARG, "device %d out of range (have %d)", device, nd);
    HIP_TRY(hipSetDevice(device));
    DevBuf d_v, d_out;
    HIP_TRY(d_v.alloc((size_t)n * 12));
    HIP_TRY(d_out.alloc(8));
    HIP_TRY(hipMemcpy(d_v.p, vectors, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_out.p, 0, 8));
    if (n)
        hipLaunchKernelGGL(fastmath_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, (const float*)d_v.p, n,
                           (uint32_t*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(counts, d_out.p, 8, hipMemcpyDeviceToHost));
    return RTC_OK;
}

}  // extern "C"
