// rtc_kernel_core.h -- the device side of librtc_amd.so: every __device__ function and kernel of the
// MI355X (gfx950 / CDNA4) render path.
//
// Compiled two ways from this one file:
//   * ahead of time by hipcc, included from rtc_device.hip (generic instantiations);
//   * at scene-set time by hiprtc with -DRTC_SPEC_LIST=... (see rtc_device.hip, "Scene-specialised
//     kernels"): the per-object kind / flag words become compile-time constants, which removes the
//     wave-uniform kind switches from the unrolled object loops (C3: 4.06 -> 3.13 ms, same bits).
//
// One wavefront lane per pixel, 8x8 pixel tiles per 64-lane wave.  The flattened scene (structure-of-arrays
// float4 records, see SceneSoA) lives in HBM; because every lane of a wave walks the same object list -- or, for
// worlds with GroupShapes, the same depth-first entry list, as a packet -- the records are fetched with
// wave-uniform addresses (scalar loads -> SGPRs) and cost no vector registers or LDS bandwidth.  Recursion
// (reflected_color / refracted_color -> color_at) is an explicit per-lane post-order stack, so that sums are
// formed in exactly the reference's order.  No MFMA: the path is branchy scalar f32 arithmetic.
//
// Contents, in order: the libm restatements (powf, cosf, atanf / atan2f, acosf); the scene layout; shape
// intersectors and normals (sphere, plane, cube, cylinder, cone, triangle); object loops and the group-tree
// packet walk; is_shadowed and its area-light fast path (origin hoisting, caster-first passes, exact sqrt /
// divide cores, light-cone culling); patterns and texture maps; phong, n1/n2, schlick; color_at; the render
// kernel; then the ahead-of-time-only utility kernels (counter sum, quantiser, PPM formatter, batched entry points).
//
// Bit-exactness rules (see DESIGN.md "Arithmetic contract"):
//   * whole file is compiled with -ffp-contract=off and the pragma below: the Rust reference never fuses a*b+c;
//   * every sum keeps the reference's association order;
//   * '/' and sqrtf are the correctly rounded IEEE forms (hipcc default);
//   * libm calls (powf, cos, atan2, acos) are restatements of glibc 2.35's routines -- the ones a Linux build of
//     the reference calls -- not ocml's;
//   * anything that skips work (shadow-ray culling, tree pruning, the plane shortcut) does so only where the
//     skipped evaluation's outcome is known, with margins documented at the site.
//
#ifndef RTC_KERNEL_CORE_H
#define RTC_KERNEL_CORE_H

#ifdef __HIPCC_RTC__
// hiprtc: HIP builtins and vector types are pre-included; there is no C++ standard library.
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#define RTC_HOSTDEV __device__
// rtc.h needs <stddef.h>/<stdint.h>, which hiprtc does not ship: restate the few constants the device
// code uses (the ahead-of-time build below static_asserts that they agree with rtc.h).
#define RTC_STACK_DEPTH_BASE 8
enum { RTC_SPHERE = 0, RTC_PLANE = 1, RTC_CUBE = 2, RTC_CYLINDER = 3, RTC_CONE = 4, RTC_TRIANGLE = 5 };
enum { RTC_PATTERN_NONE = 0, RTC_PATTERN_STRIPES = 1, RTC_PATTERN_GRADIENT = 2, RTC_PATTERN_RINGS = 3,
       RTC_PATTERN_CHECKERS = 4, RTC_PATTERN_SINE2D = 5, RTC_PATTERN_TEXTURE_MAP = 6, RTC_PATTERN_CUBE_MAP = 7 };
enum { RTC_UV_CHECKERS = 1, RTC_UV_ALIGN_CHECK = 2, RTC_UV_IMAGE = 3 };
enum { RTC_MAP_SPHERICAL = 1, RTC_MAP_PLANAR = 2, RTC_MAP_CYLINDRICAL = 3 };
enum { RTC_LIGHT_POINT = 0, RTC_LIGHT_RECT = 1 };
enum { RTC_JITTER_CONSTANT = 0, RTC_JITTER_HASHED = 2, RTC_JITTER_SEQUENCE = 3 };
#define RTC_JITTER_SEQUENCE_MAX 16
#else
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rtc.h"
#define RTC_HOSTDEV __host__ __device__
static_assert(RTC_STACK_DEPTH_BASE == 8 && RTC_MAX_DEPTH <= 255 && RTC_SPHERE == 0 && RTC_PLANE == 1 && RTC_CUBE == 2 && RTC_CYLINDER == 3 &&
                  RTC_CONE == 4 && RTC_TRIANGLE == 5 && RTC_PATTERN_NONE == 0 && RTC_PATTERN_STRIPES == 1 && RTC_PATTERN_GRADIENT == 2 &&
                  RTC_PATTERN_RINGS == 3 && RTC_PATTERN_CHECKERS == 4 && RTC_PATTERN_SINE2D == 5 &&
                  RTC_PATTERN_TEXTURE_MAP == 6 && RTC_PATTERN_CUBE_MAP == 7 && RTC_UV_CHECKERS == 1 &&
                  RTC_UV_ALIGN_CHECK == 2 && RTC_UV_IMAGE == 3 && RTC_MAP_SPHERICAL == 1 && RTC_MAP_PLANAR == 2 &&
                  RTC_MAP_CYLINDRICAL == 3 && RTC_LIGHT_POINT == 0 && RTC_LIGHT_RECT == 1 && RTC_JITTER_CONSTANT == 0 && RTC_JITTER_HASHED == 2 && RTC_JITTER_SEQUENCE == 3 && RTC_JITTER_SEQUENCE_MAX == 16,
              "rtc_kernel_core.h restates these rtc.h constants for the hiprtc build");
#endif

#pragma clang fp contract(off)

#define DI __device__ __forceinline__
#define HDI RTC_HOSTDEV __forceinline__
#define RTC_INF __builtin_huge_valf()
#define RTC_NAN __builtin_nanf("")

namespace rtc {

// ============================================================================
//  powf: glibc 2.35 sysdeps/ieee754/flt-32/e_powf.c (Szabolcs Nagy's
//  algorithm from ARM optimized-routines), FMA variant (__powf_fma), which is
//  what f32::powf (phong_lighting.rs:56) resolves to on an x86-64 Linux host
//  with FMA.  log2(x) by a 16-entry table + degree-5 polynomial, exp2 by a
//  32-entry table + degree-3 polynomial, all in double precision.
// ============================================================================
struct PowLog2Entry {
    double invc, logc;
};
__device__ __constant__ PowLog2Entry d_pow_log2_tab[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
__device__ __constant__ uint64_t d_exp2f_tab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
HDI uint32_t f2u(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
HDI float u2f(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
HDI uint64_t d2u(double d) {
    uint64_t u;
    __builtin_memcpy(&u, &d, 8);
    return u;
}
HDI double u2d(uint64_t u) {
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
}

HDI int pow_checkint(uint32_t iy) {  // 0: not an integer, 1: odd, 2: even
    int e = (iy >> 23) & 0xff;
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
HDI bool pow_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000 - 1; }

HDI float powf_glibc(float x, float y, const PowLog2Entry* __restrict__ T, const uint64_t* __restrict__ E) {
    uint32_t sign_bias = 0;
    uint32_t ix = f2u(x), iy = f2u(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || pow_zeroinfnan(iy)) {
        if (pow_zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
            if (2 * ix == 2 * 0x3f800000u) return 1.0f;
            if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (pow_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && pow_checkint(iy) == 1) {
                x2 = -x2;
                sign_bias = 1;
            }
            if (2 * ix == 0 && (iy & 0x80000000u)) return sign_bias ? -RTC_INF : RTC_INF;
            return (iy & 0x80000000u) ? 1 / x2 : x2;
        }
        if (ix & 0x80000000u) {
            int yint = pow_checkint(iy);
            if (yint == 0) return RTC_NAN;
            if (yint == 1) sign_bias = 1u << (5 + 11);  // SIGN_BIAS = 1 << (EXP2F_TABLE_BITS + 11)
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {  // normalise a subnormal x
            ix = f2u(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline: x = 2^k z, z in [OFF, 2*OFF); log2(x) = k + log2(c) + log2(z/c)
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> (23 - 4)) % 16;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double invc = T[i].invc, logc = T[i].logc;
    double z = (double)u2f(iz);
    double r = fma(z, invc, -1.0);
    double y0 = logc + (double)k;
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    double r2 = r * r;
    double yy = fma(A0, r, A1);
    double p = fma(A2, r, A3);
    double r4 = r2 * r2;
    double q = fma(A4, r, y0);
    q = fma(p, r2, q);
    double logx = fma(yy, r4, q);
    double ylogx = (double)y * logx;
    if (((d2u(ylogx) >> 47) & 0xffff) >= (d2u(126.0) >> 47)) {  // |y*log2(x)| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -RTC_INF : RTC_INF;  // __math_oflowf
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                       // __math_uflowf
        if (ylogx < -149.0) {                                                       // __math_may_uflowf
            float tiny = 0x1.4p-75f * 0x1.4p-75f;
            return sign_bias ? -tiny : tiny;
        }
    }
    // exp2_inline: x = k/N + r, 2^x = 2^(k/N) * 2^r
    const double SHIFT = 0x1.8p+52 / 32.0;
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    double kd = ylogx + SHIFT;
    uint64_t ki = d2u(kd);
    kd -= SHIFT;
    double rr = ylogx - kd;
    uint64_t t = E[ki % 32];
    uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    double s = u2d(t);
    double zz = fma(C0, rr, C1);
    double rr2 = rr * rr;
    double res = fma(C2, rr, 1.0);
    res = fma(zz, rr2, res);
    res = res * s;
    return (float)res;
}

DI float rtc_powf_dev(float x, float y) { return powf_glibc(x, y, d_pow_log2_tab, d_exp2f_tab); }

// ============================================================================
//  cosf: glibc 2.35 sysdeps/ieee754/flt-32/s_cosf.c + s_sincosf.h (same origin
//  as powf above), FMA variant (__cosf_fma) -- what f32::cos (pattern/sine_2d.rs:40)
//  resolves to on an x86-64 Linux host with FMA.  Argument reduction by pi/2
//  (fast path for |x| < 120, 192-bit 4/pi table above that), then a degree-8
//  cosine or degree-7 sine polynomial in double precision.
// ============================================================================
struct SinCosTab {
    double sign[4];
    double hpi_inv, hpi;
    double c0, c1, c2, c3, c4;
    double s1, s2, s3;
};
// (the initialisers are macros so that rtc_device.hip can build host copies for rtc_cosf_host)
#define RTC_SINCOSF_TAB_INIT                                                                                     \
    {{{1.0, -1.0, -1.0, 1.0}, 0x1.45f306dc9c883p+23, 0x1.921fb54442d18p+0,                                        \
      0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,          \
      -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},                                       \
     {{1.0, -1.0, -1.0, 1.0}, 0x1.45f306dc9c883p+23, 0x1.921fb54442d18p+0,                                        \
      -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,         \
      -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}}
// 4/pi as overlapping 32-bit words (__inv_pio4)
#define RTC_INV_PIO4_INIT                                                                                        \
    {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,              \
     0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,              \
     0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041}
__device__ __constant__ SinCosTab d_sincosf_tab[2] = RTC_SINCOSF_TAB_INIT;
__device__ __constant__ uint32_t d_inv_pio4[24] = RTC_INV_PIO4_INIT;
// sinf_poly: n even -> sine polynomial of x, n odd -> cosine polynomial of x (x2 = x*x)
HDI float sincosf_poly(double x, double x2, const SinCosTab* p, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = fma(p->s3, x2, p->s2);
        double x7 = x3 * x2;
        double s = fma(x3, p->s1, x);
        return (float)fma(x7, s1, s);
    }
    double x4 = x2 * x2;
    double c2 = fma(p->c4, x2, p->c3);
    double c1 = fma(p->c1, x2, p->c0);
    double x6 = x4 * x2;
    double c = fma(x4, p->c2, c1);
    return (float)fma(x6, c2, c);
}
HDI float cosf_glibc(float y, const SinCosTab* __restrict__ T, const uint32_t* __restrict__ IP) {
    double x = (double)y;
    const uint32_t ix = f2u(y);
    const uint32_t top = (ix >> 20) & 0x7ffu;  // abstop12
    if (top < 0x3f4u) {                        // |y| < pi/4
        double x2 = x * x;
        if (top < 0x398u) return 1.0f;         // |y| < 2^-12
        return sincosf_poly(x, x2, &T[0], 1);
    }
    if (top < 0x42fu) {  // |y| < 120: reduce_fast
        double r = x * T[0].hpi_inv;
        int n = ((int32_t)r + 0x800000) >> 24;
        x = fma(-(double)n, T[0].hpi, x);
        double s = T[0].sign[n & 3];
        const SinCosTab* p = (n & 2) ? &T[1] : &T[0];
        return sincosf_poly(x * s, x * x, p, n ^ 1);
    }
    if (top < 0x7f8u) {  // reduce_large
        const uint32_t* arr = &IP[(ix >> 26) & 15u];
        const int shift = (int)((ix >> 23) & 7u);
        uint32_t xi = ((ix & 0xffffffu) | 0x800000u) << shift;
        uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
        uint64_t res1 = (uint64_t)xi * arr[4];
        uint64_t res2 = (uint64_t)xi * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        uint64_t nn = (res0 + (1ULL << 61)) >> 62;
        res0 -= nn << 62;
        x = (double)(long long)res0 * 0x1.921fb54442d18p-62;
        int n = (int)nn;
        int sign = (int)(ix >> 31);
        double s = T[0].sign[(n + sign) & 3];
        const SinCosTab* p = ((n + sign) & 2) ? &T[1] : &T[0];
        return sincosf_poly(x * s, x * x, p, n ^ 1);
    }
    return RTC_NAN;  // inf or NaN
}
DI float rtc_cosf_dev(float x) { return cosf_glibc(x, d_sincosf_tab, d_inv_pio4); }

// ============================================================================
//  atanf / atan2f / acosf: glibc 2.35 sysdeps/ieee754/flt-32/{s_atanf,e_atan2f,e_acosf}.c -- the fdlibm
//  single-precision routines, which an x86-64 glibc builds as plain scalar SSE (no FMA variants exist for
//  them): every operation below is one IEEE f32 operation in the order the library performs it.  What
//  f32::atan2 / f32::acos (pattern/uv.rs:108,115) resolve to on a Linux host.  Constants are given by their
//  bit patterns, as read from this image's libm.so.6.
// ============================================================================
HDI float atanf_glibc(float x) {
    const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu;
    const float atanhi[4] = {u2f(0x3eed6338u), u2f(0x3f490fdau), u2f(0x3f7b985eu), u2f(0x3fc90fdau)};
    const float atanlo[4] = {u2f(0x31ac3769u), u2f(0x33222168u), u2f(0x33140fb4u), u2f(0x33a22168u)};
    if (ix >= 0x4c000000u) {  // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;
        if ((int32_t)hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    int id;
    if (ix < 0x3ee00000u) {             // |x| < 0.4375
        if (ix < 0x31000000u) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000u) {      // |x| < 1.1875
            if (ix < 0x3f300000u) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {  // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000u) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {  // 2.4375 <= |x| < 2^25
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    // even and odd terms of the polynomial, Horner in w, exactly as compiled
    float s1 = u2f(0x3c8569d7u) * w + u2f(0x3d4bda59u);
    s1 = s1 * w + u2f(0x3d886b35u);
    s1 = s1 * w + u2f(0x3dba2e6eu);
    s1 = s1 * w + u2f(0x3e124925u);
    s1 = s1 * w + u2f(0x3eaaaaabu);
    s1 = s1 * z;
    float s2 = u2f(0xbd15a221u) * w - u2f(0x3d6ef16bu);
    s2 = s2 * w - u2f(0x3d9d8795u);
    s2 = s2 * w - u2f(0x3de38e38u);
    s2 = s2 * w - u2f(0x3e4ccccdu);
    s2 = s2 * w;
    const float t = (s1 + s2) * x;
    if (id < 0) return x - t;
    const float r = atanhi[id] - ((t - atanlo[id]) - x);
    return ((int32_t)hx < 0) ? -r : r;
}
HDI float atan2f_glibc(float y, float x) {
    const uint32_t hx = f2u(x), hy = f2u(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const float tiny = u2f(0x0da24260u), pi_o_4 = u2f(0x3f490fdbu), pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu);
    const float pi_lo = u2f(0xb3bbbd2eu);
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;  // NaN
    if (hx == 0x3f800000u) return atanf_glibc(y);            // x = 1.0
    const uint32_t m = ((hy >> 31) & 1u) | ((hx >> 30) & 2u);  // 2 * sign(x) + sign(y)
    if (iy == 0u) {                                            // y = 0
        if (m < 2u) return y;
        return m == 2u ? pi + tiny : -pi - tiny;
    }
    if (ix == 0u) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x = 0
    if (ix == 0x7f800000u) {                                                     // x = +-inf
        if (iy == 0x7f800000u) {
            if (m == 0u) return pi_o_4 + tiny;
            if (m == 1u) return -pi_o_4 - tiny;
            if (m == 2u) return 3.0f * pi_o_4 + tiny;
            return -3.0f * pi_o_4 - tiny;
        }
        if (m == 0u) return 0.0f;
        if (m == 1u) return -0.0f;
        return m == 2u ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000u) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = ((int32_t)iy - (int32_t)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                   // |y/x| > 2^60
    else if ((int32_t)hx < 0 && k < -60) z = 0.0f;           // |y|/x < -2^60
    else z = atanf_glibc(fabsf(y / x));
    if (m == 0u) return z;
    if (m == 1u) return u2f(f2u(z) ^ 0x80000000u);
    if (m == 2u) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}
HDI float acosf_glibc(float x) {
    const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu;
    const float pi = u2f(0x40490fdau), pio2_hi = u2f(0x3fc90fdau), pio2_lo = u2f(0x33a22168u);
    if (ix == 0x3f800000u) {  // |x| == 1
        if ((int32_t)hx > 0) return 0.0f;
        return pi + 2.0f * pio2_lo;
    }
    if (ix > 0x3f800000u) return (x - x) / (x - x);  // |x| > 1: NaN
    auto p_of = [](float z) {
        float p = u2f(0x3811ef08u) * z + u2f(0x3a4f7f04u);
        p = p * z - u2f(0x3d241146u);
        p = p * z + u2f(0x3e4e0aa8u);
        p = p * z - u2f(0x3ea6b090u);
        p = p * z + u2f(0x3e2aaaabu);
        return p * z;
    };
    auto q_of = [](float z) {
        float q = u2f(0x3d9dc62eu) * z - u2f(0x3f303361u);
        q = q * z + u2f(0x4001572du);
        q = q * z - u2f(0x4019d139u);
        q = q * z;
        return q + 1.0f;
    };
    if (ix < 0x3f000000u) {  // |x| < 0.5
        if (ix <= 0x32800000u) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float r = p_of(z) / q_of(z);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if ((int32_t)hx < 0) {  // x < -0.5
        const float z = (1.0f + x) * 0.5f;
        const float s = sqrtf(z);
        const float r = p_of(z) / q_of(z);
        const float w = r * s - pio2_lo;
        const float t = w + s;
        return pi - (t + t);
    }
    const float z = (1.0f - x) * 0.5f;  // x > 0.5
    const float s = sqrtf(z);
    const float df = u2f(f2u(s) & 0xfffff000u);
    const float r = p_of(z) / q_of(z);
    const float c = (z - df * df) / (s + df);
    const float w = r * s + c;
    const float t = w + df;
    return t + t;
}

// ============================================================================
//  Scene as the kernel sees it
// ============================================================================
struct V3 {
    float x, y, z;
};
DI V3 v3(float x, float y, float z) { return {x, y, z}; }
DI V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DI V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DI V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
DI V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DI V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
// tuple.rs:44-46 for vectors (the w*w term is +0 and is dropped)
DI float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// tuple.rs:29-43 for vectors: sqrt(x^2 + y^2 + z^2 [+ 0]); norm divides
DI float mag3(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
DI V3 norm3(V3 a) {
    float m = mag3(a);
    return {a.x / m, a.y / m, a.z / m};
}
// ray.rs:43  -(n*2*dot(in,n) - in)
DI V3 reflect3(V3 in, V3 n) {
    float d = dot3(in, n);
    return -(n * 2.0f * d - in);
}

struct SceneHdr {
    uint32_t n_objects;
    int32_t light_kind;
    float li[3];      // light.intensity()
    float lpos[3];    // light.position(): point position or rectangle centre
    float corner[3];
    float uvec[3];    // per-cell
    float vvec[3];
    int32_t u_steps, v_steps;
    float cells_f;    // (u_steps*v_steps) as f32, rectangle_light.rs:87
    int32_t jitter_mode;
    float jitter_const;
    uint32_t jitter_seed;
    // camera
    uint32_t width, height;
    float half_w, half_h, pixel_size;
    float cam[12];        // rows 0..2 of transform_inverse
    float cam_origin[3];  // transform_inverse * point(0,0,0), camera.rs:70
    uint32_t has_patterns;  // some material carries a pattern (wave-uniform switch around the pattern code)
    uint32_t n_trav;        // entries in SceneSoA::trav; 0: the world is a flat object list
    float light_y_lo, light_y_hi;  // world-space y range of the area light's sample points, widened (light-cone culling)
    uint32_t uvrec_off;            // where SceneSoA::uvrec starts inside the scene buffer, in float4 units (host use)
    uint32_t all_cast;             // every object casts shadows: a shadow ray may stop at its first hit before the light
    // Triangle pre-culling (tree walks, see tri_precull): some triangle carries a box in SceneSoA::tbox; rays that start
    // within cull_r2 (squared) of cull_c are the ones the boxes' padding was computed for
    uint32_t has_tbox;
    float cull_c[3], cull_r2;
    float tri_guard;  // TRI_GUARD (the boxes' padding is derived from it on the host)
    // Small trees in the unrolled kernels (rtc_device.hip flatten): group boxes as gates.  gate_box[g] = min.xyz, max.xyz;
    // bit g of gate_mask[i]: object i sits inside group g and is only intersected by rays that hit g's box
    uint32_t has_scene_box;   // scene_box bounds everything a primary ray can hit (render_body)
    float scene_box[6];       // min.xyz, max.xyz, padded
    uint32_t internal_boxes;  // the traversal stream's boxes are all the library's own (build_flat_bvh): padded, not semantic
    uint32_t n_gates;
    float gate_box[8][6];
    uint32_t gate_mask[8];
    // Light-cone culling switches (light_cull_mask), set by the host from RTC_AMD_LIGHT_CULL / RTC_AMD_DARK (default: both
    // on): bit 0 -- cull at all; bit 1 -- the `dark` shortcut (every sample blocked on the far side of a casting sphere),
    // the one decision that asserts hits instead of removing tests; bit 2 -- shadow_fast (RTC_AMD_FAST_SHADOW), the
    // margin-guarded decision of a sample without normalising its ray; bit 3 -- block cones (RTC_AMD_CELL_CULL,
    // blocks_usable): 2 x 2 blocks of the light's cells called lit by the wave before any sample is drawn.  Off, every shadow ray is tested against every
    // object the exact way: the image and the ray counts must not change (tests/test_gpu_fullsize.py, whole frames).
    uint32_t cull_flags;
    uint32_t max_leaf_run;  // the longest run of consecutive leaf entries in SceneSoA::trav (host: lanes per pixel, for_each_leaf_shared)
    // RTC_JITTER_SEQUENCE (test/utils.rs:19-24 hardcoded_jitter; the batched rtc_intensity_at / rtc_point_on_light only): draw k of
    // a call is jitter_seq[k mod jitter_seq_len]
    uint32_t jitter_seq_len;
    float jitter_seq[RTC_JITTER_SEQUENCE_MAX];
    // block cones (intensity_at): half the diagonal of one cell of the area light (world units), padded by 1 % and by the rounding of a
    // sample's own position (8 u x the light's largest coordinate); 0: not available (jitter outside [0, 1], ...)
    float cell_hd;
};
constexpr uint32_t CULL_ENABLED = 1u, CULL_DARK = 2u, CULL_FAST_SHADOW = 4u, CULL_CELLS = 8u;
constexpr uint32_t RTC_MAX_GATES = 8;

// Structure-of-arrays scene records in HBM: 4 float4 of geometry (64 B) and
// 3 float4 of material (48 B) per object.  Every lane of a wave reads the same
// record, so these loads are wave-uniform (s_load -> SGPRs).  The geometry is
// split so that the common case -- a shadow ray against a scale+translate-only
// object -- touches a single 16-byte record (`geo`):
//   geo  = { m00, m11, m22, bits }          diagonal of t_inverse + kind/flags
//   off0 = { m01, m02, m03, min_y }         off-diagonals, translation column,
//   off1 = { m10, m12, m13, max_y }         cylinder bounds
//   off2 = { m20, m21, m23, 0 }
struct SceneSoA {
    const float4* __restrict__ geo;
    const float4* __restrict__ off0;
    const float4* __restrict__ off1;
    const float4* __restrict__ off2;
    const float4* __restrict__ mat_a;  // {r, g, b, ambient}
    const float4* __restrict__ mat_b;  // {diffuse, specular, shininess, reflective}
    const float4* __restrict__ mat_c;  // {transparency, refractive_index, 0, 0}
    // 5 records per object, read once per shaded hit and only if hdr.has_patterns:
    //   {a.rgb, kind}, {b.rgb -- or distance = b - a for gradient / sine_2d --, 0}, rows 0..2 of the pattern's t_inverse
    const float4* __restrict__ pat;
    // Worlds with GroupShapes (shape/group.rs): the object records are the tree's leaves in depth-first order and
    // `trav` is that traversal written out, TRAV_STRIDE (three) float4 per entry:
    //   group: { bounds.min.xyz, skip }, { bounds.max.xyz, slack }, unused   skip = entry index after the group's subtree,
    //          slack = 1e-3 * the box's largest |coordinate| (pruning margin, see for_each_object)
    //   leaf : { 0, 0, 0, object index }, { 0, 0, 0, -1 }, unused
    //   leaf, a triangle with a pre-culling box (tri_precull; world space, padded on the host -- rtc_device.hip triangle_box):
    //          { box.min.xyz, object index }, { box.max.xyz, -2 }, { unit normal.xyz, 1 if the next entry is such a leaf too }
    const float4* __restrict__ trav;
    // Triangles (shape/triangle.rs:9-17), 3 records per object, read only for RTC_TRIANGLE objects:
    //   { p1.xyz, normal.x }, { e1.xyz, normal.y }, { e2.xyz, normal.z }
    const float4* __restrict__ tri;
    // Light-cone culling: the area light's four corners (around the parallelogram) in each object's own space,
    // 3 records per object: { c0.xyz, c1.x }, { c1.yz, c2.xy }, { c2.z, c3.xyz }.  Approximate values (margins apply).
    const float4* __restrict__ lcorn;
    // { m03, m13, m23, 0 }: the translation column again, so that scale+translate-only objects need two records
    // (geo + trn = 32 B) instead of four
    const float4* __restrict__ trn;
    // { centre.xyz, radius } of a world-space sphere around the object; radius = +inf for unbounded objects.  Approximate
    // (margins apply): used to leave non-casters out of shadow tests when they lie behind every caster (light_cull_mask)
    const float4* __restrict__ bsph;
    // Texture maps (pattern/uv.rs).  A TEXTURE_MAP / CUBE_MAP pattern's second record is { mapping, first UV pattern,
    // 0, 0 }; each UV pattern is six records in `uvrec`:
    //   { kind, width, height, first texel }, { image width, image height, 0, 0 }, then five RGB colours packed
    //   ({ c0.rgb, c1.r }, { c1.gb, c2.rg }, { c2.b, c3.rgb }, { c4.rgb, 0 }); `texels`: every UVImage's Canvas, RGB f32.
    const float4* __restrict__ uvrec;
    const float* __restrict__ texels;
};
constexpr float TRAV_LEAF_TAG = -1.0f;  // e1.w of a leaf entry; a group's e1.w is its pruning slack (>= 0)
constexpr float TRAV_BOXED_LEAF_TAG = -2.0f;
constexpr uint32_t TRAV_STRIDE = 3;
enum : uint32_t {
    SHAPE_KIND_MASK = 0xffu,
    SHAPE_NONE = 0xffu,      // padding record: never intersects (arrays are padded to a multiple of 8)
    SHAPE_CASTS = 1u << 8,   // BaseShape.casts_shadow
    SHAPE_CLOSED = 1u << 9,  // Cylinder.closed
    SHAPE_DIAG = 1u << 10,   // t_inverse has no off-diagonal 3x3 terms (scale + translate only)
    // The object sits in a GroupShape whose box does not contain it (group.rs:15 caches the box and never updates it, so a
    // child added later can lie outside): rays that would hit it may be turned away at the group.  Shortcuts that
    // ASSERT a hit on the object (light_cull_mask's `dark`) must stand down; those that only remove tests are unaffected.
    SHAPE_LOOSE = 1u << 11,
    // SHAPE_DIAG with all three diagonal entries equal: a uniformly scaled (and translated) object.  Only the margin-guarded
    // fast decision of a shadow sample (shadow_fast) looks at it -- approximate arithmetic may factor the scale out.
    SHAPE_UNIFORM = 1u << 12,
};

// ---- scene specialisation (hiprtc compile only) --------------------------------------------------
// -DRTC_SPEC_LIST=b0,b1,...  the per-object `bits` words of THIS scene (kind | casts | closed | diag),
// -DRTC_SPEC_NOBJ=n, -DRTC_SPEC_SIMPLE=0|1, -DRTC_SPEC_LIGHT_KIND=k, -DRTC_SPEC_JITTER=m, -DRTC_SPEC_PATTERNS=0|1.
// Inside the fully unrolled object loops `i` is a constant, so spec_bits(i, ...) folds and every
// `kind == ...` / `bits & flag` test below disappears at compile time.  Values (matrices, materials,
// light geometry, camera) stay run-time data: one specialisation serves every scene of that shape.
#ifdef RTC_SPEC_LIST
#ifdef RTC_SPEC_UNIFORM_BITS
// any number of objects, all with the same kind / flags word (C5: 64 scale+translate spheres)
DI uint32_t spec_bits(uint32_t, uint32_t) { return RTC_SPEC_UNIFORM_BITS; }
#elif defined(RTC_SPEC_RUNTIME_BITS)
// a tree of mixed objects: only the light / jitter / pattern switches are compile-time
DI uint32_t spec_bits(uint32_t, uint32_t runtime_bits) { return runtime_bits; }
#else
DI uint32_t spec_bits(uint32_t i, uint32_t) {
    constexpr uint32_t table[8] = {RTC_SPEC_LIST};
    return table[i & 7u];
}
#endif
DI int32_t spec_light_kind(int32_t) { return RTC_SPEC_LIGHT_KIND; }
DI int32_t spec_jitter_mode(int32_t) { return RTC_SPEC_JITTER; }
DI bool spec_has_patterns(uint32_t) { return RTC_SPEC_PATTERNS != 0; }
#ifdef RTC_SPEC_TBOX
DI bool spec_has_tbox(uint32_t) { return RTC_SPEC_TBOX != 0; }  // the scene has pre-culling boxes (tree walks)
DI bool spec_has_nodes(uint32_t) { return RTC_SPEC_TBOX >= 2; }  // ... and nodes of the library's own among its entries (has_tbox == 2)
#else
DI bool spec_has_tbox(uint32_t h) { return h != 0; }
DI bool spec_has_nodes(uint32_t h) { return h >= 2u; }
#endif
#ifdef RTC_SPEC_GATES
DI bool spec_has_gates(uint32_t) { return RTC_SPEC_GATES != 0; }
#else
DI bool spec_has_gates(uint32_t n) { return n != 0; }
#endif
#else
DI uint32_t spec_bits(uint32_t, uint32_t runtime_bits) { return runtime_bits; }
DI int32_t spec_light_kind(int32_t k) { return k; }
DI int32_t spec_jitter_mode(int32_t m) { return m; }
DI bool spec_has_patterns(uint32_t h) { return h != 0; }
DI bool spec_has_gates(uint32_t n) { return n != 0; }
DI bool spec_has_tbox(uint32_t h) { return h != 0; }
DI bool spec_has_nodes(uint32_t h) { return h >= 2u; }
#endif

// Scene facts a scene-compiled kernel knows (hiprtc, -DRTC_SPEC_ANY_REFL / _ANY_REFR / _REG_LEVELS): whether any material
// reflects / transmits at all -- a scene without either has no recursion and carries no frame code, a mirror-only scene
// no refraction half -- and how many levels of the recursion stack live in REGISTERS.
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_ANY_REFL)
constexpr bool ANY_REFL = RTC_SPEC_ANY_REFL != 0, ANY_REFR = RTC_SPEC_ANY_REFR != 0;
#else
constexpr bool ANY_REFL = true, ANY_REFR = true;
#endif
// ... and whether any material has a specular highlight that needs powf (phong)
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_ANY_SPECULAR)
constexpr bool ANY_SPECULAR = RTC_SPEC_ANY_SPECULAR != 0;
#else
constexpr bool ANY_SPECULAR = true;
#endif
// ... and which components of the area light's cell vectors are exactly zero (bits 0..2: u_vec.xyz, 3..5: v_vec.xyz; an
// axis-aligned light has four).  point_on_light (rectangle_light.rs:60-66) adds `vec * (cell + jitter)` per component: a
// product with an exact zero is +-0 for the finite factors the host vouches for (hashed jitter, or a finite constant), and
// adding +-0 changes at most the sign of a zero sum -- the terms are dropped like the other exact zeros of this file.
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_LIGHT_ZEROS)
constexpr uint32_t LIGHT_ZEROS = RTC_SPEC_LIGHT_ZEROS;
#else
constexpr uint32_t LIGHT_ZEROS = 0u;
#endif
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_SELECT)
constexpr bool SELECT_RECORDS = RTC_SPEC_SELECT != 0;
#else
constexpr bool SELECT_RECORDS = false;
#endif
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_STASH)
constexpr bool USE_STASH = RTC_SPEC_STASH != 0;
#else
constexpr bool USE_STASH = true;
#endif
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_LDS_FRAMES)
constexpr int LDS_FRAME_LEVELS = RTC_SPEC_LDS_FRAMES;
#else
constexpr int LDS_FRAME_LEVELS = 0;
#endif
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_REG_LEVELS)
constexpr int STACK_REG_LEVELS = RTC_SPEC_REG_LEVELS;
#else
constexpr int STACK_REG_LEVELS = 0;
#endif
// Levels of the recursion's frame stack (color_at): RTC_STACK_DEPTH_BASE in every kernel but the ones the host compiles
// for a deeper recursion (-DRTC_SPEC_MAX_DEPTH=16 / 32 / ...: rtc_device.hip deep_kernel; camera.rs:76 takes any depth).
// A DEEP kernel differs in bookkeeping only: the path code no longer fits the bits the shallow kernels pack it into, so it
// is parked in a word of its own and every frame keeps its caller's path (a 32-bit code wraps beyond 31 levels, exactly
// as the oracle's u32 does; restoring the saved word is then the only way back to the caller's).
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_MAX_DEPTH)
#define RTC_STACK_DEPTH RTC_SPEC_MAX_DEPTH
#else
#define RTC_STACK_DEPTH RTC_STACK_DEPTH_BASE
#endif
#define RTC_DEEP_STACK (RTC_STACK_DEPTH > RTC_STACK_DEPTH_BASE)
static_assert(RTC_STACK_DEPTH >= RTC_STACK_DEPTH_BASE && RTC_STACK_DEPTH <= 255, "RTC_SPEC_MAX_DEPTH");
static_assert(!RTC_DEEP_STACK || LDS_FRAME_LEVELS == 0, "deep kernels keep their frames in scratch");
static_assert(STACK_REG_LEVELS >= 0 && STACK_REG_LEVELS <= RTC_STACK_DEPTH, "RTC_SPEC_REG_LEVELS");

constexpr float PLANE_EPS = 1.1920929e-7f * 10000.0f;  // plane.rs:49  f32::EPSILON * 10000.0
constexpr float SELF_EPS = 1.1920929e-7f * 10000.0f;   // world.rs:210
constexpr float CLOSE_TO_ZERO = 0.000001f;             // cylinder.rs:82

struct Obj {
    float4 geo, off0, off1, off2, trn;
    uint32_t bits;
    DI float min_y() const { return off0.w; }
    DI float max_y() const { return off1.w; }
};
DI Obj load_obj(const SceneSoA& S, uint32_t i) {
    Obj o;
    o.geo = S.geo[i];
    o.off0 = S.off0[i];
    o.off1 = S.off1[i];
    o.off2 = S.off2[i];
    o.trn = S.trn[i];
    o.bits = __float_as_uint(o.geo.w);
    return o;
}
// Record i of an object LOOP (i is the same in every lane).  Two things keep the compiler from fetching it
// with scalar loads by itself: inside divergent control flow it carries a run-time loop counter in a VGPR,
// and it cannot prove that the read-only scene is not clobbered by the kernel's own stores.  On C5 (64
// spheres, generic loop) that meant per-lane vector loads and 64 % of wave-cycles waiting on memory.
// readfirstlane states the uniformity, the constant address space states the immutability: the record
// then comes through the scalar cache into SGPRs.  In the unrolled kernels i is a compile-time constant
// and the bits may be too (spec_bits).
typedef float RawF4 __attribute__((ext_vector_type(4)));
typedef const RawF4 __attribute__((address_space(4))) * ConstF4Ptr;
DI float4 load_uniform(const float4* base, uint32_t i) {
    const uint32_t iu = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
    const RawF4 r = ((ConstF4Ptr)(unsigned long)base)[iu];
    return make_float4(r.x, r.y, r.z, r.w);
}
// RUNTIME_INDEX: the generic any-count loop.  (The unrolled kernels index with constants; forcing the
// constant address space there makes the loads freely hoistable and costs SGPR spills: C3 4.06 -> 4.37 ms.)
template <bool RUNTIME_INDEX>
DI Obj load_obj_static(const SceneSoA& S, uint32_t i) {
    Obj o;
    if constexpr (RUNTIME_INDEX) {
        o.geo = load_uniform(S.geo, i);
        o.bits = spec_bits(i, __float_as_uint(o.geo.w));
#ifdef RTC_SPEC_UNIFORM_BITS
        // the kind / flags word is a compile-time constant: scale+translate-only objects that keep no bounds
        // need just two records (geo + trn = 32 B)
        constexpr uint32_t kind = RTC_SPEC_UNIFORM_BITS & SHAPE_KIND_MASK;
        if constexpr ((RTC_SPEC_UNIFORM_BITS & SHAPE_DIAG) && kind != RTC_CYLINDER && kind != RTC_CONE) {
            o.trn = load_uniform(S.trn, i);
            o.off0 = o.off1 = o.off2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            return o;
        }
#endif
        o.off0 = load_uniform(S.off0, i);
        o.off1 = load_uniform(S.off1, i);
        o.off2 = load_uniform(S.off2, i);
        o.trn = make_float4(o.off0.z, o.off1.z, o.off2.z, load_uniform(S.trn, i).w);  // (.w: light_cull_mask's E; unused elsewhere, and then not loaded)
        return o;
    } else {
        o = load_obj(S, i);
    }
    o.bits = spec_bits(i, __float_as_uint(o.geo.w));
    return o;
}

// shape.rs:57-70 + ray.rs:26-31 for an affine inverse: o' = M*o (w = 1), d' = M*d (w = 0).
// When the 3x3 part is diagonal the products with the (exactly zero) off-diagonal
// entries are +-0 and adding them changes nothing, so they are skipped.
DI V3 obj_point(const Obj& b, V3 p) {
    if (b.bits & SHAPE_DIAG) return {b.geo.x * p.x + b.trn.x, b.geo.y * p.y + b.trn.y, b.geo.z * p.z + b.trn.z};
    return {b.geo.x * p.x + b.off0.x * p.y + b.off0.y * p.z + b.off0.z,
            b.off1.x * p.x + b.geo.y * p.y + b.off1.y * p.z + b.off1.z,
            b.off2.x * p.x + b.off2.y * p.y + b.geo.z * p.z + b.off2.z};
}
DI V3 obj_vector(const Obj& b, V3 v) {
    if (b.bits & SHAPE_DIAG) return {b.geo.x * v.x, b.geo.y * v.y, b.geo.z * v.z};
    return {b.geo.x * v.x + b.off0.x * v.y + b.off0.y * v.z, b.off1.x * v.x + b.geo.y * v.y + b.off1.y * v.z,
            b.off2.x * v.x + b.off2.y * v.y + b.geo.z * v.z};
}
// normal_to_world (shape.rs:72-146): transpose(t_inverse) * n, w := 0, normalise
DI V3 obj_normal_to_world(const Obj& b, V3 n) {
    if (b.bits & SHAPE_DIAG) return norm3(v3(b.geo.x * n.x, b.geo.y * n.y, b.geo.z * n.z));  // the other terms are +-0 (see obj_point)
    V3 w = {b.geo.x * n.x + b.off1.x * n.y + b.off2.x * n.z, b.off0.x * n.x + b.geo.y * n.y + b.off2.y * n.z,
            b.off0.y * n.x + b.off1.y * n.y + b.geo.z * n.z};
    return norm3(w);
}

// Calls f(t) for every intersection the reference's local_intersect would
// push, in push order.  o, d: object-space ray.
// `c` is the origin-only term of the sphere / cylinder quadratic (quadratic_c): rays that share
// an origin (area-light shadow samples) compute it once.
DI float quadratic_c(uint32_t kind, V3 o) {
    if (kind == RTC_SPHERE) return (o.x * o.x + o.y * o.y + o.z * o.z) - 1.0f;  // sphere.rs:56
    if (kind == RTC_CYLINDER) return o.x * o.x + o.z * o.z - 1.0f;              // cylinder.rs:95
    if (kind == RTC_CONE) return o.x * o.x - o.y * o.y + o.z * o.z;             // cone.rs:143-146
    return 0.0f;
}
// HITS_ONLY: the caller only looks at distances >= 0 (hit selection, shadow tests), so an
// intersection that is provably negative need not be evaluated; refraction_indices() needs the
// negative ones too and passes false.
// tuple.rs:47-55
DI V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// triangle.rs:45-68 (Moeller-Trumbore as written there); t0..t2: the object's three `tri` records
template <class F>
DI void triangle_intersect(float4 t0, float4 t1, float4 t2, V3 o, V3 d, F&& f) {
    const V3 p1 = v3(t0.x, t0.y, t0.z), e1 = v3(t1.x, t1.y, t1.z), e2 = v3(t2.x, t2.y, t2.z);
    V3 dir_cross_e2 = cross3(d, e2);
    float determinant = dot3(e1, dir_cross_e2);
    if (!(fabsf(determinant) < 0.0000001f)) {
        float fi = 1.0f / determinant;
        V3 p1_to_origin = o - p1;
        float u = fi * dot3(p1_to_origin, dir_cross_e2);
        if (!(u < 0.0f || u > 1.0f)) {
            V3 origin_cross_e1 = cross3(p1_to_origin, e1);
            float v = fi * dot3(d, origin_cross_e1);
            if (!(v < 0.0f || (u + v) > 1.0f)) f(fi * dot3(e2, origin_cross_e1));
        }
    }
}

// `tri` / `i`: the triangle records and this object's index (wave-uniform); only read for RTC_TRIANGLE.
// LANE_IDX: `i` differs from lane to lane (the leaf-sharing tree walk): the triangle records come through vector loads.
template <bool HITS_ONLY, bool LANE_IDX = false, class F>
DI void local_intersect_c(uint32_t bits, float min_y, float max_y, const float4* __restrict__ tri, uint32_t i, V3 o, V3 d,
                          float c, F&& f) {
    const uint32_t kind = bits & SHAPE_KIND_MASK;
    if (kind == RTC_SPHERE) {  // sphere.rs:47-70
        float a = d.x * d.x + d.y * d.y + d.z * d.z;
        float b = 2.0f * (d.x * o.x + d.y * o.y + d.z * o.z);
        float disc = b * b - 4.0f * a * c;
        if (!(disc < 0.0f)) {
            float two_a = 2.0f * a;
            float sq = sqrtf(disc);
#ifndef RTC_NO_LAZY_ROOT
            if constexpr (HITS_ONLY) {
                // t1 <= t2 (two_a > 0, sq >= 0), and a caller that only looks at distances >= 0 takes the smaller of those:
                // t1 if it is >= 0 -- which it is exactly when its numerator is (-0 included) -- and t2 otherwise.  One
                // IEEE division instead of two; the quotient evaluated is the very one the reference would have used.
                const float n1 = -b - sq;
                f((n1 >= 0.0f ? n1 : -b + sq) / two_a);
            } else
#endif
            {
                f((-b - sq) / two_a);
                f((-b + sq) / two_a);
            }
        }
    } else if (kind == RTC_PLANE) {  // plane.rs:45-56
        if (!(fabsf(d.y) < PLANE_EPS)) {
            // -o.y / d.y is strictly negative when o.y and d.y have the same sign and the quotient cannot
            // underflow to -0 (|o.y| >= 2^-100, |d.y| <= 2^20): the ray leaves the plane behind.
            const bool behind = HITS_ONLY && fabsf(d.y) <= 0x1p20f &&
                                ((o.y >= 0x1p-100f && d.y > 0.0f) || (o.y <= -0x1p-100f && d.y < 0.0f));
            if (!behind) f(-o.y / d.y);
        }
    } else if (kind == RTC_CUBE) {  // cube.rs:55-63, 90-129; reciprocals from Ray::new (ray.rs:16)
        float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
        float x0 = (-1.0f - o.x) * ix, x1 = (1.0f - o.x) * ix;
        float tmin = fminf(x0, x1), tmax = fmaxf(x0, x1);
        float y0 = (-1.0f - o.y) * iy, y1 = (1.0f - o.y) * iy;
        tmin = fmaxf(tmin, fminf(y0, y1));
        tmax = fminf(tmax, fmaxf(y0, y1));
        float z0 = (-1.0f - o.z) * iz, z1 = (1.0f - o.z) * iz;
        tmin = fmaxf(tmin, fminf(z0, z1));
        tmax = fminf(tmax, fmaxf(z0, z1));
        if (tmax >= fmaxf(0.0f, tmin)) {
            f(tmin);
            f(tmax);
        }
    } else if (kind == RTC_CYLINDER) {  // cylinder.rs:52-59, 84-151
        int pushed = 0;
        float two_a = 2.0f * (d.x * d.x + d.z * d.z);
        if (!(fabsf(two_a) < CLOSE_TO_ZERO)) {
            float b = 2.0f * (o.x * d.x + o.z * d.z);
            float disc = b * b - 2.0f * two_a * c;
            if (!(disc < 0.0f)) {
                float sq = sqrtf(disc);
                float d1 = (-b - sq) / two_a;
                float d2 = (-b + sq) / two_a;
                if (d1 > d2) {
                    float t = d1;
                    d1 = d2;
                    d2 = t;
                }
                float y1 = o.y + d1 * d.y;
                if (min_y < y1 && y1 < max_y) {
                    f(d1);
                    pushed++;
                }
                float y2 = o.y + d2 * d.y;
                if (min_y < y2 && y2 < max_y) {
                    f(d2);
                    pushed++;
                }
            }
        }
        if (pushed < 2 && (bits & SHAPE_CLOSED)) {  // intersect_caps, cylinder.rs:132-151
            float t = (min_y - o.y) / d.y;
            float cx = o.x + t * d.x, cz = o.z + t * d.z;
            if ((cx * cx + cz * cz) <= 1.0f + CLOSE_TO_ZERO) f(t);
            t = (max_y - o.y) / d.y;
            cx = o.x + t * d.x;
            cz = o.z + t * d.z;
            if ((cx * cx + cz * cz) <= 1.0f + CLOSE_TO_ZERO) f(t);
        }
    } else if (kind == RTC_TRIANGLE) {
        const float4 t0 = LANE_IDX ? tri[3u * i] : load_uniform(tri, 3u * i), t1 = LANE_IDX ? tri[3u * i + 1u] : load_uniform(tri, 3u * i + 1u),
                     t2 = LANE_IDX ? tri[3u * i + 2u] : load_uniform(tri, 3u * i + 2u);
        triangle_intersect(t0, t1, t2, o, d, f);
    } else if (kind == RTC_CONE) {  // cone.rs:52-57: sides (:89-141), then -- always -- caps (:156-175)
        float two_a = 2.0f * (d.x * d.x - d.y * d.y + d.z * d.z);
        float b = 2.0f * (o.x * d.x - o.y * d.y + o.z * d.z);
        if (fabsf(two_a) < CLOSE_TO_ZERO) {
            // ray parallel to one half of the cone: a single root, pushed without a y-range test (:99-107)
            if (!(fabsf(b) < CLOSE_TO_ZERO)) f(-c / (2.0f * b));
        } else {
            float disc = b * b - 2.0f * two_a * c;
            if (!(disc < 0.0f)) {
                float sq = sqrtf(disc);
                float d1 = (-b - sq) / two_a;
                float d2 = (-b + sq) / two_a;
                if (d1 > d2) {
                    float t = d1;
                    d1 = d2;
                    d2 = t;
                }
                float y1 = o.y + d1 * d.y;
                if (min_y < y1 && y1 < max_y) f(d1);
                float y2 = o.y + d2 * d.y;
                if (min_y < y2 && y2 < max_y) f(d2);
            }
        }
        if (bits & SHAPE_CLOSED) {  // check_cap (:148-154) compares x^2 + z^2 with |y|, as written there
            float t = (min_y - o.y) / d.y;
            float cx = o.x + t * d.x, cz = o.z + t * d.z;
            if ((cx * cx + cz * cz) <= fabsf(min_y) + CLOSE_TO_ZERO) f(t);
            t = (max_y - o.y) / d.y;
            cx = o.x + t * d.x;
            cz = o.z + t * d.z;
            if ((cx * cx + cz * cz) <= fabsf(max_y) + CLOSE_TO_ZERO) f(t);
        }
    }
}

template <bool HITS_ONLY, bool LANE_IDX = false, class F>
DI void local_intersect(uint32_t bits, float min_y, float max_y, const float4* __restrict__ tri, uint32_t i, V3 o, V3 d, F&& f) {
    local_intersect_c<HITS_ONLY, LANE_IDX>(bits, min_y, max_y, tri, i, o, d, quadratic_c(bits & SHAPE_KIND_MASK, o), f);
}

// local_norm_at (sphere.rs:71-73, plane.rs:57-59, cube.rs:66-80, cylinder.rs:62-72, cone.rs:60-73)
DI V3 local_normal(uint32_t kind, float min_y, float max_y, const float4* __restrict__ tri, V3 p) {
    if (kind == RTC_SPHERE) return p;
    if (kind == RTC_TRIANGLE) return v3(tri[0].w, tri[1].w, tri[2].w);  // triangle.rs:70-72: the stored normal
    if (kind == RTC_PLANE) return v3(0.0f, 1.0f, 0.0f);
    if (kind == RTC_CUBE) {
        float xa = fabsf(p.x), ya = fabsf(p.y), za = fabsf(p.z);
        float max_c = fmaxf(xa, fmaxf(ya, za));
        if (xa == max_c) return v3(p.x, 0.0f, 0.0f);
        if (ya == max_c) return v3(0.0f, p.y, 0.0f);
        return v3(0.0f, 0.0f, p.z);
    }
    float dist_square = p.x * p.x + p.z * p.z;
    if (dist_square < 1.0f) {
        if (p.y >= max_y - CLOSE_TO_ZERO) return v3(0.0f, 1.0f, 0.0f);
        if (p.y <= min_y + CLOSE_TO_ZERO) return v3(0.0f, -1.0f, 0.0f);
    }
    if (kind == RTC_CONE) {  // the cap test above uses radius 1 for the cone too, as cone.rs:61-68 does
        float y = sqrtf(dist_square);
        if (p.y > 0.0f) y = -y;
        return v3(p.x, y, p.z);
    }
    return v3(p.x, 0.0f, p.z);
}

struct Hit {
    float t;
    int obj;  // -1: none
};

// per-lane work counters (reduced per workgroup at kernel end)
struct Counters {
    uint32_t rays;    // World::intersect evaluations
    uint32_t shaded;  // bits 0..11: shade_hit evaluations (<= 2^depth per pixel); bits 12..31: of `rays`, the shadow
                      // rays answered by the light-cone cull without testing any object (<= 2^depth * cells)
    // Sample-parallel rendering (render_body): 2^share_log2 adjacent lanes trace the SAME pixel and split the area light's
    // cells between them.  Wave-uniform.  Everything but the shadow rays is then computed identically by all of a
    // pixel's lanes and counted by the first of them only.
    // Compiled in only where it is used (-DRTC_SPEC_SHARE=1, hiprtc): with SHARE_LANES false everything below folds to
    // "one lane per pixel" (the 4096^2 headline kernel is 3 % slower with the general form).
    uint32_t share_log2_;
    // Kernels compiled for a deep recursion (RTC_DEEP_STACK: up to 255 levels) can shade more than 4 095 points in one pixel (a glass
    // ball in a hall of mirrors doubles its rays per level) and answer more than 2^20 shadow rays by the cull: the packed word
    // would carry from one field into the other.  They count the culled rays in a word of their own; the shallow kernels
    // (<= 8 levels: <= 255 shade points, <= 255 x 128 cells) keep the packing, which costs them no register.
    uint32_t culled_deep_;
    DI void add_culled(uint32_t n) {
#if RTC_DEEP_STACK
        culled_deep_ += n;
#else
        shaded += n << 12;
#endif
    }
    DI uint32_t shaded_count() const { return RTC_DEEP_STACK ? shaded : (shaded & 0xfffu); }
    DI uint32_t culled_count() const { return RTC_DEEP_STACK ? culled_deep_ : (shaded >> 12); }
#if defined(RTC_SPEC_SHARE) && RTC_SPEC_SHARE
    static constexpr bool SHARE_LANES = true;
#else
    static constexpr bool SHARE_LANES = false;
#endif
    DI uint32_t share_log2() const { return SHARE_LANES ? share_log2_ : 0u; }
    DI uint32_t share_mask() const { return (1u << share_log2()) - 1u; }
    DI uint32_t sub() const { return threadIdx.x & share_mask(); }           // this lane's place among its pixel's lanes
    DI uint32_t lead() const { return sub() == 0u ? 1u : 0u; }               // 1 on the lane that counts the shared work
    DI uint32_t my_cells(uint32_t cells) const { return (cells + share_mask() - sub()) >> share_log2(); }  // cells c with c % 2^s == sub
};
constexpr uint32_t CNT_SHADED_MASK = 0xfffu, CNT_CULLED_SHIFT = 12u;

#ifdef RTC_DEBUG_STEPS  // development (tools/walk_steps.py): per lane, what the tree walks did -- the frame holds counts instead of colours
DI uint32_t* dbg_steps() {
    __shared__ uint32_t c[256 * 4];
    return c + threadIdx.x * 4;
}
#define RTC_DBG_STEP(j) (dbg_steps()[j]++)
#else
#define RTC_DBG_STEP(j) ((void)0)
#endif
// cube.rs:90-129 aabb_intersection(..).is_some() for a world-space box: `inv` are the reciprocals Ray::new keeps
// (ray.rs:16).  fminf / fmaxf return the non-NaN operand, as Rust's f32::min / max do.
DI bool aabb_hit(V3 o, V3 inv, float4 mn, float4 mx, float& tmin) {
    float x0 = (mn.x - o.x) * inv.x, x1 = (mx.x - o.x) * inv.x;
    tmin = fminf(x0, x1);
    float tmax = fmaxf(x0, x1);
    float y0 = (mn.y - o.y) * inv.y, y1 = (mx.y - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(y0, y1));
    tmax = fminf(tmax, fmaxf(y0, y1));
    float z0 = (mn.z - o.z) * inv.z, z1 = (mx.z - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(z0, z1));
    tmax = fminf(tmax, fmaxf(z0, z1));
    return tmax >= fmaxf(0.0f, tmin);
}
// The world-space ray as GroupShape::local_intersect sees it (group.rs:115-133: groups do not transform the ray).
// `limit` (tree walks only): the caller has no use for intersections beyond this ray parameter -- the nearest hit so
// far, or the distance to the light for a shadow ray -- so a group whose box the ray ENTERS later than that need not
// be opened (see for_each_object); -inf: this lane needs nothing more at all.  +inf switches the pruning off.
struct WorldRay {
    V3 o, inv;
    float limit;
    V3 d;
    float guard;  // TRI_GUARD * |d|, or +inf when this ray may not pre-cull triangles (see tri_precull)
    uint32_t closed;  // unrolled kernels with gates: the groups whose box this ray misses (SceneHdr::gate_mask)
};
constexpr float TRI_GUARD = 0.05f;  // sine of the smallest ray-to-plane angle at which a triangle may be pre-culled
template <int NOBJ>
DI WorldRay world_ray(const SceneHdr& H, V3 o, V3 d) {
    if constexpr (NOBJ < 0) {
        const V3 c = o - v3(H.cull_c[0], H.cull_c[1], H.cull_c[2]);
        const bool near = H.has_tbox != 0u && dot3(c, c) <= H.cull_r2;  // NaN: false
        // ray.rs:16 keeps IEEE reciprocals, and a GroupShape's box must be tested with them; the library's own boxes are
        // padded by 10 %, which a 1-ulp v_rcp_f32 cannot cross (three divisions are a quarter of a ray that hits nothing)
        const V3 inv = H.internal_boxes ? v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z))
                                        : v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        return {o, inv, RTC_INF, d, near ? H.tri_guard * sqrtf(dot3(d, d)) : RTC_INF, 0u};
    } else {
        WorldRay wr = {o, o, RTC_INF, o, RTC_INF, 0u};  // unused but for `closed`
        if (NOBJ > 0 && spec_has_gates(H.n_gates)) {
            const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // ray.rs:16
            for (uint32_t g = 0; g < H.n_gates; g++) {
                float tmin;
                const float* b = H.gate_box[g];
                if (!aabb_hit(o, inv, make_float4(b[0], b[1], b[2], 0.0f), make_float4(b[3], b[4], b[5], 0.0f), tmin)) wr.closed |= 1u << g;
            }
        }
        return wr;
    }
}
// a WorldRay for loops that run over ALL objects whatever their groups (per-shade-point preparation)
DI WorldRay ungated_ray(V3 o) { return {o, o, RTC_INF, o, RTC_INF, 0u}; }

// Tree walks meet long runs of triangle leaves (the reference's divide() keeps every child that straddles the split
// plane in the parent group: a closed mesh leaves a ring of them at every level), and the exact test costs ~150
// operations per leaf (general inverse transform of the ray + Moeller-Trumbore with an IEEE division).  Most of those
// rays pass nowhere near the triangle.  This is a ~25-operation test that says so in a way the exact f32 evaluation
// is GUARANTEED to agree with -- i.e. it only skips triangles for which triangle.rs:45-68, evaluated in f32 on this
// ray, returns no intersection:
//   * the ray's line (no t >= 0 restriction: the n1/n2 walk wants intersections behind the origin as well) misses
//     the triangle's world-space bounding box, PADDED on the host by a distance P derived below, and
//   * the ray is not within asin(TRI_GUARD) of the triangle's plane (|d.n| >= TRI_GUARD |d|), and
//   * the ray starts inside the ball the padding was derived for (WorldRay::guard is +inf otherwise).
// Why that is safe (object-space quantities; D = |origin - p1|, s = longest edge, theta = angle between ray and plane,
// alpha = smallest interior angle, q = where the ray's line meets the plane, eps = 2^-24):  the three rejections of
// the reference's test compare N_u = p1o.(d x e2), N_v = d.(p1o x e1) and their sum with 0 and det = e1.(d x e2).
// Geometrically |N_u| = |d||e2| m_u sin(theta) with m_u the in-plane distance from q to the edge line through p1
// along e2 (likewise for the other two edges), and the rounding of each N is at most 8 eps D |d||e|, that of det at
// most 7 eps |e1||d||e2|, so a rejection that holds in exact arithmetic can only be lost to rounding if
//       m_j sin(theta) <= eps rho (8 D + 10 s),    rho = max(1, (|e1| + |e2|) / |e2 - e1|).
// A line that misses the padded box passes at least P from the triangle, so q lies at least P outside it in the
// plane, which puts it at least P sin(alpha / 2) beyond one of the three edge lines.  With sin(theta) >= TRI_GUARD /
// kappa (kappa: condition number of the object's transform, which distorts the angle) the host chooses
//       P = 4 eps rho kappa (8 D + 10 s) / (TRI_GUARD sin(alpha / 2))          (4: safety factor; it also covers the
// rounding of the ray's own transformation into object space, which moves q by ~4 eps D / sin(theta)),
// with D bounded through the ball: every ray that pre-culls starts within it, every triangle lies within it.  The sign of
// det is safe as well (|det| = |d| |e1 x e2| sin(theta) >> 7 eps |e1||d||e2| under the same conditions; triangles too
// thin for that get no box).  NaNs fail the comparisons and keep the triangle.
DI bool tri_precull(const WorldRay& wr, float4 b0, float4 b1, float4 b2) {
    const float dn = wr.d.x * b2.x + wr.d.y * b2.y + wr.d.z * b2.z;
    if (!(fabsf(dn) >= wr.guard)) return false;
    const float x0 = (b0.x - wr.o.x) * wr.inv.x, x1 = (b1.x - wr.o.x) * wr.inv.x;
    const float y0 = (b0.y - wr.o.y) * wr.inv.y, y1 = (b1.y - wr.o.y) * wr.inv.y;
    const float z0 = (b0.z - wr.o.z) * wr.inv.z, z1 = (b1.z - wr.o.z) * wr.inv.z;
    const float lo = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float hi = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    return lo > hi;
}
// A NODE of the library's own over part of a long run of boxed triangles (host: cluster_leaf_runs; e2.w > 0 marks it among
// the group entries): its box is the hull of its triangles' padded boxes and its third record a vector a' with
// |d.a'| >= TRI_GUARD |d|  =>  |d.n| >= TRI_GUARD |d| for every triangle normal n under it (the axis of a cone around
// those normals, scaled).  A ray may pass the node by exactly when tri_precull would have skipped each of its triangles
// one by one -- the line misses the hull, hence every box, and no triangle is near-parallel to it -- or, under the same
// guard, when it enters the hull beyond what the caller still wants (see for_each_object).  A ray for which the guard
// fails opens the node whatever its box says: the triangles under it take their own tests, as without the node.
DI bool node_precull(const WorldRay& wr, float4 b0, float4 b1, float4 b2) {
    const float dn = wr.d.x * b2.x + wr.d.y * b2.y + wr.d.z * b2.z;
    if (!(fabsf(dn) >= wr.guard)) return false;
    const float x0 = (b0.x - wr.o.x) * wr.inv.x, x1 = (b1.x - wr.o.x) * wr.inv.x;
    const float y0 = (b0.y - wr.o.y) * wr.inv.y, y1 = (b1.y - wr.o.y) * wr.inv.y;
    const float z0 = (b0.z - wr.o.z) * wr.inv.z, z1 = (b1.z - wr.o.z) * wr.inv.z;
    const float lo = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float hi = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    return lo > hi || lo > wr.limit + (b1.w + 1e-4f * fabsf(wr.limit));
}

#ifdef RTC_NO_SHARED_WALK  // development: a pixel's lanes all walk every leaf, as before for_each_leaf_shared existed
constexpr bool SHARED_WALK = false;
#else
constexpr bool SHARED_WALK = true;
#endif
#ifdef RTC_NO_SHARED_HIT
constexpr bool SHARED_WALK_HIT = false;
#else
constexpr bool SHARED_WALK_HIT = true;
#endif
#ifdef RTC_NO_SHARED_N12
constexpr bool SHARED_WALK_N12 = false;
#else
constexpr bool SHARED_WALK_N12 = true;
#endif
template <bool B>
struct BoolConstant {
    static constexpr bool value = B;
};
// e2.w of a leaf entry, written by the host (mark_leaf_runs): (consecutive leaf entries of the same group from this one
// on, itself included) * 8 + (4 if they are a MESH run: boxed triangles of consecutive object indices that all share one
// transform and one kind / flags word) + (boxed triangle leaves right after this one, at most 3) -- a small whole number,
// exact as a float
DI uint32_t trav_more(float w) { return (uint32_t)w & 3u; }
DI uint32_t trav_run(float w) { return (uint32_t)w >> 3; }
DI bool trav_mesh_run(float w) { return ((uint32_t)w & 4u) != 0u; }

// The tree walk of a kernel whose pixels are traced by 2^s adjacent lanes each (Counters::SHARE_LANES), for a ray that
// all of a pixel's lanes share -- the pixel's primary and secondary rays, a point light's shadow ray.  One lane per
// pixel walks a mesh as ONE chain of dependent tests: the reference's divide() keeps every child that straddles the
// split plane in the parent group (group.rs:46-73), a ring of dozens to hundreds of direct triangle children at every
// level, and a frame's time is that of its slowest wave -- here_be_dragons at 1000 x 400 kept 3 % of the chip busy.
// Here the lanes of a pixel SPLIT every run of consecutive leaves between them: lane `sub` takes the run's entries
// sub, sub + 2^s, ...  (fetched per lane: vector loads), group entries are tested by all of them alike, and after
// each run `after_run()` lets the caller pool what the lanes found (nearest_hit: the minimum over the pixel's lanes
// becomes every lane's pruning limit).  Each leaf is still visited exactly once per pixel, by exactly one lane; what
// the callers compute from the visits is order-independent in the tree kernels (ties go by object index), so the
// split cannot change an answer.
// `on_tri(i, t)`: a triangle of a mesh run has its one intersection at distance t (what body(i) would have found).  The
// triangles of a parsed mesh share their group's baked transform (group.rs:39-44), so the ray is taken into their object
// space once per run -- the same operations on the same matrix as per leaf, hence the same bits -- from records that
// arrive through scalar loads; per leaf a lane then needs its own entry (box + normal, for tri_precull) and, for the few
// leaves that pass, the three triangle records.  The next entry is fetched while the current one is being tested.
template <class F, class T, class G>
DI void for_each_leaf_shared(const SceneHdr& H, const SceneSoA& S, WorldRay& wr, const Counters& cnt, F&& body, T&& on_tri, G&& after_run) {
    uint32_t resume = 0;  // this lane ignores entries below `resume`
    const uint32_t stride = 1u << cnt.share_log2(), sub = cnt.sub();
    for (uint32_t k = 0; k < H.n_trav;) {
        const ConstF4Ptr ep = ((ConstF4Ptr)(unsigned long)S.trav) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(TRAV_STRIDE * k));
        const RawF4 r0 = ep[0], r1 = ep[1];
        const bool active = k >= resume && wr.limit > -RTC_INF;
        RTC_DBG_STEP(3);
        if (!(r1.w < 0.0f)) {  // a group (see for_each_object): every lane of the pixel takes the same decision
            if (active) RTC_DBG_STEP(0);
            const uint32_t skip = __float_as_uint(r0.w);
            bool inside = false;
            if (spec_has_nodes(H.has_tbox) && ep[2].w > 0.0f) {  // wave-uniform: one of the library's own nodes (node_precull)
                const RawF4 r2 = ep[2];
                if (active) {
                    inside = !node_precull(wr, make_float4(r0.x, r0.y, r0.z, r0.w), make_float4(r1.x, r1.y, r1.z, r1.w),
                                           make_float4(r2.x, r2.y, r2.z, r2.w));
                    if (!inside) resume = skip;
                }
            } else if (active) {
                float tmin;
                inside = aabb_hit(wr.o, wr.inv, make_float4(r0.x, r0.y, r0.z, r0.w), make_float4(r1.x, r1.y, r1.z, r1.w), tmin);
                if (inside && tmin > wr.limit + (r1.w + 1e-4f * fabsf(wr.limit)) + (2e-3f * tmin + ep[2].x * tmin * tmin)) inside = false;  // (see for_each_object)
                if (!inside) resume = skip;
            }
            k = __any(inside) ? k + 1u : skip;
        } else {
            const float w = ep[2].w;
            const uint32_t run = trav_run(w);  // wave-uniform, >= 1
            if (__any(active)) {
                const float4* ent = S.trav + (size_t)TRAV_STRIDE * k;
                if (spec_has_tbox(H.has_tbox) && trav_mesh_run(w)) {  // wave-uniform
                    const uint32_t obj0 = __float_as_uint(r0.w);  // the run's first object: all of them share its transform
                    const Obj ob = load_obj_static<true>(S, obj0);
                    const V3 po = obj_point(ob, wr.o), pd = obj_vector(ob, wr.d);
                    uint32_t m = sub;
                    float4 b0 = ent[0], b1 = ent[1], b2 = ent[2];
                    if (active && m < run) b0 = ent[TRAV_STRIDE * m], b1 = ent[TRAV_STRIDE * m + 1u], b2 = ent[TRAV_STRIDE * m + 2u];
                    while (active && m < run) {
                        const uint32_t mn = m + stride;
                        float4 n0 = b0, n1 = b1, n2 = b2;
                        if (mn < run) n0 = ent[TRAV_STRIDE * mn], n1 = ent[TRAV_STRIDE * mn + 1u], n2 = ent[TRAV_STRIDE * mn + 2u];
                        RTC_DBG_STEP(1);
                        if (!tri_precull(wr, b0, b1, b2)) {
                            RTC_DBG_STEP(2);
                            const uint32_t i = __float_as_uint(b0.w);
                            const float4* tr = S.tri + (size_t)3u * i;
                            triangle_intersect(tr[0], tr[1], tr[2], po, pd, [&](float t) { on_tri(i, t); });
                        }
                        b0 = n0, b1 = n1, b2 = n2;
                        m = mn;
                    }
                } else {
                    for (uint32_t m = sub; active && m < run; m += stride) {
                        const float4 b0 = ent[TRAV_STRIDE * m], b1 = ent[TRAV_STRIDE * m + 1u];  // this lane's own entry
                        bool visit = true;
                        RTC_DBG_STEP(1);
                        if (spec_has_tbox(H.has_tbox) && b1.w == TRAV_BOXED_LEAF_TAG) visit = !tri_precull(wr, b0, b1, ent[TRAV_STRIDE * m + 2u]);
                        if (visit) RTC_DBG_STEP(2);
                        if (visit) body(__float_as_uint(b0.w));
                    }
                }
                after_run();
            }
            k += run;
        }
    }
}

// Applies `body(i)` to every object the reference's World::intersect would reach.  NOBJ > 0: the scene has at
// most NOBJ objects and the loop is fully unrolled (record loads become loop-invariant SGPR values, per-object
// state can live in registers); NOBJ == 0: any count; NOBJ < 0: the world contains GroupShapes -- the depth-first
// entry list is walked as a PACKET: the entry index k is wave-uniform (records still arrive through scalar
// loads), a lane that misses a group's bounding box sits out until k reaches that group's skip index, and the
// wave jumps over a subtree only when no lane is inside it.  Every lane therefore visits exactly the leaves, in
// exactly the order, the reference's recursive child loop visits for its ray.
template <int NOBJ, class F>
DI void for_each_object(const SceneHdr& H, const SceneSoA& S, WorldRay& wr, F&& body) {
    if constexpr (NOBJ < 0) {
        // ERROR_BUDGET.md B6.  Pruning by wr.limit cannot change what the caller computes: everything a group contributes lies
        // inside its box up to rounding (bounds are hulls of transformed corners; a hit satisfies the shape's equation under
        // the rounded inverse transform: E1, a relative ~1e-6 of the coordinates), the computed entry parameter tmin is within
        // 3 ulp of the exact one, and the group is only left closed when tmin exceeds the limit by a margin of four parts:
        //   * 4e-3 of the box's largest |coordinate| (its `slack`, stored with the entry; infinite or NaN bounds never prune):
        //     E1 -- also as amplified by a grazing hit, sqrt(2 r E1) <= slack while r <= 20 |coordinate|, which a box around
        //     the object implies -- and the part of E2 below that is proportional to the object's own size;
        //   * 1e-4 of the limit (the limit is itself a computed distance);
        //   * 2e-3 tmin: E2's cap -- a sphere / cylinder / cone quadratic near a tangent reports its root up to
        //     sqrt(16 u) |o| / |d| = 1e-3 D early, whatever the object's size (|o| / |d|, object space, is the distance D when
        //     the ray heads for the object, and only then is there a root at all);
        //   * q tmin^2, q = 1e-6 / (the smallest r below the group: 1 / |inverse 3x3|): E2 away from the tangent, u D^2 / 2 r -- two
        //     units for a sphere of 0.25 seen from 4 000, fuzz world 84 -- sixteen times over.
        uint32_t resume = 0;  // this lane ignores entries below `resume`
        for (uint32_t k = 0; k < H.n_trav;) {
            // one address computation for the entry's three records (the walk executes as many scalar as vector instructions)
            const ConstF4Ptr ep = ((ConstF4Ptr)(unsigned long)S.trav) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(TRAV_STRIDE * k));
            const RawF4 r0 = ep[0], r1 = ep[1], r2 = ep[2];
            const float4 e0 = make_float4(r0.x, r0.y, r0.z, r0.w), e1 = make_float4(r1.x, r1.y, r1.z, r1.w),
                         e2 = make_float4(r2.x, r2.y, r2.z, r2.w);
            const bool active = k >= resume && wr.limit > -RTC_INF;
            RTC_DBG_STEP(3);
            if (!(e1.w < 0.0f)) {  // a group: e1.w is its slack (>= 0, inf or NaN); leaves carry -1
                if (active) RTC_DBG_STEP(0);
                const uint32_t skip = __float_as_uint(e0.w);
                bool inside = false;
                if (spec_has_nodes(H.has_tbox) && e2.w > 0.0f) {  // wave-uniform: one of the library's own nodes (node_precull)
                    if (active) {
                        inside = !node_precull(wr, e0, e1, e2);
                        if (!inside) resume = skip;
                    }
                } else if (active) {
                    float tmin;
                    inside = aabb_hit(wr.o, wr.inv, e0, e1, tmin);
                    if (inside && tmin > wr.limit + (e1.w + 1e-4f * fabsf(wr.limit)) + (2e-3f * tmin + e2.x * tmin * tmin)) inside = false;  // nothing of interest in there
                    if (!inside) resume = skip;
                }
                k = __any(inside) ? k + 1u : skip;
            } else if (spec_has_tbox(H.has_tbox) && e1.w == TRAV_BOXED_LEAF_TAG && trav_more(e2.w) >= 3u) {  // wave-uniform: three more such leaves follow
                float4 f[3][3];
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const RawF4 q = ep[3 * (j + 1) + r];
                        f[j][r] = make_float4(q.x, q.y, q.z, q.w);
                    }
                const bool c0 = tri_precull(wr, e0, e1, e2), c1 = tri_precull(wr, f[0][0], f[0][1], f[0][2]),
                           c2 = tri_precull(wr, f[1][0], f[1][1], f[1][2]), c3 = tri_precull(wr, f[2][0], f[2][1], f[2][2]);
#ifdef RTC_DEBUG_STEPS
                RTC_DBG_STEP(3), RTC_DBG_STEP(3), RTC_DBG_STEP(3);
                if (active) { RTC_DBG_STEP(1); if (!c0) RTC_DBG_STEP(2); }
                if (k + 1u >= resume && wr.limit > -RTC_INF) { RTC_DBG_STEP(1); if (!c1) RTC_DBG_STEP(2); }
                if (k + 2u >= resume && wr.limit > -RTC_INF) { RTC_DBG_STEP(1); if (!c2) RTC_DBG_STEP(2); }
                if (k + 3u >= resume && wr.limit > -RTC_INF) { RTC_DBG_STEP(1); if (!c3) RTC_DBG_STEP(2); }
#endif
                if (active && !c0) body(__float_as_uint(e0.w));
                if (k + 1u >= resume && wr.limit > -RTC_INF && !c1) body(__float_as_uint(f[0][0].w));
                if (k + 2u >= resume && wr.limit > -RTC_INF && !c2) body(__float_as_uint(f[1][0].w));
                if (k + 3u >= resume && wr.limit > -RTC_INF && !c3) body(__float_as_uint(f[2][0].w));
                k += 4u;
            } else if (spec_has_tbox(H.has_tbox) && e1.w == TRAV_BOXED_LEAF_TAG && trav_more(e2.w) > 0u) {  // wave-uniform
                // a boxed triangle followed by another one (e2.w, set on the host): both pre-culling tests at once.  In the
                // long runs of such leaves the walk is one wave's chain of dependent instructions; two independent chains
                // interleave.  Visiting order, and what each visit sees of the other's result (wr.limit), are unchanged.
                const RawF4 q0 = ep[3], q1 = ep[4], q2 = ep[5];
                const float4 f0 = make_float4(q0.x, q0.y, q0.z, q0.w), f1 = make_float4(q1.x, q1.y, q1.z, q1.w),
                             f2 = make_float4(q2.x, q2.y, q2.z, q2.w);
                const bool cull0 = tri_precull(wr, e0, e1, e2), cull1 = tri_precull(wr, f0, f1, f2);
#ifdef RTC_DEBUG_STEPS
                RTC_DBG_STEP(3);
                if (active) { RTC_DBG_STEP(1); if (!cull0) RTC_DBG_STEP(2); }
                if (k + 1u >= resume && wr.limit > -RTC_INF) { RTC_DBG_STEP(1); if (!cull1) RTC_DBG_STEP(2); }
#endif
                if (active && !cull0) body(__float_as_uint(e0.w));
                if (k + 1u >= resume && wr.limit > -RTC_INF && !cull1) body(__float_as_uint(f0.w));
                k += 2u;
            } else {
                bool visit = active;
                if (active) RTC_DBG_STEP(1);
                if (spec_has_tbox(H.has_tbox) && e1.w == TRAV_BOXED_LEAF_TAG) visit = visit && !tri_precull(wr, e0, e1, e2);  // wave-uniform branch
                if (visit) RTC_DBG_STEP(2);
                if (visit) body(__float_as_uint(e0.w));
                k++;
            }
        }
    } else if constexpr (NOBJ > 0) {
        // no `i < n_objects` guard: the record arrays are padded with SHAPE_NONE entries, so every
        // load is unconditional and can be hoisted / issued ahead of the arithmetic that needs it
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)NOBJ; i++) {
            if (spec_has_gates(H.n_gates) && (H.gate_mask[i & 7u] & wr.closed) != 0u) continue;  // inside a group this ray does not open
            body(i);
        }
    } else {
        // any object count: the record arrays are padded to a multiple of 8 with SHAPE_NONE entries (skipped by
        // the bodies), so the loop advances four records at a time and their loads are issued together
#ifdef RTC_SPEC_UNIFORM_BITS
        // uniform specialisation: the bodies no longer look at the padding records' kind, so stop at n_objects
        uint32_t i = 0;
        for (; i + 3u < H.n_objects; i += 4) {
            body(i);
            body(i + 1);
            body(i + 2);
            body(i + 3);
        }
        for (; i < H.n_objects; i++) body(i);
#else
        const uint32_t padded = (H.n_objects + 7u) & ~7u;
        for (uint32_t i = 0; i < padded; i += 4) {
            body(i);
            body(i + 1);
            body(i + 2);
            body(i + 3);
        }
#endif
    }
}

// World::intersect + Intersection::hit (world.rs:52-60, intersection.rs:30-35)
// without materialising or sorting the list: the hit is the first entry, in
// (object order, push order), of the minimum among distances >= 0 -- which is
// what a stable sort followed by a first-minimum scan selects.
// Tree walks only: `t_max` -- hits beyond it do not matter to the caller (a shadow ray's distance to the light);
// `any_hit` -- the caller only asks whether there is a hit below t_max (a shadow ray in a world where every object
// casts shadows: the nearest hit is then a caster whichever it is), so a lane stops at its first such hit.
// SHARED (tree kernels compiled for several lanes per pixel): every lane of the pixel traces this very ray; they split
// the leaves between them (for_each_leaf_shared) and pool their nearest hits, so that all of them return the pixel's.
template <int NOBJ, bool SHARED = false>
DI Hit nearest_hit(const SceneHdr& H, const SceneSoA& S, V3 o, V3 d, const Counters& cnt, float t_max = RTC_INF, bool any_hit = false,
                   uint32_t skip = 0u) {
    Hit best = {0.0f, -1};
    WorldRay wr = world_ray<NOBJ>(H, o, d);
    wr.limit = t_max;
    // (a block of one lane per pixel -- RenderArgs::tiles mixes both kinds in one launch -- keeps the packet walk below,
    // whose records arrive through scalar loads: wave-uniform choice)
    if constexpr (NOBJ < 0 && SHARED && Counters::SHARE_LANES && SHARED_WALK && SHARED_WALK_HIT)
      if (cnt.share_log2() != 0u) {
        auto offer = [&](uint32_t i, float t) {
            const bool better = best.obj < 0 || t < best.t || (t == best.t && (int)i < best.obj);
            if (t >= 0.0f && better) {
                best.t = t;
                best.obj = (int)i;
                wr.limit = (any_hit && t < t_max) ? -RTC_INF : fminf(wr.limit, t);
            }
        };
        for_each_leaf_shared(
            H, S, wr, cnt,
            [&](uint32_t i) {
                if (i < 32u && ((skip >> i) & 1u)) return;  // light-cone culled for this shade point
                Obj ob = load_obj(S, i);                    // per-lane index: vector loads
                ob.bits = spec_bits(i, ob.bits);
                V3 po = obj_point(ob, o);
                V3 pd = obj_vector(ob, d);
                local_intersect<true, true>(ob.bits, ob.min_y(), ob.max_y(), S.tri, i, po, pd, [&](float t) { offer(i, t); });
            },
            [&](uint32_t i, float t) {
                if (i < 32u && ((skip >> i) & 1u)) return;
                offer(i, t);
            },
            [&]() {  // the minimum over the pixel's lanes, in the order the reference's stable sort gives: (t, object index)
                for (uint32_t m = 1u; m < (1u << cnt.share_log2()); m <<= 1) {
                    const float ot = __shfl_xor(best.t, (int)m, 64);
                    const int oo = __shfl_xor(best.obj, (int)m, 64);
                    if (oo >= 0 && (best.obj < 0 || ot < best.t || (ot == best.t && oo < best.obj))) {
                        best.t = ot;
                        best.obj = oo;
                    }
                }
                if (best.obj >= 0) wr.limit = (any_hit && best.t < t_max) ? -RTC_INF : fminf(wr.limit, best.t);
            });
        return best;
    }
    for_each_object<NOBJ>(H, S, wr, [&](uint32_t i) {
        if (i < 32u && ((skip >> i) & 1u)) return;  // light-cone culled for this shade point (wave-uniform)
        Obj ob = load_obj_static<NOBJ <= 0>(S, i);
        if ((ob.bits & SHAPE_KIND_MASK) == SHAPE_NONE) return;  // padding record (wave-uniform)
        V3 po = obj_point(ob, o);
        V3 pd = obj_vector(ob, d);
        local_intersect<true>(ob.bits, ob.min_y(), ob.max_y(), S.tri, i, po, pd, [&](float t) {
            // first-minimum in list order: the flat loops visit objects in that order; a tree walk may not (the
            // library's own bounding-volume hierarchy over a flat world reorders visits), so there ties go to the lower index
            bool better = best.obj < 0 || t < best.t;
            if constexpr (NOBJ < 0) better = better || (t == best.t && (int)i < best.obj);
            if (t >= 0.0f && better) {
                best.t = t;
                best.obj = (int)i;
                if constexpr (NOBJ < 0) wr.limit = (any_hit && t < t_max) ? -RTC_INF : fminf(wr.limit, t);
            }
        });
    });
    return best;
}

// nearest_hit for the rays of color_at in a TREE kernel, gathering on the way what precompute_values' n1 / n2 walk
// (world.rs:235-263; refraction_indices below has the argument) needs from the intersections BEHIND the origin: the two
// innermost odd-parity containers, and whether the hit object is one.  Round 2 walked the tree a second time for them (with the
// limit at t = 0) whenever the hit was transparent -- a third of a glass mesh's walks.  One walk serves both: a group the
// container walk opens (its box holds the origin: entry parameter <= 0 within the pruning slack) the nearest-hit walk opens as
// well (its limit is the nearest hit's t >= 0, so its condition is the weaker one), leaves and nodes are pre-culled by the same
// line tests, and an object in a box the ray enters at t > 0 has no intersection behind the origin -- unless its group is
// "loose", and those are never pruned by either walk.  Every object is visited once: all its intersections are formed
// (HITS_ONLY = false: the same expressions, so the hit distances are the ones nearest_hit returns), the negative ones counted,
// the smallest non-negative one offered as the hit together with the object's parity.
template <int NOBJ>
DI Hit nearest_hit_and_containers(const SceneHdr& H, const SceneSoA& S, V3 o, V3 d, const Counters& cnt, float& t1, int& c1, float& t2, int& c2,
                                  bool& hit_inside) {
    static_assert(NOBJ < 0, "tree kernels");
    Hit best = {0.0f, -1};
    bool best_inside = false;
    t1 = t2 = 0.0f;
    c1 = c2 = -1;
    WorldRay wr = world_ray<NOBJ>(H, o, d);
    wr.limit = RTC_INF;
    auto after = [&](float ta, int ia, float tb, int ib) { return ta > tb || (ta == tb && ia > ib); };  // does (ta, ia) sort after (tb, ib)?
    auto offer_container = [&](float t, int c) {
        if (c1 < 0 || after(t, c, t1, c1)) {
            t2 = t1, c2 = c1;
            t1 = t, c1 = c;
        } else if (c2 < 0 || after(t, c, t2, c2)) {
            t2 = t, c2 = c;
        }
    };
    auto offer_hit = [&](uint32_t i, float t, bool inside) {  // t >= 0
        if (best.obj < 0 || t < best.t || (t == best.t && (int)i < best.obj)) {
            best.t = t, best.obj = (int)i, best_inside = inside;
            wr.limit = fminf(wr.limit, t);
        }
    };
    auto per_object = [&](uint32_t i, const Obj& ob, auto lane_idx) {
        const V3 po = obj_point(ob, o), pd = obj_vector(ob, d);
        int negatives = 0;
        float tmax = 0.0f, tpos = 0.0f;
        bool has = false;
        local_intersect<false, decltype(lane_idx)::value>(ob.bits, ob.min_y(), ob.max_y(), S.tri, i, po, pd, [&](float t) {
            if (t < 0.0f) {
                if (negatives == 0 || t > tmax) tmax = t;
                negatives++;
            } else if (t >= 0.0f && (!has || t < tpos)) {  // (NaN: neither)
                tpos = t, has = true;
            }
        });
        const bool inside = (negatives & 1) != 0;
        if (inside) offer_container(tmax, (int)i);
        if (has) offer_hit(i, tpos, inside);
    };
    using LaneIdx = BoolConstant<true>;
    using UniformIdx = BoolConstant<false>;
    if constexpr (Counters::SHARE_LANES && SHARED_WALK && SHARED_WALK_HIT && SHARED_WALK_N12)
      if (cnt.share_log2() != 0u) {  // the pixel's lanes split the leaf runs (for_each_leaf_shared) and pool what they find
        for_each_leaf_shared(
            H, S, wr, cnt,
            [&](uint32_t i) {
                Obj ob = load_obj(S, i);
                ob.bits = spec_bits(i, ob.bits);
                per_object(i, ob, LaneIdx());
            },
            [&](uint32_t i, float t) {  // a triangle of a mesh run: its one intersection
                if (t < 0.0f) offer_container(t, (int)i);
                else if (t >= 0.0f) offer_hit(i, t, false);
            },
            [&]() {  // after a run: the pixel's nearest hit so far, in (t, object index) order, with its parity
                for (uint32_t m = 1u; m < (1u << cnt.share_log2()); m <<= 1) {
                    const float ot = __shfl_xor(best.t, (int)m, 64);
                    const int oo = __shfl_xor(best.obj, (int)m, 64);
                    const int oi = __shfl_xor((int)best_inside, (int)m, 64);
                    if (oo >= 0 && (best.obj < 0 || ot < best.t || (ot == best.t && oo < best.obj))) best.t = ot, best.obj = oo, best_inside = oi != 0;
                }
                if (best.obj >= 0) wr.limit = fminf(wr.limit, best.t);
            });
        for (uint32_t m = 1u; m < (1u << cnt.share_log2()); m <<= 1) {  // every object was examined by exactly one lane: no duplicates
            const float pt1 = __shfl_xor(t1, (int)m, 64), pt2 = __shfl_xor(t2, (int)m, 64);
            const int pc1 = __shfl_xor(c1, (int)m, 64), pc2 = __shfl_xor(c2, (int)m, 64);
            if (pc1 >= 0) offer_container(pt1, pc1);
            if (pc2 >= 0) offer_container(pt2, pc2);
        }
        hit_inside = best_inside;
        return best;
      }
    for_each_object<NOBJ>(H, S, wr, [&](uint32_t i) {
        const Obj ob = load_obj_static<true>(S, i);
        if ((ob.bits & SHAPE_KIND_MASK) == SHAPE_NONE) return;  // padding record (wave-uniform)
        per_object(i, ob, UniformIdx());
    });
    hit_inside = best_inside;
    return best;
}

// world.rs:104-119
// SHARED: see nearest_hit (the ray is counted once per pixel)
template <int NOBJ, bool SHARED = false>
DI bool is_shadowed(const SceneHdr& H, const SceneSoA& S, V3 light_position, V3 p, Counters& cnt, uint32_t skip = 0u) {
    V3 v = light_position - p;
    float distance = mag3(v);
    V3 direction = norm3(v);
    cnt.rays += SHARED ? cnt.lead() : 1u;
    Hit h = nearest_hit<NOBJ, SHARED>(H, S, p, direction, cnt, distance, H.all_cast != 0u, skip);
    if (h.obj < 0) return false;
    bool casts;
    if constexpr (NOBJ > 0) {  // the flags of a few objects as one wave-uniform mask (a constant in a scene-compiled kernel) instead of a gather
        uint32_t casters = 0u;
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)(NOBJ > 0 ? NOBJ : 1); i++)
            if (spec_bits(i, __float_as_uint(S.geo[i].w)) & SHAPE_CASTS) casters |= 1u << i;
        casts = ((casters >> h.obj) & 1u) != 0u;
    } else {
        casts = (__float_as_uint(S.geo[h.obj].w) & SHAPE_CASTS) != 0;
    }
    return casts && h.t < distance;
}

// ---- exact sqrt / divide without the range handling ---------------------------------------------
// hipcc expands IEEE-correct sqrtf(x) and a/b into v_sqrt_f32 / v_rcp_f32 plus FMA correction steps
// wrapped in range handling (v_div_scale / v_div_fmas / v_div_fixup, denormal pre-scaling, class
// checks).  When the operands are comfortably inside the normal range that wrapping is the identity,
// so the bare correction sequence below returns the SAME bits -- it is the compiler's own sequence
// with the no-op steps removed (checked bit-for-bit on the device by rtc_selftest_fastmath).  Three
// quotients by one denominator also share the reciprocal refinement.
DI float sqrt_core(float x) {  // == sqrtf(x) for x in [2^-80, 2^40]
    float s = __builtin_amdgcn_sqrtf(x);
    float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    float r_dn = __builtin_fmaf(-s_dn, s, x);
    float r_up = __builtin_fmaf(-s_up, s, x);
    float r = (r_dn <= 0.0f) ? s_dn : s;
    return (r_up > 0.0f) ? s_up : r;
}
struct RcpCore {  // refined reciprocal of b, shared by every quotient n / b
    float b, r1;
    DI explicit RcpCore(float b_) : b(b_) {
        float r0 = __builtin_amdgcn_rcpf(b);
        float e0 = __builtin_fmaf(-b, r0, 1.0f);
        r1 = __builtin_fmaf(e0, r0, r0);
    }
    DI float div(float n) const {  // == n / b when b in [2^-40, 2^20] and n == 0 or |n| in [2^-100, b]
        float q0 = n * r1;
        float e1 = __builtin_fmaf(-b, q0, n);
        float q1 = __builtin_fmaf(e1, r1, q0);
        float e2 = __builtin_fmaf(-b, q1, n);
        return __builtin_fmaf(e2, r1, q1);
    }
};
// distance = |v| and dir = v / |v| exactly as mag3 / norm3 compute them.  The cheap sequence is used
// when EVERY active lane of the wave is in its validity range (wave-uniform branch); otherwise the
// whole wave takes the general path -- both produce identical bits, so the choice is invisible.
DI bool normalize_in_core_range(V3 v, float sum) {
    // v_k == 0, or |v_k| >= 2^-100: as integers, (|bits| - 1) wraps zero to 0xffffffff
    uint32_t ax = (__float_as_uint(v.x) & 0x7fffffffu) - 1u, ay = (__float_as_uint(v.y) & 0x7fffffffu) - 1u,
             az = (__float_as_uint(v.z) & 0x7fffffffu) - 1u;
    uint32_t lo = ax < ay ? ax : ay;
    lo = lo < az ? lo : az;
    return sum >= 0x1p-80f && sum <= 0x1p40f && lo >= 0x0d800000u - 1u;  // 0x0d800000 = 2^-100
}
DI void normalize_exact(V3 v, float& distance, V3& dir) {
    float sum = v.x * v.x + v.y * v.y + v.z * v.z;
    if (__all(normalize_in_core_range(v, sum))) {
        distance = sqrt_core(sum);
        RcpCore rc(distance);
        dir = v3(rc.div(v.x), rc.div(v.y), rc.div(v.z));
    } else {
        distance = sqrtf(sum);
        dir = v3(v.x / distance, v.y / distance, v.z / distance);
    }
}

// ---- area-light shadow samples: many rays from one point -------------------------------------
// Per shade point and object, the parts of World::is_shadowed that depend on the ray ORIGIN only:
// the object-space origin (shape.rs:57-61) and the quadratic's constant term.
struct ShadowPre {
    V3 o;
    float c;
};
template <int NOBJ>
DI void shadow_prepare(const SceneHdr& H, const SceneSoA& S, V3 p, ShadowPre* pre) {
    static_assert(NOBJ > 0, "flat unrolled scenes only");
    WorldRay wr = ungated_ray(p);
    for_each_object<NOBJ>(H, S, wr, [&](uint32_t i) {
        Obj ob = load_obj_static<NOBJ <= 0>(S, i);
        pre[i].o = obj_point(ob, p);
        pre[i].c = quadratic_c(ob.bits & SHAPE_KIND_MASK, pre[i].o);
    });
}
// World::is_shadowed (world.rs:104-119) for light sample `lp`, given shadow_prepare's output.
// Per object only its 16-byte `geo` record is fetched unless it is rotated/sheared or a cylinder.
//
// The reference answers "is the NEAREST hit a shadow caster closer than the light?".  That is a
// pure function of the hit list, so it is evaluated in two passes without changing the answer:
//   1. nearest hit among shadow CASTERS only, (t_c, i_c) in (distance, object order) order;
//      if there is none, or t_c >= distance, the point is lit whatever the non-casters do;
//   2. otherwise a non-caster hides that caster iff it has a hit t_n >= 0 that sorts before
//      (t_c, i_c).  Only then are non-casters (the soft_shadows lampshade) intersected at all.
// SIMPLE: every object is scale+translate-only, none is a cylinder or cone and no material has a pattern
// (decided on the host), so the
// loop-invariant uniform working set is 4 SGPRs per object and stays resident across the sample loop.
constexpr uint32_t LIGHT_CULL_ALL_CASTERS = 0x80000000u;
// squared inflation of the bounding spheres in light_cull_mask: 3 % in radius.  A culled ray then passes at least 0.03
// radius outside: discriminant <= -4a * 0.06 against a rounding error of ~1e-7 * 4a * |o|^2 <= 4e-3 a within the
// 100-radii limit -- a factor 60.  (10 % was used first; 3 % leaves 7 % fewer shade points of C3 to be tested.)
constexpr float LIGHT_CULL_INFLATE2 = 1.0609f;
// `skip`: wave-uniform mask from light_cull_mask() -- objects that provably have no intersection at t >= 0 with
// any ray from this shade point to the light, and non-casters that provably lie behind every caster left: neither
// can change either pass.
template <int NOBJ, bool SIMPLE>
DI bool is_shadowed_pre(const SceneHdr& H, const SceneSoA& S, const ShadowPre* pre, V3 lp, V3 p, Counters& cnt,
                        uint32_t skip) {
    V3 v = lp - p;
    float distance;
    V3 dir;
    normalize_exact(v, distance, dir);  // == mag3(v), norm3(v)
    cnt.rays++;
    WorldRay wr = world_ray<NOBJ>(H, p, dir);  // unused: NOBJ > 0 here
    auto object_ts = [&](uint32_t i, const float4 g, uint32_t bits, auto&& f) {
        V3 pd;
        if (SIMPLE || (bits & SHAPE_DIAG)) {
            pd = v3(g.x * dir.x, g.y * dir.y, g.z * dir.z);
        } else {
            const float4 a = S.off0[i], b = S.off1[i], c = S.off2[i];
            pd = v3(g.x * dir.x + a.x * dir.y + a.y * dir.z, b.x * dir.x + g.y * dir.y + b.y * dir.z,
                    c.x * dir.x + c.y * dir.y + g.z * dir.z);
        }
        float mn = 0.0f, mx = 0.0f;
        if (!SIMPLE && ((bits & SHAPE_KIND_MASK) == RTC_CYLINDER || (bits & SHAPE_KIND_MASK) == RTC_CONE)) {
            mn = S.off0[i].w;
            mx = S.off1[i].w;
        }
        local_intersect_c<true>(bits, mn, mx, S.tri, i, pre[i].o, pd, pre[i].c, f);
    };
    // pass 1: shadow casters
    bool found = false;
    float t_c = 0.0f;
    uint32_t i_c = 0;
    for_each_object<NOBJ>(H, S, wr, [&](uint32_t i) {
        const float4 g = S.geo[i];
        const uint32_t bits = spec_bits(i, __float_as_uint(g.w));
        if ((bits & SHAPE_KIND_MASK) == SHAPE_NONE || !(bits & SHAPE_CASTS)) return;  // wave-uniform
        if (skip & (1u << i)) return;                                                 // wave-uniform
        object_ts(i, g, bits, [&](float t) {
            if (t >= 0.0f && (!found || t < t_c)) {
                t_c = t;
                i_c = i;
                found = true;
            }
        });
    });
    bool shadowed = found && t_c < distance;
    // pass 2: can a non-caster hide that caster?
    if (shadowed) {
        for_each_object<NOBJ>(H, S, wr, [&](uint32_t i) {
            const float4 g = S.geo[i];
            const uint32_t bits = spec_bits(i, __float_as_uint(g.w));
            if ((bits & SHAPE_KIND_MASK) == SHAPE_NONE || (bits & SHAPE_CASTS)) return;
            if (skip & (1u << i)) return;
            object_ts(i, g, bits, [&](float t) {
                if (t >= 0.0f && (t < t_c || (t == t_c && i < i_c))) shadowed = false;
            });
        });
    }
    return shadowed;
}

// ---- light-cone culling -----------------------------------------------------------------------------
// All shadow rays of one shade point run from p into the light's parallelogram.  Per object, once per shade
// point, decide CONSERVATIVELY whether any such ray can reach the object at t >= 0; if none can -- for every
// active lane of the wave -- the object is left out of all (100) is_shadowed evaluations of this shade point.
// This is to the shadow rays what a bounding-volume hierarchy is to primary rays: it removes tests whose outcome
// is known, and therefore cannot change the image -- PROVIDED a culled object really yields no hit in the exact
// f32 evaluation.  That is ensured by margins far above the evaluation's rounding error (DESIGN.md "Light-cone
// culling"): the test works in the object's own space (where the exact quadratic is evaluated) with the unit
// sphere / cube blown up by 3 % in radius (LIGHT_CULL_INFLATE2), the cone of directions widened by 1e-3 in cosine, and it is only
// trusted when the shade point is within 100 radii (beyond that the quadratic's cancellation error grows and
// the object is simply kept).  Everything here is approximate arithmetic; NaNs fail every comparison and so
// keep the object.  Spheres, cubes, bounded cylinders and planes under any affine transform are handled; cones,
// unbounded cylinders and triangles are always kept.
// ERROR_BUDGET.md B1 - B5 derive every constant below from three errors of the exact f32 evaluation the cull must agree with:
// E1, the rounding of object-space positions (the per-object constant E in trn.w, host: flatten), E2, the cancellation of
// the sphere / cylinder quadratic (a relative 16 u |o|^2: hence "within 100 radii", in the OBJECT's units), E3, the slabs'.
// `dark` (per lane): every sample of this shade point is shadowed -- it sits on the far side of a casting sphere, see
// the end of per_object -- so intensity_at may answer 0 without a test.
template <int NOBJ>
DI uint32_t light_cull_mask(const SceneHdr& H, const SceneSoA& S, V3 p, bool& dark) {
    uint32_t mask = 0;
    dark = false;
    bool casters_left = false;  // wave-uniform
    // B5: the reference reports a sphere's / cylinder's hits up to sqrt(1 + 16 u oo) of its radius away from its centre / axis (E2);
    // the largest such growth (squared, with 32 u) among the objects this lane has looked at: what their bounding spheres
    // are inflated by before distances to them are compared (`reach`)
    float grow2 = 1.0f;
    const float pmax = fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z));
    // the samples stay inside the parallelogram only for jitter in [0, 1] (the hashed source draws from (0, 1])
    const bool hashed = spec_jitter_mode(H.jitter_mode) == RTC_JITTER_HASHED;
    if (!(H.cull_flags & CULL_ENABLED)) return 0u;
    if (!hashed && !(H.jitter_const >= 0.0f && H.jitter_const <= 1.0f)) return 0u;
    // a decision costs about as much as four or five ray-object tests: not worth it for a handful of samples
    if (H.u_steps * H.v_steps < 8) return 0u;
    // unrolled kernels: all NOBJ records; any-count loops and tree walks: up to 32 objects (the mask's width)
    if (NOBJ <= 0 && H.n_objects > 31u) return 0u;
    // casters first: when none of them is left the answer is known and the non-casters need not be looked at at all
    auto per_object = [&](uint32_t i, bool casters) {
        const Obj ob = load_obj_static<NOBJ <= 0>(S, i);
        const uint32_t bits = ob.bits;
        const uint32_t kind = bits & SHAPE_KIND_MASK;
        if (kind == SHAPE_NONE || ((bits & SHAPE_CASTS) != 0u) != casters) return;  // wave-uniform
        // Bounding sphere of the shape in its own space (centre = origin), inflated by 3 % (LIGHT_CULL_INFLATE2).  Every intersection
        // these kinds report lies on the shape.  (Not so for cones -- the near-parallel branch, cone.rs:99-107,
        // returns a root of the unbounded double cone without a range check -- nor for unbounded cylinders;
        // triangles are kept as well.)
        float r2 = 0.0f;  // 0: keep
        if (kind == RTC_SPHERE) r2 = LIGHT_CULL_INFLATE2;
        else if (kind == RTC_CUBE) r2 = 3.0f * LIGHT_CULL_INFLATE2;
        else if (kind == RTC_CYLINDER) {
            const float hy = fmaxf(fabsf(ob.min_y()), fabsf(ob.max_y()));
            if (hy < 1e6f) r2 = LIGHT_CULL_INFLATE2 * (1.0f + hy * hy);
        }
        if (kind != RTC_PLANE && !(r2 > 0.0f)) {  // wave-uniform
            if (bits & SHAPE_CASTS) casters_left = true;
            return;  // (cones and unbounded cylinders have an infinite bounding sphere, a triangle's errors are linear: `reach` needs nothing)
        }
#ifdef RTC_NO_EGUARD  // development: A/B (the cull as it was before the E1 terms)
        const float E = 0.0f;
#else
        const float E = ob.trn.w;  // E1: how far this object's pyramid may sit from the rays the exact test traces, object units (host: flatten)
#endif
        // the object-space shade point, computed as the exact tests compute it (same function, same value)
        const V3 o = obj_point(ob, p);
        const float4 l0 = S.lcorn[3 * i], l1 = S.lcorn[3 * i + 1], l2 = S.lcorn[3 * i + 2];  // the light's corners, wave-uniform
        auto corner = [&](int k) {
            return k == 0 ? v3(l0.x, l0.y, l0.z) : k == 1 ? v3(l0.w, l1.x, l1.y) : k == 2 ? v3(l1.z, l1.w, l2.x) : v3(l2.y, l2.z, l2.w);
        };
        bool cull;
        if (kind == RTC_PLANE) {
            // plane.rs:45-56 gives a t >= 0 only if the object-space origin height o.y and the direction's d.y have
            // strictly opposite signs, and d.y is a positive multiple of L.y - o.y for the sample's object-space
            // position L.  If every corner -- hence every sample -- is at least as far from the plane as o on o's own
            // side (by a margin far above the rounding of either), every sample runs parallel (rejected) or leaves the
            // plane behind.  o.y is the very value the exact test uses.
            const float lo = fminf(fminf(corner(0).y, corner(1).y), fminf(corner(2).y, corner(3).y));
            const float hi = fmaxf(fmaxf(corner(0).y, corner(1).y), fmaxf(corner(2).y, corner(3).y));
            // B1 (plane): the margin is 1e-4 of the magnitudes (their own rounding is u) plus twice E1 -- the corners' and o.y's
            // distance from the true images of the light and of p, which is absolute: a floor a thousand units below the
            // world's origin rounds o.y to 1e-4 whatever its value -- and, for a plane that is rotated or sheared, the part
            // of E1 that grows with |p| (o.y = row . p + t cancels on the horizon of a tilted floor): errB |p|_inf.
            float m = 1e-4f * (fabsf(lo) + fabsf(hi) + fabsf(o.y)) + 2.0f * E;
            if (!(bits & SHAPE_DIAG)) m += 2.0f * ob.off2.w * pmax;
            cull = (o.y > 0.0f && lo - o.y >= m) || (o.y < 0.0f && hi - o.y <= -m);
        } else {
            // The rays fill the pyramid with apex o over the (object-space) parallelogram: the intersection of the four
            // half-spaces through o bounded by its faces.  The inflated bounding sphere lies outside as soon as ONE
            // face plane separates it: signed distance of the origin beyond that face > radius.
            // Unnormalised throughout: n.(-o) > R |n|  <=>  n.(-o) > 0 and (n.o)^2 > R^2 |n|^2.
            const float oo = o.x * o.x + o.y * o.y + o.z * o.z;
            auto edge = [&](int k) { return corner(k) - o; };  // object-space vector from the shade point to corner k
            const V3 m = edge(0) + edge(2);                    // towards the parallelogram's centre: inside the pyramid
#ifndef RTC_NO_CONE_PRETEST  // development: A/B
            if (kind == RTC_SPHERE) {
                // B1 (cone first).  Most shade points see the light far from the sphere: there the cone around the direction to the
                // light's centre that holds the whole parallelogram (half its longer diagonal, off2.w from the host, plus what E1 can
                // move it by) misses the inflated sphere -- theta > alpha + beta, in cosines c . w < sqrt(cc - hd^2) sqrt(oo - R^2) - hd R
                // -- and the four face planes of the pyramid (a hundred operations) need not be formed.  Decided by the wave.
                grow2 = fmaxf(grow2, 1.0f + 1.9073486e-6f * oo);  // (B5, `reach`: also for a sphere this test removes)
                const float R2 = LIGHT_CULL_INFLATE2 * (1.0f + 1.9073486e-6f * oo), ro = __builtin_amdgcn_sqrtf(oo);
                const float hd = ob.off2.w + 2.0f * E + 7.2e-7f * ro;  // (4 u |o|, three times over: o's own rounding beyond E's 103 radii)
                const V3 c = m * 0.5f;
                const float cc = dot3(c, c);
                const float rhs = hd * __builtin_amdgcn_sqrtf(R2) - __builtin_amdgcn_sqrtf(cc - hd * hd) * __builtin_amdgcn_sqrtf(oo - R2);  // NaN: inside, or too close
                if (__all(oo > 1.001f * R2 && oo < 1e12f && dot3(c, o) > rhs && ob.off2.w > 0.0f)) {
                    mask |= 1u << i;
                    return;
                }
            }
#endif
            bool narrow = true, outside = false, leaving = true, entering = true;
            const float c_own = oo - 1.0f;  // the sphere quadratic's constant term, as the exact test computes it
            // Cubes and bounded cylinders: their own box [-1, 1] x [min_y, max_y] x [-1, 1] instead of the sphere around it
            // (a tall cylinder's sphere swallows the floor it stands on), inflated by 3 % of each half extent (+1e-4 of the
            // bounds): a face plane separates the box when the box centre lies further out than the box's half width
            // along the plane's normal.  Every intersection these kinds report lies in the box (cylinder.rs: both roots
            // are range-checked against min_y / max_y, the caps against the radius; cube.rs: the slabs themselves).
            const bool boxy = kind == RTC_CUBE || kind == RTC_CYLINDER;  // wave-uniform
            // B1 (sphere, E2): the reference's quadratic reports a hit for lines that pass within sqrt(1 + 16 u oo) radii -- 1 % at a
            // hundred radii, the sphere twice its size at a thousand.  Up to round 3 the cull simply stood back beyond 100 radii
            // -- and the far end of C3's floor, a twentieth of its pixels, ran all hundred samples for two spheres a pixel wide.
            // Now the inflated sphere grows with the distance instead (32 u oo: twice the bound): what the pyramid must miss is
            // what the reference could at most report.
#ifdef RTC_NO_GROW  // development: A/B
            const float r2e = r2;
#else
            const float r2e = kind == RTC_SPHERE ? r2 * (1.0f + 1.9073486e-6f * oo) : r2;
#endif
            float cy = 0.0f, hy = 1.03f;
            float ee_min = RTC_INF;  // the nearest corner (squared distance from o)
            bool gentle = true;      // cylinders: no corner direction steeper than 10 : 1 against the axis
            if (kind == RTC_CYLINDER) {
                cy = 0.5f * (ob.min_y() + ob.max_y());
                // B1 (cylinder): the wall's roots carry E2's relative error -- up to 2e-3 of t, so the height o.y + t d.y the
                // reference range-checks is off by up to 2e-3 of the height difference to o (four times that here), and near a
                // tangent by 1e-3 of the slope (`gentle`: slopes of at most ten, 0.01 of a radius)
                hy = 1.03f * (0.5f * (ob.max_y() - ob.min_y())) + 1e-4f * (fabsf(ob.min_y()) + fabsf(ob.max_y()) + 1.0f) +
                     (8e-3f * (fabsf(o.y) + fabsf(ob.min_y()) + fabsf(ob.max_y())) + 0.01f);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const V3 e0 = edge(k), e1 = edge((k + 1) & 3);
                const float me = dot3(m, e0);  // every corner within ~66 degrees of the axis: a proper convex cone
                const float ee = dot3(e0, e0);
                ee_min = fminf(ee_min, ee);
                if (kind == RTC_CYLINDER) gentle = gentle && e0.x * e0.x + e0.z * e0.z >= 0.01f * ee;
                narrow = narrow && me > 0.0f && me * me > 0.16f * dot3(m, m) * ee;
                const float oe = dot3(o, e0);  // corner direction at least ~3 degrees above the sphere's tangent plane at o
                leaving = leaving && oe > 0.0f && oe * oe > 0.0025f * oo * ee;
                // corner direction at least ~2 sqrt(c) below the tangent plane, and the corner beyond the sphere's far side
                entering = entering && oe < 0.0f && oe * oe > 4.2f * c_own * oo * dot3(e0, e0) && oe * oe > 4.84f * oo;
                V3 n = cross3(e0, e1);
                if (dot3(n, m) > 0.0f) n = -n;  // outward
                const float h = -dot3(n, o);    // (distance of the origin outside face k) * |n|
                if (boxy) outside = outside || h + n.y * cy > 1.005f * (1.03f * (fabsf(n.x) + fabsf(n.z)) + hy * fabsf(n.y));
                else outside = outside || (h > 0.0f && h * h > 1.01f * r2e * dot3(n, n));
            }
            // `near_enough`: a sphere's E2 is in r2e; a cube's slabs err by 4 u |o| (E3: 1e-3 at |o| = 3 000); a cylinder keeps the
            // hundred radii (its wall's heights, E2(iii), have no such closed form), on xz as well.
            // `aimed`: E1 turns the pyramid by up to E / (distance to the light's nearest point, at least 0.4 of the nearest corner's
            // under `narrow`), which moves it at the object, at most sqrt(oo) + sqrt(r2e) away, by less than 0.01 when
            // ee_min > 1.25e5 E^2 (oo + r2e); beyond the 103 radii E was computed for, o's own rounding adds 4 u |o| to it.
#ifdef RTC_NO_GROW
            const bool near_enough = oo < 1e4f * r2 && (kind != RTC_CYLINDER || o.x * o.x + o.z * o.z < 1e4f * LIGHT_CULL_INFLATE2);
#else
            const bool near_enough = kind == RTC_SPHERE ? oo < 1e12f : kind == RTC_CUBE ? oo < 1e7f
                                     : (oo < 1e4f * r2 && o.x * o.x + o.z * o.z < 1e4f * LIGHT_CULL_INFLATE2);
#endif
            const bool aimed = ee_min > 1.25e5f * (2.0f * (E * E) + 1.15e-13f * oo) * (oo + r2e);  // (NaN / infinite E: false)
            if (kind == RTC_SPHERE) grow2 = fmaxf(grow2, 1.0f + 1.9073486e-6f * oo);  // 32 u oo
            // (a cylinder's wall also reports heights off by 2e-3 of the height difference to o, E2(iii): bounded here by a tenth of
            // its radius, or its bounds mean nothing)
            if (kind == RTC_CYLINDER) grow2 = (near_enough && fabsf(o.y) < 10.0f + fabsf(ob.min_y()) + fabsf(ob.max_y())) ? fmaxf(grow2, 1.21f) : RTC_INF;
            cull = (boxy || oo > r2e) && near_enough && aimed && narrow && outside && gentle;
            if (boxy) {
                // A shade point beyond one face of the box -- typically ON that face of the object itself -- whose every sample
                // moves further out along that axis: the slab's parameter interval [(lo - o) / d, (hi - o) / d] is negative
                // (or -inf for d = 0), so cube.rs:90-129 reports nothing, cylinder.rs' range checks o.y + t d.y against
                // min_y / max_y fail for every t >= 0 and its caps lie at t < 0.  Likewise radially for a cylinder: with
                // c = ox^2 + oz^2 - 1 > 0 and b = 2 (ox dx + oz dz) >= 0 the radius only grows with t, which rules out the
                // wall and the caps' radius check.  Margins: 1e-5 on the position (over_point is 1.2e-3 / scale out),
                // 1e-3 |e| on the direction component (its rounding is ~1e-7 |e|).  Sample directions are positive
                // combinations of the corner directions, so the corner with the smallest component decides.
                const V3 e0 = edge(0), e1 = edge(1), e2 = edge(2), e3 = edge(3);
                // (+ 2 E: the corners' own distance from the true image of the light, E1)
                const float tol = 1e-3f * sqrtf(fmaxf(fmaxf(dot3(e0, e0), dot3(e1, e1)), fmaxf(dot3(e2, e2), dot3(e3, e3)))) + 2.0f * E;
                auto beyond = [&](float oa, float lo_a, float hi_a, float a0, float a1, float a2, float a3) {
                    const float emin = fminf(fminf(a0, a1), fminf(a2, a3)), emax = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
                    const float m = 1e-5f * (fabsf(lo_a) + fabsf(hi_a) + 1.0f);
                    return (oa > hi_a + m && emin >= tol) || (oa < lo_a - m && emax <= -tol);
                };
                const float ylo = kind == RTC_CYLINDER ? ob.min_y() : -1.0f, yhi = kind == RTC_CYLINDER ? ob.max_y() : 1.0f;
                bool away = beyond(o.y, ylo, yhi, e0.y, e1.y, e2.y, e3.y);
                if (kind == RTC_CUBE) {
                    away = away || beyond(o.x, -1.0f, 1.0f, e0.x, e1.x, e2.x, e3.x) || beyond(o.z, -1.0f, 1.0f, e0.z, e1.z, e2.z, e3.z);
                } else {
                    const float rr = o.x * o.x + o.z * o.z, rtol = tol * sqrtf(rr);
                    away = away || (rr - 1.0f > 1e-4f && o.x * e0.x + o.z * e0.z >= rtol && o.x * e1.x + o.z * e1.z >= rtol &&
                                    o.x * e2.x + o.z * e2.z >= rtol && o.x * e3.x + o.z * e3.z >= rtol);
                }
                cull = cull || (away && near_enough);
            }
            // A shade point sitting just outside the unit sphere whose whole light pyramid points away from it: with
            // c = |o|^2 - 1 > 0 and b = 2 pd.o > 0 both roots (-b -+ sqrt(b^2 - 4ac)) / 2a are negative, and by more
            // than rounding can undo because 4ac / b^2 >= c / (1 + c) >= 8e-5 (c is the exact test's own value).
            // (B2; the corner directions are good to E / |e|: a hundredth of the three degrees when the nearest corner is 100 E away)
            const bool corners_sure = ee_min > 1e4f * (E * E);
            if (kind == RTC_SPHERE) cull = cull || (oo - 1.0f > 1e-4f && oo <= 1.21f && leaving && corners_sure);
            // The mirror image: a shade point just outside a CASTING unit sphere whose whole light pyramid points into it.
            // The directions to the samples are positive combinations of the corner directions, and {d : o.d <= -s |d|}
            // is a convex cone, so every sample direction satisfies (o.d)^2 > 4.2 c |o|^2 |d|^2 with o.d < 0: the
            // discriminant 4 ((o.d)^2 - |d|^2 c) is positive by three quarters of its value, the first root
            // 2c / (|b| + sqrt(disc)) is positive (c > 1e-4 keeps the cancellation in (-b - sqrt(disc)) / 2a at parts in
            // 1e3 of |b|), and it lies before the light because every sample is more than 2.2 radii beyond o along -o
            // while the sphere ends within 2.1.  So is_shadowed finds a hit below `distance`; the nearest hit may belong
            // to another object, but if that is a caster the answer is the same, and non-casters are dealt with below.
            if (kind == RTC_SPHERE && (bits & SHAPE_CASTS) && !(bits & SHAPE_LOOSE) && c_own > 1e-4f && oo <= 1.21f && entering && corners_sure) dark = true;  // B3
        }
        if (__all(cull)) mask |= 1u << i;
        else if (bits & SHAPE_CASTS) casters_left = true;
    };
    if constexpr (NOBJ > 0) {
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)NOBJ; i++) per_object(i, true);
    } else {
        for (uint32_t i = 0; i < H.n_objects; i++) per_object(i, true);
    }
    // no shadow caster can be reached from this shade point: every sample is lit whatever else is in the way
    // (world.rs:104-119 asks for the nearest hit to BE a caster), see intensity_at
    if (!casters_left) {
        mask |= LIGHT_CULL_ALL_CASTERS;
        return mask;  // (dark is false for every lane: the sphere it sits behind would have been kept)
    }
    if (H.all_cast == 0u) {
        if constexpr (NOBJ > 0) {
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)NOBJ; i++) per_object(i, false);
        } else {
            for (uint32_t i = 0; i < H.n_objects; i++) per_object(i, false);
        }
    }
    bool noncasters_left = false;  // wave-uniform
    // Objects that cast no shadow (the demo's lampshade around its area light) matter to a shadow ray only by being hit
    // BEFORE the nearest caster.  One that lies, as seen from p, wholly behind every caster still in play cannot be:
    // nearest point of its bounding sphere farther than the farthest point of theirs.  ERROR_BUDGET.md B5: the distances
    // the reference reports are off by up to 1e-3 of themselves (E2's cap) -- hence 0.999 x the one against 1.001 x the
    // other -- and the hits of a sphere or cylinder seen from afar lie up to sqrt(1 + 16 u oo) radii from its centre / axis
    // (E2; a disc scaled 1e-3 across, seen from ten units, seven radii): the bounding spheres grow by that (`grow2`).
    if (H.all_cast == 0u) {
        float far_casters = 0.0f;  // per lane
        const float grow = __builtin_amdgcn_sqrtf(grow2);
        auto reach = [&](uint32_t i, bool caster) {
            const uint32_t bits = spec_bits(i, __float_as_uint(load_obj_static<NOBJ <= 0>(S, i).geo.w));
            if ((bits & SHAPE_KIND_MASK) == SHAPE_NONE || ((bits & SHAPE_CASTS) != 0u) != caster || ((mask >> i) & 1u)) return;
            const float4 b = S.bsph[i];  // wave-uniform
            const V3 c = p - v3(b.x, b.y, b.z);
            const float dist = sqrtf(dot3(c, c));
            const float radius = b.w * grow;
            if (caster) {
                far_casters = fmaxf(far_casters, dist + radius);  // +inf for an unbounded caster (and NaN-safe: fmaxf keeps inf)
                if (!(radius < RTC_INF)) far_casters = RTC_INF;
            } else if (__all(0.999f * dist - radius > 1.001f * far_casters)) {
                mask |= 1u << i;
            } else {
                noncasters_left = true;
            }
        };
        if constexpr (NOBJ > 0) {
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)NOBJ; i++) reach(i, true);
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)NOBJ; i++) reach(i, false);
        } else {
            for (uint32_t i = 0; i < H.n_objects; i++) reach(i, true);
            for (uint32_t i = 0; i < H.n_objects; i++) reach(i, false);
        }
    }
    if (noncasters_left || !(H.cull_flags & CULL_DARK)) dark = false;  // a non-caster might be hit first
    return mask;
}

// ---- fast decision of an area-light shadow sample -------------------------------------------------------------------
// World::is_shadowed (world.rs:104-119) answers a yes / no question, and for most samples the answer is far from close:
// the ray passes well inside or well outside a sphere, well before or well beyond the light.  When every object still
// in play for this shade point (light_cull_mask) is a shadow-casting sphere or plane of a scale+translate-only scene,
// "the nearest hit is a caster closer than the light" is simply "SOME caster has its first intersection at a ray
// parameter in [0, distance)", and that can be decided per caster from the UNNORMALISED vector v = sample - point:
// with pd = t_inverse v, a = pd.pd, hb = pd.o, q = hb^2 - a c (a quarter of the discriminant, c the exact test's own
// constant term) the roots along v are (-hb -+ sqrt q) / a, the light sits at parameter 1, so
//     no hit      <=>  q < 0, or both numerators < 0
//     blocked     <=>  the smallest non-negative numerator n satisfies n < a
// -- no normalisation (an exact square root and three exact divisions per sample), no division, an approximate square
// root.  The answer is only taken when every comparison holds with a RELATIVE margin of 1e-3 against the magnitudes
// that went into it (the exact evaluation's own rounding, and this one's, are ~1e-6 of those same magnitudes: where
// both are three orders away from a boundary they are on the same side of it); anything closer, any NaN, any
// magnitude outside [1e-30, 1e30] is "uncertain" and the sample takes the exact path (is_shadowed_pre) as before.
// RTC_AMD_FAST_SHADOW=0 switches it off; whole C3 frames with it on and off are compared value for value and count for
// count (tests/test_gpu_fullsize.py), as are the boundary-hugging scenes of tests/test_gpu_light_cull.py.
constexpr float FAST_MARGIN = 1e-3f;
enum { SHADOW_LIT = 0, SHADOW_BLOCKED = 1, SHADOW_UNCERTAIN = 2 };
// wave-uniform: may the samples of this shade point be decided by shadow_fast?
template <int NOBJ, bool SIMPLE>
DI bool shadow_fast_usable(const SceneHdr& H, const SceneSoA& S, uint32_t skip) {
    if constexpr (!SIMPLE || NOBJ <= 0) return false;
    if (!(H.cull_flags & CULL_FAST_SHADOW)) return false;
    bool ok = true;
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)(NOBJ > 0 ? NOBJ : 1); i++) {
        const uint32_t bits = spec_bits(i, __float_as_uint(S.geo[i].w)), kind = bits & SHAPE_KIND_MASK;
        if (kind == SHAPE_NONE || ((skip >> i) & 1u)) continue;
        if (!(bits & SHAPE_CASTS) || (kind != RTC_SPHERE && kind != RTC_PLANE)) ok = false;
    }
    return ok;
}
template <int NOBJ>
DI int shadow_fast(const SceneSoA& S, const ShadowPre* pre, V3 v, uint32_t skip) {
    bool blocked = false, uncertain = false;
    const float vv = dot3(v, v);
    // x in [1e-30, 1e30] (not NaN, not negative) as ONE unsigned compare of the bit patterns
    auto in_range = [](float x) { return __float_as_uint(x) - 0x0da24260u <= 0x7149f2cau - 0x0da24260u; };
    if (!in_range(vv)) return SHADOW_UNCERTAIN;
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)(NOBJ > 0 ? NOBJ : 1); i++) {
        const float4 g = S.geo[i];
        const uint32_t bits = spec_bits(i, __float_as_uint(g.w)), kind = bits & SHAPE_KIND_MASK;
        if (kind == SHAPE_NONE || ((skip >> i) & 1u)) continue;  // wave-uniform
        const V3 o = pre[i].o;
        if (kind == RTC_SPHERE) {
            // straight-line code: every quantity is formed, the verdicts are combined at the end (the branches this
            // replaces cost more than the few operations they skipped)
            // a = |t_inverse v|^2 and hb = (t_inverse v) . o; a uniformly scaled sphere (t_inverse = s I + translation) has
            // a = s^2 |v|^2 and hb = s (v . o) -- other roundings of the same quantities, which is all this decision needs
            // (its margins are three orders of magnitude above any of them)
            float a, hb;
#ifdef RTC_NO_UNIFORM_FAST  // development: A/B
            constexpr bool use_uniform = false;
#else
            constexpr bool use_uniform = true;
#endif
            if (use_uniform && (bits & SHAPE_UNIFORM)) {
                a = (g.x * g.x) * vv;
                hb = g.x * dot3(v, o);
            } else {
                const V3 pd = v3(g.x * v.x, g.y * v.y, g.z * v.z);
                a = dot3(pd, pd), hb = dot3(pd, o);
            }
            const float hb2 = hb * hb, ac = a * pre[i].c;
            const float q = hb2 - ac;
            const bool sure_q = in_range(a) && fabsf(q) >= FAST_MARGIN * (hb2 + fabsf(ac));  // NaN: false
            const float sq = __builtin_amdgcn_sqrtf(fabsf(q)), tol = FAST_MARGIN * (fabsf(hb) + sq);  // (q < 0: nothing below is looked at)
            const float n1 = -hb - sq, n2 = -hb + sq;  // numerators of the two roots, n1 <= n2
            const float n = n1 >= 0.0f ? n1 : n2;      // the first intersection at a parameter >= 0, if n >= 0
            // min(|n1|, |n2|) = ||hb| - sq|, the same subtraction as whichever numerator it is
            const bool sure_n = fabsf(fabsf(hb) - sq) >= tol && fabsf(n - a) >= FAST_MARGIN * (fabsf(n) + a);
            const bool misses = q < 0.0f || n < 0.0f;  // the line misses the sphere, or the sphere lies behind
            if (!sure_q || (!(q < 0.0f) && !sure_n)) uncertain = true;
            else if (!misses && n < a) blocked = true;
            // (Round 3 also built the decision from SIGNS alone -- roots exist, the side of t = 1 from f(1) = A + 2 B + C and the
            // vertex A + B: no square root, no numerator whose cancellation decides.  Same image; slower, 0.86 -> 1.03 ms on C3:
            // a dozen more lane masks to combine, 74 spilled scalar registers.  LABNOTES.md "Round 3".)
        } else {  // RTC_PLANE (plane.rs:45-56): parallel if |d.y| < PLANE_EPS for the normalised direction d = pd / |v|
            const float pdy = g.y * v.y, pdy2 = pdy * pdy, lim = PLANE_EPS * PLANE_EPS * vv;
            if (!(fabsf(pdy2 - lim) >= FAST_MARGIN * (pdy2 + lim)) || !(fabsf(o.y) >= 1e-30f)) {
                uncertain = true;
                continue;
            }
            if (pdy2 < lim) continue;                                    // parallel: no intersection
            if ((o.y > 0.0f) == (pdy > 0.0f)) continue;                  // t = -o.y / pd.y < 0: the plane lies behind
            const float ay = fabsf(o.y), ap = fabsf(pdy);                // t = ay / ap against the light's parameter 1
            if (!(fabsf(ay - ap) >= FAST_MARGIN * (ay + ap))) uncertain = true;
            else if (ay < ap) blocked = true;
        }
    }
    return blocked ? SHADOW_BLOCKED : uncertain ? SHADOW_UNCERTAIN : SHADOW_LIT;
}
// one sample of intensity_at's loop (unrolled kernels): the fast decision where it applies and is certain, else the exact one
template <int NOBJ, bool SIMPLE>
DI bool sample_blocked(const SceneHdr& H, const SceneSoA& S, const ShadowPre* pre, V3 lp, V3 p, Counters& cnt, uint32_t skip, bool fast) {
    if constexpr (SIMPLE && NOBJ > 0) {
        if (fast) {  // wave-uniform
            const int r = shadow_fast<NOBJ>(S, pre, lp - p, skip);
            if (r != SHADOW_UNCERTAIN) {
                cnt.rays++;
                return r == SHADOW_BLOCKED;
            }
        }
    }
#ifdef RTC_COUNT_EXACT  // development: the `culled` statistic counts the samples that took the exact path instead
    cnt.add_culled(1u);
#endif
    return is_shadowed_pre<NOBJ, SIMPLE>(H, S, pre, lp, p, cnt, skip);
}

// Pinned jitter (DESIGN.md "Jitter"): counter-based hash keyed by (pixel, path code, cell, draw):
// h = mix32(base + (2 cell + draw) * 0x85EBCA6B), one 32-bit mix per draw, mapped to the 23-bit grid of rand's
// OpenClosed01 (rectangle_light.rs:46): ((h >> 9) + 1) * 2^-23 in (0, 1].
DI uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
DI uint32_t jitter_base(uint32_t seed, uint32_t pixel, uint32_t path) {
    uint32_t a = mix32(pixel ^ seed);
    return mix32(a + path * 0x9E3779B9u);
}
// ((h >> 9) + 1) * 2^-23 without the int -> float conversion: {0x7f, h} >> 9 is the float 1 + (h >> 9) 2^-23 in [1, 2),
// and subtracting 1 - 2^-23 from it is exact (the difference, (k + 1) 2^-23 <= 1, is representable).  Two instructions;
// the same bits as the plain formula, which is what the oracle evaluates.
DI float jitter_value(uint32_t h) {
    return __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, h, 9u)) - 0.99999988079071044921875f;
}
// rectangle_light.rs:60-66: corner + u_vec * (u + jitter1) + v_vec * (v + jitter2), per component, left to right; the terms
// LIGHT_ZEROS names are exact zeros and are not formed
DI V3 point_on_light(V3 corner, V3 uvec, V3 vvec, float a, float b) {
    V3 lp = corner;
    if (!(LIGHT_ZEROS & 1u)) lp.x = lp.x + uvec.x * a;
    if (!(LIGHT_ZEROS & 2u)) lp.y = lp.y + uvec.y * a;
    if (!(LIGHT_ZEROS & 4u)) lp.z = lp.z + uvec.z * a;
    if (!(LIGHT_ZEROS & 8u)) lp.x = lp.x + vvec.x * b;
    if (!(LIGHT_ZEROS & 16u)) lp.y = lp.y + vvec.y * b;
    if (!(LIGHT_ZEROS & 32u)) lp.z = lp.z + vvec.z * b;
    return lp;
}

// ---- block cones: parts of the light decided before any sample is drawn (ERROR_BUDGET.md B10) ------------------------------
// The whole-light cull (light_cull_mask) answers a shade point whose light is entirely clear of the casters, or entirely
// behind one.  What is left is the penumbra, where every lane of the wave ran all u_steps x v_steps samples -- hash two jitters,
// place the sample, test it -- although the casters' silhouettes cross only a part of the light.  A sample of cell (u, v) lies in
// that cell whatever its jitter (rectangle_light.rs:60-66: corner + u_vec (u + j1) + v_vec (v + j2), j in [0, 1]); a block of 2 x 2
// cells lies within `hd2x`, the cells' diagonal, of its centre: the rays into it lie in the cone around the direction c to that
// centre with sin(alpha) = hd2x / |c|.  A casting sphere seen from the shade point fills the cone around w (towards its centre) with
// sin(beta) = R / |w|.  With theta the angle between c and w:  theta > alpha + beta for every caster in play  =>  no ray into the
// block meets a caster: its four samples are LIT -- total += 4, four rays counted, no jitter hashed.  In cosines, multiplied
// through by |c| |w|:  c . w < sqrt(cc - hd2x^2) sqrt(ww - R^2) - hd2x R.  The decision is taken by the WAVE (a vote): the blocks some
// lane cannot call lit are sampled by every lane exactly as before, so the loop stays the loop it was -- one cell at a time, the
// same cell in every lane.  (The same vote for single cells inside a block that failed: 6 % fewer samples, and 20 - 28 % slower, rolled
// or unrolled.  Round 4 first built the finer thing -- cells decided one by one, LIT and BLOCKED, every lane sampling
// its own list of undecided cells: 3.5 times fewer samples on C3, and slower: lanes that sample different cells share no hash
// schedule, no uniform branches, and nearly every round has some lane on the exact path.  LABNOTES.md "Round 4".)
// Everything is evaluated in the sphere's own space, where the exact test lives: o (the exact test's own object-space shade
// point) gives w = -o, ww = oo; a uniformly scaled sphere maps the world vector c_w = centre - p to c = g c_w, so that |g| cancels:
//     -sign(g) (c_w . o) < sqrt(cc_w - hd2x^2) sqrt(oo - R^2) - hd2x R.
// Margins (B10): R^2 = 1.0609 (1 + 32 u oo) -- 3 % in radius, and what E2 lets the reference's quadratic report at this distance,
// twice over; the sample's own rounding (3 u |L|) is in hd's padding (host: SceneHdr::cell_hd); no E1 exposure -- the cone's apex
// is the exact test's o and its axis the same world-space difference the exact test transforms.  Only LIT is claimed, which holds
// whatever else is in play as long as every CASTER in play is such a sphere: a non-caster never shadows (world.rs:117).
template <int NOBJ>
DI bool blocks_usable(const SceneHdr& H, const SceneSoA& S, uint32_t skip) {  // wave-uniform
    if (!(H.cull_flags & CULL_CELLS) || !(H.cell_hd > 0.0f) || ((H.u_steps | H.v_steps) & 1) != 0) return false;
    if (spec_jitter_mode(H.jitter_mode) == RTC_JITTER_SEQUENCE) return false;  // (its draws are numbered in the reference's cell order: the plain loop)
    bool ok = true;
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)(NOBJ > 0 ? NOBJ : 1); i++) {
        const uint32_t bits = spec_bits(i, __float_as_uint(S.geo[i].w)), kind = bits & SHAPE_KIND_MASK;
        if (kind == SHAPE_NONE || ((skip >> i) & 1u) || !(bits & SHAPE_CASTS)) continue;
        if (kind != RTC_SPHERE || !(bits & SHAPE_UNIFORM)) ok = false;
    }
    return ok;
}

// Light::intensity_at: point_light.rs:28-34, rectangle_light.rs:60-66, 76-88
template <int NOBJ, bool SIMPLE>
DI float intensity_at(const SceneHdr& H, const SceneSoA& S, V3 p, uint32_t pixel, uint32_t path, Counters& cnt) {
    if (spec_light_kind(H.light_kind) == RTC_LIGHT_POINT) {
        return is_shadowed<NOBJ, true>(H, S, v3(H.lpos[0], H.lpos[1], H.lpos[2]), p, cnt) ? 0.0f : 1.0f;
    }
#ifdef RTC_LIGHT_VGPRS
    // The light's geometry as per-lane values: the sample loop then multiplies VGPR by VGPR (an SGPR operand puts v_mul / v_add
    // in the half-rate issue class, profiles/ubench_valu_r03.txt) and the compiler has nine scalar registers less to keep --
    // or to re-load from the argument block inside the loop -- across it.
    auto in_vgpr = [](float x) {
        float r;
        asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
        return r;
    };
#else
    auto in_vgpr = [](float x) { return x; };
#endif
    const bool hashed = spec_jitter_mode(H.jitter_mode) == RTC_JITTER_HASHED;
    // every shadow ray of this shade point starts at p: do the origin-only work once per object
    constexpr bool PRE = NOBJ > 0;
    ShadowPre pre[PRE ? NOBJ : 1];
    bool dark;
    const uint32_t skip = light_cull_mask<NOBJ>(H, S, p, dark);  // before pre[] becomes live: the cull needs registers of its own
    if (skip & LIGHT_CULL_ALL_CASTERS) {
        // wave-uniform: no shadow caster is reachable, so each of the u_steps * v_steps is_shadowed() calls answers
        // "lit": total = 1.0 + ... + 1.0 = cells exactly (an integer below 2^24), and cells / cells = 1.0
        const uint32_t cells = (uint32_t)(H.u_steps * H.v_steps);
        cnt.rays += cnt.my_cells(cells);
#ifndef RTC_COUNT_EXACT
        cnt.add_culled(cnt.my_cells(cells));  // statistics: rays answered without an object test
#endif
        return (float)cells / H.cells_f;
    }
    if (dark) {
        // per lane: every is_shadowed() call answers "shadowed": total stays 0.0, and 0.0 / cells = 0.0
        const uint32_t cells = (uint32_t)(H.u_steps * H.v_steps);
        cnt.rays += cnt.my_cells(cells);
#ifndef RTC_COUNT_EXACT
        cnt.add_culled(cnt.my_cells(cells));
#endif
        return 0.0f / H.cells_f;
    }
    const V3 corner = v3(in_vgpr(H.corner[0]), in_vgpr(H.corner[1]), in_vgpr(H.corner[2]));
    const V3 uvec = v3(in_vgpr(H.uvec[0]), in_vgpr(H.uvec[1]), in_vgpr(H.uvec[2]));
    const V3 vvec = v3(in_vgpr(H.vvec[0]), in_vgpr(H.vvec[1]), in_vgpr(H.vvec[2]));
    uint32_t key = hashed ? jitter_base(H.jitter_seed, pixel, path) : 0u;
    if constexpr (PRE) shadow_prepare<NOBJ>(H, S, p, pre);
    const bool fast = shadow_fast_usable<NOBJ, SIMPLE>(H, S, skip);  // wave-uniform
    // The cells of the light in the reference's order (v outer, u inner), this lane's share of them: cell = sub, sub +
    // 2^s, ...  `total` only ever holds a whole number of at most `cells` (< 2^24), so the order of the additions and
    // their split over lanes cannot change it.
    float total = 0.0f;
#ifndef RTC_NO_CELLS  // development: A/B
    if constexpr (SIMPLE && NOBJ > 0 && !Counters::SHARE_LANES) {
        if (blocks_usable<NOBJ>(H, S, skip)) {  // wave-uniform: block cones (see above)
            constexpr int N = NOBJ;
            float S_lit[N];
            const float hd = 2.0f * H.cell_hd, hd2 = hd * hd;  // a block's cone: the cells' whole diagonal
            // hd R, with the largest R any lane may use: 1.03 sqrt(1 + 32 u oo) at oo = 2.5e5 (500 radii; beyond, a lane calls nothing lit).
            // The same for every lane and sphere -- a scalar, where the exact product would be a register per sphere.
            const float T_lit = hd * 1.2535f;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const uint32_t bits = spec_bits(i, __float_as_uint(S.geo[i].w));
                if ((bits & SHAPE_KIND_MASK) != RTC_SPHERE || !(bits & SHAPE_CASTS) || ((skip >> i) & 1u)) continue;  // wave-uniform
                const float oo = pre[i].c + 1.0f, R2 = 1.0609f * (1.0f + 1.9073486e-6f * oo);
                S_lit[i] = (oo > 1.001f * R2 && oo < 2.5e5f) ? __builtin_amdgcn_sqrtf(oo - R2) : RTC_NAN;  // (NaN: no comparison holds -- this lane calls nothing lit)
            }
            for (int vb = 0; vb < H.v_steps; vb += 2) {
                for (int ub = 0; ub < H.u_steps; ub += 2) {
                    // the block's centre as seen from p (formed anew per block from the loop counters: two vectors less to keep across the samples)
                    const V3 c = point_on_light(corner, uvec, vvec, (float)(ub + 1), (float)(vb + 1)) - p;
                    const float cc = __builtin_fmaf(c.x, c.x, __builtin_fmaf(c.y, c.y, c.z * c.z));
                    const float sq = __builtin_amdgcn_sqrtf(cc - hd2);  // NaN when the shade point is within the cone's base radius of the centre
                    bool lit = true;
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        const uint32_t bits = spec_bits(i, __float_as_uint(S.geo[i].w));
                        if ((bits & SHAPE_KIND_MASK) != RTC_SPHERE || !(bits & SHAPE_CASTS) || ((skip >> i) & 1u)) continue;
                        const float dot = __builtin_fmaf(c.x, pre[i].o.x, __builtin_fmaf(c.y, pre[i].o.y, c.z * pre[i].o.z));
                        const float d = S.geo[i].x < 0.0f ? dot : -dot;  // -sign(g) (c . o)   (wave-uniform choice)
                        lit = lit & (d < __builtin_fmaf(sq, S_lit[i], -T_lit));
                    }
                    if (__all(lit)) {  // four rays the reference casts and finds lit
                        total += 4.0f;
                        cnt.rays += 4u;
#ifndef RTC_COUNT_EXACT
                        cnt.add_culled(4u);
#endif
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int u = ub + (q & 1), v = vb + (q >> 1);
                            float j1 = H.jitter_const, j2 = H.jitter_const;
                            if (hashed) {  // key = base + (2 cell + draw) * 0x85EBCA6B, cell = v * u_steps + u
                                const uint32_t k = key + 2u * (uint32_t)(v * H.u_steps + u) * 0x85EBCA6Bu;
                                j1 = jitter_value(mix32(k));
                                j2 = jitter_value(mix32(k + 0x85EBCA6Bu));
                            }
                            const V3 lp = point_on_light(corner, uvec, vvec, (float)u + j1, (float)v + j2);
                            if (!sample_blocked<NOBJ, SIMPLE>(H, S, pre, lp, p, cnt, skip, fast)) total += 1.0f;
                        }
                    }
                }
            }
            return total / H.cells_f;
        }
    }
#endif
    if constexpr (!Counters::SHARE_LANES) {
        const bool sequence = spec_jitter_mode(H.jitter_mode) == RTC_JITTER_SEQUENCE;  // (never in a render kernel: the host refuses)
        uint32_t draw = 0u;  // sequence: draws taken so far in this call, modulo the list's length (wave-uniform)
        for (int v = 0; v < H.v_steps; v++) {
            for (int u = 0; u < H.u_steps; u++) {
                float j1 = H.jitter_const, j2 = H.jitter_const;
                if (hashed) {  // key = base + (2 cell + draw) * 0x85EBCA6B, cell = v * u_steps + u
                    j1 = jitter_value(mix32(key));
                    j2 = jitter_value(mix32(key + 0x85EBCA6Bu));
                    key += 2u * 0x85EBCA6Bu;
                } else if (sequence) {  // the closure's cycle: u's draw, then v's (rectangle_light.rs:62-63)
                    j1 = H.jitter_seq[draw];
                    draw = draw + 1u == H.jitter_seq_len ? 0u : draw + 1u;
                    j2 = H.jitter_seq[draw];
                    draw = draw + 1u == H.jitter_seq_len ? 0u : draw + 1u;
                }
                // corner + u_vec * (u + jitter1) + v_vec * (v + jitter2)
                V3 lp = point_on_light(corner, uvec, vvec, (float)u + j1, (float)v + j2);
                bool blocked;
                if constexpr (PRE) blocked = sample_blocked<NOBJ, SIMPLE>(H, S, pre, lp, p, cnt, skip, fast);
                else blocked = is_shadowed<NOBJ>(H, S, lp, p, cnt, skip & 0x7fffffffu);
                if (!blocked) total += 1.0f;
            }
        }
        return total / H.cells_f;
    }
    const uint32_t stride = 1u << cnt.share_log2(), cells = (uint32_t)(H.u_steps * H.v_steps);
    uint32_t cell = cnt.sub();
    int u = (int)cell, v = 0;
    while (u >= H.u_steps) u -= H.u_steps, v++;
    key += 2u * cell * 0x85EBCA6Bu;  // key = base + (2 cell + draw) * 0x85EBCA6B, cell = v * u_steps + u
    for (; cell < cells; cell += stride) {
        float j1 = H.jitter_const, j2 = H.jitter_const;
        if (hashed) {
            j1 = jitter_value(mix32(key));
            j2 = jitter_value(mix32(key + 0x85EBCA6Bu));
            key += 2u * stride * 0x85EBCA6Bu;
        }
        // corner + u_vec * (u + jitter1) + v_vec * (v + jitter2)
        V3 lp = point_on_light(corner, uvec, vvec, (float)u + j1, (float)v + j2);
        bool blocked;
        if constexpr (PRE) blocked = sample_blocked<NOBJ, SIMPLE>(H, S, pre, lp, p, cnt, skip, fast);
        else blocked = is_shadowed<NOBJ>(H, S, lp, p, cnt, skip & 0x7fffffffu);
        if (!blocked) total += 1.0f;
        u += (int)stride;
        while (u >= H.u_steps) u -= H.u_steps, v++;
    }
    for (uint32_t m = 1u; m < stride; m <<= 1) total += __shfl_xor(total, (int)m, 64);  // the pixel's lanes are adjacent
    return total / H.cells_f;
}

// Rust `f as i32`: saturating, NaN -> 0
DI int32_t rust_f32_as_i32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return (int32_t)f;
}
// Pattern::color_at_world for the five procedural patterns (stripes.rs:39-45, gradient.rs:33-36, rings.rs:38-50,
// checkers.rs:38-46, sine_2d.rs:39-44).  pa = {a.rgb, kind}, pb = {b.rgb | distance.rgb, 0}.
// `v % 2 == 0` on an i32 is `(v & 1) == 0` for either sign.
// Rust `f as usize`: saturating, NaN -> 0 (32 bits are plenty: it indexes an image)
DI uint32_t rust_f32_as_index(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
// f32::rem_euclid: r = a % rhs; if r < 0 { r + |rhs| } else { r }   (`%` on f32 is fmodf, which is exact)
DI float rem_euclid_f32(float a, float rhs) {
    float r = fmodf(a, rhs);
    return r < 0.0f ? r + fabsf(rhs) : r;
}
// UVPattern::color_at (uv.rs:45-55 UVCheckers, :145-165 AlignCheck, :366-376 UVImage)
DI V3 uv_color_at(const SceneSoA& S, uint32_t index, float u, float v) {
    const float4* r = S.uvrec + 6u * index;
    const float4 h0 = r[0];
    const uint32_t kind = __float_as_uint(h0.x);
    if (kind == RTC_UV_IMAGE) {
        const float4 h1 = r[1];
        const uint32_t iw = __float_as_uint(h1.x), ih = __float_as_uint(h1.y);
        const float vv = 1.0f - v;
        const float x = u * (float)(iw - 1u);
        const float y = vv * (float)(ih - 1u);
        const uint32_t xi = rust_f32_as_index(roundf(x)), yi = rust_f32_as_index(roundf(y));
        if (xi >= iw || yi >= ih) return v3(0.0f, 0.0f, 0.0f);  // the reference panics on the out-of-range index
        const float* px = S.texels + ((size_t)__float_as_uint(h0.w) + (size_t)yi * iw + xi) * 3u;
        return v3(px[0], px[1], px[2]);
    }
    const float4 c0 = r[2], c1 = r[3], c2 = r[4], c3 = r[5];
    const V3 col[5] = {v3(c0.x, c0.y, c0.z), v3(c0.w, c1.x, c1.y), v3(c1.z, c1.w, c2.x), v3(c2.y, c2.z, c2.w), v3(c3.x, c3.y, c3.z)};
    if (kind == RTC_UV_CHECKERS) {
        const int32_t u2 = rust_f32_as_i32(floorf(u * h0.y)), v2 = rust_f32_as_i32(floorf(v * h0.z));
        return (((uint32_t)u2 + (uint32_t)v2) & 1u) == 0u ? col[0] : col[1];
    }
    // AlignCheck: main, ul, ur, bl, br
    if (v > 0.8f) {
        if (u < 0.2f) return col[1];
        if (u > 0.8f) return col[2];
    } else if (v < 0.2f) {
        if (u < 0.2f) return col[3];
        if (u > 0.8f) return col[4];
    }
    return col[0];
}
constexpr float UV_FRAC_1_PI = 0.318309886183790671537767526745028724f;  // std::f32::consts::FRAC_1_PI
constexpr float UV_PI = 3.14159265358979323846264338327950288f;
constexpr float UV_FRAC_1_2PI = 1.0f / (2.0f * UV_PI);                   // uv.rs:12 (folded in f32, as rustc does)
DI float u_from_azimuth(V3 p) {  // uv.rs:107-113
    const float theta = atan2f_glibc(p.x, p.z);
    const float raw_u = theta * UV_FRAC_1_2PI;
    return 1.0f - (raw_u + 0.5f);
}
// TextureMap / CubicMap::color_at_world (uv.rs:84-88, 248-261)
DI V3 texture_color_at_world(const SceneSoA& S, uint32_t kind, uint32_t mapping, uint32_t first_uv, V3 p) {
    float u, v;
    uint32_t index = first_uv;
    if (kind == RTC_PATTERN_CUBE_MAP) {
        // face_from_point (uv.rs:264-283), then the face's own projection (:285-319); faces in CubicMap::new's order
        const float coord = fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z));
        if (coord == p.x) {  // right
            index += 3u;
            u = fmodf(1.0f - p.z, 2.0f) / 2.0f;
            v = fmodf(p.y + 1.0f, 2.0f) / 2.0f;
        } else if (coord == -p.x) {  // left
            index += 2u;
            u = fmodf(p.z + 1.0f, 2.0f) / 2.0f;
            v = fmodf(p.y + 1.0f, 2.0f) / 2.0f;
        } else if (coord == p.y) {  // up
            index += 4u;
            u = fmodf(p.x + 1.0f, 2.0f) / 2.0f;
            v = fmodf(1.0f - p.z, 2.0f) / 2.0f;
        } else if (coord == -p.y) {  // down
            index += 5u;
            u = fmodf(p.x + 1.0f, 2.0f) / 2.0f;
            v = fmodf(p.z + 1.0f, 2.0f) / 2.0f;
        } else if (coord == p.z) {  // front
            u = fmodf(p.x + 1.0f, 2.0f) / 2.0f;
            v = fmodf(p.y + 1.0f, 2.0f) / 2.0f;
        } else {  // back
            index += 1u;
            u = fmodf(1.0f - p.x, 2.0f) / 2.0f;
            v = fmodf(p.y + 1.0f, 2.0f) / 2.0f;
        }
    } else if (mapping == RTC_MAP_SPHERICAL) {  // uv.rs:93-105
        u = u_from_azimuth(p);
        const float radius = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
        const float phi = acosf_glibc(p.y / radius);
        v = 1.0f - phi * UV_FRAC_1_PI;
    } else if (mapping == RTC_MAP_PLANAR) {  // uv.rs:182-186
        u = rem_euclid_f32(p.x, 1.0f);
        v = rem_euclid_f32(p.z, 1.0f);
    } else {  // cylindrical, uv.rs:190-197
        u = u_from_azimuth(p);
        v = rem_euclid_f32(p.y, 2.0f * UV_PI) * UV_FRAC_1_2PI;
    }
    return uv_color_at(S, index, u, v);
}
DI V3 pattern_color_at_world(const SceneSoA& S, float4 pa, float4 pb, V3 pt) {
    const uint32_t kind = __float_as_uint(pa.w);
    if (kind >= RTC_PATTERN_TEXTURE_MAP) return texture_color_at_world(S, kind, __float_as_uint(pb.x), __float_as_uint(pb.y), pt);
    const V3 a = v3(pa.x, pa.y, pa.z), b = v3(pb.x, pb.y, pb.z);
    if (kind == RTC_PATTERN_GRADIENT) {
        float fraction = pt.x - floorf(pt.x);
        return a + (b * fraction);
    }
    if (kind == RTC_PATTERN_SINE2D) {
        float cosine = rtc_cosf_dev(pt.x + pt.z);
        float fraction = (-cosine + 1.0f) / 2.0f;
        return a + (b * fraction);
    }
    float key = pt.x;  // RTC_PATTERN_STRIPES
    if (kind == RTC_PATTERN_RINGS) key = sqrtf(pt.x * pt.x + pt.z * pt.z);
    if (kind == RTC_PATTERN_CHECKERS) key = fabsf(pt.x) + fabsf(pt.y) + fabsf(pt.z);
    return (rust_f32_as_i32(floorf(key)) & 1) == 0 ? a : b;
}
// Pattern::color_at_object (pattern.rs:15-19): world -> object -> pattern space.  Both matrices are affine
// (checked on the host), so the w component stays exactly 1.
DI V3 pattern_color_at_object(const SceneSoA& S, const float4* __restrict__ pat, const Obj& rec, V3 world_point) {
    const float4 pa = pat[0], pb = pat[1], r0 = pat[2], r1 = pat[3], r2 = pat[4];
    V3 op = obj_point(rec, world_point);
    V3 pp = v3(r0.x * op.x + r0.y * op.y + r0.z * op.z + r0.w, r1.x * op.x + r1.y * op.y + r1.z * op.z + r1.w,
               r2.x * op.x + r2.y * op.y + r2.z * op.z + r2.w);
    return pattern_color_at_world(S, pa, pb, pp);
}

// light/phong_lighting.rs:12-63; `material_color` is material.color or the pattern's colour (:24-27)
DI V3 phong(const SceneHdr& H, V3 material_color, float4 ma, float4 mb, V3 p, V3 eye, V3 n, float light_intensity) {
    const V3 li = v3(H.li[0], H.li[1], H.li[2]);
    V3 effective = material_color * li;
    V3 ambient = effective * ma.w;
    if (light_intensity == 0.0f) return ambient;
    V3 to_light = norm3(v3(H.lpos[0], H.lpos[1], H.lpos[2]) - p);
    float light_normal_cosine = dot3(to_light, n);
    V3 diffuse = v3(0.0f, 0.0f, 0.0f), specular = v3(0.0f, 0.0f, 0.0f);
    if (!(light_normal_cosine < 0.0f)) {
        diffuse = effective * mb.x * light_normal_cosine;
        V3 surface_reflection = reflect3(-to_light, n);
        float reflection_eye_cosine = dot3(surface_reflection, eye);
        if (!(reflection_eye_cosine <= 0.0f)) {
            // A material without a highlight (specular == 0: every object of the soft_shadows demo, the floor of most
            // scenes): `light.intensity() * 0 * factor` is the zero `intensity * 0` again for any finite positive factor,
            // and powf(c, s) is one for c in (0, 1 + 1e-6] -- a cosine of two unit vectors -- and 0 <= s <= 1e6.  The
            // power (some 25 f64 operations) is then not evaluated; a scene-compiled kernel of a world without any
            // highlight does not contain it at all.
            const bool no_highlight = !ANY_SPECULAR || (mb.y == 0.0f && mb.z >= 0.0f && mb.z <= 1e6f);
            float factor = 1.0f;
            if (ANY_SPECULAR && !no_highlight) factor = rtc_powf_dev(reflection_eye_cosine, mb.z);
            specular = li * mb.y * factor;
        }
    }
    return ambient + (diffuse + specular) * light_intensity;
}

// The hit object's records for shading.  `ob` differs from lane to lane, so `S.geo[ob]` is a gather through the vector
// memory path -- two dependent round trips per shade point (geometry for the normal, then the material).  In an
// unrolled kernel of a few objects every record is a wave-uniform scalar load the intersection loops hold in SGPRs
// anyway: take them all and keep the one whose index matches (-DRTC_SPEC_SELECT=1).
// (The values pass through readfirstlane -- a no-op for a scalar-loaded value -- because the compiler otherwise turns
// "select between two loaded values" back into "load from a selected address", i.e. into the gather this replaces.)
DI float uniform_value(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
DI float4 uniform_value(float4 x) { return make_float4(uniform_value(x.x), uniform_value(x.y), uniform_value(x.z), uniform_value(x.w)); }
DI float4 pick(bool mine, float4 a, float4 b) { return make_float4(mine ? a.x : b.x, mine ? a.y : b.y, mine ? a.z : b.z, mine ? a.w : b.w); }
template <int NOBJ>
DI Obj select_obj(const SceneSoA& S, int ob) {
    Obj r;
    r.geo = r.off0 = r.off1 = r.off2 = r.trn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    r.bits = 0u;
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)(NOBJ > 0 ? NOBJ : 1); i++) {
        const Obj t = load_obj_static<false>(S, i);
        const bool mine = ob == (int)i;
        r.geo = pick(mine, uniform_value(t.geo), r.geo);
        r.trn = pick(mine, uniform_value(t.trn), r.trn);
        if (mine) r.bits = t.bits;
        if (!(t.bits & SHAPE_DIAG) || (t.bits & SHAPE_KIND_MASK) == RTC_CYLINDER || (t.bits & SHAPE_KIND_MASK) == RTC_CONE) {  // compile-time in these kernels
            r.off0 = pick(mine, uniform_value(t.off0), r.off0);
            r.off1 = pick(mine, uniform_value(t.off1), r.off1);
            r.off2 = pick(mine, uniform_value(t.off2), r.off2);
        }
    }
    return r;
}
template <int NOBJ>
DI void select_material(const SceneSoA& S, int ob, float4& ma, float4& mb, float4& mc) {
    ma = mb = mc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)(NOBJ > 0 ? NOBJ : 1); i++) {
        const bool mine = ob == (int)i;
        ma = pick(mine, uniform_value(S.mat_a[i]), ma);
        mb = pick(mine, uniform_value(S.mat_b[i]), mb);
        mc = pick(mine, uniform_value(S.mat_c[i]), mc);
    }
}

// n1/n2 of precompute_values (world.rs:235-263) without the sorted list.
// Every intersection listed before the hit has t < 0 (the hit is the first
// non-negative minimum of a stably sorted list).  Walking those toggles each
// object in/out of an insertion-ordered set; an object ends up inside iff it
// has an odd number of negative intersections, and its insertion slot is that
// of its largest negative t (ties between objects: object order).  So the
// "innermost container" is the odd-parity object with the largest
// (t_max_negative, index); toggling the hit object then gives n2.
// (Tree kernels with several lanes per pixel: the lanes split the leaves -- for_each_leaf_shared -- and merge their two
// best containers at the end; an object is examined by exactly one lane, so the merged lists hold no duplicates.)
template <int NOBJ>
DI void refraction_indices(const SceneHdr& H, const SceneSoA& S, V3 o, V3 d, int hit_obj, float& n1, float& n2, const Counters& cnt) {
    WorldRay wr = world_ray<NOBJ>(H, o, d);
    wr.limit = 0.0f;             // only intersections behind the origin (t < 0) matter here
    float t1 = 0.0f, t2 = 0.0f;  // best and runner-up container keys
    int c1 = -1, c2 = -1;
    bool hit_inside = false;
    // the later object (in list order) wins ties: it sorts after.  Flat loops visit in list order; tree walks
    // compare indices explicitly (see nearest_hit).
    auto after = [&](float ta, int ia, float tb, int ib) {  // does (ta, ia) sort after (tb, ib)?
        if constexpr (NOBJ < 0) return ta > tb || (ta == tb && ia > ib);
        else return ta >= tb;
    };
    auto offer = [&](float t, int c) {  // a container candidate
        if (c1 < 0 || after(t, c, t1, c1)) {
            t2 = t1;
            c2 = c1;
            t1 = t;
            c1 = c;
        } else if (c2 < 0 || after(t, c, t2, c2)) {
            t2 = t;
            c2 = c;
        }
    };
    auto per_object = [&](uint32_t i, const Obj& ob, auto lane_idx) {
        V3 po = obj_point(ob, o);
        V3 pd = obj_vector(ob, d);
        int negatives = 0;
        float tmax = 0.0f;
        local_intersect<false, decltype(lane_idx)::value>(ob.bits, ob.min_y(), ob.max_y(), S.tri, i, po, pd, [&](float t) {
            if (t < 0.0f) {
                if (negatives == 0 || t > tmax) tmax = t;
                negatives++;
            }
        });
        if (negatives & 1) {
            if ((int)i == hit_obj) hit_inside = true;
            offer(tmax, (int)i);
        }
    };
    using LaneIdx = BoolConstant<true>;
    using UniformIdx = BoolConstant<false>;
    bool walked = false;
    if constexpr (NOBJ < 0 && Counters::SHARE_LANES && SHARED_WALK && SHARED_WALK_N12)
      if (cnt.share_log2() != 0u) {
        walked = true;
        for_each_leaf_shared(
            H, S, wr, cnt,
            [&](uint32_t i) {
                Obj ob = load_obj(S, i);
                ob.bits = spec_bits(i, ob.bits);
                per_object(i, ob, LaneIdx());
            },
            [&](uint32_t i, float t) {  // a triangle has one intersection: behind the origin it is an odd count
                if (t < 0.0f) {
                    if ((int)i == hit_obj) hit_inside = true;
                    offer(t, (int)i);
                }
            },
            [&]() {});
        for (uint32_t m = 1u; m < (1u << cnt.share_log2()); m <<= 1) {
            const float pt1 = __shfl_xor(t1, (int)m, 64), pt2 = __shfl_xor(t2, (int)m, 64);
            const int pc1 = __shfl_xor(c1, (int)m, 64), pc2 = __shfl_xor(c2, (int)m, 64);
            if (pc1 >= 0) offer(pt1, pc1);
            if (pc2 >= 0) offer(pt2, pc2);
            const int partner_inside = __shfl_xor((int)hit_inside, (int)m, 64);  // (every lane must take part: no short circuit)
            hit_inside = hit_inside || partner_inside != 0;
        }
      }
    if (!walked) for_each_object<NOBJ>(H, S, wr, [&](uint32_t i) { per_object(i, load_obj_static<NOBJ <= 0>(S, i), UniformIdx()); });
    const float vacuum = 1.0f;  // REFRACTION_VACCUM, constants.rs:6
    auto index_of = [&](int c) {  // Material.refractive_index of object c
        if constexpr (SELECT_RECORDS && NOBJ > 0) {
            float r = 0.0f;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)(NOBJ > 0 ? NOBJ : 1); i++) r = c == (int)i ? uniform_value(S.mat_c[i].y) : r;
            return r;
        } else {
            return S.mat_c[c].y;
        }
    };
    n1 = c1 >= 0 ? index_of(c1) : vacuum;
    if (!hit_inside) {
        n2 = index_of(hit_obj);  // entering: the hit object becomes the innermost container
    } else if (c1 == hit_obj) {
        n2 = c2 >= 0 ? index_of(c2) : vacuum;
    } else {
        n2 = n1;
    }
}

// world.rs:285-303
DI float schlick(V3 eye, V3 n, float n1, float n2) {
    float cosine = dot3(eye, n);
    if (n1 > n2) {
        float r = n1 / n2;
        float sin2 = r * r * (1.0f - cosine * cosine);
        if (sin2 > 1.0f) return 1.0f;
        cosine = sqrtf(1.0f - sin2);
    }
    float q = (n1 - n2) / (n1 + n2);
    float r0 = q * q;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x4 = x2 * x2;
    float x5 = x * x4;  // powi(5)
    return r0 + (1.0f - r0) * x5;
}

// Per-lane parking space in LDS for the state that must survive the light-sampling loop but is not
// used inside it (ray, hit, normal, recursion bookkeeping).  Parked there it costs no VGPRs during
// the 100-sample loop and no scratch (HBM-side) traffic; slot k of lane t lives at lds[k*stride + t],
// so a wave's accesses are consecutive dwords (conflict-free).
// (tree kernels park two more words: the hit's n1 / n2, known when the nearest-hit walk ends -- nearest_hit_and_containers)
#if defined(RTC_SPEC_LIST) && RTC_SPEC_NOBJ >= 0
constexpr int STASH_N12 = 0;
#else
constexpr int STASH_N12 = 2;
#endif
constexpr int STASH_N12_SLOT0 = RTC_DEEP_STACK ? 14 : 13;
constexpr int STASH_SLOTS = STASH_N12_SLOT0 + STASH_N12;
#ifdef RTC_NO_MERGED_N12  // development (A/B): the containers from a second walk (refraction_indices), as up to round 2
constexpr bool MERGED_N12 = false;
#else
constexpr bool MERGED_N12 = true;
#endif
// COMPACT frames (kernels that select the hit object's records from scalar loads, -DRTC_SPEC_SELECT=1: one or two
// objects): a suspended shade_hit keeps the object's INDEX instead of its two material coefficients and reads them again
// when a child returns -- a select between values the wave holds anyway.  The reflection half of a frame is then 5 dwords:
// five levels of it are 25.6 KB of LDS per workgroup, SIX workgroups per CU instead of five (the occupancy the kernel is
// compiled for), and the refraction half 6 dwords of scratch instead of 7.
#if defined(RTC_SPEC_LIST) && defined(RTC_SPEC_SELECT) && RTC_SPEC_SELECT && !RTC_DEEP_STACK && !defined(RTC_NO_COMPACT_FRAMES)  // (the last: development, A/B)
#define RTC_COMPACT_FRAMES 1
#else
#define RTC_COMPACT_FRAMES 0
#endif
constexpr int FRAME_LDS_DWORDS = RTC_COMPACT_FRAMES ? 5 : 6;
constexpr int FRAME_LDS_SLOT0 = USE_STASH ? STASH_SLOTS : 0;                  // recursion frames kept in LDS come after the parking slots
constexpr int LDS_SLOTS = FRAME_LDS_SLOT0 + FRAME_LDS_DWORDS * LDS_FRAME_LEVELS > 0 ? FRAME_LDS_SLOT0 + FRAME_LDS_DWORDS * LDS_FRAME_LEVELS : 1;
struct LaneStash {
    float* base;
    uint32_t stride;
    DI void put(int k, float v) const { base[k * stride] = v; }
    DI void putu(int k, uint32_t v) const { base[k * stride] = __uint_as_float(v); }
    DI float get(int k) const { return base[k * stride]; }
    DI uint32_t getu(int k) const { return __float_as_uint(base[k * stride]); }
};

// One suspended shade_hit (world.rs:62-86) waiting for a child colour.
// Split in two so that the common frame -- a mirror-like hit waiting for its reflection only -- moves 6 dwords
// instead of 13: the refraction half is written and read only when there is a refraction child.
struct Frame {
    V3 acc;       // surface colour, later surface + reflected[*R]
#if !RTC_COMPACT_FRAMES
    float reflective;
#endif
    float R;
    uint32_t flags;  // bit0: waiting for the refraction child; bit1: has refraction child; bit2: Schlick; COMPACT: bits 8.. the hit object
#if RTC_DEEP_STACK
    uint32_t path;   // the suspended call's own path code
#endif
};
struct FrameRefr {
    V3 ro, rd;    // pending refraction ray (under_point, direction)
#if !RTC_COMPACT_FRAMES
    float transparency;
#endif
};
enum { F_WAIT_REFR = 1, F_HAS_REFR = 2, F_SCHLICK = 4 };

// The post-order stack of color_at: one Frame (+ FrameRefr) per suspended shade_hit, at most `depth` of them.
// A per-lane array indexed by a per-lane stack pointer lives in scratch memory: every push and pop is a round trip
// through the vector memory path, and with a quarter of a million waves each owning 28 KB of it the lines do not stay
// in L2 -- on the glass-and-mirror scene (BASELINE C4) that was 2.5 GB of L2<->fabric traffic per 0.2 GB frame with 63 %
// of the wave-cycles spent waiting.  So the first STACK_REG_LEVELS levels are kept in registers instead: level u is
// the struct `reg[u]`, only ever indexed with constants, and a push / pop visits the levels some lane of the wave is
// at (wave-uniform test per level, then a per-lane select) -- typically one or two.  Deeper levels (depth > levels, or
// a kernel compiled without register levels) use the scratch arrays as before.  Which of the two holds a frame
// changes nothing about the values stored: the image cannot differ (tests/test_gpu_parity.py, every level count).
// (Two objects, because the compiler only turns a local aggregate into registers when EVERY access to it has a
// constant offset: the dynamically indexed scratch levels must not share an allocation with the register levels.)
struct FrameRegs {
    static constexpr int R = STACK_REG_LEVELS;
    Frame f[R > 0 ? R : 1];
    FrameRefr r[R > 0 ? R : 1];
};
struct FrameMem {
    static constexpr int M = RTC_STACK_DEPTH - STACK_REG_LEVELS;
    Frame f[M > 0 ? M : 1];
    FrameRefr r[M > 0 ? M : 1];
};
// A third home for the reflection half of a frame (6 dwords): the workgroup's LDS, for the first LDS_FRAME_LEVELS levels
// (-DRTC_SPEC_LDS_FRAMES).  A point-light kernel that no longer parks state there (RTC_SPEC_STASH=0) has the LDS free:
// five levels are 30 dwords per lane, 30 KB per workgroup, five workgroups per CU -- and the mirror floor of the
// glass-and-mirror scene, whose frames are reflection-only, stops writing its recursion to memory.
struct FrameStack {
    static constexpr int R = FrameRegs::R, M = FrameMem::M, L = LDS_FRAME_LEVELS;
    FrameRegs& reg;
    FrameMem& mem;
    LaneStash lds;  // slots FRAME_LDS_SLOT0 + 6 * level + k of this lane
    DI bool in_lds(int sp) const { return L > 0 && sp >= R && sp < R + L; }
    DI void put(int sp, const Frame& f) const {
#pragma unroll
        for (int u = 0; u < R; u++)
            if (__any(sp == u)) {  // wave-uniform: some lane is at this level
                if (sp == u) reg.f[u] = f;
            }
        if (in_lds(sp)) {
            const int b = FRAME_LDS_SLOT0 + FRAME_LDS_DWORDS * (sp - R);
            lds.put(b, f.acc.x), lds.put(b + 1, f.acc.y), lds.put(b + 2, f.acc.z), lds.put(b + 3, f.R), lds.putu(b + 4, f.flags);
#if !RTC_COMPACT_FRAMES
            lds.put(b + 5, f.reflective);
#endif
            return;
        }
        if (M > 0 && sp >= R) mem.f[sp - R] = f;
    }
    DI void put_refr(int sp, const FrameRefr& f) const {
#pragma unroll
        for (int u = 0; u < R; u++)
            if (__any(sp == u)) {
                if (sp == u) reg.r[u] = f;
            }
        if (M > 0 && sp >= R) mem.r[sp - R] = f;
    }
    DI Frame get(int sp) const {
        Frame f = {};
#pragma unroll
        for (int u = 0; u < R; u++)
            if (__any(sp == u)) {
                if (sp == u) f = reg.f[u];
            }
        if (in_lds(sp)) {
            const int b = FRAME_LDS_SLOT0 + FRAME_LDS_DWORDS * (sp - R);
            f.acc = v3(lds.get(b), lds.get(b + 1), lds.get(b + 2));
            f.R = lds.get(b + 3), f.flags = lds.getu(b + 4);
#if !RTC_COMPACT_FRAMES
            f.reflective = lds.get(b + 5);
#endif
            return f;
        }
        if (M > 0 && sp >= R) f = mem.f[sp - R];
        return f;
    }
    DI FrameRefr get_refr(int sp) const {
        FrameRefr f = {};
#pragma unroll
        for (int u = 0; u < R; u++)
            if (__any(sp == u)) {
                if (sp == u) f = reg.r[u];
            }
        if (M > 0 && sp >= R) f = mem.r[sp - R];
        return f;
    }
};

// World::color_at (world.rs:88-101) with reflected_color / refracted_color
// recursion (world.rs:121-162) unrolled into an explicit post-order stack.
// `path` is the jitter path code: 1 at the root, 2p for the reflection child
// of p, 2p+1 for its refraction child.
template <int NOBJ, bool SIMPLE>
DI V3 color_at(const SceneHdr& H, const SceneSoA& S, V3 o, V3 d, int depth, uint32_t pixel, Counters& cnt,
               const LaneStash stash) {
    FrameRegs stack_regs;
    FrameMem stack_mem;
    const FrameStack stack = {stack_regs, stack_mem, stash};
    int sp = 0;
    int rem = depth;
    uint32_t path = 1;
    V3 ret = v3(0.0f, 0.0f, 0.0f);
    for (;;) {
        // ---------------- color_at(ray(o, d), rem)
        cnt.rays += cnt.lead();
        Hit h;
        float kt1 = 0.0f, kt2 = 0.0f;  // tree kernels: the ray's containers, from the same walk (nearest_hit_and_containers)
        int kc1 = -1, kc2 = -1;
        bool k_inside = false;
        constexpr bool TREE_N12 = NOBJ < 0 && ANY_REFR && MERGED_N12;
        if constexpr (TREE_N12) h = nearest_hit_and_containers<NOBJ>(H, S, o, d, cnt, kt1, kc1, kt2, kc2, k_inside);
        else h = nearest_hit<NOBJ, true>(H, S, o, d, cnt);
        bool descend = false;
        ret = v3(0.0f, 0.0f, 0.0f);
        if (h.obj >= 0) {
            // precompute_values, world.rs:212-233.  Only what the light sampling needs is computed before
            // it (point, normal, over_point); everything else is (re)derived afterwards so that it is not
            // live across the 100-sample loop -- register pressure there decides occupancy.
            int ob = h.obj;
            V3 n;
            bool inside;
            V3 over_point;
            float tree_n1 = 1.0f, tree_n2 = 1.0f;  // n1 / n2 of precompute_values (world.rs:235-263), for a transparent hit of a tree kernel
            if constexpr (TREE_N12) {
                const float4 mc0 = S.mat_c[ob];
                if (mc0.x != 0.0f) {  // (what refraction_indices derives from the same containers)
                    tree_n1 = kc1 >= 0 ? S.mat_c[kc1].y : 1.0f;  // REFRACTION_VACCUM, constants.rs:6
                    tree_n2 = !k_inside ? mc0.y : kc1 == ob ? (kc2 >= 0 ? S.mat_c[kc2].y : 1.0f) : tree_n1;
                }
            }
            {
                Obj rec = SELECT_RECORDS && NOBJ > 0 ? select_obj<NOBJ>(S, ob) : load_obj(S, ob);
                V3 point = o + d * h.t;
                V3 op = obj_point(rec, point);
                n = obj_normal_to_world(rec, local_normal(rec.bits & SHAPE_KIND_MASK, rec.min_y(), rec.max_y(), S.tri + 3 * ob, op));
                inside = dot3(n, -d) < 0.0f;
                if (inside) n = -n;
                over_point = point + n * SELF_EPS;
            }

            // shade_hit, world.rs:62-86
            cnt.shaded += cnt.lead();
            // park everything the sampling loop does not touch (a kernel compiled for a point light has no such loop --
            // one shadow ray -- and keeps the state where it is: -DRTC_SPEC_STASH=0)
            float li;
            if constexpr (!USE_STASH) {
                li = intensity_at<NOBJ, SIMPLE>(H, S, over_point, pixel, path, cnt);
            } else {
            stash.put(0, o.x), stash.put(1, o.y), stash.put(2, o.z);
            stash.put(3, d.x), stash.put(4, d.y), stash.put(5, d.z);
            stash.put(6, n.x), stash.put(7, n.y), stash.put(8, n.z);
            stash.put(9, h.t);
            stash.putu(10, (uint32_t)ob | (inside ? 0x80000000u : 0u));
            stash.putu(11, pixel);
#if RTC_DEEP_STACK
            stash.putu(12, path);
            stash.putu(13, (uint32_t)rem | ((uint32_t)sp << 8) | ((uint32_t)depth << 16));
#else
            stash.putu(12, path | ((uint32_t)rem << 16) | ((uint32_t)sp << 20) | ((uint32_t)depth << 24));
#endif
            if constexpr (TREE_N12 && STASH_N12 == 2) stash.put(STASH_N12_SLOT0, tree_n1), stash.put(STASH_N12_SLOT0 + 1, tree_n2);
            asm volatile("" ::: "memory");
            li = intensity_at<NOBJ, SIMPLE>(H, S, over_point, pixel, path, cnt);
            asm volatile("" ::: "memory");
            o = v3(stash.get(0), stash.get(1), stash.get(2));
            d = v3(stash.get(3), stash.get(4), stash.get(5));
            n = v3(stash.get(6), stash.get(7), stash.get(8));
            h.t = stash.get(9);
            {
                const uint32_t w10 = stash.getu(10), w12 = stash.getu(12);
                ob = (int)(w10 & 0x7fffffffu);
                inside = (w10 >> 31) != 0;
                pixel = stash.getu(11);
#if RTC_DEEP_STACK
                const uint32_t w13 = stash.getu(13);
                path = w12;
                rem = (int)(w13 & 0xffu);
                sp = (int)((w13 >> 8) & 0xffu);
                depth = (int)(w13 >> 16);
#else
                path = w12 & 0xffffu;
                rem = (int)((w12 >> 16) & 0xfu);
                sp = (int)((w12 >> 20) & 0xfu);
                depth = (int)(w12 >> 24);
#endif
            }
            if constexpr (TREE_N12 && STASH_N12 == 2) tree_n1 = stash.get(STASH_N12_SLOT0), tree_n2 = stash.get(STASH_N12_SLOT0 + 1);
            }

            V3 eye = -d;
            V3 reflectv = reflect3(d, inside ? -n : n);  // world.rs:221 uses the normal before the inside flip
            V3 point = o + d * h.t;
            V3 under_point = point - n * SELF_EPS;
            float4 ma, mb, mc;
            if constexpr (SELECT_RECORDS && NOBJ > 0) {
                select_material<NOBJ>(S, ob, ma, mb, mc);
            } else {
                ma = S.mat_a[ob], mb = S.mat_b[ob], mc = S.mat_c[ob];
            }
            const float reflective = mb.w, transparency = mc.x;
            V3 material_color = v3(ma.x, ma.y, ma.z);
            if constexpr (!SIMPLE) {
                if (spec_has_patterns(H.has_patterns)) {
                    const float4* pat = S.pat + 5 * ob;
                    if (__float_as_uint(pat[0].w) != RTC_PATTERN_NONE)
                        material_color = pattern_color_at_object(S, pat, load_obj(S, ob), over_point);
                }
            }
            V3 surface = phong(H, material_color, ma, mb, over_point, eye, n, li);

            bool has_refl = ANY_REFL && !(reflective == 0.0f || rem < 1);  // world.rs:126
            bool has_refr = false;
            bool use_schlick = ANY_REFL && ANY_REFR && reflective > 0.0f && transparency > 0.0f;  // world.rs:80
            float R = 0.0f;
            V3 rdir = v3(0.0f, 0.0f, 0.0f);
            if (ANY_REFR && transparency != 0.0f) {
                float n1, n2;
                if constexpr (TREE_N12) n1 = tree_n1, n2 = tree_n2;
                else refraction_indices<NOBJ>(H, S, o, d, ob, n1, n2, cnt);
                if (use_schlick) R = schlick(eye, n, n1, n2);
                if (rem != 0) {  // refracted_color, world.rs:140-161
                    float n_ratio = n1 / n2;
                    float cos_i = dot3(eye, n);
                    float sin2 = n_ratio * n_ratio * (1.0f - cos_i * cos_i);
                    if (!(sin2 > 1.0f)) {
                        float cos_t = sqrtf(1.0f - sin2);
                        rdir = n * (n_ratio * cos_i - cos_t) - (eye * n_ratio);
                        has_refr = true;
                    }
                }
            }
            if (!has_refl && !has_refr) {
                const V3 black = v3(0.0f, 0.0f, 0.0f);
                ret = use_schlick ? surface + black * R + black * (1.0f - R) : surface + black + black;
            } else {
                Frame f;
                f.R = R;
                f.flags = (has_refr ? F_HAS_REFR : 0) | (use_schlick ? F_SCHLICK : 0);
#if RTC_COMPACT_FRAMES
                f.flags |= (uint32_t)ob << 8;
#else
                f.reflective = reflective;
#endif
#if RTC_DEEP_STACK
                f.path = path;
#endif
                if (has_refr) {
                    FrameRefr fr;
                    fr.ro = under_point;
                    fr.rd = rdir;
#if !RTC_COMPACT_FRAMES
                    fr.transparency = transparency;
#endif
                    stack.put_refr(sp, fr);
                }
                if (has_refl) {
                    f.acc = surface;
                    o = over_point;
                    d = reflectv;
                    path = path * 2u;
                } else {
                    const V3 black = v3(0.0f, 0.0f, 0.0f);
                    f.acc = use_schlick ? surface + black * R : surface + black;
                    f.flags |= F_WAIT_REFR;
                    o = under_point;
                    d = rdir;
                    path = path * 2u + 1u;
                }
                stack.put(sp++, f);
                rem--;
                descend = true;
            }
        }
        if (descend) continue;
        // ---------------- return `ret` to the suspended callers
        for (;;) {
            if (sp == 0) return ret;
            Frame f = stack.get(sp - 1);
            rem++;
#if RTC_DEEP_STACK
            path = f.path;
#else
            path >>= 1;
#endif
#if RTC_COMPACT_FRAMES
            // the suspended hit's material coefficients, selected again from the scalar-loaded records (select_material)
            float f_reflective = 0.0f, f_transparency = 0.0f;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)(NOBJ > 0 ? NOBJ : 1); i++) {
                const bool mine = (f.flags >> 8) == i;
                f_reflective = mine ? uniform_value(S.mat_b[i].w) : f_reflective;
                f_transparency = mine ? uniform_value(S.mat_c[i].x) : f_transparency;
            }
#else
            const float f_reflective = f.reflective;
#endif
            if (!(f.flags & F_WAIT_REFR)) {
                V3 reflected = ret * f_reflective;  // world.rs:131
                V3 partial = (f.flags & F_SCHLICK) ? f.acc + reflected * f.R : f.acc + reflected;
                if (ANY_REFR && (f.flags & F_HAS_REFR)) {
                    f.acc = partial;
                    f.flags |= F_WAIT_REFR;
                    stack.put(sp - 1, f);
                    const FrameRefr fr = stack.get_refr(sp - 1);
                    o = fr.ro;
                    d = fr.rd;
                    rem--;
                    path = path * 2u + 1u;
                    break;
                }
                const V3 black = v3(0.0f, 0.0f, 0.0f);
                ret = (f.flags & F_SCHLICK) ? partial + black * (1.0f - f.R) : partial + black;
                sp--;
            } else {
#if RTC_COMPACT_FRAMES
                V3 refracted = ret * f_transparency;  // world.rs:159-160
#else
                V3 refracted = ret * stack.get_refr(sp - 1).transparency;  // world.rs:159-160
#endif
                ret = (f.flags & F_SCHLICK) ? f.acc + refracted * (1.0f - f.R) : f.acc + refracted;
                sp--;
            }
        }
    }
}

// ============================================================================
//  Kernels
// ============================================================================
struct RenderArgs {
    SceneHdr hdr;
    SceneSoA soa;
    float* out;            // compact rows of this partition: [rows][width][3] ...
    uint8_t* out_u8;       // ... or, when not null, the same rows as the bytes Canvas::to_ppm prints (scale_color, canvas.rs:39-43)
    uint4* block_counts;   // one partial {rays, shaded hits, culled shadow rays, 0} per wave (4 per workgroup)
    unsigned long long* total;  // {rays, shaded hits, culled}: zeroed here, accumulated by sum_counts_kernel
    uint32_t rows;         // rows in `out`
    uint32_t band_rows, n_parts, part;
    int32_t depth;
    uint32_t share_log2;   // 2^share_log2 lanes per pixel (see Counters): > 0 only for area lights on small images
    // Block list: workgroup b renders the block tiles[b] = (s & 3) << 30 | (x0 / 4) << 16 | (s >> 2) << 15 | (y0 / 4) -- pixel origin
    // (x0, local row y0 < 2^17) and, in kernels compiled for lane sharing, ITS OWN lanes-per-pixel 2^s, s = 0 .. 4 -- instead of block
    // (blockIdx.x, blockIdx.y) of a regular grid.  For tree worlds with long leaf runs the host lists the blocks of image regions a mesh projects to first and
    // with more lanes per pixel, the rest after them with one (rtc_device.hip build_block_list): such a frame's time is that of
    // its slowest waves, so those start first and are cut up.  From a scene's second frame on, any list is made from the wave
    // times of the frame before (refine_block_list).  nullptr: regular grid.
    const uint32_t* tiles;
    // Kernels compiled for lane sharing; nullptr: not asked for.  Wave w of workgroup b (of a block list, or b = blockIdx.y *
    // gridDim.x + blockIdx.x) leaves its own running time here, [4 b + w], in ticks of the 100 MHz clock: what the host orders
    // and cuts the NEXT frames' list by.  (The other kernels' lists are ordered by the waves' work counts, block_counts.)
    uint32_t* wave_ticks;
    uint32_t blocks_y;  // regular grid: blocks rendered by one workgroup, stacked vertically (>= 1)
    // Regular grid, not 0: which block a workgroup renders is not (blockIdx.x, blockIdx.y) but a permutation of it within four
    // rows of the grid (an even number of columns, a multiple of four rows: the host pads).  Workgroups are started in index
    // order and dealt to the eight XCDs round robin; in image order XCD r gets every eighth block of a row, and all eight sweep
    // the same few rows at the same time.  Here workgroup j of a group of four rows renders row j mod 8 / 2, column 2 (j / 8) +
    // j mod 2: an XCD works along half a row of its own.  Measured as fixed block lists first (profiles/r03_ab_block_orders.txt:
    // first_textures 1.07 -> 0.92 ms, reflect_refract 0.98 -> 0.92, hexagons 0.50 -> 0.475, C3 0.897 -> 0.870; eight bands or 4 x 2
    // regions, one per XCD: 1.9 and 2.4 ms -- the XCDs' shares must be alike); rows still finish four at a time from the
    // top, which is all rtc_render_ex's progress words need.
    uint32_t swizzle;
    // regular grid: the launch covers the blocks from (block_x0, block_y0) on -- of a frame whose scene can only be seen
    // inside a rectangle of pixel columns [fill_x0, fill_x1) x local rows [fill_y0, fill_y1) only that rectangle is rendered
    // (rtc_device.hip: scene rectangle); fill_wg_rows rows of the grid, spread evenly among the rendering ones, are
    // workgroups that zero-fill everything outside it instead, fill_rows local rows each.
    uint32_t block_x0, block_y0;
    uint32_t fill_wg_rows, fill_rows, fill_x0, fill_x1, fill_y0, fill_y1;
    uint32_t fill_period;  // row j * fill_period of the grid is the j-th row of filling workgroups (j < fill_wg_rows), the others render
    // Progress reporting (rtc_render_ex, regular grid only; nullptr otherwise): the frame is ONE launch, and its rows leave
    // for the host while later rows are still being rendered.  A workgroup whose four waves have stored their pixels adds one
    // to its block row's counter; the workgroup that completes a block row adds one to the counter of the row's CHUNK
    // (chunk_block_rows consecutive block rows); the one that completes a chunk writes `epoch` into the chunk's word of
    // `done` -- page-locked host memory the caller polls before it starts that chunk's copy.
    // What makes the pixels be in memory before the word that announces them: in this mode they are stored WRITE-THROUGH
    // (system-scope stores: no dirty line is left in any XCD's L2), a wave counts itself done only after its stores have
    // been acknowledged (s_waitcnt vmcnt(0)), and the counters are device-scope atomics chained by what each returns.
    // (The textbook form -- release / acquire fences at device scope -- writes this XCD's whole L2 back per workgroup, the
    // recursion's scratch included: measured 0.97 -> 2.47 ms for the C3 frame.)  Counters are PROGRESS_STRIDE dwords apart.
    uint32_t* progress;       // [gridDim.y] block-row counters, then [n_chunks] chunk counters; zeroed before the launch
    uint32_t* done;           // [n_chunks], host memory
    uint32_t chunk_block_rows, epoch;
};
constexpr uint32_t PROGRESS_STRIDE = 16;

// One of the launch's first workgroups (render_body): zero the part outside the scene rectangle of its share of the rows.
// Memory-bound work running beside the arithmetic-bound rendering; 16-byte stores where rows and segments are aligned to
// that (widths that are multiples of 4; the rectangle's columns are multiples of 16 pixels).
DI void fill_outside(const RenderArgs& A, uint32_t fill_row) {
    const uint32_t w = fill_row * gridDim.x + blockIdx.x, W3 = A.hdr.width * 3u;
    const uint32_t r0 = w * A.fill_rows, r1 = min(A.rows, r0 + A.fill_rows);
    const bool wide = (A.hdr.width & 3u) == 0u && ((unsigned long)A.out & 15ul) == 0ul;  // wave-uniform
    for (uint32_t row = r0; row < r1; row++) {
        float* dst = A.out + (size_t)row * W3;
        const bool beside = row >= A.fill_y0 && row < A.fill_y1;  // a row the rectangle crosses: what is left and right of it
        const uint32_t e0 = (beside ? A.fill_x0 : A.hdr.width) * 3u, b1 = (beside ? min(A.fill_x1, A.hdr.width) : A.hdr.width) * 3u;
        if (wide) {
            float4* d4 = (float4*)dst;
            const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (uint32_t i = threadIdx.x; i < e0 / 4u; i += 256u) d4[i] = z;
            for (uint32_t i = b1 / 4u + threadIdx.x; i < W3 / 4u; i += 256u) d4[i] = z;
        } else {
            for (uint32_t i = threadIdx.x; i < e0; i += 256u) dst[i] = 0.0f;
            for (uint32_t i = b1 + threadIdx.x; i < W3; i += 256u) dst[i] = 0.0f;
        }
    }
}

// Camera::render (camera.rs:76-91): one lane per pixel, 8x8 pixel tile per
// wave, 2x2 waves per 256-thread workgroup.  NOBJ: see for_each_object.
#ifndef RTC_WAVES_PER_SIMD
#define RTC_WAVES_PER_SIMD 6
#endif
template <int NOBJ, bool SIMPLE>
DI void render_body(const RenderArgs& A) {
    const SceneHdr& H = A.hdr;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // Sample-parallel mode (small images under an area light, chosen by the host): a frame of a few thousand waves,
    // each working through 64 pixels x 100 shadow rays, leaves most of the chip idle for the length of one wave.
    // With 2^s lanes per pixel a wave takes a tile of 64 >> s pixels (8x8, 8x4, 4x4, 4x2) and every lane a 2^-s share of
    // each shade point's light cells; all other work is replicated across a pixel's lanes (same inputs, same bits).
    // Scene rectangle launches (the ahead-of-time kernels, and scene kernels compiled with -DRTC_SPEC_RECT=1: the few
    // extra argument loads and operations in front of every wave cost the frames of very short waves 6 - 10 % --
    // first_plane, first_patterns -- so kernels of scenes that have no use for it do not carry them): is this one of the
    // workgroups that zero-fill (wave-uniform; it renders nothing), and which block of the rendering grid is it otherwise?
#if !defined(RTC_SPEC_LIST) || (defined(RTC_SPEC_RECT) && RTC_SPEC_RECT)
    constexpr bool RECT_LAUNCH = true;
#else
    constexpr bool RECT_LAUNCH = false;
#endif
    // A workgroup of the regular grid renders `blocks_y` blocks, one below the other (host: frames whose waves are so
    // short -- C5: 95 % of 67 M pixels miss the scene's box -- that launching them is what the frame costs)
    // Compiled in only where the host asks for it (-DRTC_SPEC_BLOCKS_Y=1): the loop's carried state costs other kernels
    // registers (first_patterns: 6 -> 18 spilled VGPRs, +45 %).
#if defined(RTC_SPEC_BLOCKS_Y) && RTC_SPEC_BLOCKS_Y
    const uint32_t blocks_y = A.tiles != nullptr ? 1u : A.blocks_y;
#else
    constexpr uint32_t blocks_y = 1u;
#endif
    // Where this lane's pixel is.  (These kernels are compiled to the last register of their occupancy, and what is added in
    // front of color_at decides on which side of a cliff the allocation lands -- reflect_refract's kernel, 102 registers: 21
    // spilled with the block list's entry fetched by a scalar load, 69 with the same entry fetched by a vector load, 69 with a
    // clock read at the wave's start.  Hence: a plain load here, and wave times only in the kernels that cut their lists.)
    struct Where {
        uint32_t x, yl, sl, grid_y;
        bool fills;
    };
    auto where = [&](uint32_t rep) {
        Where w;
        w.sl = Counters::SHARE_LANES ? A.share_log2 : 0u;  // lanes per pixel (log2)
        w.fills = false;
        w.grid_y = blockIdx.y;
        uint32_t bx0, by0;  // pixel origin of this workgroup's block
        if (A.tiles != nullptr) {
            const uint32_t t = A.tiles[blockIdx.x];  // wave-uniform: a scalar load
            if (Counters::SHARE_LANES) w.sl = (t >> 30) | ((t >> 13) & 4u);  // (a kernel without lane sharing is only ever given lists of whole blocks)
            bx0 = ((t >> 16) & 0x3fffu) << 2;
            by0 = (t & 0x7fffu) << 2;
        } else {
            uint32_t block_x0 = 0u, block_y0 = 0u;
            if constexpr (RECT_LAUNCH) {
                block_x0 = A.block_x0, block_y0 = A.block_y0;
                if (A.fill_wg_rows != 0u) {
                    const uint32_t j = blockIdx.y / A.fill_period;
                    w.fills = j < A.fill_wg_rows && blockIdx.y == j * A.fill_period;
                    if (w.fills) w.grid_y = j;
                    else w.grid_y = blockIdx.y - min(A.fill_wg_rows, j + 1u);
                }
            }
            uint32_t gx = blockIdx.x, gy = w.grid_y;
            if (A.swizzle != 0u) {  // (RenderArgs::swizzle; never together with a rectangle, fills or several blocks per workgroup)
                const uint32_t j = (gy & 3u) * gridDim.x + gx, r = j & 7u;
                gy = (gy & ~3u) + (r >> 1);
                gx = 2u * (j >> 3) + (r & 1u);
            }
            bx0 = (gx + block_x0) << (4u - (w.sl >> 1));
            by0 = (gy * blocks_y + rep + block_y0) << (4u - ((w.sl + 1u) >> 1));
        }
        const uint32_t q = lane >> w.sl;  // q: the pixel's slot in the wave's tile
        const uint32_t tw_log2 = 3u - (w.sl >> 1), th_log2 = 3u - ((w.sl + 1u) >> 1);
        w.x = bx0 + ((wave & 1u) << tw_log2) + (q & ((1u << tw_log2) - 1u));
        w.yl = by0 + ((wave >> 1) << th_log2) + (q >> tw_log2);
        return w;
    };
    const Where w0 = where(0u);
    const uint32_t sl = w0.sl;
    Counters cnt = {0u, 0u, sl};
    const bool timed = Counters::SHARE_LANES && A.wave_ticks != nullptr;  // wave-uniform
    uint32_t ticks0 = 0u;
    if (timed) ticks0 = (uint32_t)wall_clock64();
#ifdef RTC_DEBUG_TIMELINE  // development (tools/wave_timeline.py): the frame holds each wave's start / end / place instead of colours
    const uint32_t t_start = (uint32_t)wall_clock64();
#endif
#ifdef RTC_DEBUG_STEPS
    dbg_steps()[0] = dbg_steps()[1] = dbg_steps()[2] = dbg_steps()[3] = 0u;
#endif
    __shared__ float stash_lds[LDS_SLOTS * 256];
    const LaneStash stash = {stash_lds + threadIdx.x, 256u};
    __shared__ uint32_t waves_done;  // progress reporting: how many of this workgroup's waves have stored their pixels
    if (A.progress != nullptr) {     // wave-uniform (a kernel argument)
        if (threadIdx.x == 0) waves_done = 0u;
        __syncthreads();
    }
    if constexpr (RECT_LAUNCH) {
        if (w0.fills) fill_outside(A, w0.grid_y);
    }
    for (uint32_t rep = 0; rep < blocks_y; rep++) {
    const Where w = where(rep);
    const uint32_t x = w.x, yl = w.yl;
    if ((!RECT_LAUNCH || !w.fills) && x < H.width && yl < A.rows) {
        // compact local row -> global row of the image
        const uint32_t band = yl / A.band_rows;
        const uint32_t y = (band * A.n_parts + A.part) * A.band_rows + (yl - band * A.band_rows);
        V3 col = v3(0.0f, 0.0f, 0.0f);
        // camera.rs:80-81: `0..height-1` x `0..width-1` -- the last row and column stay black
        if (x < H.width - 1u && y < H.height - 1u) {
            // ray_for_pixel, camera.rs:60-74
            float x_offset = ((float)x + 0.5f) * H.pixel_size;
            float y_offset = ((float)y + 0.5f) * H.pixel_size;
            float world_x = H.half_w - x_offset;
            float world_y = H.half_h - y_offset;
            const float* c = H.cam;
            V3 pixel = {c[0] * world_x + c[1] * world_y + c[2] * -1.0f + c[3],
                        c[4] * world_x + c[5] * world_y + c[6] * -1.0f + c[7],
                        c[8] * world_x + c[9] * world_y + c[10] * -1.0f + c[11]};
            V3 origin = v3(H.cam_origin[0], H.cam_origin[1], H.cam_origin[2]);
            // A box around everything a primary ray could hit (SceneHdr::scene_box, built and padded on the host when every
            // top-level object is bounded): a ray that misses it -- tested with the unnormalised direction and approximate
            // reciprocals, which move the ray by parts in 1e6 against a padding of 10 % of the box / 1 % of the camera's
            // distance -- hits nothing, so color_at would return black after one counted ray.  Saves the exact square root
            // and divisions of norm() and the walk on every such pixel (95 % of C5's).
            bool sees_nothing = false;
            if (H.has_scene_box) {
                const V3 du = pixel - origin;
                const V3 iu = v3(__builtin_amdgcn_rcpf(du.x), __builtin_amdgcn_rcpf(du.y), __builtin_amdgcn_rcpf(du.z));
                float tmin;
                sees_nothing = !aabb_hit(origin, iu, make_float4(H.scene_box[0], H.scene_box[1], H.scene_box[2], 0.0f),
                                         make_float4(H.scene_box[3], H.scene_box[4], H.scene_box[5], 0.0f), tmin);
            }
            if (sees_nothing) {
                cnt.rays += cnt.lead();
            } else {
                V3 direction = norm3(pixel - origin);
                col = color_at<NOBJ, SIMPLE>(H, A.soa, origin, direction, A.depth, y * H.width + x, cnt, stash);
            }
        }
#ifdef RTC_DEBUG_TIMELINE
        col = v3(__uint_as_float(t_start), __uint_as_float((uint32_t)wall_clock64()),
                 __uint_as_float((__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16)));
#endif
#ifdef RTC_DEBUG_STEPS  // group tests | exact tests << 20, leaf box tests, entries the WAVE stepped through
        col = v3(__uint_as_float(dbg_steps()[0] | dbg_steps()[2] << 20), __uint_as_float(dbg_steps()[1]), __uint_as_float(dbg_steps()[3]));
#endif
        // The canvas store (canvas.rs:26-43: row-major RGB): three dword stores per lane at a 12-byte stride.  (Round 4 measured the
        // obvious alternative -- the wave's 8 x 8 tile transposed through LDS and stored as 48 sixteen-byte pieces, six per row: C3 +3 %,
        // C5 +3 %, first_plane +6 %, profiles/r04_ab_wide_stores.txt.  The memory system merges the partial lines; the transpose costs
        // an LDS round trip and registers in front of every wave's exit.)
        if (cnt.lead()) {
            const bool through = A.progress != nullptr;  // wave-uniform: write-through stores (RenderArgs::progress)
            const Where ws = where(rep);
            if (A.out_u8 != nullptr) {  // wave-uniform: scale_color on the way out (the arithmetic of quantize_kernel)
                uint8_t* dst = A.out_u8 + ((size_t)ws.yl * H.width + ws.x) * 3;
                const uint8_t r = (uint8_t)fmaxf(fminf(col.x * 255.0f, 255.0f), 0.0f), g = (uint8_t)fmaxf(fminf(col.y * 255.0f, 255.0f), 0.0f),
                              b = (uint8_t)fmaxf(fminf(col.z * 255.0f, 255.0f), 0.0f);
                if (through) {
                    __hip_atomic_store(dst, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(dst + 1, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(dst + 2, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    dst[0] = r, dst[1] = g, dst[2] = b;
                }
            } else {
                float* dst = A.out + ((size_t)ws.yl * H.width + ws.x) * 3;
                if (through) {
                    __hip_atomic_store(dst, col.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(dst + 1, col.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(dst + 2, col.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    dst[0] = col.x;
                    dst[1] = col.y;
                    dst[2] = col.z;
                }
            }
        }
    }
    }
    // work statistics: wave reduce, then one partial per wave
    uint32_t rays = cnt.rays, shaded = cnt.shaded_count(), culled = cnt.culled_count();
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off, 64);
        shaded += __shfl_down(shaded, off, 64);
        culled += __shfl_down(culled, off, 64);
    }
    const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4u;
    // one partial per WAVE and no workgroup barrier: a wave that is done leaves (where neighbouring tiles differ a lot in
    // depth of recursion -- the edge of a glass ball -- the waiting waves were holding the slots of the next workgroup)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 3) A.total[threadIdx.x] = 0ull;  // for sum_counts_kernel's atomics
    if (lane == 0) A.block_counts[slot + wave] = make_uint4(rays, shaded, culled, 0u);
    if (timed && lane == 0) A.wave_ticks[slot + wave] = (uint32_t)wall_clock64() - ticks0;
    if (A.progress != nullptr) {  // see RenderArgs::progress
        // every store of this wave has been acknowledged -- and, being write-through, is in memory -- before the wave counts
        // itself done
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        uint32_t last = 0u;
        if (lane == 0) last = __hip_atomic_fetch_add(&waves_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 3u ? 1u : 0u;
        if (__builtin_amdgcn_readfirstlane((int)last) != 0 && lane == 0) {
            const uint32_t row = blockIdx.y;
            if (__hip_atomic_fetch_add(&A.progress[row * PROGRESS_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x) {
                const uint32_t ch = row / A.chunk_block_rows;
                const uint32_t in_chunk = min(A.chunk_block_rows, gridDim.y - ch * A.chunk_block_rows);
                if (__hip_atomic_fetch_add(&A.progress[(gridDim.y + ch) * PROGRESS_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == in_chunk)
                    __hip_atomic_store(&A.done[ch], A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

#ifdef RTC_SPEC_LIST
}  // namespace rtc
// The one kernel of a scene-specialised (hiprtc) build.
extern "C" __global__ __launch_bounds__(256, RTC_WAVES_PER_SIMD) void render_kernel_spec(rtc::RenderArgs A) {
    rtc::render_body<RTC_SPEC_NOBJ, RTC_SPEC_SIMPLE != 0>(A);
}
namespace rtc {
#else
template <int NOBJ, bool SIMPLE>
__global__ __launch_bounds__(256, RTC_WAVES_PER_SIMD) void render_kernel(RenderArgs A) {
    render_body<NOBJ, SIMPLE>(A);
}


// Workgroup b sums entries [b * SUM_COUNTS_SLICE, (b + 1) * SUM_COUNTS_SLICE) and adds its result to `total`, which the
// render kernel left zeroed (one workgroup took 62 us for the 262 144 partials of an 8192^2 frame).
constexpr uint32_t SUM_COUNTS_SLICE = 8192;
// `extra_rays`: rays of pixels the launch did not cover (each such pixel misses the scene's box: one ray, no hit).
__global__ __launch_bounds__(1024) void sum_counts_kernel(const uint4* __restrict__ block_counts, uint32_t n_all,
                                                          unsigned long long* __restrict__ total, unsigned long long extra_rays) {
    unsigned long long rays = (blockIdx.x == 0 && threadIdx.x == 0) ? extra_rays : 0ull, shaded = 0, culled = 0;
    const uint32_t n = min(n_all, (blockIdx.x + 1u) * SUM_COUNTS_SLICE);
    uint32_t i = blockIdx.x * SUM_COUNTS_SLICE + threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {  // four independent loads in flight per lane
        uint4 a = block_counts[i], b = block_counts[i + 1024], c = block_counts[i + 2048], d = block_counts[i + 3072];
        rays += (unsigned long long)a.x + b.x + c.x + d.x;
        shaded += (unsigned long long)a.y + b.y + c.y + d.y;
        culled += (unsigned long long)a.z + b.z + c.z + d.z;
    }
    for (; i < n; i += 1024) {
        uint4 c = block_counts[i];
        rays += c.x;
        shaded += c.y;
        culled += c.z;
    }
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off, 64);
        shaded += __shfl_down(shaded, off, 64);
        culled += __shfl_down(culled, off, 64);
    }
    __shared__ unsigned long long part[16][3];
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = rays;
        part[threadIdx.x >> 6][1] = shaded;
        part[threadIdx.x >> 6][2] = culled;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r = 0, sh = 0, cu = 0;
        for (int w = 0; w < 16; w++) {
            r += part[w][0];
            sh += part[w][1];
            cu += part[w][2];
        }
        atomicAdd(&total[0], r);
        atomicAdd(&total[1], sh);
        atomicAdd(&total[2], cu);
    }
}

// Scene tiles (rtc_device.hip ctx_render_slot): the tiles of the canvas no entry of the world projects to are black.  One
// workgroup per job {x0 | n << 16 (tiles), local row y0}: sixteen rows of n tiles' columns, zeroed with 16-byte stores where rows are
// aligned to that (f32 rows of widths that are multiples of four; byte rows always start on a multiple of 48 columns' worth).
__global__ __launch_bounds__(256) void fill_tiles_kernel(const uint2* __restrict__ jobs, uint32_t n_jobs, uint8_t* __restrict__ out, uint32_t width, uint32_t rows,
                                                         uint32_t bytes_per_pixel) {
  for (uint32_t j = blockIdx.x; j < n_jobs; j += gridDim.x) {  // (a few hundred workgroups share the jobs: they leave the chip's wave slots to the render kernel)
    const uint2 job = jobs[j];
    const uint32_t x0 = (job.x & 0xffffu) * 16u, x1 = min(width, x0 + (job.x >> 16) * 16u), y0 = job.y, y1 = min(rows, y0 + 16u);
    if (x0 >= x1) continue;
    const size_t row_bytes = (size_t)width * bytes_per_pixel;
    const uint32_t span = (x1 - x0) * bytes_per_pixel;  // bytes per row of this job
    const bool wide = (row_bytes & 15u) == 0u && ((unsigned long)out & 15ul) == 0ul && ((x0 * bytes_per_pixel) & 15u) == 0u && (span & 15u) == 0u;
    for (uint32_t y = y0; y < y1; y++) {
        uint8_t* dst = out + (size_t)y * row_bytes + (size_t)x0 * bytes_per_pixel;
        if (wide) {
            float4* d4 = (float4*)dst;
            for (uint32_t i = threadIdx.x; i < span / 16u; i += 256u) d4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            for (uint32_t i = threadIdx.x; i < span; i += 256u) dst[i] = 0;
        }
    }
  }
}

// canvas.rs:39-43
__global__ void quantize_kernel(const float* __restrict__ rgb, uint64_t n, uint8_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = fmaxf(fminf(rgb[i] * 255.0f, 255.0f), 0.0f);
        out[i] = (uint8_t)v;
    }
}

// ---- Canvas::to_ppm (canvas.rs:47-96) on the device -----------------------------------------------
// The P3 body is a stream of tokens (one per colour channel, 3*w per image row); every token emits its
// 1-3 digits followed by exactly ONE separator byte: ' ' while the current text line is shorter than 67
// characters, '\n' otherwise (write_rgb_separator, canvas.rs:47-55), and '\n' after a row's last token
// (canvas.rs:90-93).  So byte offsets are a plain prefix sum of (digits + 1); only the KIND of each
// separator depends on the running line length, a tiny sequential automaton per image row.
DI uint32_t ppm_channel(float c) {  // scale_color, canvas.rs:39-43
    float v = fmaxf(fminf(c * 255.0f, 255.0f), 0.0f);
    return (uint32_t)(uint8_t)v;
}
DI uint32_t ppm_digits(uint32_t v) { return 1u + (v >= 10u) + (v >= 100u); }

// Pass 1, one thread per image row: row text length, and one bit per token: 1 = its separator is '\n'.
__global__ void ppm_row_scan_kernel(const float* __restrict__ rgb, uint32_t w, uint32_t h,
                                    unsigned long long* __restrict__ row_len, uint32_t* __restrict__ sep_bits,
                                    uint32_t words_per_row) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= h) return;
    const uint32_t n_tok = 3u * w;
    const float* src = rgb + (size_t)row * n_tok;
    uint32_t* bits = sep_bits + (size_t)row * words_per_row;
    uint32_t line = 0, word = 0;
    unsigned long long len = 0;
    for (uint32_t k = 0; k < n_tok; k++) {
        const uint32_t d = ppm_digits(ppm_channel(src[k]));
        len += d + 1u;
        line += d;
        bool newline;
        if (k + 1u == n_tok) {
            newline = true;  // end of the image row
        } else if (line < 67u) {
            newline = false;
            line += 1u;
        } else {
            newline = true;
            line = 0u;
        }
        if (newline) word |= 1u << (k & 31u);
        if ((k & 31u) == 31u || k + 1u == n_tok) {
            bits[k >> 5] = word;
            word = 0u;
        }
    }
    row_len[row] = len;
}

// Pass 2: exclusive prefix sum of the row lengths (h entries, one workgroup), plus the total.
__global__ __launch_bounds__(1024) void ppm_row_offsets_kernel(unsigned long long* __restrict__ row_len, uint32_t h,
                                                               unsigned long long base,
                                                               unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[1024];
    const uint32_t per = (h + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per, e = b + per < h ? b + per : h;
    unsigned long long s = 0;
    for (uint32_t i = b; i < e; i++) s += row_len[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = base;
        for (int t = 0; t < 1024; t++) {
            unsigned long long v = part[t];
            part[t] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint32_t i = b; i < e; i++) {
        unsigned long long v = row_len[i];
        row_len[i] = run;
        run += v;
    }
}

// Pass 3, one wave per image row, 64 tokens per step: a wave-level prefix sum places each token's bytes.
__global__ __launch_bounds__(256) void ppm_emit_kernel(const float* __restrict__ rgb, uint32_t w, uint32_t h,
                                                       const unsigned long long* __restrict__ row_off,
                                                       const uint32_t* __restrict__ sep_bits, uint32_t words_per_row,
                                                       char* __restrict__ text) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= h) return;
    const uint32_t n_tok = 3u * w;
    const float* src = rgb + (size_t)row * n_tok;
    const uint32_t* bits = sep_bits + (size_t)row * words_per_row;
    unsigned long long base = row_off[row];
    for (uint32_t k0 = 0; k0 < n_tok; k0 += 64u) {
        const uint32_t k = k0 + lane;
        const bool active = k < n_tok;
        uint32_t v = 0, d = 0, nbytes = 0;
        bool newline = false;
        if (active) {
            v = ppm_channel(src[k]);
            d = ppm_digits(v);
            nbytes = d + 1u;
            newline = (bits[k >> 5] >> (k & 31u)) & 1u;
        }
        // inclusive wave scan of nbytes
        uint32_t incl = nbytes;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t up = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += up;
        }
        if (active) {
            char* dst = text + base + (incl - nbytes);
            uint32_t p = 0;
            if (v >= 100u) dst[p++] = (char)('0' + v / 100u);
            if (v >= 10u) dst[p++] = (char)('0' + (v / 10u) % 10u);
            dst[p++] = (char)('0' + v % 10u);
            dst[p] = newline ? '\n' : ' ';
        }
        base += __shfl(incl, 63, 64);
    }
}

// The batched entry points are test/utility paths: they use the generic loop.
__global__ void color_at_kernel(SceneHdr H, SceneSoA S, const float4* __restrict__ origins,
                                const float4* __restrict__ directions, uint32_t n, int depth,
                                float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Counters cnt = {0u, 0u};
    __shared__ float stash_lds[STASH_SLOTS * 64];
    const LaneStash stash = {stash_lds + threadIdx.x, 64u};
    float4 o = origins[i], d = directions[i];
    V3 c = H.n_trav ? color_at<-1, false>(H, S, v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), depth, i, cnt, stash)
                    : color_at<0, false>(H, S, v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), depth, i, cnt, stash);
    out[i * 3 + 0] = c.x;
    out[i * 3 + 1] = c.y;
    out[i * 3 + 2] = c.z;
}

__global__ void intensity_at_kernel(SceneHdr H, SceneSoA S, const float4* __restrict__ points, uint32_t n,
                                    float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Counters cnt = {0u, 0u};
    float4 p = points[i];
    out[i] = intensity_at<4, false>(H, S, v3(p.x, p.y, p.z), i, 1u, cnt);
}
// ... a world of at most four scale+translate-only spheres / planes / cubes without patterns: the SIMPLE instantiation, the
// only one that takes the fast decision of a sample (shadow_fast) -- what the render kernels run for such a world
__global__ void intensity_at_kernel_simple(SceneHdr H, SceneSoA S, const float4* __restrict__ points, uint32_t n,
                                           float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Counters cnt = {0u, 0u};
    float4 p = points[i];
    out[i] = intensity_at<4, true>(H, S, v3(p.x, p.y, p.z), i, 1u, cnt);
}
__global__ void intensity_at_kernel_generic(SceneHdr H, SceneSoA S, const float4* __restrict__ points, uint32_t n,
                                            float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Counters cnt = {0u, 0u};
    float4 p = points[i];
    out[i] = H.n_trav ? intensity_at<-1, false>(H, S, v3(p.x, p.y, p.z), i, 1u, cnt)
                      : intensity_at<0, false>(H, S, v3(p.x, p.y, p.z), i, 1u, cnt);
}

// RectangleLight::point_on_light (rectangle_light.rs:60-66) for caller-supplied cells, each as the first call on a new light
__global__ void point_on_light_kernel(SceneHdr H, const int2* __restrict__ cells, uint32_t n, float4* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = cells[i].x, v = cells[i].y;
    float j1 = H.jitter_const, j2 = H.jitter_const;
    if (H.jitter_mode == RTC_JITTER_HASHED) {
        const uint32_t key = jitter_base(H.jitter_seed, 0u, 1u) + 2u * (uint32_t)(v * H.u_steps + u) * 0x85EBCA6Bu;
        j1 = jitter_value(mix32(key));
        j2 = jitter_value(mix32(key + 0x85EBCA6Bu));
    } else if (H.jitter_mode == RTC_JITTER_SEQUENCE) {
        j1 = H.jitter_seq[0];
        j2 = H.jitter_seq[H.jitter_seq_len > 1u ? 1u : 0u];
    }
    const V3 lp = point_on_light(v3(H.corner[0], H.corner[1], H.corner[2]), v3(H.uvec[0], H.uvec[1], H.uvec[2]), v3(H.vvec[0], H.vvec[1], H.vvec[2]),
                                 (float)u + j1, (float)v + j2);
    out[i] = make_float4(lp.x, lp.y, lp.z, 1.0f);
}

__global__ void is_shadowed_kernel(SceneHdr H, SceneSoA S, const float4* __restrict__ lights,
                                   const float4* __restrict__ points, uint32_t n, int32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Counters cnt = {0u, 0u};
    float4 l = lights[i], p = points[i];
    const bool s = H.n_trav ? is_shadowed<-1>(H, S, v3(l.x, l.y, l.z), v3(p.x, p.y, p.z), cnt)
                            : is_shadowed<0>(H, S, v3(l.x, l.y, l.z), v3(p.x, p.y, p.z), cnt);
    out[i] = s ? 1 : 0;
}

__global__ void powf_kernel(const float* __restrict__ x, const float* __restrict__ y, uint32_t n,
                            float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rtc_powf_dev(x[i], y[i]);
}

__global__ void atan2f_kernel(const float* __restrict__ y, const float* __restrict__ x, uint32_t n, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = atan2f_glibc(y[i], x[i]);
}
__global__ void acosf_kernel(const float* __restrict__ x, uint32_t n, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = acosf_glibc(x[i]);
}
__global__ void cosf_kernel(const float* __restrict__ x, uint32_t n, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rtc_cosf_dev(x[i]);
}

// Shape::local_intersect for caller-supplied object-space rays: up to 4 distances per ray, in push order.
__global__ void local_intersect_kernel(Obj ob, const float4* __restrict__ tri, const float4* __restrict__ origins,
                                       const float4* __restrict__ directions, uint32_t n, float4* __restrict__ out_t,
                                       int32_t* __restrict__ out_count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 o = origins[i], d = directions[i];
    float ts[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int count = 0;
    local_intersect<false>(ob.bits, ob.min_y(), ob.max_y(), tri, 0u, v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), [&](float t) {
        if (count < 4) ts[count] = t;
        count++;
    });
    out_t[i] = make_float4(ts[0], ts[1], ts[2], ts[3]);
    out_count[i] = count;
}

// Shape::normal_at (shape.rs:72-154) for caller-supplied world points.
__global__ void normal_at_kernel(Obj ob, const float4* __restrict__ tri, const float4* __restrict__ points, uint32_t n,
                                 float4* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = points[i];
    V3 op = obj_point(ob, v3(p.x, p.y, p.z));
    V3 nn = obj_normal_to_world(ob, local_normal(ob.bits & SHAPE_KIND_MASK, ob.min_y(), ob.max_y(), tri, op));
    out[i] = make_float4(nn.x, nn.y, nn.z, 0.0f);
}

// Pattern::color_at_object (pattern.rs:15-19) for caller-supplied world points; pat: the 5 pattern records.
__global__ void pattern_color_kernel(Obj ob, const float4* __restrict__ pat, const float4* __restrict__ uvrec,
                                     const float* __restrict__ texels, const float4* __restrict__ points, uint32_t n,
                                     float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = points[i];
    SceneSoA S = {};
    S.uvrec = uvrec;
    S.texels = texels;
    V3 c = pattern_color_at_object(S, pat, ob, v3(p.x, p.y, p.z));
    out[3 * i + 0] = c.x;
    out[3 * i + 1] = c.y;
    out[3 * i + 2] = c.z;
}

// Diagnostic: sqrt_core / RcpCore against the compiler's sqrtf and '/' on caller-supplied vectors.
// out[0] = vectors inside the core range, out[1] = of those, how many differ in any bit pattern (a
// zero's sign excepted).
__global__ void fastmath_selftest_kernel(const float* __restrict__ v, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 a = v3(v[i * 3], v[i * 3 + 1], v[i * 3 + 2]);
    float sum = a.x * a.x + a.y * a.y + a.z * a.z;
    if (!normalize_in_core_range(a, sum)) return;
    float m_ref = sqrtf(sum);
    V3 d_ref = v3(a.x / m_ref, a.y / m_ref, a.z / m_ref);
    float m = sqrt_core(sum);
    RcpCore rc(m);
    V3 d = v3(rc.div(a.x), rc.div(a.y), rc.div(a.z));
    auto same = [](float p, float q) { return __float_as_uint(p) == __float_as_uint(q) || (p == 0.0f && q == 0.0f); };
    atomicAdd(&out[0], 1u);
    if (!(same(m, m_ref) && same(d.x, d_ref.x) && same(d.y, d_ref.y) && same(d.z, d_ref.z))) atomicAdd(&out[1], 1u);
}
#endif  // !RTC_SPEC_LIST (the utility kernels above are ahead-of-time only)

}  // namespace rtc

#endif  // RTC_KERNEL_CORE_H
