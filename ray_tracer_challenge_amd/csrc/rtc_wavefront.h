// rtc_wavefront.h -- Camera::render (camera.rs:76-91) level by level: one lane per RAY instead of one lane per pixel.
//
// The per-pixel kernels (rtc_kernel_core.h) walk a pixel's whole ray tree -- world.rs:62-86: every hit that both reflects
// and refracts spawns two children, 63 rays at depth 5 -- in one lane, as an explicit post-order stack.  A frame then
// takes as long as its longest wave: on a glass mesh a few dozen waves over the rim of the mesh run for most of the
// frame while the rest of the chip is idle (LABNOTES.md section 9, "when the waves run": mesh 2048^2 is throughput until
// 2.6 ms and ends at 3.9).  Here the unit of scheduling is one ray:
//
//   level 0        one lane per pixel traces the primary ray (ray_for_pixel + World::color_at's hit and shade_hit, world.rs:
//                  62-101); a hit with children becomes a NODE in HBM -- the suspended shade_hit: surface colour, the
//                  material's reflective / transparency, Schlick's R, where the result goes -- and its child rays are
//                  appended to the next level's ray lists (reflection children to one list, refraction children to the other:
//                  neighbouring pixels' children stay neighbours, which is what the packet walk's economy rests on);
//   level 1 .. d   one lane per listed ray does the same; a ray without children hands its colour to its parent node (or,
//                  for a primary ray, to the canvas);
//   then bottom-up, level d .. 0: one lane per node forms `surface + reflected + refracted` exactly as shade_hit does
//                  (same operations, same order -- world.rs:77-85) and hands the result up.
//
// Same rays, same order of floating-point operations per ray and per sum, same jitter keys (global pixel, path code): the
// image and the ray counts are those of the per-pixel kernels and of the oracle, bit for bit (tests/test_wavefront.py).
// Launch sizes never come back to the host: every level is launched over its lists' CAPACITY and lanes beyond the counts the
// previous level left in memory leave at once.  A list or the node pool running full raises a flag; the host then renders the
// frame with the per-pixel kernel instead.
#ifndef RTC_WAVEFRONT_H
#define RTC_WAVEFRONT_H

#include "rtc_kernel_core.h"

#pragma clang fp contract(off)

namespace rtc {

struct WfRay {  // 40 bytes
    float ox, oy, oz, dx, dy, dz;
    uint32_t pixel, path;  // the jitter's key: global pixel index y * width + x, path code (1 at the root, 2p / 2p + 1)
    int32_t parent;        // >= 0: node index; < 0: -(1 + index of the pixel in the partition's compact rows)
    uint32_t rem_kind;     // remaining depth | (which child of its parent: 0 reflection, 1 refraction) << 16
};
struct WfNode {  // 64 bytes: one suspended shade_hit
    float acc[3];                       // surface colour (phong_lighting)
    float R, reflective, transparency;  // Schlick's reflectance, Material.reflective, Material.transparency
    uint32_t flags;                     // F_HAS_REFR / F_SCHLICK as in color_at, WF_HAS_REFL
    int32_t parent;                     // as WfRay::parent
    uint32_t kind;                      // which child of its parent this node is
    float c0[3], c1[3];                 // the reflection / refraction child's colour, once known
    uint32_t pad;
};
enum { WF_HAS_REFL = 8 };
// counters in device memory (uint32): [0] nodes allocated, [1] overflow, then per level L: rays in its reflection list, rays in
// its refraction list, nodes allocated once the level has been traced, and the two work counters its passes draw 64-lane
// chunks from (tracing the level's rays; combining its nodes)
constexpr uint32_t WF_CTR_NODES = 0, WF_CTR_OVERFLOW = 1, WF_CTR_LEVEL0 = 2, WF_CTR_PER_LEVEL = 5, WF_MAX_LEVELS = 10;
constexpr uint32_t WF_CTR_WORDS = WF_CTR_LEVEL0 + WF_CTR_PER_LEVEL * (WF_MAX_LEVELS + 1);

struct WfArgs {
    SceneHdr hdr;
    SceneSoA soa;
    float* out;       // compact rows of this partition, f32 RGB ...
    uint8_t* out_u8;  // ... or scale_color'd bytes
    uint32_t rows, band_rows, n_parts, part;
    int32_t depth;
    uint32_t level;
    const WfRay* in_refl;  // this level's rays
    const WfRay* in_refr;
    WfRay* out_refl;       // the next level's
    WfRay* out_refr;
    WfNode* nodes;
    uint32_t* ctr;
    uint32_t cap_rays, cap_nodes;
    // {rays, shaded hits, culled shadow rays, 0} per wave of every launch of the frame: this launch's waves write slots
    // count_base + their index; sum_counts_kernel adds them up at the end of the frame (three atomics per wave on the frame's
    // totals were what the first version of this file spent its time on: 400 k same-address atomics, 8 ms)
    uint4* wave_counts;
    uint32_t count_base;
};

DI void wf_store_pixel(const WfArgs& A, uint32_t local, V3 col) {
    if (A.out_u8 != nullptr) {
        uint8_t* dst = A.out_u8 + (size_t)local * 3;
        dst[0] = (uint8_t)fmaxf(fminf(col.x * 255.0f, 255.0f), 0.0f);
        dst[1] = (uint8_t)fmaxf(fminf(col.y * 255.0f, 255.0f), 0.0f);
        dst[2] = (uint8_t)fmaxf(fminf(col.z * 255.0f, 255.0f), 0.0f);
    } else {
        float* dst = A.out + (size_t)local * 3;
        dst[0] = col.x, dst[1] = col.y, dst[2] = col.z;
    }
}
// a finished colour goes to its parent node's child slot, or to the canvas
DI void wf_deliver(const WfArgs& A, int32_t parent, uint32_t kind, V3 col) {
    if (parent < 0) {
        wf_store_pixel(A, (uint32_t)(-(parent + 1)), col);
    } else {
        float* c = kind ? A.nodes[parent].c1 : A.nodes[parent].c0;
        c[0] = col.x, c[1] = col.y, c[2] = col.z;
    }
}
// rank of this lane among the lanes of `mask` below it, and one atomic per wave reserving popcount(mask) slots
DI uint32_t wf_reserve(uint32_t* counter, unsigned long long mask, uint32_t lane, uint32_t& rank) {
    rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    uint32_t base = 0u;
    if (mask != 0ull) {  // wave-uniform
        const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
        if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, (int)leader, 64);
    }
    return base;
}

// One level of the frame: PRIMARY -- one lane per pixel of the partition (8 x 8 tile per wave, as render_body); else one lane per
// ray of the level's two lists.
template <bool PRIMARY>
__global__ __launch_bounds__(256, 6) void wf_trace_kernel(WfArgs A) {
    const SceneHdr& H = A.hdr;
    const SceneSoA& S = A.soa;
    const uint32_t lane = threadIdx.x & 63u;
    Counters cnt = {0u, 0u, 0u};
    // A level's rays are drawn in chunks of 64 from a work counter by however many waves the launch has (a fixed few thousand:
    // launching a workgroup per 256 rays of the lists' CAPACITY cost more in empty workgroups than the frame in rays -- 1.2 M of
    // them on a 4096 x 2048 frame, 7.8 ms against 0.56 for the per-pixel kernel).  PRIMARY: one tile per wave, one pass.
    uint32_t n_refl = 0u, n_refr = 0u;
    if constexpr (!PRIMARY) {
        const uint32_t* lc = A.ctr + WF_CTR_LEVEL0 + WF_CTR_PER_LEVEL * A.level;
        n_refl = min(lc[0], A.cap_rays), n_refr = min(lc[1], A.cap_rays);
    }
    for (;;) {
    uint32_t chunk = 0u;
    if constexpr (!PRIMARY) {
        if (lane == 0) chunk = atomicAdd(A.ctr + WF_CTR_LEVEL0 + WF_CTR_PER_LEVEL * A.level + 3u, 1u);
        chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
        if (chunk * 64u >= n_refl + n_refr) break;  // wave-uniform
    }
    bool active = false;
    V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 1.0f);
    uint32_t pixel = 0u, path = 1u, kind = 0u;
    int32_t parent = 0;
    // (PRIMARY: every ray starts with the call's depth.  Assigned HERE, not beside `active = true` below: the value is
    // wave-uniform on either side of that divergent branch -- depth or the initial 0 -- and the compiler (ROCm 7.2, gfx950) merged
    // the two in a SCALAR register: one lane of the wave leaving through the "sees nothing" / last-column path zeroed the
    // depth of all 64, and every hit of such a wave came out childless.  Found on here_be_dragons, whose scene box makes
    // such lanes; the ISA shows `s_mov_b32 s34, 0` on that path feeding `v_mov_b32 v4, s34` for the whole wave.)
    int rem = PRIMARY ? A.depth : 0;
    if constexpr (PRIMARY) {
        const uint32_t wave = threadIdx.x >> 6;
        const uint32_t x = (blockIdx.x << 4) + ((wave & 1u) << 3) + (lane & 7u);
        const uint32_t yl = (blockIdx.y << 4) + ((wave >> 1) << 3) + (lane >> 3);
        if (x < H.width && yl < A.rows) {
            const uint32_t band = yl / A.band_rows;
            const uint32_t y = (band * A.n_parts + A.part) * A.band_rows + (yl - band * A.band_rows);
            parent = -(int32_t)(1u + yl * H.width + x);
            // camera.rs:80-81: the last row and column stay black
            if (x < H.width - 1u && y < H.height - 1u) {
                // ray_for_pixel, camera.rs:60-74 (as render_body)
                float x_offset = ((float)x + 0.5f) * H.pixel_size;
                float y_offset = ((float)y + 0.5f) * H.pixel_size;
                float world_x = H.half_w - x_offset;
                float world_y = H.half_h - y_offset;
                const float* c = H.cam;
                V3 pix = {c[0] * world_x + c[1] * world_y + c[2] * -1.0f + c[3], c[4] * world_x + c[5] * world_y + c[6] * -1.0f + c[7],
                          c[8] * world_x + c[9] * world_y + c[10] * -1.0f + c[11]};
                V3 origin = v3(H.cam_origin[0], H.cam_origin[1], H.cam_origin[2]);
                bool sees_nothing = false;
                if (H.has_scene_box) {  // render_body's early-out: a ray that misses the padded box of everything is black after one counted ray
                    const V3 du = pix - origin;
                    const V3 iu = v3(__builtin_amdgcn_rcpf(du.x), __builtin_amdgcn_rcpf(du.y), __builtin_amdgcn_rcpf(du.z));
                    float tmin;
                    sees_nothing = !aabb_hit(origin, iu, make_float4(H.scene_box[0], H.scene_box[1], H.scene_box[2], 0.0f),
                                             make_float4(H.scene_box[3], H.scene_box[4], H.scene_box[5], 0.0f), tmin);
                }
                if (sees_nothing) {
                    cnt.rays += 1u;
                    wf_store_pixel(A, yl * H.width + x, v3(0.0f, 0.0f, 0.0f));
                } else {
                    o = origin;
                    d = norm3(pix - origin);
                    pixel = y * H.width + x;
                    active = true;
                }
            } else {
                wf_store_pixel(A, yl * H.width + x, v3(0.0f, 0.0f, 0.0f));
            }
        }
    } else {
        const uint32_t t = chunk * 64u + lane;
        if (t < n_refl + n_refr) {
            const WfRay r = t < n_refl ? A.in_refl[t] : A.in_refr[t - n_refl];
            o = v3(r.ox, r.oy, r.oz), d = v3(r.dx, r.dy, r.dz);
            pixel = r.pixel, path = r.path, parent = r.parent;
            rem = (int)(r.rem_kind & 0xffffu), kind = r.rem_kind >> 16;
            active = true;
        }
    }
    // ---- World::color_at for this ray (world.rs:88-101), up to where its children would be called
    // what the hit leaves to do, as ONE per-lane word (bit 0: reflection child, bit 1: refraction child, bit 2: Schlick) that is
    // pinned to a vector register below: as three bools assigned inside the nested branches of the hit these lived in scalar
    // lane masks across the tree walks of intensity_at / refraction_indices, and the compiled kernel lost them on one scene
    // (here_be_dragons: every hit came out childless) while a build with one more use of the same values kept them
    uint32_t todo = 0u;
    bool has_refl = false, has_refr = false, use_schlick = false;
    V3 surface = v3(0.0f, 0.0f, 0.0f), over_point = o, under_point = o, reflectv = d, rdir = d;
    float R = 0.0f, reflective = 0.0f, transparency = 0.0f;
    if (active) {
        cnt.rays += 1u;
        const Hit h = nearest_hit<-1, false>(H, S, o, d, cnt);
        V3 ret = v3(0.0f, 0.0f, 0.0f);
        if (h.obj >= 0) {
            // precompute_values (world.rs:212-233) and shade_hit (world.rs:62-86), operation for operation as color_at has them
            const int ob = h.obj;
            const Obj rec = load_obj(S, ob);
            const V3 point = o + d * h.t;
            const V3 op = obj_point(rec, point);
            V3 n = obj_normal_to_world(rec, local_normal(rec.bits & SHAPE_KIND_MASK, rec.min_y(), rec.max_y(), S.tri + 3 * ob, op));
            const bool inside = dot3(n, -d) < 0.0f;
            if (inside) n = -n;
            over_point = point + n * SELF_EPS;
            cnt.shaded += 1u;
            const float li = intensity_at<-1, false>(H, S, over_point, pixel, path, cnt);
            const V3 eye = -d;
            reflectv = reflect3(d, inside ? -n : n);  // world.rs:221 uses the normal before the inside flip
            under_point = point - n * SELF_EPS;
            const float4 ma = S.mat_a[ob], mb = S.mat_b[ob], mc = S.mat_c[ob];
            reflective = mb.w, transparency = mc.x;
            V3 material_color = v3(ma.x, ma.y, ma.z);
            if (spec_has_patterns(H.has_patterns)) {
                const float4* pat = S.pat + 5 * ob;
                if (__float_as_uint(pat[0].w) != RTC_PATTERN_NONE) material_color = pattern_color_at_object(S, pat, load_obj(S, ob), over_point);
            }
            surface = phong(H, material_color, ma, mb, over_point, eye, n, li);
            has_refl = !(reflective == 0.0f || rem < 1);                  // world.rs:126
            use_schlick = reflective > 0.0f && transparency > 0.0f;       // world.rs:80
            if (transparency != 0.0f) {
                float n1, n2;
                refraction_indices<-1>(H, S, o, d, ob, n1, n2, cnt);
                if (use_schlick) R = schlick(eye, n, n1, n2);
                if (rem != 0) {  // refracted_color, world.rs:140-161
                    const float n_ratio = n1 / n2;
                    const float cos_i = dot3(eye, n);
                    const float sin2 = n_ratio * n_ratio * (1.0f - cos_i * cos_i);
                    if (!(sin2 > 1.0f)) {
                        const float cos_t = sqrtf(1.0f - sin2);
                        rdir = n * (n_ratio * cos_i - cos_t) - (eye * n_ratio);
                        has_refr = true;
                    }
                }
            }
            if (!has_refl && !has_refr) {
                const V3 black = v3(0.0f, 0.0f, 0.0f);
                ret = use_schlick ? surface + black * R + black * (1.0f - R) : surface + black + black;
            }
            todo = (has_refl ? 1u : 0u) | (has_refr ? 2u : 0u) | (use_schlick ? 4u : 0u);
        }
        asm volatile("" : "+v"(todo));
        if ((todo & 3u) == 0u) wf_deliver(A, parent, kind, ret);
    }
    asm volatile("" : "+v"(todo));
    has_refl = (todo & 1u) != 0u, has_refr = (todo & 2u) != 0u, use_schlick = (todo & 4u) != 0u;
    bool want_node = (todo & 3u) != 0u;
    // ---- the suspended shade_hit becomes a node, its children the next level's rays (one reservation per wave and list)
    uint32_t rank;
    uint32_t base = wf_reserve(A.ctr + WF_CTR_NODES, __ballot(want_node), lane, rank);
    const uint32_t node = base + rank;
    if (want_node && node >= A.cap_nodes) {
        A.ctr[WF_CTR_OVERFLOW] = 1u;
        want_node = false;
    }
    if (want_node) {
        WfNode nd;
        nd.acc[0] = surface.x, nd.acc[1] = surface.y, nd.acc[2] = surface.z;
        nd.R = R, nd.reflective = reflective, nd.transparency = transparency;
        nd.flags = (has_refr ? F_HAS_REFR : 0) | (use_schlick ? F_SCHLICK : 0) | (has_refl ? WF_HAS_REFL : 0);
        nd.parent = parent, nd.kind = kind;
        nd.c0[0] = nd.c0[1] = nd.c0[2] = nd.c1[0] = nd.c1[1] = nd.c1[2] = 0.0f;
        nd.pad = 0u;
        A.nodes[node] = nd;
    }
    uint32_t* next = A.ctr + WF_CTR_LEVEL0 + WF_CTR_PER_LEVEL * (A.level + 1u);
    for (uint32_t which = 0; which < 2u; which++) {
        const bool emit = want_node && (which ? has_refr : has_refl);
        base = wf_reserve(next + which, __ballot(emit), lane, rank);
        const uint32_t idx = base + rank;
        if (emit && idx >= A.cap_rays) A.ctr[WF_CTR_OVERFLOW] = 1u;
        else if (emit) {
            WfRay r;
            const V3 ro = which ? under_point : over_point, rd = which ? rdir : reflectv;
            r.ox = ro.x, r.oy = ro.y, r.oz = ro.z, r.dx = rd.x, r.dy = rd.y, r.dz = rd.z;
            r.pixel = pixel, r.path = path * 2u + which;
            r.parent = (int32_t)node;
            r.rem_kind = (uint32_t)(rem - 1) | (which << 16);
            (which ? A.out_refr : A.out_refl)[idx] = r;
        }
    }
    if constexpr (PRIMARY) break;
    }
    // ---- statistics: one set of atomics per wave
    uint32_t rays = cnt.rays, shaded = cnt.shaded & CNT_SHADED_MASK, culled = cnt.shaded >> CNT_CULLED_SHIFT;
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off, 64);
        shaded += __shfl_down(shaded, off, 64);
        culled += __shfl_down(culled, off, 64);
    }
    if (lane == 0) A.wave_counts[A.count_base + ((blockIdx.y * gridDim.x + blockIdx.x) << 2) + (threadIdx.x >> 6)] = make_uint4(rays, shaded, culled, 0u);
}

// after a level has been traced: how many nodes exist now (the combine pass's range for that level)
__global__ void wf_snapshot_kernel(uint32_t* ctr, uint32_t level, uint32_t cap_nodes) {
    ctr[WF_CTR_LEVEL0 + WF_CTR_PER_LEVEL * level + 2u] = min(ctr[WF_CTR_NODES], cap_nodes);
}

// shade_hit's sum (world.rs:77-85) for the nodes of one level, children's colours being final: the operations of color_at's
// return path, in its order
__global__ __launch_bounds__(256) void wf_combine_kernel(WfArgs A) {
    const uint32_t begin = A.level == 0u ? 0u : A.ctr[WF_CTR_LEVEL0 + WF_CTR_PER_LEVEL * (A.level - 1u) + 2u];
    const uint32_t end = A.ctr[WF_CTR_LEVEL0 + WF_CTR_PER_LEVEL * A.level + 2u];
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {  // chunks of 64 nodes from the level's work counter (see wf_trace_kernel)
    uint32_t chunk = 0u;
    if (lane == 0) chunk = atomicAdd(A.ctr + WF_CTR_LEVEL0 + WF_CTR_PER_LEVEL * A.level + 4u, 1u);
    chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
    if (begin + chunk * 64u >= end) break;
    const uint32_t i = begin + chunk * 64u + lane;
    if (i >= end) continue;
    const WfNode nd = A.nodes[i];
    const V3 black = v3(0.0f, 0.0f, 0.0f);
    const V3 acc = v3(nd.acc[0], nd.acc[1], nd.acc[2]);
    V3 partial;
    if (nd.flags & WF_HAS_REFL) {
        const V3 reflected = v3(nd.c0[0], nd.c0[1], nd.c0[2]) * nd.reflective;  // world.rs:131
        partial = (nd.flags & F_SCHLICK) ? acc + reflected * nd.R : acc + reflected;
    } else {
        partial = (nd.flags & F_SCHLICK) ? acc + black * nd.R : acc + black;
    }
    V3 ret;
    if (nd.flags & F_HAS_REFR) {
        const V3 refracted = v3(nd.c1[0], nd.c1[1], nd.c1[2]) * nd.transparency;  // world.rs:159-160
        ret = (nd.flags & F_SCHLICK) ? partial + refracted * (1.0f - nd.R) : partial + refracted;
    } else {
        ret = (nd.flags & F_SCHLICK) ? partial + black * (1.0f - nd.R) : partial + black;
    }
    wf_deliver(A, nd.parent, nd.kind, ret);
    }
}

}  // namespace rtc

#endif  // RTC_WAVEFRONT_H
