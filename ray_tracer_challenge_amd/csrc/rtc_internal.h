// rtc_internal.h -- declarations shared by the host and device translation
// units of librtc_amd.so.  Not part of the public ABI (that is include/rtc.h).
#ifndef RTC_INTERNAL_H
#define RTC_INTERNAL_H

#include <cstdarg>
#include <cstdint>

#include "rtc.h"

namespace rtc {

rtc_status fail(rtc_status code, const char* fmt, ...);

// matrix.rs:163-182 -- drop one row and one column of a row-major NxN matrix
template <int N>
inline void submatrix(const float* a, int remove_row, int remove_col, float* out) {
    int k = 0;
    for (int r = 0; r < N; r++) {
        if (r == remove_row) continue;
        for (int c = 0; c < N; c++) {
            if (c == remove_col) continue;
            out[k++] = a[r * N + c];
        }
    }
}

float determinant(const float* a, int n);
void inverse4(const float a[16], float out[16]);
void mat_mul4(const float a[16], const float b[16], float out[16]);
void mat_vec4(const float a[16], const float v[4], float out[4]);
float magnitude4(const float v[4]);
void norm4(const float v[4], float out[4]);
void cross4(const float a[4], const float b[4], float out[4]);
bool is_affine(const float m[16]);

// Band partition of rows (rtc_partition in rtc.h): resolved form.
struct Partition {
    uint32_t band_rows, n_parts, part;
};
inline Partition resolve(const rtc_partition* p) {
    Partition r = {64u, 1u, 0u};
    if (p) {
        if (p->band_rows) r.band_rows = p->band_rows;
        if (p->n_parts) r.n_parts = p->n_parts;
        r.part = p->part;
    }
    return r;
}
inline uint32_t partition_rows(uint32_t height, const rtc_partition* p) {
    Partition q = resolve(p);
    if (q.part >= q.n_parts) return 0;
    uint32_t n_bands = (height + q.band_rows - 1) / q.band_rows;
    uint32_t rows = 0;
    for (uint32_t b = q.part; b < n_bands; b += q.n_parts) {
        uint32_t y0 = b * q.band_rows;
        uint32_t y1 = y0 + q.band_rows < height ? y0 + q.band_rows : height;
        rows += y1 - y0;
    }
    return rows;
}

// Hooks of rtc_device.hip for rtc_render_ex (rtc_oneshot.hip): a frame rendered as several launches, each with a
// counter slot of its own.  `stream`: a hipStream_t.
constexpr uint32_t CTX_TOTAL_SLOTS = 64;
// Progress reporting of one launch (RenderArgs::progress): in -- where the host words live, how many chunks the caller
// would like, this frame's epoch; out -- how the launch's rows were cut (n_chunks == 0: this launch cannot report -- a
// block list, a scene rectangle, several blocks per workgroup -- and the caller waits for the stream instead).
struct ProgressPlan {
    uint32_t* d_done;      // device-visible address of the page-locked host words, one per chunk (>= PROGRESS_MAX_CHUNKS)
    uint32_t want_chunks, epoch;
    uint32_t n_chunks, chunk_rows;  // out: chunk j = local rows [j * chunk_rows, min(rows, (j + 1) * chunk_rows))
};
constexpr uint32_t PROGRESS_MAX_CHUNKS = 64;
// out_u8: the frame is stored as the bytes Canvas::to_ppm prints (scale_color, canvas.rs:39-43) -- `d_out` then holds
// rows * width * 3 bytes -- instead of f32 RGB.
rtc_status ctx_render_slot(rtc_ctx* c, int32_t depth, const rtc_partition* part, void* d_out, void* stream, uint32_t slot,
                           ProgressPlan* plan = nullptr, bool out_u8 = false);
rtc_status ctx_collect(rtc_ctx* c, uint32_t n_slots, rtc_stats* out);
// A context of the one-call seam (rtc_render_ex): a scene whose kernel is neither in memory nor in the disk cache is rendered by the
// ahead-of-time kernels the first time this process sees it and compiled when it is rendered again (rtc_device.hip jit_get).
void ctx_mark_one_shot(rtc_ctx* c);

}  // namespace rtc

#endif
