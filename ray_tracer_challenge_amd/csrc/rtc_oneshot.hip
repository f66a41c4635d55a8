// rtc_oneshot.hip -- Camera::render (camera.rs:76-91) in ONE call, host buffer out: rtc_render / rtc_render_ex.
//
// What a caller of the reference's seam pays for is the whole call, not the kernel: context, scene upload, kernel
// compile, output allocation, the kernel, and 201 MB (4096^2 f32) back over PCIe.  Everything that can be kept between
// calls is kept per device (context with its compiled kernel and resident scene, output buffer, pinned staging memory,
// streams, events), and the rest is a pipeline:
//
//   * the image's 64-row bands are dealt round-robin over the devices of opts->devices (pixels are independent,
//     camera.rs:80-85; the jitter key is the global pixel index, so any split assembles to the same image);
//   * each device's share is rendered as several launches (row chunks: part k + D*j of D*C parts, rtc_partition), so
//     chunk j travels while chunk j+1 renders;
//   * host output: every device copies its own chunks to the host over its own PCIe link -- straight into `out` when
//     that is page-locked memory (rtc_host_alloc), else through two pinned staging buffers that a small thread pool
//     empties into `out` while the next DMA runs (a plain hipMemcpy into pageable memory runs at ~10 GB/s here);
//   * device output (opts->out_on_device): peers' bands go GPU-to-GPU into devices[0]'s buffer, one xGMI hop each.
//
// No RCCL here: a star of point-to-point copies is all the path has (bench.py --gpus N, one process per GPU, does the
// same gather with RCCL send/recv through torch.distributed).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "rtc_internal.h"

using namespace rtc;

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(RTC_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define RTC_TRY(expr)                  \
    do {                               \
        rtc_status s_ = (expr);        \
        if (s_ != RTC_OK) return s_;   \
    } while (0)

namespace {

// ---- a small persistent thread pool for the staging -> `out` copies -------------------------------------------
class CopyPool {
  public:
    static CopyPool& get() {
        static CopyPool p;
        return p;
    }
    // runs fn(i) for i in [0, n) on the pool's threads and the caller's; returns when all are done
    void parallel_for(size_t n, const std::function<void(size_t)>& fn) {
        if (n == 0) return;
        if (n == 1 || threads_.empty()) {
            for (size_t i = 0; i < n; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> l(m_);
            fn_ = &fn;
            n_ = n;
            next_ = 0;
            pending_ = n;
            generation_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
    size_t width() const { return threads_.size() + 1; }

  private:
    CopyPool() {
        unsigned hw = std::thread::hardware_concurrency();
        unsigned n = hw > 1 ? std::min(hw - 1, 7u) : 0u;  // + the calling thread
        for (unsigned i = 0; i < n; i++) threads_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
            generation_++;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void work() {
        std::unique_lock<std::mutex> l(m_);
        while (next_ < n_) {  // items are claimed under the lock (a few hundred ~1 MB pieces per frame)
            const size_t i = next_++;
            const std::function<void(size_t)>* fn = fn_;
            l.unlock();
            (*fn)(i);
            l.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
            }
            work();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0, pending_ = 0;
    size_t next_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// ---- what is kept per entry of opts->devices between calls ---------------------------------------------------
struct DevState {
    int device = -1;
    rtc_ctx* ctx = nullptr;
    char* d_out = nullptr;  // this device's rows (f32), chunk after chunk
    size_t d_out_cap = 0;
    char* d_u8 = nullptr;  // the same rows as bytes (opts->quantize)
    size_t d_u8_cap = 0;
    char* h_stage[2] = {nullptr, nullptr};  // pinned
    size_t stage_cap = 0;
    hipStream_t s_render = nullptr, s_copy2[2] = {nullptr, nullptr};  // chunks alternate between two copy streams: two DMA engines
    std::vector<hipEvent_t> ev_render, ev_copy;

    void release() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        if (ctx) rtc_ctx_destroy(ctx);
        if (d_out) (void)hipFree(d_out);
        if (d_u8) (void)hipFree(d_u8);
        for (char*& h : h_stage)
            if (h) (void)hipHostFree(h), h = nullptr;
        for (hipEvent_t e : ev_render) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_copy) (void)hipEventDestroy(e);
        if (s_render) (void)hipStreamDestroy(s_render);
        for (hipStream_t& c : s_copy2)
            if (c) (void)hipStreamDestroy(c), c = nullptr;
        *this = DevState();
    }
};
std::mutex g_mutex;  // rtc_render_ex calls are serialised: they share the per-device state
std::vector<DevState> g_state;

rtc_status grow_device(char** p, size_t* cap, size_t need) {
    if (need <= *cap) return RTC_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void**)p, need));
    *cap = need;
    return RTC_OK;
}

// On any early return after work has been queued: wait for everything queued on the devices touched so far, so that no
// DMA is still writing into the caller's `out` (or reading staging memory the pool is about to reuse) and no render is
// left running with nobody to wait for it, when the caller sees the error and frees its buffer.
struct DrainOnError {
    std::vector<DevState*> touched;
    bool ok = false;
    ~DrainOnError() {
        if (ok) return;
        for (DevState* S : touched) {
            if (S->device < 0) continue;
            (void)hipSetDevice(S->device);
            if (S->s_render) (void)hipStreamSynchronize(S->s_render);
            for (hipStream_t c : S->s_copy2)
                if (c) (void)hipStreamSynchronize(c);
        }
        (void)hipGetLastError();
    }
};

struct Chunk {  // one launch: part `part` of `n_parts`
    uint32_t part, rows;
    size_t row0;  // first row inside the device's compact buffer
};
struct Band {  // one contiguous run of image rows inside a chunk
    size_t src_row;  // row inside the device's compact buffer
    uint32_t y0, rows;
};

}  // namespace

extern "C" {

void* rtc_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void rtc_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

void rtc_render_release(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto& s : g_state) s.release();
    g_state.clear();
}

rtc_status rtc_render_ex(const rtc_scene* scene, const rtc_camera* camera, int32_t depth, const rtc_opts* opts, void* out,
                         rtc_stats* stats) {
    if (!out || !camera) return fail(RTC_ERR_INVALID_ARG, "rtc_render: null argument");
    if (camera->width == 0 || camera->height == 0) return fail(RTC_ERR_INVALID_ARG, "empty canvas");
    const int32_t dev0 = 0;
    const int32_t* devices = (opts && opts->devices && opts->n_devices) ? opts->devices : &dev0;
    const uint32_t D = (opts && opts->devices && opts->n_devices) ? opts->n_devices : 1u;
    const uint32_t band_rows = (opts && opts->band_rows) ? opts->band_rows : 64u;
    const bool quantize = opts && opts->quantize != 0, on_device = opts && opts->out_on_device != 0;
    int n_visible = 0;
    if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible <= 0) {
        (void)hipGetLastError();
        return fail(RTC_ERR_NO_DEVICE, "no HIP device visible; librtc_amd has no CPU fallback");
    }
    if (D > 64) return fail(RTC_ERR_INVALID_ARG, "rtc_render_ex: %u devices", D);
    for (uint32_t k = 0; k < D; k++)
        if (devices[k] < 0 || devices[k] >= n_visible) return fail(RTC_ERR_INVALID_ARG, "device %d out of range (have %d)", devices[k], n_visible);

    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_state.size() < D) g_state.resize(D);
    const uint32_t W = camera->width, H = camera->height;
    const size_t px_bytes = quantize ? 3 : 12, row_out = (size_t)W * px_bytes, row_f32 = (size_t)W * 12;
    const uint32_t n_bands = (H + band_rows - 1) / band_rows;
    // chunks per device: ~24 MB of output each, at most 16, at least one band each
    const size_t share_bytes = (size_t)((n_bands + D - 1) / D) * band_rows * row_out;
    uint32_t C = (uint32_t)std::min<size_t>(16, std::max<size_t>(1, (share_bytes + (24u << 20) - 1) / (24u << 20)));
    C = std::min(C, std::max(1u, (n_bands + D - 1) / D));
    const uint32_t n_parts = D * C;

    // is `out` page-locked host memory (then DMA goes straight into it)?
    bool out_pinned = false;
    if (!on_device) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, out) == hipSuccess) out_pinned = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }

    std::vector<std::vector<Chunk>> chunks(D);
    std::vector<std::vector<std::vector<Band>>> bands(D);
    const auto t_start = std::chrono::steady_clock::now();
    DrainOnError drain;
    // ---- phase 1: every device's renders are queued (asynchronous) ------------------------------------------------
    for (uint32_t k = 0; k < D; k++) {
        DevState& S = g_state[k];
        if (S.device != devices[k]) {
            S.release();
            S.device = devices[k];
        }
        HIP_TRY(hipSetDevice(S.device));
        drain.touched.push_back(&S);
        if (!S.ctx) RTC_TRY(rtc_ctx_create(S.device, &S.ctx));
        if (!S.s_render) HIP_TRY(hipStreamCreateWithFlags(&S.s_render, hipStreamNonBlocking));
        for (hipStream_t& c : S.s_copy2)
            if (!c) HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        while (S.ev_render.size() < C) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            S.ev_render.push_back(e);
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            S.ev_copy.push_back(e);
        }
        RTC_TRY(rtc_ctx_set_scene(S.ctx, scene, camera));  // a no-op when this very scene is resident already
        size_t rows_total = 0, chunk_max = 0;
        for (uint32_t j = 0; j < C; j++) {
            rtc_partition part = {band_rows, n_parts, k + D * j};
            Chunk c = {part.part, rtc_partition_rows(H, &part), rows_total};
            std::vector<Band> bl;
            size_t r = rows_total;
            for (uint32_t b = c.part; b < n_bands; b += n_parts) {
                const uint32_t y0 = b * band_rows, y1 = std::min(H, y0 + band_rows);
                bl.push_back({r, y0, y1 - y0});
                r += y1 - y0;
            }
            rows_total += c.rows;
            chunk_max = std::max(chunk_max, (size_t)c.rows);
            chunks[k].push_back(c);
            bands[k].push_back(bl);
        }
        RTC_TRY(grow_device(&S.d_out, &S.d_out_cap, std::max<size_t>(1, rows_total * row_f32)));
        if (quantize) RTC_TRY(grow_device(&S.d_u8, &S.d_u8_cap, std::max<size_t>(1, rows_total * (size_t)W * 3)));
        if (!on_device && !out_pinned && chunk_max * row_out > S.stage_cap) {
            for (char*& h : S.h_stage) {
                if (h) (void)hipHostFree(h);
                h = nullptr;
            }
            S.stage_cap = 0;
            for (char*& h : S.h_stage) HIP_TRY(hipHostMalloc((void**)&h, chunk_max * row_out, hipHostMallocDefault));
            S.stage_cap = chunk_max * row_out;
        }
        for (uint32_t j = 0; j < C; j++) {
            const Chunk& c = chunks[k][j];
            rtc_partition part = {band_rows, n_parts, c.part};
            RTC_TRY(ctx_render_slot(S.ctx, depth, &part, S.d_out + c.row0 * row_f32, S.s_render, j));
            if (quantize && c.rows)
                RTC_TRY(rtc_ctx_quantize(S.ctx, S.d_out + c.row0 * row_f32, (uint64_t)c.rows * W * 3, S.d_u8 + c.row0 * (size_t)W * 3, S.s_render));
            HIP_TRY(hipEventRecord(S.ev_render[j], S.s_render));
        }
    }
    // ---- phase 2: the rows travel ---------------------------------------------------------------------------------
    auto src_of = [&](uint32_t k) { return quantize ? g_state[k].d_u8 : g_state[k].d_out; };
    if (on_device || out_pinned) {
        // band by band straight to where it belongs (device memory on devices[0], or page-locked host memory)
        for (uint32_t j = 0; j < C; j++)
            for (uint32_t k = 0; k < D; k++) {
                DevState& S = g_state[k];
                HIP_TRY(hipSetDevice(S.device));
                hipStream_t s_copy = S.s_copy2[j & 1];
                HIP_TRY(hipStreamWaitEvent(s_copy, S.ev_render[j], 0));
                for (const Band& b : bands[k][j]) {
                    char* dst = (char*)out + (size_t)b.y0 * row_out;
                    const char* src = src_of(k) + b.src_row * row_out;
                    const size_t n = (size_t)b.rows * row_out;
                    if (!on_device) HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s_copy));
                    else if (S.device == devices[0]) HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s_copy));
                    else HIP_TRY(hipMemcpyPeerAsync(dst, devices[0], src, S.device, n, s_copy));
                }
            }
        for (uint32_t k = 0; k < D; k++) {
            HIP_TRY(hipSetDevice(g_state[k].device));
            for (hipStream_t c : g_state[k].s_copy2) HIP_TRY(hipStreamSynchronize(c));
        }
    } else {
        // pageable host memory: DMA into pinned staging (two slots per device), emptied into `out` by the copy pool
        // while the next chunk's DMA runs
        auto enqueue = [&](uint32_t k, uint32_t j) -> rtc_status {
            DevState& S = g_state[k];
            const Chunk& c = chunks[k][j];
            HIP_TRY(hipSetDevice(S.device));
            hipStream_t s_copy = S.s_copy2[j & 1];  // (slot j & 1 of the staging memory belongs to this stream alone)
            HIP_TRY(hipStreamWaitEvent(s_copy, S.ev_render[j], 0));
            if (c.rows) HIP_TRY(hipMemcpyAsync(S.h_stage[j & 1], src_of(k) + c.row0 * row_out, (size_t)c.rows * row_out, hipMemcpyDeviceToHost, s_copy));
            HIP_TRY(hipEventRecord(S.ev_copy[j], s_copy));
            return RTC_OK;
        };
        for (uint32_t k = 0; k < D; k++)
            for (uint32_t j = 0; j < std::min(2u, C); j++) RTC_TRY(enqueue(k, j));
        CopyPool& pool = CopyPool::get();
        for (uint32_t j = 0; j < C; j++)
            for (uint32_t k = 0; k < D; k++) {
                DevState& S = g_state[k];
                HIP_TRY(hipSetDevice(S.device));
                HIP_TRY(hipEventSynchronize(S.ev_copy[j]));
                // split the chunk's bands into pieces of ~1 MB so that every thread of the pool has work
                struct Piece {
                    char* dst;
                    const char* src;
                    size_t n;
                };
                std::vector<Piece> pieces;
                const char* stage = S.h_stage[j & 1];
                const size_t row0 = chunks[k][j].row0;
                for (const Band& b : bands[k][j]) {
                    const size_t total = (size_t)b.rows * row_out, step = std::max<size_t>(row_out, ((1u << 20) / row_out) * row_out);
                    for (size_t off = 0; off < total; off += step)
                        pieces.push_back({(char*)out + (size_t)b.y0 * row_out + off, stage + (b.src_row - row0) * row_out + off, std::min(step, total - off)});
                }
                pool.parallel_for(pieces.size(), [&](size_t i) { std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].n); });
                if (j + 2 < C) RTC_TRY(enqueue(k, j + 2));
            }
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    // ---- statistics -------------------------------------------------------------------------------------------------
    rtc_stats total;
    std::memset(&total, 0, sizeof(total));
    for (uint32_t k = 0; k < D; k++) {
        DevState& S = g_state[k];
        HIP_TRY(hipSetDevice(S.device));
        HIP_TRY(hipStreamSynchronize(S.s_render));
        rtc_stats s;
        RTC_TRY(ctx_collect(S.ctx, C, &s));
        total.rays += s.rays;
        total.shaded_hits += s.shaded_hits;
        total.culled_shadow_rays += s.culled_shadow_rays;
        total.kernel_ms = std::max(total.kernel_ms, s.kernel_ms);
        total.launches += s.launches;
        total.flags |= s.flags;
    }
    total.pixels = (uint64_t)(W - 1) * (H - 1);
    total.rows = H;
    total.gather_ms = (float)wall_ms;
    if (stats) *stats = total;
    drain.ok = true;
    return RTC_OK;
}

rtc_status rtc_render(const rtc_scene* scene, const rtc_camera* camera, int32_t depth, int32_t device, float* out_rgb,
                      rtc_stats* stats) {
    rtc_opts o;
    std::memset(&o, 0, sizeof(o));
    o.devices = &device;
    o.n_devices = 1;
    return rtc_render_ex(scene, camera, depth, &o, out_rgb, stats);
}

}  // extern "C"
