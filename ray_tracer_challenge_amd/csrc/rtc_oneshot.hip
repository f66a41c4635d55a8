// rtc_oneshot.hip -- Camera::render (camera.rs:76-91) in ONE call, host buffer out: rtc_render / rtc_render_ex.
//
// What a caller of the reference's seam pays for is the whole call, not the kernel: context, scene upload, kernel
// compile, output allocation, the kernel, and 201 MB (4096^2 f32) back over PCIe.  Everything that can be kept between
// calls is kept per device (context with its compiled kernel and resident scene, output buffer, pinned staging memory,
// streams, events), and the rest is a pipeline:
//
//   * the image's 64-row bands are dealt round-robin over the devices of opts->devices (pixels are independent,
//     camera.rs:80-85; the jitter key is the global pixel index, so any split assembles to the same image);
//   * each device's share is rendered by ONE launch whose kernel reports, chunk of rows by chunk of rows, what it has
//     finished (RenderArgs::progress: a word per chunk in page-locked host memory), so chunk j travels while the rows
//     after it are still being rendered -- and the frame's kernel time is that of a single launch;
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
#include <cstdio>
#include <cstdlib>
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
    char* d_out = nullptr;  // this device's rows, band after band: f32 RGB, or bytes (opts->quantize)
    size_t d_out_cap = 0;
    char* h_stage[2] = {nullptr, nullptr};  // pinned
    size_t stage_cap = 0;
    hipStream_t s_render = nullptr, s_copy2[2] = {nullptr, nullptr};  // chunks alternate between two copy streams: two DMA engines
    std::vector<hipEvent_t> ev_copy;
    uint32_t* h_done = nullptr;  // page-locked host words the render kernel reports finished chunks in (RenderArgs::done)
    uint32_t* d_done = nullptr;  // ... as the device addresses them
    uint32_t epoch = 0;          // this call's value of a finished chunk's word

    void release() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        if (ctx) rtc_ctx_destroy(ctx);
        if (d_out) (void)hipFree(d_out);
        for (char*& h : h_stage)
            if (h) (void)hipHostFree(h), h = nullptr;
        if (h_done) (void)hipHostFree(h_done);
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

struct Band {  // one contiguous run of image rows inside a device's compact buffer
    size_t src_row;  // row inside the compact buffer
    uint32_t y0, rows;
};
// One device's share of a frame while it is in flight: ONE launch, cut into chunks of consecutive compact rows that the
// kernel reports finished one by one (RenderArgs::progress) -- or, for launches that cannot report, one chunk that is
// finished when the stream is.
struct Share {
    uint32_t rows = 0, n_chunks = 0, chunk_rows = 0;
    bool reports = false;        // the kernel writes S.h_done[j] = S.epoch as chunk j completes
    bool stream_done = false;    // the render stream has been seen idle: every chunk is complete whatever the words say
    std::vector<Band> bands;     // all bands of the device, in compact order
    uint32_t next_copy = 0;      // chunks [0, next_copy) have their transfer queued
    uint32_t next_unstage = 0;   // pageable output: chunks [0, next_unstage) are in `out`
    uint32_t chunk_row0(uint32_t j) const { return std::min(rows, j * chunk_rows); }
    uint32_t chunk_row1(uint32_t j) const { return std::min(rows, (j + 1) * chunk_rows); }
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
    const size_t px_bytes = quantize ? 3 : 12, row_out = (size_t)W * px_bytes;
    const uint32_t n_bands = (H + band_rows - 1) / band_rows;

    // is `out` page-locked host memory (then DMA goes straight into it)?
    bool out_pinned = false;
    if (!on_device) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, out) == hipSuccess) out_pinned = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    const bool staged = !on_device && !out_pinned;

    std::vector<Share> shares(D);
    const auto t_start = std::chrono::steady_clock::now();
#ifdef RTC_DEV_SWITCHES
    // development (librtc_amd_dev.so only): RTC_AMD_SEAM_TRACE=1 prints where a call's time goes
    static const bool seam_trace = std::getenv("RTC_AMD_SEAM_TRACE") != nullptr;
    double t_queued = 0.0, t_first = 0.0, t_last = 0.0, t_copied = 0.0;
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
#endif
    DrainOnError drain;
    // ---- phase 1: every device's render is queued: ONE launch each, asynchronous ---------------------------------------
    // (Round 2 cut a device's share into ~24 MB launches so that one could travel while the next rendered; every launch
    // then ended in its own tail of a few long waves, and the frame's kernel time was 2.5 times that of a single launch.
    // Now the kernel itself says which rows are done.)
    for (uint32_t k = 0; k < D; k++) {
        DevState& S = g_state[k];
        Share& sh = shares[k];
        if (S.device != devices[k]) {
            S.release();
            S.device = devices[k];
        }
        HIP_TRY(hipSetDevice(S.device));
        drain.touched.push_back(&S);
        if (!S.ctx) {
            RTC_TRY(rtc_ctx_create(S.device, &S.ctx));
            rtc::ctx_mark_one_shot(S.ctx);  // (a scene's kernel is compiled when the scene comes a second time, or is in the disk cache)
        }
        if (!S.s_render) HIP_TRY(hipStreamCreateWithFlags(&S.s_render, hipStreamNonBlocking));
        for (hipStream_t& c : S.s_copy2)
            if (!c) HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        while (S.ev_copy.size() < PROGRESS_MAX_CHUNKS) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            S.ev_copy.push_back(e);
        }
        if (!S.h_done) {
            HIP_TRY(hipHostMalloc((void**)&S.h_done, PROGRESS_MAX_CHUNKS * sizeof(uint32_t), hipHostMallocMapped));
            std::memset(S.h_done, 0, PROGRESS_MAX_CHUNKS * sizeof(uint32_t));
            HIP_TRY(hipHostGetDevicePointer((void**)&S.d_done, S.h_done, 0));
        }
        if (++S.epoch == 0u) {  // (wrapped: words of 2^32 frames ago must not read as this frame's)
            std::memset(S.h_done, 0, PROGRESS_MAX_CHUNKS * sizeof(uint32_t));
            S.epoch = 1u;
        }
        RTC_TRY(rtc_ctx_set_scene(S.ctx, scene, camera));  // a no-op when this very scene is resident already
        rtc_partition part = {band_rows, D, k};
        sh.rows = rtc_partition_rows(H, &part);
        size_t r = 0;
        for (uint32_t b = k; b < n_bands; b += D) {
            const uint32_t y0 = b * band_rows, y1 = std::min(H, y0 + band_rows);
            if (!sh.bands.empty() && sh.bands.back().y0 + sh.bands.back().rows == y0) sh.bands.back().rows += y1 - y0;  // (one device: one run)
            else sh.bands.push_back({r, y0, y1 - y0});
            r += y1 - y0;
        }
        RTC_TRY(grow_device(&S.d_out, &S.d_out_cap, std::max<size_t>(1, (size_t)sh.rows * row_out)));
        // chunks of about 4 MB of output, at most 24: the call ends one chunk's transfer after the kernel does, so the last
        // should be short (a 50 MB u8 frame: 6 chunks 1.46 ms, the copy of the last alone 0.46; 12 chunks: see bench.py
        // one_shot), and a DMA of a few MB already runs at the link's rate
        ProgressPlan plan;
        plan.d_done = S.d_done;
        plan.epoch = S.epoch;
        plan.want_chunks = (uint32_t)std::min<size_t>(24, std::max<size_t>(1, ((size_t)sh.rows * row_out + (4u << 20) - 1) / (4u << 20)));
        plan.n_chunks = plan.chunk_rows = 0u;
        RTC_TRY(ctx_render_slot(S.ctx, depth, &part, S.d_out, S.s_render, 0u, &plan, quantize));
        sh.reports = plan.n_chunks != 0u;
        if (sh.reports) {
            sh.n_chunks = sh.rows == 0u ? 0u : plan.n_chunks;
            sh.chunk_rows = plan.chunk_rows;
        } else {
            // a launch that cannot report (block lists, scene rectangles, several blocks per workgroup) is complete when its stream
            // is: its rows still leave in the same ~4 MB chunks -- chunk_ready() answers true for every one of them once
            // stream_done is set -- so that the transfers and the unstaging copies pipeline and the pinned staging stays two
            // chunks, not two whole shares (C5 8192^2 f32: 2 x 805 MB)
            const uint32_t want = std::min<uint32_t>(plan.want_chunks, (uint32_t)PROGRESS_MAX_CHUNKS);
            sh.chunk_rows = sh.rows == 0u ? 0u : (sh.rows + want - 1u) / want;
            sh.n_chunks = sh.rows == 0u ? 0u : (sh.rows + sh.chunk_rows - 1u) / sh.chunk_rows;
        }
        if (staged && sh.rows) {
            const size_t need = (size_t)std::min(sh.rows, sh.chunk_rows) * row_out;
            if (need > S.stage_cap) {
                for (char*& h : S.h_stage) {
                    if (h) (void)hipHostFree(h);
                    h = nullptr;
                }
                S.stage_cap = 0;
                for (char*& h : S.h_stage) HIP_TRY(hipHostMalloc((void**)&h, need, hipHostMallocDefault));
                S.stage_cap = need;
            }
        }
    }
#ifdef RTC_DEV_SWITCHES
    t_queued = since();
#endif
    // ---- phase 2: the rows travel as their chunks are reported finished -------------------------------------------------
    // chunk j of device k is complete when the kernel has stored the frame's epoch into its word, or when the device's
    // render stream has been seen idle (launches that cannot report; and the safety net: a stream that is done has
    // written everything)
    auto chunk_ready = [&](uint32_t k, uint32_t j) {
        DevState& S = g_state[k];
        Share& sh = shares[k];
        if (sh.stream_done) return true;
        if (sh.reports && __atomic_load_n(&S.h_done[j], __ATOMIC_ACQUIRE) == S.epoch) return true;
        return false;
    };
    auto poll_stream = [&](uint32_t k) -> rtc_status {
        DevState& S = g_state[k];
        HIP_TRY(hipSetDevice(S.device));
        const hipError_t q = hipStreamQuery(S.s_render);
        if (q == hipSuccess) shares[k].stream_done = true;
        else if (q != hipErrorNotReady) return fail(RTC_ERR_DEVICE, "render failed: %s", hipGetErrorString(q));
        else (void)hipGetLastError();
        return RTC_OK;
    };
    // the compact rows [r0, r1) of device k as runs of image rows
    auto for_each_run = [&](uint32_t k, uint32_t r0, uint32_t r1, const std::function<rtc_status(size_t, uint32_t, uint32_t)>& fn) -> rtc_status {
        for (const Band& b : shares[k].bands) {
            const size_t lo = std::max<size_t>(b.src_row, r0), hi = std::min<size_t>(b.src_row + b.rows, r1);
            if (lo < hi) RTC_TRY(fn(lo, (uint32_t)(b.y0 + (lo - b.src_row)), (uint32_t)(hi - lo)));
        }
        return RTC_OK;
    };
    auto queue_copy = [&](uint32_t k, uint32_t j) -> rtc_status {
        DevState& S = g_state[k];
        Share& sh = shares[k];
        HIP_TRY(hipSetDevice(S.device));
        hipStream_t s_copy = S.s_copy2[j & 1];
        const uint32_t r0 = sh.chunk_row0(j), r1 = sh.chunk_row1(j);
        if (staged) {  // DMA into pinned staging slot j & 1 (this stream's alone), emptied into `out` by the copy pool
            if (r1 > r0) HIP_TRY(hipMemcpyAsync(S.h_stage[j & 1], S.d_out + (size_t)r0 * row_out, (size_t)(r1 - r0) * row_out, hipMemcpyDeviceToHost, s_copy));
        } else {  // straight to where the rows belong: page-locked host memory, or device memory on devices[0]
            RTC_TRY(for_each_run(k, r0, r1, [&](size_t src_row, uint32_t y0, uint32_t nrows) -> rtc_status {
                char* dst = (char*)out + (size_t)y0 * row_out;
                const char* src = S.d_out + src_row * row_out;
                const size_t n = (size_t)nrows * row_out;
                if (!on_device) HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s_copy));
                else if (S.device == devices[0]) HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s_copy));
                else HIP_TRY(hipMemcpyPeerAsync(dst, devices[0], src, S.device, n, s_copy));
                return RTC_OK;
            }));
        }
        HIP_TRY(hipEventRecord(S.ev_copy[j], s_copy));
        return RTC_OK;
    };
    CopyPool& pool = CopyPool::get();
    auto unstage = [&](uint32_t k, uint32_t j) -> rtc_status {  // staging slot j & 1 -> `out`, in pieces of ~1 MB so that every thread of the pool has work
        DevState& S = g_state[k];
        Share& sh = shares[k];
        struct Piece {
            char* dst;
            const char* src;
            size_t n;
        };
        std::vector<Piece> pieces;
        const char* stage = S.h_stage[j & 1];
        const uint32_t r0 = sh.chunk_row0(j), r1 = sh.chunk_row1(j);
        RTC_TRY(for_each_run(k, r0, r1, [&](size_t src_row, uint32_t y0, uint32_t nrows) -> rtc_status {
            const size_t total = (size_t)nrows * row_out, step = std::max<size_t>(row_out, ((1u << 20) / row_out) * row_out);
            for (size_t off = 0; off < total; off += step)
                pieces.push_back({(char*)out + (size_t)y0 * row_out + off, stage + (src_row - r0) * row_out + off, std::min(step, total - off)});
            return RTC_OK;
        }));
        pool.parallel_for(pieces.size(), [&](size_t i) { std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].n); });
        return RTC_OK;
    };
    for (uint32_t idle = 0;;) {
        bool all_done = true, progressed = false;
        for (uint32_t k = 0; k < D; k++) {
            DevState& S = g_state[k];
            Share& sh = shares[k];
            // queue the transfer of the next finished chunk (pageable output: while one of the two staging slots is free)
            if (sh.next_copy < sh.n_chunks && (!staged || sh.next_copy < sh.next_unstage + 2u) && chunk_ready(k, sh.next_copy)) {
                RTC_TRY(queue_copy(k, sh.next_copy));
                sh.next_copy++;
                progressed = true;
#ifdef RTC_DEV_SWITCHES
                if (t_first == 0.0) t_first = since();
                t_last = since();
#endif
            }
            // pageable output: empty the oldest staging slot whose DMA has landed
            if (staged && sh.next_unstage < sh.next_copy) {
                HIP_TRY(hipSetDevice(S.device));
                const hipError_t q = hipEventQuery(S.ev_copy[sh.next_unstage]);
                if (q == hipSuccess) {
                    RTC_TRY(unstage(k, sh.next_unstage));
                    sh.next_unstage++;
                    progressed = true;
                } else if (q != hipErrorNotReady) {
                    return fail(RTC_ERR_DEVICE, "transfer failed: %s", hipGetErrorString(q));
                } else {
                    (void)hipGetLastError();
                }
            }
            if (sh.next_copy < sh.n_chunks || (staged && sh.next_unstage < sh.n_chunks)) all_done = false;
        }
        if (all_done) break;
        if (progressed) {
            idle = 0;
            continue;
        }
        // nothing to do yet: look at the render streams now and then (a finished stream finishes all its chunks), else
        // spin -- the wait is microseconds to a few hundred of them
        if ((++idle & 255u) == 0u || D > 1) {
            for (uint32_t k = 0; k < D; k++)
                if (!shares[k].stream_done && shares[k].next_copy < shares[k].n_chunks) RTC_TRY(poll_stream(k));
        }
        if (idle > 4096u) std::this_thread::yield();
    }
    for (uint32_t k = 0; k < D; k++) {
        HIP_TRY(hipSetDevice(g_state[k].device));
        for (hipStream_t c : g_state[k].s_copy2) HIP_TRY(hipStreamSynchronize(c));
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
#ifdef RTC_DEV_SWITCHES
    t_copied = since();
#endif
    // ---- statistics -------------------------------------------------------------------------------------------------
    rtc_stats total;
    std::memset(&total, 0, sizeof(total));
    for (uint32_t k = 0; k < D; k++) {
        DevState& S = g_state[k];
        HIP_TRY(hipSetDevice(S.device));
        HIP_TRY(hipStreamSynchronize(S.s_render));
        rtc_stats s;
        RTC_TRY(ctx_collect(S.ctx, 1u, &s));
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
#ifdef RTC_DEV_SWITCHES
    if (seam_trace)
        std::fprintf(stderr, "rtc_render_ex: queued %.3f ms, first chunk's copy queued %.3f, last %.3f, copies done %.3f, stats %.3f; kernel %.3f ms, %u chunk(s)\n",
                     t_queued, t_first, t_last, t_copied, since(), total.kernel_ms, shares[0].n_chunks);
#endif
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
