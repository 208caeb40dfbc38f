/*
 * fr_host.hip — from HBM into the caller's Vec<RGB>-shaped HOST buffer (src/lib.rs:253-270: get_image
 * returns a freshly allocated Vec every call, src/lib.rs:266-267).
 *
 * What the pieces cost on an MI355X box (tools/ubench/host_path.hip, 805 MB = one 16384^2 image):
 *   D2H into pinned memory                        14.1 ms (57 GB/s, the PCIe link)
 *   hipHostRegister of a never-touched buffer     55-77 ms (the driver faults every 4-KiB page in, serially)
 *   first touch by T threads, 4-KiB pages         ~55 ms whatever T is (the faults serialise on the mm lock)
 *   first touch by 8 / 16 threads after madvise(MADV_HUGEPAGE)   4.4 / 3.0 ms
 *   hipHostRegister of touched memory             ~0.9 ms per 64 MiB;  hipHostUnregister ~0.02 ms
 *   a pinned staging ring + memcpy threads        16-19 ms resident, 57-78 ms fresh: worse on both counts
 * So: the image is rendered band by band into device scratch (all kernels are enqueued up front), and
 * the caller's buffer is walked in page-aligned chunks of 64 MiB — huge-page hint + parallel first
 * touch where the pages do not exist yet, pin, DMA behind the band that completes the chunk — so that
 * faulting, pinning, rendering and copying all overlap and the call costs about what the slowest of
 * them does.  Copies go straight to their final place: no staging, no second pass over the bytes.
 *
 * GUI-SIZED frames (up to 3840 x 2160 RGBA, round 4) take another road: host_render_staged.  The device copies (a kernel of
 * the library's own: fr_launch_copy_out) into a pinned buffer of the LIBRARY's and the calling thread (with a few helpers for the larger frames) copies the bytes out —
 * the caller's pages are never mapped, pinned or registered with the driver.  Why: the reference's GUI gets a fresh Vec
 * from every get_image and drops it after the upload (src/gui.rs:56-82); pages that the driver had registered for DMA —
 * by hipHostRegister here or by the runtime's own in-place pinning of a pageable copy — make the kernel driver
 * revalidate the process's user-pointer mappings when they are unmapped, with the process's hardware queues stopped
 * meanwhile (a 1 ms frame measured at 7-28 ms, profiles/r03_gui_fresh_buffer_pattern.txt).  Rendering, D2H and the copy
 * out are pipelined in bands, so the second pass over the bytes costs little (profiles/r04_gui_staging.txt).
 */
#include <sys/mman.h>
#include <unistd.h>

#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "fr_ctx.h"

#ifndef MADV_HUGEPAGE
#define MADV_HUGEPAGE 14
#endif

namespace fr {

namespace {

constexpr size_t kPage = 4096;
constexpr size_t kHuge = (size_t)2 << 20;
constexpr size_t kChunk = (size_t)64 << 20;
constexpr size_t kPinThreshold = (size_t)16 << 20; /* below: one kernel + one plain copy */
constexpr size_t kStageMax = (size_t)40 << 20;     /* up to a 3840 x 2160 RGBA frame (33.2 MB): the staged road */
constexpr size_t kSdmaMin = (size_t)5 << 20;       /* bands from here up leave HBM through the copy engine, smaller ones through a kernel */

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

bool trace_enabled() {
    static const bool on = getenv("FR_TRACE") != nullptr;
    return on;
}

int touch_threads() {
    static const int n = [] {
        const char *e = getenv("FR_TOUCH_THREADS");
        if (e && atoi(e) > 0) return atoi(e) > 64 ? 64 : atoi(e);
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        return (int)(cpus >= 16 ? 8 : cpus >= 4 ? cpus / 2 : 1);
    }();
    return n;
}

/* are (a sample of) the pages of [p, p+len) resident?  One mincore call; unknown = "yes". */
bool looks_resident(uint8_t *p, size_t len) {
    uint8_t *a = reinterpret_cast<uint8_t *>(reinterpret_cast<uintptr_t>(p) & ~(kPage - 1));
    const size_t span = (size_t)(p + len - a);
    const size_t pages = (span + kPage - 1) / kPage;
    std::vector<unsigned char> vec(pages);
    if (mincore(a, span, vec.data()) != 0) return true;
    size_t missing = 0;
    for (size_t k = 0; k < pages; k++) missing += !(vec[k] & 1);
    return missing * 16 < pages; /* a few swapped-out pages are not worth the threads */
}

/* Write-fault every page of [p, p+len) without changing a byte (volatile read, same value back). */
void touch_range(uint8_t *p, size_t len) {
    volatile uint8_t *v = p;
    for (size_t k = 0; k < len; k += kPage) v[k] = v[k];
    if (len) v[len - 1] = v[len - 1];
}

}  // namespace

/* First touch of a fresh buffer: ask for huge pages (2 MiB: 512x fewer faults, and the kernel zeroes
 * them outside the contended lock), then fault the range in from several threads. */
void prefault(void *ptr, size_t len) {
    uint8_t *p = static_cast<uint8_t *>(ptr);
    uint8_t *ha = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(p) + kHuge - 1) & ~(kHuge - 1));
    uint8_t *hb = reinterpret_cast<uint8_t *>(reinterpret_cast<uintptr_t>(p + len) & ~(kHuge - 1));
    if (hb > ha) (void)madvise(ha, (size_t)(hb - ha), MADV_HUGEPAGE); /* a hint; failure is fine */
    const int T = touch_threads();
    if (T <= 1 || len < 8 * kHuge) {
        touch_range(p, len);
        return;
    }
    std::atomic<size_t> next{0};
    const size_t piece = 4 * kHuge;
    auto work = [&] {
        for (;;) {
            const size_t a = next.fetch_add(piece);
            if (a >= len) break;
            touch_range(p + a, a + piece < len ? piece : len - a);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

/* ---- ChunkPinner: walk a host buffer in page-aligned chunks, faulting (background) and pinning ---- */

/* End of the chunk that starts at byte `a` of a buffer at `out`: 64 MiB further (16 MiB over the last
 * stretch, so that little remains to be copied when the last kernel ends), moved back to a page boundary
 * of the HOST address so that neighbouring pins never share a page. */
size_t ChunkPinner::chunk_end(const uint8_t *out, size_t need, size_t a) {
    const size_t left = need - a;
    size_t b = a + (left <= kChunk + kChunk / 4 ? kChunk / 4 : kChunk);
    if (b >= need) return need;
    b -= (reinterpret_cast<uintptr_t>(out) + b) & (kPage - 1);
    return b > a ? b : need;
}

ChunkPinner::ChunkPinner(uint8_t *out, size_t need, bool portable, const std::vector<size_t> *byte_order)
    : out_(out), need_(need), flags_(portable ? hipHostRegisterPortable : hipHostRegisterDefault) {
    for (size_t a = 0; a < need;) {
        const size_t b = chunk_end(out, need, a);
        bounds_.push_back(a);
        a = b;
    }
    bounds_.push_back(need);
    const size_t n = chunks();
    state_.reset(new std::atomic<int>[n]);
    const bool fresh = !looks_resident(out, need < 4 * kChunk ? need : 4 * kChunk);
    for (size_t k = 0; k < n; k++) state_[k].store(fresh ? 0 : 1);
    pinned_.assign(n, 0);
    if (fresh) {
        /* the toucher runs ahead of the pinning: chunks in the order they will be needed */
        std::vector<size_t> order;
        std::vector<char> seen(n, 0);
        if (byte_order)
            for (size_t byte : *byte_order) {
                const size_t k = chunk_of(byte < need ? byte : need - 1);
                if (!seen[k]) seen[k] = 1, order.push_back(k);
            }
        for (size_t k = 0; k < n; k++)
            if (!seen[k]) order.push_back(k);
        toucher_ = std::thread([this, order] {
            for (size_t k : order) {
                prefault(out_ + bounds_[k], bounds_[k + 1] - bounds_[k]);
                state_[k].store(1, std::memory_order_release);
            }
        });
    }
}

size_t ChunkPinner::chunk_of(size_t byte) const {
    size_t lo = 0, hi = chunks(); /* bounds_[lo] <= byte < bounds_[hi] */
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (bounds_[mid] <= byte) lo = mid;
        else hi = mid;
    }
    return lo;
}

/* waits until chunk k's pages exist, then pins it (once); false = it cannot be pinned: plain copies */
bool ChunkPinner::pin(size_t k) {
    if (pinned_[k]) return pinned_[k] == 1;
    const uintptr_t base = reinterpret_cast<uintptr_t>(out_);
    const size_t a = bounds_[k], b = bounds_[k + 1];
    const double t0 = now_ms();
    while (state_[k].load(std::memory_order_acquire) == 0) std::this_thread::yield();
    const double t1 = now_ms();
    /* interior boundaries are page boundaries of the host address; the two outer ones are rounded outwards */
    uint8_t *ra = reinterpret_cast<uint8_t *>((base + a) & ~(kPage - 1));
    uint8_t *rb = b < need_ ? out_ + b : reinterpret_cast<uint8_t *>((base + b + kPage - 1) & ~(kPage - 1));
    const hipError_t re = hipHostRegister(ra, (size_t)(rb - ra), flags_);
    bool ok = re == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        ok = re == hipErrorHostMemoryAlreadyRegistered; /* the caller pinned it (fr_pin_host_buffer): even better */
    }
    const double t2 = now_ms();
    {
        std::lock_guard<std::mutex> lk(reg_mu_);
        if (re == hipSuccess) regs_.push_back(ra);
        t_touch += t1 - t0;
        t_reg += t2 - t1;
    }
    pinned_[k] = ok ? 1 : 2;
    return ok;
}

bool ChunkPinner::next(size_t &a, size_t &b, bool &pinned) {
    if (pos_ >= chunks()) return false;
    a = bounds_[pos_];
    b = bounds_[pos_ + 1];
    pinned = pin(pos_);
    pos_++;
    return true;
}

void ChunkPinner::release() {
    if (toucher_.joinable()) toucher_.join();
    for (uint8_t *r : regs_) (void)hipHostUnregister(r);
    regs_.clear();
}

ChunkPinner::~ChunkPinner() { release(); }

/* ---- the copy threads of the staged road ----------------------------------------------------------- */

/* A few threads that sleep until the calling thread has bytes to move: a frame's band is cut into pieces, the caller
 * takes the first and the helpers the rest.  kind 0: memcpy(dst, src, len); kind 1: first touch of [dst, dst + len). */
struct CopyPool {
    struct Piece {
        uint8_t *dst;
        const uint8_t *src;
        size_t len;
        int kind;
    };
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable work_cv, done_cv;
    std::vector<Piece> pieces;
    size_t outstanding = 0;
    bool stop = false;

    explicit CopyPool(int helpers) {
        for (int t = 0; t < helpers; t++) threads.emplace_back([this] { loop(); });
    }
    static void run(const Piece &p) {
        if (p.kind == 0) memcpy(p.dst, p.src, p.len);
        else prefault(p.dst, p.len);
    }
    void loop() {
        for (;;) {
            Piece p;
            {
                std::unique_lock<std::mutex> lk(m);
                work_cv.wait(lk, [&] { return stop || !pieces.empty(); });
                if (pieces.empty()) return; /* stop */
                p = pieces.back();
                pieces.pop_back();
            }
            run(p);
            std::lock_guard<std::mutex> lk(m);
            if (--outstanding == 0) done_cv.notify_all();
        }
    }
    /* hand out pieces; returns at once (the caller works on its own piece, then calls wait()) */
    void post(const std::vector<Piece> &ps) {
        if (ps.empty()) return;
        {
            std::lock_guard<std::mutex> lk(m);
            for (const Piece &p : ps) pieces.push_back(p);
            outstanding += ps.size();
        }
        work_cv.notify_all();
    }
    /* the caller helps with whatever is still queued, then waits for the pieces in flight */
    void wait() {
        for (;;) {
            Piece p;
            {
                std::lock_guard<std::mutex> lk(m);
                if (pieces.empty()) break;
                p = pieces.back();
                pieces.pop_back();
            }
            run(p);
            std::lock_guard<std::mutex> lk(m);
            --outstanding;
        }
        std::unique_lock<std::mutex> lk(m);
        done_cv.wait(lk, [&] { return outstanding == 0; });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        work_cv.notify_all();
        for (auto &t : threads) t.join();
    }
};

void destroy_copy_pool(CopyPool *pool) { delete pool; }

namespace {

bool staging_enabled() {
    static const bool on = [] {
        const char *e = getenv("FR_HOST_STAGING"); /* tuning aid: 0 = round 3's road for GUI-sized frames too */
        return !(e && atoi(e) == 0);
    }();
    return on;
}

int copy_helpers() {
    static const int n = [] {
        const char *e = getenv("FR_COPY_THREADS");
        if (e && atoi(e) >= 1) return atoi(e) > 16 ? 15 : atoi(e) - 1;
        const long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        return (int)(cpus >= 8 ? 3 : cpus >= 4 ? 1 : 0);
    }();
    return n;
}

/* wait for an event with little latency: poll for a while (the DMA of a band takes a tenth of a millisecond), then sleep */
hipError_t wait_event(hipEvent_t e) {
    const double t0 = now_ms();
    for (;;) {
        const hipError_t q = hipEventQuery(e);
        if (q != hipErrorNotReady) return q;
        for (int k = 0; k < 32; k++) _mm_pause();
        if (now_ms() - t0 > 0.5) return hipEventSynchronize(e);
    }
}

/* wait until the copy kernel of a band has published `seq` in the band's flag (pinned host memory): poll — a band takes tens
 * of microseconds — looking at the stream now and then, so that a failed launch or a lost device ends the wait */
hipError_t wait_flag(const unsigned long long *flag, unsigned long long seq, hipStream_t st) {
    const double t0 = now_ms();
    double next_check = 2.0;
    for (;;) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
        for (int k = 0; k < 16; k++) _mm_pause();
        const double dt = now_ms() - t0;
        if (dt > next_check) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) /* everything enqueued has run: the flag is there, or it never will be */
                return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq ? hipSuccess : hipErrorUnknown;
            if (q != hipErrorNotReady) return q;
            next_check = dt + 2.0;
        }
    }
}

}  // namespace

/* GUI-sized frames: render in up to four bands of whole 8-row tiles on two streams, DMA each finished band into the
 * library's pinned staging buffer, copy it out to the caller's buffer while the next band renders and travels. */
int host_render_staged(Ctx &ctx, const fr_config *cfg, int precision, const Opts &o, uint32_t y0, uint32_t y1, uint8_t *out,
                       unsigned bpp) {
    const size_t row_bytes = (size_t)bpp * cfg->width;
    const size_t need = row_bytes * (size_t)(y1 - y0);
    int rc = ctx.reserve(ctx.rgb, need);
    if (rc != FR_OK) return rc;
    uint8_t *scratch = static_cast<uint8_t *>(ctx.rgb.ptr);
    uint8_t *stage = static_cast<uint8_t *>(ctx.stage);
    const bool trace = trace_enabled();
    const double t_start = now_ms();

    /* bands: ~6 MiB each, at most four (a 3840 x 2160 RGB frame: four of 6.2 MB; 1920 x 1080: two; 750 x 500: one).  Few and
     * large: at these sizes the host's HIP calls (5-10 us each) are a visible share of the frame, and every band costs three */
    const uint32_t rows = y1 - y0;
    uint32_t nb = (uint32_t)((need + ((size_t)6 << 20) - 1) / ((size_t)6 << 20));
    if (need >= ((size_t)3 << 20) && nb < 2) nb = 2;
    if (nb > 4) nb = 4;
    if (nb < 1) nb = 1;
    uint32_t band_rows = ((rows + nb - 1) / nb + 7u) & ~7u;
    if (band_rows == 0) band_rows = 8;
    nb = (rows + band_rows - 1) / band_rows;

    Opts ob = o;
    decide_kernel(ctx, cfg, precision, y0, y1, ob, ctx.stream, true); /* ONE view (and one sample) for the frame, not one per band */
    const int pending_sample = ob.pending_sample;
    ob.pending_sample = -1;
    std::vector<double> marks; /* FR_TRACE: host time after every step of the enqueue (which HIP call was slow?) */
    auto mark = [&] {
        if (trace) marks.push_back(now_ms() - t_start);
    };
    mark();
    auto render_on = [&](hipStream_t st, uint32_t ya, uint32_t yb, uint8_t *dst) -> int {
        fr_kparams p;
        fill_params(cfg, ob, p);
        p.nrows = yb - ya;
        p.y_first = ya;
        p.block_rows = p.nrows;
        p.y_stride = 0;
        p.out_rgba = bpp == 4 ? 1u : 0u;
        return render_device(ctx, cfg, p, precision, ob, dst, st);
    };

    hipError_t err = hipSuccess;
    const char *what = "";
    hipStream_t last = ctx.stream;
    const unsigned long long seq0 = ctx.stage_seq + 1; /* band b publishes seq0 + b */
    ctx.stage_seq += nb;
    for (uint32_t b = 0; b < nb && rc == FR_OK && err == hipSuccess; b++) {
        const uint32_t ya = y0 + b * band_rows, yb = ya + band_rows < y1 ? ya + band_rows : y1;
        const size_t a = row_bytes * (size_t)(ya - y0), len = row_bytes * (size_t)(yb - ya);
        hipStream_t st = (b & 1) ? ctx.stream2 : ctx.stream;
        last = st;
        rc = render_on(st, ya, yb, scratch + a);
        mark();
        if (rc != FR_OK) break;
        /* the band leaves HBM through a copy KERNEL behind its render kernel on the same stream — 16-byte stores into the
         * pinned staging buffer, the band's sequence number into its flag when the last workgroup is done — not through the
         * runtime's copy machinery: in a process that had moved gigabytes through registered host buffers, hipMemcpyAsync was
         * seen to hold the calling thread for 6-8 ms now and then (FR_TRACE marks, bench.py's gui_latency block); a launch
         * never did.  Two HIP calls a band, no event.  Bands alternate between two streams: band b + 1 renders while band b
         * travels */
        if (len < kSdmaMin) {
            if ((err = fr_launch_copy_out(scratch + a, static_cast<uint8_t *>(ctx.stage_dev) + a, len, ctx.stage_counters + b,
                                          ctx.stage_flags_dev + b, seq0 + b, st)) != hipSuccess)
                what = "fr_launch_copy_out";
        } else {
            /* ... except the 6 MB bands of a 3840 x 2160 frame: the copy ENGINE moves those at the link's full rate (a band in
             * 0.11 ms against 0.15 through the kernel's stores: 0.76 ms a frame against 0.94), and what a rare slow call costs
             * weighs less on a frame of that size */
            hipEvent_t ec;
            rc = ctx.event(b, &ec);
            if (rc != FR_OK) break;
            if ((err = hipMemcpyAsync(stage + a, scratch + a, len, hipMemcpyDeviceToHost, st)) != hipSuccess) what = "hipMemcpyAsync";
            else if ((err = hipEventRecord(ec, st)) != hipSuccess) what = "hipEventRecord";
        }
        mark();
    }
    ctx.post_sample(pending_sample, last); /* a first frame of the view: its statistics, behind the last band */
    mark();
    const double t_enqueued = now_ms();

    /* the copy out: the caller alone for small bands, with the helpers from 2 MiB a band; pages the caller has never
     * touched (a fresh Vec) are faulted in — huge-page hint, all threads — while the first band renders */
    const int helpers = need >= ((size_t)3 << 20) ? copy_helpers() : 0;
    /* made by the first frame that wants them: 0.1-0.3 ms of that frame */
    if (helpers > 0 && !ctx.copy_pool) ctx.copy_pool = new CopyPool(helpers);
    CopyPool *pool = helpers > 0 ? ctx.copy_pool : nullptr;

    auto spread = [&](uint8_t *dst, const uint8_t *src, size_t len, int kind) {
        if (!pool || len < ((size_t)1 << 20)) {
            CopyPool::run(CopyPool::Piece{dst, src, len, kind});
            return;
        }
        const size_t parts = (size_t)helpers + 1;
        const size_t piece = ((len + parts - 1) / parts + 4095) & ~(size_t)4095;
        std::vector<CopyPool::Piece> ps;
        for (size_t off = piece; off < len; off += piece)
            ps.push_back(CopyPool::Piece{dst + off, src ? src + off : nullptr, off + piece < len ? piece : len - off, kind});
        pool->post(ps);
        CopyPool::run(CopyPool::Piece{dst, src, piece < len ? piece : len, kind});
        pool->wait();
    };
    if (rc == FR_OK && err == hipSuccess && need >= ((size_t)1 << 20) && !looks_resident(out, need)) spread(out, nullptr, need, 1);
    const double t_touched = now_ms();
    for (uint32_t b = 0; b < nb && rc == FR_OK && err == hipSuccess; b++) {
        const uint32_t ya = y0 + b * band_rows, yb = ya + band_rows < y1 ? ya + band_rows : y1;
        const size_t a = row_bytes * (size_t)(ya - y0), len = row_bytes * (size_t)(yb - ya);
        if (len < kSdmaMin) {
            err = wait_flag(ctx.stage_flags + b, seq0 + b, (b & 1) ? ctx.stream2 : ctx.stream);
        } else {
            hipEvent_t ec;
            rc = ctx.event(b, &ec);
            if (rc != FR_OK) break;
            err = wait_event(ec);
        }
        if (err != hipSuccess) {
            what = "waiting for a band to arrive in the staging buffer";
            break;
        }
        spread(out + a, stage + a, len, 0);
    }
    /* drain whatever an error left in flight: the scratch and the staging buffer are reused by the next call */
    if (rc != FR_OK || err != hipSuccess) {
        (void)hipStreamSynchronize(ctx.stream);
        (void)hipStreamSynchronize(ctx.stream2);
        (void)hipStreamSynchronize(ctx.copy_stream);
    }
    if (trace) {
        fprintf(stderr, "[fr_host] staged %zu bytes in %u bands: enqueued at %.3f ms, first touch until %.3f, copied out at %.3f", need, nb,
                t_enqueued - t_start, t_touched - t_start, now_ms() - t_start);
        if (t_enqueued - t_start > 1.0) { /* a slow enqueue: after decide | per band: launch, memcpy, event | sample */
            fprintf(stderr, "  [marks:");
            for (double m : marks) fprintf(stderr, " %.3f", m);
            fprintf(stderr, "]");
        }
        fprintf(stderr, "\n");
    }
    if (rc != FR_OK) return rc;
    if (err != hipSuccess) return fail_hip(err, what);
    return FR_OK;
}

/* ---- rows [y0, y1) into a host buffer ----------------------------------------------------------- */

int host_render_rows(Ctx &ctx, const fr_config *cfg, int precision, const Opts &o, uint32_t y0, uint32_t y1,
                     uint8_t *out, unsigned bpp) {
    const size_t row_bytes = (size_t)bpp * cfg->width;
    const size_t need = row_bytes * (size_t)(y1 - y0);
    /* GUI-sized frames: through the library's own pinned staging buffer (the caller's pages are never registered) */
    if (need <= kStageMax && staging_enabled() && ctx.reserve_stage(need > ((size_t)40 << 20) ? need : (size_t)40 << 20) == FR_OK)
        return host_render_staged(ctx, cfg, precision, o, y0, y1, out, bpp);
    int rc = ctx.reserve(ctx.rgb, need);
    if (rc != FR_OK) return rc;
    uint8_t *scratch = static_cast<uint8_t *>(ctx.rgb.ptr);
    const bool trace = trace_enabled();
    const double t_start = now_ms();

    const Opts *use_opts = &o;
    auto render_on = [&](hipStream_t st, uint32_t ya, uint32_t yb, uint8_t *dst) -> int {
        fr_kparams p;
        fill_params(cfg, *use_opts, p);
        p.nrows = yb - ya;
        p.y_first = ya;
        p.block_rows = p.nrows;
        p.y_stride = 0;
        p.out_rgba = bpp == 4 ? 1u : 0u;
        return render_device(ctx, cfg, p, precision, *use_opts, dst, st);
    };

    if (need < kPinThreshold) {
        /* small image (GUI frames): one kernel, one copy; the runtime's own staging is the fastest here */
        rc = render_on(ctx.stream, y0, y1, scratch);
        if (rc != FR_OK) return rc;
        HIP_TRY(hipMemcpyAsync(out, scratch, need, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        return FR_OK;
    }

    /* 0. which kernel suits the view is decided ONCE, from a sample of all the rows (each band would otherwise take
     *    its own sample, blocking, in front of its launch) */
    Opts ob = o;
    decide_kernel(ctx, cfg, precision, y0, y1, ob, ctx.stream, true);
    const int pending_sample = ob.pending_sample; /* a first frame of a GUI-sized view: its sample goes behind the bands */
    ob.pending_sample = -1;
    use_opts = &ob;
    /* 1. every band's kernel, enqueued up front: the GPU renders while the host prepares the buffer.  Bands
     *    are ~64 MiB of whole 8-row tiles (smaller over the last stretch) and alternate between two streams,
     *    so that the tail of one band's kernel — its few longest strips — overlaps the start of the next.
     *    They are issued in BIT-REVERSED order: the set's interior makes neighbouring bands similarly cheap
     *    or similarly expensive (default view: 0.3 ms at the top, 4 ms in the middle), and in image order the
     *    copy engine first starves behind the expensive middle, then finds a third of the image finished at
     *    once; a bit-reversed prefix samples the image evenly, so finished bytes arrive at a steady rate. */
    struct Band {
        uint32_t ya, yb;
        size_t a, b; /* byte range of the image */
    };
    std::vector<Band> bands;
    {
        uint32_t ya = y0;
        size_t a = 0;
        while (ya < y1) {
            const size_t target = ChunkPinner::chunk_end(out, need, a);
            uint64_t rows = (target - a + row_bytes - 1) / row_bytes;
            rows = (rows + 7) / 8 * 8; /* whole 8-row tiles */
            const uint32_t yb = (uint32_t)((uint64_t)ya + rows < y1 ? ya + rows : y1);
            const size_t b = row_bytes * (size_t)(yb - y0);
            bands.push_back(Band{ya, yb, a, b});
            ya = yb;
            a = b;
        }
    }
    std::vector<size_t> order;
    {
        size_t pow2 = 1;
        int bits = 0;
        while (pow2 < bands.size()) pow2 <<= 1, bits++;
        for (size_t i = 0; i < pow2; i++) {
            size_t r = 0;
            for (int k = 0; k < bits; k++) r |= ((i >> k) & 1u) << (bits - 1 - k);
            if (r < bands.size()) order.push_back(r);
        }
    }
    hipError_t err = hipSuccess;
    const char *what = "";
    std::vector<hipEvent_t> tk, tc; /* FR_TRACE only: when each band's kernel / copy finished */
    hipEvent_t t_zero = nullptr;
    if (trace && hipEventCreate(&t_zero) == hipSuccess) (void)hipEventRecord(t_zero, ctx.stream);
    for (size_t i = 0; i < order.size() && rc == FR_OK && err == hipSuccess; i++) {
        const Band &bd = bands[order[i]];
        hipStream_t st = (i & 1) ? ctx.stream2 : ctx.stream; /* three streams measured worse: the first band finishes later */
        rc = render_on(st, bd.ya, bd.yb, scratch + bd.a);
        if (rc != FR_OK) break;
        hipEvent_t e;
        rc = ctx.event(i, &e);
        if (rc != FR_OK) break;
        if ((err = hipEventRecord(e, st)) != hipSuccess) what = "hipEventRecord";
        if (t_zero) {
            hipEvent_t te;
            if (hipEventCreate(&te) == hipSuccess) {
                (void)hipEventRecord(te, st);
                tk.push_back(te);
            }
        }
    }
    ctx.post_sample(pending_sample, (order.size() & 1) ? ctx.stream : ctx.stream2); /* behind the LAST band enqueued */
    const double t_launched = now_ms();

    /* 2. the caller's buffer, band by band in the same order: first touch (a background thread, ahead of us)
     *    and pin of the 64 MiB chunks the band lies in, then its DMA behind its kernel — split where two pins
     *    meet (one DMA must not span two registrations) */
    std::vector<size_t> byte_order;
    for (size_t k : order) byte_order.push_back(bands[k].a);
    ChunkPinner pinner(out, need, false, &byte_order);
    for (size_t i = 0; i < order.size() && rc == FR_OK && err == hipSuccess; i++) {
        const Band &bd = bands[order[i]];
        hipEvent_t e;
        rc = ctx.event(i, &e); /* recorded above */
        if (rc != FR_OK) break;
        bool waited = false;
        for (size_t pos = bd.a; pos < bd.b && err == hipSuccess;) {
            const size_t k = pinner.chunk_of(pos);
            const size_t end = pinner.bound(k + 1) < bd.b ? pinner.bound(k + 1) : bd.b;
            if (pinner.pin(k)) {
                if (!waited) {
                    if ((err = hipStreamWaitEvent(ctx.copy_stream, e, 0)) != hipSuccess) what = "hipStreamWaitEvent";
                    waited = true;
                }
                if (err == hipSuccess &&
                    (err = hipMemcpyAsync(out + pos, scratch + pos, end - pos, hipMemcpyDeviceToHost, ctx.copy_stream)) != hipSuccess)
                    what = "hipMemcpyAsync";
            } else {
                /* memory that cannot be pinned: a plain (staged) copy once the band is done */
                if ((err = hipEventSynchronize(e)) != hipSuccess) what = "hipEventSynchronize";
                else if ((err = hipMemcpy(out + pos, scratch + pos, end - pos, hipMemcpyDeviceToHost)) != hipSuccess) what = "hipMemcpy";
            }
            pos = end;
        }
        if (t_zero) {
            hipEvent_t te;
            if (hipEventCreate(&te) == hipSuccess) {
                (void)hipEventRecord(te, ctx.copy_stream);
                tc.push_back(te);
            }
        }
    }
    const double t_enqueued = now_ms();
    /* always drain both streams and unpin before returning, error or not */
    hipError_t e1 = hipStreamSynchronize(ctx.stream);
    const hipError_t e1b = hipStreamSynchronize(ctx.stream2);
    if (e1 == hipSuccess) e1 = e1b;
    hipError_t e2 = hipStreamSynchronize(ctx.copy_stream);
    const double t_synced = now_ms();
    pinner.release();
    if (trace)
        fprintf(stderr,
                "[fr_host] %zu bytes: kernels enqueued at %.2f ms, waited for first touch %.2f, pinning %.2f, copies "
                "enqueued at %.2f, drained at %.2f, unpinned at %.2f\n",
                need, t_launched - t_start, pinner.t_touch, pinner.t_reg, t_enqueued - t_start, t_synced - t_start,
                now_ms() - t_start);
    if (t_zero) {
        fprintf(stderr, "[fr_host]   band kernels done at (ms):");
        for (hipEvent_t e : tk) {
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, t_zero, e);
            fprintf(stderr, " %.1f", ms);
            (void)hipEventDestroy(e);
        }
        fprintf(stderr, "\n[fr_host]   chunk copies done at (ms):");
        for (hipEvent_t e : tc) {
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, t_zero, e);
            fprintf(stderr, " %.1f", ms);
            (void)hipEventDestroy(e);
        }
        fprintf(stderr, "\n");
        (void)hipEventDestroy(t_zero);
    }
    if (rc != FR_OK) return rc;
    if (err != hipSuccess) return fail_hip(err, what);
    if (e1 != hipSuccess) return fail_hip(e1, "hipStreamSynchronize(stream)");
    if (e2 != hipSuccess) return fail_hip(e2, "hipStreamSynchronize(copy_stream)");
    return FR_OK;
}

}  // namespace fr

using namespace fr;

/* Shared body of fr_render_rgb8 / fr_render_rows_rgb8(_opts) / fr_render_rows_rgba8. */
int fr_host_render_rows(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out, size_t out_len,
                        unsigned bpp, const fr_render_opts *opts) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (y0 > y1) return fail(FR_ERR_INVALID_ARGUMENT, "y0 > y1");
    if (y1 > cfg->height) return fail(FR_ERR_INVALID_ARGUMENT, "y1 > height");
    int rc = check_precision(precision);
    Opts o;
    if (rc == FR_OK) rc = resolve_opts(opts, o);
    if (rc != FR_OK) return rc;
    const size_t need = (size_t)bpp * cfg->width * (size_t)(y1 - y0);
    if (need == 0) return FR_OK;
    if (!out) return fail(FR_ERR_INVALID_ARGUMENT, "out is NULL");
    if (out_len < need)
        return fail(FR_ERR_BUFFER_TOO_SMALL, bpp == 4 ? "out_len < 4*width*(y1-y0)" : "out_len < 3*width*(y1-y0)");
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return host_render_rows(*ctx, cfg, precision, o, y0, y1, out, bpp);
}
