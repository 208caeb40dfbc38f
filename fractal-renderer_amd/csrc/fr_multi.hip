/*
 * fr_multi.hip — get_image across a set of devices from ONE process.
 *
 * The reference is one process calling get_image(&Config) once (src/main.rs:16, src/lib.rs:253) and
 * its rayon loop spreads the rows over every core (src/lib.rs:256-258).  Here the rows are spread over
 * every device of a set, row-block-cyclically (block b -> device b % n: the set's interior sits in
 * the middle rows of the default view, contiguous bands would be badly unbalanced), by ONE host thread
 * with its own streams per device.  Pixels are independent (src/lib.rs:259-264), so the only exchange is
 * getting the finished bytes to where the caller wants them:
 *
 *   fr_render_rgb8_multi         the caller's HOST buffer (what get_image returns): every device DMAs
 *                                each finished block straight to its final place over its OWN PCIe link;
 *   fr_render_rgb8_multi_device  HBM of the set's first device: peer-to-peer DMA over xGMI, or grouped
 *                                ncclSend / ncclRecv on a communicator from ncclCommInitAll (RCCL).
 *
 * A device renders several of its blocks per kernel launch ("chunk": big chunks first — launches of
 * >= 1024 rows run at the single-launch rate, but never more than a quarter of the device's blocks — then
 * a geometric tail 4, 2, 1 so that only one block's bytes remain to be moved when the last kernel ends),
 * chunks alternate between two streams so that a
 * kernel's tail overlaps the next kernel's start, and chunk c is on the wire while chunk c+1 renders.
 */
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <cstdlib>
#include <thread>

#include "fr_ctx.h"

namespace fr {

namespace {

constexpr uint32_t kDefaultBlockRows = 256;
constexpr uint32_t kMaxChunkBlocks = 8;
constexpr const char *kEchoError = "another device of the set failed";

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

/* ---- RCCL, loaded on first use (the library itself links only the HIP runtime) ---------------- */

typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclUint8 = 1; /* ncclDataType_t: ncclUint8 (rccl.h) */

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<ncclComm_t> comms;

    int load() {
        if (handle) return FR_OK;
        /* FR_RCCL_LIBRARY names the library to load instead of searching (tests use it to make the load fail).
         * RTLD_NOLOAD first: a process that already carries RCCL (PyTorch bundles one) must not get a second */
        const char *forced = getenv("FR_RCCL_LIBRARY");
        for (int flags : {RTLD_NOW | RTLD_NOLOAD, RTLD_NOW}) {
            if (forced && *forced) {
                handle = dlopen(forced, flags);
            } else {
                for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                    handle = dlopen(name, flags);
                    if (handle) break;
                }
            }
            if (handle) break;
        }
        if (!handle) {
            const char *m = dlerror(); /* ONE call: it clears the message it returns */
            return fail(FR_ERR_HIP, std::string("cannot load librccl: ") + (m ? m : "not found"));
        }
        auto sym = [&](const char *n) { return dlsym(handle, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(sym("ncclCommAbort")); /* optional */
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            dlclose(handle);
            handle = nullptr;
            return fail(FR_ERR_HIP, "librccl lacks a required symbol");
        }
        return FR_OK;
    }
    int check(ncclResult_t r, const char *what) {
        if (r == 0) return FR_OK;
        return fail(FR_ERR_HIP, std::string(what) + ": " + GetErrorString(r));
    }
    /* Ownership rule: communicator r is used, aborted and destroyed ONLY by logical device r's worker
     * thread — nobody ever pulls a handle from under a thread that may be inside an RCCL call on it.  A rank that
     * fails, or that sees the set's abort flag while it waits for its transfers, aborts ITS OWN communicator
     * (ncclCommAbort releases the operations in flight on it: its peers' matching device-side waits then fail or
     * finish, and each of them aborts its own in turn).  After a failed render run_multi() has every worker drop
     * what is left (abort, not destroy: a communicator whose peers were aborted may not be destroyed collectively)
     * and the next RCCL render makes new ones. */
    void abort_own(uint32_t r) {
        if (r >= comms.size() || !comms[r]) return;
        if (CommAbort) (void)CommAbort(comms[r]);
        comms[r] = nullptr; /* (without ncclCommAbort in the library: leaked rather than hung on) */
        aborted.store(true, std::memory_order_release);
    }
    void destroy_own(uint32_t r) {
        if (r >= comms.size() || !comms[r]) return;
        (void)CommDestroy(comms[r]);
        comms[r] = nullptr;
    }
    std::atomic<bool> aborted{false};
};

/* ---- one worker thread per logical device ----------------------------------------------------- */

struct Worker {
    Ctx ctx;
    int index = 0;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, stop = false, done = false;
    int rc = FR_OK;
    std::string err;
    std::vector<hipEvent_t> tev; /* timing events, two per chunk */

    void loop(int device) {
        int r = ctx.create(device);
        finish(r);
        for (;;) {
            std::function<int()> t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has_task || stop; });
                if (stop) break;
                t = std::move(task);
                has_task = false;
            }
            finish(t());
        }
        for (hipEvent_t e : tev) (void)hipEventDestroy(e);
        tev.clear();
        ctx.destroy();
    }
    void finish(int r) {
        std::lock_guard<std::mutex> lk(m);
        rc = r;
        err = r == FR_OK ? std::string() : last_error(); /* the worker's thread-local message */
        done = true;
        cv.notify_all();
    }
    void post(std::function<int()> t) {
        std::lock_guard<std::mutex> lk(m);
        task = std::move(t);
        has_task = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    int timing_event(size_t k, hipEvent_t *out) {
        while (tev.size() <= k) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            tev.push_back(e);
        }
        *out = tev[k];
        return FR_OK;
    }
};

struct DeviceSet {
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<int> devices;
    bool distinct = true;
    std::mutex call_mu; /* one multi-device render at a time */
    Rccl rccl;
};
DeviceSet *g_set = nullptr;
thread_local fr_multi_stats tl_stats;

/* Local block index ranges [j0, j1) a device renders per launch (see the file header). */
std::vector<std::pair<uint32_t, uint32_t>> chunk_schedule(uint32_t nb) {
    /* at most a quarter of the device's blocks per launch: its transfer takes about as long as its rendering
     * (both scale as 1/N), so the link has to start early and stay busy */
    const uint32_t cap = std::max(1u, std::min(kMaxChunkBlocks, nb / 4));
    std::vector<uint32_t> sizes;
    uint32_t rem = nb, s = 1;
    while (rem > 0) {
        const uint32_t t = std::min(std::min(s, rem), cap);
        sizes.push_back(t);
        rem -= t;
        s *= 2;
    }
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    uint32_t j = 0;
    for (auto it = sizes.rbegin(); it != sizes.rend(); ++it) {
        ranges.emplace_back(j, j + *it);
        j += *it;
    }
    return ranges;
}

/* Who renders which row blocks (partition.py: shares / rank_chunks — the same dealing, so that both multi-GPU paths can be
 * read against each other).  q = 1: block b -> device b % n.  q = 2, 4: the SINK (device 0, which also receives every other
 * device's bytes) keeps every q-th of its blocks and the rest of them are dealt round-robin to the other devices; q = 0: the
 * sink renders nothing.  Every share is a short list of arithmetic progressions (first block, stride): one progression is
 * what one launch of the block-cyclic render covers. */
struct Share {
    uint32_t first, stride;
};
std::vector<Share> shares_of(uint32_t r, uint32_t n, int q) {
    if (n == 1 || q == 1) return {Share{r, n}};
    const uint32_t peers = n - 1;
    if (r == 0) return q == 0 ? std::vector<Share>{} : std::vector<Share>{Share{0, (uint32_t)q * n}};
    std::vector<Share> out{Share{r, n}};
    if (q == 0) out.push_back(Share{(r - 1) * n, peers * n});
    else
        for (uint32_t c = 1; c < (uint32_t)q; c++) out.push_back(Share{(c + (uint32_t)q * (r - 1)) * n, (uint32_t)q * peers * n});
    return out;
}
/* one launch: blocks first, first + stride, ... (count of them) */
struct Chunk {
    uint32_t first, stride, count;
};
std::vector<Chunk> rank_chunks(uint32_t r, uint32_t n, uint32_t nblocks, int q) {
    std::vector<Chunk> out;
    for (const Share &sh : shares_of(r, n, q)) {
        if (sh.first >= nblocks) continue;
        const uint32_t mine = (nblocks - sh.first + sh.stride - 1) / sh.stride;
        /* the cuts of the LONGEST progression of this stride, so that devices one block short cut at the same places */
        for (const auto &rg : chunk_schedule((nblocks + sh.stride - 1) / sh.stride)) {
            const uint32_t j0 = rg.first, j1 = std::min(rg.second, mine);
            if (j1 > j0) out.push_back(Chunk{sh.first + j0 * sh.stride, sh.stride, j1 - j0});
        }
    }
    return out;
}
std::atomic<int> g_root_share{1};

enum class Sink { Host, PeerCopy, Rccl };

/* Host sink: the caller's buffer is made DMA-able in page-aligned 64 MiB chunks, front to back (every device's
 * blocks advance through the image together), by a background thread while the devices render; device threads sleep
 * on a condition variable until the chunks under their block are pinned.  What pinning costs was measured
 * (tools/ubench/host_path.hip, profiles/r03_host_register_scaling.txt): hipHostRegister does NOT scale with calling
 * threads — memory the caller touched with 4-KiB pages pins at 42 GB/s from one thread and 32-36 GB/s from 2-8 (the
 * kernel's mm lock) — but memory backed by huge pages, which is what this library's own first touch produces for a
 * buffer whose pages do not exist yet (a fresh Vec: fr_host.hip), pins at 430-550 GB/s whatever the chunk size or
 * thread count, and memory that was registered before re-registers for next to nothing.  Hence ONE pinning thread
 * (FR_PIN_THREADS overrides), and fr_pin_host_buffer for callers that keep a buffer. */
struct PinProgress {
    ChunkPinner pinner;
    std::mutex m;
    std::condition_variable cv;
    std::vector<char> state; /* per chunk: 0 pending, 1 pinned, 2 cannot be pinned */
    bool failed = false;
    std::atomic<size_t> next{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> pool;

    static int threads() {
        static const int n = [] {
            const char *e = getenv("FR_PIN_THREADS");
            if (e && atoi(e) > 0) return atoi(e) > 16 ? 16 : atoi(e);
            return 1;
        }();
        return n;
    }
    PinProgress(uint8_t *dst, size_t need) : pinner(dst, need, true), state(pinner.chunks(), 0) {
        const size_t n = pinner.chunks();
        const int T = (int)std::min<size_t>((size_t)threads(), n);
        for (int t = 0; t < T; t++)
            pool.emplace_back([this, n] {
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= n || stop.load(std::memory_order_acquire)) break;
                    const bool ok = pinner.pin(k);
                    {
                        std::lock_guard<std::mutex> lk(m);
                        state[k] = ok ? 1 : 2;
                        if (!ok) failed = true;
                    }
                    cv.notify_all();
                    if (!ok) break;
                }
            });
    }
    /* sleeps until bytes [a, b) are pinned; false = some chunk cannot be pinned (plain copies) or `abort` was set */
    bool wait_for(size_t a, size_t b, const std::atomic<bool> &abort) {
        const size_t k0 = pinner.chunk_of(a), k1 = pinner.chunk_of(b - 1);
        std::unique_lock<std::mutex> lk(m);
        bool ready = false;
        cv.wait(lk, [&] {
            if (failed || abort.load(std::memory_order_acquire)) return true;
            for (size_t k = k0; k <= k1; k++)
                if (state[k] != 1) return false;
            return ready = true;
        });
        return ready;
    }
    void wake() {
        std::lock_guard<std::mutex> lk(m);
        cv.notify_all();
    }
    /* the devices have drained their streams: stop pinning what nobody will use, then unpin everything */
    void finish() {
        stop.store(true, std::memory_order_release);
        for (auto &t : pool) t.join();
        pool.clear();
        pinner.release();
    }
    ~PinProgress() { finish(); }
};

struct Job {
    const fr_config *cfg;
    int precision;
    Opts opts;
    uint32_t block_rows, nblocks, n;
    Sink sink;
    uint8_t *dst; /* host buffer, or the image in the first device's memory */
    size_t dst_len;
    int root_device;
    PinProgress *pins; /* Host sink only */
    DeviceSet *set;
    std::atomic<bool> *abort; /* a device failed: the others stop starting new work (RCCL: no new groups) */
    int root_share;           /* see shares_of */
};

/* test aid (fr_debug_inject_multi_failure): logical device `g_inject_device` fails before its chunk
 * `g_inject_chunk` of the next multi-device render, once */
std::atomic<int> g_inject_device{-1}, g_inject_chunk{0};

void block_range(const Job &j, uint32_t b, uint32_t &y0, uint32_t &y1) {
    y0 = b * j.block_rows;
    const uint64_t e = (uint64_t)y0 + j.block_rows;
    y1 = e < j.cfg->height ? (uint32_t)e : j.cfg->height;
}

/* what device `r` of the set does for one image */
int device_job(Worker &w, const Job &j, fr_multi_stats *stats) {
    Ctx &ctx = w.ctx;
    const uint32_t r = (uint32_t)w.index, n = j.n;
    const size_t row_bytes = (size_t)3 * j.cfg->width;
    const size_t need_total = row_bytes * (size_t)j.cfg->height;
    HIP_TRY(hipSetDevice(ctx.hip_device));
    const double t_job = now_ms();
    const bool in_place = j.sink != Sink::Host && r == 0; /* the root renders straight into the image */
    hipEvent_t tc0 = nullptr, tc1 = nullptr; /* around this device's transfers on its copy stream (the last two timing events) */
    bool tc_started = false;
    uint64_t bytes_moved = 0;
    /* every device's launches, step by step: step s renders chunk s of this device and moves chunk s of every peer */
    std::vector<std::vector<Chunk>> all(n);
    size_t nsteps = 0;
    for (uint32_t d = 0; d < n; d++) {
        all[d] = rank_chunks(d, n, j.nblocks, j.root_share);
        nsteps = std::max(nsteps, all[d].size());
    }
    const std::vector<Chunk> &chunks = all[r];
    uint64_t my_rows = 0;
    for (const Chunk &ch : chunks)
        for (uint32_t k = 0; k < ch.count; k++) {
            uint32_t y0, y1;
            block_range(j, ch.first + k * ch.stride, y0, y1);
            my_rows += y1 - y0;
        }
    int rc = FR_OK;
    if (!in_place) {
        rc = ctx.reserve(ctx.rgb, my_rows * row_bytes);
        if (rc != FR_OK) return rc;
    }
    uint8_t *scratch = static_cast<uint8_t *>(ctx.rgb.ptr);
    Rccl &rc_lib = j.set->rccl;
    if (w.timing_event(2 * nsteps, &tc0) != FR_OK || w.timing_event(2 * nsteps + 1, &tc1) != FR_OK) return FR_ERR_HIP;
    auto transfers_begin = [&] { /* first transfer of the job: stamp the copy stream where it becomes ready to start */
        if (!tc_started && hipEventRecord(tc0, ctx.copy_stream) == hipSuccess) tc_started = true;
    };
    ncclComm_t comm = j.sink == Sink::Rccl ? rc_lib.comms[r] : nullptr;
    size_t local_off = 0, nkernels = 0;
    bool plain_copies = false;
    std::vector<std::pair<size_t, size_t>> deferred; /* (dst offset, scratch offset) of blocks to copy unpinned */
    std::vector<size_t> deferred_len;

#define HIP_BRK(expr)                          \
    {                                          \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) {                \
            rc = ::fr::fail_hip(e_, #expr);    \
            break;                             \
        }                                      \
    }
    /* Every exit from this loop falls through to the drain below (kernels and DMAs of earlier chunks may be in
     * flight on three streams into memory the caller frees when we return): no `return` in here. */
    for (size_t c = 0; c < nsteps && rc == FR_OK; c++) {
        if (j.abort->load(std::memory_order_acquire)) {
            rc = fail(FR_ERR_HIP, kEchoError);
            break;
        }
        if (g_inject_device.load() == (int)r && g_inject_chunk.load() == (int)c) {
            g_inject_device.store(-1);
            rc = fail(FR_ERR_HIP, "injected failure (fr_debug_inject_multi_failure)");
            break;
        }
        const Chunk mine = c < chunks.size() ? chunks[c] : Chunk{0, 1, 0};
        const uint32_t first = mine.first, count = mine.count, stride = mine.stride;
        hipStream_t stream = (c & 1) ? ctx.stream2 : ctx.stream;
        hipEvent_t done = nullptr;
        size_t chunk_off = local_off;
        if (count) {
            hipEvent_t t0, t1;
            rc = w.timing_event(2 * nkernels, &t0);
            if (rc == FR_OK) rc = w.timing_event(2 * nkernels + 1, &t1);
            if (rc != FR_OK) break;
            HIP_BRK(hipEventRecord(t0, stream));
            uint64_t rows = 0;
            if (in_place)
                rc = render_block_cyclic(ctx, j.cfg, j.precision, j.opts, j.block_rows, first, stride, count, 1, j.dst, j.dst_len,
                                         stream, &rows);
            else
                rc = render_block_cyclic(ctx, j.cfg, j.precision, j.opts, j.block_rows, first, stride, count, 0,
                                         scratch + local_off, (size_t)(my_rows * row_bytes - local_off), stream, &rows);
            if (rc != FR_OK) break;
            HIP_BRK(hipEventRecord(t1, stream));
            nkernels++;
            local_off += in_place ? 0 : rows * row_bytes;
            rc = ctx.event(c, &done);
            if (rc != FR_OK) break;
            HIP_BRK(hipEventRecord(done, stream));
        }
        /* ---- move chunk c while chunk c+1 renders */
        if (j.sink == Sink::Rccl) {
            if (n == 1) continue;
            if (r == 0) {
                /* the root posts the receives of every peer's chunk c, per peer in the peer's sending order,
                 * straight into the blocks' final places (they do not depend on the root's own kernels) */
                transfers_begin();
                rc = rc_lib.check(rc_lib.GroupStart(), "ncclGroupStart");
                for (uint32_t src = 1; src < n && rc == FR_OK; src++) {
                    if (c >= all[src].size()) continue;
                    const Chunk &pc = all[src][c];
                    for (uint32_t k = 0; k < pc.count && rc == FR_OK; k++) {
                        uint32_t y0, y1;
                        block_range(j, pc.first + k * pc.stride, y0, y1);
                        rc = rc_lib.check(rc_lib.Recv(j.dst + row_bytes * y0, row_bytes * (y1 - y0), kNcclUint8, (int)src, comm,
                                                      ctx.copy_stream), "ncclRecv");
                    }
                }
                int rc2 = rc_lib.check(rc_lib.GroupEnd(), "ncclGroupEnd");
                if (rc == FR_OK) rc = rc2;
            } else if (count) {
                HIP_BRK(hipStreamWaitEvent(ctx.copy_stream, done, 0));
                transfers_begin();
                rc = rc_lib.check(rc_lib.GroupStart(), "ncclGroupStart");
                size_t off = chunk_off;
                for (uint32_t k = 0; k < count && rc == FR_OK; k++) {
                    uint32_t y0, y1;
                    block_range(j, first + k * stride, y0, y1);
                    const size_t bytes = row_bytes * (y1 - y0);
                    rc = rc_lib.check(rc_lib.Send(scratch + off, bytes, kNcclUint8, 0, comm, ctx.copy_stream), "ncclSend");
                    off += bytes;
                    bytes_moved += bytes;
                }
                int rc2 = rc_lib.check(rc_lib.GroupEnd(), "ncclGroupEnd");
                if (rc == FR_OK) rc = rc2;
            }
            continue;
        }
        if (!count || in_place) continue;
        HIP_BRK(hipStreamWaitEvent(ctx.copy_stream, done, 0));
        transfers_begin();
        size_t off = chunk_off;
        for (uint32_t k = 0; k < count && rc == FR_OK; k++) {
            uint32_t y0, y1;
            block_range(j, first + k * stride, y0, y1);
            const size_t bytes = row_bytes * (y1 - y0), dst_off = row_bytes * y0;
            bytes_moved += bytes;
            if (j.sink == Sink::PeerCopy) {
                if (ctx.hip_device == j.root_device) {
                    HIP_BRK(hipMemcpyAsync(j.dst + dst_off, scratch + off, bytes, hipMemcpyDeviceToDevice, ctx.copy_stream));
                } else {
                    HIP_BRK(hipMemcpyPeerAsync(j.dst + dst_off, j.root_device, scratch + off, ctx.hip_device, bytes,
                                               ctx.copy_stream));
                }
            } else {
                /* the pinner makes the buffer DMA-able front to back while we render: wait for this block's bytes */
                if (!plain_copies && !j.pins->wait_for(dst_off, dst_off + bytes, *j.abort)) plain_copies = true;
                if (j.abort->load(std::memory_order_acquire)) {
                    rc = fail(FR_ERR_HIP, kEchoError);
                    break;
                }
                if (plain_copies) {
                    deferred.emplace_back(dst_off, off);
                    deferred_len.push_back(bytes);
                } else {
                    /* one DMA must not span two pins: split the block where the pinner's chunks end */
                    size_t done_b = 0;
                    while (done_b < bytes && rc == FR_OK) {
                        size_t ca = 0, cb = 0; /* the chunk [ca, cb) that holds byte dst_off + done_b */
                        while ((cb = ChunkPinner::chunk_end(j.dst, need_total, ca)) <= dst_off + done_b) ca = cb;
                        const size_t part = std::min(bytes - done_b, cb - (dst_off + done_b));
                        HIP_BRK(hipMemcpyAsync(j.dst + dst_off + done_b, scratch + off + done_b, part, hipMemcpyDeviceToHost,
                                               ctx.copy_stream));
                        done_b += part;
                    }
                }
            }
            off += bytes;
        }
    }
#undef HIP_BRK
    if (rc != FR_OK) {
        /* tell the others, and — RCCL — release the peers that wait for transfers this device will never post: this
         * rank's own communicator only, from this thread, its owner (see Rccl::abort_own) */
        j.abort->store(true, std::memory_order_release);
        if (j.pins) j.pins->wake();
        if (j.sink == Sink::Rccl) rc_lib.abort_own(r);
    } else if (j.sink == Sink::Rccl && n > 1) {
        /* A healthy rank waits for its transfers by POLLING, so that it can see the abort flag: a peer that failed will
         * never post the matching operation, and a device-side wait on it would block hipStreamSynchronize for ever.
         * Seeing the flag, it aborts its own communicator — which ends its operations in flight — and reports an echo. */
        for (;;) {
            const hipError_t q = hipStreamQuery(ctx.copy_stream);
            if (q != hipErrorNotReady) {
                if (q != hipSuccess) (void)hipGetLastError();
                break;
            }
            if (j.abort->load(std::memory_order_acquire)) {
                rc_lib.abort_own(r);
                rc = fail(FR_ERR_HIP, kEchoError);
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    const bool tc_ended = tc_started && hipEventRecord(tc1, ctx.copy_stream) == hipSuccess;
    /* drain, error or not */
    hipError_t e1 = hipStreamSynchronize(ctx.stream);
    hipError_t e2 = hipStreamSynchronize(ctx.stream2);
    hipError_t e3 = hipStreamSynchronize(ctx.copy_stream);
    if (rc != FR_OK) return rc;
    if (e1 != hipSuccess) return fail_hip(e1, "hipStreamSynchronize(stream)");
    if (e2 != hipSuccess) return fail_hip(e2, "hipStreamSynchronize(stream2)");
    if (e3 != hipSuccess) return fail_hip(e3, "hipStreamSynchronize(copy_stream)");
    for (size_t k = 0; k < deferred.size(); k++) /* host memory that could not be pinned */
        HIP_TRY(hipMemcpy(j.dst + deferred[k].first, scratch + deferred[k].second, deferred_len[k], hipMemcpyDeviceToHost));
    float total = 0.0f;
    for (size_t k = 0; k < nkernels; k++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, w.tev[2 * k], w.tev[2 * k + 1]));
        total += ms;
    }
    stats->kernels[r] = (uint32_t)nkernels;
    stats->kernel_ms[r] = total;
    stats->rows[r] = my_rows;
    float span = 0.0f;
    if (tc_ended && hipEventElapsedTime(&span, tc0, tc1) != hipSuccess) {
        (void)hipGetLastError();
        span = 0.0f;
    }
    stats->transfer_span_ms[r] = span;
    stats->bytes_moved[r] = bytes_moved;
    stats->job_ms[r] = now_ms() - t_job;
    return FR_OK;
}

/* Every worker drops its own communicator on its own thread (abort after a failed gather, destroy at shutdown). */
void drop_comms(DeviceSet &set, bool abort) {
    Rccl *lib = &set.rccl;
    if (lib->comms.empty()) return;
    for (auto &w : set.workers) {
        Worker *wp = w.get();
        wp->post([wp, lib, abort] {
            (void)hipSetDevice(wp->ctx.hip_device);
            if (abort) lib->abort_own((uint32_t)wp->index);
            else lib->destroy_own((uint32_t)wp->index);
            return (int)FR_OK;
        });
    }
    for (auto &w : set.workers) (void)w->wait();
    lib->comms.clear();
    lib->aborted.store(false, std::memory_order_release);
}

int ensure_rccl(DeviceSet &set) {
    if (!set.distinct) return fail(FR_ERR_INVALID_ARGUMENT, "FR_GATHER_RCCL needs distinct devices (a communicator cannot hold one GPU twice)");
    int rc = set.rccl.load();
    if (rc != FR_OK) return rc;
    if (set.rccl.comms.empty()) {
        set.rccl.comms.assign(set.devices.size(), nullptr);
        rc = set.rccl.check(set.rccl.CommInitAll(set.rccl.comms.data(), (int)set.devices.size(), set.devices.data()),
                            "ncclCommInitAll");
        if (rc != FR_OK) set.rccl.comms.clear();
    }
    return rc;
}

int run_multi(const fr_config *cfg, int precision, uint32_t block_rows, Sink sink, uint8_t *dst, size_t dst_len) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    int rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    if (block_rows == 0) block_rows = kDefaultBlockRows;
    if (block_rows % 8 != 0) return fail(FR_ERR_INVALID_ARGUMENT, "block_rows must be a multiple of 8");
    const size_t need = (size_t)3 * cfg->width * (size_t)cfg->height;
    if (need == 0) return FR_OK;
    if (!dst) return fail(FR_ERR_INVALID_ARGUMENT, "output buffer is NULL");
    if (dst_len < need) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 3*width*height");
    LifeShared ls;
    DeviceSet *set = g_set;
    if (!set) return fail(FR_ERR_INVALID_ARGUMENT, "no device set: call fr_init_devices first");
    std::lock_guard<std::mutex> lk(set->call_mu);
    const double t_start = now_ms();
    const uint32_t n = (uint32_t)set->workers.size();
    if ((uint64_t)block_rows * n > 0xFFFFFFFFull) return fail(FR_ERR_INVALID_ARGUMENT, "block_rows * devices overflows u32");
    if (sink == Sink::Rccl) {
        rc = ensure_rccl(*set);
        if (rc != FR_OK) return rc;
    }
    std::atomic<bool> abort{false};
    /* Host sink: while the devices render, a pool makes the caller's buffer DMA-able front to back */
    std::unique_ptr<PinProgress> pins;
    if (sink == Sink::Host) pins.reset(new PinProgress(dst, need));
    Job job{cfg, precision, default_opts(), block_rows, (uint32_t)(((uint64_t)cfg->height + block_rows - 1) / block_rows), n, sink,
            dst, dst_len, set->devices[0], pins.get(), set, &abort, g_root_share.load()};
    /* which kernel suits the view: decided once for the whole image (on the first device), not by every chunk launch */
    if (hipSetDevice(set->devices[0]) == hipSuccess) decide_kernel(set->workers[0]->ctx, cfg, precision, 0, cfg->height, job.opts, nullptr, false);
    (void)hipGetLastError();
    fr_multi_stats stats;
    memset(&stats, 0, sizeof stats);
    stats.n_devices = n;
    for (auto &w : set->workers) {
        Worker *wp = w.get();
        wp->post([wp, &job, &stats] { return device_job(*wp, job, &stats); });
    }
    /* report the failure that started it, not the "another device of the set failed" of those that stopped for it */
    std::string first_err;
    bool have_cause = false;
    for (auto &w : set->workers) {
        const int r = w->wait();
        if (r == FR_OK) continue;
        const bool echo = w->err.rfind(kEchoError, 0) == 0;
        if (rc == FR_OK || (!have_cause && !echo)) {
            rc = r;
            first_err = "device " + std::to_string(w->index) + ": " + w->err;
            have_cause = !echo;
        }
    }
    if (pins) pins->finish(); /* every device has drained its streams */
    if (sink == Sink::Rccl && (abort.load() || set->rccl.aborted.load())) drop_comms(*set, true); /* the next RCCL render makes new ones */
    stats.wall_ms = now_ms() - t_start;
    tl_stats = stats;
    if (rc != FR_OK) return fail(rc, first_err);
    return FR_OK;
}

}  // namespace

/* exclusive lifetime lock held by the caller */
void multi_shutdown_locked() {
    DeviceSet *set = g_set;
    g_set = nullptr;
    if (!set) return;
    drop_comms(*set, false);
    for (auto &w : set->workers) {
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->stop = true;
            w->cv.notify_all();
        }
        if (w->th.joinable()) w->th.join();
    }
    delete set;
}

}  // namespace fr

using namespace fr;

extern "C" {

int fr_init_devices(const int *devices, int n) {
    if (!devices || n <= 0 || n > FR_MAX_DEVICES) return fail(FR_ERR_INVALID_ARGUMENT, "need 1..16 device indices");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(FR_ERR_NO_DEVICE, "no HIP device available (libfractal_hip has no CPU fallback)");
    }
    for (int k = 0; k < n; k++)
        if (devices[k] < 0 || devices[k] >= count) return fail(FR_ERR_NO_DEVICE, "device index out of range");
    LifeExclusive lx;
    multi_shutdown_locked();
    std::unique_ptr<DeviceSet> set(new DeviceSet);
    set->devices.assign(devices, devices + n);
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            if (devices[a] == devices[b]) set->distinct = false;
    int rc = FR_OK;
    std::string err;
    for (int k = 0; k < n; k++) {
        std::unique_ptr<Worker> w(new Worker);
        w->index = k;
        Worker *wp = w.get();
        const int dev = devices[k];
        wp->th = std::thread([wp, dev] { wp->loop(dev); });
        set->workers.push_back(std::move(w));
    }
    for (auto &w : set->workers) {
        const int r = w->wait(); /* context creation */
        if (r != FR_OK && rc == FR_OK) {
            rc = r;
            err = w->err;
        }
    }
    /* peer access from every device to the root's memory (the peer-copy gather writes there) */
    if (rc == FR_OK) {
        const int root = devices[0];
        for (auto &w : set->workers) {
            Worker *wp = w.get();
            wp->post([wp, root] {
                if (wp->ctx.hip_device == root) return (int)FR_OK;
                int can = 0;
                HIP_TRY(hipDeviceCanAccessPeer(&can, wp->ctx.hip_device, root));
                if (!can) return (int)FR_OK; /* hipMemcpyPeerAsync then stages through the host */
                hipError_t pe = hipDeviceEnablePeerAccess(root, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail_hip(pe, "hipDeviceEnablePeerAccess");
                (void)hipGetLastError();
                return (int)FR_OK;
            });
        }
        for (auto &w : set->workers) {
            const int r = w->wait();
            if (r != FR_OK && rc == FR_OK) {
                rc = r;
                err = w->err;
            }
        }
    }
    g_set = set.release();
    if (rc != FR_OK) {
        multi_shutdown_locked();
        return fail(rc, err);
    }
    return FR_OK;
}

int fr_set_multi_root_share(int q) {
    if (q != 0 && q != 1 && q != 2 && q != 4) return fail(FR_ERR_INVALID_ARGUMENT, "root share must be 1 (all of its blocks), 2, 4 (a half, a quarter) or 0 (none)");
    g_root_share.store(q);
    return FR_OK;
}

int fr_multi_device_count(int *count) {
    if (!count) return fail(FR_ERR_INVALID_ARGUMENT, "count is NULL");
    LifeShared ls;
    *count = g_set ? (int)g_set->workers.size() : 0;
    return FR_OK;
}

int fr_render_rgb8_multi(const fr_config *cfg, int precision, uint32_t block_rows, uint8_t *out, size_t out_len) {
    return run_multi(cfg, precision, block_rows, Sink::Host, out, out_len);
}

int fr_render_rgb8_multi_device(const fr_config *cfg, int precision, uint32_t block_rows, int gather, void *d_out,
                                size_t out_len) {
    if (gather != FR_GATHER_PEER_COPY && gather != FR_GATHER_RCCL)
        return fail(FR_ERR_INVALID_ARGUMENT, "gather must be FR_GATHER_PEER_COPY or FR_GATHER_RCCL");
    return run_multi(cfg, precision, block_rows, gather == FR_GATHER_RCCL ? Sink::Rccl : Sink::PeerCopy,
                     static_cast<uint8_t *>(d_out), out_len);
}

int fr_multi_last_stats(fr_multi_stats *stats) {
    if (!stats) return fail(FR_ERR_INVALID_ARGUMENT, "stats is NULL");
    *stats = tl_stats;
    return FR_OK;
}

int fr_debug_rccl_probe(void) {
    Rccl probe; /* a private copy of the loader: nothing of the device set is touched */
    const int rc = probe.load();
    if (rc == FR_OK && probe.handle) dlclose(probe.handle);
    return rc;
}

int fr_debug_inject_multi_failure(int device_index, int chunk) {
    g_inject_chunk.store(chunk < 0 ? 0 : chunk);
    g_inject_device.store(device_index);
    return FR_OK;
}

int fr_pin_host_buffer(void *ptr, size_t len) {
    if (!ptr || len == 0) return fail(FR_ERR_INVALID_ARGUMENT, "ptr is NULL or len is 0");
    HIP_TRY(hipHostRegister(ptr, len, hipHostRegisterPortable));
    return FR_OK;
}

int fr_unpin_host_buffer(void *ptr) {
    if (!ptr) return fail(FR_ERR_INVALID_ARGUMENT, "ptr is NULL");
    HIP_TRY(hipHostUnregister(ptr));
    return FR_OK;
}

int fr_debug_rccl_selftest(size_t bytes) {
    if (bytes == 0) return FR_OK;
    LifeShared ls;
    DeviceSet *set = g_set;
    if (!set) return fail(FR_ERR_INVALID_ARGUMENT, "no device set: call fr_init_devices first");
    std::lock_guard<std::mutex> lk(set->call_mu);
    int rc = ensure_rccl(*set);
    if (rc != FR_OK) return rc;
    Worker *w = set->workers[0].get();
    Rccl *lib = &set->rccl;
    w->post([w, lib, bytes] {
        Ctx &ctx = w->ctx;
        HIP_TRY(hipSetDevice(ctx.hip_device));
        int rc = ctx.reserve(ctx.misc, 2 * bytes);
        if (rc != FR_OK) return rc;
        uint8_t *src = static_cast<uint8_t *>(ctx.misc.ptr), *dst = src + bytes;
        HIP_TRY(hipMemsetAsync(src, 0xA7, bytes, ctx.copy_stream));
        HIP_TRY(hipMemsetAsync(dst, 0, bytes, ctx.copy_stream));
        rc = lib->check(lib->GroupStart(), "ncclGroupStart");
        if (rc == FR_OK) rc = lib->check(lib->Recv(dst, bytes, kNcclUint8, 0, lib->comms[0], ctx.copy_stream), "ncclRecv");
        if (rc == FR_OK) rc = lib->check(lib->Send(src, bytes, kNcclUint8, 0, lib->comms[0], ctx.copy_stream), "ncclSend");
        int rc2 = lib->check(lib->GroupEnd(), "ncclGroupEnd");
        if (rc == FR_OK) rc = rc2;
        if (rc != FR_OK) return rc;
        std::vector<uint8_t> host(bytes);
        HIP_TRY(hipMemcpyAsync(host.data(), dst, bytes, hipMemcpyDeviceToHost, ctx.copy_stream));
        HIP_TRY(hipStreamSynchronize(ctx.copy_stream));
        for (uint8_t v : host)
            if (v != 0xA7) return fail(FR_ERR_HIP, "RCCL self send/recv delivered wrong bytes");
        return (int)FR_OK;
    });
    rc = w->wait();
    if (rc != FR_OK) return fail(rc, w->err);
    return FR_OK;
}

} /* extern "C" */
