/*
 * fr_api.hip — the C ABI of include/fractal_hip.h over the gfx950 kernels.
 *
 * Host-side responsibilities of the path: argument validation, mapping calc::Config
 * (calc/src/lib.rs:21-37) to kernel arguments, device scratch for the host-buffer entry points,
 * and the D2H copy into the caller's Vec<RGB>-shaped buffer (src/lib.rs:253-270).
 *
 * There is deliberately no CPU fallback: every compute entry point fails with FR_ERR_NO_DEVICE /
 * FR_ERR_HIP when the device path is unavailable.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "fr_ctx.h"

static_assert(sizeof(fr_config) == 104 && offsetof(fr_config, limit) == 16 && offsetof(fr_config, inside) == 72 &&
                  offsetof(fr_config, primary_color) == 74 && offsetof(fr_config, color_weight) == 80 &&
                  offsetof(fr_config, julia_set) == 88,
              "fr_config must stay the #[repr(C)] image of calc::Config");
static_assert(sizeof(fr_render_opts) == 32, "fr_render_opts is part of the ABI");

namespace fr {

namespace {

thread_local std::string tl_error;
thread_local Profiling tl_prof;
std::atomic<bool> g_process_exiting{false};

std::shared_mutex g_life;
std::mutex g_primary_mu; /* creation / switch of the primary context */
Ctx g_primary;
bool g_primary_inited = false;

/* process-wide defaults of the per-call selectors (fr_set_*) */
std::atomic<int> g_tile{0};
std::atomic<int> g_palette_enabled{1};
std::atomic<int> g_cycle_shortcut{0};
std::atomic<int> g_refill_minrun{-1}, g_refill_quit16{-1}; /* -1: each kernel's own default */
std::atomic<int> g_loop_mode{-1}; /* -1 auto, 0 unscaled, 2 / 4 scaled with that check interval */
std::atomic<int> g_colour_filter{1};
std::atomic<unsigned long long *> g_queue_trace{nullptr}; /* tuning aid, see fr_debug_set_queue_trace */
std::atomic<uint32_t> g_two_pass_list_entries{0};          /* test aid, see fr_debug_set_two_pass_capacity */

bool valid_tile(int tile) {
    switch (tile) {
    case 0: case 1: case 2: case 4: case 8: case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: case 6401: case 3202: case 1604: case 808:
        return true;
    default:
        return false;
    }
}

}  // namespace

int fail(int code, const char *what) {
    tl_error = what;
    return code;
}
int fail(int code, const std::string &what) {
    tl_error = what;
    return code;
}

int fail_hip(hipError_t e, const char *what) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    tl_error = buf;
    (void)hipGetLastError(); /* clear the sticky error */
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? FR_ERR_NO_DEVICE : FR_ERR_HIP;
}

const std::string &last_error() { return tl_error; }

LifeShared::LifeShared() { g_life.lock_shared(); }
LifeShared::~LifeShared() { g_life.unlock_shared(); }
LifeExclusive::LifeExclusive() { g_life.lock(); }
LifeExclusive::~LifeExclusive() { g_life.unlock(); }

Profiling &profiling() { return tl_prof; }
Profiling::~Profiling() {
    /* a thread that profiled gives its two events back — unless the process is already tearing the
     * HIP runtime down (static destruction order is not ours to rely on) */
    if (e0 && !g_process_exiting.load()) {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    e0 = e1 = nullptr;
}

Opts default_opts() {
    Opts o;
    o.tile = g_tile.load();
    o.loop_mode = g_loop_mode.load();
    o.palette = g_palette_enabled.load();
    o.cycle_shortcut = g_cycle_shortcut.load();
    o.refill_minrun = g_refill_minrun.load();
    o.refill_quit16 = g_refill_quit16.load();
    o.colour_filter = g_colour_filter.load();
    return o;
}

int resolve_opts(const fr_render_opts *in, Opts &o) {
    o = default_opts();
    if (!in) return FR_OK;
    if (in->size < sizeof(fr_render_opts)) return fail(FR_ERR_INVALID_ARGUMENT, "fr_render_opts.size is too small (use fr_render_opts_init)");
    if (!valid_tile(in->tile)) return fail(FR_ERR_INVALID_ARGUMENT, "opts.tile must be 0, 1, 2, 4, 8, 9, 10 ... 16, 6401, 3202, 1604 or 808");
    if (in->loop_mode != -1 && in->loop_mode != 0 && in->loop_mode != 2 && in->loop_mode != 4 && in->loop_mode != 5)
        return fail(FR_ERR_INVALID_ARGUMENT, "opts.loop_mode must be -1 (auto), 0, 2, 4 or 5");
    if (in->refill_minrun < -1 || in->refill_quit16 < -1 || in->refill_quit16 == 0 || in->refill_quit16 > 16)
        return fail(FR_ERR_INVALID_ARGUMENT, "opts: refill_minrun >= 0, 1 <= refill_quit16 <= 16 (or -1: the kernel's default)");
    o.tile = in->tile;
    o.loop_mode = in->loop_mode;
    o.palette = in->palette != 0;
    o.cycle_shortcut = in->cycle_shortcut != 0;
    o.refill_minrun = in->refill_minrun;
    o.refill_quit16 = in->refill_quit16;
    o.colour_filter = in->colour_filter == 2 ? 2 : in->colour_filter != 0;
    return FR_OK;
}

/* ---- Ctx --------------------------------------------------------------------------------------- */

int Ctx::create(int device) {
    HIP_TRY(hipSetDevice(device));
    hip_device = device; /* from here on destroy() releases whatever the steps below managed to create */
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
    /* D2H and peer copies of registered memory run as shader (blit) kernels on this runtime: give their
     * stream the highest priority so that they get the few CUs they need while render kernels fill the chip
     * (the PCIe link, not the copy kernel, is the limit) */
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) {
        (void)hipGetLastError();
        prio_greatest = 0;
    }
    HIP_TRY(hipStreamCreateWithPriority(&copy_stream, hipStreamNonBlocking, prio_greatest));
    HIP_TRY(hipStreamCreateWithPriority(&aux_stream, hipStreamNonBlocking, prio_greatest));
    HIP_TRY(hipStreamCreateWithPriority(&aux2_stream, hipStreamNonBlocking, prio_least));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sample_counters), 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(sample_counters, 0, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sample_result), (kViewChoices + 1) * 8 * sizeof(unsigned long long), hipHostMallocMapped));
    memset(sample_result, 0, (kViewChoices + 1) * 8 * sizeof(unsigned long long));
    for (ViewChoice &v : view_choices) HIP_TRY(hipEventCreateWithFlags(&v.after, hipEventDisableTiming));
    /* every palette / claim-counter slot and every ring event now, so that no render allocates them on its way */
    constexpr size_t kSlotWords = FR_MAX_PALETTE_ENTRIES + FR_SURV_QUEUES * FR_SURV_COUNT_STRIDE;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&palette_block), sizeof(uint32_t) * kSlotWords * kPaletteSlots));
    for (int k = 0; k < kPaletteSlots; k++) {
        palette_slots[k].dev = palette_block + (size_t)k * kSlotWords;
        HIP_TRY(hipEventCreateWithFlags(&palette_slots[k].done, hipEventDisableTiming));
    }
    for (SurvSlot &ss : surv_slots) HIP_TRY(hipEventCreateWithFlags(&ss.done, hipEventDisableTiming));
    /* ... and, BEST EFFORT, the survivor-list ring for every frame up to 3840 x 2160 (one entry per eight pixels, at most 44
     * bytes each: 46 MB a slot), so that a GUI's first large Julia frame does not wait ~15 ms for the allocation
     * (acquire_surv re-makes the ring only for a launch that needs more), and the host-buffer entry points' device image
     * buffer with room for a 3840 x 2160 RGBA frame (a 4K first frame: 2.5-3.4 ms against 0.7-0.9 steady).  On a device too
     * full for these ~210 MB the context is still made: both grow lazily (acquire_surv, reserve) when a render needs them. */
    {
        constexpr size_t kGuiSlot = (size_t)48 << 20;
        if (hipMalloc(&surv_block, kGuiSlot * kSurvSlots) == hipSuccess) {
            surv_slot_cap = kGuiSlot;
            for (int k = 0; k < kSurvSlots; k++) surv_slots[k].dev = static_cast<char *>(surv_block) + (size_t)k * kGuiSlot;
        } else {
            (void)hipGetLastError();
            surv_block = nullptr;
        }
        if (hipMalloc(&rgb.ptr, (size_t)64 << 20) == hipSuccess) {
            rgb.cap = (size_t)64 << 20;
        } else {
            (void)hipGetLastError();
            rgb = Scratch();
        }
        (void)reserve_stage((size_t)40 << 20); /* the pinned staging buffer of GUI-sized host renders; lazily otherwise */
        /* ... and one copy-engine transfer of a 3840 x 2160 frame's band size out of HBM into that buffer on each of the two
         * band streams, here instead of inside the first such frame: the process's FIRST hipMemcpyAsync of this kind took
         * 6-8 ms in every bench.py run (gui_latency: the first 3840 x 2160 frame, 6.5 / 6.5 / 8.2 ms against a 0.85 median;
         * the second 4K shape of the same process never) — the runtime sets its copy path up on first use.  Best effort. */
        if (rgb.ptr && stage && stage_cap >= ((size_t)12 << 20) && rgb.cap >= ((size_t)12 << 20)) {
            const size_t band = (size_t)6 << 20;
            hipError_t e = hipMemcpyAsync(stage, rgb.ptr, band, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(static_cast<char *>(stage) + band, static_cast<char *>(rgb.ptr) + band, band, hipMemcpyDeviceToHost, stream2);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream2);
            if (e != hipSuccess) (void)hipGetLastError();
        }
        /* ... and one KERNEL on each of the two band streams (the copy kernel, 4 KB, flags 6 and 7 — frames use 0 to 3): a
         * stream gets its hardware queue at its first kernel, and where the process has meanwhile made large device
         * allocations of its own that first kernel came 25-30 ms late — the first two-band host frame of bench.py runs that
         * had cloned the 805 MB image (5 of 12).  Made here, while the process is small. */
        if (rgb.ptr && stage_dev && stage_flags_dev && stage_counters) {
            hipError_t e = fr_launch_copy_out(rgb.ptr, stage_dev, 4096, stage_counters + 6, stage_flags_dev + 6, 1ull, stream);
            if (e == hipSuccess)
                e = fr_launch_copy_out(static_cast<char *>(rgb.ptr) + 4096, static_cast<char *>(stage_dev) + 4096, 4096, stage_counters + 7,
                                       stage_flags_dev + 7, 1ull, stream2);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream2);
            if (e != hipSuccess) (void)hipGetLastError();
        }
    }
    return FR_OK;
}

int Ctx::reserve_stage(size_t bytes) {
    if (bytes <= stage_cap) return FR_OK;
    if (stage) (void)hipHostFree(stage);
    stage = stage_dev = nullptr, stage_cap = 0;
    hipError_t e = hipHostMalloc(&stage, bytes, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&stage_dev, stage, 0);
    if (e == hipSuccess && !stage_flags) {
        e = hipHostMalloc(reinterpret_cast<void **>(&stage_flags), 8 * sizeof(unsigned long long), hipHostMallocMapped);
        if (e == hipSuccess) {
            memset(stage_flags, 0, 8 * sizeof(unsigned long long));
            e = hipHostGetDevicePointer(reinterpret_cast<void **>(&stage_flags_dev), stage_flags, 0);
        }
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&stage_counters), 8 * sizeof(unsigned int));
        if (e == hipSuccess) e = hipMemset(stage_counters, 0, 8 * sizeof(unsigned int));
    }
    if (e != hipSuccess) {
        if (stage) (void)hipHostFree(stage);
        stage = stage_dev = nullptr;
        return fail_hip(e, "staging buffer (hipHostMalloc)");
    }
    stage_cap = bytes;
    memset(stage, 0, bytes); /* every page exists before the first frame */
    return FR_OK;
}

void Ctx::destroy() {
    if (hip_device < 0) return;
    (void)hipSetDevice(hip_device);
    /* hipFree waits for the device, so nothing below can still be in use by a kernel in flight */
    for (Scratch *s : {&rgb, &z, &iters, &misc}) {
        if (s->ptr) (void)hipFree(s->ptr);
        *s = Scratch();
    }
    {
        std::lock_guard<std::mutex> pl(palette_mu);
        for (PaletteSlot &ps : palette_slots) {
            if (ps.done) (void)hipEventDestroy(ps.done);
            ps = PaletteSlot();
        }
        if (palette_block) (void)hipFree(palette_block);
        palette_block = nullptr;
        for (SurvSlot &ss : surv_slots) {
            if (ss.done) (void)hipEventDestroy(ss.done);
            ss = SurvSlot();
        }
        if (surv_block) (void)hipFree(surv_block);
        surv_block = nullptr, surv_slot_cap = 0;
    }
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    events.clear();
    if (copy_pool) destroy_copy_pool(copy_pool);
    copy_pool = nullptr;
    if (stage) (void)hipHostFree(stage);
    stage = stage_dev = nullptr, stage_cap = 0;
    if (stage_flags) (void)hipHostFree(stage_flags);
    if (stage_counters) (void)hipFree(stage_counters);
    stage_flags = stage_flags_dev = nullptr, stage_counters = nullptr;
    for (hipStream_t *st : {&aux_stream, &aux2_stream})
        if (*st) (void)hipStreamSynchronize(*st); /* a sample in flight writes to sample_result */
    if (sample_counters) (void)hipFree(sample_counters);
    if (sample_result) (void)hipHostFree(sample_result);
    sample_counters = sample_result = nullptr;
    for (ViewChoice &v : view_choices) {
        if (v.after) (void)hipEventDestroy(v.after);
        v = ViewChoice();
    }
    for (hipStream_t *st : {&stream, &stream2, &copy_stream, &aux_stream, &aux2_stream}) {
        if (*st) {
            (void)hipStreamSynchronize(*st);
            (void)hipStreamDestroy(*st);
        }
        *st = nullptr;
    }
    hip_device = -1;
}

int Ctx::reserve(Scratch &s, size_t bytes) {
    if (bytes <= s.cap) return FR_OK;
    if (s.ptr) {
        HIP_TRY(hipFree(s.ptr));
        s = Scratch();
    }
    HIP_TRY(hipMalloc(&s.ptr, bytes));
    s.cap = bytes;
    return FR_OK;
}

int Ctx::event(size_t k, hipEvent_t *out) {
    while (events.size() <= k) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        events.push_back(e);
    }
    *out = events[k];
    return FR_OK;
}

/* Both rings: a slot is `busy` from acquire to release (a thread is enqueueing work that uses it) and `pending`
 * from release until the event recorded behind that work has been waited for.  A caller that finds every slot
 * busy waits for a release; the waits on events and the allocations happen outside the lock. */
int Ctx::acquire_palette(PaletteSlot **out) {
    PaletteSlot *s = nullptr;
    bool pending;
    {
        std::unique_lock<std::mutex> lk(palette_mu);
        while (!s) {
            for (int tries = 0; tries < kPaletteSlots && !s; tries++) {
                PaletteSlot &c = palette_slots[palette_next++ % kPaletteSlots];
                if (!c.busy) s = &c;
            }
            if (!s) slot_cv.wait(lk);
        }
        s->busy = true;
        pending = s->pending;
        s->pending = false;
    }
    hipError_t e = hipSuccess;
    if (pending) e = hipEventSynchronize(s->done); /* blocks only if 16 renders are in flight */
    if (e != hipSuccess) {
        {
            std::lock_guard<std::mutex> lk(palette_mu);
            s->busy = false;
        }
        slot_cv.notify_all(); /* (waiters of both rings share the condition variable) */
        return fail_hip(e, "palette slot");
    }
    *out = s;
    return FR_OK;
}

/* the slot may be reused once everything enqueued so far on `stream` has run */
void Ctx::release_palette(PaletteSlot *slot, hipStream_t st) {
    if (!slot) return;
    const bool recorded = slot->done && hipEventRecord(slot->done, st) == hipSuccess;
    if (!recorded) (void)hipStreamSynchronize(st); /* no event to wait on later: wait now */
    {
        std::lock_guard<std::mutex> lk(palette_mu);
        slot->pending = recorded;
        slot->busy = false;
    }
    slot_cv.notify_all(); /* (waiters of both rings share the condition variable) */
}

int Ctx::acquire_surv(size_t bytes, SurvSlot **out) {
    SurvSlot *s = nullptr;
    bool pending = false;
    hipError_t e = hipSuccess;
    {
        std::unique_lock<std::mutex> lk(palette_mu);
        for (;;) {
            if (surv_regrowing) {
                slot_cv.wait(lk);
                continue;
            }
            if (surv_slot_cap >= bytes) {
                for (int tries = 0; tries < kSurvSlots && !s; tries++) {
                    SurvSlot &c = surv_slots[surv_next++ % kSurvSlots];
                    if (!c.busy) s = &c;
                }
                if (s) break;
                slot_cv.wait(lk); /* more renders being enqueued at once than there are buffers */
                continue;
            }
            /* The ring is (re)made in one allocation — the one blocking allocation of the render path (the first launch
             * that needs lists larger than any before).  One thread does it, with the lock DROPPED over the waits, the
             * free and the malloc (hipFree is a device-wide synchronisation, the malloc may be gigabytes): palette slots
             * keep circulating meanwhile; other callers that need lists sleep until the ring is republished. */
            bool someone_busy = false;
            for (const SurvSlot &c : surv_slots) someone_busy = someone_busy || c.busy;
            if (someone_busy) {
                slot_cv.wait(lk);
                continue;
            }
            surv_regrowing = true;
            void *old = surv_block;
            bool was_pending[kSurvSlots];
            for (int k = 0; k < kSurvSlots; k++) {
                was_pending[k] = surv_slots[k].pending;
                surv_slots[k].pending = false;
                surv_slots[k].dev = nullptr;
            }
            surv_block = nullptr, surv_slot_cap = 0;
            lk.unlock();
            for (int k = 0; k < kSurvSlots; k++) /* the renders that used the old buffers */
                if (was_pending[k]) {
                    const hipError_t se = hipEventSynchronize(surv_slots[k].done);
                    if (e == hipSuccess) e = se;
                }
            if (old) { /* freed whatever the waits said (hipFree itself waits for the device): nothing leaks on an error path */
                const hipError_t fe = hipFree(old);
                if (e == hipSuccess) e = fe;
            }
            const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
            void *blk = nullptr;
            if (e == hipSuccess) e = hipMalloc(&blk, cap * kSurvSlots);
            lk.lock();
            surv_regrowing = false;
            if (e != hipSuccess) { /* the ring is empty (capacity 0): the next launch that needs lists tries again */
                lk.unlock();
                slot_cv.notify_all();
                return fail_hip(e, "survivor-list ring");
            }
            surv_block = blk, surv_slot_cap = cap;
            for (int k = 0; k < kSurvSlots; k++) surv_slots[k].dev = static_cast<char *>(surv_block) + (size_t)k * cap;
            slot_cv.notify_all();
        }
        s->busy = true;
        pending = s->pending;
        s->pending = false;
    }
    if (pending) e = hipEventSynchronize(s->done); /* the buffer's previous render (three are in flight) */
    if (e != hipSuccess) {
        {
            std::lock_guard<std::mutex> lk(palette_mu);
            s->busy = false;
        }
        slot_cv.notify_all();
        return fail_hip(e, "survivor-list buffer");
    }
    *out = s;
    return FR_OK;
}

void Ctx::release_surv(SurvSlot *slot, hipStream_t st) {
    if (!slot) return;
    const bool recorded = slot->done && hipEventRecord(slot->done, st) == hipSuccess;
    if (!recorded) (void)hipStreamSynchronize(st);
    {
        std::lock_guard<std::mutex> lk(palette_mu);
        slot->pending = recorded;
        slot->busy = false;
    }
    slot_cv.notify_all();
}

namespace {

/* caller holds g_primary_mu */
int init_primary_locked(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(FR_ERR_NO_DEVICE, "no HIP device available (libfractal_hip has no CPU fallback)");
    }
    if (device < 0) device = g_primary_inited ? g_primary.hip_device : 0;
    if (device >= n) return fail(FR_ERR_NO_DEVICE, "device index out of range");
    if (g_primary_inited && g_primary.hip_device == device) return FR_OK;
    if (g_primary_inited) {
        g_primary.destroy(); /* switching device: drop state that lives on the old one */
        g_primary_inited = false;
    }
    static std::once_flag once;
    std::call_once(once, [] { atexit([] { g_process_exiting.store(true); }); });
    int rc = g_primary.create(device);
    if (rc != FR_OK) {
        g_primary.destroy();
        return rc;
    }
    g_primary_inited = true;
    return FR_OK;
}

}  // namespace

/* hipSetDevice is per host thread, so every entry point re-asserts it */
int primary(Ctx **out) {
    {
        std::lock_guard<std::mutex> lk(g_primary_mu);
        if (!g_primary_inited) {
            int rc = init_primary_locked(-1);
            if (rc != FR_OK) return rc;
        }
    }
    HIP_TRY(hipSetDevice(g_primary.hip_device));
    *out = &g_primary;
    return FR_OK;
}

/* calc::Config -> kernel arguments; the local grid is filled in by the caller */
void fill_params(const fr_config *cfg, const Opts &o, fr_kparams &p) {
    memset(&p, 0, sizeof p);
    p.algo = cfg->algo;
    p.width = cfg->width;
    p.height = cfg->height;
    p.iterations = cfg->iterations;
    p.limit = cfg->limit;
    p.stable_limit = cfg->stable_limit;
    p.pos_re = cfg->pos.re;
    p.pos_im = cfg->pos.im;
    p.scale_re = cfg->scale.re;
    p.scale_im = cfg->scale.im;
    p.exposure = cfg->exposure;
    p.julia_re = cfg->julia_set.re;
    p.julia_im = cfg->julia_set.im;
    p.inside = cfg->inside ? 1u : 0u;
    p.smooth = cfg->smooth ? 1u : 0u;
    p.prim[0] = cfg->primary_color.r;
    p.prim[1] = cfg->primary_color.g;
    p.prim[2] = cfg->primary_color.b;
    p.sec[0] = cfg->secondary_color.r;
    p.sec[1] = cfg->secondary_color.g;
    p.sec[2] = cfg->secondary_color.b;
    for (int k = 0; k < 3; k++) {
        p.prim_f[k] = (double)p.prim[k];
        p.sec_f[k] = (double)p.sec[k];
    }
    p.iterations_f64 = (double)cfg->iterations;
    const uint32_t n = cfg->iterations;
    p.inv_iterations = (n != 0 && (n & (n - 1)) == 0) ? 1.0 / (double)n : 0.0; /* exact: n = 2^k */
    p.ncols = cfg->width;
    p.nrows = 0;
    p.x_first = 0;
    p.x_stride = 1;
    p.block_rows = 1;
    p.y_first = 0;
    p.y_stride = 1;
    /* episode policies (measured defaults): the patch-refill kernel ends an episode once half of its running
     * lanes have finished and 32 iterations were done; the work-queue kernel, whose refill is much cheaper,
     * once 24 lanes are free and 8 iterations were done */
    p.refill_minrun = o.refill_minrun < 0 ? 32u : (uint32_t)o.refill_minrun;
    p.refill_quit16 = o.refill_quit16 < 0 ? 8u : (uint32_t)o.refill_quit16;
    p.queue_minrun = (o.refill_minrun < 0 || (o.tile >= 11 && o.tile <= 16)) ? 8u : (uint32_t)o.refill_minrun;
    /* tile 11: minrun = the first pass's episode length, quit16 = the lanes (in 16ths of a wave) a tile must
     * keep running to stay in the first pass; the second pass keeps its own defaults */
    p.first_keep = ((o.tile >= 11 && o.tile <= 16) && o.refill_quit16 > 0) ? 4u * (uint32_t)o.refill_quit16 : 0u;
    p.two_pass_cap = ((o.tile >= 11 && o.tile <= 16) && o.refill_minrun > 0) ? (uint32_t)o.refill_minrun : 0u; /* tile 11: minrun = first_cap */
    p.queue_want = (o.refill_quit16 < 0 || (o.tile >= 11 && o.tile <= 16)) ? 24u : (64u * (uint32_t)o.refill_quit16 + 15u) / 16u;
    {
        /* tuning aid: the second pass's own policy under the two-pass render (whose fr_set_refill_policy numbers
         * steer the first pass): FR_DEBUG_QUEUE_WANT / FR_DEBUG_QUEUE_MINRUN */
        static const int dbg_want = getenv("FR_DEBUG_QUEUE_WANT") ? atoi(getenv("FR_DEBUG_QUEUE_WANT")) : 0;
        static const int dbg_minrun = getenv("FR_DEBUG_QUEUE_MINRUN") ? atoi(getenv("FR_DEBUG_QUEUE_MINRUN")) : -1;
        if (dbg_want > 0 && dbg_want <= 64) p.queue_want = (uint32_t)dbg_want;
        if (dbg_minrun >= 0) p.queue_minrun = (uint32_t)dbg_minrun;
    }
    {
        /* measurement aid (WRONG IMAGES; DESIGN.md 7, tools/c4_ablation.sh): what the hand-over's stores and the second pass cost */
        static const int dbg_ablate = getenv("FR_DEBUG_ABLATE") ? atoi(getenv("FR_DEBUG_ABLATE")) : 0;
        p.debug_ablate = (uint32_t)dbg_ablate & 3u;
    }
    /* the flag bit of the loop's return value needs iterations < 2^31; keep a margin */
    p.cycle_shortcut = (o.cycle_shortcut && cfg->iterations < (1u << 30)) ? 1u : 0u;
    /* the colour filter's constants (fr_kernels.hip: colour_outside_filtered) and the conditions under
     * which its error analysis holds: a positive cap, a finite, not absurd exposure */
    p.filt_k = n != 0 ? cfg->exposure / (double)n : 0.0;
    const bool filter_ok = o.colour_filter && cfg->smooth && n != 0 && std::isfinite(cfg->exposure) &&
                           std::fabs(p.filt_k) <= 1e100 && cfg->stable_limit >= 0.0;
    p.colour_filter = filter_ok ? 1u : 0u;
    const double ak = std::fabs(p.filt_k);
    p.colour_filter32 = (filter_ok && o.colour_filter == 1 && n < (1u << 24) && ak >= 0x1p-60 && ak <= 0x1p60) ? 1u : 0u;
    p.filt_k32 = (float)p.filt_k;
    p.filt_c32 = p.colour_filter32 ? std::nextafterf((float)(ak * FR_NU_BRACKET * (1.0 + 0x1p-9)), INFINITY) : 0.0f;
    for (int k = 0; k < 3; k++) {
        p.filt_d[k] = p.prim_f[k] * ak * FR_NU_BRACKET * (1.0 + 0x1p-20);
        p.filt_d32[k] = p.colour_filter32 ? std::nextafterf((float)(p.prim_f[k] * ak * FR_NU_BRACKET * (1.0 + 0x1p-10)), INFINITY) : 0.0f;
        p.prim32[k] = (float)p.prim[k];
    }
    const double lo = std::fmax(cfg->stable_limit, 2.0) * (1.0 + 0x1p-20);
    p.filt_lo32 = (p.colour_filter32 && lo < 1e30) ? std::nextafterf((float)lo, INFINITY) : INFINITY;
}

/* coord_to_space — calc/src/lib.rs:182-184 — evaluated on the host ONLY to bound |c| over a launch
 * (the kernels compute every coordinate themselves). */
static double host_coord(double coord, double max, double offset, double pos, double scale) {
    return ((coord / max) - offset) / scale + pos;
}

constexpr uint32_t kSpecQuiet = 16;

/* Choose the orbit-loop plan for a launch whose local grid is already set in `p`
 * (see fr_kernels.hip, "orbit loop, scaled form", for what the kernel does with it and why it is
 * exact).  loop_mode = M in {4, 2} and skip_t = T such that
 *     dist_k <= T  and  every |c| component <= Cmax   =>   dist_{k+1..k+M-1} <= limit^2,
 * using dist' <= g(dist) = 2 * (dist + Cmax)^2 * (1 + slack); T is found by inverting g M-1 times
 * from limit^2.  Falls back to the unscaled loop (0) whenever the bound is useless or any
 * parameter is outside the range the scaled form is proven for. */
void plan_loop(const fr_config *cfg, int precision, const Opts &o, fr_kparams &p) {
    p.loop_mode = 0;
    p.skip_t = 0.0;
    p.loop_spec = 0;
    /* 5 = automatic, but without the speculative long blocks (A/B and tests) */
    const bool no_spec = o.loop_mode == 5;
    const int forced = no_spec ? -1 : o.loop_mode;
    if (cfg->algo != FR_ALGO_MANDELBROT && cfg->algo != FR_ALGO_JULIA) return;
    if (p.ncols == 0 || p.nrows == 0) return;
    const bool f32 = precision == FR_PRECISION_F32;
    const double range_hi = f32 ? 0x1p30 : 0x1p400;
    const double slack = f32 ? 1.0 + 0x1p-18 : 1.0 + 0x1p-30;
    double lim2; /* limit^2 as the kernels compute it (may be +inf or NaN) */
    if (f32) {
        const float lf = (float)cfg->limit;
        lim2 = (double)(lf * lf);
    } else {
        lim2 = cfg->limit * cfg->limit;
    }
    auto mag = [f32](double v) { return std::fabs(f32 ? (double)(float)v : v); };
    /* the largest start component of the launch: the map is monotone in x and in y, so the extremes are at the ends */
    const double w = (double)cfg->width, h = (double)cfg->height;
    const uint64_t x_last = (uint64_t)p.x_first + (uint64_t)(p.ncols - 1) * p.x_stride;
    const uint32_t r_last = p.nrows - 1;
    const uint64_t y_last = (uint64_t)p.y_first + (uint64_t)(r_last / p.block_rows) * p.y_stride + r_last % p.block_rows;
    if (x_last > 0xFFFFFFFFull || y_last > 0xFFFFFFFFull) return;
    auto nmax = [](double a, double b) { return (a != a || b != b) ? NAN : std::fmax(a, b); }; /* a NaN stays (std::fmax drops it) */
    double smax = 0.0;
    for (double x : {(double)p.x_first, (double)x_last})
        smax = nmax(smax, mag(host_coord(x, h, (w / h) / 2.0, cfg->pos.re, cfg->scale.re)));
    for (double y : {(double)p.y_first, (double)y_last})
        smax = nmax(smax, mag(host_coord(y, h, 0.5, cfg->pos.im, cfg->scale.im)));
    /* Mandelbrot: c = the pixel's coordinate; Julia: c = julia_set (calc/src/lib.rs:209-210) */
    const double cmax = cfg->algo == FR_ALGO_JULIA ? nmax(mag(cfg->julia_set.re), mag(cfg->julia_set.im)) : smax;
    /* Speculative long blocks (fr_kernels.hip: FR_SC_SPEC_BODY, FR_ORBIT_ASM, FR_FB_SPEC_LOOP) — in whichever loop form a
     * wave runs — need an escape inside a block to be visible at its end: once dist > limit^2 >= 16, with every |c|
     * component <= limit^2 / 8, each step multiplies |z| by more than 2.9 (|z'| >= |z|^2 - sqrt(2) Cmax - rounding >=
     * |z|^2 (3/4 - 8u), |z| > 4), so the computed distances grow from there — through +inf to NaN at worst — and the end
     * test `NOT (bound >= dist)` (bound = T or limit^2, both <= limit^2) is true for every one of them.  limit^2 must be
     * finite with room above it (an orbit must pass it BEFORE anything overflows: no NaN without an escape), and the
     * starts finite (a NaN start never escapes and would fail every block's test).  Quiet stretch before a wave
     * speculates: 16 iterations, doubling with every block thrown away. */
    if (!no_spec && lim2 >= 16.0 && lim2 <= (f32 ? 0x1p100 : 0x1p1000) && cmax <= lim2 / 8.0 && smax <= (f32 ? 0x1p120 : 0x1p1000)) {
        static const int dbg = getenv("FR_DEBUG_SPEC_QUIET") ? atoi(getenv("FR_DEBUG_SPEC_QUIET")) : 0; /* tuning aid */
        p.loop_spec = dbg > 0 ? (uint32_t)dbg : kSpecQuiet;
    }
    if (forced == 0) return;
    if (!(std::fabs(cfg->limit) <= range_hi)) return; /* also rejects NaN */
    if (!(cmax <= range_hi)) return; /* NaN / inf / huge */
    for (int m : {4, 2}) {
        if (forced > 0 && forced != m) continue;
        double d = lim2;
        for (int j = 1; j < m && d > 0.0; j++) d = std::sqrt(d / (2.0 * slack)) - cmax;
        double t = d * (1.0 - 0x1p-20);
        if (f32) { /* the kernel compares in f32: round T toward zero */
            float tf = (float)t;
            if ((double)tf > t) tf = std::nextafterf(tf, 0.0f);
            t = (double)tf;
        }
        /* points of the set keep |z|^2 <= 4; below that the fast path would hardly ever run */
        if (t >= 4.5 || (forced == m && t > 0.0)) {
            p.loop_mode = (uint32_t)m;
            p.skip_t = t;
            if (!(t < lim2)) p.loop_spec = 0; /* (cannot happen: T <= sqrt(limit^2 / 2); the scaled blocks test against T) */
            return;
        }
    }
}

static int check_rows(const fr_config *cfg, uint32_t y0, uint32_t y1) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (y0 > y1) return fail(FR_ERR_INVALID_ARGUMENT, "y0 > y1");
    if (y1 > cfg->height) return fail(FR_ERR_INVALID_ARGUMENT, "y1 > height");
    return FR_OK;
}

int check_precision(int precision) {
    if (precision != FR_PRECISION_F64 && precision != FR_PRECISION_F32)
        return fail(FR_ERR_INVALID_ARGUMENT, "precision must be FR_PRECISION_F64 or FR_PRECISION_F32");
    return FR_OK;
}

std::atomic<int> g_dispatch_sampling{1};

/* Which kernel for a launch?  Decided from the IMAGE, not from the algorithm's name.  A sample of 16 x 16 tiles of the
 * launch goes through the plain loop (view_sample_kernel; ~20-70 us of device time) and tells, for the tiles it saw:
 *   capped    the share of pixels still running at the sample's cap (interior, or deep boundary);
 *   handed    the share of pixels the first pass's own episode schedule would hand over to the survivor lists;
 *   waste     the lane-iterations that finishing those stragglers IN PLACE would idle away, as a share of the work —
 *             what the two-pass render's lists exist to save;
 *   mean      executed iterations per pixel.
 * Three kernel families (measured on 13 views x 2 precisions, tools/two_pass_views.py; profiles/r03_kernel_choice_views.txt
 * for 8192^2, profiles/r04_kernel_choice_midsize.txt for 2048^2 and 3840 x 2160):
 *   two passes        taken when the first pass's schedule would hand over at least one pixel in 500;
 *   first pass alone  (strips in episodes, frozen lanes finished once per tile, nothing handed over): 10-13 % ahead of the
 *                     strip kernel on views of short orbits, level with it on long ones;
 *   strips            1-3 % ahead on interior-heavy views (its loop is 6.5 vector instructions per iteration from the first).
 * WHEN the sample is taken depends on the launch:
 *   >= 131 072 tiles (4096 x 2048): BLOCKING, in front of the first launch of a view (40 us beside a render of milliseconds;
 *      the one host-blocking step of the device-pointer entry points; fr_set_dispatch_sampling(0) removes it);
 *   4096 .. 131 072 tiles (every frame a GUI asks for, src/gui.rs:56-82): NON-BLOCKING.  The first frame of a view is
 *      dispatched as before (by algorithm and size) and the sample is enqueued BEHIND its render on a stream of the
 *      library's own; the totals land in host-mapped memory, the view's key last.  The next frame of the same view — what
 *      every slider move re-requests — finds them and is dispatched from measured numbers.  Nothing waits, ever: a frame
 *      that comes before the totals goes by name once more.
 * The key hashes exactly the fields that determine ORBITS (algo, size, cap, limit, pos, scale, julia_set, the launch's grid,
 * the precision): a GUI changing colours, exposure, smooth or inside (src/gui.rs:183-203) keeps its view.  The last 32
 * views are remembered.  No sample while the caller's stream is being captured into a graph (neither a host wait nor a
 * cross-stream event belongs in a capture).  Returns 1 two passes, 2 first pass alone, 0 strips, -1 no opinion. */
constexpr uint64_t kSampleMinTiles = 131072; /* from here up the sample is taken in front of the first launch (blocking) */
constexpr uint64_t kMidRuleTiles = 524288;   /* under 8192 x 4096: the rule fitted to GUI-sized launches */
constexpr uint64_t kAsyncMinTiles = 4096;
constexpr uint32_t kSampleCap = 4096;

static uint64_t view_key(const fr_config *cfg, const fr_kparams &p, int precision) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void *data, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(data);
        for (size_t k = 0; k < n; k++) h = (h ^ b[k]) * 1099511628211ull;
    };
    /* field by field: no padding bytes (indeterminate in a caller-built struct), nothing the colour map alone reads */
    const uint32_t ints[4] = {cfg->algo, cfg->width, cfg->height, cfg->iterations};
    const double reals[7] = {cfg->limit, cfg->pos.re, cfg->pos.im, cfg->scale.re, cfg->scale.im, cfg->julia_set.re, cfg->julia_set.im};
    mix(ints, sizeof ints);
    mix(reals, sizeof reals);
    const uint32_t grid[8] = {p.ncols, p.nrows, p.x_first, p.x_stride, p.block_rows, p.y_first, p.y_stride, (uint32_t)precision};
    mix(grid, sizeof grid);
    return h ? h : 1;
}

/* blocking sample into result set `set` (kViewChoices = the debug hook's own) */
int sample_view(Ctx &ctx, const fr_kparams &p, int precision, int set, double out[7]) {
    std::lock_guard<std::mutex> lk(ctx.sample_mu);
    void *d_result = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d_result, ctx.sample_result, 0));
    unsigned long long *res = ctx.sample_result + 8 * set;
    const uint32_t cap_s = p.iterations < kSampleCap ? p.iterations : kSampleCap;
    HIP_TRY(fr_launch_view_sample(p, precision, 16, cap_s, 64, 48, ctx.sample_counters, static_cast<unsigned long long *>(d_result) + 8 * set, 0ull,
                                  ctx.aux_stream));
    HIP_TRY(hipStreamSynchronize(ctx.aux_stream));
    for (int k = 0; k < 7; k++) out[k] = (double)__atomic_load_n(res + k, __ATOMIC_RELAXED);
    return FR_OK;
}

struct Decision {
    int choice;
    bool one_band;
    uint32_t strip_tiles;
    bool no_spec; /* nothing in the view stays long enough for a speculative block to pay (plan_loop's loop_spec is cleared) */
};

static Decision decide_from_sample(const double st[7], uint64_t tiles, const fr_kparams &p, int precision, bool two_pass_ok) {
    const double lanes = 64.0 * st[2];
    const double capped = st[3] / lanes, waste = st[5] / st[0], mean = st[0] / lanes, handed = st[4] / lanes;
    /* Speculative long blocks (fr_kernels.hip: FR_SC_SPEC_BODY) pay where waves stay quiet for dozens of iterations.  In a
     * view in which no sampled pixel reaches the cap and orbits are short on average, the tiles that stay in the TWO-PASS
     * render's first kernel have no such waves (their stragglers leave for the lists) — C4's dust: mean 44, and the first
     * pass's speculative form costs it 1.5 % in set-up per tile that stays an episode — so that kernel runs in its plain form
     * there.  Only that kernel: where stragglers are finished in place (strips, the first pass alone, the unscaled loop of a
     * Julia constant with a zero component) they ARE the quiet waves, however short the mean — the dendrite c = i at 1920 x
     * 1080, mean 6.5: 0.058 ms without the blocks, 0.035 with — and the blocks cost those loops nothing when unused.
     * (Long orbits that all escape in the end — filaments at high caps — gain in the first kernel too: mean >= 96.) */
    Decision d{2, false, 0u, capped < 0.0005 && mean < 96.0};
    if (tiles >= kMidRuleTiles) {
        if (capped >= 0.10 && waste < 0.01)
            d.choice = 0; /* long orbits dominate and tiles stay full: the strip kernel's ground */
        else if (handed >= 0.002 || (handed * (double)p.ncols * (double)p.nrows >= 4096.0 && st[6] >= 128.0 * st[4]))
            d.choice = 1; /* thinned-out tiles would hand over at least one pixel in 500 — or fewer, but thousands of them with long
                           * tails (128 iterations and more to go, on average): finished in place those are a few workgroups' serial
                           * chains, in the lists they spread over the chip — the lists pay (measured from 131 072-tile
                           * launches up; an estimate of the idle time saved against the second pass's cost was tried as the
                           * criterion and mispredicted wide launches, whose second pass costs next to nothing) */
        else
            d.choice = 2;
        /* long orbits: workgroups of four strips (28 tiles) differ too much in cost to balance over the chip — C2 through
         * the first pass alone: 14.2 ms with four strips per workgroup, strips 13.3 (tools/c2c3_choice.py) */
        d.one_band = mean >= 128.0;
        return d;
    }
    /* Launches under 8192 x 4096 — every frame a GUI asks for (profiles/r04_kernel_choice_midsize.txt, 13 views x 2 precisions
     * at 1920 x 1080, 2048^2, 3840 x 2160, 4096^2).  What decides at these sizes is filling and balancing the chip, so the
     * one-tile strip kernel (the by-size default) is the best or within a few per cent on every view but two kinds: */
    if (capped < 0.001 && ((mean < 16.0 && handed < 0.002) || (mean < 32.0 && waste < 1.0 && tiles >= 100000))) {
        /* short orbits everywhere and no long stragglers (thin dusts, exteriors with no piece of the set in them; from
         * 3840 x 2160 up also Julia sets of a couple of dozen iterations): per-tile overhead is all there is, and the first
         * pass's is a third of the strip kernel's (3840 x 2160 thin dust: 0.040 ms against 0.050 for 4-tile and 0.069 for
         * 1-tile strips; julia 0.285+0.01i there: 0.065 against 0.081).  Strip length: 7 tiles for the very shortest orbits
         * and from 4096^2 up, 4 below.  A constant the scaled loop may not run with (the dendrite c = i) takes 4-tile strips of
         * the strip kernel instead */
        d.choice = two_pass_ok ? 2 : 0;
        d.strip_tiles = !two_pass_ok ? 4u : (mean < 8.0 || tiles >= 200000) ? 7u : 4u;
    } else if (two_pass_ok && handed >= 0.05 && waste >= 2.0 && tiles >= (precision == FR_PRECISION_F64 ? 60000u : 200000u)) {
        /* a Julia dust: most tiles thin out within an episode and their stragglers are long — the lists pay from 2048^2 up
         * in f64 (0.179 ms against 0.206 for 1-tile strips) and from 4096^2 up in f32 (0.268 against 0.350); below, the
         * strips are level or ahead.  waste >= 2 keeps Mandelbrot filaments out (waste 1.0-1.6): their lists carry c and
         * lose to strips by 2x */
        d.choice = 1;
        d.strip_tiles = tiles >= 200000 ? 7u : 4u;
    } else {
        d.choice = 0;
        d.strip_tiles = 1;
    }
    return d;
}

void Ctx::post_sample(int idx, hipStream_t stream) {
    if (idx < 0 || idx >= kViewChoices) return;
    std::lock_guard<std::mutex> lk(sample_mu);
    ViewChoice &v = view_choices[idx];
    if (v.state != 1) return;
    void *d_result = nullptr;
    const uint32_t cap_s = v.grid.iterations < kSampleCap ? v.grid.iterations : kSampleCap;
    const bool ok = hipHostGetDevicePointer(&d_result, sample_result, 0) == hipSuccess &&
                    hipEventRecord(v.after, stream) == hipSuccess && hipStreamWaitEvent(aux2_stream, v.after, 0) == hipSuccess &&
                    fr_launch_view_sample(v.grid, v.precision, 16, cap_s, 64, 48, sample_counters + 8,
                                          static_cast<unsigned long long *>(d_result) + 8 * idx, v.key, aux2_stream) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        v.state = 0, v.key = 0; /* the slot is free again; the view goes by name */
    }
}

static int choose_kernel(Ctx &ctx, const fr_config *cfg, const fr_kparams &p, int precision, const Opts &o, hipStream_t stream,
                         bool allow_async, bool *one_band, uint32_t *strip_tiles, int *pending, bool *no_spec) {
    *one_band = false;
    *no_spec = false;
    *strip_tiles = 0;
    *pending = -1;
    if (o.tile != 0 || !g_dispatch_sampling.load()) return -1;
    const uint64_t tiles = (((uint64_t)p.ncols + 7) / 8) * (((uint64_t)p.nrows + 7) / 8);
    if (tiles < kAsyncMinTiles) return -1;
    fr_kparams q = p; /* would two passes be possible at all?  If not, the statistics still choose the strip length */
    const bool two_pass_ok = fr_wants_two_pass(q, precision, 0, 1);
    if (!two_pass_ok && (tiles >= kMidRuleTiles || (cfg->algo != FR_ALGO_MANDELBROT && cfg->algo != FR_ALGO_JULIA))) return -1;
    const uint64_t key = view_key(cfg, p, precision);
    int slot_idx = -1;
    {
        std::lock_guard<std::mutex> lk(ctx.sample_mu);
        for (int k = 0; k < Ctx::kViewChoices; k++) {
            Ctx::ViewChoice &v = ctx.view_choices[k];
            if (v.key != key || v.state == 0) continue;
            if (v.state == 1) {
                /* a sample is in flight: are its totals there?  (the kernel writes the key LAST) */
                const unsigned long long *res = ctx.sample_result + 8 * k;
                if (__atomic_load_n(res + 7, __ATOMIC_ACQUIRE) != key) return -1; /* not yet: by name once more */
                double st[7];
                for (int j = 0; j < 7; j++) st[j] = (double)__atomic_load_n(res + j, __ATOMIC_RELAXED);
                if (st[1] <= 0.0 || st[0] <= 0.0) {
                    v.state = 2, v.two_pass = -1;
                } else {
                    const Decision d = decide_from_sample(st, tiles, p, precision, two_pass_ok);
                    v.state = 2, v.two_pass = d.choice, v.one_band = d.one_band, v.strip_tiles = d.strip_tiles, v.no_spec = d.no_spec;
                    v.lane_fraction = st[0] / st[1];
                }
            }
            *one_band = v.one_band;
            *strip_tiles = v.strip_tiles;
            *no_spec = v.no_spec;
            return v.two_pass;
        }
        /* a view not seen before.  No sample of either kind while the caller's stream is being captured */
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream != nullptr && hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
        if (cs != hipStreamCaptureStatusNone) return -1;
        if (tiles < kSampleMinTiles && !allow_async) return -1;
        slot_idx = (int)(ctx.view_next++ % Ctx::kViewChoices);
        Ctx::ViewChoice &v = ctx.view_choices[slot_idx];
        hipEvent_t ev = v.after;
        v = Ctx::ViewChoice();
        v.after = ev;
        v.key = key, v.state = 1, v.grid = p, v.precision = precision;
        __atomic_store_n(ctx.sample_result + 8 * slot_idx + 7, 0ull, __ATOMIC_RELEASE);
        if (tiles < kSampleMinTiles) {
            *pending = slot_idx; /* the caller posts it behind its render (Ctx::post_sample) */
            return -1;
        }
    }
    double st[7];
    const int rc = sample_view(ctx, p, precision, slot_idx, st);
    std::lock_guard<std::mutex> lk(ctx.sample_mu);
    Ctx::ViewChoice &v = ctx.view_choices[slot_idx];
    if (v.key != key || v.state != 1) return -1; /* (the slot was recycled meanwhile: 32 other new views) */
    if (rc != FR_OK || st[1] <= 0.0 || st[0] <= 0.0) { /* no opinion rather than a failed render */
        v.state = 0, v.key = 0;
        return -1;
    }
    const Decision d = decide_from_sample(st, tiles, p, precision, two_pass_ok);
    v.state = 2, v.two_pass = d.choice, v.one_band = d.one_band, v.strip_tiles = d.strip_tiles, v.lane_fraction = st[0] / st[1];
    v.no_spec = d.no_spec;
    *one_band = d.one_band;
    *strip_tiles = d.strip_tiles;
    *no_spec = d.no_spec;
    return d.choice;
}

void decide_kernel(Ctx &ctx, const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, Opts &o, hipStream_t stream,
                   bool allow_async) {
    if (o.tile != 0 || o.kernel_hint != -2 || y1 <= y0 || cfg->width == 0) return;
    fr_kparams p;
    fill_params(cfg, o, p);
    p.nrows = y1 - y0;
    p.y_first = y0;
    p.block_rows = p.nrows;
    p.y_stride = 0;
    plan_loop(cfg, precision, o, p);
    bool one_band = false, no_spec = false;
    uint32_t strip_tiles = 0;
    int pending = -1;
    o.kernel_hint = choose_kernel(ctx, cfg, p, precision, o, stream, allow_async, &one_band, &strip_tiles, &pending, &no_spec);
    o.one_band = one_band;
    o.no_spec = no_spec;
    o.strip_tiles = strip_tiles;
    o.pending_sample = pending;
}

/* device-pointer render of an arbitrary local grid; no shared scratch except the slots `ctx` lends: re-entrant.  Host
 * synchronisation: none, with ONE exception — the first launch of a view of 131 072 tiles and more takes a blocking
 * 40 us sample (choose_kernel; off under stream capture and with fr_set_dispatch_sampling(0)). */
int render_device(Ctx &ctx, const fr_config *cfg, fr_kparams &p, int precision, const Opts &o, void *d_out,
                  hipStream_t stream) {
    plan_loop(cfg, precision, o, p);
    /* smooth == false: the outside colour is a function of the escape index alone, so build the
     * (iterations + 1)-entry palette once (into a context-owned slot) and let every workgroup stage it
     * in LDS.  Larger palettes would cost occupancy; they are computed per pixel instead. */
    const bool escape_algo = cfg->algo == FR_ALGO_MANDELBROT || cfg->algo == FR_ALGO_JULIA;
    PaletteSlot *slot = nullptr;
    const bool want_palette = !cfg->smooth && escape_algo && o.palette && cfg->iterations < FR_MAX_PALETTE_ENTRIES && o.tile <= 16;
    bool one_band = o.one_band, no_spec = o.no_spec;
    uint32_t strip_tiles = o.strip_tiles;
    int pending = -1;
    const int hint = o.kernel_hint != -2 ? o.kernel_hint
                                         : choose_kernel(ctx, cfg, p, precision, o, stream, true, &one_band, &strip_tiles, &pending, &no_spec);
    p.first_no_spec = no_spec ? 1u : 0u; /* the view's statistics: nothing stays (decide_from_sample) — the two-pass render's first kernel runs in its plain form */
    struct SampleGuard { /* whatever happens below, a slot that was promised a sample gets it (or is freed) */
        Ctx &ctx;
        int idx;
        hipStream_t stream;
        ~SampleGuard() { ctx.post_sample(idx, stream); }
    } sample_guard{ctx, pending, stream};
    const bool want_two_pass = fr_wants_two_pass(p, precision, o.tile, hint);
    if (o.tile == 0 && hint >= 0 && strip_tiles) p.strip_tiles = strip_tiles;
    p.first_one_band = one_band ? 1u : 0u;
    p.second_v1 = (o.tile == 12 || o.tile == 14) ? 1u : 0u; /* 12 = round 2's two kernels, 14 = its second pass behind this round's first */
    {
        static const int dbg = getenv("FR_DEBUG_FIRST_ONE_BAND") ? atoi(getenv("FR_DEBUG_FIRST_ONE_BAND")) : -1; /* tuning aid */
        if (dbg >= 0) p.first_one_band = dbg ? 1u : 0u;
    }
    const bool want_lists = want_two_pass && !p.first_only;
    const bool want_queue = want_lists || fr_wants_work_queue(p, o.tile);
    if (want_palette || want_queue) {
        int rc = ctx.acquire_palette(&slot);
        if (rc != FR_OK) return rc;
    }
    SurvSlot *surv = nullptr;
    if (want_lists) {
        /* room for an eighth of the pixels (C4 hands over one in fifteen; the views of tools/two_pass_views.py at most
         * one in twelve); what does not fit is finished by the first pass itself, so the size is a matter of speed only */
        const uint64_t npix = (uint64_t)p.ncols * p.nrows;
        uint64_t entries = npix / 8;
        if (entries < 65536) entries = 65536;
        if (entries > (256ull << 20)) entries = 256ull << 20;
        uint64_t sub = (entries + FR_SURV_QUEUES - 1) / FR_SURV_QUEUES;
        if (const uint32_t forced = g_two_pass_list_entries.load()) sub = forced;
        sub = (sub + FR_SURV_CHUNK - 1) / FR_SURV_CHUNK * FR_SURV_CHUNK;
        /* the 64 lists fill at about the same rate: keep their write heads off a common power-of-two stride (the same
         * memory channel for all of them) */
        if ((sub & 4095u) == 0 && !g_two_pass_list_entries.load()) sub += 3 * FR_SURV_CHUNK;
        p.surv_sub_capacity = (uint32_t)sub;
        const fr_two_pass_layout lay = fr_two_pass_bytes(p, precision, p.surv_sub_capacity);
        int rc = ctx.acquire_surv(lay.total, &surv);
        if (rc != FR_OK) {
            ctx.release_palette(slot, stream);
            return rc;
        }
        char *base = static_cast<char *>(surv->dev);
        p.surv_z = base + lay.z_off;
        p.surv_pos = reinterpret_cast<uint32_t *>(base + lay.pos_off);
        p.surv_cnt = reinterpret_cast<uint32_t *>(base + lay.cnt_off);
        p.surv_c = base + lay.c_off;
        p.surv_counts = reinterpret_cast<uint32_t *>(base + lay.counts_off);
        hipError_t e = hipMemsetAsync(p.surv_counts, 0, FR_SURV_QUEUES * FR_SURV_COUNT_STRIDE * sizeof(uint32_t), stream);
        if (e != hipSuccess) {
            ctx.release_surv(surv, stream);
            ctx.release_palette(slot, stream);
            return fail_hip(e, "hipMemsetAsync(survivor counters)");
        }
    }
    if (want_queue) { /* the persistent waves' claim counters live behind the slot's palette words */
        p.work_counter = slot->dev + FR_MAX_PALETTE_ENTRIES;
        hipError_t e = hipMemsetAsync(p.work_counter, 0, sizeof(uint32_t) * FR_SURV_QUEUES * FR_SURV_COUNT_STRIDE, stream);
        if (e != hipSuccess) {
            ctx.release_surv(surv, stream);
            ctx.release_palette(slot, stream);
            return fail_hip(e, "hipMemsetAsync(work counter)");
        }
    }
    if (want_palette) {
        p.palette = slot->dev;
        p.palette_entries = cfg->iterations + 1;
        hipError_t e = fr_launch_palette(p, slot->dev, stream);
        if (e != hipSuccess) {
            ctx.release_surv(surv, stream);
            ctx.release_palette(slot, stream);
            return fail_hip(e, "fr_launch_palette");
        }
    }
    struct SlotGuard {
        Ctx &ctx;
        PaletteSlot *slot;
        SurvSlot *surv;
        hipStream_t stream;
        ~SlotGuard() {
            ctx.release_surv(surv, stream);
            ctx.release_palette(slot, stream);
        }
    } guard{ctx, slot, surv, stream};
    fr_kout out{};
    out.rgb = static_cast<uint8_t *>(d_out);
    out.trace = g_queue_trace.load();
    Profiling &pr = tl_prof;
    if (pr.enabled) {
        if (!pr.e0) {
            HIP_TRY(hipEventCreate(&pr.e0));
            HIP_TRY(hipEventCreate(&pr.e1));
        }
        HIP_TRY(hipEventRecord(pr.e0, stream));
    }
    const char *kname = "";
    HIP_TRY(fr_launch_escape(p, precision, FR_OUT_RGB, out, o.tile, stream, &kname));
    if (pr.enabled) {
        HIP_TRY(hipEventRecord(pr.e1, stream));
        pr.have = true;
        pr.kernel = kname;
    }
    return FR_OK;
}

int render_block_cyclic(Ctx &ctx, const fr_config *cfg, int precision, const Opts &o, uint32_t block_rows,
                        uint32_t first_block, uint32_t block_stride, uint32_t max_blocks, int dest_is_image,
                        void *d_out, size_t out_len, hipStream_t stream, uint64_t *rows_written) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    int rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    if (block_rows == 0 || block_stride == 0)
        return fail(FR_ERR_INVALID_ARGUMENT, "block_rows and block_stride must be > 0");
    if ((uint64_t)block_rows * block_stride > 0xFFFFFFFFull)
        return fail(FR_ERR_INVALID_ARGUMENT, "block_rows * block_stride overflows u32");
    if (dest_is_image && block_rows % 8 != 0)
        return fail(FR_ERR_INVALID_ARGUMENT, "in-place block-cyclic rendering needs block_rows % 8 == 0");
    /* rows of blocks first_block, first_block + stride, ... (at most max_blocks of them; 0 = all) */
    uint64_t rows = 0, blocks = 0;
    for (uint64_t b = first_block; b * block_rows < cfg->height && (max_blocks == 0 || blocks < max_blocks);
         b += block_stride, blocks++) {
        const uint64_t left = cfg->height - b * block_rows;
        rows += left < block_rows ? left : block_rows;
    }
    if (rows_written) *rows_written = rows;
    const size_t need = dest_is_image ? (size_t)3 * cfg->width * (size_t)cfg->height : (size_t)3 * cfg->width * (size_t)rows;
    if (rows == 0 || cfg->width == 0) return FR_OK;
    if (!d_out) return fail(FR_ERR_INVALID_ARGUMENT, "d_out is NULL");
    if (out_len < need)
        return fail(FR_ERR_BUFFER_TOO_SMALL, dest_is_image ? "out_len < 3*width*height" : "out_len < 3*width*rows");
    fr_kparams p;
    fill_params(cfg, o, p);
    p.nrows = (uint32_t)rows;
    p.block_rows = block_rows;
    p.y_first = first_block * block_rows;
    p.y_stride = block_rows * block_stride;
    p.out_in_place = dest_is_image ? 1u : 0u;
    return render_device(ctx, cfg, p, precision, o, d_out, stream);
}

}  // namespace fr

using namespace fr;



extern "C" {

int fr_abi_version(void) { return FR_ABI_VERSION; }

#ifndef FR_BUILD_ID
#define FR_BUILD_ID "unknown"
#endif
const char *fr_build_id(void) { return FR_BUILD_ID; }

const char *fr_last_error(void) { return last_error().c_str(); }

int fr_device_count(int *count) {
    if (!count) return fail(FR_ERR_INVALID_ARGUMENT, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return FR_OK;
}

int fr_init(int device) {
    {
        /* same device (or "keep"): nothing is torn down, so calls in flight need not be waited for */
        LifeShared ls;
        std::lock_guard<std::mutex> lk(g_primary_mu);
        if (g_primary_inited && (device < 0 || device == g_primary.hip_device)) return FR_OK;
    }
    LifeExclusive lx; /* waits for every entry point that is using library state */
    std::lock_guard<std::mutex> lk(g_primary_mu);
    return init_primary_locked(device);
}

int fr_shutdown(void) {
    LifeExclusive lx;
    multi_shutdown_locked();
    std::lock_guard<std::mutex> lk(g_primary_mu);
    if (!g_primary_inited) return FR_OK;
    g_primary.destroy();
    g_primary_inited = false;
    return FR_OK;
}

int fr_device_name(char *buf, size_t buf_len) {
    if (!buf || buf_len == 0) return fail(FR_ERR_INVALID_ARGUMENT, "buf is NULL or empty");
    LifeShared ls;
    Ctx *ctx;
    int rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->hip_device));
    snprintf(buf, buf_len, "%s", prop.gcnArchName);
    return FR_OK;
}

/* Config::new — calc/src/lib.rs:39-69 (the stored RGB fields follow RGB::new(r, b, g), :129-131) */
void fr_config_new(fr_config *cfg, uint32_t algo) {
    if (!cfg) return;
    const bool fern = algo == FR_ALGO_BARNSLEY_FERN;
    memset(cfg, 0, sizeof *cfg);
    cfg->algo = algo;
    cfg->width = 2000;
    cfg->height = 1000;
    cfg->iterations = fern ? 10000000u : 50u;
    cfg->limit = 65536.0;
    cfg->stable_limit = 2.0;
    cfg->scale.re = 0.4;
    cfg->scale.im = 0.4;
    cfg->exposure = 2.0;
    cfg->inside = 1;
    cfg->smooth = 1;
    if (fern) {
        cfg->primary_color = fr_rgb{4, 3, 100};       /* new(4, 100, 3)     */
        cfg->secondary_color = fr_rgb{240, 240, 240}; /* new(240, 240, 240) */
    } else {
        cfg->primary_color = fr_rgb{40, 255, 40};   /* new(40, 40, 255)  */
        cfg->secondary_color = fr_rgb{240, 0, 170}; /* new(240, 170, 0)  */
    }
    cfg->color_weight = 0.01;
}

void fr_render_opts_init(fr_render_opts *opts) {
    if (!opts) return;
    const Opts o = default_opts();
    opts->size = sizeof(fr_render_opts);
    opts->tile = o.tile;
    opts->loop_mode = o.loop_mode;
    opts->palette = o.palette;
    opts->cycle_shortcut = o.cycle_shortcut;
    opts->refill_minrun = o.refill_minrun;
    opts->refill_quit16 = o.refill_quit16;
    opts->colour_filter = o.colour_filter;
}

/* ---- device-pointer renders: asynchronous, lock-free apart from the palette slot ring ---------- */

static int render_rows_device(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                              size_t out_len, void *hip_stream, unsigned bytes_per_pixel, const fr_render_opts *opts) {
    int rc = check_rows(cfg, y0, y1);
    if (rc == FR_OK) rc = check_precision(precision);
    Opts o;
    if (rc == FR_OK) rc = resolve_opts(opts, o);
    if (rc != FR_OK) return rc;
    const size_t need = (size_t)bytes_per_pixel * cfg->width * (size_t)(y1 - y0);
    if (need == 0) return FR_OK;
    if (!d_out) return fail(FR_ERR_INVALID_ARGUMENT, "d_out is NULL");
    if (out_len < need) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < bytes_per_pixel*width*(y1-y0)");
    if (bytes_per_pixel == 4 && (reinterpret_cast<uintptr_t>(d_out) & 3u))
        return fail(FR_ERR_INVALID_ARGUMENT, "RGBA8 output must be 4-byte aligned");
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    fr_kparams p;
    fill_params(cfg, o, p);
    p.nrows = y1 - y0;
    p.y_first = y0;
    p.block_rows = p.nrows;
    p.y_stride = 0;
    p.out_rgba = bytes_per_pixel == 4 ? 1u : 0u;
    return render_device(*ctx, cfg, p, precision, o, d_out, static_cast<hipStream_t>(hip_stream));
}

int fr_render_rows_rgb8_device_opts(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                                    size_t out_len, void *hip_stream, const fr_render_opts *opts) {
    return render_rows_device(cfg, precision, y0, y1, d_out, out_len, hip_stream, 3, opts);
}

int fr_render_rows_rgba8_device_opts(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                                     size_t out_len, void *hip_stream, const fr_render_opts *opts) {
    return render_rows_device(cfg, precision, y0, y1, d_out, out_len, hip_stream, 4, opts);
}

int fr_render_rows_rgb8_device(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                               size_t out_len, void *hip_stream) {
    return render_rows_device(cfg, precision, y0, y1, d_out, out_len, hip_stream, 3, nullptr);
}

int fr_render_rows_rgba8_device(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                                size_t out_len, void *hip_stream) {
    return render_rows_device(cfg, precision, y0, y1, d_out, out_len, hip_stream, 4, nullptr);
}

uint64_t fr_block_cyclic_rows(uint32_t height, uint32_t block_rows, uint32_t first_block, uint32_t block_stride) {
    if (block_rows == 0 || block_stride == 0) return 0;
    uint64_t rows = 0;
    for (uint64_t b = first_block; b * block_rows < height; b += block_stride) {
        const uint64_t start = b * block_rows;
        const uint64_t left = height - start;
        rows += left < block_rows ? left : block_rows;
    }
    return rows;
}

int fr_render_block_cyclic_range_rgb8_device_opts(const fr_config *cfg, int precision, uint32_t block_rows,
                                                  uint32_t first_block, uint32_t block_stride, uint32_t max_blocks,
                                                  int dest_is_image, void *d_out, size_t out_len, void *hip_stream,
                                                  uint64_t *rows_written, const fr_render_opts *opts) {
    Opts o;
    int rc = resolve_opts(opts, o);
    if (rc != FR_OK) return rc;
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    return render_block_cyclic(*ctx, cfg, precision, o, block_rows, first_block, block_stride, max_blocks, dest_is_image,
                               d_out, out_len, static_cast<hipStream_t>(hip_stream), rows_written);
}

int fr_render_block_cyclic_range_rgb8_device(const fr_config *cfg, int precision, uint32_t block_rows,
                                             uint32_t first_block, uint32_t block_stride, uint32_t max_blocks,
                                             int dest_is_image, void *d_out, size_t out_len, void *hip_stream,
                                             uint64_t *rows_written) {
    return fr_render_block_cyclic_range_rgb8_device_opts(cfg, precision, block_rows, first_block, block_stride,
                                                         max_blocks, dest_is_image, d_out, out_len, hip_stream,
                                                         rows_written, nullptr);
}

int fr_render_block_cyclic_rgb8_device(const fr_config *cfg, int precision, uint32_t block_rows,
                                       uint32_t first_block, uint32_t block_stride, void *d_out, size_t out_len,
                                       void *hip_stream, uint64_t *rows_written) {
    return fr_render_block_cyclic_range_rgb8_device_opts(cfg, precision, block_rows, first_block, block_stride, 0, 0,
                                                         d_out, out_len, hip_stream, rows_written, nullptr);
}

/* ---- host-buffer renders: the pipeline lives in fr_host.hip ----------------------------------- */

int fr_render_block_cyclic_rgb8(const fr_config *cfg, int precision, uint32_t block_rows, uint32_t first_block,
                                uint32_t block_stride, uint8_t *out, size_t out_len, uint64_t *rows_written) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (block_rows == 0 || block_stride == 0)
        return fail(FR_ERR_INVALID_ARGUMENT, "block_rows and block_stride must be > 0");
    const uint64_t rows = fr_block_cyclic_rows(cfg->height, block_rows, first_block, block_stride);
    if (rows_written) *rows_written = rows;
    const size_t need = (size_t)3 * cfg->width * (size_t)rows;
    if (need == 0) return check_precision(precision);
    if (!out) return fail(FR_ERR_INVALID_ARGUMENT, "out is NULL");
    if (out_len < need) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 3*width*rows");
    LifeShared ls;
    Ctx *ctx;
    int rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    rc = ctx->reserve(ctx->rgb, need);
    if (rc != FR_OK) return rc;
    rc = render_block_cyclic(*ctx, cfg, precision, default_opts(), block_rows, first_block, block_stride, 0, 0,
                             ctx->rgb.ptr, need, ctx->stream, nullptr);
    if (rc != FR_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->rgb.ptr, need, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

int fr_render_rows_rgb8_opts(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out,
                             size_t out_len, const fr_render_opts *opts) {
    return fr_host_render_rows(cfg, precision, y0, y1, out, out_len, 3, opts);
}

int fr_render_rows_rgb8(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out,
                        size_t out_len) {
    return fr_host_render_rows(cfg, precision, y0, y1, out, out_len, 3, nullptr);
}

int fr_render_rows_rgba8(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out, size_t out_len) {
    return fr_host_render_rows(cfg, precision, y0, y1, out, out_len, 4, nullptr);
}

int fr_render_rgb8(const fr_config *cfg, uint8_t *out, size_t out_len) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    return fr_host_render_rows(cfg, FR_PRECISION_F64, 0, cfg->height, out, out_len, 3, nullptr);
}

int fr_pixel_p(const fr_config *cfg, int precision, uint32_t x, uint32_t y, fr_rgb *out) {
    if (!cfg || !out) return fail(FR_ERR_INVALID_ARGUMENT, "cfg or out is NULL");
    int rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    /* get_recursive_pixel takes any u32 x, y — it does not clamp to width/height */
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    rc = ctx->reserve(ctx->misc, 256);
    if (rc != FR_OK) return rc;
    const Opts o = default_opts();
    fr_kparams p;
    fill_params(cfg, o, p);
    p.ncols = 1;
    p.nrows = 1;
    p.x_first = x;
    p.y_first = y;
    plan_loop(cfg, precision, o, p);
    fr_kout ko{};
    ko.rgb = static_cast<uint8_t *>(ctx->misc.ptr);
    HIP_TRY(fr_launch_escape(p, precision, FR_OUT_RGB, ko, o.tile, ctx->stream, nullptr));
    uint8_t rgb[3];
    HIP_TRY(hipMemcpyAsync(rgb, ctx->misc.ptr, 3, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    out->r = rgb[0];
    out->g = rgb[1];
    out->b = rgb[2];
    return FR_OK;
}

int fr_pixel(const fr_config *cfg, uint32_t x, uint32_t y, fr_rgb *out) {
    return fr_pixel_p(cfg, FR_PRECISION_F64, x, y, out);
}

int fr_recursive_batch(uint32_t iterations, const fr_imaginary *start, const fr_imaginary *c, size_t n,
                       double limit, int precision, fr_imaginary *out_pos, uint32_t *out_iters) {
    int rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    if (n == 0) return FR_OK;
    if (!start || !c || !out_pos || !out_iters) return fail(FR_ERR_INVALID_ARGUMENT, "NULL array");
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t zb = n * sizeof(fr_imaginary);
    rc = ctx->reserve(ctx->z, 3 * zb);
    if (rc == FR_OK) rc = ctx->reserve(ctx->iters, n * sizeof(uint32_t));
    if (rc != FR_OK) return rc;
    double *d_start = static_cast<double *>(ctx->z.ptr);
    double *d_c = d_start + 2 * n;
    double *d_pos = d_c + 2 * n;
    HIP_TRY(hipMemcpyAsync(d_start, start, zb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_c, c, zb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(fr_launch_recursive_batch(iterations, d_start, d_c, n, limit, precision, d_pos,
                                      static_cast<uint32_t *>(ctx->iters.ptr), ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_pos, d_pos, zb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_iters, ctx->iters.ptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

int fr_recursive(uint32_t iterations, fr_imaginary start, fr_imaginary c, double limit, fr_imaginary *out_pos,
                 uint32_t *out_iters) {
    return fr_recursive_batch(iterations, &start, &c, 1, limit, FR_PRECISION_F64, out_pos, out_iters);
}

int fr_escape_rows(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, double *z_re_im,
                   uint32_t *iters) {
    int rc = check_rows(cfg, y0, y1);
    if (rc == FR_OK) rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    const size_t npx = (size_t)cfg->width * (size_t)(y1 - y0);
    if (npx == 0 || (!z_re_im && !iters)) return FR_OK;
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (z_re_im) rc = ctx->reserve(ctx->z, npx * 2 * sizeof(double));
    if (rc == FR_OK && iters) rc = ctx->reserve(ctx->iters, npx * sizeof(uint32_t));
    if (rc != FR_OK) return rc;
    const Opts o = default_opts();
    fr_kparams p;
    fill_params(cfg, o, p);
    p.nrows = y1 - y0;
    p.y_first = y0;
    p.block_rows = p.nrows;
    p.y_stride = 0;
    plan_loop(cfg, precision, o, p);
    fr_kout ko{};
    ko.z = z_re_im ? static_cast<double *>(ctx->z.ptr) : nullptr;
    ko.iters = iters ? static_cast<uint32_t *>(ctx->iters.ptr) : nullptr;
    HIP_TRY(fr_launch_escape(p, precision, FR_OUT_ESCAPE, ko, o.tile, ctx->stream, nullptr));
    if (z_re_im) HIP_TRY(hipMemcpyAsync(z_re_im, ctx->z.ptr, npx * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (iters) HIP_TRY(hipMemcpyAsync(iters, ctx->iters.ptr, npx * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

int fr_colour_rgb8(const fr_config *cfg, const double *z_re_im, const uint32_t *iters, size_t n, uint8_t *out,
                   size_t out_len) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (n == 0) return FR_OK;
    if (!z_re_im || !iters || !out) return fail(FR_ERR_INVALID_ARGUMENT, "NULL array");
    if (out_len < 3 * n) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 3*n");
    LifeShared ls;
    Ctx *ctx;
    int rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    rc = ctx->reserve(ctx->z, n * 2 * sizeof(double));
    if (rc == FR_OK) rc = ctx->reserve(ctx->iters, n * sizeof(uint32_t));
    if (rc == FR_OK) rc = ctx->reserve(ctx->rgb, 3 * n);
    if (rc != FR_OK) return rc;
    fr_kparams p;
    fill_params(cfg, default_opts(), p);
    HIP_TRY(hipMemcpyAsync(ctx->z.ptr, z_re_im, n * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->iters.ptr, iters, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(fr_launch_colour(p, static_cast<const double *>(ctx->z.ptr), static_cast<const uint32_t *>(ctx->iters.ptr), n,
                             static_cast<uint8_t *>(ctx->rgb.ptr), ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, ctx->rgb.ptr, 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

int fr_colour_rgb8_device(const fr_config *cfg, const void *d_z_re_im, const void *d_iters, size_t n, void *d_out,
                          size_t out_len, void *hip_stream) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (n == 0) return FR_OK;
    if (!d_z_re_im || !d_iters || !d_out) return fail(FR_ERR_INVALID_ARGUMENT, "NULL array");
    if (out_len < 3 * n) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 3*n");
    fr_kparams p;
    fill_params(cfg, default_opts(), p);
    HIP_TRY(fr_launch_colour(p, static_cast<const double *>(d_z_re_im), static_cast<const uint32_t *>(d_iters), n,
                             static_cast<uint8_t *>(d_out), static_cast<hipStream_t>(hip_stream)));
    return FR_OK;
}

int fr_count_iterations(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint32_t sx, uint32_t sy,
                        uint64_t *total, uint64_t *pixels) {
    int rc = check_rows(cfg, y0, y1);
    if (rc == FR_OK) rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    if (!total) return fail(FR_ERR_INVALID_ARGUMENT, "total is NULL");
    if (sx == 0) sx = 1;
    if (sy == 0) sy = 1;
    *total = 0;
    const Opts o = default_opts();
    fr_kparams p;
    fill_params(cfg, o, p);
    p.ncols = (uint32_t)(((uint64_t)cfg->width + sx - 1) / sx);
    p.x_stride = sx;
    const uint64_t yf = ((uint64_t)y0 + sy - 1) / sy * sy; /* first sampled row >= y0 */
    p.nrows = yf < y1 ? (uint32_t)((y1 - 1 - yf) / sy + 1) : 0;
    p.y_first = (uint32_t)yf;
    p.block_rows = 1;
    p.y_stride = sy;
    if (pixels) *pixels = (uint64_t)p.ncols * p.nrows;
    if (p.ncols == 0 || p.nrows == 0) return FR_OK;
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t slot_bytes = sizeof(unsigned long long) * FR_COUNT_SLOTS;
    rc = ctx->reserve(ctx->misc, slot_bytes);
    if (rc != FR_OK) return rc;
    HIP_TRY(hipMemsetAsync(ctx->misc.ptr, 0, slot_bytes, ctx->stream));
    plan_loop(cfg, precision, o, p);
    fr_kout ko{};
    ko.count = static_cast<unsigned long long *>(ctx->misc.ptr);
    HIP_TRY(fr_launch_escape(p, precision, FR_OUT_COUNT, ko, o.tile, ctx->stream, nullptr));
    std::vector<unsigned long long> host(FR_COUNT_SLOTS);
    HIP_TRY(hipMemcpyAsync(host.data(), ctx->misc.ptr, slot_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long sum = 0;
    for (unsigned long long v : host) sum += v;
    *total = sum;
    return FR_OK;
}

int fr_set_profiling(int enabled) {
    Profiling &pr = profiling();
    pr.enabled = enabled != 0;
    if (!enabled) {
        pr.have = false;
        if (pr.e0) { /* give the events back: a thread that stops profiling holds nothing */
            (void)hipEventDestroy(pr.e0);
            (void)hipEventDestroy(pr.e1);
            pr.e0 = pr.e1 = nullptr;
        }
    }
    return FR_OK;
}

int fr_last_kernel_ms(float *ms) {
    if (!ms) return fail(FR_ERR_INVALID_ARGUMENT, "ms is NULL");
    Profiling &pr = profiling();
    if (!pr.have) return fail(FR_ERR_INVALID_ARGUMENT, "no profiled kernel on this thread");
    HIP_TRY(hipEventSynchronize(pr.e1));
    HIP_TRY(hipEventElapsedTime(ms, pr.e0, pr.e1));
    return FR_OK;
}

int fr_last_kernel_name(char *buf, size_t buf_len) {
    if (!buf || buf_len == 0) return fail(FR_ERR_INVALID_ARGUMENT, "buf is NULL or empty");
    Profiling &pr = profiling();
    if (!pr.have) return fail(FR_ERR_INVALID_ARGUMENT, "no profiled kernel on this thread");
    snprintf(buf, buf_len, "%s", pr.kernel);
    return FR_OK;
}

int fr_set_tile(int tile) {
    if (!valid_tile(tile)) return fail(FR_ERR_INVALID_ARGUMENT, "tile must be 0, 1, 2, 4, 8, 9, 10 ... 16, 6401, 3202, 1604 or 808");
    g_tile.store(tile);
    return FR_OK;
}

int fr_set_refill_policy(int minrun, int quit16) {
    if (minrun < -1 || quit16 < -1 || quit16 == 0 || quit16 > 16)
        return fail(FR_ERR_INVALID_ARGUMENT, "minrun >= 0, 1 <= quit16 <= 16 (or -1: the kernel's default)");
    g_refill_minrun.store(minrun);
    g_refill_quit16.store(quit16);
    return FR_OK;
}

int fr_set_cycle_shortcut(int enabled) {
    g_cycle_shortcut.store(enabled ? 1 : 0);
    return FR_OK;
}

int fr_set_palette(int enabled) {
    g_palette_enabled.store(enabled ? 1 : 0);
    return FR_OK;
}

int fr_set_colour_filter(int enabled) {
    g_colour_filter.store(enabled == 2 ? 2 : enabled ? 1 : 0); /* 2: the f64 stage only (tests) */
    return FR_OK;
}

int fr_set_loop_mode(int mode) {
    if (mode != -1 && mode != 0 && mode != 2 && mode != 4 && mode != 5)
        return fail(FR_ERR_INVALID_ARGUMENT, "loop mode must be -1 (auto), 0, 2, 4 or 5");
    g_loop_mode.store(mode);
    return FR_OK;
}

int fr_set_dispatch_sampling(int enabled) {
    g_dispatch_sampling.store(enabled ? 1 : 0);
    return FR_OK;
}

/* what choose_kernel measures, for tools and tests: out[0..5] = executed iterations, 64 x sum of per-tile maxima, tiles
 * sampled, lanes at the sample's cap (1024), lanes handed over after a 64-iteration episode (keep 48), lane-iterations
 * wasted by finishing those in place; out[6] = out[0] / out[1], the useful-lane fraction of one tile per wave; out[7] =
 * iterations the handed-over lanes still have to run */
int fr_debug_sample_view(const fr_config *cfg, int precision, double out[8]) {
    if (!cfg || !out) return fail(FR_ERR_INVALID_ARGUMENT, "cfg or out is NULL");
    int rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    fr_kparams p;
    fill_params(cfg, default_opts(), p);
    p.nrows = cfg->height;
    p.block_rows = cfg->height ? cfg->height : 1;
    p.y_stride = 0;
    if (p.ncols == 0 || p.nrows == 0) return fail(FR_ERR_INVALID_ARGUMENT, "empty image");
    double st[7];
    rc = sample_view(*ctx, p, precision, Ctx::kViewChoices, st);
    if (rc != FR_OK) return rc;
    for (int k = 0; k < 6; k++) out[k] = st[k];
    out[6] = st[1] > 0.0 ? st[0] / st[1] : 0.0;
    out[7] = st[6];
    return FR_OK;
}

/* the orbit-loop plan of (cfg, precision) rendered as one launch, with the process's current selectors: host arithmetic only
 * (no device needed), so that the CPU suite can pin when the scaled loop and its speculative blocks are allowed */
int fr_debug_loop_plan(const fr_config *cfg, int precision, uint32_t *loop_mode, double *skip_t, uint32_t *spec_quiet) {
    if (!cfg || !loop_mode || !skip_t || !spec_quiet) return fail(FR_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    const Opts o = default_opts();
    fr_kparams p;
    fill_params(cfg, o, p);
    p.nrows = cfg->height;
    p.block_rows = cfg->height ? cfg->height : 1;
    p.y_stride = 0;
    plan_loop(cfg, precision, o, p);
    *loop_mode = p.loop_mode;
    *skip_t = p.skip_t;
    *spec_quiet = p.loop_spec;
    return FR_OK;
}

int fr_debug_view_choice(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, int *state, int *choice,
                         uint32_t *strip_tiles) {
    if (!cfg || !state || !choice || !strip_tiles) return fail(FR_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_rows(cfg, y0, y1);
    if (rc == FR_OK) rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    LifeShared ls;
    Ctx *ctx;
    rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    fr_kparams p;
    fill_params(cfg, default_opts(), p);
    p.nrows = y1 - y0;
    p.y_first = y0;
    p.block_rows = p.nrows ? p.nrows : 1;
    p.y_stride = 0;
    const uint64_t key = view_key(cfg, p, precision);
    *state = 0, *choice = -1, *strip_tiles = 0;
    std::lock_guard<std::mutex> lk(ctx->sample_mu);
    for (int k = 0; k < Ctx::kViewChoices; k++) {
        const Ctx::ViewChoice &v = ctx->view_choices[k];
        if (v.key != key || v.state == 0) continue;
        *state = v.state;
        if (v.state == 1 && __atomic_load_n(ctx->sample_result + 8 * k + 7, __ATOMIC_ACQUIRE) == key) *state = 3; /* totals are in, not read yet */
        *choice = v.two_pass;
        *strip_tiles = v.strip_tiles;
        break;
    }
    return FR_OK;
}

/* tuning aid: device buffer the work-queue kernel's waves write their start / end time, work counts and per-phase
 * cycles to — 16 u64 (128 bytes) per persistent wave, at most 32 waves per CU (8192 waves, 1 MiB, on this
 * device); NULL = off.  The buffer must stay allocated until the pointer has been reset to NULL and the renders
 * that saw it have finished. */
int fr_debug_set_queue_trace(void *d_trace) {
    g_queue_trace.store(static_cast<unsigned long long *>(d_trace));
    return FR_OK;
}

/* test aid: entries per survivor list of the two-pass render (0 = sized from the image); a tiny value makes
 * the lists overflow, which the first pass must absorb */
int fr_debug_set_two_pass_capacity(uint32_t entries_per_list) {
    g_two_pass_list_entries.store(entries_per_list);
    return FR_OK;
}

/* test hook (not part of the reference surface): elementwise device arithmetic over host arrays, so
 * tests can compare the device's roundings with the host's (see include/fractal_hip.h). */
int fr_debug_math(int which, const double *in, double *out, size_t n) {
    if (n == 0) return FR_OK;
    if (!in || !out) return fail(FR_ERR_INVALID_ARGUMENT, "NULL array");
    LifeShared ls;
    Ctx *ctx;
    int rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    rc = ctx->reserve(ctx->z, 2 * n * sizeof(double));
    if (rc != FR_OK) return rc;
    double *d_in = static_cast<double *>(ctx->z.ptr), *d_out = d_in + n;
    if (which == 4) { /* colour-filter scan: in[0], in[1] = first and last f32 bit pattern; out[0] = worst error */
        if (n < 2) return fail(FR_ERR_INVALID_ARGUMENT, "which == 4 needs n >= 2");
        HIP_TRY(fr_launch_nu_scan((uint32_t)in[0], (uint32_t)in[1], d_out, ctx->stream));
        HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return FR_OK;
    }
    if (which == 6) { /* packed-cast scan: in[0], in[1] = first and last f32 bit pattern; out[0] = mismatches */
        if (n < 2) return fail(FR_ERR_INVALID_ARGUMENT, "which == 6 needs n >= 2");
        HIP_TRY(fr_launch_cast_scan((uint32_t)in[0], (uint32_t)in[1], d_out, ctx->stream));
        unsigned long long bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, d_out, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        out[0] = (double)bad;
        return FR_OK;
    }
    HIP_TRY(hipMemcpyAsync(d_in, in, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(fr_launch_math_probe(which, d_in, d_out, n, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

} /* extern "C" */
