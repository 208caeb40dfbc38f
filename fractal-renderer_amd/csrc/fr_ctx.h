/*
 * fr_ctx.h — internal (C++) state shared by the translation units of the C ABI:
 *   fr_api.hip    the single-device entry points (get_image / get_recursive_pixel / recursive)
 *   fr_host.hip   getting a finished image from HBM into the caller's Vec<RGB>-shaped host buffer
 *   fr_multi.hip  get_image across a set of devices from ONE process (src/lib.rs:253 has one caller)
 * Not installed; the public boundary is include/fractal_hip.h.
 */
#ifndef FR_CTX_H
#define FR_CTX_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fractal_hip.h"
#include "fr_kernels.h"

namespace fr {

/* ---- errors: code + thread-local message (fr_last_error) ----------------------------------- */
int fail(int code, const char *what);
int fail(int code, const std::string &what);
int fail_hip(hipError_t e, const char *what);
const std::string &last_error();

#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return ::fr::fail_hip(e_, #expr); \
    } while (0)

/* ---- implementation selectors of one render (fr_render_opts resolved against the defaults) --- */
struct Opts {
    int tile = 0;
    int loop_mode = -1;
    int palette = 1;
    int cycle_shortcut = 0;
    int refill_minrun = -1, refill_quit16 = -1; /* -1: each kernel's own default */
    int colour_filter = 1;
    /* the default dispatch's choice of kernel, when the caller has made it for a whole image that it renders in several
     * launches (bands of the host path, chunks of the multi-device path): -2 = decide per launch (fr_api.hip:
     * choose_kernel), else choose_kernel's answer (-1 none, 0 strips, 1 two passes, 2 first pass alone) and its
     * one-strip-per-workgroup flag */
    int kernel_hint = -2;
    bool one_band = false;
    bool no_spec = false; /* with kernel_hint >= 0: the view's statistics say nothing stays — no speculative blocks (fr_api.hip: decide_from_sample) */
    uint32_t strip_tiles = 0; /* with kernel_hint >= 0: the strip length the view's statistics call for (0 = by launch size) */
    /* a non-blocking view sample the caller still has to post BEHIND its render (Ctx::post_sample): slot index, -1 none */
    int pending_sample = -1;
};
Opts default_opts();                             /* what the fr_set_* calls have set */
int resolve_opts(const fr_render_opts *o, Opts &out); /* NULL = defaults; validates */

/* ---- one logical device: a HIP device + the library's streams and scratch on it -------------- */
struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
};

/* Palette scratch for smooth == false renders: a small ring of device buffers owned by the context
 * (no allocation on the render path, usable from any stream of that device).  A slot is handed out
 * only after the event recorded behind its last user has completed. */
struct PaletteSlot {
    uint32_t *dev = nullptr; /* FR_MAX_PALETTE_ENTRIES palette words, then the work-queue kernel's claim counters */
    hipEvent_t done = nullptr;
    bool pending = false; /* `done` was recorded and not yet waited for */
    bool busy = false;    /* a thread is between acquire and its event record */
};
constexpr int kPaletteSlots = 16;

/* Survivor lists of the two-pass render (fr_kernels.hip): a ring of three buffers cut from ONE device allocation
 * (Ctx::surv_block), made with the context for frames up to 3840 x 2160 and re-made only for a launch that needs more
 * than a slot holds — the one allocation the device-pointer entry points can block on (a larger two-pass frame than
 * any before); handed out like the palette slots. */
struct SurvSlot {
    void *dev = nullptr;
    hipEvent_t done = nullptr;
    bool pending = false, busy = false;
};
constexpr int kSurvSlots = 3;

struct Ctx {
    int hip_device = -1;
    std::mutex mu; /* serialises the host-buffer entry points (they share streams + scratch) */
    hipStream_t stream = nullptr;      /* kernels of the host-buffer entry points */
    hipStream_t stream2 = nullptr;     /* band / chunk kernels alternate between stream and stream2 */
    hipStream_t copy_stream = nullptr; /* D2H / peer copies, overlapped with the next band's kernel */
    std::vector<hipEvent_t> events;
    Scratch rgb, z, iters, misc;
    PaletteSlot palette_slots[kPaletteSlots];
    SurvSlot surv_slots[kSurvSlots];
    void *surv_block = nullptr; /* kSurvSlots x surv_slot_cap bytes */
    size_t surv_slot_cap = 0;
    bool surv_regrowing = false; /* a thread is re-making the ring with palette_mu dropped (acquire_surv) */
    uint32_t *palette_block = nullptr; /* kPaletteSlots slots, allocated with the context */
    /* view samples (fr_api.hip: choose_kernel).  Large launches: a BLOCKING sample on aux_stream — a stream of its own: the
     * sample must not wait behind whatever the caller has queued on its stream.  GUI-sized launches: a NON-BLOCKING one on
     * aux2_stream, behind the render of the frame it was asked for (an event of the caller's stream), read by the NEXT
     * frame of the same view.  Each has its own eight device counters; every remembered view its own eight result words
     * (host-mapped; the kernel writes the view's key into the eighth LAST), plus one set for fr_debug_sample_view. */
    hipStream_t aux_stream = nullptr, aux2_stream = nullptr;
    unsigned long long *sample_counters = nullptr; /* device: 2 x 8 words */
    unsigned long long *sample_result = nullptr;   /* pinned host memory, mapped: (kViewChoices + 1) x 8 words */
    std::mutex sample_mu; /* the view table AND the order of launches on the two sample streams */
    struct ViewChoice {
        uint64_t key = 0;
        int state = 0; /* 0 free, 1 a sample is (about to be) in flight, 2 decided */
        int two_pass = -1; /* -1 no opinion, 0 strips, 1 two passes, 2 the first pass alone */
        bool one_band = false;
        bool no_spec = false;
        uint32_t strip_tiles = 0;
        double lane_fraction = 0.0;
        fr_kparams grid; /* state 1: what the sample is launched with */
        int precision = 0;
        hipEvent_t after = nullptr; /* recorded on the caller's stream behind the render the sample follows */
    };
    static constexpr int kViewChoices = 32; /* (a rank of a multi-GPU run renders its share in up to a dozen chunk launches per image, each its own view) */
    ViewChoice view_choices[kViewChoices];
    unsigned view_next = 0;
    /* enqueue slot `idx`'s sample behind everything `stream` holds now; never blocks, never fails the render (a sample that
     * cannot be posted frees its slot) */
    void post_sample(int idx, hipStream_t stream);
    std::mutex palette_mu; /* guards both rings */
    std::condition_variable slot_cv; /* a slot of either ring was released */
    unsigned palette_next = 0, surv_next = 0;

    /* GUI-sized host-buffer renders (fr_host.hip: host_render_staged): a pinned staging buffer of the library's own — the
     * device never maps, pins or DMAs into the CALLER's pages for frames up to 3840 x 2160 RGBA — and a few copy threads */
    void *stage = nullptr;       /* pinned host memory (hipHostMalloc, mapped) */
    void *stage_dev = nullptr;   /* its device address: bands leave HBM through a copy KERNEL (fr_launch_copy_out) */
    size_t stage_cap = 0;
    unsigned int *stage_counters = nullptr;     /* device: one per band in flight (4) */
    unsigned long long *stage_flags = nullptr;  /* pinned, mapped: the bands' completion flags (4) + their device address */
    unsigned long long *stage_flags_dev = nullptr;
    unsigned long long stage_seq = 0;           /* flags carry the sequence number of the band that set them */
    struct CopyPool *copy_pool = nullptr;
    int reserve_stage(size_t bytes);

    int create(int device); /* hipSetDevice + streams; the calling thread stays on `device` */
    void destroy();         /* frees everything (caller made sure nothing is in flight) */
    int reserve(Scratch &s, size_t bytes);
    int event(size_t k, hipEvent_t *out); /* k-th reusable event (created on demand) */
    int acquire_palette(PaletteSlot **out);
    void release_palette(PaletteSlot *slot, hipStream_t stream); /* record + make reusable */
    int acquire_surv(size_t bytes, SurvSlot **out);
    void release_surv(SurvSlot *slot, hipStream_t stream);
};

/* calc::Config -> kernel arguments (the local grid is filled in by the caller) and the loop plan */
void fill_params(const fr_config *cfg, const Opts &o, fr_kparams &p);
void plan_loop(const fr_config *cfg, int precision, const Opts &o, fr_kparams &p);

/* device-pointer render of the local grid set in `p`; `ctx` lends the palette slot (it must live
 * on the device `stream` belongs to).  No host synchronisation. */
int render_device(Ctx &ctx, const fr_config *cfg, fr_kparams &p, int precision, const Opts &o, void *d_out,
                  hipStream_t stream);

/* row-block-cyclic chunk: blocks first_block, first_block + stride, ... (at most max_blocks; 0 = all) */
int render_block_cyclic(Ctx &ctx, const fr_config *cfg, int precision, const Opts &o, uint32_t block_rows,
                        uint32_t first_block, uint32_t block_stride, uint32_t max_blocks, int dest_is_image,
                        void *d_out, size_t out_len, hipStream_t stream, uint64_t *rows_written);

int check_precision(int precision);

/* choose_kernel for rows [y0, y1) of the image as ONE launch, recorded in `o` (tile 0 only): callers that render those
 * rows in several launches then sample the view once, not once per launch.  The calling thread must be on ctx's device. */
void decide_kernel(Ctx &ctx, const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, Opts &o, hipStream_t stream,
                   bool allow_async);

/* the primary context (what fr_init selected); locks and lazily creates it.  Callers hold
 * `life_shared()` while they use it. */
int primary(Ctx **out);

/* Lifetime lock: entry points that enqueue on library state hold it shared; fr_init (device switch),
 * fr_init_devices and fr_shutdown hold it exclusively, so state is never torn down under a call. */
struct LifeShared {
    LifeShared();
    ~LifeShared();
};
struct LifeExclusive {
    LifeExclusive();
    ~LifeExclusive();
};

/* ---- host buffers (fr_host.hip) ------------------------------------------------------------- */

/* Walks a host buffer in page-aligned chunks of 64 MiB, making each DMA-able in turn: where the pages
 * do not exist yet (a fresh Vec) a background thread faults them in ahead of the walk (huge-page hint,
 * several threads), then the chunk is pinned with hipHostRegister.  `portable` = for every device. */
class ChunkPinner {
  public:
    /* byte_order (optional): image byte offsets in the order the caller will need them, so that the first
     * touch runs ahead in that order */
    ChunkPinner(uint8_t *out, size_t need, bool portable, const std::vector<size_t> *byte_order = nullptr);
    ~ChunkPinner();
    size_t chunks() const { return bounds_.size() - 1; }
    size_t bound(size_t k) const { return bounds_[k]; } /* chunk k = bytes [bound(k), bound(k + 1)) */
    size_t chunk_of(size_t byte) const;
    /* waits for the chunk's first touch, pins it once; false: it cannot be pinned (plain copy).  Different chunks
     * may be pinned from different threads at the same time (fr_multi.hip's pool); one chunk by one thread. */
    bool pin(size_t k);
    /* the chunks in address order: pins the next one, bytes [a, b); returns false when the walk is over */
    bool next(size_t &a, size_t &b, bool &pinned);
    void release(); /* unpin everything (the caller drained its streams first) */
    /* where the chunk starting at byte `a` ends: a pure function of (out, need, a), so that other threads
     * can split their copies at the same places (one DMA must not span two pins) */
    static size_t chunk_end(const uint8_t *out, size_t need, size_t a);
    double t_touch = 0.0, t_reg = 0.0; /* ms spent waiting for the toucher / in hipHostRegister */

  private:
    uint8_t *out_;
    size_t need_, pos_ = 0;
    unsigned flags_;
    std::vector<size_t> bounds_;
    std::unique_ptr<std::atomic<int>[]> state_; /* 0: pages may not exist yet, 1: touched */
    std::vector<char> pinned_;                  /* 0: not tried, 1: pinned, 2: cannot be pinned */
    std::thread toucher_;
    std::mutex reg_mu_; /* regs_ and the two timers, when chunks are pinned from several threads */
    std::vector<uint8_t *> regs_;
};
void prefault(void *ptr, size_t len);
void destroy_copy_pool(CopyPool *pool); /* stops and joins the threads (fr_host.hip) */

/* per-thread kernel timing (fr_set_profiling / fr_last_kernel_ms) */
struct Profiling {
    bool enabled = false;
    bool have = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char *kernel = "";
    ~Profiling();
};
Profiling &profiling();

/* multi-device teardown hook, called by fr_shutdown / fr_init_devices with the exclusive lock held */
void multi_shutdown_locked();

}  // namespace fr

/* shared body of the host-buffer row renders (fr_host.hip) */
int fr_host_render_rows(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out, size_t out_len,
                        unsigned bytes_per_pixel, const fr_render_opts *opts);

#endif
