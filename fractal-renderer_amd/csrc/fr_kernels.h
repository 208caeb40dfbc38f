/*
 * fr_kernels.h — internal (C++) interface between the C ABI (fr_api.hip) and the gfx950 kernels
 * (fr_kernels.hip).  Not installed; the public boundary is include/fractal_hip.h.
 */
#ifndef FR_KERNELS_H
#define FR_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* Kernel arguments: calc::Config's hot-path fields (calc/src/lib.rs:21-37) plus the mapping from
 * the launch's local pixel grid to image coordinates. */
struct fr_kparams {
    uint32_t algo;
    uint32_t width, height; /* of the IMAGE (used by the coordinate map only) */
    uint32_t iterations;
    double limit, stable_limit;
    double pos_re, pos_im, scale_re, scale_im;
    double exposure;
    double julia_re, julia_im;
    uint32_t inside, smooth;
    uint32_t prim[3], sec[3]; /* stored r, g, b fields of primary_color / secondary_color */
    /* colour-map constants prepared by the host so that they arrive as scalars instead of being
     * converted by every wave: the colour fields and config.iterations as f64 (exact), and the exact
     * reciprocal of iterations when it is a power of two (x / 2^k == x * 2^-k), else 0 */
    double prim_f[3], sec_f[3];
    double iterations_f64, inv_iterations;
    /* local grid: ncols x nrows pixels; local (cx, r) is image pixel
     *   x = x_first + cx * x_stride
     *   y = y_first + (r / block_rows) * y_stride + r % block_rows                     */
    uint32_t ncols, nrows;
    uint32_t x_first, x_stride;
    uint32_t block_rows, y_first, y_stride;
    /* RGB output addressing: 0 = packed (local row r at 3*ncols*r), 1 = in place (local row r at its
     * IMAGE row y: the destination is the whole image).  In place needs block_rows % 8 == 0 so that
     * an 8-row tile never straddles two blocks. */
    uint32_t out_in_place;
    /* RGB output pixel format: 0 = packed r,g,b (the reference's Vec<RGB>), 1 = r,g,b,255 (RGBA8, what
     * the GUI converts the image to before uploading it, src/gui.rs:71-72) */
    uint32_t out_rgba;
    /* orbit-loop plan chosen by the host (fr_api.hip: plan_loop): 0 = unscaled loop, escape
     * check every iteration; 4 / 2 = scaled loop, escape check every 4th / 2nd iteration while
     * every live lane of the wave has |z|^2 <= skip_t (see fr_kernels.hip). */
    uint32_t loop_mode;
    double skip_t;
    /* Speculative long blocks (fr_kernels.hip: FR_SC_SPEC_BODY, FR_ORBIT_ASM, FR_FB_SPEC_LOOP): a wave in which no lane has
     * passed skip_t (loop_mode 4) / escaped (the unscaled loop) for this many iterations goes on in blocks of FR_SPEC_M
     * unchecked iterations that keep their start state in a second register set; 0 = never (the host could not prove that an
     * escape inside a block is visible at its end — fr_api.hip: plan_loop — was asked not to, or the view's statistics say
     * that nothing stays).  The two-iteration scaled blocks (loop_mode 2) do not speculate. */
    uint32_t loop_spec;
    uint32_t first_no_spec; /* 1: the two-pass render's first kernel runs in its plain form whatever loop_spec says (the view's
                             * statistics: nothing stays in its tiles; fr_api.hip: decide_from_sample) */
    /* smooth == false only: palette[i] = packed r | g << 8 | b << 16 of an OUTSIDE pixel whose
     * escape index is i (0 .. iterations), built by fr_launch_palette; NULL = compute per pixel */
    const uint32_t *palette;
    uint32_t palette_entries;
    /* refilling kernel: an episode may end early once `refill_quit16`/16 of its running lanes have
     * finished and `refill_minrun` iterations were done (see fr_kernels.hip) */
    uint32_t refill_minrun, refill_quit16;
    /* work-queue kernel: while more pixels wait, an episode ends once `queue_want` lanes have finished and
     * `queue_minrun` iterations were done */
    uint32_t queue_minrun, queue_want;
    /* exact periodicity shortcut (refilling kernel, scaled loops): an orbit found bitwise back at an
     * earlier state is fast-forwarded to the cap instead of being iterated there; 0 = off */
    uint32_t cycle_shortcut;
    /* smooth colouring: bracket log2(log2(sqrt(dist))/2) with the hardware f32 log first and take the
     * f64 software log2 only for pixels whose bracket straddles a byte boundary; 0 = always f64 */
    uint32_t colour_filter;
    /* work-queue kernel: FR_SURV_QUEUES device counters FR_SURV_COUNT_STRIDE words apart, zeroed on the launch
     * stream, through which its persistent waves claim patches / chunks; NULL = that kernel is not available to
     * this launch */
    uint32_t *work_counter;
    double filt_k;    /* exposure / iterations (any rounding) */
    double filt_d[3]; /* per stored colour field: |field * filt_k| * FR_NU_BRACKET * (1 + 2^-20) */
    /* the filter's f32 first stage: on only when iterations < 2^24 and 2^-60 <= |filt_k| <= 2^60 */
    uint32_t colour_filter32;
    float filt_k32;    /* (float)filt_k */
    float filt_c32;    /* |filt_k| * FR_NU_BRACKET * (1 + 2^-9), rounded up: the bracket's half-width per unit of colour field
                        * (field * filt_c32 >= filt_d32[field's index]); the first pass's form of the f32 stage */
    float filt_d32[3]; /* the same half-widths, times (1 + 2^-10), rounded up to f32 */
    float prim32[3];   /* the stored colour fields as f32 (exact) */
    /* f32 renders: max(stable_limit, 2) * (1 + 2^-20) rounded up — an f32 squared distance at or above it proves
     * dist > stable_limit and dist >= 2 (fr_kernels.hip: colour_pixel); +inf = never take that shortcut */
    float filt_lo32;
    /* two-pass rendering (fr_kernels.hip, "first pass + survivor list"): the first pass runs every pixel
     * `first_cap` iterations and appends the orbits still going to one of FR_SURV_QUEUES lists in device
     * memory; the second pass (the work-queue kernel, drawing from those lists) finishes them.
     * first_cap == 0: not in use.  List q holds entries [q * surv_sub_capacity, (q + 1) * surv_sub_capacity);
     * its counter (zeroed on the launch stream by the caller) is surv_counts[q * FR_SURV_COUNT_STRIDE] and may
     * run past the capacity: what did not fit was finished by the first pass itself. */
    uint32_t first_cap;    /* length of a first-pass episode */
    uint32_t first_keep;   /* a tile stays in the first pass while at least this many of its lanes are running */
    uint32_t two_pass_cap; /* the caller's wish for first_cap (0 = the default) */
    uint32_t first_only;   /* 1: the first pass keeps every tile to its end (no lists, no second pass) */
    uint32_t second_v1;    /* 1: the survivor lists are drained by round 2's kernel, escape_queue_kernel<.., 1> (comparison only) */
    uint32_t first_one_band; /* 1: one 7-tile strip per workgroup whatever the launch size (views of long orbits: workgroups of
                              * 28 tiles differ too much in cost to balance) */
    uint32_t strip_tiles; /* strip length asked for by the caller: first pass 4 (GUI-sized launches) else 7; strip kernel
                           * (tile 0, RGB) 1 / 2 / 4 / 7, 0 = by launch size */
    uint32_t debug_ablate;  /* measurement aid, never set by the product path (FR_DEBUG_ABLATE; WRONG IMAGES): bit 0 = the first pass
                             * claims its list slots but does not store the entries, bit 1 = the second pass is not launched */
    uint32_t surv_sub_capacity;
    void *surv_z;           /* T[2] per entry: the position after first_cap iterations */
    uint32_t *surv_pos;     /* uint32[2] per entry: output column, output row */
    uint32_t *surv_cnt;     /* iterations the entry's pixel has done */
    void *surv_c;           /* T[2] per entry: c (Mandelbrot; unused for Julia, whose c is julia_set) */
    uint32_t *surv_counts;
};

constexpr uint32_t FR_SURV_QUEUES = 64;       /* = the wave size: the second pass scans the counters one per lane */
constexpr uint32_t FR_SURV_COUNT_STRIDE = 32; /* words between counters: one 128-byte line each */
constexpr uint32_t FR_SURV_CHUNK = 64;        /* entries a second-pass wave claims at a time: one per lane */

/* half-width of the colour filter's bracket around its f32 estimate of nu (fr_kernels.hip) */
constexpr double FR_NU_BRACKET = 0x1p-18;

enum fr_out_mode {
    FR_OUT_RGB = 0,    /* packed r,g,b at 3*(r*ncols + cx)                    */
    FR_OUT_ESCAPE = 1, /* z (2 doubles) and/or escape index per local pixel   */
    FR_OUT_COUNT = 2   /* sum of executed iterations into one uint64          */
};

/* MODE COUNT accumulates into this many uint64 partial sums (fr_kout::count points at them) */
constexpr uint32_t FR_COUNT_SLOTS = 512;

struct fr_kout {
    uint8_t *rgb;
    double *z;
    uint32_t *iters;
    unsigned long long *count;
    /* tuning aid (fr_debug_set_queue_trace): the work-queue kernel's waves record 16 u64 (128 bytes) each here —
     * start, end (100 MHz realtime ticks), patches opened, episodes, colour passes, iterations run, cycles per
     * phase (open | refill << 32, loop | retire << 32, finish), 7 unused */
    unsigned long long *trace;
};

/* tile = kernel-variant selector (see fr_set_tile in include/fractal_hip.h); 0 = default.
 * *kernel_name (may be NULL) receives a static string naming the kernel that was launched. */
hipError_t fr_launch_escape(const fr_kparams &p, int precision, int mode, const fr_kout &out, int tile,
                            hipStream_t stream, const char **kernel_name);

/* Would fr_launch_escape(p, ..., FR_OUT_RGB, ..., tile) pick the work-queue kernel if p.work_counter were set?
 * (The caller then lends a counter and zeroes it on the launch stream.) */
bool fr_wants_work_queue(const fr_kparams &p, int tile);

/* Would fr_launch_escape(p, ..., FR_OUT_RGB, ..., tile) render in two passes if the survivor lists were set?
 * (The caller then lends them — fr_two_pass_bytes() says how large for `entries` per list — and zeroes
 * surv_counts and work_counter on the launch stream.)  Sets p.first_cap when it answers yes. */
bool fr_wants_two_pass(fr_kparams &p, int precision, int tile, int hint = -1); /* hint (tile 0 only): -1 none, 0 strips, 1 two passes, 2 the first pass alone */
struct fr_two_pass_layout {
    size_t z_off, pos_off, cnt_off, c_off, counts_off, total; /* byte offsets into one allocation */
};
fr_two_pass_layout fr_two_pass_bytes(const fr_kparams &p, int precision, uint32_t sub_capacity);

/* View sample (fr_kernels.hip: view_sample_kernel): side x side 8x8 tiles of the launch through the plain loop capped
 * at cap_s; the last wave writes SEVEN totals — {executed iterations, 64 x sum of per-tile maxima, tiles, lanes at the cap,
 * lanes the first pass's episode schedule (`episode` iterations, doubling from the ninth on) would hand over (tiles with
 * fewer than `keep` lanes left), lane-iterations wasted by finishing those in place, iterations the handed-over lanes still
 * have to run} — to result[0..6] (host-mapped), then `tag` to result[7] (release: a host polling for the tag reads complete
 * totals), and zeroes `counters` (8 device words, zero before the first use; one sample at a time per counter set). */
hipError_t fr_launch_view_sample(const fr_kparams &p, int precision, uint32_t side, uint32_t cap_s, uint32_t episode, uint32_t keep,
                                 unsigned long long *counters, unsigned long long *result, unsigned long long tag,
                                 hipStream_t stream);

/* Device -> pinned host copy BY A KERNEL, with its own completion flag (fr_host.hip: the staged road of GUI-sized frames):
 * `bytes` from `src` (device memory) to `dst` (the device address of pinned, mapped host memory; src and dst congruent
 * modulo 16), 16 bytes per store.  Every workgroup fences its stores at system scope and counts itself in on `counter` (a
 * device word, zero between launches); the last one stores `seq` to `flag` (pinned, mapped host memory) — a host that
 * polls the flag for `seq` finds the bytes there.  No runtime copy machinery, no event: two launches a band. */
hipError_t fr_launch_copy_out(const void *src, void *dst, size_t bytes, unsigned int *counter, unsigned long long *flag,
                              unsigned long long seq, hipStream_t stream);

/* Largest palette the render kernel will stage in LDS (entries of 4 bytes): beyond it the LDS
 * footprint per one-wave workgroup would cut occupancy, and the colour is computed per pixel. */
constexpr uint32_t FR_MAX_PALETTE_ENTRIES = 1280;

/* palette[i], i = 0 .. p.iterations, for smooth == false (see fr_kparams::palette) */
hipError_t fr_launch_palette(const fr_kparams &p, uint32_t *palette, hipStream_t stream);

/* colour map only (calc/src/lib.rs:214-234) over n stored recursive() results: z (re, im
 * interleaved) and iters -> packed r,g,b.  Device arrays. */
hipError_t fr_launch_colour(const fr_kparams &p, const double *z, const uint32_t *iters, size_t n, uint8_t *rgb,
                            hipStream_t stream);

/* n independent orbits, device arrays (re, im interleaved) */
hipError_t fr_launch_recursive_batch(uint32_t iterations, const double *start, const double *c, size_t n,
                                     double limit, int precision, double *out_pos, uint32_t *out_iters,
                                     hipStream_t stream);

/* test hooks: elementwise device log2 / sqrt over n doubles */
hipError_t fr_launch_math_probe(int which, const double *in, double *out, size_t n, hipStream_t stream);

/* test hook: largest |filter bracket centre - f64 nu| over every f32 bit pattern in [lo, hi] -> out[0] (device) */
hipError_t fr_launch_nu_scan(uint32_t lo_bits, uint32_t hi_bits, double *out, hipStream_t stream);

/* test hook: the number of f32 bit patterns in [lo, hi] on which the packed `as u8` cast differs from the plain
 * one -> the 8 bytes at out (device) as a uint64 */
hipError_t fr_launch_cast_scan(uint32_t lo_bits, uint32_t hi_bits, double *out, hipStream_t stream);

#endif
