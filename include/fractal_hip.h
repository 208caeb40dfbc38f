/*
 * fractal_hip.h — C ABI of libfractal_hip.so, the MI355X (gfx950) implementation of
 * Icelk/fractal-renderer's escape-time hot path.
 *
 * The reference is pure Rust and has NO FFI of its own; the seam this library fills is the
 * three public functions + four types of the `calc` crate and the host library's get_image:
 *
 *   pub fn recursive(iterations, start, c, limit) -> (Imaginary, u32)   calc/src/lib.rs:245
 *   pub fn get_recursive_pixel(&Config, x, y) -> RGB                    calc/src/lib.rs:199
 *   pub fn get_image(&Config) -> Vec<RGB>   (Mandelbrot | Julia arm)    src/lib.rs:253-270
 *   Config / Imaginary / RGB / Algo                                     calc/src/lib.rs:21-37,79-82,121-125,150-154
 *
 * Every entry point below names the reference interface it replaces.  INTEGRATION.md shows the
 * Rust `extern "C"` block and the patch to src/lib.rs:253-270 a maintainer would add.
 *
 * Conventions
 *   - Plain pointers and sizes only.  The CALLER owns every buffer; the library never frees or
 *     keeps a caller pointer past the call.
 *   - Every function returns FR_OK (0) or an fr_status error code and never aborts the process;
 *     fr_last_error() returns a thread-local message for the last failing call on this thread.
 *     (The reference's get_image is infallible; the Rust shim maps an error to a panic or to its
 *     own CPU arm — INTEGRATION.md.)
 *   - Re-entrant: may be called concurrently from several host threads with different configs,
 *     as the GUI's render thread and screenshot thread do (src/gui.rs:56-60, 322-326).
 *   - Output pixel layout is the reference's Vec<RGB>: row-major, tightly packed bytes r,g,b;
 *     pixel (x, y) of the image lives at byte 3*(y*width + x).
 *   - There is no CPU fallback: without a usable gfx950 device every compute call fails with
 *     FR_ERR_NO_DEVICE.
 */
#ifndef FRACTAL_HIP_H
#define FRACTAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_ABI_VERSION 3

typedef enum fr_status {
    FR_OK = 0,
    FR_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, y0 > y1, y1 > height, ... */
    FR_ERR_BUFFER_TOO_SMALL = 2, /* out_len < bytes the call must write */
    FR_ERR_NO_DEVICE = 3,        /* no HIP device / not initialised and auto-init failed */
    FR_ERR_HIP = 4,              /* a HIP runtime call failed; see fr_last_error() */
    FR_ERR_UNSUPPORTED_ALGO = 5  /* reserved; Algo::BarnsleyFern is NOT an error (renders black) */
} fr_status;

/* enum Algo — calc/src/lib.rs:150-154, in declaration order */
typedef enum fr_algo {
    FR_ALGO_MANDELBROT = 0,
    FR_ALGO_BARNSLEY_FERN = 1, /* on this path: every pixel RGB::BLACK (calc/src/lib.rs:211) */
    FR_ALGO_JULIA = 2
} fr_algo;

/* struct Imaginary — calc/src/lib.rs:79-82 */
typedef struct fr_imaginary {
    double re;
    double im;
} fr_imaginary;

/* struct RGB — calc/src/lib.rs:121-125.  These are the STORED fields.  RGB::new(r, b, g) takes
 * blue second (calc/src/lib.rs:129-131), so e.g. Config::new's RGB::new(40, 40, 255) is stored
 * {r:40, g:255, b:40}; pass the stored struct verbatim — the library reproduces color_multiply's
 * swap (calc/src/lib.rs:133-139) internally. */
typedef struct fr_rgb {
    uint8_t r;
    uint8_t g;
    uint8_t b;
} fr_rgb;

/* struct Config — calc/src/lib.rs:21-37, field for field (`#[repr(C)]` image; bools as u8,
 * enum as u32).  sizeof == 104. */
typedef struct fr_config {
    uint32_t algo; /* fr_algo */
    uint32_t width;
    uint32_t height;
    uint32_t iterations;
    double limit;        /* escape RADIUS; squared inside recursive() (calc/src/lib.rs:246) */
    double stable_limit; /* compared with the SQUARED distance, un-squared (calc/src/lib.rs:216) */
    fr_imaginary pos;
    fr_imaginary scale; /* per-axis pair, not a complex number */
    double exposure;
    uint8_t inside;
    uint8_t smooth;
    fr_rgb primary_color;
    fr_rgb secondary_color;
    double color_weight; /* fern only; ignored on this path */
    fr_imaginary julia_set;
} fr_config;

/* Arithmetic of the z = z^2 + c loop.  F64 is the reference's (and the only one with a parity
 * claim against it).  F32 is this build's fast path for shallow zooms, defined as: coordinates
 * (calc/src/lib.rs:181-197) in f64, start / c / limit narrowed to f32, recursive() evaluated in
 * f32 with the same operation order, final position widened to f64, colour mapping in f64. */
typedef enum fr_precision { FR_PRECISION_F64 = 0, FR_PRECISION_F32 = 1 } fr_precision;

/* ---- lifetime ---------------------------------------------------------------------------- */

/* Select the HIP device the single-device entry points render on (-1 = keep the current one /
 * device 0) and create the library's state.  Optional: compute calls auto-initialise on device 0.
 * Switching to another device waits for every call in flight and frees the state on the old one. */
int fr_init(int device);
/* Release streams and scratch memory.  Safe to call twice; the library can be re-initialised. */
int fr_shutdown(void);
int fr_device_count(int *count);
/* gfx architecture name of the active device, e.g. "gfx950:sramecc+:xnack-" */
int fr_device_name(char *buf, size_t buf_len);
const char *fr_last_error(void);
int fr_abi_version(void);
/* SHA-256 prefix over the sources and flags this binary was built from (fractal-renderer_amd/build.py) */
const char *fr_build_id(void);

/* Config::new(algo) — calc/src/lib.rs:39-69 */
void fr_config_new(fr_config *cfg, uint32_t algo);

/* ---- get_image — src/lib.rs:253-270 ------------------------------------------------------- */

/* Whole image into a HOST buffer of at least 3*width*height bytes (f64 arithmetic).
 * Replaces `get_image(config)` for Algo::Mandelbrot | Algo::Julia. */
int fr_render_rgb8(const fr_config *cfg, uint8_t *out, size_t out_len);

/* Rows [y0, y1) of the image into a HOST buffer of at least 3*width*(y1-y0) bytes: the unit the
 * reference's rayon loop parallelises over (src/lib.rs:256-264).  y0 == y1 is legal (no-op). */
int fr_render_rows_rgb8(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out,
                        size_t out_len);

/* Same, into DEVICE memory, asynchronously on `hip_stream` (a hipStream_t, NULL = the null
 * stream).  For callers that keep the image in HBM (multi-GPU gather, GUI upload). */
int fr_render_rows_rgb8_device(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1,
                               void *d_out, size_t out_len, void *hip_stream);

/* The same rows as RGBA8 (bytes r,g,b,255; 4*width*(y1-y0) bytes; device pointer 4-byte aligned): what
 * the reference's GUI converts the Vec<RGB> to on the CPU before uploading it (src/gui.rs:71-72). */
int fr_render_rows_rgba8(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out,
                         size_t out_len);
int fr_render_rows_rgba8_device(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                                size_t out_len, void *hip_stream);

/* Row-block-cyclic share of the image for multi-GPU rendering: blocks of `block_rows` rows,
 * this call renders blocks first_block, first_block + block_stride, ... and packs them
 * contiguously into d_out (device memory).  *rows_written (may be NULL) receives the number of
 * rows produced.  fr_block_cyclic_rows() returns that count without rendering. */
int fr_render_block_cyclic_rgb8_device(const fr_config *cfg, int precision, uint32_t block_rows,
                                       uint32_t first_block, uint32_t block_stride, void *d_out,
                                       size_t out_len, void *hip_stream, uint64_t *rows_written);
/* The same with two refinements used by the pipelined multi-GPU gather: at most `max_blocks` blocks
 * (0 = all), and `dest_is_image` != 0 to write every row at its place in the WHOLE image (d_out is
 * then the image base, out_len >= 3*width*height, block_rows % 8 == 0) instead of packing. */
int fr_render_block_cyclic_range_rgb8_device(const fr_config *cfg, int precision, uint32_t block_rows,
                                             uint32_t first_block, uint32_t block_stride, uint32_t max_blocks,
                                             int dest_is_image, void *d_out, size_t out_len, void *hip_stream,
                                             uint64_t *rows_written);

/* Same share into a HOST buffer (packed blocks, 3*width*rows bytes). */
int fr_render_block_cyclic_rgb8(const fr_config *cfg, int precision, uint32_t block_rows, uint32_t first_block,
                                uint32_t block_stride, uint8_t *out, size_t out_len, uint64_t *rows_written);
uint64_t fr_block_cyclic_rows(uint32_t height, uint32_t block_rows, uint32_t first_block,
                              uint32_t block_stride);

/* ---- per-call implementation selectors ------------------------------------------------------- */

/* Everything the fr_set_* calls below select process-wide, as an argument of ONE call, so that two
 * threads (the GUI's render and screenshot threads, src/gui.rs:56-60, 322-326) never share a knob.
 * None of them changes a single output byte.  Fill with fr_render_opts_init() (the process
 * defaults), change what you need, pass to an *_opts entry point; NULL means "the defaults". */
typedef struct fr_render_opts {
    uint32_t size;          /* sizeof(fr_render_opts): lets the struct grow without an ABI break */
    int32_t tile;           /* see fr_set_tile */
    int32_t loop_mode;      /* see fr_set_loop_mode */
    int32_t palette;        /* see fr_set_palette */
    int32_t cycle_shortcut; /* see fr_set_cycle_shortcut */
    int32_t refill_minrun;  /* see fr_set_refill_policy */
    int32_t refill_quit16;
    int32_t colour_filter;  /* see fr_set_colour_filter */
} fr_render_opts;
void fr_render_opts_init(fr_render_opts *opts);

int fr_render_rows_rgb8_device_opts(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                                    size_t out_len, void *hip_stream, const fr_render_opts *opts);
int fr_render_rows_rgba8_device_opts(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                                     size_t out_len, void *hip_stream, const fr_render_opts *opts);
int fr_render_rows_rgb8_opts(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out,
                             size_t out_len, const fr_render_opts *opts);
int fr_render_block_cyclic_range_rgb8_device_opts(const fr_config *cfg, int precision, uint32_t block_rows,
                                                  uint32_t first_block, uint32_t block_stride, uint32_t max_blocks,
                                                  int dest_is_image, void *d_out, size_t out_len, void *hip_stream,
                                                  uint64_t *rows_written, const fr_render_opts *opts);

/* ---- get_image across several GPUs from ONE process ---------------------------------------- */

/* The reference is one process calling get_image once (src/main.rs:16, src/lib.rs:253); its rayon
 * loop spreads the rows over every core (src/lib.rs:256-258).  The counterpart here spreads row
 * blocks over every device of a set: block b (block_rows rows) is rendered by device b % n — the
 * set's interior sits in the middle rows of the default view, so contiguous bands would be badly
 * unbalanced — by one host thread and one set of streams per device.
 *
 * fr_init_devices: `devices` lists HIP device indices; an index may repeat ("logical devices" that
 * share a GPU: how a one-GPU box exercises the whole path).  Replaces any earlier set.  n <= 16. */
int fr_init_devices(const int *devices, int n);
int fr_multi_device_count(int *count);

/* Whole image into the caller's HOST buffer (>= 3*width*height bytes): every device DMAs each of its
 * finished row blocks straight to the block's final place over its own PCIe link while it renders
 * the next ones.  This is what a multi-GPU get_image costs its caller.  block_rows = 0: default (256). */
int fr_render_rgb8_multi(const fr_config *cfg, int precision, uint32_t block_rows, uint8_t *out, size_t out_len);

/* Whole image gathered into DEVICE memory of the set's first device (d_out, >= 3*width*height bytes,
 * allocated by the caller on that device): the first device renders its blocks in place, the others
 * send each finished block to its place over xGMI while they render the next ones —
 *   FR_GATHER_PEER_COPY  peer-to-peer DMA (hipMemcpyPeerAsync), works for repeated device indices too;
 *   FR_GATHER_RCCL       grouped ncclSend / ncclRecv on a communicator from ncclCommInitAll
 *                        (librccl is loaded on first use; needs distinct devices). */
/* The call renders on streams of the library's own and returns when the image is complete: work the CALLER has queued on
 * d_out on a stream of its own (a fill, an earlier reader) must have finished before the call. */
typedef enum fr_gather { FR_GATHER_PEER_COPY = 0, FR_GATHER_RCCL = 1 } fr_gather;
int fr_render_rgb8_multi_device(const fr_config *cfg, int precision, uint32_t block_rows, int gather, void *d_out,
                                size_t out_len);

/* Who renders the sink's row blocks.  The set's first device is both a renderer and the sink of the gather; by default it
 * renders all of its blocks (plain cyclic dealing, q = 1).  q = 2 / 4: it keeps every 2nd / 4th of them and the rest are dealt
 * round-robin to the other devices; q = 0: it renders nothing and only receives.  Same bytes; a tuning knob for the first
 * real multi-GPU run (bench.py --root-share, which prints where every device's time went).  Process-wide. */
int fr_set_multi_root_share(int q);

/* Timing of the calling thread's last multi-device render: per device, the time its render kernels
 * took (HIP events, summed over its chunks) and the host-side wall time of the whole call. */
#define FR_MAX_DEVICES 16
typedef struct fr_multi_stats {
    uint32_t n_devices;
    uint32_t kernels[FR_MAX_DEVICES];   /* render kernels launched on each device */
    float kernel_ms[FR_MAX_DEVICES];    /* their summed duration */
    uint64_t rows[FR_MAX_DEVICES];      /* rows each device rendered */
    double wall_ms;                     /* the call, entry to return */
    /* ABI 3: where each device's time went, for the first real multi-GPU run to be read against DESIGN.md 4's prediction */
    float transfer_span_ms[FR_MAX_DEVICES]; /* device time from its first transfer (DMA / ncclSend; the sink: ncclRecv) being
                                             * ready to start to its last one done; 0 for a device that moves nothing */
    double job_ms[FR_MAX_DEVICES];          /* host wall time of the device's whole job: first launch to streams drained */
    uint64_t bytes_moved[FR_MAX_DEVICES];   /* bytes the device sent to the sink (host buffer or first device's HBM) */
} fr_multi_stats;
int fr_multi_last_stats(fr_multi_stats *stats);

/* Test hook: one grouped self send/recv of `bytes` bytes through RCCL on the set's first device
 * (loads librccl, creates the communicator) — the only RCCL traffic a one-GPU box can carry. */
int fr_debug_rccl_selftest(size_t bytes);
/* Test hook: can librccl be loaded and does it export what the RCCL gather needs?  Touches no device (usable
 * without a GPU).  The environment variable FR_RCCL_LIBRARY, when set, names the library to load instead of the
 * default search — an unloadable name must yield FR_ERR_HIP and a message, never a crash. */
int fr_debug_rccl_probe(void);
/* Test hook: logical device `device_index` of the set fails before its chunk `chunk` of the NEXT multi-device
 * render (once; -1 disarms).  The render must return an error with every stream drained — no hang, no DMA
 * left in flight into the caller's buffer — and the render after it must succeed. */
int fr_debug_inject_multi_failure(int device_index, int chunk);

/* Optional, for callers that render into the SAME host buffer again and again (a GUI's frame buffer,
 * src/gui.rs:56-82): pin it once.  fr_render_rgb8 / fr_render_rgb8_multi make every large host buffer they are
 * handed DMA-able for the duration of the call (hipHostRegister, ~0.9 ms per 64 MiB) and release it before they
 * return; a buffer pinned through this call is found already registered and that cost disappears.  The library
 * does NOT cache registrations by itself: get_image returns a fresh Vec each call (src/lib.rs:266-267) and a
 * registration outliving its allocation would pin — and later DMA into — memory that belongs to someone else.
 * The caller must unpin before freeing.  Portable (valid on every device of the process). */
int fr_pin_host_buffer(void *ptr, size_t len);
int fr_unpin_host_buffer(void *ptr);

/* ---- get_image, Algo::BarnsleyFern arm — src/lib.rs:271-319, fern() :417-463 ------------------- */

/* The chaos game on the GPU.  The reference gives every rayon thread an image filled with secondary_color
 * and iterations / threads points (each point darkens the pixel it falls on: Image::subtract_pixel,
 * src/lib.rs:383-401), then "reduces" the images with combine_images — which adds a into b and returns a
 * (src/lib.rs:275-284, 305-316), so ONE thread's image comes back.  `threads` is that rayon thread count
 * (>= 1).  The reference's RNG is seeded from entropy (src/lib.rs:428): its output is a sample of a
 * distribution, and so is this one — drawn with Philox4x32-10 keyed by `seed` over `walkers` parallel
 * orbits (0 = chosen from the point count).  Deterministic for a given (seed, walkers); bit-identical to the
 * CPU restatement with the same RNG (oracle/); statistically a single sequential orbit (tests).
 * fr_render_rgb8 itself keeps rendering BarnsleyFern BLACK, as calc::get_recursive_pixel does (:211). */
int fr_render_fern_rgb8(const fr_config *cfg, uint32_t threads, uint64_t seed, uint32_t walkers, uint8_t *out,
                        size_t out_len);

/* ---- get_recursive_pixel — calc/src/lib.rs:199-235 ---------------------------------------- */

int fr_pixel(const fr_config *cfg, uint32_t x, uint32_t y, fr_rgb *out);
int fr_pixel_p(const fr_config *cfg, int precision, uint32_t x, uint32_t y, fr_rgb *out);

/* ---- recursive — calc/src/lib.rs:245-257 --------------------------------------------------- */

/* One orbit: returns the final position and the escape index (== iterations on exhaustion). */
int fr_recursive(uint32_t iterations, fr_imaginary start, fr_imaginary c, double limit,
                 fr_imaginary *out_pos, uint32_t *out_iters);

/* n independent orbits (host arrays): start[k], c[k] -> out_pos[k], out_iters[k]. */
int fr_recursive_batch(uint32_t iterations, const fr_imaginary *start, const fr_imaginary *c, size_t n,
                       double limit, int precision, fr_imaginary *out_pos, uint32_t *out_iters);

/* recursive() results of every pixel of rows [y0, y1) (host arrays, either may be NULL):
 * z_re_im[2k], z_re_im[2k+1] = final position, iters[k] = escape index, k = (y-y0)*width + x. */
int fr_escape_rows(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, double *z_re_im,
                   uint32_t *iters);

/* The colour map alone (calc/src/lib.rs:214-234 + color_multiply) over n stored recursive() results
 * — e.g. the arrays fr_escape_rows returned — into packed r,g,b.  This is what the GUI's exposure,
 * smooth/inside and colour controls need (src/gui.rs:183-203 change only inputs of the colour map):
 * re-colouring without re-iterating.  Uses cfg's iterations, stable_limit, exposure, inside, smooth
 * and colours; host arrays, or device arrays + stream for the _device form. */
int fr_colour_rgb8(const fr_config *cfg, const double *z_re_im, const uint32_t *iters, size_t n, uint8_t *out,
                   size_t out_len);
int fr_colour_rgb8_device(const fr_config *cfg, const void *d_z_re_im, const void *d_iters, size_t n, void *d_out,
                          size_t out_len, void *hip_stream);

/* ---- measurement --------------------------------------------------------------------------- */

/* Exact sum of EXECUTED loop iterations over the pixels (x, y) with x % sx == 0, y % sy == 0 of
 * rows [y0, y1): a pixel that escapes at index i executed i+1, one that exhausts the cap executed
 * `iterations`.  Computed on the device. */
int fr_count_iterations(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint32_t sx,
                        uint32_t sy, uint64_t *total, uint64_t *pixels);

/* When enabled, device-pointer renders record HIP events around the escape+colour kernel on the
 * stream they are launched on; fr_last_kernel_ms() waits for that kernel and returns its
 * duration.  State is per calling thread. */
int fr_set_profiling(int enabled);
int fr_last_kernel_ms(float *ms);
/* Name of the render kernel the calling thread's last device-pointer render launched (profiling on),
 * e.g. "escape_strip_kernel<double, 7 tiles>": bench.py reports what actually ran. */
int fr_last_kernel_name(char *buf, size_t buf_len);

/* Kernel-variant selector for tuning studies and tests; every variant produces the same bytes.
 * 0 = default: strips of 8x8 tiles (one tile per workgroup under 8192 x 4096 pixels, seven from there up) unless the view's
 *     own statistics call for another kernel or strip length — see fr_set_dispatch_sampling; without statistics (the first
 *     frame of a GUI-sized view, sampling off): two passes (11) for Julia images of 4096^2 pixels and more with a cap of
 *     512 and more, strips otherwise;
 * 1, 2, 4 = the strip kernel with a fixed strip length of 1, 2, 4 tiles, 8 = of 7 tiles (the longest);
 * 9 = 7-tile strips with lane refill;
 * 10 = the work-queue kernel (persistent waves drawing 64x32-pixel patches from a device-wide counter, unchecked
 *      blocks of iterations, results finished and coloured 64 at a time; RGB renders of an escape-time algorithm
 *      whose loop plan allows the scaled form — otherwise it acts as 9);
 * 11 = two passes: 7-tile strips run every pixel through episodes of a few
 *      dozen iterations and colour what has escaped, tile by tile; a tile whose running lanes fall under a
 *      threshold hands them — position, iterations done, output position — to lists in device memory, which the
 *      persistent waves of a second kernel then finish.  Same conditions as 10 (otherwise it acts as 9).  The
 *      lists live in a context-owned ring of three buffers cut from one allocation: per entry 20 bytes (f32 Julia),
 *      28 (f64 Julia; f32 Mandelbrot), 44 (f64 Mandelbrot), one entry per eight pixels of the launch (at most 2^28
 *      entries) — C4 in f32: 671 MB per buffer, 2 GB for the ring.  The ring exists from the context's creation on
 *      for frames up to 3840 x 2160 (3 x 48 MiB, best effort) and is re-allocated only for a launch that needs more: the one
 *      allocation a device-pointer render can block on (~15 ms, once per size never seen before; other threads' renders that
 *      need no lists are not held up by it).  A list that is full costs speed only;
 * 12 = two passes with round 2's two kernels (kept for comparisons: tools/c4_ab.py);
 * 13 = the first pass of 11 alone: no tile is handed over (every lane finishes in place), no lists, no second kernel;
 * 14 = 11 with round 2's second-pass kernel behind this round's first pass (kept for comparisons);
 * 15 = 11 with the first pass in 4-tile strips, 16 = 13 in 4-tile strips (GUI-sized launches: four times as many
 *      workgroups to balance over the chip);
 * 6401, 3202, 1604, 808 = the 4-wave-workgroup kernel with a 64x1 / 32x2 / 16x4 / 8x8 per-wave
 * pixel footprint. */
int fr_set_tile(int tile);

/* The default dispatch (tile 0) chooses its kernel from the IMAGE: a sample of 16 x 16 tiles of the launch goes through the
 * plain loop (capped at 4096 iterations; ~20-70 us of device time) and reports the share of pixels still running at the cap
 * (`capped`), the share the two-pass render would hand over to its lists (`handed`), the lane-iterations that finishing those
 * in place would idle away as a share of the work (`waste`) and the mean iteration count.
 *   Launches of 131 072 tiles (4096 x 2048 pixels) and more — the sample is taken in FRONT of the first launch of a view,
 *   on a stream of the library's own, and the calling thread waits for it (~40 us): the ONE step of the device-pointer entry
 *   points that blocks.  Rule: capped >= 0.10 and waste < 0.01 -> strips; else handed >= 0.002 (or >= 4096 handed-over
 *   pixels with >= 128 iterations to go on average) -> two passes; else the first pass alone.
 *   (Launches under 524 288 tiles, 8192 x 4096, use the rule of the next paragraph instead.)
 *   Launches of 4096 .. 131 072 tiles — every frame the reference's GUI asks for (src/gui.rs:56-82) — NEVER block: the first
 *   frame of a view is dispatched by size as described under fr_set_tile, the sample is enqueued BEHIND its render, and the
 *   next frame of the same view is dispatched from the measured numbers.  Rule: capped < 0.001 and either mean < 16 with
 *   handed < 0.002 (orbits of a dozen iterations everywhere) or, from 100 000 tiles up, mean < 32 with waste < 1 -> the first
 *   pass alone (4- or 7-tile strips; 4-tile strips of the strip kernel for a Julia constant the scaled loop may not use);
 *   handed >= 0.05 and waste >= 2 (a Julia dust) from 60 000 tiles up in f64, 200 000 in f32 -> two passes; everything else
 *   -> one-tile strips.
 * A view is identified by the fields that determine orbits (algo, width, height, iterations, limit, pos, scale, julia_set),
 * the launch's rows and the precision: changing colours, exposure, smooth or inside keeps it.  The last 32 views are
 * remembered.  No sample is taken while `hip_stream` is being captured into a graph.  0 switches sampling off (dispatch
 * by algorithm and size only).  Same bytes either way. */
int fr_set_dispatch_sampling(int enabled);
/* Tool / test hook: the (blocking) sample of the whole image. out[0..5] = executed iterations, 64 x the sum of the tiles'
 * longest orbits, tiles, lanes at the sample's cap (min(iterations, 4096)), lanes the two-pass render's first-pass schedule
 * would hand over, lane-iterations that finishing those in place would waste; out[6] = out[0] / out[1], the useful-lane
 * fraction of one-tile-per-wave rendering; out[7] = iterations the handed-over lanes would still have to run. */
int fr_debug_sample_view(const fr_config *cfg, int precision, double out[8]);
/* Tool / test hook: what the default dispatch has on record for the view (cfg, rows [y0, y1), precision) as ONE launch:
 * *state = 0 nothing, 1 a non-blocking sample is in flight, 3 its totals have arrived (the next frame reads them), 2 decided; *choice = -1 none, 0 strips, 1 two passes, 2 the first
 * pass alone; *strip_tiles = the strip length it asks for (0 = by launch size).  Touches nothing. */
int fr_debug_view_choice(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, int *state, int *choice,
                         uint32_t *strip_tiles);

/* Policy of the lane-refilling kernels (tuning studies): an orbit episode may end early, so that
 * idle lanes get new pixels, once quit16/16 of its running lanes (work-queue kernel: of the wave's 64 lanes)
 * have finished and at least `minrun` iterations were done.  With tile 11 the two numbers steer its first pass
 * instead: `minrun` = the length of an episode (default 64), `quit16` x 4 = the running lanes a tile needs to stay
 * in the first pass for another episode (default 12: 48 lanes).  -1 = the kernel's own measured default.  Does
 * not affect results. */
int fr_set_refill_policy(int minrun, int quit16);

/* Exact periodicity shortcut, OFF by default.  When on, large images are rendered by the refilling
 * kernel and an orbit that returns BITWISE to a state it has already visited (the floating-point map
 * z -> z^2 + c is a deterministic function of the state, so it is then exactly periodic and can never
 * escape) is fast-forwarded to the iteration cap — (cap - k) mod d further steps — instead of being
 * iterated there.  Output bytes, escape indices and final positions are identical to the plain loop;
 * only the work differs, so bench.py's headline is measured with it off and reports the on-number
 * separately. */
int fr_set_cycle_shortcut(int enabled);

/* smooth == false renders look the outside colour up in an LDS-staged palette (one entry per
 * escape index, built once per call on the device) when iterations < 1280; 0 disables that and
 * computes the colour per pixel.  Same bytes either way (tests compare them). */
int fr_set_palette(int enabled);

/* Smooth colouring needs log2(log2(sqrt(dist)) / 2) per outside pixel (calc/src/lib.rs:222-223).  With
 * the filter on (default) the kernel first brackets that value with the hardware's f32 log and a
 * proven error bound, evaluates the rest of the colour map in f64 at both ends of the bracket and
 * keeps the bytes if they agree (the map is monotone); only pixels whose bracket straddles a byte
 * boundary take the full f64 software log2.  1 (default) first makes the same test in f32 arithmetic, with a
 * correspondingly wider window, and goes to f64 only for the waves that fail it; 2 = the f64 test only;
 * 0 = always the software log2.  Same bytes in all three (tests compare them). */
int fr_set_colour_filter(int enabled);

/* Orbit-loop selector for tuning studies and tests: -1 = automatic (default); 0 = the unscaled
 * loop with an escape check every iteration; 4 / 2 = the scaled loop that checks every 4th / 2nd
 * iteration, used only where it is provably bit-identical (otherwise the call still falls back to
 * 0).  In the four-iteration scaled loop and in the unscaled loop a wave whose lanes have all stayed quiet
 * (far inside the limit / none escaping) for 16 iterations goes on in speculative blocks of 16 unchecked
 * iterations that keep their start state in a second register set: one test at the block's end, the block
 * thrown away and re-run with checks if it fails (only where 16 <= limit^2 <= 2^1000 — f32: 2^100 —, every
 * |c| component <= limit^2 / 8 and the view's coordinates are finite: an orbit past the limit then grows
 * monotonically and passes the limit before anything overflows, so an escape inside a block cannot be
 * missed at its end; fr_debug_loop_plan shows the plan).
 * 5 = automatic without those speculative blocks (A/B).  Every mode produces the same bytes. */
int fr_set_loop_mode(int mode);
/* Tool / test hook, host arithmetic only (works without a device): the loop plan of (cfg, precision) as one launch
 * under the current selectors — *loop_mode = 0 / 2 / 4 as above, *skip_t = the squared distance under which escape
 * checks are skipped, *spec_quiet = iterations a wave must stay under it before it speculates (0 = never). */
int fr_debug_loop_plan(const fr_config *cfg, int precision, uint32_t *loop_mode, double *skip_t, uint32_t *spec_quiet);

/* Test hook (not part of the reference surface): elementwise DEVICE arithmetic over host arrays —
 * which = 0: the kernels' software log2, 1: sqrt, 2: in[k] / in[(k+1) % n], 3: the `as u8` cast —
 * so tests can compare the device's roundings with the host's; which = 4: the colour filter's
 * bracket centre against the f64 nu over EVERY f32 bit pattern in [in[0], in[1]], out[0] = worst error;
 * which = 5: the packed form of the cast ((float)in[k] into byte 1 of the word 0xAABBCCDD, returned whole);
 * which = 6: the number of f32 bit patterns in [in[0], in[1]] on which the packed and the plain cast differ. */
int fr_debug_math(int which, const double *in, double *out, size_t n);
/* Tuning aid: a device buffer of 16 uint64 per persistent wave (8192 waves is enough) to which the waves of the
 * persistent kernels (tile 10; the second pass of 11 / 12 / 14) write their start / end times (100 MHz ticks) and
 * work counts (round 2's kernel also the cycles per phase); NULL turns it off. */
int fr_debug_set_queue_trace(void *d_trace);
/* Test aid: entries per survivor list of the two-pass render (tile 11); 0 = sized from the image.  A tiny
 * value makes the lists overflow, which the first pass absorbs by finishing those pixels itself. */
int fr_debug_set_two_pass_capacity(uint32_t entries_per_list);

#ifdef __cplusplus
}
#endif
#endif /* FRACTAL_HIP_H */
