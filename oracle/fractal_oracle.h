/*
 * fractal_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's escape-time hot path:
 *   calc/src/lib.rs:83-117   Imaginary::square / squared_distance / Add
 *   calc/src/lib.rs:121-139  RGB, RGB::new(r, b, g) argument-order quirk, color_multiply
 *   calc/src/lib.rs:181-197  coord_to_space / xy_to_imaginary
 *   calc/src/lib.rs:199-235  get_recursive_pixel
 *   calc/src/lib.rs:244-257  recursive
 *   src/lib.rs:253-270       get_image, Mandelbrot/Julia arm (row-parallel, row-major)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (fractal-renderer_amd/) never does.
 *
 * Parity status: the reference holds no tests, fixtures or golden vectors for this
 * path and cannot be built here (no Rust toolchain), so this oracle
 * is pinned by the hand-derived known-answer tests of SURVEY.md §8c only
 * ("parity unpinned" at the f64::log2 boundary, which the reference delegates to the
 * platform libm).
 */
#ifndef FRACTAL_ORACLE_H
#define FRACTAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* calc/src/lib.rs:150-154 (declaration order) */
enum { FRO_ALGO_MANDELBROT = 0, FRO_ALGO_BARNSLEY_FERN = 1, FRO_ALGO_JULIA = 2 };

/* calc/src/lib.rs:79-82 */
typedef struct fro_imaginary {
    double re, im;
} fro_imaginary;

/* calc/src/lib.rs:121-125 — the STORED fields r, g, b */
typedef struct fro_rgb {
    uint8_t r, g, b;
} fro_rgb;

/* calc/src/lib.rs:21-37, field for field */
typedef struct fro_config {
    uint32_t algo;
    uint32_t width;
    uint32_t height;
    uint32_t iterations;
    double limit;
    double stable_limit;
    fro_imaginary pos;
    fro_imaginary scale;
    double exposure;
    uint8_t inside;
    uint8_t smooth;
    fro_rgb primary_color;
    fro_rgb secondary_color;
    double color_weight;
    fro_imaginary julia_set;
} fro_config;

/* arithmetic the escape loop runs in: F64 is the reference; F32 is the build-defined
 * fast path of SURVEY.md §8a (coordinates in f64, start/c/limit narrowed to f32, loop in
 * f32, final z widened back, colour in f64). */
enum { FRO_F64 = 0, FRO_F32 = 1 };

/* which log2 get_recursive_pixel's smooth colouring calls:
 *   LIBM  — the platform libm log2, which is what Rust's f64::log2 lowers to
 *           (calc/src/lib.rs:222-223);
 *   SOFT  — the deterministic software log2 the HIP kernels use (fr_log2, from the
 *           product header csrc/fr_math.h) so host and device agree bit for bit. */
enum { FRO_LOG2_LIBM = 0, FRO_LOG2_SOFT = 1 };
void fro_set_log2_mode(int mode);
int fro_get_log2_mode(void);

/* RGB::new(r, b, g) — calc/src/lib.rs:129-131 (second parameter is BLUE) */
fro_rgb fro_rgb_new(uint8_t r, uint8_t b, uint8_t g);

/* Config::new(algo) — calc/src/lib.rs:39-69 */
void fro_config_new(fro_config *cfg, uint32_t algo);

/* recursive() — calc/src/lib.rs:245-257; returns the escape index (== iterations when
 * the cap is exhausted) and stores the final position in *out_pos. */
uint32_t fro_recursive(uint32_t iterations, fro_imaginary start, fro_imaginary c, double limit,
                       fro_imaginary *out_pos);
uint32_t fro_recursive_f32(uint32_t iterations, fro_imaginary start, fro_imaginary c, double limit,
                           fro_imaginary *out_pos);

/* xy_to_imaginary() with the arguments get_recursive_pixel passes — calc/src/lib.rs:186-207 */
fro_imaginary fro_xy_to_imaginary(const fro_config *cfg, uint32_t x, uint32_t y);

/* get_recursive_pixel() — calc/src/lib.rs:199-235 */
fro_rgb fro_get_recursive_pixel(const fro_config *cfg, uint32_t x, uint32_t y);
fro_rgb fro_get_recursive_pixel_p(const fro_config *cfg, int precision, uint32_t x, uint32_t y);

/* get_image() rows [y0, y1) — src/lib.rs:253-270; out holds 3*width*(y1-y0) bytes,
 * row-major, tightly packed r,g,b.  One task per row, dynamically scheduled over
 * `threads` host threads (0 = all online cores).  Returns the threads used. */
int fro_get_image_rows(const fro_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out,
                       int threads);

/* Raw recursive() results for rows [y0, y1): z[2*k], z[2*k+1] = final position,
 * iters[k] = escape index, k = (y-y0)*width + x.  Either pointer may be NULL. */
int fro_escape_rows(const fro_config *cfg, int precision, uint32_t y0, uint32_t y1, double *z,
                    uint32_t *iters, int threads);

/* The colour map of get_recursive_pixel alone (calc/src/lib.rs:214-234 + color_multiply) over n stored
 * recursive() results, e.g. the arrays fro_escape_rows produced; out holds 3*n bytes r,g,b.  Uses the log2
 * mode currently selected. */
int fro_colour_rows(const fro_config *cfg, const double *z, const uint32_t *iters, size_t n, uint8_t *out,
                    int threads);

/* Strided sample of the image (every sx-th column, sy-th row, same per-pixel function):
 * returns Σ executed iterations over the sampled pixels (i+1 on escape at
 * index i, `iterations` on exhaustion) and, if out != NULL, their colours. */
uint64_t fro_sample_image(const fro_config *cfg, int precision, uint32_t sx, uint32_t sy,
                          uint8_t *out, int threads, uint64_t *out_pixels);

/* Σ executed iterations over rows [y0, y1) */
uint64_t fro_count_iterations_rows(const fro_config *cfg, int precision, uint32_t y0, uint32_t y1,
                                   int threads);

/* the log2 currently selected, exposed for ulp studies in tests */
double fro_log2(double x);

/* Algo::BarnsleyFern — src/lib.rs:271-319, 369-401, 417-463 with a deterministic RNG (Philox4x32-10 keyed by
 * `seed`).  threads = rayon::current_num_threads() of the machine being modelled; walkers == 1 is the
 * reference's single sequential orbit, > 1 the same orbit cut into independently played pieces.  Writes
 * 3 * width * height bytes r, g, b.  Returns 0, or -1 on a degenerate argument. */
int fro_fern_image(const fro_config *cfg, uint32_t threads, uint64_t seed, uint32_t walkers, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
