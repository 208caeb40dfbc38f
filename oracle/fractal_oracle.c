/*
 * fractal_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See fractal_oracle.h.
 *
 * Build: -O2 -ffp-contract=off, no -ffast-math (Rust never contracts a*b+c, so every multiply
 * and add below keeps its own IEEE rounding exactly as calc/src/lib.rs evaluates it).
 *
 * Each function cites the reference lines it restates.  Nothing here is shared with the HIP
 * kernels; FRO_LOG2_SOFT mode uses a COPY (soft_log2.h, checked verbatim by a test) of the product's
 * software log2, which is what lets the parity tests demand byte equality rather than "equal up to libm".
 * The oracle builds from the files of this directory alone.
 */
#include "fractal_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "soft_log2.h" /* a verbatim copy of the product's fr_math.h, kept in oracle/ so the checker builds alone */

static const double fro_log2_table[FR_LOG2_N][3] = FR_LOG2_TABLE_INIT;

static int g_log2_mode = FRO_LOG2_LIBM;

void fro_set_log2_mode(int mode) { g_log2_mode = mode; }
int fro_get_log2_mode(void) { return g_log2_mode; }

double fro_log2(double x) {
    if (g_log2_mode == FRO_LOG2_SOFT) return fr_log2_tab(x, &fro_log2_table[0][0]);
    return log2(x); /* f64::log2 -> platform libm (calc/src/lib.rs:222-223) */
}

/* ---- calc/src/lib.rs:121-139 ------------------------------------------------------------ */

/* `pub const fn new(r: u8, b: u8, g: u8) -> Self { Self { r, g, b } }` — :129-131 */
fro_rgb fro_rgb_new(uint8_t r, uint8_t b, uint8_t g) {
    fro_rgb c;
    c.r = r;
    c.g = g;
    c.b = b;
    return c;
}

/* Rust `expr as u8` for an f64 (saturating since 1.45; NaN -> 0), used at :135-137 */
static uint8_t rust_f64_as_u8(double v) {
    if (v != v) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v; /* truncates toward zero */
}

/* color_multiply — :133-139.  Note it passes (r, g, b) products to new(r, b, g). */
static fro_rgb color_multiply(fro_rgb color, double mult) {
    return fro_rgb_new(rust_f64_as_u8((double)color.r * mult), rust_f64_as_u8((double)color.g * mult),
                       rust_f64_as_u8((double)color.b * mult));
}

/* ---- calc/src/lib.rs:39-69 -------------------------------------------------------------- */

void fro_config_new(fro_config *cfg, uint32_t algo) {
    int fern = algo == FRO_ALGO_BARNSLEY_FERN;
    memset(cfg, 0, sizeof *cfg);
    cfg->width = 2000;
    cfg->height = 1000;
    cfg->iterations = fern ? 10000000u : 50u;
    cfg->limit = 65536.0; /* 2.0_f64.powi(16) */
    cfg->stable_limit = 2.0;
    cfg->pos.re = 0.0;
    cfg->pos.im = 0.0;
    cfg->scale.re = 1.0 * 0.4; /* Imaginary::ONE * 0.4, ONE = {1, 1} (:85) */
    cfg->scale.im = 1.0 * 0.4;
    cfg->exposure = 2.0;
    cfg->inside = 1;
    cfg->smooth = 1;
    cfg->primary_color = fern ? fro_rgb_new(4, 100, 3) : fro_rgb_new(40, 40, 255);
    cfg->secondary_color = fern ? fro_rgb_new(240, 240, 240) : fro_rgb_new(240, 170, 0);
    cfg->color_weight = 0.01;
    cfg->julia_set.re = 0.0;
    cfg->julia_set.im = 0.0;
    cfg->algo = algo;
}

/* ---- calc/src/lib.rs:83-107, 244-257 ---------------------------------------------------- */

uint32_t fro_recursive(uint32_t iterations, fro_imaginary start, fro_imaginary c, double limit,
                       fro_imaginary *out_pos) {
    double squared = limit * limit; /* :246 */
    fro_imaginary previous = start; /* :247 */
    for (uint32_t i = 0; i < iterations; i++) {
        /* previous.square() — :87-92 */
        double sq_re = (previous.re * previous.re) - (previous.im * previous.im);
        double sq_im = 2.0 * previous.re * previous.im; /* (2.0 * re) * im */
        /* + c — :98-107 */
        fro_imaginary next;
        next.re = sq_re + c.re;
        next.im = sq_im + c.im;
        /* squared_distance — :94-96 */
        double dist = next.re * next.re + next.im * next.im;
        if (dist > squared) { /* :251-253 */
            *out_pos = next;
            return i;
        }
        previous = next;
    }
    *out_pos = previous; /* :256 */
    return iterations;
}

/* Build-defined f32 fast path (the reference has none; SURVEY.md §8a note): recursive()
 * templated on f32 — start, c and limit narrowed with `as f32`, identical operation order,
 * final position widened back to f64. */
uint32_t fro_recursive_f32(uint32_t iterations, fro_imaginary start, fro_imaginary c, double limit,
                           fro_imaginary *out_pos) {
    float lim = (float)limit;
    float squared = lim * lim;
    float pre = (float)start.re, pim = (float)start.im;
    float cre = (float)c.re, cim = (float)c.im;
    for (uint32_t i = 0; i < iterations; i++) {
        float sq_re = (pre * pre) - (pim * pim);
        float sq_im = 2.0f * pre * pim;
        float nre = sq_re + cre;
        float nim = sq_im + cim;
        float dist = nre * nre + nim * nim;
        if (dist > squared) {
            out_pos->re = (double)nre;
            out_pos->im = (double)nim;
            return i;
        }
        pre = nre;
        pim = nim;
    }
    out_pos->re = (double)pre;
    out_pos->im = (double)pim;
    return iterations;
}

/* ---- calc/src/lib.rs:181-197 ------------------------------------------------------------ */

static double coord_to_space(double coord, double max, double offset, double pos, double scale) {
    return ((coord / max) - offset) / scale + pos; /* :183 */
}

fro_imaginary fro_xy_to_imaginary(const fro_config *cfg, uint32_t x, uint32_t y) {
    double width = (double)cfg->width;   /* config.width as f64  — :203 */
    double height = (double)cfg->height; /* config.height as f64 — :204 */
    fro_imaginary z;
    z.re = coord_to_space((double)x, height, (width / height) / 2.0, cfg->pos.re, cfg->scale.re); /* :194 */
    z.im = coord_to_space((double)y, height, 0.5, cfg->pos.im, cfg->scale.im);                    /* :195 */
    return z;
}

/* ---- calc/src/lib.rs:199-235 ------------------------------------------------------------ */

static uint32_t escape_pixel(const fro_config *cfg, int precision, uint32_t x, uint32_t y,
                             fro_imaginary *pos, int *is_escape_algo) {
    fro_imaginary start = fro_xy_to_imaginary(cfg, x, y);
    fro_imaginary c;
    *is_escape_algo = 1;
    if (cfg->algo == FRO_ALGO_MANDELBROT) {
        c = start; /* :209 */
    } else if (cfg->algo == FRO_ALGO_JULIA) {
        c = cfg->julia_set; /* :210 */
    } else {
        *is_escape_algo = 0; /* :211 `_ => return RGB::BLACK` */
        pos->re = pos->im = 0.0;
        return 0;
    }
    if (precision == FRO_F32) return fro_recursive_f32(cfg->iterations, start, c, cfg->limit, pos);
    return fro_recursive(cfg->iterations, start, c, cfg->limit, pos);
}

static fro_rgb colour_of(const fro_config *cfg, fro_imaginary pos, uint32_t iters_u) {
    double dist = pos.re * pos.re + pos.im * pos.im; /* :214 */
    if (dist > cfg->stable_limit) {                  /* :216 (stable_limit is NOT squared) */
        double iters = (double)iters_u;              /* :217 */
        if (cfg->smooth) {
            double log_zn = fro_log2(sqrt(dist)) / 2.0; /* :222 */
            double nu = fro_log2(log_zn);               /* :223 */
            iters += 1.0 - nu;                          /* :225 */
        }
        double mult = iters / (double)cfg->iterations * cfg->exposure; /* :228 */
        return color_multiply(cfg->primary_color, mult);               /* :229 */
    } else if (cfg->inside) {
        return color_multiply(cfg->secondary_color, dist); /* :231 */
    }
    return fro_rgb_new(0, 0, 0); /* :233 RGB::BLACK */
}

fro_rgb fro_get_recursive_pixel_p(const fro_config *cfg, int precision, uint32_t x, uint32_t y) {
    fro_imaginary pos;
    int ok;
    uint32_t iters = escape_pixel(cfg, precision, x, y, &pos, &ok);
    if (!ok) return fro_rgb_new(0, 0, 0);
    return colour_of(cfg, pos, iters);
}

fro_rgb fro_get_recursive_pixel(const fro_config *cfg, uint32_t x, uint32_t y) {
    return fro_get_recursive_pixel_p(cfg, FRO_F64, x, y);
}

/* ---- src/lib.rs:253-270 ----------------------------------------------------------------- */

static int pick_threads(int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_num_procs();
    return threads;
#else
    (void)threads;
    return 1;
#endif
}

/* One task per row, dynamic scheduling (rayon's into_par_iter over 0..height, :256-258);
 * rows land at their final row-major offset, which is what flatten().collect() yields (:266-267). */
int fro_get_image_rows(const fro_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out,
                       int threads) {
    int nt = pick_threads(threads);
    const uint64_t width = cfg->width;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int64_t y = (int64_t)y0; y < (int64_t)y1; y++) {
        uint8_t *row = out + 3u * width * (uint64_t)(y - y0);
        for (uint32_t x = 0; x < cfg->width; x++) {
            fro_rgb p = fro_get_recursive_pixel_p(cfg, precision, x, (uint32_t)y);
            row[3 * (uint64_t)x + 0] = p.r;
            row[3 * (uint64_t)x + 1] = p.g;
            row[3 * (uint64_t)x + 2] = p.b;
        }
    }
    return nt;
}

int fro_escape_rows(const fro_config *cfg, int precision, uint32_t y0, uint32_t y1, double *z,
                    uint32_t *iters, int threads) {
    int nt = pick_threads(threads);
    const uint64_t width = cfg->width;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int64_t y = (int64_t)y0; y < (int64_t)y1; y++) {
        for (uint32_t x = 0; x < cfg->width; x++) {
            fro_imaginary pos;
            int ok;
            uint32_t it = escape_pixel(cfg, precision, x, (uint32_t)y, &pos, &ok);
            uint64_t k = (uint64_t)(y - y0) * width + x;
            if (z) {
                z[2 * k] = pos.re;
                z[2 * k + 1] = pos.im;
            }
            if (iters) iters[k] = it;
        }
    }
    return nt;
}

/* get_recursive_pixel's colour map alone — calc/src/lib.rs:214-234 — over n stored recursive() results
 * (z[2k], z[2k+1] = final position, iters[k] = escape index): what the rest of get_recursive_pixel does
 * with `recursive`'s return value (:212).  Lets a caller colour one escape pass with both log2 modes. */
int fro_colour_rows(const fro_config *cfg, const double *z, const uint32_t *iters, size_t n, uint8_t *out,
                    int threads) {
    int nt = pick_threads(threads);
    const int escape_algo = cfg->algo == FRO_ALGO_MANDELBROT || cfg->algo == FRO_ALGO_JULIA;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t k = 0; k < (int64_t)n; k++) {
        fro_imaginary pos;
        pos.re = z[2 * k];
        pos.im = z[2 * k + 1];
        fro_rgb p = escape_algo ? colour_of(cfg, pos, iters[k]) : fro_rgb_new(0, 0, 0); /* :211 */
        out[3 * k + 0] = p.r;
        out[3 * k + 1] = p.g;
        out[3 * k + 2] = p.b;
    }
    return nt;
}

static uint64_t executed_of(const fro_config *cfg, uint32_t iters) {
    /* escape at 0-based index i ran i+1 loop bodies; exhaustion ran `iterations` */
    return iters < cfg->iterations ? (uint64_t)iters + 1u : (uint64_t)cfg->iterations;
}

uint64_t fro_sample_image(const fro_config *cfg, int precision, uint32_t sx, uint32_t sy,
                          uint8_t *out, int threads, uint64_t *out_pixels) {
    int nt = pick_threads(threads);
    if (sx == 0) sx = 1;
    if (sy == 0) sy = 1;
    const uint64_t ncols = (cfg->width + sx - 1) / sx;
    const int64_t nrows = (cfg->height + sy - 1) / sy;
    uint64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) reduction(+ : total)
    for (int64_t j = 0; j < nrows; j++) {
        uint32_t y = (uint32_t)j * sy;
        for (uint64_t i = 0; i < ncols; i++) {
            uint32_t x = (uint32_t)i * sx;
            fro_imaginary pos;
            int ok;
            uint32_t it = escape_pixel(cfg, precision, x, y, &pos, &ok);
            if (ok) total += executed_of(cfg, it);
            if (out) {
                fro_rgb p = ok ? colour_of(cfg, pos, it) : fro_rgb_new(0, 0, 0);
                uint8_t *o = out + 3u * ((uint64_t)j * ncols + i);
                o[0] = p.r;
                o[1] = p.g;
                o[2] = p.b;
            }
        }
    }
    if (out_pixels) *out_pixels = ncols * (uint64_t)nrows;
    return total;
}

uint64_t fro_count_iterations_rows(const fro_config *cfg, int precision, uint32_t y0, uint32_t y1,
                                   int threads) {
    int nt = pick_threads(threads);
    uint64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) reduction(+ : total)
    for (int64_t y = (int64_t)y0; y < (int64_t)y1; y++) {
        for (uint32_t x = 0; x < cfg->width; x++) {
            fro_imaginary pos;
            int ok;
            uint32_t it = escape_pixel(cfg, precision, x, (uint32_t)y, &pos, &ok);
            if (ok) total += executed_of(cfg, it);
        }
    }
    return total;
}

/* ---- Algo::BarnsleyFern: src/lib.rs:271-319 (get_image's fern arm), :369-401 (Image), :417-463 (fern) ----
 *
 * Restated with a deterministic RNG (the reference's SmallRng::from_entropy, :428, cannot be matched by
 * anything, itself included): Philox4x32-10 keyed by `seed`, counter = (walker, step / 2, ...), two
 * 53-bit uniforms per block, r = (u64 >> 11) * 2^-53 like rand's Standard f64.
 *
 * `walkers` == 1 is the reference's shape: ONE orbit of iterations / threads points from
 * (pos.re * width, pos.im * height), every point plotted.  (get_image "reduces" the per-thread images with
 * combine_images(a, b), which adds a INTO b and then returns a (:275-284, :305-316): the sum is dropped and
 * the result is one thread's image — per_thread_iterations points, :286-287.)
 * `walkers` > 1 cuts that orbit into pieces played independently: walker 0 as above, every other walker
 * first takes 64 unplotted steps from the same start (it then sits on the attractor).  The image is built
 * exactly as subtract_pixel builds it: pixel by pixel, hit by hit, in this restatement's own order (the
 * result does not depend on the order: every hit applies the same function to the pixel it lands on). */

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

/* Rust `expr as usize` for an f64: saturating, NaN -> 0 */
static uint64_t rust_f64_as_usize(double v) {
    if (v != v || v <= 0.0) return 0;
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}

/* Image::subtract_pixel — :383-401 (pixel_mut's bounds rules :376-382) */
static void subtract_pixel(fro_rgb *contents, uint64_t width, uint64_t len, uint64_t x, uint64_t y, fro_rgb value,
                           double amount) {
    if (x > width) return; /* :377 (x == width is let through and lands on the next row) */
    if (y > len) return;   /* y * width + x would exceed len (and could wrap) */
    uint64_t index = y * width + x;
    if (len < index) return; /* :381 */
    if (index >= len) return; /* get_mut -> None */
    fro_rgb *pixel = &contents[index];
    *pixel = fro_rgb_new(
        rust_f64_as_u8((double)pixel->r * 1.0 / ((((1.0 / ((double)value.r / 255.0)) - 1.0) * amount) + 1.0)),
        rust_f64_as_u8((double)pixel->g * 1.0 / ((((1.0 / ((double)value.g / 255.0)) - 1.0) * amount) + 1.0)),
        rust_f64_as_u8((double)pixel->b * 1.0 / ((((1.0 / ((double)value.b / 255.0)) - 1.0) * amount) + 1.0)));
}

int fro_fern_image(const fro_config *cfg, uint32_t threads, uint64_t seed, uint32_t walkers, uint8_t *out) {
    const uint64_t w = cfg->width, h = cfg->height, len = w * h;
    if (threads == 0 || walkers == 0 || len == 0) return -1;
    fro_rgb *contents = (fro_rgb *)out; /* 3 packed bytes r, g, b: what the reference transmutes (:13-15) */
    for (uint64_t k = 0; k < len; k++) contents[k] = cfg->secondary_color; /* :294-295 */
    const uint64_t steps = cfg->iterations / threads;                      /* :286-287 */
    const double width = (double)cfg->width, height = (double)cfg->height; /* :418-419 */
    /* 0.006 just works fine — :424-425 */
    const double effective_scale_x = 65.0 * cfg->scale.re * (double)cfg->height * 0.006;
    const double effective_scale_y = 37.0 * cfg->scale.im * (double)cfg->height * 0.006;
    const fro_rgb color = cfg->primary_color; /* :430 */
    if ((uint64_t)walkers > steps) walkers = steps ? (uint32_t)steps : 1u;
    for (uint32_t wk = 0; wk < walkers; wk++) {
        const uint64_t n = steps / walkers + (wk < steps % walkers ? 1u : 0u);
        const uint64_t burn = wk == 0 ? 0u : 64u;
        double x = cfg->pos.re * width, y = cfg->pos.im * height; /* :420-421 */
        uint32_t rnd[4] = {0, 0, 0, 0};
        for (uint64_t s = 0; s < n + burn; s++) {
            if (s >= burn)
                subtract_pixel(contents, w, len, rust_f64_as_usize(((x - cfg->pos.re) * effective_scale_x) + width / 2.0),
                               rust_f64_as_usize(height - ((y + (cfg->pos.im - 5.0) - 0.5) * effective_scale_y + height / 2.0)),
                               color, cfg->color_weight); /* :433-441 */
            if ((s & 1) == 0) {
                rnd[0] = wk, rnd[1] = (uint32_t)(s >> 1), rnd[2] = (uint32_t)(s >> 33), rnd[3] = 0;
                philox4x32_10(rnd, (uint32_t)seed, (uint32_t)(seed >> 32));
            }
            uint64_t bits = ((uint64_t)rnd[(s & 1) * 2 + 1] << 32) | rnd[(s & 1) * 2];
            double r = (double)(bits >> 11) * 0x1p-53; /* :443 `rng.gen::<f64>()` */
            /* :446-461 */
            if (r < 0.01) {
                double old_x = x;
                x = 0.00 * x + 0.00 * y;
                y = 0.00 * old_x + 0.16 * y + 0.00;
            } else if (r < 0.86) {
                double old_x = x;
                x = 0.85 * x + 0.04 * y;
                y = -0.04 * old_x + 0.85 * y + 1.60;
            } else if (r < 0.93) {
                double old_x = x;
                x = 0.20 * x - 0.26 * y;
                y = 0.23 * old_x + 0.22 * y + 1.60;
            } else {
                double old_x = x;
                x = -0.15 * x + 0.28 * y;
                y = 0.26 * old_x + 0.24 * y + 0.44;
            }
        }
        (void)h;
    }
    return 0;
}
