/*
 * fr_fern.hip — Algo::BarnsleyFern on the GPU: get_image's fern arm (src/lib.rs:271-319),
 * fern() (src/lib.rs:417-463) and Image::subtract_pixel (src/lib.rs:383-401).
 *
 * What the reference computes.  Every rayon thread starts from an image filled with secondary_color and
 * plays the chaos game for iterations / threads steps from (pos.re * width, pos.im * height): plot the
 * point (subtract_pixel darkens the pixel it falls on by a fixed rule), draw r in [0, 1) from an unseeded
 * SmallRng (src/lib.rs:428), apply one of four affine maps (r < 0.01, < 0.86, < 0.93, else).  The
 * per-thread images are then "reduced" with combine_images, which adds a into b and returns a
 * (src/lib.rs:275-284, 305-316): the sum is dropped, so get_image returns ONE thread's image —
 * iterations / threads points.  `threads` is therefore an input here (rayon::current_num_threads() on the
 * caller's machine).  The RNG being entropy-seeded, the reference's output is not reproducible even by
 * itself; what can be matched is its distribution.
 *
 * What subtract_pixel does to a pixel depends only on the pixel's current value (value = primary_color and
 * amount = color_weight are constants), so the final pixel is F^m(secondary_color) with m = the number of
 * points that fell on it — F includes RGB::new(r, b, g)'s channel swap (calc/src/lib.rs:129-131), so g and b
 * trade places on every hit.  Hence:
 *   1. hit counts: `walkers` independent orbits play the game in parallel and count hits per pixel with
 *      global atomics (integers: the result does not depend on the order).  Walker 0 starts where the
 *      reference starts and plots every point; the others first run 64 unplotted steps from there — after
 *      which they sit on the attractor to within 0.85^64 — so that together they are statistically one
 *      long orbit cut into pieces (the reference's own argument for its parallel loop, src/lib.rs:290-291).
 *   2. colours: the host iterates F from secondary_color until it cycles (it does within a few hundred
 *      steps: channels shrink to 0, or are untouched when the matching primary channel is 255) and the
 *      device maps count -> colour through that table.
 * Random numbers: Philox4x32-10 keyed by `seed`, counter = (walker, step / 2), two 53-bit uniforms per block —
 * build-defined, deterministic, and restated independently by the test oracle (oracle/), against which the
 * device result is BIT-exact (same seed, same walkers); the match with a single sequential orbit, the
 * reference's shape, is statistical (tests/test_fern.py).
 */
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "fr_ctx.h"

namespace {

constexpr uint32_t kBurnIn = 64;
constexpr uint32_t kMaxWalkers = 1u << 18;

struct FernParams {
    double width, height;      /* config.width / height as f64 */
    double x0, y0;             /* pos.re * width, pos.im * height */
    double pos_re, pos_im;
    double esx, esy;           /* effective_scale_x / _y (src/lib.rs:424-425) */
    uint64_t w, h, len;        /* image extent, as usize */
    uint64_t steps;            /* iterations / threads */
    uint32_t walkers;
    uint32_t key0, key1;
};

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* Rust's `f64 as usize` (saturating; NaN -> 0) */
__device__ inline uint64_t as_usize(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551616.0) return ~0ull;
    return (uint64_t)v;
}

__global__ __launch_bounds__(256) void fern_walk_kernel(const FernParams p, uint32_t *counts) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= p.walkers) return;
    uint64_t n = p.steps / p.walkers + (w < p.steps % p.walkers ? 1u : 0u);
    const uint32_t burn = w == 0 ? 0u : kBurnIn;
    double x = p.x0, y = p.y0;
    uint32_t rnd[4];
    const uint64_t total = n + burn;
    for (uint64_t s = 0; s < total; s++) {
        if (s >= burn) {
            /* subtract_pixel's target (src/lib.rs:433-440) and its bounds rules (:379-389) */
            const uint64_t px = as_usize(((x - p.pos_re) * p.esx) + p.width / 2.0);
            const uint64_t py = as_usize(p.height - ((y + (p.pos_im - 5.0) - 0.5) * p.esy + p.height / 2.0));
            if (px <= p.w && py <= p.len) { /* `x > width` is rejected; x == width spills into the next row */
                const uint64_t index = py * p.w + px;
                if (index < p.len) atomicAdd(counts + index, 1u);
            }
        }
        if ((s & 1) == 0) philox4x32_10(w, (uint32_t)(s >> 1), (uint32_t)(s >> 33), 0u, p.key0, p.key1, rnd);
        const uint32_t lo = rnd[(s & 1) * 2], hi = rnd[(s & 1) * 2 + 1];
        const double r = (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1p-53;
        /* https://en.wikipedia.org/wiki/Barnsley_fern — src/lib.rs:445-461, every product and sum rounded on
         * its own (this file is compiled with -ffp-contract=off) */
        const double old_x = x;
        if (r < 0.01) {
            x = 0.00 * x + 0.00 * y;
            y = 0.00 * old_x + 0.16 * y + 0.00;
        } else if (r < 0.86) {
            x = 0.85 * x + 0.04 * y;
            y = -0.04 * old_x + 0.85 * y + 1.60;
        } else if (r < 0.93) {
            x = 0.20 * x - 0.26 * y;
            y = 0.23 * old_x + 0.22 * y + 1.60;
        } else {
            x = -0.15 * x + 0.28 * y;
            y = 0.26 * old_x + 0.24 * y + 0.44;
        }
    }
}

/* count -> colour: table[m] for m < mu + lambda, then periodic with period lambda */
__global__ __launch_bounds__(256) void fern_colour_kernel(const uint32_t *counts, uint64_t len, const uint32_t *table, uint32_t mu,
                                                          uint32_t lambda, uint8_t *rgb) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= len) return;
    uint32_t m = counts[k];
    if (m >= mu + lambda) m = mu + (m - mu) % lambda;
    const uint32_t v = table[m];
    rgb[3 * k + 0] = (uint8_t)v;
    rgb[3 * k + 1] = (uint8_t)(v >> 8);
    rgb[3 * k + 2] = (uint8_t)(v >> 16);
}

uint8_t rust_as_u8(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

/* One subtract_pixel (src/lib.rs:383-401) on stored fields {r, g, b} packed r | g << 8 | b << 16. */
uint32_t subtract_once(uint32_t pix, const fr_rgb &value, double amount) {
    auto f = [amount](uint32_t c, uint8_t v) {
        return rust_as_u8((double)c * 1.0 / ((((1.0 / ((double)v / 255.0)) - 1.0) * amount) + 1.0));
    };
    const uint8_t a = f(pix & 255u, value.r), b = f((pix >> 8) & 255u, value.g), c = f((pix >> 16) & 255u, value.b);
    /* RGB::new(a, b, c) stores {r: a, g: c, b: b} (calc/src/lib.rs:129-131) */
    return (uint32_t)a | ((uint32_t)c << 8) | ((uint32_t)b << 16);
}

}  // namespace

using namespace fr;

extern "C" int fr_render_fern_rgb8(const fr_config *cfg, uint32_t threads, uint64_t seed, uint32_t walkers, uint8_t *out,
                                   size_t out_len) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (threads == 0) return fail(FR_ERR_INVALID_ARGUMENT, "threads must be >= 1 (rayon::current_num_threads() of the caller)");
    const uint64_t len = (uint64_t)cfg->width * (uint64_t)cfg->height;
    if (len == 0) return FR_OK;
    if (len > 0xFFFFFFFFull) return fail(FR_ERR_INVALID_ARGUMENT, "width * height overflows u32, as in the reference (src/lib.rs:295)");
    if (!out) return fail(FR_ERR_INVALID_ARGUMENT, "out is NULL");
    if (out_len < 3 * len) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 3*width*height");

    FernParams p;
    memset(&p, 0, sizeof p);
    p.width = (double)cfg->width;
    p.height = (double)cfg->height;
    p.x0 = cfg->pos.re * p.width;
    p.y0 = cfg->pos.im * p.height;
    p.pos_re = cfg->pos.re;
    p.pos_im = cfg->pos.im;
    p.esx = 65.0 * cfg->scale.re * (double)cfg->height * 0.006;
    p.esy = 37.0 * cfg->scale.im * (double)cfg->height * 0.006;
    p.w = cfg->width;
    p.h = cfg->height;
    p.len = len;
    p.steps = cfg->iterations / threads; /* per_thread_iterations, src/lib.rs:286-287 */
    if (walkers == 0) { /* enough orbits to fill the chip, each long enough to dwarf its burn-in */
        uint64_t wk = p.steps / 256;
        walkers = (uint32_t)(wk < 1 ? 1 : wk > 65536 ? 65536 : wk);
    }
    if (walkers > kMaxWalkers) return fail(FR_ERR_INVALID_ARGUMENT, "walkers must be <= 262144");
    if (walkers > p.steps) walkers = p.steps ? (uint32_t)p.steps : 1u;
    p.walkers = walkers;
    p.key0 = (uint32_t)seed;
    p.key1 = (uint32_t)(seed >> 32);

    /* count -> colour: iterate subtract_pixel from secondary_color until the value cycles */
    const uint32_t start = (uint32_t)cfg->secondary_color.r | ((uint32_t)cfg->secondary_color.g << 8) |
                           ((uint32_t)cfg->secondary_color.b << 16);
    std::vector<uint32_t> table;
    std::unordered_map<uint32_t, uint32_t> seen;
    uint32_t v = start, mu = 0, lambda = 1;
    for (;;) {
        auto it = seen.find(v);
        if (it != seen.end()) {
            mu = it->second;
            lambda = (uint32_t)table.size() - mu;
            break;
        }
        seen.emplace(v, (uint32_t)table.size());
        table.push_back(v);
        v = subtract_once(v, cfg->primary_color, cfg->color_weight);
    }

    LifeShared ls;
    Ctx *ctx;
    int rc = primary(&ctx);
    if (rc != FR_OK) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    rc = ctx->reserve(ctx->iters, len * sizeof(uint32_t));
    if (rc == FR_OK) rc = ctx->reserve(ctx->rgb, 3 * len);
    if (rc == FR_OK) rc = ctx->reserve(ctx->misc, table.size() * sizeof(uint32_t));
    if (rc != FR_OK) return rc;
    uint32_t *d_counts = static_cast<uint32_t *>(ctx->iters.ptr);
    uint32_t *d_table = static_cast<uint32_t *>(ctx->misc.ptr);
    HIP_TRY(hipMemsetAsync(d_counts, 0, len * sizeof(uint32_t), ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    if (p.steps > 0) {
        hipLaunchKernelGGL(fern_walk_kernel, dim3((walkers + 255) / 256), dim3(256), 0, ctx->stream, p, d_counts);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(fern_colour_kernel, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, ctx->stream, d_counts, len, d_table,
                       mu, lambda, static_cast<uint8_t *>(ctx->rgb.ptr));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, ctx->rgb.ptr, 3 * len, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* also keeps `table` alive until its copy is done */
    return FR_OK;
}
