/*
 * fr_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the escape-time hot path.
 *
 * Written for gfx950 only.  MUST be compiled with -ffp-contract=off and without fast-math: the
 * reference (Rust) rounds every multiply and add separately (calc/src/lib.rs:87-107), so an implicit
 * v_fma_f64 in the orbit loop breaks bit parity.  (The loops contain exactly one explicit fma, which
 * is exact by construction — see "orbit loop, scaled form".)
 *
 * What runs where (reference lines in brackets):
 *   coordinate map   [calc/src/lib.rs:181-197]  re depends only on x and im only on y, so a strip's
 *                    56 column values and 8 row values are evaluated in one pass by the wave that
 *                    renders it (the reference's 3 IEEE divisions per pixel become 2 per 7 pixels,
 *                    same values) and handed to the pixel lanes by cross-lane reads;
 *   orbit loop       [calc/src/lib.rs:245-257, 87-107]  one wavefront lane per pixel, state in VGPRs,
 *                    hand-written ISA; a wave leaves it when every lane has escaped (EXEC == 0), when
 *                    the wave-uniform counter reaches the cap, or — refilling kernel — when enough
 *                    lanes are idle to be worth handing them new pixels;
 *   colour map       [calc/src/lib.rs:199-235, 133-139]  fused behind the loop; log2 is the software
 *                    fr_log2 (fr_math.h) whose 3 KB table is staged in LDS; with smooth == false the
 *                    outside colour comes from an LDS-staged palette instead;
 *   image assembly   [src/lib.rs:253-270]  bytes r,g,b at 3*(row*ncols + col); a wave writes 8-pixel
 *                    (24-byte) runs of 8 rows per tile, 192-byte runs per strip row.
 *
 * Kernels: escape_strip_kernel (small launches; large interior-heavy views); escape_first_kernel (round 3: 7-tile strips
 * in episodes, lanes past T frozen and finished once per tile, the common tile in one asm block) — alone (tile = 13) or
 * followed by escape_queue_kernel<.., 1> over the survivor lists (two passes, tile = 11), which of the three a large
 * launch gets being decided from view_sample_kernel's sample of the image (fr_api.hip: choose_kernel);
 * escape_first_v1_kernel (round 2's first pass, tile = 12, kept for A/B); escape_queue_kernel<.., 0> (the work queue
 * over the image, tile = 10); escape_refill_kernel (tile = 9, the opt-in periodicity shortcut, COUNT / ESCAPE outputs of
 * large Julia images); escape_kernel (the first, 4-wave design; kept for the tile-shape study); palette_kernel,
 * colour_kernel, recursive_batch_kernel; math_probe_kernel, nu_scan_kernel, cast_scan_kernel (test hooks).
 */
#include "fr_kernels.h"

#include <cmath>

#include <cstdlib>

#include "fr_math.h"

namespace {

__device__ const double g_log2_tab[FR_LOG2_N][3] = FR_LOG2_TABLE_INIT;

constexpr int kWaves = 4; /* 256-thread workgroups */

/* ---- colour map: calc/src/lib.rs:214-234 -------------------------------------------------- */

struct ColourConsts {
    double stable_limit, exposure;
    double iterations_f64; /* config.iterations as f64 */
    double inv_iterations; /* exact reciprocal when iterations is a power of two, else 0 */
    uint32_t inside, smooth;
    double prim[3], sec[3]; /* stored r, g, b fields as f64 */
    uint32_t filter;        /* smooth colouring: try the f32 bracket first (see colour_outside_filtered) */
    double filt_k;          /* exposure / iterations, any rounding */
    double filt_d[3];       /* prim[k] * |filt_k| * FR_NU_BRACKET * (1 + 2^-20): the bracket's half-width in byte units */
    uint32_t filter32;      /* ... and before that, the same test carried out in f32 */
    float filt_k32, filt_d32[3], prim32[3];
    float filt_lo32;        /* f32 renders: an f32 squared distance at or above this is surely > stable_limit and >= 2 */
};

template <typename P>
__device__ __forceinline__ ColourConsts make_colour_consts(const P &p) {
    ColourConsts c;
    c.stable_limit = p.stable_limit;
    c.exposure = p.exposure;
    c.iterations_f64 = p.iterations_f64;
    c.inv_iterations = p.inv_iterations;
    c.inside = p.inside;
    c.smooth = p.smooth;
    for (int k = 0; k < 3; k++) {
        c.prim[k] = p.prim_f[k];
        c.sec[k] = p.sec_f[k];
        c.filt_d[k] = p.filt_d[k];
    }
    c.filter = p.colour_filter;
    c.filt_k = p.filt_k;
    c.filter32 = p.colour_filter32;
    c.filt_k32 = p.filt_k32;
    for (int k = 0; k < 3; k++) {
        c.filt_d32[k] = p.filt_d32[k];
        c.prim32[k] = p.prim32[k];
    }
    c.filt_lo32 = p.filt_lo32;
    return c;
}

/* Rust's `f64 as u8` (truncate, saturate, NaN -> 0) in two instructions: v_cvt_u32_f64 truncates
 * toward zero, saturates out-of-range inputs (negative -> 0, huge / +inf -> 0xFFFFFFFF) and maps
 * NaN to 0; the min brings it to u8 range.  (Inline asm because a plain C cast of an out-of-range
 * value is undefined behaviour to the optimiser.)  Checked against the host's fr_sat_u8 by
 * test_device_saturating_cast. */
__device__ __forceinline__ uint32_t sat_u8_dev(double v) {
    uint32_t u;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(u) : "v"(v));
    return u < 255u ? u : 255u;
}

__device__ __forceinline__ uint32_t sat_u8_dev(float v) { /* the same, from an f32 */
    uint32_t u;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(v));
    return u < 255u ? u : 255u;
}

/* The same cast, from an f32, written straight into byte `k` of a packed word: v_floor_f32 + v_cvt_pk_u8_f32.
 * The pack instruction rounds to nearest even and saturates (NaN -> 0); behind a floor that is truncation for
 * every value that does not saturate to 0 anyway.  Equal to sat_u8_dev(float) on EVERY f32 bit pattern
 * (tools/ubench/cvt_pk_u8.hip scans them all; test_device_packed_saturating_cast does the same through the
 * library).  Two half-cost instructions instead of a convert and a full-cost integer min, and the three bytes
 * of a pixel arrive packed. */
template <int K>
__device__ __forceinline__ uint32_t sat_u8_pack(float v, uint32_t acc) {
    /* the compiler's own v_floor_f32 / v_cvt_pk_u8_f32 (inline asm here drew an s_nop before every dependent use) */
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(v), (uint32_t)K, acc);
}

__device__ __forceinline__ void colour_multiply(const double col[3], double mult, uint8_t out[3]) {
    out[0] = (uint8_t)sat_u8_dev(col[0] * mult);
    out[1] = (uint8_t)sat_u8_dev(col[2] * mult);
    out[2] = (uint8_t)sat_u8_dev(col[1] * mult);
}

/* outside colouring without the smooth term: a function of the escape index alone */
__device__ __forceinline__ void colour_outside_flat(const ColourConsts &c, uint32_t iters_u, uint8_t out[3]) {
    const double iters = (double)iters_u;
    const double q = (c.inv_iterations != 0.0) ? iters * c.inv_iterations : iters / c.iterations_f64;
    colour_multiply(c.prim, q * c.exposure, out); /* :228-229 */
}

/* Smooth colouring without the two f64 software log2s, when that is provably the same bytes.
 *
 * The reference computes (calc/src/lib.rs:222-229)
 *     nu   = log2(log2(sqrt(dist)) / 2)                  [= log2(log2(dist) / 4) over the reals]
 *     byte = (col * ((iters + 1 - nu) / iterations * exposure)) as u8
 * and every step after nu is monotone in nu (IEEE add, mul and div by a positive constant are monotone,
 * so is the truncating cast).  So if nu is known to lie in [a - E, a + E] and the value col * (...) taken
 * at a, widened by what E and the roundings can move it, stays strictly inside one integer cell, the byte
 * is decided without knowing nu any better.
 *
 * a comes from the hardware's f32 log2 (v_log_f32, twice: 2 + 2 VALU slots instead of ~110 f64
 * instructions).  Error budget on a, for 2 <= dist <= 2^120 (L = log2(dist) in [1, 120]):
 *     dist -> f32                 relative 2^-24, i.e. 8.6e-8 absolute on L
 *     v_log_f32                   <= 1 ulp of L (2^-23 relative)            [measured: fr_debug_math(4)]
 *     second v_log_f32            relative error of L times 1/ln 2, + 1 ulp of |nu| < 8 (4.8e-7)
 *     the f64 path's own nu       differs from the real nu by < 1e-14
 * total < 1e-6; the bracket used is FR_NU_BRACKET = 2^-18 = 3.8e-6 (fr_kernels.h; applied by the host in filt_d).  test_gpu_parity.py scans EVERY f32 in
 * [2, 2^120] on the device and asserts the composite error of a stays under 1.5e-6.
 * Pixels outside that range of dist, and pixels whose widened value touches a cell boundary (about one in
 * 10^5), take the exact path.  Same bytes either way; fr_set_colour_filter(0) forces the exact path. */
/* Stage 1, all in f32 (instructions at half the f64 cost), from an f32 squared distance d32:
 * v32 = col * ((iters + 1 - nu32) * K32).  Four f32 roundings and K's own put it within |v| * 2.4e-7 of
 * col * (iters + 1 - nu32) * K, which is within col * |K| * E of the real value (nu is within E of nu32); the
 * window used is |v32| * 2^-21 + filt_d32 (the host rounds that term up), twice the relative part, so that the
 * roundings of the window's own ends are covered too.  (iterations < 2^24 and 2^-60 <= |K| <= 2^60 — the host
 * checks — keep every step exact enough: iters + 1 converts exactly, nothing under- or overflows.)  Both ends of
 * the window are cast into packed bytes, so one comparison decides all three channels.  Returns, per lane,
 * whether the byte triple is decided (and then out[] holds it). */
/* itp1 = (float)(iters + 1), exact (iterations < 2^24); `lo` = the bytes r | g << 8 | b << 16 when decided */
__device__ __forceinline__ bool colour_filter_stage1_packed(const ColourConsts &c, float d32, float itp1, float &nu32,
                                                            uint32_t &lo) {
    const float l1 = __builtin_amdgcn_logf(d32);
    nu32 = __builtin_amdgcn_logf(l1 * 0.25f);
    const int ch[3] = {0, 2, 1}; /* color_multiply's RGB::new(r, b, g) swap, as in colour_multiply() */
    const float m32 = (itp1 - nu32) * c.filt_k32;
    const float v0 = c.prim32[ch[0]] * m32, v1 = c.prim32[ch[1]] * m32, v2 = c.prim32[ch[2]] * m32;
    const float w0 = __builtin_fmaf(__builtin_fabsf(v0), 0x1p-21f, c.filt_d32[ch[0]]);
    const float w1 = __builtin_fmaf(__builtin_fabsf(v1), 0x1p-21f, c.filt_d32[ch[1]]);
    const float w2 = __builtin_fmaf(__builtin_fabsf(v2), 0x1p-21f, c.filt_d32[ch[2]]);
    lo = sat_u8_pack<2>(v2 - w2, sat_u8_pack<1>(v1 - w1, sat_u8_pack<0>(v0 - w0, 0u)));
    const uint32_t hi = sat_u8_pack<2>(v2 + w2, sat_u8_pack<1>(v1 + w1, sat_u8_pack<0>(v0 + w0, 0u)));
    return lo == hi;
}
__device__ __forceinline__ bool colour_filter_stage1(const ColourConsts &c, float d32, bool in_range, uint32_t iters_u,
                                                     float &nu32, uint8_t out[3]) {
    uint32_t lo;
    const bool same = colour_filter_stage1_packed(c, d32, (float)(iters_u + 1u), nu32, lo);
    out[0] = (uint8_t)lo, out[1] = (uint8_t)(lo >> 8), out[2] = (uint8_t)(lo >> 16);
    return in_range && same;
}

__device__ __forceinline__ bool colour_outside_filtered(const ColourConsts &c, double dist, uint32_t iters_u, uint8_t out[3]) {
    const bool in_range = dist >= 2.0 && dist <= 0x1p120;
    const int ch[3] = {0, 2, 1};
    float nu32;
    /* A wave whose lanes all pass stage 1 is done; about one wave in fifty is not and goes on to stage 2. */
    if (c.filter32) {
        const bool same32 = colour_filter_stage1(c, (float)dist, in_range, iters_u, nu32, out);
        if (__ballot(!same32) == 0ull) return true;
    } else {
        nu32 = __builtin_amdgcn_logf(__builtin_amdgcn_logf((float)dist) * 0.25f);
    }
    /* Stage 2: the same test with the arithmetic after nu32 in f64 (window: |v| * 2^-46 + filt_d) */
    const double it2 = ((double)iters_u + 1.0) - (double)nu32; /* iters + 1 is exact */
    const double m = it2 * c.filt_k;
    bool same = in_range;
    for (int k = 0; k < 3; k++) {
        const double v = c.prim[ch[k]] * m;
        const double w = __builtin_fma(__builtin_fabs(v), 0x1p-46, c.filt_d[ch[k]]);
        const uint32_t lo = sat_u8_dev(v - w), hi = sat_u8_dev(v + w);
        same = same && lo == hi;
        out[k] = (uint8_t)lo;
    }
    return same;
}

/* `palette` (LDS) is non-NULL only when smooth == false: "LDS-staged palette lookup". */
__device__ __forceinline__ void colour_of(const ColourConsts &c, double dist, uint32_t iters_u,
                                          const double *lds_tab, const uint32_t *palette, uint8_t out[3]) {
    if (dist > c.stable_limit) { /* :216 */
        if (palette) {
            const uint32_t v = palette[iters_u];
            out[0] = (uint8_t)v;
            out[1] = (uint8_t)(v >> 8);
            out[2] = (uint8_t)(v >> 16);
        } else if (c.smooth) {
            bool exact = true;
            if (c.filter) exact = !colour_outside_filtered(c, dist, iters_u, out);
            /* the exact path is ~4x the filter: taken by the whole wave only when one of its lanes needs it */
            if (__ballot(exact) != 0ull && exact) {
                double iters = (double)iters_u;
                double log_zn = fr_log2_tab(__builtin_sqrt(dist), lds_tab) * 0.5; /* :222, x/2.0 == x*0.5 */
                double nu = fr_log2_tab(log_zn, lds_tab);                         /* :223 */
                iters += 1.0 - nu;                                                /* :225 */
                double q = (c.inv_iterations != 0.0) ? iters * c.inv_iterations : iters / c.iterations_f64;
                colour_multiply(c.prim, q * c.exposure, out); /* :228-229 */
            }
        } else {
            colour_outside_flat(c, iters_u, out);
        }
    } else if (c.inside) {
        colour_multiply(c.sec, dist, out); /* :231 */
    } else {
        out[0] = out[1] = out[2] = 0; /* :233 */
    }
}

/* The colour of the pixel recursive() left at (re, im) (r2 = re*re, i2 = im*im) after `iters_u`.
 *
 * f64 renders: dist = r2 + i2, the reference's squared_distance() (:214).  f32 renders: the reference's arithmetic
 * on the f32 position is zre*zre + zim*zim in f64 — two conversions, two multiplies, an add, and then three f64
 * compares and a conversion back before the filter's f32 stage can start: a third of the colour map's cost.  So
 * the f32 kernels first try with d32 = fl32(re*re + im*im), which is within 2^-23 (relative) of that f64 value:
 *   - d32 >= filt_lo32 (the host's max(stable_limit, 2) * (1 + 2^-20), rounded up) and d32 <= 2^120 (1 - 2^-20)
 *     PROVE dist > stable_limit and 2 <= dist <= 2^120, the branch and the range the filter needs;
 *   - as the logarithm's argument d32 moves nu by at most 2^-23 / (ln 2)^2 = 2.5e-7 on top of the 1.5e-6 the
 *     scan over every f32 allows (test_colour_filter_bracket_holds_for_every_f32): 1.75e-6, inside the bracket
 *     E = 2^-18 = 3.8e-6 the windows are built from.
 * A wave in which every lane passes both, and stage 1 decides every lane's bytes, never touches f64; any other
 * wave takes the general path below, which starts again from the f64 distance.  Same bytes either way. */
template <typename T>
__device__ __forceinline__ void colour_pixel(const ColourConsts &c, T re, T im, T r2, T i2, uint32_t iters_u,
                                             const double *lds_tab, const uint32_t *palette, uint8_t out[3]) {
    double dist;
    if constexpr (sizeof(T) == 8) {
        dist = (double)(r2 + i2);
    } else {
        if (c.filter32 && c.smooth && palette == nullptr) { /* wave-uniform */
            const float d32 = r2 + i2;
            const bool sure = d32 >= c.filt_lo32 && d32 <= 0x1.ffffep119f;
            if (__ballot(!sure) == 0ull) {
                float nu32;
                const bool decided = colour_filter_stage1(c, d32, true, iters_u, nu32, out);
                if (__ballot(!decided) == 0ull) return;
            }
        }
        const double zre = (double)re, zim = (double)im;
        dist = zre * zre + zim * zim;
    }
    colour_of(c, dist, iters_u, lds_tab, palette, out);
}

/* ---- orbit loop: calc/src/lib.rs:245-257 --------------------------------------------------- */

/* Runs recursive() for one lane.  On return (re, im) is the position recursive() returns (`next`
 * on escape, `previous` on exhaustion — the same registers, updated in place), r2 = re*re and
 * i2 = im*im of that position, and the result is the escape index.
 *
 * Per iteration, exactly the reference's roundings:
 *   square():            (re*re) - (im*im)          |  (2.0*re)*im      (2.0*re as re+re: exact)
 *   + c:                 ... + c.re                 |  ... + c.im
 *   squared_distance():  re'*re' + im'*im'    (re'*re' and im'*im' are reused by the next square())
 *
 * The loop is hand-written gfx950 ISA because its cost IS the kernel's cost: every VALU
 * instruction here issues at 4 cycles per wave on the 16-lane f64 datapath (32-bit integer ops
 * too — measured, tools/ubench/valu_rates.hip), so the floor is 9 VALU instructions per iteration
 * (8 arithmetic + 1 compare) and everything else must be scalar:
 *   - v_cmpx_nlt writes EXEC directly: a lane that escapes drops out with its registers frozen
 *     at `next`, which is what recursive() returns;
 *   - the iteration counter lives in an SGPR; a lane's escape index is written (one masked
 *     v_mov_b32) only on the iterations where EXEC actually changed, found with s_xor_b64 on the
 *     scalar unit;
 *   - the wave leaves the loop when EXEC == 0 (every lane escaped) or the counter hits the cap;
 *   - unrolled x4 so the scalar loop control is amortised; the remainder runs first.
 * EXEC is restored before the block ends, so the compiler's view of control flow is unchanged. */
#define FR_ORBIT_STEP(SFX, TAG)                    \
    "v_add_" SFX " %[t], %[r2], -%[i2]\n"          \
    "v_add_" SFX " %[x], %[re], %[re]\n"           \
    "v_add_" SFX " %[re], %[t], %[cre]\n"          \
    "v_mul_" SFX " %[x], %[x], %[im]\n"            \
    "v_add_" SFX " %[im], %[x], %[cim]\n"          \
    "v_mul_" SFX " %[r2], %[re], %[re]\n"          \
    "v_mul_" SFX " %[i2], %[im], %[im]\n"          \
    "v_add_" SFX " %[t], %[r2], %[i2]\n"           \
    "s_mov_b64 %[sprev], exec\n"                   \
    "v_cmpx_nlt_" SFX " %[lim2], %[t]\n"           \
    "s_xor_b64 %[sdiff], %[sprev], exec\n"         \
    "s_cbranch_scc1 .Lrec" TAG "_%=\n"             \
    ".Lcont" TAG "_%=:\n"

/* Taken only on iterations where EXEC changed: write the escape index of the lanes that just left,
 * then decide whether the wave keeps going.  It stops when no lane is left, or — for the refilling
 * kernel — when at most `thr` lanes are still running and at least `minrun` iterations have been
 * done, so that the idle lanes can be given new pixels (thr = 0 and minrun = 0 reproduce "run until
 * every lane has escaped").  %[si] leaves the block holding the number of iterations completed. */
#define FR_ORBIT_RECORD_(TAG, OFFSET, REARM)       \
    ".Lrec" TAG "_%=:\n" REARM                     \
    "s_add_u32 %[stmp], %[si], " OFFSET "\n"       \
    "s_mov_b64 %[sprev], exec\n"                   \
    "s_mov_b64 exec, %[sdiff]\n"                   \
    "v_mov_b32 %[it], %[stmp]\n"                   \
    "s_mov_b64 exec, %[sprev]\n"                   \
    "s_cbranch_execz .Lquit" TAG "_%=\n"           \
    "s_bcnt1_i32_b64 %[scnt], exec\n"              \
    "s_cmp_gt_u32 %[scnt], %[thr]\n"               \
    "s_cbranch_scc1 .Lcont" TAG "_%=\n"            \
    "s_cmp_lt_u32 %[stmp], %[minrun]\n"            \
    "s_cbranch_scc1 .Lcont" TAG "_%=\n"            \
    ".Lquit" TAG "_%=:\n"                          \
    "s_add_u32 %[si], %[stmp], 1\n"                \
    "s_branch .Ldone_%=\n"
#define FR_ORBIT_RECORD(TAG, OFFSET) FR_ORBIT_RECORD_(TAG, OFFSET, "")
/* the unscaled loop's form: a lane that escapes re-arms the quiet stretch before the wave may speculate (FR_ORBIT_ASM) */
#define FR_ORBIT_RECORD_A(TAG, OFFSET)             \
    FR_ORBIT_RECORD_(TAG, OFFSET, "s_add_u32 %[sspec], %[si], %[specq]\n" "s_cselect_b32 %[sspec], -1, %[sspec]\n" "s_min_u32 %[sspec], %[sspec], %[n]\n")

/* length of a speculative block (FR_ORBIT_ASM, FR_SC_SPEC_BODY, FR_FB_SPEC_LOOP): FR_SPEC_M iterations */
#ifndef FR_SPEC_M
#define FR_SPEC_M 16
#endif
#if FR_SPEC_M == 8
#define FR_SPEC_MSTR "8"
#define FR_SC_SPEC_REST(SFX, S) FR_SC_IT_R5(SFX, S) FR_SC_IT_R(SFX, S, S) FR_SC_IT_R(SFX, S, S)
#define FR_ORBIT_SPEC_REST(SFX, S) FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R(SFX, S, S) FR_ORBIT_IT_R(SFX, S, S)
#elif FR_SPEC_M == 16
#define FR_SPEC_MSTR "16"
#define FR_SC_SPEC_REST(SFX, S) FR_SC_IT_R5(SFX, S) FR_SC_IT_R5(SFX, S) FR_SC_IT_R5(SFX, S)
#define FR_ORBIT_SPEC_REST(SFX, S) FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R5(SFX, S)
#elif FR_SPEC_M == 32
#define FR_SPEC_MSTR "32"
#define FR_SC_SPEC_REST(SFX, S) \
    FR_SC_IT_R5(SFX, S) FR_SC_IT_R5(SFX, S) FR_SC_IT_R5(SFX, S) FR_SC_IT_R5(SFX, S) FR_SC_IT_R5(SFX, S) FR_SC_IT_R5(SFX, S) FR_SC_IT_R(SFX, S, S)
#define FR_ORBIT_SPEC_REST(SFX, S) \
    FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R5(SFX, S) FR_ORBIT_IT_R(SFX, S, S)
#else
#error "FR_SPEC_M must be 8, 16 or 32"
#endif
/* The unscaled iteration without its distance add and compare, reading the state in register set S and writing it to
 * set D ("" = re, im, r2, i2; "1" = re1, im1, r21, i21): 7 vector instructions. */
#define FR_ORBIT_IT_R(SFX, S, D)                           \
    "v_add_" SFX " %[t], %[r2" S "], -%[i2" S "]\n"        \
    "v_add_" SFX " %[x], %[re" S "], %[re" S "]\n"         \
    "v_add_" SFX " %[re" D "], %[t], %[cre]\n"             \
    "v_mul_" SFX " %[x], %[x], %[im" S "]\n"               \
    "v_add_" SFX " %[im" D "], %[x], %[cim]\n"             \
    "v_mul_" SFX " %[r2" D "], %[re" D "], %[re" D "]\n"   \
    "v_mul_" SFX " %[i2" D "], %[im" D "], %[im" D "]\n"
#define FR_ORBIT_IT_R5(SFX, S) \
    FR_ORBIT_IT_R(SFX, S, S) FR_ORBIT_IT_R(SFX, S, S) FR_ORBIT_IT_R(SFX, S, S) FR_ORBIT_IT_R(SFX, S, S) FR_ORBIT_IT_R(SFX, S, S)
#define FR_ORBIT_SPEC_MOVS(MOV) \
    MOV " %[re], %[re1]\n" MOV " %[im], %[im1]\n" MOV " %[r2], %[r21]\n" MOV " %[i2], %[i21]\n"

/* The loop of recursive() as written (9 vector instructions per iteration), with the speculative long blocks of the scaled
 * loop below (FR_SC_SPEC_BODY has the argument; round 4): this is the loop of every wave with a lane the scaled form is not
 * proven for — a Julia constant with a zero component (c = -1, i, 0.25: the named sets), the strip on the real axis — and
 * of launches whose limit leaves no room for skipped checks.  A wave in which no lane has escaped for %[specq] iterations
 * runs FR_SPEC_M iterations without the distance add and the compare (7 instructions each), the first one writing to the
 * second register set, and tests `NOT (limit^2 >= dist)` once at the end — true for an orbit that escaped inside the block
 * (it only grows from there, through +inf to NaN at worst: the host's conditions, plan_loop) — in which case the block is
 * thrown away and run again with checks from its intact start state.  There is no early warning here (no T): every escape
 * after a quiet stretch costs one block, and the stretch doubles each time. */
#define FR_ORBIT_ASM(SFX, MOV)                                                     \
    "s_mov_b64 %[sorig], exec\n"                                                   \
    "v_mov_b32 %[it], %[n]\n"                                                      \
    "s_mov_b32 %[si], 0\n"                                                         \
    "s_min_u32 %[sspec], %[specq], %[n]\n"                                         \
    "s_cbranch_execz .Ldone_%=\n"                                                  \
    "s_and_b32 %[nrem], %[n], 3\n"                                                 \
    "s_cbranch_scc0 .Lmainentry_%=\n"                                              \
    ".Lrem_%=:\n" FR_ORBIT_STEP(SFX, "R")                                          \
    "s_add_u32 %[si], %[si], 1\n"                                                  \
    "s_cmp_lt_u32 %[si], %[nrem]\n"                                                \
    "s_cbranch_scc1 .Lrem_%=\n"                                                    \
    ".Lmainentry_%=:\n"                                                            \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc0 .Ldone_%=\n"                                                   \
    ".Lmain_%=:\n" FR_ORBIT_STEP(SFX, "A") FR_ORBIT_STEP(SFX, "B") FR_ORBIT_STEP(SFX, "C") FR_ORBIT_STEP(SFX, "D") \
    "s_add_u32 %[si], %[si], 4\n"                                                  \
    "s_cmp_lt_u32 %[si], %[sspec]\n"  /* the one bound: min(n, end of the quiet stretch) */ \
    "s_cbranch_scc1 .Lmain_%=\n"                                                   \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc0 .Ldone_%=\n"                                                   \
    "s_sub_u32 %[stmp], %[n], %[si]\n"                                             \
    "s_cmp_ge_u32 %[stmp], " FR_SPEC_MSTR "\n"                                     \
    "s_cbranch_scc1 .Lspec_%=\n"                                                   \
    "s_mov_b32 %[sspec], %[n]\n"                                                   \
    "s_branch .Lmain_%=\n"                                                         \
    ".Lspec_%=:\n" FR_ORBIT_IT_R(SFX, "", "1") FR_ORBIT_SPEC_REST(SFX, "1")        \
    "v_add_" SFX " %[t], %[r21], %[i21]\n"                                         \
    "v_cmp_nge_" SFX " vcc, %[lim2], %[t]\n"                                       \
    "s_cbranch_vccnz .LrbA_%=\n"                                                   \
    "s_add_u32 %[si], %[si], " FR_SPEC_MSTR "\n"                                   \
    "s_sub_u32 %[stmp], %[n], %[si]\n"                                             \
    "s_cmp_ge_u32 %[stmp], " FR_SPEC_MSTR "\n"                                     \
    "s_cbranch_scc0 .LexA_%=\n"                                                    \
    FR_ORBIT_IT_R(SFX, "1", "") FR_ORBIT_SPEC_REST(SFX, "")                        \
    "v_add_" SFX " %[t], %[r2], %[i2]\n"                                           \
    "v_cmp_nge_" SFX " vcc, %[lim2], %[t]\n"                                       \
    "s_cbranch_vccnz .LrbB_%=\n"                                                   \
    "s_add_u32 %[si], %[si], " FR_SPEC_MSTR "\n"                                   \
    "s_sub_u32 %[stmp], %[n], %[si]\n"                                             \
    "s_cmp_ge_u32 %[stmp], " FR_SPEC_MSTR "\n"                                     \
    "s_cbranch_scc1 .Lspec_%=\n"                                                   \
    "s_branch .Lspecout_%=\n"                                                      \
    ".LexA_%=:\n" FR_ORBIT_SPEC_MOVS(MOV)                                          \
    ".Lspecout_%=:\n"                                                              \
    "s_mov_b32 %[sspec], %[n]\n"                                                   \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc1 .Lmain_%=\n"                                                   \
    "s_branch .Ldone_%=\n"                                                         \
    ".LrbB_%=:\n" FR_ORBIT_SPEC_MOVS(MOV)                                          \
    ".LrbA_%=:\n"                                                                  \
    "s_lshl_b32 %[specq], %[specq], 1\n"                                           \
    "s_min_u32 %[specq], %[specq], 0x8000\n"                                       \
    "s_add_u32 %[sspec], %[si], %[specq]\n"                                        \
    "s_cselect_b32 %[sspec], -1, %[sspec]\n"                                       \
    "s_min_u32 %[sspec], %[sspec], %[n]\n"                                         \
    "s_branch .Lmain_%=\n"                                                         \
    FR_ORBIT_RECORD_A("R", "0") FR_ORBIT_RECORD_A("A", "0") FR_ORBIT_RECORD_A("B", "1") \
    FR_ORBIT_RECORD_A("C", "2") FR_ORBIT_RECORD_A("D", "3")                         \
    ".Ldone_%=:\n"                                                                 \
    "s_mov_b64 exec, %[sorig]\n"

/* The wave-control knobs of one run of the loop ("episode"): stop early when at most `thr` lanes are
 * still running and at least `minrun` iterations have been done.  {0, 0} = run until every lane has
 * escaped or the cap is reached. */
struct EpisodeCtl {
    uint32_t thr, minrun;
};

/* Runs up to `iterations` steps of recursive() for the active lanes.  Returns, per lane, the index
 * (0-based, within this run) of the iteration at which it escaped, or a value >= `completed` if it
 * is still running; `completed` (wave-uniform) = iterations every still-running lane has done. */
template <typename T>
__device__ __forceinline__ uint32_t orbit_run(uint32_t iterations, T &re, T &im, T cre, T cim, T squared, T &r2,
                                              T &i2, EpisodeCtl ctl, uint32_t &completed, uint32_t spec_quiet = 0u) {
    uint32_t it;
    T t, x;
    T re1, im1, r21, i21; /* the second register set of the speculative blocks */
    unsigned long long sorig, sprev, sdiff;
    uint32_t si, stmp, nrem, scnt, sspec;
    uint32_t specq = __builtin_amdgcn_readfirstlane(spec_quiet ? spec_quiet : 0xFFFFFFFFu); /* 0 = never; doubles with every block thrown away */
    const uint32_t n = __builtin_amdgcn_readfirstlane(iterations);
    const uint32_t thr = __builtin_amdgcn_readfirstlane(ctl.thr), minrun = __builtin_amdgcn_readfirstlane(ctl.minrun);
#define FR_ORBIT_OPERANDS                                                                                          \
    : [re] "+v"(re), [im] "+v"(im), [r2] "+v"(r2), [i2] "+v"(i2), [it] "=&v"(it), [t] "=&v"(t), [x] "=&v"(x),      \
      [sorig] "=&s"(sorig), [sprev] "=&s"(sprev), [sdiff] "=&s"(sdiff), [si] "=&s"(si), [stmp] "=&s"(stmp),        \
      [nrem] "=&s"(nrem), [scnt] "=&s"(scnt), [re1] "=&v"(re1), [im1] "=&v"(im1), [r21] "=&v"(r21),                \
      [i21] "=&v"(i21), [sspec] "=&s"(sspec), [specq] "+&s"(specq)                                                 \
    : [cre] "v"(cre), [cim] "v"(cim), [lim2] "s"(lim2), [n] "s"(n), [thr] "s"(thr), [minrun] "s"(minrun)           \
    : "vcc", "scc"
    if constexpr (sizeof(T) == 8) {
        /* limit^2 is wave-uniform: pin it in an SGPR pair (v_cmpx's src0) */
        const uint64_t sq_bits = fr_bits_of(squared);
        const uint64_t lim2 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sq_bits) |
                              ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sq_bits >> 32)) << 32);
        asm volatile(FR_ORBIT_ASM("f64", "v_mov_b64") FR_ORBIT_OPERANDS);
    } else {
        const uint32_t lim2 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, squared));
        asm volatile(FR_ORBIT_ASM("f32", "v_mov_b32") FR_ORBIT_OPERANDS);
    }
    (void)re1, (void)im1, (void)r21, (void)i21, (void)sspec;
    completed = si;
    return it;
}

/* recursive() from the start: (re, im) in/out, r2 = re*re and i2 = im*im of the returned position,
 * result = the escape index (== iterations on exhaustion). */
template <typename T>
__device__ __forceinline__ uint32_t orbit(uint32_t iterations, T &re, T &im, T cre, T cim, T squared,
                                          T &r2, T &i2, uint32_t spec_quiet = 0u) {
    r2 = re * re;
    i2 = im * im;
    uint32_t completed;
    return orbit_run<T>(iterations, re, im, cre, cim, squared, r2, i2, EpisodeCtl{0, 0}, completed, spec_quiet);
}

/* ---- orbit loop, scaled form ------------------------------------------------------------------
 *
 * Same recursion, carried as X = 2*re, Y = 2*im, A = X*X (= 4*re*re), B = Y*Y (= 4*im*im):
 *
 *     t4 = A - B                  = 4 * fl(re*re - im*im)
 *     q  = X * Y                  = 2 * fl((2*re)*im)
 *     X' = fma(t4, 0.5, 2*c.re)   = 2 * fl(fl(re*re - im*im) + c.re)     <- t4*0.5 is exact, so the
 *     Y' = q + 2*c.im             = 2 * fl(fl((2*re)*im) + c.im)            fma rounds ONCE, exactly
 *     A' = X'*X',  B' = Y'*Y'                                               where the reference does
 *
 * 6 VALU instructions instead of 8: the doubling folds into the state and the exact-by-
 * construction fma replaces an add.  Scaling by 2 or 4 commutes with IEEE rounding as long as no
 * intermediate is subnormal or overflows; lane_is_scalable() below admits only lanes for which
 * that is provable (every |c| and start component in [2^-300, 2^400], f32: [2^-30, 2^30]: then
 * every iterate is 0 or >= 2^-353 in magnitude, so no product is subnormal), and the host admits
 * only limit <= 2^400 (f32: 2^30).  Any wave with an inadmissible lane runs the unscaled loop.
 *
 * Escape checks are skipped where they are provably false.  If dist_k = fl(re^2 + im^2) <= T and
 * every |c| component <= Cmax, then dist_{k+1} <= g(T) = 2*(T + Cmax)^2 * (1 + 2^-30) (each of
 * |re'|, |im'| <= (T + Cmax)(1 + few ulp)).  The host picks M in {4, 2} and the largest T with
 * g^(M-1)(T) <= limit^2, so after a check `dist <= T` the next M-1 iterations cannot escape and
 * need neither the distance add nor the compare; the M-th iteration is checked against T
 * (v_cmp, no EXEC write) and, only if some lane exceeds T, exactly against limit^2 (v_cmpx).
 * A wave with a lane above T ("in transit") runs M fully-checked iterations at a time until all
 * its live lanes are back under T.  The fast block is M*6 + 2 VALU instructions. */
#define FR_SC_IT(SFX)                              \
    "v_add_" SFX " %[t], %[A], -%[B]\n"            \
    "v_mul_" SFX " %[q], %[X], %[Y]\n"             \
    "v_fma_" SFX " %[X], %[t], 0.5, %[c2re]\n"     \
    "v_add_" SFX " %[Y], %[q], %[c2im]\n"          \
    "v_mul_" SFX " %[A], %[X], %[X]\n"             \
    "v_mul_" SFX " %[B], %[Y], %[Y]\n"

/* The same iteration reading the state in register set S and writing it to set D ("" = X, Y, A, B;
 * "1" = X1, Y1, A1, B1): the speculative blocks below keep their start state by never writing to it. */
#define FR_SC_IT_R(SFX, S, D)                              \
    "v_add_" SFX " %[t], %[A" S "], -%[B" S "]\n"          \
    "v_mul_" SFX " %[q], %[X" S "], %[Y" S "]\n"           \
    "v_fma_" SFX " %[X" D "], %[t], 0.5, %[c2re]\n"        \
    "v_add_" SFX " %[Y" D "], %[q], %[c2im]\n"             \
    "v_mul_" SFX " %[A" D "], %[X" D "], %[X" D "]\n"      \
    "v_mul_" SFX " %[B" D "], %[Y" D "], %[Y" D "]\n"
#define FR_SC_IT_R5(SFX, S) FR_SC_IT_R(SFX, S, S) FR_SC_IT_R(SFX, S, S) FR_SC_IT_R(SFX, S, S) FR_SC_IT_R(SFX, S, S) FR_SC_IT_R(SFX, S, S)

/* Speculative long blocks (round 4).  A wave that has been QUIET for %[specq] iterations — no lane above T at any block
 * end: waves of interior pixels, which carry 97 % of C2's work — stops paying the distance add and the compare every
 * fourth iteration: it runs blocks of FR_SPEC_M iterations with NO check and tests `dist <= T` once at the block's end
 * (6 + 2/M instructions per iteration: 6.125 at M = 16 against 6.5).  What makes that exact although a lane may
 * escape in the middle of such a block:
 *   - the block never overwrites its start state: its first iteration reads one register set and writes the other,
 *     the remaining ones work in place there, and the next block does the same in the opposite direction;
 *   - once dist_k > limit^2 the computed distances only grow (the host enables this only for limit^2 >= 16 and
 *     |c| components <= limit^2 / 8: then |z'| >= |z|^2 (3/4 - 8u) > 2.9 |z|), through +inf to NaN at worst, and
 *     v_cmp_nge (NOT "T >= dist") is true for all of them: an escape inside the block cannot pass the end test;
 *   - a block whose end test fails is thrown away: the wave returns to the block's START state (it is still in
 *     its registers: no copy for the set0 -> set1 block, four moves for the other) and runs the checked machinery
 *     above from there, which finds the escape at its exact iteration; speculation stays off for the next %[specq]
 *     iterations.  A lane merely in transit at the block's end (above T, not yet above the limit) takes the same road.
 * Lanes that have escaped are not in EXEC and keep their final state in set 0, where the checked code left it. */
/* the checked loop's own bound is %[sspec] = min(n, where the quiet stretch ends): no scalar instruction more per block
 * than without speculation; what the bound meant is sorted out when it is reached */
#define FR_SC_SPEC_ENTRY(MSTR)                                                     \
    "s_cmp_lt_u32 %[si], %[sspec]\n"                                               \
    "s_cbranch_scc1 .Lfast_%=\n"                                                   \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc0 .Ldone_%=\n"                                                   \
    "s_sub_u32 %[stmp], %[n], %[si]\n"                                             \
    "s_cmp_ge_u32 %[stmp], " MSTR "\n"                                             \
    "s_cbranch_scc1 .Lspec_%=\n"                                                   \
    "s_mov_b32 %[sspec], %[n]\n"                                                   \
    "s_branch .Lfast_%=\n"
#define FR_SC_NOSPEC_ENTRY                                                         \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc1 .Lfast_%=\n"                                                   \
    "s_branch .Ldone_%=\n"
/* sspec = min(n, si + specq), saturating: the iteration count from which the wave may speculate (again), or the cap */
#define FR_SC_SPEC_ARM                                                             \
    "s_add_u32 %[sspec], %[si], %[specq]\n"                                        \
    "s_cselect_b32 %[sspec], -1, %[sspec]\n"                                       \
    "s_min_u32 %[sspec], %[sspec], %[n]\n"
#define FR_SC_SPEC_BODY(SFX, MOV, MSTR, REST)                                      \
    ".Lspec_%=:\n" FR_SC_IT_R(SFX, "", "1") REST(SFX, "1")                         \
    "v_add_" SFX " %[t], %[A1], %[B1]\n"                                           \
    "v_cmp_nge_" SFX " vcc, %[t4lim], %[t]\n"                                      \
    "s_cbranch_vccnz .LrbA_%=\n"                                                   \
    "s_add_u32 %[si], %[si], " MSTR "\n"                                           \
    "s_sub_u32 %[stmp], %[n], %[si]\n"                                             \
    "s_cmp_ge_u32 %[stmp], " MSTR "\n"                                             \
    "s_cbranch_scc0 .LexA_%=\n"                                                    \
    FR_SC_IT_R(SFX, "1", "") REST(SFX, "")                                         \
    "v_add_" SFX " %[t], %[A], %[B]\n"                                             \
    "v_cmp_nge_" SFX " vcc, %[t4lim], %[t]\n"                                      \
    "s_cbranch_vccnz .LrbB_%=\n"                                                   \
    "s_add_u32 %[si], %[si], " MSTR "\n"                                           \
    "s_sub_u32 %[stmp], %[n], %[si]\n"                                             \
    "s_cmp_ge_u32 %[stmp], " MSTR "\n"                                             \
    "s_cbranch_scc1 .Lspec_%=\n"                                                   \
    "s_branch .Lspecout_%=\n"                                                      \
    ".LexA_%=:\n"                                                                  \
    MOV " %[X], %[X1]\n" MOV " %[Y], %[Y1]\n" MOV " %[A], %[A1]\n" MOV " %[B], %[B1]\n" \
    ".Lspecout_%=:\n"                                                              \
    "s_mov_b32 %[sspec], %[n]\n"                                                   \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc1 .Lfast_%=\n"                                                   \
    "s_branch .Ldone_%=\n"                                                         \
    ".LrbB_%=:\n"                                                                  \
    MOV " %[X], %[X1]\n" MOV " %[Y], %[Y1]\n" MOV " %[A], %[A1]\n" MOV " %[B], %[B1]\n" \
    ".LrbA_%=:\n"                                                                  \
    "s_lshl_b32 %[specq], %[specq], 1\n"                                           \
    "s_min_u32 %[specq], %[specq], 0x8000\n"                                       \
    FR_SC_SPEC_ARM                                                                 \
    "s_branch .Lfast_%=\n"

#define FR_SC_CHECKED_STEP(SFX, TAG)               \
    FR_SC_IT(SFX)                                  \
    "v_add_" SFX " %[t], %[A], %[B]\n"             \
    "s_mov_b64 %[sprev], exec\n"                   \
    "v_cmpx_nlt_" SFX " %[lim4], %[t]\n"           \
    "s_xor_b64 %[sdiff], %[sprev], exec\n"         \
    "s_cbranch_scc1 .Lrec" TAG "_%=\n"             \
    ".Lcont" TAG "_%=:\n"

#define FR_SC_ASM(SFX, MSTR, FAST_ITS, SLOW_STEPS, SLOW_RECORDS, CYC_F, CYC_S, CYC_H, SP_INIT, SP_ARM, SP_ENTRY, SP_BODY)  \
    "s_mov_b64 %[sorig], exec\n"                                                   \
    "v_mov_b32 %[it], %[n]\n"                                                      \
    "s_mov_b32 %[si], 0\n"                                                         \
    SP_INIT                                                                        \
    "s_cbranch_execz .Ldone_%=\n"                                                  \
    "v_add_" SFX " %[t], %[A], %[B]\n"                                                         \
    "s_and_b32 %[nrem], %[n], " MSTR "-1\n"                                        \
    "s_cbranch_scc0 .Lmodesel_%=\n"                                                \
    ".Lrem_%=:\n" FR_SC_CHECKED_STEP(SFX, "R")                                     \
    "s_add_u32 %[si], %[si], 1\n"                                                  \
    "s_cmp_lt_u32 %[si], %[nrem]\n"                                                \
    "s_cbranch_scc1 .Lrem_%=\n"                                                    \
    ".Lmodesel_%=:\n"                                                              \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc0 .Ldone_%=\n"                                                   \
    "v_cmp_lt_" SFX " vcc, %[t4lim], %[t]\n"                                       \
    "s_cbranch_vccnz .Lslow_%=\n"                                                  \
    ".Lfast_%=:\n" FAST_ITS                                                        \
    "v_add_" SFX " %[t], %[A], %[B]\n"                                             \
    "v_cmp_lt_" SFX " vcc, %[t4lim], %[t]\n"                                       \
    "s_add_u32 %[si], %[si], " MSTR "\n"                                           \
    "s_cbranch_vccnz .Lfastexit_%=\n"                                              \
    CYC_F                                                                          \
    SP_ENTRY                                                                       \
    ".Lfastexit_%=:\n"                                                             \
    SP_ARM                                                                         \
    "s_mov_b64 %[sprev], exec\n"                                                   \
    "v_cmpx_nlt_" SFX " %[lim4], %[t]\n"                                           \
    "s_xor_b64 %[sdiff], %[sprev], exec\n"                                         \
    "s_cbranch_scc0 .Lfxnorec_%=\n"                                                \
    "s_sub_u32 %[stmp], %[si], 1\n"                                                \
    "s_mov_b64 %[sprev], exec\n"                                                   \
    "s_mov_b64 exec, %[sdiff]\n"                                                   \
    "v_mov_b32 %[it], %[stmp]\n"                                                   \
    "s_mov_b64 exec, %[sprev]\n"                                                   \
    "s_cbranch_execz .Ldone_%=\n"                                                  \
    "s_bcnt1_i32_b64 %[scnt], exec\n"                                              \
    "s_cmp_gt_u32 %[scnt], %[thr]\n"                                               \
    "s_cbranch_scc1 .Lfxnorec_%=\n"                                                \
    "s_cmp_lt_u32 %[si], %[minrun]\n"                                              \
    "s_cbranch_scc0 .Ldone_%=\n"                                                   \
    ".Lfxnorec_%=:\n"                                                              \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc0 .Ldone_%=\n"                                                   \
    ".Lslow_%=:\n" SLOW_STEPS                                                      \
    "s_add_u32 %[si], %[si], " MSTR "\n"                                           \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc0 .Ldone_%=\n"                                                   \
    CYC_S                                                                          \
    "v_cmp_lt_" SFX " vcc, %[t4lim], %[t]\n"                                       \
    "s_cbranch_vccnz .Lslow_%=\n"                                                  \
    SP_ARM                                                                         \
    "s_branch .Lfast_%=\n"                                                         \
    FR_ORBIT_RECORD("R", "0") SLOW_RECORDS CYC_H SP_BODY                           \
    ".Ldone_%=:\n"                                                                 \
    "s_mov_b64 exec, %[sorig]\n"

/* Exact periodicity check (optional, see orbit_scaled_run): at a block end, a lane whose state
 * (X, Y) is BITWISE equal to the state saved earlier on the same orbit (Xs, Ys) has entered an exact
 * cycle of the floating-point map; it leaves the loop with bit 31 set in its index. */
#define FR_SC_CYC_CHECK(EQ, TAG)                   \
    EQ " %[scyc], %[X], %[Xs]\n"                   \
    EQ " vcc, %[Y], %[Ys]\n"                       \
    "s_and_b64 %[scyc], %[scyc], vcc\n"            \
    "s_cbranch_scc1 .Lcyc" TAG "_%=\n"             \
    ".Lcyccont" TAG "_%=:\n"                       \
    "s_cmp_ge_u32 %[si], %[snext]\n"               \
    "s_cbranch_scc1 .Lsave" TAG "_%=\n"            \
    ".Lsavecont" TAG "_%=:\n"

/* Brent's schedule inside the run: when the run's iteration count passes 32, 64, 128, ... every
 * running lane remembers its state and the count it was taken at. */
#define FR_SC_CYC_SAVE(MOV, TAG)                   \
    ".Lsave" TAG "_%=:\n"                          \
    MOV " %[Xs], %[X]\n"                           \
    MOV " %[Ys], %[Y]\n"                           \
    "v_mov_b32 %[vsaved], %[si]\n"                 \
    "s_lshl_b32 %[snext], %[snext], 1\n"           \
    "s_branch .Lsavecont" TAG "_%=\n"

#define FR_SC_CYC_HANDLER(TAG)                     \
    ".Lcyc" TAG "_%=:\n"                           \
    "s_or_b32 %[stmp], %[si], 0x80000000\n"        \
    "s_mov_b64 %[sprev], exec\n"                   \
    "s_mov_b64 exec, %[scyc]\n"                    \
    "v_mov_b32 %[it], %[stmp]\n"                   \
    "s_andn2_b64 exec, %[sprev], %[scyc]\n"        \
    "s_cbranch_execz .Ldone_%=\n"                  \
    "s_bcnt1_i32_b64 %[scnt], exec\n"              \
    "s_cmp_gt_u32 %[scnt], %[thr]\n"               \
    "s_cbranch_scc1 .Lcyccont" TAG "_%=\n"         \
    "s_cmp_lt_u32 %[si], %[minrun]\n"              \
    "s_cbranch_scc1 .Lcyccont" TAG "_%=\n"         \
    "s_branch .Ldone_%=\n"

#define FR_SC_ASM_M4_(SFX, CYC_F, CYC_S, CYC_H, SP_INIT, SP_ARM, SP_ENTRY, SP_BODY)                 \
    FR_SC_ASM(SFX, "4", FR_SC_IT(SFX) FR_SC_IT(SFX) FR_SC_IT(SFX) FR_SC_IT(SFX),                    \
              FR_SC_CHECKED_STEP(SFX, "A") FR_SC_CHECKED_STEP(SFX, "B") FR_SC_CHECKED_STEP(SFX, "C") \
                  FR_SC_CHECKED_STEP(SFX, "D"),                                                     \
              FR_ORBIT_RECORD("A", "0") FR_ORBIT_RECORD("B", "1") FR_ORBIT_RECORD("C", "2")         \
                  FR_ORBIT_RECORD("D", "3"),                                                        \
              CYC_F, CYC_S, CYC_H, SP_INIT, SP_ARM, SP_ENTRY, SP_BODY)
#define FR_SC_ASM_M2_(SFX, CYC_F, CYC_S, CYC_H)                                            \
    FR_SC_ASM(SFX, "2", FR_SC_IT(SFX) FR_SC_IT(SFX),                                        \
              FR_SC_CHECKED_STEP(SFX, "A") FR_SC_CHECKED_STEP(SFX, "B"),                    \
              FR_ORBIT_RECORD("A", "0") FR_ORBIT_RECORD("B", "1"), CYC_F, CYC_S, CYC_H, "", "", FR_SC_NOSPEC_ENTRY, "")
#define FR_SC_ASM_M4(SFX) FR_SC_ASM_M4_(SFX, "", "", "", "", "", FR_SC_NOSPEC_ENTRY, "")
#define FR_SC_ASM_M2(SFX) FR_SC_ASM_M2_(SFX, "", "", "")
#define FR_SC_ASM_M4_SPEC(SFX, MOV)                                                                 \
    FR_SC_ASM_M4_(SFX, "", "", "", "s_min_u32 %[sspec], %[specq], %[n]\n", FR_SC_SPEC_ARM, FR_SC_SPEC_ENTRY(FR_SPEC_MSTR), \
                  FR_SC_SPEC_BODY(SFX, MOV, FR_SPEC_MSTR, FR_SC_SPEC_REST))
#define FR_SC_CYC_HANDLERS(MOV) \
    FR_SC_CYC_HANDLER("F") FR_SC_CYC_HANDLER("S") FR_SC_CYC_SAVE(MOV, "F") FR_SC_CYC_SAVE(MOV, "S")
#define FR_SC_ASM_M4_CYC(SFX, EQ, MOV) \
    FR_SC_ASM_M4_(SFX, FR_SC_CYC_CHECK(EQ, "F"), FR_SC_CYC_CHECK(EQ, "S"), FR_SC_CYC_HANDLERS(MOV), "", "", FR_SC_NOSPEC_ENTRY, "")
#define FR_SC_ASM_M2_CYC(SFX, EQ, MOV) \
    FR_SC_ASM_M2_(SFX, FR_SC_CYC_CHECK(EQ, "F"), FR_SC_CYC_CHECK(EQ, "S"), FR_SC_CYC_HANDLERS(MOV))

template <typename T>
struct ScalableRange;
template <>
struct ScalableRange<double> {
    static constexpr double lo = 0x1p-300, hi = 0x1p400;
};
template <>
struct ScalableRange<float> {
    static constexpr float lo = 0x1p-30f, hi = 0x1p30f;
};

/* may this lane run the scaled loop with bit-identical results?  (see the proof sketch above) */
template <typename T>
__device__ __forceinline__ bool lane_is_scalable(T re0, T im0, T cre, T cim) {
    constexpr T lo = ScalableRange<T>::lo, hi = ScalableRange<T>::hi;
    auto c_ok = [=](T v) { return __builtin_fabs(v) >= lo && __builtin_fabs(v) <= hi; };
    auto s_ok = [=](T v) { return v == (T)0 || c_ok(v); };
    return c_ok(cre) && c_ok(cim) && s_ok(re0) && s_ok(im0);
}

/* One run of the scaled loop on state (X, Y, A, B) = (2re, 2im, X*X, Y*Y); M = 4 or 2.  Same
 * return convention as orbit_run().  `squared` = limit^2 (of T), `skip_t` = the host's threshold T
 * on fl(re^2+im^2).
 *
 * CYC = true adds the exact periodicity check: (Xs, Ys) is a state this lane's orbit passed through
 * earlier (or a NaN pattern = "none"); a lane found bitwise back at it after `k` completed iterations
 * of this run returns 0x80000000 | k and stops.  The map z -> z^2 + c is a deterministic function of
 * the state, so from then on the orbit repeats with a period dividing the distance between the two
 * visits: the caller fast-forwards it exactly (refill_strip).  Costs 2 VALU per block. */
template <typename T, int M, bool CYC>
__device__ __forceinline__ uint32_t orbit_scaled_run(uint32_t iterations, T &X, T &Y, T &A, T &B, T c2re, T c2im,
                                                     T squared, T skip_t, EpisodeCtl ctl, uint32_t &completed,
                                                     T *pXs = nullptr, T *pYs = nullptr,
                                                     uint32_t *saved_index = nullptr, uint32_t spec_quiet = 0u) {
    uint32_t it;
    T t, q;
    T X1, Y1, A1, B1; /* M == 4 && !CYC: the second register set of the speculative blocks (FR_SC_SPEC_BODY) */
    uint32_t sspec;
    uint32_t specq = __builtin_amdgcn_readfirstlane(spec_quiet ? spec_quiet : 0xFFFFFFFFu); /* 0 = never; doubles with every block thrown away */
    T Xs = CYC ? *pXs : T(0), Ys = CYC ? *pYs : T(0);
    uint32_t vsaved = 0xFFFFFFFFu; /* CYC: the run's iteration count at this lane's latest save, if any */
    uint32_t snext = 32u;          /* CYC: next save point of Brent's schedule within this run */
    unsigned long long sorig, sprev, sdiff, scyc;
    uint32_t si, stmp, nrem, scnt;
    const uint32_t n = __builtin_amdgcn_readfirstlane(iterations);
    const uint32_t thr = __builtin_amdgcn_readfirstlane(ctl.thr), minrun = __builtin_amdgcn_readfirstlane(ctl.minrun);
    const T lim4_v = (T)4 * squared, t4_v = (T)4 * skip_t;
#define FR_SC_OUTPUTS                                                                                           \
    [X] "+v"(X), [Y] "+v"(Y), [A] "+v"(A), [B] "+v"(B), [it] "=&v"(it), [t] "=&v"(t), [q] "=&v"(q),             \
        [sorig] "=&s"(sorig), [sprev] "=&s"(sprev), [sdiff] "=&s"(sdiff), [si] "=&s"(si), [stmp] "=&s"(stmp),   \
        [nrem] "=&s"(nrem), [scnt] "=&s"(scnt)
#define FR_SC_INPUTS                                                                                            \
    [c2re] "v"(c2re), [c2im] "v"(c2im), [lim4] "s"(lim4), [t4lim] "s"(t4lim), [n] "s"(n), [thr] "s"(thr),       \
        [minrun] "s"(minrun)
#define FR_SC_OPERANDS : FR_SC_OUTPUTS : FR_SC_INPUTS : "vcc", "scc"
#define FR_SC_OPERANDS_SPEC                                                                                     \
    : FR_SC_OUTPUTS, [X1] "=&v"(X1), [Y1] "=&v"(Y1), [A1] "=&v"(A1), [B1] "=&v"(B1), [sspec] "=&s"(sspec),      \
      [specq] "+&s"(specq)                                                                                      \
    : FR_SC_INPUTS                                                                                              \
    : "vcc", "scc"
#define FR_SC_OPERANDS_CYC                                                                                      \
    : FR_SC_OUTPUTS, [scyc] "=&s"(scyc), [Xs] "+v"(Xs), [Ys] "+v"(Ys), [vsaved] "+v"(vsaved), [snext] "+s"(snext) \
    : FR_SC_INPUTS                                                                                              \
    : "vcc", "scc"
    if constexpr (sizeof(T) == 8) {
        const uint64_t lb = fr_bits_of(lim4_v), tb = fr_bits_of(t4_v);
        const uint64_t lim4 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)lb) |
                              ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(lb >> 32)) << 32);
        const uint64_t t4lim = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)tb) |
                               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(tb >> 32)) << 32);
        if constexpr (M == 4 && CYC)
            asm volatile(FR_SC_ASM_M4_CYC("f64", "v_cmp_eq_u64", "v_mov_b64") FR_SC_OPERANDS_CYC);
        else if constexpr (M == 4)
            asm volatile(FR_SC_ASM_M4_SPEC("f64", "v_mov_b64") FR_SC_OPERANDS_SPEC);
        else if constexpr (CYC)
            asm volatile(FR_SC_ASM_M2_CYC("f64", "v_cmp_eq_u64", "v_mov_b64") FR_SC_OPERANDS_CYC);
        else
            asm volatile(FR_SC_ASM_M2("f64") FR_SC_OPERANDS);
    } else {
        const uint32_t lim4 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, lim4_v));
        const uint32_t t4lim = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, t4_v));
        if constexpr (M == 4 && CYC)
            asm volatile(FR_SC_ASM_M4_CYC("f32", "v_cmp_eq_u32", "v_mov_b32") FR_SC_OPERANDS_CYC);
        else if constexpr (M == 4)
            asm volatile(FR_SC_ASM_M4_SPEC("f32", "v_mov_b32") FR_SC_OPERANDS_SPEC);
        else if constexpr (CYC)
            asm volatile(FR_SC_ASM_M2_CYC("f32", "v_cmp_eq_u32", "v_mov_b32") FR_SC_OPERANDS_CYC);
        else
            asm volatile(FR_SC_ASM_M2("f32") FR_SC_OPERANDS);
    }
    (void)scyc;
    (void)snext;
    (void)X1, (void)Y1, (void)A1, (void)B1, (void)sspec, (void)specq;
    if constexpr (CYC) {
        *pXs = Xs;
        *pYs = Ys;
        *saved_index = vsaved;
    }
    completed = si;
    return it;
}

/* recursive() from the start through the scaled loop.  Same contract as orbit(). */
template <typename T, int M>
__device__ __forceinline__ uint32_t orbit_scaled(uint32_t iterations, T &re, T &im, T cre, T cim, T squared,
                                                 T skip_t, T &r2, T &i2, uint32_t spec_quiet = 0u) {
    T X = re + re, Y = im + im, A = X * X, B = Y * Y;
    uint32_t completed;
    const uint32_t it = orbit_scaled_run<T, M, false>(iterations, X, Y, A, B, cre + cre, cim + cim, squared, skip_t,
                                               EpisodeCtl{0, 0}, completed, nullptr, nullptr, nullptr, spec_quiet);
    re = X * (T)0.5; /* exact */
    im = Y * (T)0.5;
    r2 = re * re; /* recomputed from the final position: the reference's own re*re, im*im */
    i2 = im * im;
    return it;
}

/* Dispatch for one lane of a wave whose active lanes all call this together: the scaled loop if
 * the host allowed it (loop_mode 4 or 2) AND every active lane of the wave is admissible,
 * otherwise the unscaled loop.  The choice is wave-uniform. */
template <typename T>
__device__ __forceinline__ uint32_t orbit_auto(uint32_t loop_mode, uint32_t iterations, T &re, T &im, T cre, T cim,
                                               T squared, T skip_t, T &r2, T &i2, int strip_scalable = -1,
                                               uint32_t spec_quiet = 0u) {
    if (loop_mode != 0 && strip_scalable != 0) {
        /* admissibility: decided once per strip by the caller (1), or per call from the lanes' values */
        bool all_ok = strip_scalable == 1;
        if (strip_scalable < 0) all_ok = __ballot(!lane_is_scalable<T>(re, im, cre, cim)) == 0ull;
        if (all_ok) {
            if (loop_mode == 4) return orbit_scaled<T, 4>(iterations, re, im, cre, cim, squared, skip_t, r2, i2, spec_quiet);
            return orbit_scaled<T, 2>(iterations, re, im, cre, cim, squared, skip_t, r2, i2);
        }
    }
    return orbit<T>(iterations, re, im, cre, cim, squared, r2, i2, spec_quiet);
}

/* coord_to_space — calc/src/lib.rs:182-184 */
__device__ __forceinline__ double coord_to_space(double coord, double max, double offset, double pos,
                                                 double scale) {
    return ((coord / max) - offset) / scale + pos;
}

/* store one finished pixel: packed r,g,b at 3*(row*ncols + col), or one r,g,b,255 dword at 4*(...) */
__device__ __forceinline__ void store_pixel(uint32_t ncols, uint32_t out_rgba, uint8_t *base, uint32_t row, uint32_t col,
                                            const uint8_t rgb[3]) {
    const uint64_t k = (uint64_t)row * ncols + col;
    if (out_rgba) {
        reinterpret_cast<uint32_t *>(base)[k] =
            (uint32_t)rgb[0] | ((uint32_t)rgb[1] << 8) | ((uint32_t)rgb[2] << 16) | 0xFF000000u;
    } else {
        uint8_t *o = base + 3ull * k;
        o[0] = rgb[0];
        o[1] = rgb[1];
        o[2] = rgb[2];
    }
}
__device__ __forceinline__ void store_pixel(const fr_kparams &p, uint8_t *base, uint32_t row, uint32_t col, const uint8_t rgb[3]) {
    store_pixel(p.ncols, p.out_rgba, base, row, col, rgb);
}

/* Kernel arguments for the COLD parts of a kernel whose hot loop needs the scalar registers: re-read from
 * the kernel-argument segment where they are used (scalar loads through a pointer the optimiser cannot see
 * through) instead of being held in SGPRs from the kernel's entry on, as arguments normally are — the ~50
 * values of the colour map and the addressing overflow the scalar file otherwise, and every use becomes a
 * v_readlane spill reload (224 of them per tile in this kernel's first version). */
typedef const __attribute__((address_space(4))) fr_kparams *KArgs;
#define FR_COLD_PARAMS(NAME)                                                          \
    KArgs NAME = (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); /* `p` is argument 0 */ \
    asm volatile("" : "+s"(NAME)) /* opaque: the loads through it stay where they are written */

/* the lanes (of EXEC) with `b` set: the builtin on the i1 itself — s_and_b64 with EXEC; HIP's __ballot() converts the
 * boolean to an integer and compares it again, two vector instructions */
__device__ __forceinline__ unsigned long long ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

/* bit `lane` of a wave-uniform mask, as a per-lane boolean: a lane mask in scalar registers IS the machine's form of a
 * per-lane boolean, and the builtin says so to the compiler — no instruction at all, where v_cndmask + v_cmp (two or
 * three vector instructions per use) made one from the mask */
__device__ __forceinline__ bool lane_in(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

/* The filter's first stage as the first pass evaluates it (same test, fewer instructions; constants fetched by
 * tile_fast).  The three channels are p_k * m with ONE m, and the window of channel k is p_k * w with
 *     w = |m| * 2^-20 + c,      c = |K| * E * (1 + 2^-9) rounded up     (p_k * c >= filt_d32[k], p_k >= 0),
 * so both ends of all three windows come from two numbers, m - w and m + w.  The relative part is 2^-20 where
 * colour_filter_stage1 has 2^-21: the ends here carry two more roundings (m -/+ w, then the fma), 2^-24 each, on
 * top of the 2.4e-7 of m itself — 3.6e-7 against a window of 9.5e-7.  And the truncating cast of an end x is
 * v_cvt_pk_u8_f32(x - 0.5), the -0.5 folded into the fma: round-to-nearest-even of x - 0.5 IS floor(x) unless x is
 * an exact integer, where it may give x - 1; at the lower end that can only turn "decided" into "undecided", and at
 * the upper end x - 1 is the right answer, because the true value lies STRICTLY inside the window (the slack above).
 * Saturation and NaN as in sat_u8_pack.  Returns whether the byte triple is decided (then `lo` holds it). */
struct Filter32 {
    float lo, k, c, p0, p1, p2; /* channels in OUTPUT order (color_multiply's swap applied where this is filled) */
};
__device__ __forceinline__ uint32_t cvt_pk_u8_at(float v, uint32_t acc, int K) {
    return __builtin_amdgcn_cvt_pk_u8_f32(v, (uint32_t)K, acc); /* v_cvt_pk_u8_f32 (the builtin: no s_nop padding around it, as inline asm gets) */
}
__device__ __forceinline__ bool colour_fast32(const Filter32 &f, float d32, float itp1, uint32_t &lo) {
    const float l1 = __builtin_amdgcn_logf(d32);
    const float nu32 = __builtin_amdgcn_logf(l1 * 0.25f);
    const float m = (itp1 - nu32) * f.k;
    const float w = __builtin_fmaf(__builtin_fabsf(m), 0x1p-20f, f.c);
    const float ml = m - w, mh = m + w;
    lo = cvt_pk_u8_at(__builtin_fmaf(f.p2, ml, -0.5f), cvt_pk_u8_at(__builtin_fmaf(f.p1, ml, -0.5f), cvt_pk_u8_at(__builtin_fmaf(f.p0, ml, -0.5f), 0u, 0), 1), 2);
    const uint32_t hi = cvt_pk_u8_at(__builtin_fmaf(f.p2, mh, -0.5f), cvt_pk_u8_at(__builtin_fmaf(f.p1, mh, -0.5f), cvt_pk_u8_at(__builtin_fmaf(f.p0, mh, -0.5f), 0u, 0), 1), 2);
    return lo == hi;
}

/* one pixel per lane to `base + off` (a wave-uniform base in scalar registers, a 32-bit byte offset per lane: no
 * vector address arithmetic): r,g,b as one 16-bit store — unaligned, as the compiler itself emits them on this
 * target — plus the high byte of the same register, or r,g,b,255 as one dword.
 * The s_nop: a memory instruction that reads a scalar register written by a VECTOR instruction needs five wait states
 * on this target, and the compiler, which inserts them in its own code, does not look inside an asm statement: when
 * it restores a spilled `base` with v_readlane right in front of this one, the store went out with the register's OLD
 * upper half (escape_second_kernel<double>, round 3: "memory access fault", "beyond the largest legal address"). */
__device__ __forceinline__ void store_packed(uint8_t *base, uint32_t off, uint32_t packed, uint32_t bpp) {
    if (bpp == 4u) {
        const uint32_t v = packed | 0xFF000000u;
        asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
    } else {
        asm volatile("s_nop 4\n\tglobal_store_short %0, %1, %2\n\tglobal_store_byte_d16_hi %0, %1, %2 offset:2" : : "v"(off), "v"(packed), "s"(base) : "memory");
    }
}

/* Where a strip kernel's tile goes, for the packed store: a wave-uniform base (the tile's first pixel) and one 32-bit
 * byte offset per lane; fast_colour: the f32 stage of the colour filter applies to this render */
struct TileOut {
    uint8_t *base;
    uint32_t lane_off, bpp;
    bool narrow, fast_colour;
};

/* The colour of one pixel per lane, packed r | g << 8 | b << 16, decided PER LANE by the cheapest rule that proves its
 * bytes (round 3; the strip kernel's colour map was 8 % of C2's instructions, as much again as the first pass's whole
 * tile path):
 *   1. dist surely outside (d32 >= filt_lo32, <= 2^120): the filter's f32 stage in the first pass's form (colour_fast32),
 *      whatever the index — colour_of's first branch looks at dist alone (calc/src/lib.rs:216);
 *   2. !(dist > stable_limit): the `inside` colour or black (:230-233) — the points of the set, a quarter of C2;
 *   3. whatever is left (an undecided window, a dist between the two limits, a palette, no smoothing): colour_pixel.
 * re / im / r2 / i2 / iters as colour_pixel takes them. */
template <typename T>
__device__ __forceinline__ uint32_t colour_packed(bool valid, bool fast_colour, T re, T im, T r2, T i2, uint32_t iters,
                                                  const double *lds_tab, const uint32_t *palette) {
    uint32_t packed = 0;
    unsigned long long todo = ballot64(valid);
    if (fast_colour) {
        Filter32 f;
        {
            FR_COLD_PARAMS(kp);
            f.lo = kp->filt_lo32, f.k = kp->filt_k32, f.c = kp->filt_c32;
            f.p0 = kp->prim32[0], f.p1 = kp->prim32[2], f.p2 = kp->prim32[1]; /* colour_multiply's RGB::new(r, b, g) swap */
        }
        float d32;
        if constexpr (sizeof(T) == 8)
            d32 = (float)(r2 + i2);
        else
            d32 = r2 + i2; /* within 2^-23 of the reference's f64 sum: see colour_pixel */
        uint32_t pk;
        const bool decided = colour_fast32(f, d32, (float)(iters + 1u), pk); /* exact: iterations < 2^24 (the host checks) */
        const bool ok = valid && d32 >= f.lo && d32 <= 0x1.ffffep119f && decided;
        packed = ok ? pk : 0u;
        todo &= ~ballot64(ok);
    }
    if (todo != 0ull) { /* wave-uniform */
        bool inside_done = false;
        if (lane_in(todo)) {
            double dist;
            if constexpr (sizeof(T) == 8) {
                dist = (double)(r2 + i2);
            } else {
                const double zre = (double)re, zim = (double)im;
                dist = zre * zre + zim * zim;
            }
            FR_COLD_PARAMS(kp);
            inside_done = !(dist > kp->stable_limit);
            if (inside_done && kp->inside) /* color_multiply(secondary_color, dist), with its RGB::new(r, b, g) swap */
                packed = sat_u8_dev(kp->sec_f[0] * dist) | (sat_u8_dev(kp->sec_f[2] * dist) << 8) | (sat_u8_dev(kp->sec_f[1] * dist) << 16);
        }
        todo &= ~ballot64(inside_done);
    }
    if (todo != 0ull) {
        if (lane_in(todo)) {
            FR_COLD_PARAMS(kp);
            const ColourConsts cc = make_colour_consts(*kp);
            uint8_t rgb[3];
            colour_pixel<T>(cc, re, im, r2, i2, iters, lds_tab, palette, rgb);
            packed = (uint32_t)rgb[0] | ((uint32_t)rgb[1] << 8) | ((uint32_t)rgb[2] << 16);
        }
    }
    return packed;
}

/* One pixel per lane, from its start coordinate to its output: orbit loop, then the colour map
 * (MODE RGB), the raw recursive() result (MODE ESCAPE) or the executed-iteration sum (MODE COUNT).
 * Every lane of the wave calls this together; `valid` masks lanes that fall outside the image. */
template <typename T, int MODE>
__device__ __forceinline__ void render_pixel(const fr_kparams &p, const fr_kout &out, const double *s_tab,
                                             const uint32_t *s_pal, double sre, double sim, bool valid,
                                             uint32_t cx, uint32_t r, uint32_t lane, uint32_t r_out,
                                             int strip_scalable = -1, const TileOut *to = nullptr) {
    double zre = 0.0, zim = 0.0;
    T tre = 0, tim = 0, tr2 = 0, ti2 = 0; /* the final position and its squares in the render's own type */
    uint32_t iters = 0;
    const bool escape_algo = p.algo == 0 /* Mandelbrot */ || p.algo == 2 /* Julia */;
    if (valid && escape_algo) {
        const double cre = p.algo == 0 ? sre : p.julia_re; /* calc/src/lib.rs:209-210 */
        const double cim = p.algo == 0 ? sim : p.julia_im;
        if constexpr (sizeof(T) == 8) {
            tre = sre;
            tim = sim;
            iters = orbit_auto<double>(p.loop_mode, p.iterations, tre, tim, cre, cim, p.limit * p.limit, p.skip_t,
                                       tr2, ti2, strip_scalable, p.loop_spec);
            zre = tre;
            zim = tim;
        } else {
            tre = (float)sre, tim = (float)sim;
            const float lim = (float)p.limit;
            iters = orbit_auto<float>(p.loop_mode, p.iterations, tre, tim, (float)cre, (float)cim, lim * lim,
                                      (float)p.skip_t, tr2, ti2, strip_scalable, p.loop_spec);
            zre = (double)tre;
            zim = (double)tim;
        }
    }

    if constexpr (MODE == FR_OUT_RGB) {
        if (to != nullptr) { /* the strip kernel: per-lane colour rules, packed store (wave-uniform control up to the store) */
            uint32_t packed = 0;
            if (escape_algo) packed = colour_packed<T>(valid, to->fast_colour, tre, tim, tr2, ti2, iters, s_tab, s_pal);
            if (valid) {
                if (to->narrow) {
                    store_packed(to->base, to->lane_off, packed, to->bpp);
                } else {
                    const uint8_t rgb[3] = {(uint8_t)packed, (uint8_t)(packed >> 8), (uint8_t)(packed >> 16)};
                    store_pixel(p, out.rgb, r_out, cx, rgb);
                }
            }
        } else if (valid) {
            uint8_t rgb[3] = {0, 0, 0};
            if (escape_algo) {
                const ColourConsts cc = make_colour_consts(p);
                colour_pixel<T>(cc, tre, tim, tr2, ti2, iters, s_tab, s_pal, rgb); /* :214-234 */
            }
            store_pixel(p, out.rgb, r_out, cx, rgb);
        }
    } else if constexpr (MODE == FR_OUT_ESCAPE) {
        if (valid) {
            const uint64_t k = (uint64_t)r * p.ncols + cx;
            if (out.z) {
                out.z[2 * k] = zre;
                out.z[2 * k + 1] = zim;
            }
            if (out.iters) out.iters[k] = iters;
        }
    } else {
        unsigned long long n = 0;
        if (valid && escape_algo) n = iters < p.iterations ? (unsigned long long)iters + 1ull : p.iterations;
        for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
        /* FR_COUNT_SLOTS partial sums (the host adds them): one hot address would serialise */
        if (lane == 0 && n) atomicAdd(out.count + ((blockIdx.x + 131u * blockIdx.y) % FR_COUNT_SLOTS), n);
    }
}

template <typename T, int TW, int TH, int WX, int WY, int MODE>
__global__ __launch_bounds__(64 * kWaves) void escape_kernel(const fr_kparams p, const fr_kout out) {
    static_assert(TW * TH == 64 && WX * WY == kWaves, "one lane per pixel, 4 waves per workgroup");
    constexpr int BW = TW * WX, BH = TH * WY;
    __shared__ double s_tab[FR_LOG2_N * 3];
    __shared__ double s_re[BW];
    __shared__ double s_im[BH];

    const uint32_t tid = threadIdx.x;
    const uint32_t tiles_x = (p.ncols + BW - 1) / BW;
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t col0 = bx * BW, row0 = by * BH;

    /* stage the log2 table (3 KB) and this tile's coordinates in LDS */
    if (MODE == FR_OUT_RGB) {
        const double *gt = &g_log2_tab[0][0];
        for (uint32_t k = tid; k < FR_LOG2_N * 3; k += 64 * kWaves) s_tab[k] = gt[k];
    }
    if (tid < BW + BH) {
        const double width = (double)p.width, height = (double)p.height;
        if (tid < BW) {
            const uint32_t x = p.x_first + (col0 + tid) * p.x_stride;
            s_re[tid] = coord_to_space((double)x, height, (width / height) / 2.0, p.pos_re, p.scale_re);
        } else {
            const uint32_t r = row0 + (tid - BW);
            const uint32_t y = p.y_first + (r / p.block_rows) * p.y_stride + r % p.block_rows;
            s_im[tid - BW] = coord_to_space((double)y, height, 0.5, p.pos_im, p.scale_im);
        }
    }
    __syncthreads();

    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t lx = (wave % WX) * TW + lane % TW;
    const uint32_t ly = (wave / WX) * TH + lane / TW;
    const uint32_t cx = col0 + lx, r = row0 + ly;
    const bool valid = cx < p.ncols && r < p.nrows;
    render_pixel<T, MODE>(p, out, s_tab, nullptr, s_re[lx], s_im[ly], valid, cx, r, lane, r);
}

/* May a whole strip run the scaled loop (see "orbit loop, scaled form")?  Every c and start
 * component must be admissible; for a strip they are its column and row coordinates (Mandelbrot:
 * c = start; Julia: c = julia_set, start = the coordinates).  coords_admissible() tests one
 * coordinate per lane (`relevant` masks lanes whose column / row lies past the image edge and never
 * becomes a pixel) and returns the wave-uniform verdict. */
template <typename T>
__device__ __forceinline__ bool coords_admissible(bool julia, double julia_re, double julia_im, double coord, bool relevant) {
    constexpr T lo = ScalableRange<T>::lo, hi = ScalableRange<T>::hi;
    const T v = (T)coord;
    const T av = __builtin_fabs(v);
    const bool in_range = av >= lo && av <= hi;
    bool lane_ok;
    if (julia) {
        const T jr = __builtin_fabs((T)julia_re), ji = __builtin_fabs((T)julia_im);
        lane_ok = (v == (T)0 || in_range) && jr >= lo && jr <= hi && ji >= lo && ji <= hi;
    } else {
        lane_ok = in_range;
    }
    return __ballot(relevant && !lane_ok) == 0ull;
}
template <typename T>
__device__ __forceinline__ bool coords_admissible(const fr_kparams &p, double coord, bool relevant) {
    return coords_admissible<T>(p.algo == 2, p.julia_re, p.julia_im, coord, relevant);
}

/* the strip kernel's layout: columns on lanes 0-55, the 8 rows on lanes 56-63 of one register */
template <typename T>
__device__ __forceinline__ bool strip_is_scalable(const fr_kparams &p, double coord_lane, uint32_t tile0, uint32_t row0,
                                                  uint32_t lane) {
    const bool relevant = lane >= 56 ? (row0 + (lane - 56) < p.nrows) : (tile0 * 8u + lane < p.ncols);
    return coords_admissible<T>(p, coord_lane, relevant);
}

/* Default kernel: ONE WAVE PER WORKGROUP renders a horizontal strip of kStripTiles 8x8 tiles
 * (64 x 8 pixels at 8 tiles), one tile at a time.
 *
 * Why: with one tile per wave, a workgroup's fixed costs — launch, staging the 3 KB log2 table,
 * the barrier behind it — are paid once per 256 pixels, and outside the set (three quarters of the
 * default view, ~20 iterations per pixel) they dominate.  Here they are paid once per strip and
 * there is no barrier at all.  (Multi-wave workgroups with strips were measured and are WORSE the
 * longer the strip: a CU does not backfill the slots of a workgroup's finished waves while one of
 * its waves is still deep inside the set — 4 waves x 8 tiles ran C2 in 22.6 ms against 14.6 ms for
 * 4 waves x 1 tile.  A one-wave workgroup has nothing to wait for.)
 *   - the coordinate map (calc/src/lib.rs:181-197) of the whole strip is evaluated once by the wave
 *     itself (re depends only on x, im only on y: column lanes + 8 row lanes, one pass of the two
 *     IEEE divisions) and handed to the 64 pixel lanes with cross-lane reads (ds_bpermute) — no
 *     LDS storage, no barrier;
 *   - there are still >> 256 workgroups (C2: 524 288) for the dispatcher to balance.
 * The grid is 2-D/3-D, so no integer division is needed to find a tile. */

template <typename T, int MODE, int kStripTiles>
__global__ __launch_bounds__(64) void escape_strip_kernel(const fr_kparams p, const fr_kout out) {
    /* one LDS array, two uses: the log2 table (smooth colouring) or the palette (smooth == false) */
    __shared__ double s_tab[(FR_LOG2_N * 3 * 8 > FR_MAX_PALETTE_ENTRIES * 4 ? FR_LOG2_N * 3 * 8 : FR_MAX_PALETTE_ENTRIES * 4) / 8];
    const uint32_t tid = threadIdx.x;
    const uint32_t *s_pal = nullptr;
    if (MODE == FR_OUT_RGB) {
        if (p.palette != nullptr) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(s_tab);
            for (uint32_t k = tid; k < p.palette_entries; k += 64) dst[k] = p.palette[k];
            s_pal = dst;
        } else if (p.smooth) {
            const double *gt = &g_log2_tab[0][0];
            for (uint32_t k = tid; k < FR_LOG2_N * 3; k += 64) s_tab[k] = gt[k];
        }
        __syncthreads();
    }
    const uint32_t lane = tid;
    const uint32_t row0 = (blockIdx.y + gridDim.y * blockIdx.z) * 8u;
    if (row0 >= p.nrows) return; /* whole workgroup (uniform) */
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t r = row0 + ly;
    const double width = (double)p.width, height = (double)p.height;

    /* The coordinate map of the whole strip in ONE pass: lanes 0 .. 8*kStripTiles-1 evaluate the
     * strip's column coordinates, lanes 56-63 its 8 row coordinates — the same expression
     * ((coord / height) - offset) / scale + pos with per-lane operands (calc/src/lib.rs:182-197). */
    static_assert(kStripTiles <= 7, "lanes 56-63 are the row lanes");
    const bool row_lane = lane >= 56;
    const uint32_t tile0 = blockIdx.x * kStripTiles;
    uint32_t coord_u;
    if (row_lane) {
        const uint32_t rr = row0 + (lane - 56);
        coord_u = p.y_first + (rr / p.block_rows) * p.y_stride + rr % p.block_rows;
    } else {
        coord_u = p.x_first + (tile0 * 8u + lane) * p.x_stride;
    }
    const double coord_lane = coord_to_space((double)coord_u, height, row_lane ? 0.5 : (width / height) / 2.0,
                                             row_lane ? p.pos_im : p.pos_re, row_lane ? p.scale_im : p.scale_re);
    const double sim = __shfl(coord_lane, 56 + ly, 64);
    /* where this strip's rows go: packed, or at their image rows (in place; the tile's 8 rows are in
     * one block because block_rows % 8 == 0, so the map is evaluated once, wave-uniformly) */
    uint32_t out_row0 = row0;
    if (p.out_in_place) out_row0 = p.y_first + (row0 / p.block_rows) * p.y_stride + row0 % p.block_rows;
    const uint32_t r_out = out_row0 + ly;
    const int strip_scalable = (p.loop_mode != 0 && strip_is_scalable<T>(p, coord_lane, tile0, row0, lane)) ? 1 : 0;
    /* RGB: where the strip's tiles go (a scalar base per tile + one 32-bit offset per lane, as in the first pass) */
    TileOut to;
    to.bpp = p.out_rgba ? 4u : 3u;
    to.narrow = (uint64_t)p.ncols * to.bpp * 8u <= 0xFFFFFFFFull;
    to.lane_off = ly * (p.ncols * to.bpp) + lx * to.bpp;
    to.fast_colour = p.colour_filter32 && p.smooth && p.palette == nullptr;
    uint8_t *const strip_base = MODE == FR_OUT_RGB ? out.rgb + ((uint64_t)out_row0 * p.ncols + (uint64_t)tile0 * 8u) * to.bpp : nullptr;

    for (int k = 0; k < kStripTiles; k++) {
        const uint32_t col0 = (tile0 + k) * 8u;
        if (col0 >= p.ncols) break; /* wave-uniform */
        const double sre = __shfl(coord_lane, k * 8 + lx, 64);
        const uint32_t cx = col0 + lx;
        const bool valid = cx < p.ncols && r < p.nrows;
        to.base = strip_base + (size_t)((uint32_t)k * 8u * to.bpp);
        render_pixel<T, MODE>(p, out, s_tab, s_pal, sre, sim, valid, cx, r, lane, r_out, strip_scalable, MODE == FR_OUT_RGB ? &to : nullptr);
    }
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

/* ---- first pass + survivor list (two-pass rendering, part 1) ---------------------------------------
 *
 * C4's orbits (a Julia dust: mean 44 iterations, none near the cap of 4096, 58 % of the pixels gone after 8)
 * are spatially coherent while they are short: an 8x8 tile run only to iteration 128 keeps 91 % of its lanes
 * busy (to 64: 98 %), against 34 % when it runs until its slowest pixel is done — the waste is all in the
 * tail.  So the strip kernel's cheap, static, fully fused form takes every pixel the first `first_cap`
 * iterations; a pixel that has escaped by then (89 % of them at 128) is coloured and stored on the spot,
 * coalesced, with its tile.  The others — position after first_cap iterations, output position (and c, for
 * Mandelbrot) — are appended to a list in device memory, which the work-queue kernel (below, SRC = 1) drains
 * with full waves.  Its per-pixel costs (hand-out, stack, finishing pass with scattered stores) are then paid
 * by one pixel in ten.
 *
 * The list is FR_SURV_QUEUES lists with a counter each (one atomic per tile that has survivors — a million
 * of them on C4 — would queue up on a single address); a strip appends to the list its position hashes to.
 * A list that is full is not an error: the lanes that did not get a slot run their orbit to the end right
 * here, as the strip kernel would have.  Same bytes whichever way a pixel goes: the state after first_cap
 * iterations is recursive()'s own, and the second pass continues it with recursive()'s own arithmetic. */
template <typename T>
struct Pair;
template <>
struct Pair<float> {
    typedef float2 type;
};
template <>
struct Pair<double> {
    typedef double2 type;
};

/* is this coordinate admissible as a start / c component of the scaled loop?  (see lane_is_scalable) */
template <typename T>
__device__ __forceinline__ bool coord_is_scalable(bool julia, double coord) {
    constexpr T lo = ScalableRange<T>::lo, hi = ScalableRange<T>::hi;
    const T v = (T)coord;
    const T av = __builtin_fabs(v);
    const bool in_range = av >= lo && av <= hi;
    return julia ? (v == (T)0 || in_range) : in_range; /* Julia: a start component; Mandelbrot: start and c */
}

/* One wave renders kBands strips of kStripTiles tiles, one below the other: what a workgroup pays once — launch,
 * the scalar loads of its arguments, the coordinate map's IEEE divisions (one per column and one per row of the
 * whole block) — is spread over 4 x 7 tiles instead of 7; with orbits this short a 7-tile strip is three
 * microseconds of work and those fixed costs were a tenth of it. */
template <typename T, int M, int kStripTiles, int kBands>
__global__ __launch_bounds__(64) void escape_first_v1_kernel(const fr_kparams p, const fr_kout out) {
    /* LDS holds the palette only (smooth == false).  The log2 table of the exact colour path is NOT staged here:
     * with the filter on, one wave in fifty needs it — 3 KB of loads, LDS writes and a barrier in front of every
     * workgroup were a measurable part of its time.  The table is read where it lies (device constant data: it
     * stays in L2). */
    __shared__ uint32_t s_palette[FR_MAX_PALETTE_ENTRIES];
    const double *const s_tab = &g_log2_tab[0][0];
    typedef typename Pair<T>::type T2;
    static_assert(kStripTiles <= 8 && kBands <= 8, "one column / one row coordinate per lane");
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t *s_pal = nullptr;
    uint32_t row0, tile0, ncols, nrows, bpp, lane_pitch;
    double cols, rows; /* the block's coordinate map: column tile0*8 + lane, row row0 + lane (calc/src/lib.rs:181-197) */
    bool cols_scalable;
    unsigned long long bad_rows; /* row lanes whose coordinate the scaled loop may not start from */
    bool narrow; /* the 8 rows of a strip span less than 4 GiB: always, short of a 178-million-pixel-wide image */
    {
        FR_COLD_PARAMS(kp);
        const auto &P = *kp;
        if (P.palette != nullptr) {
            const uint32_t n = P.palette_entries;
            const uint32_t *src = P.palette;
            for (uint32_t k = lane; k < n; k += 64) s_palette[k] = src[k];
            s_pal = s_palette;
            __syncthreads();
        }
        row0 = (blockIdx.y + gridDim.y * blockIdx.z) * (8u * kBands);
        nrows = P.nrows, ncols = P.ncols;
        if (row0 >= nrows) return;
        const double width = (double)P.width, height = (double)P.height;
        tile0 = blockIdx.x * kStripTiles;
        const uint32_t block_rows = P.block_rows;
        const uint32_t x = P.x_first + (tile0 * 8u + lane) * P.x_stride;
        cols = coord_to_space((double)x, height, (width / height) / 2.0, P.pos_re, P.scale_re);
        const uint32_t rr = row0 + lane;
        const uint32_t y = P.y_first + (rr / block_rows) * P.y_stride + rr % block_rows;
        rows = coord_to_space((double)y, height, 0.5, P.pos_im, P.scale_im);
        bpp = P.out_rgba ? 4u : 3u;
        narrow = (uint64_t)ncols * bpp * 8u <= 0xFFFFFFFFull;
        lane_pitch = ly * (ncols * bpp) + lx * bpp; /* this lane's byte offset inside a tile's 8 rows */
        /* may the block's strips run the scaled loop?  (strip_is_scalable, per band) */
        const bool is_julia = P.algo == 2;
        bool c_ok = true;
        if (is_julia) {
            constexpr T lo = ScalableRange<T>::lo, hi = ScalableRange<T>::hi;
            const T jr = __builtin_fabs((T)P.julia_re), ji = __builtin_fabs((T)P.julia_im);
            c_ok = jr >= lo && jr <= hi && ji >= lo && ji <= hi;
        }
        const bool col_relevant = lane < 8u * kStripTiles && tile0 * 8u + lane < ncols;
        const bool row_relevant = lane < 8u * kBands && rr < nrows;
        cols_scalable = c_ok && __ballot(col_relevant && !coord_is_scalable<T>(is_julia, cols)) == 0ull;
        bad_rows = __ballot(row_relevant && !coord_is_scalable<T>(is_julia, rows));
    }
    /* what the loops need, held in scalar registers across the block */
    const uint32_t k1 = p.first_cap, cap = p.iterations, keep = p.first_keep; /* the host guarantees 0 < k1 < cap */
    const bool julia = p.algo == 2;
    const T jre = (T)p.julia_re, jim = (T)p.julia_im;
    const T squared = sizeof(T) == 8 ? (T)(p.limit * p.limit) : (T)((float)p.limit * (float)p.limit);
    const T skip_t = (T)p.skip_t;

  for (int band = 0; band < kBands; band++) {
    const uint32_t rb = row0 + 8u * (uint32_t)band; /* the strip's first row */
    if (rb >= nrows) break;                         /* wave-uniform */
    const double sim = __shfl(rows, band * 8 + (int)ly, 64);
    const bool strip_scalable = cols_scalable && ((bad_rows >> (8 * band)) & 0xFFull) == 0ull;
    /* where the strip's pixels go: a wave-uniform base and one 32-bit byte offset per lane — the general form,
     * (row * ncols + col) * 3 in 64 bits per pixel, is three quarter-rate v_mad_u64_u32 per tile.  (In place: the
     * strip's 8 rows lie in one row block, block_rows % 8 == 0.) */
    uint32_t out_row0 = rb;
    {
        FR_COLD_PARAMS(kp);
        if (kp->out_in_place) out_row0 = kp->y_first + (rb / kp->block_rows) * kp->y_stride + rb % kp->block_rows;
    }
    const uint32_t r_out = out_row0 + ly;
    uint8_t *const strip_base = out.rgb + ((uint64_t)out_row0 * ncols + (uint64_t)tile0 * 8u) * bpp;
    const uint32_t off_lane = lane_pitch;
    const uint32_t r = rb + ly;
    /* neighbouring strips append to different lists */
    const uint32_t list = (blockIdx.x + 5u * (rb >> 3)) & (FR_SURV_QUEUES - 1u);

    for (int k = 0; k < kStripTiles; k++) {
        const uint32_t col0 = (tile0 + k) * 8u;
        if (col0 >= ncols) break; /* wave-uniform */
        const double sre = __shfl(cols, k * 8 + (int)lx, 64);
        const uint32_t cx = col0 + lx;
        const bool valid = cx < ncols && r < nrows;
        T re = (T)sre, im = (T)sim, r2 = 0, i2 = 0;
        const T cre = julia ? jre : re, cim = julia ? jim : im; /* calc/src/lib.rs:209-210 */
        uint32_t iters = 0, start = 0;
        bool slow = valid; /* lanes that run recursive()'s plain loop to the end, from `start` */
        bool handed_over = false;
        if (strip_scalable) {
            /* episodes of k1 iterations for as long as the tile is worth a wave of its own: at least `keep`
             * lanes still running (a tile inside the set stays here to the cap, at full lanes, exactly as in
             * the strip kernel); then what is left of it goes to the list, with the iterations it has done */
            slow = false;
            bool running = valid;
            uint32_t done = 0, len = k1; /* a tile still here after 8 episodes is most likely inside a filled set: from
                                          * then on every episode is twice the last (up to 8 x k1) — fewer restarts of
                                          * the loop; a dust's dense tiles keep the short looks that suit them */
            unsigned long long sm;
            for (;;) {
                const uint32_t n = cap - done < len ? cap - done : len;
                if (running) {
                    const uint32_t it = orbit_scaled<T, M>(n, re, im, cre, cim, squared, skip_t, r2, i2);
                    if (it < n) iters = done + it, running = false; /* escaped: (re, im) = `next`, frozen from here on */
                }
                done += n;
                sm = __ballot(running);
                if (sm == 0ull) break;
                if (done == cap) { /* no escape within the cap: recursive() returns (iterations, previous) */
                    if (running) iters = cap;
                    sm = 0ull;
                    break;
                }
                if ((uint32_t)__builtin_popcountll(sm) < keep) break;
                if (done >= 8u * k1 && len < 8u * k1) len += len;
            }
            if (sm != 0ull) { /* hand the running lanes over: (re, im) is the position after `done` iterations */
                FR_COLD_PARAMS(kp);
                const uint32_t sub_cap = kp->surv_sub_capacity;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(kp->surv_counts + list * FR_SURV_COUNT_STRIDE, (uint32_t)__builtin_popcountll(sm));
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
                if (running) {
                    if (slot < sub_cap) {
                        const size_t e = (size_t)list * sub_cap + slot;
                        T2 zz;
                        zz.x = re, zz.y = im;
                        static_cast<T2 *>(kp->surv_z)[e] = zz;
                        reinterpret_cast<uint2 *>(kp->surv_pos)[e] = make_uint2(cx, r_out);
                        kp->surv_cnt[e] = done;
                        if (!julia) {
                            T2 cc2;
                            cc2.x = cre, cc2.y = cim;
                            static_cast<T2 *>(kp->surv_c)[e] = cc2;
                        }
                        handed_over = true;
                    } else {
                        slow = true, start = done; /* the list is full: finish here */
                    }
                }
            }
        }
        /* the plain loop: a strip that may not use the scaled form (wave-uniform: start = 0 for all), or the
         * lanes whose list was full (all of one tile: the same `start`) */
        if (__ballot(slow) != 0ull && slow) {
            const uint32_t rest = cap - start;
            const uint32_t more = orbit<T>(rest, re, im, cre, cim, squared, r2, i2);
            iters = more < rest ? start + more : cap;
        }
        if (valid && !handed_over) {
            FR_COLD_PARAMS(kp);
            const ColourConsts cc = make_colour_consts(*kp);
            uint8_t rgb[3];
            colour_pixel<T>(cc, re, im, r2, i2, iters, s_tab, s_pal, rgb);
            if (narrow) {
                uint8_t *o = strip_base + (size_t)((uint32_t)k * 8u * bpp) + (size_t)off_lane;
                if (bpp == 4u) {
                    *reinterpret_cast<uint32_t *>(o) = (uint32_t)rgb[0] | ((uint32_t)rgb[1] << 8) | ((uint32_t)rgb[2] << 16) | 0xFF000000u;
                } else {
                    o[0] = rgb[0];
                    o[1] = rgb[1];
                    o[2] = rgb[2];
                }
            } else {
                store_pixel(kp->ncols, kp->out_rgba, out.rgb, r_out, cx, rgb);
            }
        }
    }
  } /* band */
}

/* ---- first pass, second form: freeze and finish (round 3) ------------------------------------------
 *
 * What the first form above pays around the orbit loop is, on C4, more than the loop itself: per 8x8 tile 364 VALU
 * instructions of which the loop at its floor is 190, and 150 scalar instructions (profiles/r03_c4_first_pass_isa_classes.txt).
 * The excess has three sources and this form removes each:
 *   1. the transit.  Lanes on their way out (|z|^2 between T and limit^2, 4-5 iterations) put the WHOLE wave on the
 *      fully-checked path (8 VALU + 3 SALU per iteration, a scalar handler per escape event).  Here a lane that
 *      fails the block-end test |z|^2 <= T simply FREEZES (v_cmpx, as in the work-queue kernel): the wave goes on
 *      with unchecked blocks (27 VALU per 4 iterations, 3 SALU, a per-lane f32 count instead of every escape-index
 *      handler) until its episode ends or nobody runs, and the frozen lanes' last few iterations are run ONCE per
 *      tile, together, by an exact-check loop in the scaled form (9 VALU + the count per iteration; no conversion);
 *   2. conversions.  The state stays in the scaled form (X = 2re, Y = 2im, A = X^2, B = Y^2) from the pixel's
 *      start to its colour: the filter's first stage starts from d32 = (A + B) / 4 — the very number
 *      fl(re^2 + im^2) it used before, scaling by 4 being exact — and its (iters + 1) is the lane's count;
 *      only a wave that needs the general colour path, and a hand-over, compute re and im;
 *   3. divergent control.  Which lanes run, froze, escaped or were handed over are 64-bit MASKS in scalar
 *      registers, set into EXEC inside the asm loops; the code around them is uniform.
 * Coordinates are narrowed to T once per block, not per tile; 3-byte pixels leave as one 16-bit and one 8-bit
 * store.  Same bytes: every pixel's orbit is recursive()'s arithmetic whichever loop runs it (the scaled form's
 * exactness and the choice of T are argued at "orbit loop, scaled form"). */
template <typename T>
struct UBits {
    typedef uint32_t type;
};
template <>
struct UBits<double> {
    typedef uint64_t type;
};
template <typename T>
__device__ __forceinline__ typename UBits<T>::type uniform_bits(T v) {
    if constexpr (sizeof(T) == 8) {
        const uint64_t b = fr_bits_of(v);
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32);
    } else {
        return (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, v));
    }
}

/* Blocks of M unchecked scaled iterations for the lanes of `mask` (a subset of EXEC, not empty), one |z|^2 <= T
 * test per block: a lane that fails it freezes with the state and t = A + B it has at that moment, and its count is
 * SET then — to `base` + M x the blocks run so far — by a scalar handler on the blocks in which EXEC changed; lanes
 * that keep running carry no count (the caller knows it: base + M x nblocks).  26 vector instructions per block,
 * 6.5 per iteration — the strip kernel's fast path — at 5 scalar ones: for the tiles that stay past their first episode
 * (the asm path's loop counts per lane, 27 + 3: right for tiles that are gone after a block or two).  Ends after
 * `nblocks` or when no lane runs; returns the lanes that passed every test. */
#define FR_FB_ASM(SFX, BLOCK_ITS, MSHIFT)              \
    "s_mov_b64 %[sorig], exec\n"                       \
    "s_mov_b64 exec, %[mask]\n"                        \
    ".Lfb_%=:\n" BLOCK_ITS                             \
    "v_add_" SFX " %[t], %[A], %[B]\n"                 \
    "s_mov_b64 %[sprev], exec\n"                       \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"              \
    "s_xor_b64 %[sdiff], %[sprev], exec\n"             \
    "s_cbranch_scc1 .Lfbr_%=\n"                        \
    ".Lfbc_%=:\n"                                      \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_cbranch_scc0 .Lfb_%=\n"                         \
    "s_branch .Lfbd_%=\n"                              \
    ".Lfbr_%=:\n"                                      \
    "s_sub_u32 %[stmp], %[n0], %[k]\n"                 \
    "s_lshl_b32 %[stmp], %[stmp], " MSHIFT "\n"        \
    "s_add_u32 %[stmp], %[stmp], %[base]\n"            \
    "s_mov_b64 %[sprev], exec\n"                       \
    "s_mov_b64 exec, %[sdiff]\n"                       \
    "v_cvt_f32_u32 %[cnt], %[stmp]\n"                  \
    "s_mov_b64 exec, %[sprev]\n"                       \
    "s_cbranch_execnz .Lfbc_%=\n"                      \
    ".Lfbd_%=:\n"                                      \
    "s_mov_b64 %[srun], exec\n"                        \
    "s_mov_b64 exec, %[sorig]\n"

/* the same loop with the count kept per lane (27 vector instructions per block, 3 scalar): f32 renders, whose scalar
 * unit is the co-limiter (C4-f32 with the handler form in its later episodes: +4 %) */
#define FR_FBC_ASM(SFX, BLOCK_ITS, STEP)               \
    "s_mov_b64 %[sorig], exec\n"                       \
    "s_mov_b64 exec, %[mask]\n"                        \
    ".Lfc_%=:\n" BLOCK_ITS                             \
    "v_add_" SFX " %[t], %[A], %[B]\n"                 \
    "v_add_f32 %[cnt], %[cnt], " STEP "\n"             \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"              \
    "s_cbranch_execz .Lfcd_%=\n"                       \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_cbranch_scc0 .Lfc_%=\n"                         \
    ".Lfcd_%=:\n"                                      \
    "s_mov_b64 %[srun], exec\n"                        \
    "s_mov_b64 exec, %[sorig]\n"

/* M = 4: the same two loops with the speculative long blocks of the strip kernel's loop (FR_SC_SPEC_BODY has the argument):
 * after a STRETCH of %[specq] blocks in which no lane froze, the wave runs FR_SPEC_M iterations at a time with one
 * `NOT (T >= dist)` test at their end, the first iteration writing to the second register set so that the block's start state
 * survives; a failed test throws the block away and the checked blocks run from its start state — they freeze the same lanes
 * at the same iterations with the same counts as if there had been no speculation.  (96 + 2) / 16 = 6.125 instructions per
 * iteration (f32, counting per lane: 6.19) against 6.5 (6.75).
 *
 * Counters, in blocks of four iterations: %[k] = blocks left in the current stretch - 1 (the checked loop's own counter: it
 * costs what the loop without speculation costs — C4, where nothing ever stays, was 2.5 % slower with a separate
 * comparison per block), %[kend] = blocks left after the stretch; remaining = k + 1 + kend.  A stretch ends by borrow:
 * nothing left -> done; EXEC as it was when the stretch began -> quiet: speculate while kend >= KB; else a new stretch.
 * A speculative block is KB = FR_SPEC_M / 4 blocks of kend. */
/* begin a stretch over the remaining R = %[kend] blocks (R >= 1): k = min(R, specq) - 1, kend = R - k - 1, EXEC remembered */
#define FR_FB_SPEC_STRETCH                             \
    "s_min_u32 %[k], %[kend], %[specq]\n"              \
    "s_sub_u32 %[kend], %[kend], %[k]\n"               \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_mov_b64 %[squiet], exec\n"
#define FR_FB_SPEC_MOVS(MOV) MOV " %[X], %[X1]\n" MOV " %[Y], %[Y1]\n" MOV " %[A], %[A1]\n" MOV " %[B], %[B1]\n"
/* the end of a stretch and the speculative loop behind it; COUNT = what a successful block does for the per-lane count
 * ("" in the handler form); LOOP = the checked loop's head, DONE = the exit */
#define FR_FB_SPEC_EVENT(SFX, MOV, REST, KB, COUNT, LOOP, DONE)  \
    "s_cmp_eq_u32 %[kend], 0\n"                               \
    "s_cbranch_scc1 " DONE "\n"                               \
    "s_cmp_eq_u64 %[squiet], exec\n"                          \
    "s_cbranch_scc0 .Lfnew_%=\n"                              \
    "s_cmp_lt_u32 %[kend], " KB "\n"                          \
    "s_cbranch_scc1 .Lfnew_%=\n"                              \
    ".Lfsp_%=:\n" FR_SC_IT_R(SFX, "", "1") REST(SFX, "1")     \
    "v_add_" SFX " %[t], %[A1], %[B1]\n"                      \
    "v_cmp_nge_" SFX " vcc, %[t4lim], %[t]\n"                 \
    "s_cbranch_vccnz .LfrbA_%=\n"                             \
    COUNT                                                     \
    "s_sub_u32 %[kend], %[kend], " KB "\n"                    \
    "s_cmp_lt_u32 %[kend], " KB "\n"                          \
    "s_cbranch_scc1 .LfexA_%=\n"                              \
    FR_SC_IT_R(SFX, "1", "") REST(SFX, "")                    \
    "v_add_" SFX " %[t], %[A], %[B]\n"                        \
    "v_cmp_nge_" SFX " vcc, %[t4lim], %[t]\n"                 \
    "s_cbranch_vccnz .LfrbB_%=\n"                             \
    COUNT                                                     \
    "s_sub_u32 %[kend], %[kend], " KB "\n"                    \
    "s_cmp_lt_u32 %[kend], " KB "\n"                          \
    "s_cbranch_scc0 .Lfsp_%=\n"                               \
    "s_branch .Lfrest_%=\n"                                   \
    ".LfexA_%=:\n" FR_FB_SPEC_MOVS(MOV)                       \
    ".Lfrest_%=:\n"  /* fewer than KB blocks left: the rest runs checked (or nothing is left) */ \
    "s_cmp_eq_u32 %[kend], 0\n"                               \
    "s_cbranch_scc1 " DONE "\n"                               \
    "s_branch .Lfnew_%=\n"                                    \
    ".LfrbB_%=:\n" FR_FB_SPEC_MOVS(MOV)                       \
    ".LfrbA_%=:\n"                                            \
    "s_lshl_b32 %[specq], %[specq], 1\n"                      \
    "s_min_u32 %[specq], %[specq], 0x2000\n"                  \
    ".Lfnew_%=:\n" FR_FB_SPEC_STRETCH                         \
    "s_branch " LOOP "\n"

/* handler form (f64): %[n0] = the number of blocks to run (>= 1) */
#define FR_FB_SPEC_ASM(SFX, MOV, BLOCK_ITS, MSHIFT, REST, KB) \
    "s_mov_b64 %[sorig], exec\n"                       \
    "s_mov_b64 exec, %[mask]\n"                        \
    "s_mov_b32 %[kend], %[n0]\n"                       \
    FR_FB_SPEC_STRETCH                                 \
    ".Lfb_%=:\n" BLOCK_ITS                             \
    "v_add_" SFX " %[t], %[A], %[B]\n"                 \
    "s_mov_b64 %[sprev], exec\n"                       \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"              \
    "s_xor_b64 %[sdiff], %[sprev], exec\n"             \
    "s_cbranch_scc1 .Lfbr_%=\n"                        \
    ".Lfbc_%=:\n"                                      \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_cbranch_scc0 .Lfb_%=\n"                         \
    FR_FB_SPEC_EVENT(SFX, MOV, REST, KB, "", ".Lfb_%=", ".Lfbd_%=") \
    ".Lfbr_%=:\n"  /* blocks run so far, this one included = n0 - (k + kend) */ \
    "s_add_u32 %[stmp], %[k], %[kend]\n"               \
    "s_sub_u32 %[stmp], %[n0], %[stmp]\n"              \
    "s_lshl_b32 %[stmp], %[stmp], " MSHIFT "\n"        \
    "s_add_u32 %[stmp], %[stmp], %[base]\n"            \
    "s_mov_b64 %[sprev], exec\n"                       \
    "s_mov_b64 exec, %[sdiff]\n"                       \
    "v_cvt_f32_u32 %[cnt], %[stmp]\n"                  \
    "s_mov_b64 exec, %[sprev]\n"                       \
    "s_cbranch_execnz .Lfbc_%=\n"                      \
    ".Lfbd_%=:\n"                                      \
    "s_mov_b64 %[srun], exec\n"                        \
    "s_mov_b64 exec, %[sorig]\n"

/* per-lane count (f32): %[n0] = the number of blocks to run (>= 1) */
#define FR_FBC_SPEC_ASM(SFX, MOV, BLOCK_ITS, STEP, REST, KB, SPECSTEP) \
    "s_mov_b64 %[sorig], exec\n"                       \
    "s_mov_b64 exec, %[mask]\n"                        \
    "s_mov_b32 %[kend], %[n0]\n"                       \
    FR_FB_SPEC_STRETCH                                 \
    ".Lfc_%=:\n" BLOCK_ITS                             \
    "v_add_" SFX " %[t], %[A], %[B]\n"                 \
    "v_add_f32 %[cnt], %[cnt], " STEP "\n"             \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"              \
    "s_cbranch_execz .Lfcd_%=\n"                       \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_cbranch_scc0 .Lfc_%=\n"                         \
    FR_FB_SPEC_EVENT(SFX, MOV, REST, KB, "v_add_f32 %[cnt], " SPECSTEP ", %[cnt]\n", ".Lfc_%=", ".Lfcd_%=") \
    ".Lfcd_%=:\n"                                      \
    "s_mov_b64 %[srun], exec\n"                        \
    "s_mov_b64 exec, %[sorig]\n"

#if FR_SPEC_M == 8
#define FR_SPEC_KB "2"
#define FR_SPEC_STEP "0x41000000" /* 8.0f: a literal, VOP2 takes it as src0 */
#elif FR_SPEC_M == 16
#define FR_SPEC_KB "4"
#define FR_SPEC_STEP "0x41800000" /* 16.0f: a literal, VOP2 takes it as src0 */
#else
#define FR_SPEC_KB "8"
#define FR_SPEC_STEP "0x42000000" /* 32.0f */
#endif

template <typename T, int M, bool SPEC = false>
__device__ __forceinline__ unsigned long long first_blocks(unsigned long long mask, uint32_t nblocks, uint32_t done_before, T &X, T &Y,
                                                           T &A, T &B, T &t, float &cnt, T c2re, T c2im,
                                                           typename UBits<T>::type t4lim, uint32_t spec_quiet_blocks = 0u) {
    T q;
    T X1, Y1, A1, B1; /* M == 4: the second register set of the speculative blocks */
    uint32_t specq = __builtin_amdgcn_readfirstlane(spec_quiet_blocks ? spec_quiet_blocks : 0xFFFFFFFFu); /* 0 = never; doubles with every block thrown away */
    unsigned long long sorig, srun, sprev, sdiff;
    (void)X1, (void)Y1, (void)A1, (void)B1, (void)specq;
    if constexpr (sizeof(T) == 4) {
        uint32_t kc = __builtin_amdgcn_readfirstlane(nblocks) - 1u;
        (void)done_before, (void)sdiff;
        if constexpr (M == 4 && SPEC) {
            const uint32_t nb = kc + 1u; /* the blocks to run; kc becomes the stretch counter */
            uint32_t kend;
            unsigned long long squiet;
            asm volatile(FR_FBC_SPEC_ASM("f32", "v_mov_b32", FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32"), "4.0",
                                         FR_SC_SPEC_REST, FR_SPEC_KB, FR_SPEC_STEP)
                         : [X] "+v"(X), [Y] "+v"(Y), [A] "+v"(A), [B] "+v"(B), [t] "+v"(t), [cnt] "+v"(cnt), [q] "=&v"(q),
                           [sorig] "=&s"(sorig), [srun] "=&s"(srun), [squiet] "=&s"(squiet), [k] "=&s"(kc), [kend] "=&s"(kend),
                           [X1] "=&v"(X1), [Y1] "=&v"(Y1), [A1] "=&v"(A1), [B1] "=&v"(B1), [specq] "+&s"(specq)
                         : [c2re] "v"(c2re), [c2im] "v"(c2im), [t4lim] "s"(t4lim), [mask] "s"(mask), [n0] "s"(nb)
                         : "vcc", "scc");
            return srun;
        }
        (void)sprev;
#define FR_FBC_OPERANDS                                                                                          \
    : [X] "+v"(X), [Y] "+v"(Y), [A] "+v"(A), [B] "+v"(B), [t] "+v"(t), [cnt] "+v"(cnt), [q] "=&v"(q),            \
      [sorig] "=&s"(sorig), [srun] "=&s"(srun), [k] "+s"(kc)                                                     \
    : [c2re] "v"(c2re), [c2im] "v"(c2im), [t4lim] "s"(t4lim), [mask] "s"(mask)                                   \
    : "vcc", "scc"
        if constexpr (M == 4)
            asm volatile(FR_FBC_ASM("f32", FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32"), "4.0") FR_FBC_OPERANDS);
        else
            asm volatile(FR_FBC_ASM("f32", FR_SC_IT("f32") FR_SC_IT("f32"), "2.0") FR_FBC_OPERANDS);
        return srun;
    }
    const uint32_t n0 = __builtin_amdgcn_readfirstlane(nblocks), base = __builtin_amdgcn_readfirstlane(done_before);
    uint32_t k = n0 - 1u, stmp;
#define FR_FB_OPERANDS                                                                                           \
    : [X] "+v"(X), [Y] "+v"(Y), [A] "+v"(A), [B] "+v"(B), [t] "+v"(t), [cnt] "+v"(cnt), [q] "=&v"(q),            \
      [sorig] "=&s"(sorig), [srun] "=&s"(srun), [sprev] "=&s"(sprev), [sdiff] "=&s"(sdiff), [k] "+s"(k),         \
      [stmp] "=&s"(stmp)                                                                                         \
    : [c2re] "v"(c2re), [c2im] "v"(c2im), [t4lim] "s"(t4lim), [mask] "s"(mask), [n0] "s"(n0), [base] "s"(base)   \
    : "vcc", "scc"
    if constexpr (sizeof(T) == 8) {
        if constexpr (M == 4 && SPEC)
        {
            uint32_t kend; /* blocks left behind the current stretch; k becomes the stretch counter */
            unsigned long long squiet;
            asm volatile(FR_FB_SPEC_ASM("f64", "v_mov_b64", FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64"), "2",
                                        FR_SC_SPEC_REST, FR_SPEC_KB)
                         : [X] "+v"(X), [Y] "+v"(Y), [A] "+v"(A), [B] "+v"(B), [t] "+v"(t), [cnt] "+v"(cnt), [q] "=&v"(q),
                           [sorig] "=&s"(sorig), [srun] "=&s"(srun), [sprev] "=&s"(sprev), [sdiff] "=&s"(sdiff), [k] "=&s"(k),
                           [stmp] "=&s"(stmp), [squiet] "=&s"(squiet), [kend] "=&s"(kend),
                           [X1] "=&v"(X1), [Y1] "=&v"(Y1), [A1] "=&v"(A1), [B1] "=&v"(B1), [specq] "+&s"(specq)
                         : [c2re] "v"(c2re), [c2im] "v"(c2im), [t4lim] "s"(t4lim), [mask] "s"(mask), [n0] "s"(n0), [base] "s"(base)
                         : "vcc", "scc");
        }
        else if constexpr (M == 4)
            asm volatile(FR_FB_ASM("f64", FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64"), "2") FR_FB_OPERANDS);
        else
            asm volatile(FR_FB_ASM("f64", FR_SC_IT("f64") FR_SC_IT("f64"), "1") FR_FB_OPERANDS);
    }
    return srun;
}

/* recursive()'s iteration with its escape test after every one, in the scaled form (4 * dist against 4 * limit^2:
 * the same comparison), for the lanes of `mask` (not empty): v_cmpx freezes a lane at `next`, which is what
 * recursive() returns; the count runs with the lane.  Ends after `n` iterations or when no lane is left; returns the
 * lanes that did not escape. */
#define FR_FS_ASM(SFX)                                 \
    "s_mov_b64 %[sorig], exec\n"                       \
    "s_mov_b64 exec, %[mask]\n"                        \
    ".Lfs_%=:\n" FR_SC_IT(SFX)                         \
    "v_add_" SFX " %[t], %[A], %[B]\n"                 \
    "v_add_f32 %[cnt], 1.0, %[cnt]\n"                  \
    "v_cmpx_nlt_" SFX " %[lim4], %[t]\n"               \
    "s_cbranch_execz .Lfsd_%=\n"                       \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_cbranch_scc0 .Lfs_%=\n"                         \
    ".Lfsd_%=:\n"                                      \
    "s_mov_b64 %[srun], exec\n"                        \
    "s_mov_b64 exec, %[sorig]\n"

template <typename T>
__device__ __forceinline__ unsigned long long finish_scaled(unsigned long long mask, uint32_t n, T &X, T &Y, T &A, T &B, T &t,
                                                            float &cnt, T c2re, T c2im, typename UBits<T>::type lim4) {
    T q;
    unsigned long long sorig, srun;
    uint32_t k = __builtin_amdgcn_readfirstlane(n) - 1u;
#define FR_FS_OPERANDS                                                                                           \
    : [X] "+v"(X), [Y] "+v"(Y), [A] "+v"(A), [B] "+v"(B), [t] "+v"(t), [cnt] "+v"(cnt), [q] "=&v"(q),            \
      [sorig] "=&s"(sorig), [srun] "=&s"(srun), [k] "+s"(k)                                                      \
    : [c2re] "v"(c2re), [c2im] "v"(c2im), [lim4] "s"(lim4), [mask] "s"(mask)                                     \
    : "vcc", "scc"
    if constexpr (sizeof(T) == 8)
        asm volatile(FR_FS_ASM("f64") FR_FS_OPERANDS);
    else
        asm volatile(FR_FS_ASM("f32") FR_FS_OPERANDS);
    return srun;
}

/* The common case of a tile, start to finish, in one asm block — a full tile of a strip that may use the scaled
 * form, no start beyond the limit: the first episode (`nblk` blocks of M unchecked iterations) and, if every lane has
 * frozen by its end (three tiles in five on C4 are gone after ONE block), their exact last iterations.  The scalar
 * unit is a co-limiter of the f32 render (one scalar instruction per ~4 cycles per SIMD against ~2.4 for an f32
 * vector one; profiles/r03_c4_first_pass_classes.txt): on this path the scalar work is the loops' own control
 * and a dozen instructions besides.  Returns how far it got; the state is valid for the general path to go on from:
 *   0  every lane escaped (state = `next`, cnt = its index + 1)
 *   1  lanes still run after the first episode (`srun`): further episodes or a hand-over
 *   2  nothing was done: some start lies beyond the LIMIT (a start merely beyond T stays here, frozen from the
 *      start with a count of 0: the exact loop takes it from iteration 0, and the test "did its last iteration
 *      escape" cannot fire for it because its |z|^2 is within the limit)
 *   3  64 exact iterations did not finish everybody (`srun` = the lanes still live) */
#define FR_SC_IT0(SFX)                             \
    "v_add_" SFX " %[t], %[A], -%[B0]\n"           \
    "v_mul_" SFX " %[q], %[X], %[Y0]\n"            \
    "v_fma_" SFX " %[X], %[t], 0.5, %[c2re]\n"     \
    "v_add_" SFX " %[Y], %[q], %[c2im]\n"          \
    "v_mul_" SFX " %[A], %[X], %[X]\n"             \
    "v_mul_" SFX " %[B], %[Y], %[Y]\n"
/* REST_ITS = the block's iterations after its first; the first block's first iteration reads Y0 / B0 where they
 * lie (no copies) and its count is set, not added */
#define FR_TILE_ASM(SFX, MOVT, REST_ITS, STEP)         \
    "s_load_dwordx2 %[fa], %[kargs], %[offa]\n"        \
    "s_load_dwordx2 %[fb], %[kargs], %[offb]\n"        \
    "s_load_dwordx2 %[fk], %[kargs], %[offk]\n"        \
    "v_add_" SFX " %[X], %[sre], %[sre]\n"             \
    "v_mul_" SFX " %[A], %[X], %[X]\n"                 \
    "v_add_" SFX " %[t], %[A], %[B0]\n"                \
    "v_mov_b32 %[cnt], 0\n"                            \
    "s_mov_b32 %[st], 2\n"                             \
    "v_cmp_lt_" SFX " vcc, %[lim4], %[t]\n"            \
    "s_cbranch_vccnz .Ltout_%=\n"                      \
    "s_mov_b64 %[sorig], exec\n"                       \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"              \
    "s_xor_b64 %[srun], %[sorig], exec\n"              \
    "s_cbranch_scc0 .Ltgo_%=\n"                        \
    "s_mov_b64 vcc, exec\n"                            \
    "s_mov_b64 exec, %[srun]\n"                        \
    MOVT " %[Y], %[Y0]\n"                              \
    MOVT " %[B], %[B0]\n"                              \
    "s_mov_b64 exec, vcc\n"                            \
    "s_cbranch_execz .Ltnone_%=\n"                     \
    ".Ltgo_%=:\n"                                      \
    FR_SC_IT0(SFX) REST_ITS                            \
    "v_add_" SFX " %[t], %[A], %[B]\n"                 \
    "v_mov_b32 %[cnt], " STEP "\n"                     \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"              \
    "s_cbranch_execz .Ltbd_%=\n"                       \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_cbranch_scc1 .Ltbd_%=\n"                        \
    ".Ltb_%=:\n" FR_SC_IT(SFX) REST_ITS                \
    "v_add_" SFX " %[t], %[A], %[B]\n"                 \
    "v_add_f32 %[cnt], %[cnt], " STEP "\n"             \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"              \
    "s_cbranch_execz .Ltbd_%=\n"                       \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_cbranch_scc0 .Ltb_%=\n"                         \
    ".Ltbd_%=:\n"                                      \
    "s_mov_b64 %[srun], exec\n"                        \
    "s_mov_b64 exec, %[sorig]\n"                       \
    "s_mov_b32 %[st], 1\n"                             \
    "s_cmp_lg_u64 %[srun], 0\n"                        \
    "s_cbranch_scc1 .Ltout_%=\n"                       \
    ".Ltnone_%=:\n"                                    \
    "s_mov_b64 exec, %[sorig]\n"                       \
    "s_mov_b32 %[st], 0\n"                             \
    "v_cmpx_nlt_" SFX " %[lim4], %[t]\n"               \
    "s_cbranch_execz .Ltfd_%=\n"                       \
    "s_mov_b32 %[k], 63\n"                             \
    ".Ltf_%=:\n" FR_SC_IT(SFX)                         \
    "v_add_" SFX " %[t], %[A], %[B]\n"                 \
    "v_add_f32 %[cnt], 1.0, %[cnt]\n"                  \
    "v_cmpx_nlt_" SFX " %[lim4], %[t]\n"               \
    "s_cbranch_execz .Ltfd_%=\n"                       \
    "s_sub_u32 %[k], %[k], 1\n"                        \
    "s_cbranch_scc0 .Ltf_%=\n"                         \
    "s_mov_b32 %[st], 3\n"                             \
    ".Ltfd_%=:\n"                                      \
    "s_mov_b64 %[srun], exec\n"                        \
    "s_mov_b64 exec, %[sorig]\n"                       \
    ".Ltout_%=:\n"                                     \
    "s_waitcnt lgkmcnt(0)\n"

template <typename T, int M>
__device__ __forceinline__ uint32_t tile_fast(uint32_t nblk, T sre, T Y0, T B0, T c2re, T c2im, typename UBits<T>::type t4lim,
                                              typename UBits<T>::type lim4, KArgs kargs, T &X, T &Y, T &A, T &B, T &t, float &cnt,
                                              unsigned long long &srun, unsigned long long &fa, unsigned long long &fb, unsigned long long &fk) {
    static_assert(offsetof(fr_kparams, filt_lo32) == offsetof(fr_kparams, prim32) + 12 && offsetof(fr_kparams, prim32) % 4 == 0 &&
                      offsetof(fr_kparams, filt_c32) == offsetof(fr_kparams, filt_k32) + 4,
                  "prim32[0..1] | prim32[2], filt_lo32 | filt_k32, filt_c32: three 8-byte loads");
    T q;
    unsigned long long sorig;
    uint32_t st, k = nblk - 1u;
#define FR_TILE_OPERANDS                                                                                         \
    : [X] "=&v"(X), [Y] "=&v"(Y), [A] "=&v"(A), [B] "=&v"(B), [t] "=&v"(t), [cnt] "=&v"(cnt), [q] "=&v"(q),      \
      [sorig] "=&s"(sorig), [srun] "=&s"(srun), [k] "+s"(k), [st] "=&s"(st), [fa] "=&s"(fa), [fb] "=&s"(fb),     \
      [fk] "=&s"(fk)                                                                                             \
    : [sre] "v"(sre), [Y0] "v"(Y0), [B0] "v"(B0), [c2re] "v"(c2re), [c2im] "v"(c2im), [t4lim] "s"(t4lim),        \
      [lim4] "s"(lim4), [kargs] "s"(kargs), [offa] "i"(offsetof(fr_kparams, prim32)),                            \
      [offb] "i"(offsetof(fr_kparams, prim32) + 8),                                                              \
      [offk] "i"(offsetof(fr_kparams, filt_k32))                                                                 \
    : "vcc", "scc", "memory"
    if constexpr (sizeof(T) == 8) {
        if constexpr (M == 4)
            asm volatile(FR_TILE_ASM("f64", "v_mov_b64", FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64"), "4.0") FR_TILE_OPERANDS);
        else
            asm volatile(FR_TILE_ASM("f64", "v_mov_b64", FR_SC_IT("f64"), "2.0") FR_TILE_OPERANDS);
    } else {
        if constexpr (M == 4)
            asm volatile(FR_TILE_ASM("f32", "v_mov_b32", FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32"), "4.0") FR_TILE_OPERANDS);
        else
            asm volatile(FR_TILE_ASM("f32", "v_mov_b32", FR_SC_IT("f32"), "2.0") FR_TILE_OPERANDS);
    }
    return st;
}

/* ds_bpermute of a T (lane `byte_index / 4`'s value), the index a ready-made byte offset */
template <typename T>
__device__ __forceinline__ typename UBits<T>::type bpermute_t(uint32_t byte_index, T v) {
    if constexpr (sizeof(T) == 8) {
        const uint64_t b = fr_bits_of(v);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_index, (int)(uint32_t)b);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_index, (int)(uint32_t)(b >> 32));
        return (uint64_t)lo | ((uint64_t)hi << 32);
    } else {
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_index, __builtin_bit_cast(int, v));
    }
}

template <typename T, int M, int kStripTiles, int kBands, bool SPEC = false>
__global__ __launch_bounds__(64) void escape_first_kernel(const fr_kparams p, const fr_kout out) {
    __shared__ uint32_t s_palette[FR_MAX_PALETTE_ENTRIES]; /* smooth == false only; the log2 table stays in L2 (see v1) */
    const double *const s_tab = &g_log2_tab[0][0];
    typedef typename Pair<T>::type T2;
    typedef typename UBits<T>::type UB;
    static_assert(kStripTiles <= 8 && kBands <= 8, "one column / one row coordinate per lane");
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t *s_pal = nullptr;
    uint32_t row0, tile0, ncols, nrows, bpp, lane_pitch;
    T colsT, rowsT; /* the block's coordinate map (calc/src/lib.rs:181-197), narrowed once: column tile0*8 + lane, row row0 + lane */
    bool cols_scalable;
    unsigned long long bad_rows;
    bool narrow;
    bool fast_colour; /* the f32 stage of the colour filter applies to this render (wave-uniform, per launch) */
    {
        FR_COLD_PARAMS(kp);
        const auto &P = *kp;
        if (P.palette != nullptr) {
            const uint32_t n = P.palette_entries;
            const uint32_t *src = P.palette;
            for (uint32_t k = lane; k < n; k += 64) s_palette[k] = src[k];
            s_pal = s_palette;
            __syncthreads();
        }
        row0 = (blockIdx.y + gridDim.y * blockIdx.z) * (8u * kBands);
        nrows = P.nrows, ncols = P.ncols;
        if (row0 >= nrows) return;
        const double width = (double)P.width, height = (double)P.height;
        tile0 = blockIdx.x * kStripTiles;
        const uint32_t block_rows = P.block_rows;
        const uint32_t x = P.x_first + (tile0 * 8u + lane) * P.x_stride;
        const double cols = coord_to_space((double)x, height, (width / height) / 2.0, P.pos_re, P.scale_re);
        const uint32_t rr = row0 + lane;
        const uint32_t y = P.y_first + (rr / block_rows) * P.y_stride + rr % block_rows;
        const double rows = coord_to_space((double)y, height, 0.5, P.pos_im, P.scale_im);
        colsT = (T)cols, rowsT = (T)rows;
        bpp = P.out_rgba ? 4u : 3u;
        narrow = (uint64_t)ncols * bpp * 8u <= 0xFFFFFFFFull;
        lane_pitch = ly * (ncols * bpp) + lx * bpp;
        const bool is_julia = P.algo == 2;
        bool c_ok = true;
        if (is_julia) {
            constexpr T lo = ScalableRange<T>::lo, hi = ScalableRange<T>::hi;
            const T jr = __builtin_fabs((T)P.julia_re), ji = __builtin_fabs((T)P.julia_im);
            c_ok = jr >= lo && jr <= hi && ji >= lo && ji <= hi;
        }
        const bool col_relevant = lane < 8u * kStripTiles && tile0 * 8u + lane < ncols;
        const bool row_relevant = lane < 8u * kBands && rr < nrows;
        cols_scalable = c_ok && __ballot(col_relevant && !coord_is_scalable<T>(is_julia, cols)) == 0ull;
        bad_rows = __ballot(row_relevant && !coord_is_scalable<T>(is_julia, rows));
        /* colour_multiply's RGB::new(r, b, g) swap (calc/src/lib.rs:129-139): output channel k takes field {0, 2, 1}[k] */
        fast_colour = P.colour_filter32 && P.smooth && P.palette == nullptr;
    }
    /* the host guarantees 0 < k1 < cap < 2^24; first_only: no tile is ever handed over (keep = 0), every lane finishes here */
    const uint32_t k1 = p.first_cap, cap = p.iterations, keep = p.first_only ? 0u : p.first_keep;
    const bool julia = p.algo == 2;
    const T jre = (T)p.julia_re, jim = (T)p.julia_im;
    const T squared = sizeof(T) == 8 ? (T)(p.limit * p.limit) : (T)((float)p.limit * (float)p.limit);
    const T skip_t = (T)p.skip_t;
    const T t4v = (T)4 * skip_t, lim4v = (T)4 * squared;
    const UB t4lim = uniform_bits<T>(t4v), lim4 = uniform_bits<T>(lim4v);
    T c2re = jre + jre, c2im = jim + jim; /* Julia: c = julia_set (calc/src/lib.rs:209-210); Mandelbrot: set per tile */
    /* the asm path runs a whole first episode and up to 64 exact iterations without looking at the cap */
    const bool fast_tiles = k1 % (uint32_t)M == 0u && k1 + 64u <= cap;
    const uint32_t nblk1 = k1 / (uint32_t)M;
    /* the strip's tiles: `ntiles` lie (partly) inside the image, the first `nfull` of them with all 8 columns */
    const uint32_t cols_left = ncols - tile0 * 8u; /* > 0: the grid has no workgroup past the right edge */
    const uint32_t ntiles = (cols_left + 7u) / 8u < (uint32_t)kStripTiles ? (cols_left + 7u) / 8u : (uint32_t)kStripTiles;
    const uint32_t nfull = cols_left / 8u < (uint32_t)kStripTiles ? cols_left / 8u : (uint32_t)kStripTiles;
    const uint32_t lane_x4 = lx * 4u; /* ds_bpermute index of this lane's column within tile 0 of the strip */

  for (int band = 0; band < kBands; band++) {
    const uint32_t rb = row0 + 8u * (uint32_t)band;
    if (rb >= nrows) break; /* wave-uniform */
    const T sim = __shfl(rowsT, band * 8 + (int)ly, 64);
    const T Y0 = sim + sim, B0 = Y0 * Y0;
    const bool strip_scalable = cols_scalable && ((bad_rows >> (8 * band)) & 0xFFull) == 0ull;
    const bool rows_full = rb + 8u <= nrows;
    uint32_t out_row0 = rb;
    {
        FR_COLD_PARAMS(kp);
        if (kp->out_in_place) out_row0 = kp->y_first + (rb / kp->block_rows) * kp->y_stride + rb % kp->block_rows;
    }
    const uint32_t r_out = out_row0 + ly;
    uint8_t *const strip_base = out.rgb + ((uint64_t)out_row0 * ncols + (uint64_t)tile0 * 8u) * bpp;
    const uint32_t r = rb + ly;
    const uint32_t list = (blockIdx.x + 5u * (rb >> 3)) & (FR_SURV_QUEUES - 1u); /* neighbouring strips: different lists */
    /* tiles [0, nfast) of this strip may try the asm path */
    const uint32_t nfast = (fast_tiles && strip_scalable && rows_full && narrow) ? nfull : 0u;

    for (uint32_t k = 0; k < ntiles; k++) {
        const uint32_t col0 = (tile0 + k) * 8u;
        const T sre = __builtin_bit_cast(T, bpermute_t<T>(lane_x4 + k * 32u, colsT));
        const uint32_t cx = col0 + lx;
        const bool full = rows_full && k < nfull;
        unsigned long long valid_m = ~0ull;
        if (!full) valid_m = __ballot(cx < ncols && r < nrows);
        /* where the tile's pixels go: a wave-uniform base and one 32-bit byte offset per lane (narrow images) */
        uint8_t *const tile_base = strip_base + (size_t)(k * 8u * bpp);
        unsigned long long fin; /* the lanes whose pixel is coloured and stored here */
        uint32_t packed = 0;    /* r | g << 8 | b << 16 */
        bool have_colour = false;
        if (strip_scalable) {
            T X, Y, A, B, t;
            float cnt;
            unsigned long long run = 0ull, over0 = 0ull, esc = 0ull, live = 0ull;
            uint32_t done = 0, st = 2;
            if (!julia) c2re = sre + sre, c2im = Y0; /* Mandelbrot: c = start */
            if (k < nfast) {
                /* the filter's scalars: re-read from the kernel-argument segment HERE, before the orbit loop that hides
                 * the loads' latency, instead of being held in scalar registers across the workgroup (they were
                 * spilled to vector lanes and every tile paid 16 v_readlane — vector-issue slots — to get them back):
                 * tile_fast issues the two loads before its loops and waits for them behind them */
                unsigned long long fa, fb, fk;
                st = tile_fast<T, M>(nblk1, sre, Y0, B0, c2re, c2im, t4lim, lim4, (KArgs)__builtin_amdgcn_kernarg_segment_ptr(), X, Y, A,
                                     B, t, cnt, run, fa, fb, fk);
                Filter32 fc;
                fc.lo = __builtin_bit_cast(float, (uint32_t)(fb >> 32));
                fc.k = __builtin_bit_cast(float, (uint32_t)fk), fc.c = __builtin_bit_cast(float, (uint32_t)(fk >> 32));
                /* colour_multiply's RGB::new(r, b, g) swap: output channel k takes stored field {0, 2, 1}[k] */
                fc.p0 = __builtin_bit_cast(float, (uint32_t)fa), fc.p1 = __builtin_bit_cast(float, (uint32_t)fb);
                fc.p2 = __builtin_bit_cast(float, (uint32_t)(fa >> 32));
                /* The common case to its end, apart from everything else (no state shared with the general path
                 * below, so nothing is merged or copied for it): every lane escaped, and the filter's first stage
                 * decides every lane's bytes from the f32 squared distance — (A + B) / 4 IS fl(re^2 + im^2), see
                 * colour_pixel; f64 renders round it to f32 as the filter always did — with cnt = iters + 1. */
                if (st == 0u && fast_colour) {
                    const float d32 = (float)t * 0.25f;
                    const bool unsure = !(d32 >= fc.lo && d32 <= 0x1.ffffep119f);
                    if (__ballot(unsure) == 0ull) {
                        uint32_t pk;
                        const bool decided = colour_fast32(fc, d32, cnt, pk);
                        if (__ballot(!decided) == 0ull) {
                            store_packed(tile_base, lane_pitch, pk, bpp);
                            continue;
                        }
                    }
                }
            }
            if (st == 0u) {
                fin = esc = ~0ull;
            } else {
                /* ---- the general path; picks the tile up where the asm path left it */
                if (st == 2u) {
                    X = sre + sre, Y = Y0, A = X * X, B = B0, t = A + B, cnt = 0.0f;
                    /* a start already past T never enters the unchecked blocks: it goes to the exact loop, from iteration 0 */
                    over0 = __ballot(t > t4v) & valid_m;
                    run = valid_m & ~over0;
                }
                unsigned long long handed = 0ull;
                if (st != 3u) {
                    /* episodes of k1 iterations for as long as the tile is worth a wave of its own — at least `keep`
                     * lanes still running; a tile still here after 8 episodes is most likely inside a filled set: from
                     * then on every episode is twice the last, up to 16 x k1 (doubling from the third episode on was
                     * tried: C4's dense tiles thin out within an episode or two and idled through the longer ones) */
                    uint32_t len = k1;
                    bool first = st == 1u; /* the asm path has run the first episode */
                    /* SPEC (the host launches this form when the plan allows speculation and the view's statistics, where
                     * there are any, do not say that nothing stays): quiet blocks before a wave speculates in first_blocks, read
                     * from the kernel-argument segment HERE, where tiles that stay arrive, not held in a scalar register across
                     * the workgroup (the kernel has none to spare) */
                    uint32_t spec_blocks = 0;
                    if constexpr (SPEC) {
                        FR_COLD_PARAMS(kp);
                        spec_blocks = kp->loop_spec / (uint32_t)M;
                    }
                    while (run != 0ull) {
                        if (!first) {
                            const uint32_t left = cap - done;
                            const uint32_t nblk = (left < len ? left : len) / (uint32_t)M;
                            if (nblk == 0u) break; /* fewer than M iterations to the cap: the exact loop below runs them */
                            if constexpr (SPEC)
                                run = first_blocks<T, M, true>(run, nblk, done, X, Y, A, B, t, cnt, c2re, c2im, t4lim, spec_blocks);
                            else
                                run = first_blocks<T, M>(run, nblk, done, X, Y, A, B, t, cnt, c2re, c2im, t4lim);
                            done += nblk * (uint32_t)M;
                        } else {
                            done = k1;
                            first = false;
                        }
                        if (run == 0ull || cap - done < (uint32_t)M) break;
                        if ((uint32_t)__builtin_popcountll(run) < keep) break;
                        if (done >= 8u * k1 && len < 16u * k1) len += len;
                    }
                    if (st == 1u && done == 0u) done = k1;
                    /* lanes that ran through those episodes carry no count of their own (first_blocks): it is `done` */
                    if (run != 0ull) cnt = lane_in(run) ? (float)done : cnt;
                    /* hand the running lanes over: position after `done` iterations, recursive()'s own state */
                    if (run != 0ull && cap - done >= (uint32_t)M) {
                        FR_COLD_PARAMS(kp);
                        const uint32_t sub_cap = kp->surv_sub_capacity;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(kp->surv_counts + list * FR_SURV_COUNT_STRIDE, (uint32_t)__builtin_popcountll(run));
                        base = __builtin_amdgcn_readfirstlane(base);
                        const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(run >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)run, 0u));
                        const bool mine = lane_in(run) && slot < sub_cap; /* a full list is not an error: those lanes finish here */
                        if (mine && !(kp->debug_ablate & 1u)) {
                            const size_t e = (size_t)list * sub_cap + slot;
                            T2 zz;
                            zz.x = X * (T)0.5, zz.y = Y * (T)0.5; /* exact */
                            static_cast<T2 *>(kp->surv_z)[e] = zz;
                            reinterpret_cast<uint2 *>(kp->surv_pos)[e] = make_uint2(cx, r_out);
                            kp->surv_cnt[e] = done;
                            if (!julia) {
                                T2 cc2;
                                cc2.x = sre, cc2.y = sim;
                                static_cast<T2 *>(kp->surv_c)[e] = cc2;
                            }
                        }
                        handed = __ballot(mine);
                    }
                    fin = valid_m & ~handed;
                    if (fin == 0ull) continue; /* the whole tile went to the lists */
                    /* `esc` = lanes whose last iteration escaped (their count - 1 is the index): tested AFTER an iteration
                     * only, so not for a start past T; NaN: no */
                    esc = __ballot(t > lim4v) & fin & ~over0;
                    live = fin & ~esc;
                } else {
                    fin = ~0ull, live = run, esc = ~run, done = k1 + 64u;
                }
                /* the exact loop, once per tile, for every lane that is not done: the frozen ones (4-5 iterations from
                 * limit^2), starts past T, the running lanes of a tile that reached the cap's last M - 1 iterations or
                 * found its list full */
                uint32_t bound = done; /* no lane's count exceeds it */
                while (live != 0ull) {
                    uint32_t nf = 64u;
                    if (bound + 64u > cap) {
                        /* near the cap the lanes' room differs: the largest count decides, lanes at the cap retire */
                        live &= ~__ballot((uint32_t)cnt >= cap);
                        if (live == 0ull) break;
                        nf = cap - wave_max_u32(lane_in(live) ? (uint32_t)cnt : 0u);
                        bound = cap - nf;
                    }
                    const unsigned long long still = finish_scaled<T>(live, nf, X, Y, A, B, t, cnt, c2re, c2im, lim4);
                    esc |= live & ~still;
                    live = still;
                    bound += nf;
                }
            }
            /* ---- colour (calc/src/lib.rs:214-234).  The filter's first stage from the f32 squared distance (see
             * colour_pixel): (A + B) / 4 IS fl(re^2 + im^2) — f64 renders round it to f32 as the filter always did —
             * and cnt = iters + 1 for an escaped lane */
            if (fast_colour && esc == fin) { /* wave-uniform */
                const float d32 = (float)t * 0.25f;
                Filter32 f32c;
                {
                    FR_COLD_PARAMS(kp);
                    f32c.lo = kp->filt_lo32, f32c.k = kp->filt_k32, f32c.c = kp->filt_c32;
                    f32c.p0 = kp->prim32[0], f32c.p1 = kp->prim32[2], f32c.p2 = kp->prim32[1];
                }
                const bool unsure = !(d32 >= f32c.lo && d32 <= 0x1.ffffep119f);
                if ((__ballot(unsure) & fin) == 0ull) {
                    const bool decided = colour_fast32(f32c, d32, cnt, packed);
                    have_colour = (__ballot(!decided) & fin) == 0ull;
                }
            }
            if (!have_colour) {
                const T re = X * (T)0.5, im = Y * (T)0.5; /* exact */
                const T r2 = re * re, i2 = im * im;       /* the reference's own re*re, im*im */
                const uint32_t iters = lane_in(esc) ? (uint32_t)cnt - 1u : cap;
                if (lane_in(fin)) {
                    FR_COLD_PARAMS(kp);
                    const ColourConsts cc = make_colour_consts(*kp);
                    uint8_t rgb[3];
                    colour_pixel<T>(cc, re, im, r2, i2, iters, s_tab, s_pal, rgb);
                    packed = (uint32_t)rgb[0] | ((uint32_t)rgb[1] << 8) | ((uint32_t)rgb[2] << 16);
                }
            }
        } else {
            /* a strip that may not use the scaled form: the plain loop from the start, the general colour path */
            fin = valid_m;
            T re = sre, im = sim, r2 = 0, i2 = 0;
            const T cre = julia ? jre : re, cim = julia ? jim : im;
            if (lane_in(valid_m)) {
                const uint32_t iters = orbit<T>(cap, re, im, cre, cim, squared, r2, i2);
                FR_COLD_PARAMS(kp);
                const ColourConsts cc = make_colour_consts(*kp);
                uint8_t rgb[3];
                colour_pixel<T>(cc, re, im, r2, i2, iters, s_tab, s_pal, rgb);
                packed = (uint32_t)rgb[0] | ((uint32_t)rgb[1] << 8) | ((uint32_t)rgb[2] << 16);
            }
        }
        /* ---- store (src/lib.rs:253-270) */
        if (fin == ~0ull || lane_in(fin)) {
            if (narrow) {
                store_packed(tile_base, lane_pitch, packed, bpp);
            } else {
                uint8_t *o = out.rgb + ((uint64_t)r_out * ncols + cx) * bpp;
                if (bpp == 4u) {
                    *reinterpret_cast<uint32_t *>(o) = packed | 0xFF000000u;
                } else {
                    o[0] = (uint8_t)packed, o[1] = (uint8_t)(packed >> 8), o[2] = (uint8_t)(packed >> 16);
                }
            }
        }
    }
  } /* band */
}

/* ---- strip kernel with lane refill ------------------------------------------------------------
 *
 * One wave renders a patch of 7 x 2 tiles (56 x 16 pixels), and a lane whose pixel has finished
 * (escaped, or reached the cap) is handed the next unstarted pixel of the patch instead of idling
 * until the slowest lane of its 8x8 tile is done.  Views without a large interior (Julia sets, zoomed exteriors) leave two thirds of
 * the lanes idle with one tile per wave (measured useful-lane fraction of C4: 0.35).
 *
 * The orbit loops run as EPISODES (orbit_run / orbit_scaled_run): an episode ends when every
 * running lane has escaped, when the lane closest to its cap reaches it (n = min over running lanes
 * of iterations - done), or — while unstarted pixels remain — when a set fraction of the running lanes
 * have finished and at least refill_minrun iterations were done.  Between episodes the finished lanes
 * are coloured and stored together and refilled; every lane keeps `done`, its own iteration count.
 * refill_minrun keeps tiles that escape within a few dozen iterations (most of a Mandelbrot
 * exterior) on the cheap path: one episode, one colour pass per 64 pixels, exactly as without
 * refill.  Results are independent of the schedule: a pixel's orbit never depends on its lane. */

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

/* Periodicity check: Brent's schedule runs inside orbit_scaled_run (a save when a run's iteration
 * count passes 32, 64, 128, ...; every run restarts the schedule, which keeps it valid — any earlier
 * state of the same orbit will do — and gives freshly refilled pixels their first save after 32 of
 * their own iterations); the lane compares against its latest save at every block end. */

template <typename T>
__device__ __forceinline__ T cycle_none() { /* a state no orbit passes through: NaN bits */
    if constexpr (sizeof(T) == 8)
        return __builtin_bit_cast(double, 0x7FF8000000000001ull);
    else
        return __builtin_bit_cast(float, 0x7FC00001u);
}

/* The refilling kernel's pool of pixels is a PATCH of kStripTiles x kRefillBands tiles: the larger
 * the pool a wave draws from, the fewer lanes idle while the last long orbits of the pool finish
 * (simulated useful-lane fraction on C4: 0.58 with one 7-tile strip, 0.67 with 2 bands, 0.74 with 8) —
 * but the fewer, longer-lived waves there are to balance over the chip.  Measured on C4 (f32):
 * 1 band 3.65 ms, 2 bands 3.48, 4 bands 3.53, 8 bands 3.97; 2 it is (56 x 16 pixels).  Column coordinates live one per lane in `cols` (lanes 0-55), row coordinates in `rows`
 * (lanes 0-63), the rows' output positions in `out_rows`; pixel ids run tile-major:
 * pid = ((band * kStripTiles + tile) * 64) + ly * 8 + lx. */
constexpr int kRefillBands = 2;

template <typename T, int MODE, int kStripTiles, int FORM, bool CYC>
__device__ __forceinline__ void refill_patch(const fr_kparams &p, const fr_kout &out, const double *s_tab,
                                             const uint32_t *s_pal, double cols, double rows, uint32_t out_rows,
                                             uint32_t tile0, uint32_t row0, uint32_t lane) {
    static_assert(!(CYC && FORM == 0), "the periodicity check lives in the scaled loops");
    constexpr uint32_t P = kStripTiles * kRefillBands * 64;
    const ColourConsts cc = make_colour_consts(p);
    const bool julia = p.algo == 2;
    const T squared = sizeof(T) == 8 ? (T)(p.limit * p.limit) : (T)((float)p.limit * (float)p.limit);
    const T skip_t = (T)p.skip_t;
    const uint32_t iterations = p.iterations;

    T a0 = 0, a1 = 0, a2 = 0, a3 = 0; /* FORM 0: re, im, re*re, im*im;  scaled: X, Y, A, B */
    T c0 = 0, c1 = 0;                 /* FORM 0: c.re, c.im;            scaled: 2c.re, 2c.im */
    T xs = cycle_none<T>(), ys = cycle_none<T>(); /* CYC: this orbit's state after `saved_at` iterations */
    uint32_t done = 0, next = 0, saved_at = 0;
    uint32_t px = 0, py = 0, pout = 0; /* the lane's pixel: local column, local row, output row */
    bool busy = false;
    unsigned long long count_acc = 0;

    while (true) {
        /* ---- hand unstarted pixels to the free lanes */
        const unsigned long long free_mask = __ballot(!busy);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)free_mask, 0u));
        const uint32_t cand = (busy || next + rank >= P) ? 0u : next + rank;
        const uint32_t ct = cand >> 6, cband = ct / (uint32_t)kStripTiles, ctile = ct - cband * (uint32_t)kStripTiles;
        const uint32_t ccol = ctile * 8u + (cand & 7u), crow = cband * 8u + ((cand >> 3) & 7u);
        /* cross-lane reads by ALL lanes (a masked-off source lane would not deliver its value) */
        const double sre = __shfl(cols, ccol, 64);
        const double sim = __shfl(rows, crow, 64);
        const uint32_t sout = __shfl(out_rows, crow, 64);
        if (!busy && next + rank < P) {
            const uint32_t cx = tile0 * 8u + ccol, r = row0 + crow;
            if (cx < p.ncols && r < p.nrows) {
                const T zre = (T)sre, zim = (T)sim;
                const T cre = julia ? (T)p.julia_re : zre, cim = julia ? (T)p.julia_im : zim; /* :209-210 */
                if constexpr (FORM == 0) {
                    a0 = zre, a1 = zim, a2 = zre * zre, a3 = zim * zim, c0 = cre, c1 = cim;
                } else {
                    a0 = zre + zre, a1 = zim + zim, a2 = a0 * a0, a3 = a1 * a1, c0 = cre + cre, c1 = cim + cim;
                }
                if constexpr (CYC) xs = ys = cycle_none<T>(), saved_at = 0; /* the previous pixel's save is not ours */
                px = cx, py = r, pout = sout;
                done = 0;
                busy = true;
            }
        }
        const uint32_t nfree = (uint32_t)__builtin_popcountll(free_mask);
        next = next + nfree < P ? next + nfree : P;
        const unsigned long long busy_mask = __ballot(busy);
        if (busy_mask == 0ull) {
            if (next >= P) break;
            continue; /* only out-of-image pixels were handed out; take the next ones */
        }

        /* ---- one episode: run until the lane closest to its cap gets there (or the policy quits) */
        const uint32_t n = wave_min_u32(busy ? iterations - done : 0xFFFFFFFFu);
        const uint32_t nbusy = (uint32_t)__builtin_popcountll(busy_mask);
        EpisodeCtl ctl{0u, 0u};
        if (next < P) ctl = EpisodeCtl{nbusy - (nbusy * p.refill_quit16 + 15) / 16, p.refill_minrun};
        uint32_t it = 0, completed = 0, saved_index = 0xFFFFFFFFu;
        if (busy) {
            if constexpr (FORM == 0)
                it = orbit_run<T>(n, a0, a1, c0, c1, squared, a2, a3, ctl, completed, p.loop_spec);
            else
                it = orbit_scaled_run<T, FORM, CYC>(n, a0, a1, a2, a3, c0, c1, squared, skip_t, ctl, completed, &xs, &ys,
                                                    &saved_index, p.loop_spec);
        }
        /* ---- retire the lanes that finished */
        if (busy) {
            bool cycled = false;
            if constexpr (CYC) cycled = (it & 0x80000000u) != 0u;
            const bool escaped = !cycled && it < completed;
            const uint32_t before = done;
            if constexpr (CYC) {
                /* the run saved this lane's state (xs, ys) after `saved_index` of its iterations */
                if (saved_index != 0xFFFFFFFFu) saved_at = before + saved_index;
            }
            if (cycled) {
                /* Back, after before + k iterations, at the state it had after saved_at iterations: the
                 * orbit is periodic with a period dividing d = before + k - saved_at, and no state of
                 * the cycle escapes (all were visited without escaping).  recursive() would go on to
                 * the cap; the state there is the one r = (remaining) mod d steps ahead. */
                const uint32_t k = it & 0x7FFFFFFFu;
                const uint32_t d = before + k - saved_at;
                const uint32_t remaining = iterations - (before + k);
                done = iterations - remaining % d;
                xs = ys = cycle_none<T>(); /* no second detection: it would only re-derive this */
            } else {
                done += completed;
            }
            if (escaped || done == iterations) {
                const uint32_t iters = escaped ? before + it : iterations;
                T fre, fim, fr2, fi2;
                if constexpr (FORM == 0) {
                    fre = a0, fim = a1, fr2 = a2, fi2 = a3;
                } else {
                    fre = a0 * (T)0.5, fim = a1 * (T)0.5; /* exact */
                    fr2 = fre * fre, fi2 = fim * fim;
                }
                const double zre = (double)fre, zim = (double)fim;
                const double dist = sizeof(T) == 8 ? (double)(fr2 + fi2) : zre * zre + zim * zim; /* :214 */
                if constexpr (MODE == FR_OUT_RGB) {
                    uint8_t rgb[3];
                    colour_of(cc, dist, iters, s_tab, s_pal, rgb);
                    store_pixel(p, out.rgb, pout, px, rgb);
                } else if constexpr (MODE == FR_OUT_ESCAPE) {
                    const uint64_t kk = (uint64_t)py * p.ncols + px;
                    if (out.z) {
                        out.z[2 * kk] = zre;
                        out.z[2 * kk + 1] = zim;
                    }
                    if (out.iters) out.iters[kk] = iters;
                } else {
                    count_acc += iters < iterations ? (unsigned long long)iters + 1ull : iterations;
                }
                busy = false;
            }
        }
    }
    if constexpr (MODE == FR_OUT_COUNT) {
        for (int off = 32; off > 0; off >>= 1) count_acc += __shfl_down(count_acc, off, 64);
        if (lane == 0 && count_acc)
            atomicAdd(out.count + ((blockIdx.x + 131u * blockIdx.y) % FR_COUNT_SLOTS), count_acc);
    }
}

template <typename T, int MODE, int kStripTiles, int FORM, bool CYC>
__global__ __launch_bounds__(64) void escape_refill_kernel(const fr_kparams p, const fr_kout out) {
    __shared__ double s_tab[(FR_LOG2_N * 3 * 8 > FR_MAX_PALETTE_ENTRIES * 4 ? FR_LOG2_N * 3 * 8 : FR_MAX_PALETTE_ENTRIES * 4) / 8];
    const uint32_t lane = threadIdx.x;
    const uint32_t *s_pal = nullptr;
    if (MODE == FR_OUT_RGB) {
        if (p.palette != nullptr) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(s_tab);
            for (uint32_t k = lane; k < p.palette_entries; k += 64) dst[k] = p.palette[k];
            s_pal = dst;
        } else if (p.smooth) {
            const double *gt = &g_log2_tab[0][0];
            for (uint32_t k = lane; k < FR_LOG2_N * 3; k += 64) s_tab[k] = gt[k];
        }
        __syncthreads();
    }
    static_assert(kStripTiles <= 7 && kRefillBands <= 8, "column lanes 0-55, row lanes 0-63");
    const uint32_t row0 = (blockIdx.y + gridDim.y * blockIdx.z) * (8u * kRefillBands);
    if (row0 >= p.nrows) return;
    const uint32_t tile0 = blockIdx.x * kStripTiles;
    const double width = (double)p.width, height = (double)p.height;
    /* the patch's coordinate map (calc/src/lib.rs:182-197), one column and one row per lane */
    const uint32_t x = p.x_first + (tile0 * 8u + lane) * p.x_stride;
    const double cols = coord_to_space((double)x, height, (width / height) / 2.0, p.pos_re, p.scale_re);
    const uint32_t rr = row0 + lane;
    const uint32_t y = p.y_first + (rr / p.block_rows) * p.y_stride + rr % p.block_rows;
    const double rows = coord_to_space((double)y, height, 0.5, p.pos_im, p.scale_im);
    /* where local row rr goes: packed, or its image row (in place) */
    const uint32_t out_rows = p.out_in_place ? y : rr;

    bool scaled_ok = false;
    if constexpr (FORM != 0)
        scaled_ok = coords_admissible<T>(p, cols, lane < 8u * kStripTiles && tile0 * 8u + lane < p.ncols) &&
                    coords_admissible<T>(p, rows, rr < p.nrows);
    if (FORM != 0 && scaled_ok)
        refill_patch<T, MODE, kStripTiles, FORM, CYC>(p, out, s_tab, s_pal, cols, rows, out_rows, tile0, row0, lane);
    else
        refill_patch<T, MODE, kStripTiles, 0, false>(p, out, s_tab, s_pal, cols, rows, out_rows, tile0, row0, lane);
}

template <typename T, int kStripTiles, int FORM, bool CYC>
hipError_t launch_refill_form(const fr_kparams &p, int mode, const fr_kout &out, dim3 grid, hipStream_t stream) {
    dim3 block(64);
    switch (mode) {
    case FR_OUT_RGB:
        hipLaunchKernelGGL((escape_refill_kernel<T, FR_OUT_RGB, kStripTiles, FORM, CYC>), grid, block, 0, stream, p, out);
        break;
    case FR_OUT_ESCAPE:
        hipLaunchKernelGGL((escape_refill_kernel<T, FR_OUT_ESCAPE, kStripTiles, FORM, CYC>), grid, block, 0, stream, p, out);
        break;
    default:
        hipLaunchKernelGGL((escape_refill_kernel<T, FR_OUT_COUNT, kStripTiles, FORM, CYC>), grid, block, 0, stream, p, out);
        break;
    }
    return hipGetLastError();
}

template <typename T, int kStripTiles>
hipError_t launch_refill(const fr_kparams &p, int mode, const fr_kout &out, hipStream_t stream) {
    if (p.ncols == 0 || p.nrows == 0) return hipSuccess;
    const uint64_t gx = ((uint64_t)p.ncols + 8 * kStripTiles - 1) / (8 * kStripTiles);
    const uint64_t row_patches = ((uint64_t)p.nrows + 8 * kRefillBands - 1) / (8 * kRefillBands);
    const uint64_t gy = row_patches < 32768 ? row_patches : 32768;
    const uint64_t gz = (row_patches + gy - 1) / gy;
    if (gx > 0x7FFFFFFFull || gz > 65535) return hipErrorInvalidConfiguration;
    dim3 grid((uint32_t)gx, (uint32_t)gy, (uint32_t)gz);
    if (p.loop_mode == 4 && p.cycle_shortcut) return launch_refill_form<T, kStripTiles, 4, true>(p, mode, out, grid, stream);
    if (p.loop_mode == 2 && p.cycle_shortcut) return launch_refill_form<T, kStripTiles, 2, true>(p, mode, out, grid, stream);
    if (p.loop_mode == 4) return launch_refill_form<T, kStripTiles, 4, false>(p, mode, out, grid, stream);
    if (p.loop_mode == 2) return launch_refill_form<T, kStripTiles, 2, false>(p, mode, out, grid, stream);
    return launch_refill_form<T, kStripTiles, 0, false>(p, mode, out, grid, stream);
}

/* ---- work-queue kernel ----------------------------------------------------------------------------
 *
 * For views whose orbits are mostly short with a heavy tail (Julia sets: C4's mean is 44 iterations at a
 * cap of 4096).  There the strip kernel idles two thirds of its lanes, and in the patch-refill kernel
 * above some lane of every wave is always on its way out (with ~1.5 escapes per wave-iteration, a few
 * lanes are permanently between T and limit^2), so the wave never leaves the loop's fully-checked path:
 * 8 VALU + an EXEC-bookkeeping branch per iteration instead of 6.5, plus a partial-EXEC colour pass after
 * every episode.  This kernel separates the two regimes:
 *
 *   main loop      blocks of M unchecked iterations of the scaled form (6 VALU each), then ONE test per
 *                  block: |z|^2 <= T?  A lane that fails it freezes (v_cmpx) — by the choice of T
 *                  (fr_api.hip: plan_loop) it cannot have escaped before the block's last iteration — and
 *                  is pushed, UNFINISHED, onto a per-wave LDS stack: position, c, iterations done, pixel.
 *                  A per-lane f32 counter (+M per block, frozen with the lane) replaces every escape-index
 *                  handler.  27 VALU per 4 iterations, whatever the lanes are doing.
 *   finishing pass when 64 results wait: a FULL wave runs the last few iterations of each with the exact
 *                  per-iteration check (the unscaled loop; a pixel past T is 4-5 iterations from limit^2),
 *                  then the colour map and the store.  Pixels that may not use the scaled form at all
 *                  (the im == 0 row, the re == 0 column) and lanes within M iterations of the cap take the
 *                  same road with their whole remaining orbit.
 *   persistent     waves draw 64 x 32-pixel patches from a device-wide counter (one ahead, so the atomic's
 *                  latency is never waited for): a lane's pixel outlives its patch and the only tail is
 *                  the one at the end of the image.  A patch's column / row coordinates
 *                  (calc/src/lib.rs:181-197: two IEEE divisions per column and per row, not per pixel) are
 *                  staged in LDS when it is opened; a refill is two LDS reads and two multiplies.
 * Results are independent of the schedule: a pixel's orbit never depends on its lane, its wave, the order
 * of patches or where the main loop hands it over. */
constexpr uint32_t kQPatchW = 64, kQPatchH = 32;
constexpr uint32_t kQStack = 128;

/* One run of the main loop: blocks of M unchecked scaled iterations, one |z|^2 <= T test per block.
 * In: the active lanes (EXEC) all have A + B <= 4T.  Out: `running` = the lanes that passed every test
 * (the others froze at the end of the block whose test they failed), `cnt` += M per block while running,
 * return value = blocks done.  Ends when no lane runs, after `n` blocks, or — once `minrun` blocks were done —
 * when at most `thr` lanes still run. */
#define FR_QB_ASM(SFX, BLOCK_ITS, STEP)                                            \
    "s_mov_b64 %[sorig], exec\n"                                                   \
    "s_mov_b32 %[si], 0\n"                                                         \
    "s_cbranch_execz .Lqdone_%=\n"                                                 \
    ".Lqloop_%=:\n" BLOCK_ITS                                                      \
    "v_add_" SFX " %[t], %[A], %[B]\n"                                             \
    "v_add_f32 %[cnt], %[cnt], " STEP "\n"                                         \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"                                          \
    "s_add_u32 %[si], %[si], 1\n"                                                  \
    "s_cbranch_execz .Lqdone_%=\n"                                                 \
    "s_bcnt1_i32_b64 %[scnt], exec\n"                                              \
    "s_cmp_gt_u32 %[scnt], %[thr]\n"                                               \
    "s_cbranch_scc0 .Lqmaybe_%=\n"                                                 \
    ".Lqcont_%=:\n"                                                                \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc1 .Lqloop_%=\n"                                                  \
    "s_branch .Lqdone_%=\n"                                                        \
    ".Lqmaybe_%=:\n"                                                               \
    "s_cmp_lt_u32 %[si], %[minrun]\n"                                              \
    "s_cbranch_scc1 .Lqcont_%=\n"                                                  \
    ".Lqdone_%=:\n"                                                                \
    "s_mov_b64 %[srun], exec\n"                                                    \
    "s_mov_b64 exec, %[sorig]\n"

template <typename T, int M>
__device__ __forceinline__ uint32_t queue_block_run(uint32_t nblocks, T &X, T &Y, T &A, T &B, T c2re, T c2im, T skip_t,
                                                    float &cnt, uint32_t thr, uint32_t minrun, unsigned long long &running) {
    T t;
    T q;
    unsigned long long sorig, srun;
    uint32_t si, scnt;
    const uint32_t n = __builtin_amdgcn_readfirstlane(nblocks);
    const uint32_t sthr = __builtin_amdgcn_readfirstlane(thr), smin = __builtin_amdgcn_readfirstlane(minrun);
    const T t4_v = (T)4 * skip_t;
#define FR_QB_OPERANDS                                                                                          \
    : [X] "+v"(X), [Y] "+v"(Y), [A] "+v"(A), [B] "+v"(B), [cnt] "+v"(cnt), [t] "=&v"(t), [q] "=&v"(q),          \
      [sorig] "=&s"(sorig), [srun] "=&s"(srun), [si] "=&s"(si), [scnt] "=&s"(scnt)                              \
    : [c2re] "v"(c2re), [c2im] "v"(c2im), [t4lim] "s"(t4lim), [n] "s"(n), [thr] "s"(sthr), [minrun] "s"(smin)   \
    : "vcc", "scc"
    if constexpr (sizeof(T) == 8) {
        const uint64_t tb = fr_bits_of(t4_v);
        const uint64_t t4lim = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)tb) |
                               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(tb >> 32)) << 32);
        if constexpr (M == 4)
            asm volatile(FR_QB_ASM("f64", FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64"), "4.0") FR_QB_OPERANDS);
        else
            asm volatile(FR_QB_ASM("f64", FR_SC_IT("f64") FR_SC_IT("f64"), "2.0") FR_QB_OPERANDS);
    } else {
        const uint32_t t4lim = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, t4_v));
        if constexpr (M == 4)
            asm volatile(FR_QB_ASM("f32", FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32"), "4.0") FR_QB_OPERANDS);
        else
            asm volatile(FR_QB_ASM("f32", FR_SC_IT("f32") FR_SC_IT("f32"), "2.0") FR_QB_OPERANDS);
    }
    running = srun;
    return si;
}

/* The finishing pass's loop: recursive()'s own iteration (unscaled, 8 VALU) with the escape test after every
 * one — v_cmpx freezes a lane at `next`, exactly what recursive() returns — and a per-lane f32 count of the
 * iterations run (frozen with the lane), so no escape-index handler and no scalar work beyond the loop
 * control.  In: EXEC = the lanes to finish; out: `running` = the lanes that did not escape within n. */
#define FR_FIN_ASM(SFX)                                \
    "s_mov_b64 %[sorig], exec\n"                       \
    "s_mov_b32 %[si], 0\n"                             \
    "s_cbranch_execz .Lfdone_%=\n"                     \
    ".Lfloop_%=:\n"                                    \
    "v_add_" SFX " %[t], %[r2], -%[i2]\n"              \
    "v_add_" SFX " %[x], %[re], %[re]\n"               \
    "v_add_" SFX " %[re], %[t], %[cre]\n"              \
    "v_mul_" SFX " %[x], %[x], %[im]\n"                \
    "v_add_" SFX " %[im], %[x], %[cim]\n"              \
    "v_mul_" SFX " %[r2], %[re], %[re]\n"              \
    "v_mul_" SFX " %[i2], %[im], %[im]\n"              \
    "v_add_" SFX " %[t], %[r2], %[i2]\n"               \
    "v_add_f32 %[fc], 1.0, %[fc]\n"                    \
    "v_cmpx_nlt_" SFX " %[lim2], %[t]\n"               \
    "s_cbranch_execz .Lfdone_%=\n"                     \
    "s_add_u32 %[si], %[si], 1\n"                      \
    "s_cmp_lt_u32 %[si], %[n]\n"                       \
    "s_cbranch_scc1 .Lfloop_%=\n"                      \
    ".Lfdone_%=:\n"                                    \
    "s_mov_b64 %[srun], exec\n"                        \
    "s_mov_b64 exec, %[sorig]\n"

template <typename T>
__device__ __forceinline__ unsigned long long finish_run(uint32_t iterations, T &re, T &im, T &r2, T &i2, T cre, T cim, T squared,
                                                         float &fc) {
    T t, x;
    unsigned long long sorig, srun;
    uint32_t si;
    const uint32_t n = __builtin_amdgcn_readfirstlane(iterations);
#define FR_FIN_OPERANDS                                                                                       \
    : [re] "+v"(re), [im] "+v"(im), [r2] "+v"(r2), [i2] "+v"(i2), [fc] "+v"(fc), [t] "=&v"(t), [x] "=&v"(x),  \
      [sorig] "=&s"(sorig), [srun] "=&s"(srun), [si] "=&s"(si)                                                \
    : [cre] "v"(cre), [cim] "v"(cim), [lim2] "s"(lim2), [n] "s"(n)                                            \
    : "vcc", "scc"
    if constexpr (sizeof(T) == 8) {
        const uint64_t sq_bits = fr_bits_of(squared);
        const uint64_t lim2 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sq_bits) |
                              ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sq_bits >> 32)) << 32);
        asm volatile(FR_FIN_ASM("f64") FR_FIN_OPERANDS);
    } else {
        const uint32_t lim2 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, squared));
        asm volatile(FR_FIN_ASM("f32") FR_FIN_OPERANDS);
    }
    return srun;
}

/* SRC = 0: the pixels come from 64x32 patches of the image, as described above.
 * SRC = 1: they come from the survivor lists the first pass (escape_first_kernel) left in device memory: a
 *          "patch" is a chunk of up to FR_SURV_CHUNK entries of one list, a pixel arrives with its position
 *          after first_cap iterations and starts counting there.  npatch_x / npatches_arg are unused: the
 *          number of chunks follows from the lists' counters, read once by every wave (one counter per lane). */
template <typename T, int M, int SRC>
__global__ __launch_bounds__(64) void escape_queue_kernel(const fr_kparams p, const fr_kout out, uint32_t npatch_x,
                                                          uint32_t npatches_arg) {
    static_assert(M == 4 || M == 2, "the main loop is the scaled form in blocks of M");
    static_assert(FR_SURV_QUEUES == 64, "one list counter per lane");
    typedef typename Pair<T>::type T2;
    __shared__ T s_x[kQPatchW], s_y[kQPatchH];
    __shared__ uint32_t s_orow[kQPatchH];
    /* SRC 1: the open chunk's entries (position, output position, c for Mandelbrot) */
    __shared__ T2 s_cz[FR_SURV_CHUNK], s_cc[FR_SURV_CHUNK];
    __shared__ uint2 s_cpos[FR_SURV_CHUNK];
    __shared__ uint32_t s_ccnt[FR_SURV_CHUNK];
    static_assert(FR_SURV_CHUNK == 64, "a chunk is loaded one entry per lane");
    /* the stack of unfinished results: position and c (unscaled), iterations done, output position */
    __shared__ T q_re[kQStack], q_im[kQStack], q_cre[kQStack], q_cim[kQStack];
    __shared__ uint32_t q_it[kQStack], q_px[kQStack], q_py[kQStack];
    extern __shared__ uint32_t s_dyn_palette[]; /* smooth == false: the palette, staged once */
    /* Everything the COLD phases need — opening a patch (once per 1024 pixels) and the colour pass (once
     * per 64) — is RE-READ from the kernel-argument segment there (scalar loads through a pointer the
     * optimiser cannot see through), instead of being held in SGPRs across the hot loop as kernel arguments
     * normally are: those ~50 values overflow the scalar file and every use became a v_readlane spill reload. */
    const uint32_t lane = threadIdx.x;
    const uint32_t *s_pal = nullptr;
    if (p.palette != nullptr) {
        for (uint32_t k = lane; k < p.palette_entries; k += 64) s_dyn_palette[k] = p.palette[k];
        s_pal = s_dyn_palette;
        __syncthreads();
    }
    const double *tab = &g_log2_tab[0][0]; /* the exact colour path is rare here: the table stays in L2 */
    /* hot-path constants (SGPRs) */
    const bool julia = p.algo == 2;
    const T squared = sizeof(T) == 8 ? (T)(p.limit * p.limit) : (T)((float)p.limit * (float)p.limit);
    const T skip_t = (T)p.skip_t, t4 = (T)4 * (T)p.skip_t;
    const uint32_t cap = p.iterations;
    const uint32_t queue_want = p.queue_want, queue_minblocks = (p.queue_minrun + M - 1) / M;
    const T jre = (T)p.julia_re, jim = (T)p.julia_im;
    uint32_t *const counter = p.work_counter;
    const bool tracing = out.trace != nullptr;
    const unsigned long long t_start = tracing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t tr_patches = 0, tr_episodes = 0, tr_colour = 0, tr_iters = 0;
    unsigned long long ph_open = 0, ph_refill = 0, ph_loop = 0, ph_retire = 0, ph_finish = 0, ph_mark = 0;
#define FR_PHASE_BEGIN() if (tracing) ph_mark = __builtin_amdgcn_s_memtime()
#define FR_PHASE_END(ACC) if (tracing) ACC += __builtin_amdgcn_s_memtime() - ph_mark

    /* per-lane state of a running orbit: X = 2re, Y = 2im, A = X^2, B = Y^2, 2c; iterations done (exact in f32:
     * the host sends caps >= 2^24 elsewhere); where the pixel goes */
    T X = 0, Y = 0, A = 0, B = 0, c2re = 0, c2im = 0;
    float cnt = 0.0f;
    uint32_t px = 0, py = 0;
    bool busy = false;
    /* wave-uniform state */
    uint32_t have_patch = 0, exhausted = 0, next = 0, vw = 0, vh = 0, pcol0 = 0;
    uint32_t patch_ok = 0; /* every pixel of the open patch may use the scaled form (all but a sliver of patches) */
    uint32_t qcount = 0, upper = 0; /* results waiting; upper bound of the running lanes' iteration counts */
    /* Work is claimed through FR_SURV_QUEUES counters, not one: a single address takes 88 million atomics a
     * second on this device and 64 addresses on 64 lines 4 100 million (tools/ubench/buffer_atomic_oob.hip) — a
     * quarter of a million claims on one counter would be the bound of the whole kernel.  SRC 0: shard q holds
     * the patches q, q + 64, ...; SRC 1: list q's chunks.  A claim is answered after ~3 us and the wave waits
     * for it (every way of keeping it in flight across the loop either makes the compiler wait at once — the
     * join after `if (lane == 0)` copies the result — or hides the access from it), so claims are made rare
     * instead: a wave takes a BATCH of units, sized by what is left in its shard (guided self-scheduling:
     * 1/(2 x waves per shard) of it, at most 4 — waves differ in speed by a factor of two, and a slow wave with
     * 16 chunks in hand was the tail — down to single units at the end).
     * When its shard is used up a wave looks at all 64 counters at once and moves to the next shard with work. */
    uint32_t list_len = 0, chunk_n = 0;
    if constexpr (SRC == 1) {
        const uint32_t raw = p.surv_counts[lane * FR_SURV_COUNT_STRIDE];
        list_len = raw < p.surv_sub_capacity ? raw : p.surv_sub_capacity;
    }
    const uint32_t units_lane = SRC == 1 ? (list_len + FR_SURV_CHUNK - 1u) / FR_SURV_CHUNK
                                         : (npatches_arg > lane ? (npatches_arg - lane + FR_SURV_QUEUES - 1u) / FR_SURV_QUEUES : 0u);
    uint32_t cur_q = blockIdx.x & (FR_SURV_QUEUES - 1u);
    uint32_t batch_next = 0, batch_end = 0; /* the units of shard cur_q this wave holds: [batch_next, batch_end) */
    const uint32_t claim_div = 2u * (gridDim.x / FR_SURV_QUEUES + 1u);
    /* the next unit: unit j of shard cur_q; false = no work is left anywhere */
    auto take_unit = [&](uint32_t &j) -> bool {
        for (;;) {
            if (batch_next < batch_end) {
                j = batch_next++;
                return true;
            }
            const uint32_t units_q = __builtin_amdgcn_readlane(units_lane, cur_q);
            /* what was left when this wave last looked (others have claimed since: an upper bound) */
            const uint32_t left = units_q > batch_end ? units_q - batch_end : 0u;
            uint32_t size = SRC == 1 ? left / claim_div : 1u; /* (a batch of 64x32 patches is too coarse at the end) */
            size = size < 1u ? 1u : (size > 4u ? 4u : size);
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(counter + cur_q * FR_SURV_COUNT_STRIDE, size);
            got = __builtin_amdgcn_readfirstlane(got);
            if (got < units_q) {
                batch_next = got;
                batch_end = got + size < units_q ? got + size : units_q;
                continue;
            }
            /* this shard is used up: look at every shard's counter */
            const uint32_t claimed = __hip_atomic_load(counter + lane * FR_SURV_COUNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long m = __ballot(claimed < units_lane);
            if (m == 0ull) return false; /* claims only grow: every shard is used up for good */
            const uint32_t r = (cur_q + 1u) & 63u;
            const unsigned long long m2 = r ? (m >> r) | (m << (64u - r)) : m;
            cur_q = (cur_q + 1u + (uint32_t)__builtin_ctzll(m2)) & 63u;
            batch_next = batch_end = __builtin_amdgcn_readlane(claimed, cur_q);
        }
    };
    /* SRC 1: the NEXT chunk, already on its way into registers (one entry per lane); nx_n == 0: there is none */
    uint32_t nx_n = 0;
    T2 pf_z, pf_c;
    uint2 pf_pos = make_uint2(0u, 0u);
    uint32_t pf_cnt = 0;      /* iterations the entry's pixel has done */
    const uint32_t first_cap = p.first_cap;
    uint32_t chunk_maxcnt = 0; /* the largest of them in the open chunk */
    pf_z.x = pf_z.y = pf_c.x = pf_c.y = (T)0;
    /* SRC 1: start loading the next chunk's entries into the prefetch registers; they are waited for one chunk
     * later, when their latency has long passed.  (Every lane loads — lanes past the chunk's end its entry 0 —
     * so that no branch joins behind the loads: a join would copy the registers and wait right here.) */
    auto prefetch_chunk = [&]() {
        uint32_t j;
        nx_n = 0;
        if (take_unit(j)) {
            FR_COLD_PARAMS(kp);
            const uint32_t len = __builtin_amdgcn_readlane(list_len, cur_q);
            const size_t base = (size_t)cur_q * kp->surv_sub_capacity + (size_t)j * FR_SURV_CHUNK;
            nx_n = len - j * FR_SURV_CHUNK < FR_SURV_CHUNK ? len - j * FR_SURV_CHUNK : FR_SURV_CHUNK;
            const size_t e = base + (lane < nx_n ? lane : 0u);
            pf_z = static_cast<const T2 *>(kp->surv_z)[e];
            pf_pos = reinterpret_cast<const uint2 *>(kp->surv_pos)[e];
            pf_cnt = kp->surv_cnt[e];
            if (!julia) pf_c = static_cast<const T2 *>(kp->surv_c)[e];
        }
    };
    if constexpr (SRC == 1) prefetch_chunk();

    /* the finishing pass over the top 64 (or, at the end, all remaining) stack entries */
    auto finish_and_colour = [&](uint32_t base, uint32_t count) {
        const unsigned long long f0 = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
        __syncthreads();
        const bool mine = lane < count;
        const uint32_t e = base + (mine ? lane : 0u);
        T re = q_re[e], im = q_im[e];
        const T cre = q_cre[e], cim = q_cim[e];
        uint32_t done = q_it[e];
        T r2 = re * re, i2 = im * im;
        /* did the last iteration it ran escape?  (the earlier ones of its block cannot have: see plan_loop;
         * a fresh pixel, done == 0, has not been tested yet: recursive() tests AFTER an iteration) */
        const bool escaped0 = mine && done > 0u && r2 + i2 > squared; /* NaN: false, as in the reference */
        bool live = mine && !escaped0 && done < cap;
        uint32_t iters = escaped0 ? done - 1u : cap;
        /* exact per-iteration checks for whoever is not finished.  A round runs up to 64 iterations — a pixel
         * past T is 4-5 iterations from limit^2 — unless a lane is that close to the cap: then the round's length
         * is the smallest remaining count (a wave reduction: rare) */
        float fc = (float)done;
        for (;;) {
            const unsigned long long live_mask = __ballot(live);
            if (live_mask == 0ull) break;
            uint32_t n = 64u;
            if (__ballot(live && cap - done < 64u) != 0ull) n = wave_min_u32(live ? cap - done : 0xFFFFFFFFu);
            unsigned long long running = 0ull;
            if (live) running = finish_run<T>(n, re, im, r2, i2, cre, cim, squared, fc);
            const int first = (int)__builtin_ctzll(live_mask);
            running = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(running >> 32), first) << 32) |
                      (uint32_t)__builtin_amdgcn_readlane((uint32_t)running, first);
            if (live) {
                done = (uint32_t)fc;
                if (((running >> lane) & 1ull) == 0ull) { /* escaped in the iteration that made the count `done` */
                    iters = done - 1u;
                    live = false;
                } else if (done >= cap) {
                    iters = cap;
                    live = false;
                }
            }
        }
        if (mine) {
            FR_COLD_PARAMS(kp);
            const ColourConsts cc = make_colour_consts(*kp);
            uint8_t rgb[3];
            colour_pixel<T>(cc, re, im, r2, i2, iters, tab, s_pal, rgb);
            store_pixel(kp->ncols, kp->out_rgba, out.rgb, q_py[e], q_px[e], rgb);
        }
        tr_colour++;
        __syncthreads();
        if (tracing) ph_finish += __builtin_amdgcn_s_memtime() - f0;
    };
    /* push the lanes with `who` set; re/im/c are the UNSCALED values */
    auto push = [&](bool who, T re, T im, T cre, T cim, uint32_t done) {
        const unsigned long long m = __ballot(who);
        if (m == 0ull) return;
        if (who) {
            const uint32_t slot = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            q_re[slot] = re, q_im[slot] = im, q_cre[slot] = cre, q_cim[slot] = cim;
            q_it[slot] = done, q_px[slot] = px, q_py[slot] = py;
        }
        qcount += (uint32_t)__builtin_popcountll(m);
        if (qcount >= 64u) {
            finish_and_colour(qcount - 64u, 64u);
            qcount -= 64u;
        }
    };

    for (;;) {
        /* ---- 1. open the next patch when the current one is used up */
        FR_PHASE_BEGIN();
        if constexpr (SRC == 1) {
            if (!have_patch && !exhausted) {
                if (nx_n == 0) {
                    exhausted = 1;
                } else {
                    __syncthreads(); /* earlier reads of the chunk arrays are done */
                    s_cz[lane] = pf_z;
                    s_cpos[lane] = pf_pos;
                    s_ccnt[lane] = pf_cnt;
                    if (!julia) s_cc[lane] = pf_c;
                    /* nearly every chunk's entries left the first pass after its first episode */
                    chunk_maxcnt = __ballot(lane < nx_n && pf_cnt != first_cap) == 0ull ? first_cap
                                                                                        : wave_max_u32(lane < nx_n ? pf_cnt : 0u);
                    __syncthreads();
                    chunk_n = nx_n;
                    have_patch = 1;
                    next = 0;
                    tr_patches++;
                    prefetch_chunk();
                }
            }
        } else if (!have_patch && !exhausted) {
            uint32_t unit;
            if (!take_unit(unit)) {
                exhausted = 1;
            } else {
                const uint32_t id = cur_q + unit * FR_SURV_QUEUES;
                FR_COLD_PARAMS(kp);
                const auto &P = *kp;
                /* the patch's coordinate map (calc/src/lib.rs:182-197), one column and one row per lane */
                const uint32_t pyi = id / npatch_x, pxi = id - pyi * npatch_x;
                const uint32_t col = pxi * kQPatchW + lane, rr = pyi * kQPatchH + lane;
                const uint32_t block_rows = P.block_rows;
                const uint32_t x = P.x_first + col * P.x_stride;
                const uint32_t y = P.y_first + (rr / block_rows) * P.y_stride + rr % block_rows;
                const double width = (double)P.width, height = (double)P.height;
                const double cx = coord_to_space((double)x, height, (width / height) / 2.0, P.pos_re, P.scale_re);
                const double cy = coord_to_space((double)y, height, 0.5, P.pos_im, P.scale_im);
                const uint32_t ncols = P.ncols, nrows = P.nrows, prow0 = pyi * kQPatchH;
                pcol0 = pxi * kQPatchW;
                vw = ncols - pcol0 < kQPatchW ? ncols - pcol0 : kQPatchW;
                vh = nrows - prow0 < kQPatchH ? nrows - prow0 : kQPatchH;
                const double pjre = P.julia_re, pjim = P.julia_im;
                patch_ok = coords_admissible<T>(julia, pjre, pjim, cx, col < ncols) &&
                                   coords_admissible<T>(julia, pjre, pjim, cy, lane < kQPatchH && rr < nrows)
                               ? 1u
                               : 0u;
                __syncthreads(); /* earlier reads of s_x / s_y are done */
                s_x[lane] = (T)cx;
                if (lane < kQPatchH) {
                    s_y[lane] = (T)cy;
                    s_orow[lane] = P.out_in_place ? y : rr;
                }
                __syncthreads();
                have_patch = 1;
                next = 0;
                tr_patches++;
            }
        }
        FR_PHASE_END(ph_open);
        /* ---- 2. hand unstarted pixels to the free lanes */
        FR_PHASE_BEGIN();
        unsigned long long busy_mask = __ballot(busy);
        if (have_patch && busy_mask != ~0ull) {
            const unsigned long long free_mask = ~busy_mask;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)free_mask, 0u));
            const uint32_t pix = next + rank;
            bool direct = false; /* this pixel cannot enter the main loop: straight to the finishing pass */
            T sx = 0, sy = 0, cre = 0, cim = 0;
            uint32_t start_count = 0; /* iterations the pixel has done when it arrives */
            if constexpr (SRC == 1) {
                if (!busy && pix < chunk_n) {
                    const T2 zz = s_cz[pix];
                    const uint2 pos = s_cpos[pix];
                    start_count = s_ccnt[pix];
                    sx = zz.x, sy = zz.y;
                    cre = jre, cim = jim;
                    if (!julia) {
                        const T2 cc2 = s_cc[pix];
                        cre = cc2.x, cim = cc2.y;
                    }
                    X = sx + sx, Y = sy + sy, A = X * X, B = Y * Y, c2re = cre + cre, c2im = cim + cim;
                    px = pos.x, py = pos.y;
                    cnt = (float)start_count;
                    /* (the scaled form is exact for this orbit: the first pass hands over only pixels of strips
                     * it ran in the scaled form itself — admissible start and c, see strip_is_scalable) */
                    if (A + B <= t4 && start_count + (uint32_t)M <= cap)
                        busy = true;
                    else
                        direct = true;
                }
                next += (uint32_t)__builtin_popcountll(free_mask);
                if (next >= chunk_n) have_patch = 0;
                if (upper < chunk_maxcnt) upper = chunk_maxcnt;
            } else {
                const uint32_t col = pix & (kQPatchW - 1), row = pix >> 6;
                if (!busy && col < vw && row < vh) { /* row < vh <= kQPatchH also bounds pix */
                    sx = s_x[col], sy = s_y[row];
                    cre = julia ? jre : sx, cim = julia ? jim : sy; /* calc/src/lib.rs:209-210 */
                    X = sx + sx, Y = sy + sy, A = X * X, B = Y * Y, c2re = cre + cre, c2im = cim + cim;
                    px = pcol0 + col;
                    py = s_orow[row];
                    cnt = 0.0f;
                    /* the scaled form must be provably exact for this pixel, and it must start under T */
                    if ((patch_ok || lane_is_scalable<T>(sx, sy, cre, cim)) && A + B <= t4 && cap >= (uint32_t)M)
                        busy = true;
                    else
                        direct = true;
                }
                next += (uint32_t)__builtin_popcountll(free_mask);
                if ((next >> 6) >= vh) have_patch = 0;
            }
            push(direct, sx, sy, cre, cim, start_count);
            busy_mask = __ballot(busy);
        }
        FR_PHASE_END(ph_refill);
        if constexpr (SRC == 1) { /* the chunk ran out before the free lanes did: open the next one right away */
            if (!have_patch && !exhausted && busy_mask != ~0ull) continue;
        }
        if (busy_mask == 0ull) {
            if (!have_patch && exhausted) break;
            continue;
        }
        /* ---- 3. one run of the main loop */
        FR_PHASE_BEGIN();
        if (upper + (uint32_t)M > cap) upper = wave_max_u32(busy ? (uint32_t)cnt : 0u); /* every busy lane: cnt + M <= cap */
        const uint32_t nblocks = (cap - upper) / (uint32_t)M;
        const uint32_t nbusy = (uint32_t)__builtin_popcountll(busy_mask);
        uint32_t thr = 0, minblocks = 0;
        if (have_patch || !exhausted) { /* more pixels wait: stop once `queue_want` lanes are free */
            thr = nbusy > queue_want ? nbusy - queue_want : 0u;
            minblocks = queue_minblocks;
        }
        unsigned long long running = 0ull;
        uint32_t blocks = 0;
        if (busy) blocks = queue_block_run<T, M>(nblocks, X, Y, A, B, c2re, c2im, skip_t, cnt, thr, minblocks, running);
        const int first = (int)__builtin_ctzll(busy_mask);
        blocks = __builtin_amdgcn_readlane(blocks, first); /* wave-uniform, but assigned under `if (busy)` */
        running = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(running >> 32), first) << 32) |
                  (uint32_t)__builtin_amdgcn_readlane((uint32_t)running, first);
        upper += blocks * (uint32_t)M;
        tr_episodes++;
        tr_iters += blocks * (uint32_t)M;
        FR_PHASE_END(ph_loop);
        /* ---- 4. retire: lanes that froze past T, and lanes that cannot fit another block under the cap */
        FR_PHASE_BEGIN();
        const bool still = busy && ((running >> lane) & 1ull) != 0ull;
        const bool capped = still && (uint32_t)cnt + (uint32_t)M > cap;
        const bool leave = busy && (!still || capped);
        if (leave) busy = false;
        push(leave, X * (T)0.5, Y * (T)0.5, c2re * (T)0.5, c2im * (T)0.5, (uint32_t)cnt); /* exact halvings */
        FR_PHASE_END(ph_retire);
    }
    /* ---- the last, partial batch */
    if (qcount) finish_and_colour(0u, qcount);
    if (out.trace && lane == 0) {
        unsigned long long *t = out.trace + (size_t)blockIdx.x * 16;
        t[0] = t_start, t[1] = __builtin_amdgcn_s_memrealtime(), t[2] = tr_patches, t[3] = tr_episodes, t[4] = tr_colour,
        t[5] = tr_iters, t[6] = (ph_open & 0xFFFFFFFFull) | (ph_refill << 32), t[7] = (ph_loop & 0xFFFFFFFFull) | (ph_retire << 32);
        t[8] = ph_finish;
    }
#undef FR_PHASE_BEGIN
#undef FR_PHASE_END
}

/* ---- second pass, second form (round 3) --------------------------------------------------------------
 *
 * escape_queue_kernel<.., 1> above spends, on C4, four vector instructions in ten OUTSIDE its main loop
 * (profiles/r03_c4_f64_rocprofv3.txt: 5.0e8 of them for 5.35e7 wave-iterations at 6.75): per 64-entry chunk 2.5 refills,
 * 2.5 retirements and one finishing pass, each written for the general case — results leave the loop converted back to
 * re / im (four multiplies and seven LDS words per lane), the finishing pass runs the unscaled 8-instruction iteration,
 * colours every lane through the general colour path (both stages of the filter as soon as one lane of 64 is at the cap,
 * which is six batches in seven) and stores three bytes per pixel through 64-bit addresses.  This kernel is the same
 * schedule — persistent one-wave workgroups, 64-entry chunks claimed in batches from the 64 lists, unchecked blocks with
 * freezing lanes, an LDS stack of results finished 64 at a time — with the first pass's (escape_first_kernel) means:
 *   - the state stays in the scaled form from the list to the colour: a chunk is converted ONCE, by 64 lanes, when it is
 *     opened (X = 2re, Y = 2im, 2c, the count as f32) and a refill is two or three LDS reads and two squares; a result is
 *     pushed as it stands (X, Y | count, column, row | 2c for Mandelbrot: two or three wide LDS writes, one address);
 *   - the finishing pass is finish_scaled (9 instructions per exact iteration, no conversion), then the filter's f32 stage
 *     in the first pass's form (colour_fast32) for the lanes that ESCAPED — per lane, not per wave: the lanes at the cap and
 *     the rare undecided ones alone take the general path — and one 16-bit + one 8-bit store at a 32-bit offset;
 *   - which lanes are busy, still running, or leaving are 64-bit masks; the main loop sets its own EXEC from the mask
 *     and returns scalars; the per-lane test against the cap runs only when the wave's largest count is within M of it.
 * Same bytes: a pixel's orbit is recursive()'s arithmetic whichever loop runs it (see "orbit loop, scaled form"). */
#define FR_SB_ASM(SFX, BLOCK_ITS, STEP)                                            \
    "s_mov_b64 %[sorig], exec\n"                                                   \
    "s_mov_b64 exec, %[mask]\n"                                                    \
    "s_mov_b32 %[si], 0\n"                                                         \
    ".Lsloop_%=:\n" BLOCK_ITS                                                      \
    "v_add_" SFX " %[t], %[A], %[B]\n"                                             \
    "v_add_f32 %[cnt], %[cnt], " STEP "\n"                                         \
    "v_cmpx_nlt_" SFX " %[t4lim], %[t]\n"                                          \
    "s_add_u32 %[si], %[si], 1\n"                                                  \
    "s_cbranch_execz .Lsdone_%=\n"                                                 \
    "s_bcnt1_i32_b64 %[scnt], exec\n"                                              \
    "s_cmp_gt_u32 %[scnt], %[thr]\n"                                               \
    "s_cbranch_scc0 .Lsmaybe_%=\n"                                                 \
    ".Lscont_%=:\n"                                                                \
    "s_cmp_lt_u32 %[si], %[n]\n"                                                   \
    "s_cbranch_scc1 .Lsloop_%=\n"                                                  \
    "s_branch .Lsdone_%=\n"                                                        \
    ".Lsmaybe_%=:\n"                                                               \
    "s_cmp_lt_u32 %[si], %[minrun]\n"                                              \
    "s_cbranch_scc1 .Lscont_%=\n"                                                  \
    ".Lsdone_%=:\n"                                                                \
    "s_mov_b64 %[srun], exec\n"                                                    \
    "s_mov_b64 exec, %[sorig]\n"

/* One run of the main loop for the lanes of `mask` (not empty; each with A + B <= 4T and count + M * nblocks <= cap,
 * nblocks >= 1): as queue_block_run, but the mask is an argument and everything that comes back is a scalar. */
template <typename T, int M>
__device__ __forceinline__ uint32_t second_block_run(unsigned long long mask, uint32_t nblocks, T &X, T &Y, T &A, T &B, T c2re,
                                                     T c2im, typename UBits<T>::type t4lim, float &cnt, uint32_t thr,
                                                     uint32_t minrun, unsigned long long &running) {
    T t, q;
    unsigned long long sorig, srun;
    uint32_t si, scnt;
#define FR_SB_OPERANDS                                                                                          \
    : [X] "+v"(X), [Y] "+v"(Y), [A] "+v"(A), [B] "+v"(B), [cnt] "+v"(cnt), [t] "=&v"(t), [q] "=&v"(q),          \
      [sorig] "=&s"(sorig), [srun] "=&s"(srun), [si] "=&s"(si), [scnt] "=&s"(scnt)                              \
    : [c2re] "v"(c2re), [c2im] "v"(c2im), [t4lim] "s"(t4lim), [mask] "s"(mask), [n] "s"(nblocks), [thr] "s"(thr), \
      [minrun] "s"(minrun)                                                                                      \
    : "vcc", "scc"
    if constexpr (sizeof(T) == 8) {
        if constexpr (M == 4)
            asm volatile(FR_SB_ASM("f64", FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64") FR_SC_IT("f64"), "4.0") FR_SB_OPERANDS);
        else
            asm volatile(FR_SB_ASM("f64", FR_SC_IT("f64") FR_SC_IT("f64"), "2.0") FR_SB_OPERANDS);
    } else {
        if constexpr (M == 4)
            asm volatile(FR_SB_ASM("f32", FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32") FR_SC_IT("f32"), "4.0") FR_SB_OPERANDS);
        else
            asm volatile(FR_SB_ASM("f32", FR_SC_IT("f32") FR_SC_IT("f32"), "2.0") FR_SB_OPERANDS);
    }
    running = srun;
    return si;
}

template <typename T, int M>
__global__ __launch_bounds__(64) void escape_second_kernel(const fr_kparams p, const fr_kout out) {
    static_assert(M == 4 || M == 2, "the main loop is the scaled form in blocks of M");
    static_assert(FR_SURV_QUEUES == 64 && FR_SURV_CHUNK == 64, "one list counter per lane; a chunk is loaded one entry per lane");
    typedef typename Pair<T>::type T2;
    typedef typename UBits<T>::type UB;
    /* the open chunk, converted: (X, Y), {count as f32 bits, output column, output row, count} */
    __shared__ T2 s_cxy[FR_SURV_CHUNK];
    __shared__ uint4 s_cmeta[FR_SURV_CHUNK];
    /* the stack of unfinished results, in the same form */
    __shared__ T2 q_xy[kQStack];
    __shared__ uint4 q_meta[kQStack];
    /* dynamic: 2c of the chunk's and the stack's entries — Mandelbrot only: a Julia render's c is one constant, and the
     * 3 KB (f64) they would take are three more resident waves per CU — then the palette (smooth == false), staged once */
    extern __shared__ uint4 s_dyn[];
    T2 *const s_cc2 = reinterpret_cast<T2 *>(s_dyn);
    T2 *const q_c2 = s_cc2 + FR_SURV_CHUNK;
    uint32_t *const s_dyn_palette = reinterpret_cast<uint32_t *>(s_dyn) + (p.algo == 2 ? 0u : (uint32_t)((FR_SURV_CHUNK + kQStack) * sizeof(T2) / 4));
    const uint32_t lane = threadIdx.x;
    const uint32_t *s_pal = nullptr;
    if (p.palette != nullptr) {
        for (uint32_t k = lane; k < p.palette_entries; k += 64) s_dyn_palette[k] = p.palette[k];
        s_pal = s_dyn_palette;
        __syncthreads();
    }
    const double *tab = &g_log2_tab[0][0]; /* the exact colour path is rare here: the table stays in L2 */
    const bool julia = p.algo == 2;
    const T squared = sizeof(T) == 8 ? (T)(p.limit * p.limit) : (T)((float)p.limit * (float)p.limit);
    const T t4v = (T)4 * (T)p.skip_t, lim4v = (T)4 * squared;
    const UB t4lim = uniform_bits<T>(t4v), lim4 = uniform_bits<T>(lim4v);
    const uint32_t cap = p.iterations; /* < 2^24 (the host sends larger caps elsewhere): counts are exact in f32 */
    const float capf = (float)cap;
    const uint32_t queue_want = p.queue_want, queue_minblocks = (p.queue_minrun + M - 1) / M;
    const T j2re = (T)p.julia_re + (T)p.julia_re, j2im = (T)p.julia_im + (T)p.julia_im;
    uint32_t *const counter = p.work_counter;
    const uint32_t first_cap = p.first_cap;
    bool fast_colour, narrow;
    uint32_t bpp, ncols;
    {
        FR_COLD_PARAMS(kp);
        fast_colour = kp->colour_filter32 && kp->smooth && kp->palette == nullptr;
        bpp = kp->out_rgba ? 4u : 3u;
        ncols = kp->ncols;
        const uint64_t rows = kp->out_in_place ? kp->height : kp->nrows; /* no output row lies beyond it */
        narrow = rows * (uint64_t)ncols <= 0xFFFFFFFFull / bpp;
    }
    const unsigned long long t_start = out.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t tr_chunks = 0, tr_episodes = 0, tr_colour = 0, tr_iters = 0;

    /* work claiming: as escape_queue_kernel<.., 1> (batches of chunks from 64 counters, guided self-scheduling) */
    uint32_t list_len;
    {
        const uint32_t raw = p.surv_counts[lane * FR_SURV_COUNT_STRIDE];
        list_len = raw < p.surv_sub_capacity ? raw : p.surv_sub_capacity;
    }
    const uint32_t units_lane = (list_len + FR_SURV_CHUNK - 1u) / FR_SURV_CHUNK;
    uint32_t cur_q = blockIdx.x & (FR_SURV_QUEUES - 1u);
    uint32_t batch_next = 0, batch_end = 0;
    const uint32_t claim_div = 2u * (gridDim.x / FR_SURV_QUEUES + 1u);
    auto take_unit = [&](uint32_t &j) __attribute__((always_inline)) -> bool {
        for (;;) {
            if (batch_next < batch_end) {
                j = batch_next++;
                return true;
            }
            const uint32_t units_q = __builtin_amdgcn_readlane(units_lane, cur_q);
            const uint32_t left = units_q > batch_end ? units_q - batch_end : 0u;
            uint32_t size = left / claim_div;
            size = size < 1u ? 1u : (size > 4u ? 4u : size);
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(counter + cur_q * FR_SURV_COUNT_STRIDE, size);
            got = __builtin_amdgcn_readfirstlane(got);
            if (got < units_q) {
                batch_next = got;
                batch_end = got + size < units_q ? got + size : units_q;
                continue;
            }
            const uint32_t claimed = __hip_atomic_load(counter + lane * FR_SURV_COUNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long m = ballot64(claimed < units_lane);
            if (m == 0ull) return false;
            const uint32_t r = (cur_q + 1u) & 63u;
            const unsigned long long m2 = r ? (m >> r) | (m << (64u - r)) : m;
            cur_q = (cur_q + 1u + (uint32_t)__builtin_ctzll(m2)) & 63u;
            batch_next = batch_end = __builtin_amdgcn_readlane(claimed, cur_q);
        }
    };
    /* the NEXT chunk, on its way into registers (one entry per lane) while the open one is worked on */
    uint32_t nx_n = 0;
    T2 pf_z, pf_c;
    uint2 pf_pos = make_uint2(0u, 0u);
    uint32_t pf_cnt = 0;
    pf_z.x = pf_z.y = pf_c.x = pf_c.y = (T)0;
    auto prefetch_chunk = [&]() __attribute__((always_inline)) {
        uint32_t j;
        nx_n = 0;
        if (take_unit(j)) {
            FR_COLD_PARAMS(kp);
            const uint32_t len = __builtin_amdgcn_readlane(list_len, cur_q);
            const size_t base = (size_t)cur_q * kp->surv_sub_capacity + (size_t)j * FR_SURV_CHUNK;
            nx_n = len - j * FR_SURV_CHUNK < FR_SURV_CHUNK ? len - j * FR_SURV_CHUNK : FR_SURV_CHUNK;
            const size_t e = base + (lane < nx_n ? lane : 0u);
            pf_z = static_cast<const T2 *>(kp->surv_z)[e];
            pf_pos = reinterpret_cast<const uint2 *>(kp->surv_pos)[e];
            pf_cnt = kp->surv_cnt[e];
            if (!julia) pf_c = static_cast<const T2 *>(kp->surv_c)[e];
        }
    };
    prefetch_chunk();

    /* per-lane state of a running orbit */
    T X = 0, Y = 0, A = 0, B = 0, c2re = j2re, c2im = j2im;
    float cnt = 0.0f;
    uint32_t px = 0, py = 0;
    bool busy = false;
    /* wave-uniform state */
    uint32_t have_chunk = 0, exhausted = 0, next = 0, chunk_n = 0, chunk_maxcnt = 0;
    uint32_t qcount = 0, upper = 0; /* results waiting; upper bound of the busy lanes' counts */

    /* the finishing pass over `count` stack entries from `base` */
    auto finish_and_colour = [&](uint32_t base, uint32_t count) __attribute__((always_inline)) {
        __syncthreads();
        const bool mine = lane < count;
        const uint32_t e = base + (mine ? lane : 0u);
        const T2 xy = q_xy[e];
        const uint4 meta = q_meta[e];
        T fX = xy.x, fY = xy.y, f2re = j2re, f2im = j2im;
        if (!julia) {
            const T2 c2 = q_c2[e];
            f2re = c2.x, f2im = c2.y;
        }
        T fA = fX * fX, fB = fY * fY, ft = fA + fB;
        float fc = __builtin_bit_cast(float, meta.x);
        /* did the last iteration it ran escape?  (the earlier ones of its block cannot have: plan_loop; an entry that
         * has run none has not been tested yet: recursive() tests AFTER an iteration; NaN: no, as in the reference) */
        const bool escaped0 = mine && fc > 0.0f && ft > lim4v;
        bool lv = mine && !escaped0 && fc < capf;
        unsigned long long esc = ballot64(escaped0), live = ballot64(lv);
        /* the exact loop for whoever is not done: a lane past T is 4-5 iterations from limit^2; rounds of up to 64,
         * shorter when a lane is that close to the cap (lanes AT the cap retire: they are points of the set) */
        while (live != 0ull) {
            uint32_t nf = 64u;
            if (ballot64(lv && capf - fc < 64.0f) != 0ull) {
                live &= ~ballot64(lv && fc >= capf);
                if (live == 0ull) break;
                nf = cap - __builtin_amdgcn_readfirstlane(wave_max_u32(lane_in(live) ? (uint32_t)fc : 0u));
                nf = nf > 64u ? 64u : nf;
            }
            const unsigned long long still = finish_scaled<T>(live, nf, fX, fY, fA, fB, ft, fc, f2re, f2im, lim4);
            esc |= live & ~still;
            live = still;
            lv = lane_in(live);
        }
        /* ---- colour (calc/src/lib.rs:214-234): the filter's f32 stage for the lanes that escaped — (A + B) / 4 IS
         * fl(re^2 + im^2) and the count is iters + 1, as in the first pass — the general path for the rest */
        const unsigned long long fin = count >= 64u ? ~0ull : (1ull << count) - 1ull;
        uint32_t packed = 0;
        unsigned long long coloured = 0ull;
        if (fast_colour && esc != 0ull) {
            Filter32 f;
            {
                FR_COLD_PARAMS(kp);
                f.lo = kp->filt_lo32, f.k = kp->filt_k32, f.c = kp->filt_c32;
                f.p0 = kp->prim32[0], f.p1 = kp->prim32[2], f.p2 = kp->prim32[1]; /* colour_multiply's RGB::new(r, b, g) swap */
            }
            const float d32 = (float)ft * 0.25f;
            const bool sure = d32 >= f.lo && d32 <= 0x1.ffffep119f;
            const bool decided = colour_fast32(f, d32, fc, packed);
            coloured = ballot64(sure && decided) & esc;
        }
        /* the lanes that did not escape (points of the set, at the cap) whose squared distance is within stable_limit —
         * all of them, bar contrived limits — are `inside` pixels: colour_of's last two branches (calc/src/lib.rs:230-233) */
        const unsigned long long stayed = fin & ~esc;
        bool inside_done = false;
        if (stayed != 0ull) {
            if (lane_in(stayed)) {
                double dist;
                if constexpr (sizeof(T) == 8) {
                    const T re = fX * (T)0.5, im = fY * (T)0.5;
                    dist = (double)(re * re + im * im);
                } else {
                    const double zre = (double)(fX * (T)0.5), zim = (double)(fY * (T)0.5);
                    dist = zre * zre + zim * zim;
                }
                FR_COLD_PARAMS(kp);
                inside_done = !(dist > kp->stable_limit);
                if (inside_done) {
                    packed = 0u;
                    if (kp->inside) /* color_multiply(secondary_color, dist), with its RGB::new(r, b, g) swap */
                        packed = sat_u8_dev(kp->sec_f[0] * dist) | (sat_u8_dev(kp->sec_f[2] * dist) << 8) | (sat_u8_dev(kp->sec_f[1] * dist) << 16);
                }
            }
        }
        const unsigned long long rest = fin & ~coloured & ~ballot64(inside_done);
        if (rest != 0ull) {
            if (lane_in(rest)) {
                const T re = fX * (T)0.5, im = fY * (T)0.5; /* exact */
                const T r2 = re * re, i2 = im * im;         /* the reference's own re*re, im*im */
                const uint32_t iters = lane_in(esc) ? (uint32_t)fc - 1u : cap;
                FR_COLD_PARAMS(kp);
                const ColourConsts cc = make_colour_consts(*kp);
                uint8_t rgb[3];
                colour_pixel<T>(cc, re, im, r2, i2, iters, tab, s_pal, rgb);
                packed = (uint32_t)rgb[0] | ((uint32_t)rgb[1] << 8) | ((uint32_t)rgb[2] << 16);
            }
        }
        /* ---- store (src/lib.rs:253-270) */
        if (mine) {
            if (narrow) {
                store_packed(out.rgb, (meta.z * ncols + meta.y) * bpp, packed, bpp);
            } else {
                uint8_t *o = out.rgb + ((uint64_t)meta.z * ncols + meta.y) * bpp;
                if (bpp == 4u) {
                    *reinterpret_cast<uint32_t *>(o) = packed | 0xFF000000u;
                } else {
                    o[0] = (uint8_t)packed, o[1] = (uint8_t)(packed >> 8), o[2] = (uint8_t)(packed >> 16);
                }
            }
        }
        tr_colour++;
        __syncthreads();
    };
    /* push the lanes with `who` set, as they stand */
    auto push = [&](bool who) __attribute__((always_inline)) {
        const unsigned long long m = ballot64(who);
        if (m == 0ull) return;
        if (who) {
            const uint32_t slot = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            T2 xy;
            xy.x = X, xy.y = Y;
            q_xy[slot] = xy;
            q_meta[slot] = make_uint4(__builtin_bit_cast(uint32_t, cnt), px, py, 0u);
            if (!julia) {
                T2 c2;
                c2.x = c2re, c2.y = c2im;
                q_c2[slot] = c2;
            }
        }
        qcount += (uint32_t)__builtin_popcountll(m);
        if (qcount >= 64u) {
            finish_and_colour(qcount - 64u, 64u);
            qcount -= 64u;
        }
    };

    for (;;) {
        /* ---- 1. open the next chunk when the current one is used up: convert it once, 64 entries at a time */
        if (!have_chunk && !exhausted) {
            if (nx_n == 0) {
                exhausted = 1;
            } else {
                __syncthreads(); /* earlier reads of the chunk arrays are done */
                T2 xy;
                xy.x = pf_z.x + pf_z.x, xy.y = pf_z.y + pf_z.y;
                s_cxy[lane] = xy;
                s_cmeta[lane] = make_uint4(__builtin_bit_cast(uint32_t, (float)pf_cnt), pf_pos.x, pf_pos.y, pf_cnt);
                if (!julia) {
                    T2 c2;
                    c2.x = pf_c.x + pf_c.x, c2.y = pf_c.y + pf_c.y;
                    s_cc2[lane] = c2;
                }
                /* nearly every chunk's entries left the first pass after its first episode */
                chunk_maxcnt = ballot64(lane < nx_n && pf_cnt != first_cap) == 0ull
                                   ? first_cap
                                   : __builtin_amdgcn_readfirstlane(wave_max_u32(lane < nx_n ? pf_cnt : 0u));
                __syncthreads();
                chunk_n = nx_n;
                have_chunk = 1;
                next = 0;
                tr_chunks++;
                prefetch_chunk();
            }
        }
        /* ---- 2. hand unstarted entries to the free lanes */
        unsigned long long busy_mask = ballot64(busy);
        if (have_chunk && busy_mask != ~0ull) {
            const unsigned long long free_mask = ~busy_mask;
            const uint32_t pix = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(free_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)free_mask, 0u));
            bool direct = false; /* this entry cannot enter the main loop: straight to the finishing pass */
            if (!busy && pix < chunk_n) {
                const T2 xy = s_cxy[pix];
                const uint4 meta = s_cmeta[pix];
                X = xy.x, Y = xy.y, A = X * X, B = Y * Y;
                if (!julia) {
                    const T2 c2 = s_cc2[pix];
                    c2re = c2.x, c2im = c2.y;
                }
                cnt = __builtin_bit_cast(float, meta.x);
                px = meta.y, py = meta.z;
                /* (the scaled form is exact for this orbit: the first pass hands over only pixels of strips it ran in
                 * the scaled form itself — admissible start and c, see strip_is_scalable) */
                if (A + B <= t4v && meta.w + (uint32_t)M <= cap)
                    busy = true;
                else
                    direct = true;
            }
            next += (uint32_t)__builtin_popcountll(free_mask);
            if (next >= chunk_n) have_chunk = 0;
            if (upper < chunk_maxcnt) upper = chunk_maxcnt;
            push(direct);
            busy_mask = ballot64(busy);
        }
        /* the chunk ran out before the free lanes did: open the next one right away */
        if (!have_chunk && !exhausted && busy_mask != ~0ull) continue;
        if (busy_mask == 0ull) {
            if (!have_chunk && exhausted) break;
            continue;
        }
        /* ---- 3. one run of the main loop (every busy lane: count + M <= cap) */
        if (upper + (uint32_t)M > cap) upper = __builtin_amdgcn_readfirstlane(wave_max_u32(busy ? (uint32_t)cnt : 0u));
        const uint32_t nblocks = (cap - upper) / (uint32_t)M;
        const uint32_t nbusy = (uint32_t)__builtin_popcountll(busy_mask);
        uint32_t thr = 0, minblocks = 0;
        if (have_chunk || !exhausted) { /* more entries wait: stop once `queue_want` lanes are free */
            /* (a saturating subtraction has no scalar form: the compiler computes it on the vector unit) */
            thr = __builtin_amdgcn_readfirstlane(nbusy > queue_want ? nbusy - queue_want : 0u);
            minblocks = queue_minblocks;
        }
        unsigned long long running;
        const uint32_t blocks = second_block_run<T, M>(busy_mask, nblocks, X, Y, A, B, c2re, c2im, t4lim, cnt, thr, minblocks, running);
        upper += blocks * (uint32_t)M;
        tr_episodes++;
        tr_iters += blocks * (uint32_t)M;
        /* ---- 4. retire: lanes that froze past T, and — only when the largest count is that close — lanes that cannot
         * fit another block under the cap */
        unsigned long long leave = busy_mask & ~running;
        if (upper + (uint32_t)M > cap) leave |= ballot64(lane_in(running) && (uint32_t)cnt + (uint32_t)M > cap);
        if (leave != 0ull) {
            const bool lv = lane_in(leave);
            if (lv) busy = false;
            push(lv);
        }
    }
    /* ---- the last, partial batch */
    if (qcount) finish_and_colour(0u, qcount);
    if (out.trace && lane == 0) { /* fr_debug_set_queue_trace: the same record as escape_queue_kernel's, without the phase cycles */
        unsigned long long *t = out.trace + (size_t)blockIdx.x * 16;
        t[0] = t_start, t[1] = __builtin_amdgcn_s_memrealtime(), t[2] = tr_chunks, t[3] = tr_episodes, t[4] = tr_colour, t[5] = tr_iters;
        t[6] = t[7] = t[8] = 0ull;
    }
}

template <typename T, int M, int SRC>
hipError_t launch_queue_form(const fr_kparams &p, const fr_kout &out, hipStream_t stream) {
    /* SRC 1: the number of chunks is known to the device only; size the persistent grid by an upper bound */
    const uint64_t npx = SRC == 1 ? ((uint64_t)p.surv_sub_capacity + FR_SURV_CHUNK - 1) / FR_SURV_CHUNK
                                  : ((uint64_t)p.ncols + kQPatchW - 1) / kQPatchW;
    const uint64_t npy = SRC == 1 ? FR_SURV_QUEUES : ((uint64_t)p.nrows + kQPatchH - 1) / kQPatchH;
    if (npx * npy == 0) return hipSuccess;
    if (npx * npy > 0xFFF00000ull) return hipErrorInvalidConfiguration; /* the counter overshoots by one per wave */
    size_t dyn = p.palette ? sizeof(uint32_t) * p.palette_entries : 0;
    if (SRC == 1 && !p.second_v1 && p.algo != 2) dyn += (FR_SURV_CHUNK + kQStack) * (sizeof(T) * 2); /* escape_second_kernel: 2c per entry */
    /* persistent grid: as many one-wave workgroups as the device holds at once (the occupancy query ignores
     * the SGPR budget and can answer a few too many per CU; harmless: a late workgroup finds the counter
     * exhausted and leaves) */
    static thread_local int cached_device = -1, cached_cus = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev != cached_device) {
        if ((e = hipDeviceGetAttribute(&cached_cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        cached_device = dev;
    }
    int per_cu = 0;
    if (SRC == 1 && !p.second_v1)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, escape_second_kernel<T, M>, 64, dyn);
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, escape_queue_kernel<T, M, SRC>, 64, dyn);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 32) per_cu = 32;
    uint64_t grid = (uint64_t)per_cu * (uint64_t)cached_cus;
    if (grid > npx * npy) grid = npx * npy;
    if (SRC == 1 && !p.second_v1) /* the survivor lists' own kernel; second_v1: round 2's (comparison only) */
        hipLaunchKernelGGL((escape_second_kernel<T, M>), dim3((uint32_t)grid), dim3(64), dyn, stream, p, out);
    else
        hipLaunchKernelGGL((escape_queue_kernel<T, M, SRC>), dim3((uint32_t)grid), dim3(64), dyn, stream, p, out, (uint32_t)npx,
                           (uint32_t)(npx * npy));
    return hipGetLastError();
}

/* RGB output only; needs p.work_counter (zeroed on the launch stream by the caller) */
template <typename T>
hipError_t launch_queue(const fr_kparams &p, const fr_kout &out, hipStream_t stream) {
    if (p.loop_mode == 4) return launch_queue_form<T, 4, 0>(p, out, stream);
    return launch_queue_form<T, 2, 0>(p, out, stream);
}

/* the first-pass kernel's form: later episodes in speculative blocks where the plan allows them, unless this is the two-pass
 * render (tiles hand their stragglers over) of a view whose statistics say that nothing stays */
static bool first_pass_speculates(const fr_kparams &p) {
    return p.loop_mode == 4 && p.loop_spec != 0 && !(p.first_no_spec && !p.first_only);
}

/* Two passes, RGB output only; needs the survivor lists and p.work_counter (all counters zeroed on the launch
 * stream by the caller) and 0 < p.first_cap < p.iterations */
template <typename T, int kStripTiles>
hipError_t launch_first_pass(const fr_kparams &p, const fr_kout &out, hipStream_t stream, bool v1) {
    const uint64_t gx = ((uint64_t)p.ncols + 8 * kStripTiles - 1) / (8 * kStripTiles);
    const uint64_t row_tiles = ((uint64_t)p.nrows + 7) / 8;
    /* Four strips per workgroup share its fixed costs (-1.5 % on the 16384^2 C4 in f64) but leave a quarter of the
     * workgroups to balance over the chip: at 8192^2 (37 000 of them for 8192 resident waves) C4's dust takes 0.86 ms
     * instead of 0.72, and filled sets, whose workgroups differ a thousandfold in cost, lose more.  So: only while
     * at least 131 072 workgroups remain (7-tile strips only: the short strips are for launches far below that). */
    const int bands = (kStripTiles == 7 && !p.first_one_band && gx * ((row_tiles + 3) / 4) >= 131072) ? 4 : 1;
    const uint64_t row_blocks = (row_tiles + bands - 1) / bands;
    const uint64_t gy = row_blocks < 32768 ? row_blocks : 32768;
    const uint64_t gz = (row_blocks + gy - 1) / gy;
    if (gx > 0x7FFFFFFFull || gz > 65535) return hipErrorInvalidConfiguration;
    const dim3 grid((uint32_t)gx, (uint32_t)gy, (uint32_t)gz);
    if constexpr (kStripTiles == 7) {
        if (v1 && p.loop_mode == 4 && bands == 4) /* round 2's first pass, kept for comparison (tile 12) */
            hipLaunchKernelGGL((escape_first_v1_kernel<T, 4, kStripTiles, 4>), grid, dim3(64), 0, stream, p, out);
        else if (v1 && p.loop_mode == 4)
            hipLaunchKernelGGL((escape_first_v1_kernel<T, 4, kStripTiles, 1>), grid, dim3(64), 0, stream, p, out);
        else if (bands == 4 && first_pass_speculates(p)) /* <.., true>: the later episodes may speculate (first_blocks) */
            hipLaunchKernelGGL((escape_first_kernel<T, 4, kStripTiles, 4, true>), grid, dim3(64), 0, stream, p, out);
        else if (first_pass_speculates(p))
            hipLaunchKernelGGL((escape_first_kernel<T, 4, kStripTiles, 1, true>), grid, dim3(64), 0, stream, p, out);
        else if (p.loop_mode == 4 && bands == 4)
            hipLaunchKernelGGL((escape_first_kernel<T, 4, kStripTiles, 4>), grid, dim3(64), 0, stream, p, out);
        else if (p.loop_mode == 4)
            hipLaunchKernelGGL((escape_first_kernel<T, 4, kStripTiles, 1>), grid, dim3(64), 0, stream, p, out);
        else if (bands == 4)
            hipLaunchKernelGGL((escape_first_kernel<T, 2, kStripTiles, 4>), grid, dim3(64), 0, stream, p, out);
        else
            hipLaunchKernelGGL((escape_first_kernel<T, 2, kStripTiles, 1>), grid, dim3(64), 0, stream, p, out);
    } else {
        if (first_pass_speculates(p))
            hipLaunchKernelGGL((escape_first_kernel<T, 4, kStripTiles, 1, true>), grid, dim3(64), 0, stream, p, out);
        else if (p.loop_mode == 4)
            hipLaunchKernelGGL((escape_first_kernel<T, 4, kStripTiles, 1>), grid, dim3(64), 0, stream, p, out);
        else
            hipLaunchKernelGGL((escape_first_kernel<T, 2, kStripTiles, 1>), grid, dim3(64), 0, stream, p, out);
    }
    return hipGetLastError();
}

/* Two passes, RGB output only; needs the survivor lists and p.work_counter (all counters zeroed on the launch
 * stream by the caller) and 0 < p.first_cap < p.iterations.  p.strip_tiles = 4: the first pass in 4-tile strips
 * (GUI-sized launches: four times as many workgroups to balance over the chip), else 7. */
template <typename T>
hipError_t launch_two_pass(const fr_kparams &p, const fr_kout &out, hipStream_t stream, bool v1 = false) {
    if (p.ncols == 0 || p.nrows == 0) return hipSuccess;
    const hipError_t e = (p.strip_tiles == 4 && !v1) ? launch_first_pass<T, 4>(p, out, stream, false)
                                                           : launch_first_pass<T, 7>(p, out, stream, v1);
    if (e != hipSuccess) return e;
    if (p.first_only) return hipSuccess; /* nothing was handed over: there are no lists */
    if (p.debug_ablate & 2u) return hipSuccess; /* measurement aid: the first pass's cost on its own */
    if (p.loop_mode == 4) return launch_queue_form<T, 4, 1>(p, out, stream);
    return launch_queue_form<T, 2, 1>(p, out, stream);
}

template <typename T, int kStripTiles>
hipError_t launch_strips(const fr_kparams &p, int mode, const fr_kout &out, hipStream_t stream) {
    if (p.ncols == 0 || p.nrows == 0) return hipSuccess;
    const uint64_t gx = ((uint64_t)p.ncols + 8 * kStripTiles - 1) / (8 * kStripTiles);
    const uint64_t row_tiles = ((uint64_t)p.nrows + 7) / 8;
    const uint64_t gy = row_tiles < 32768 ? row_tiles : 32768;
    const uint64_t gz = (row_tiles + gy - 1) / gy;
    if (gx > 0x7FFFFFFFull || gz > 65535) return hipErrorInvalidConfiguration;
    dim3 grid((uint32_t)gx, (uint32_t)gy, (uint32_t)gz), block(64);
    switch (mode) {
    case FR_OUT_RGB:
        hipLaunchKernelGGL((escape_strip_kernel<T, FR_OUT_RGB, kStripTiles>), grid, block, 0, stream, p, out);
        break;
    case FR_OUT_ESCAPE:
        hipLaunchKernelGGL((escape_strip_kernel<T, FR_OUT_ESCAPE, kStripTiles>), grid, block, 0, stream, p, out);
        break;
    default:
        hipLaunchKernelGGL((escape_strip_kernel<T, FR_OUT_COUNT, kStripTiles>), grid, block, 0, stream, p, out);
        break;
    }
    return hipGetLastError();
}

template <typename T, int TW, int TH, int WX, int WY>
hipError_t launch_tile(const fr_kparams &p, int mode, const fr_kout &out, hipStream_t stream) {
    constexpr int BW = TW * WX, BH = TH * WY;
    const uint64_t tiles_x = ((uint64_t)p.ncols + BW - 1) / BW;
    const uint64_t tiles_y = ((uint64_t)p.nrows + BH - 1) / BH;
    const uint64_t blocks = tiles_x * tiles_y;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    dim3 grid((uint32_t)blocks), block(64 * kWaves);
    switch (mode) {
    case FR_OUT_RGB:
        hipLaunchKernelGGL((escape_kernel<T, TW, TH, WX, WY, FR_OUT_RGB>), grid, block, 0, stream, p, out);
        break;
    case FR_OUT_ESCAPE:
        hipLaunchKernelGGL((escape_kernel<T, TW, TH, WX, WY, FR_OUT_ESCAPE>), grid, block, 0, stream, p, out);
        break;
    default:
        hipLaunchKernelGGL((escape_kernel<T, TW, TH, WX, WY, FR_OUT_COUNT>), grid, block, 0, stream, p, out);
        break;
    }
    return hipGetLastError();
}

#define FR_KNAME(base, k) (sizeof(T) == 8 ? base "<double, " k ">" : base "<float, " k ">")

template <typename T>
const char *first_pass_name(const fr_kparams &p) {
    /* which form of the first pass runs (launch_first_pass): the one whose later episodes may speculate, or the plain one */
    const bool spec = first_pass_speculates(p);
#define FR_FIRST_NAMES(SUFFIX)                                                                                                       \
    (p.strip_tiles == 4                                                                                                              \
         ? (p.first_only ? FR_KNAME("escape_first_kernel", "4-tile strips in episodes, every tile finished in place" SUFFIX)         \
                         : FR_KNAME("escape_first_kernel + escape_second_kernel",                                                    \
                                    "4-tile strips, then persistent waves over the survivor lists" SUFFIX))                          \
         : (p.first_only ? FR_KNAME("escape_first_kernel", "7-tile strips in episodes, every tile finished in place" SUFFIX)         \
                         : FR_KNAME("escape_first_kernel + escape_second_kernel",                                                    \
                                    "7-tile strips, then persistent waves over the survivor lists" SUFFIX)))
    return spec ? FR_FIRST_NAMES("; speculative blocks in the later episodes") : FR_FIRST_NAMES("");
#undef FR_FIRST_NAMES
}

template <typename T>
hipError_t launch_precision(const fr_kparams &p, int mode, const fr_kout &out, int tile, hipStream_t stream,
                            const char *&name) {
    if (p.out_in_place && tile > 16) tile = 0; /* only the strip kernels know in-place addressing */
    switch (tile) {
    case 6401:
        name = FR_KNAME("escape_kernel", "64x1");
        return launch_tile<T, 64, 1, 1, 4>(p, mode, out, stream);
    case 3202:
        name = FR_KNAME("escape_kernel", "32x2");
        return launch_tile<T, 32, 2, 2, 2>(p, mode, out, stream);
    case 1604:
        name = FR_KNAME("escape_kernel", "16x4");
        return launch_tile<T, 16, 4, 2, 2>(p, mode, out, stream);
    case 808:
        name = FR_KNAME("escape_kernel", "8x8");
        return launch_tile<T, 8, 8, 2, 2>(p, mode, out, stream);
    case 0: {
        /* strip length by image size: long strips amortise the per-workgroup setup, short ones
         * keep every SIMD supplied with several waves when the image is small (GUI frames) */
        const uint64_t tiles = (((uint64_t)p.ncols + 7) / 8) * (((uint64_t)p.nrows + 7) / 8);
        if (mode == FR_OUT_RGB && p.first_cap != 0 && (p.first_only || (p.surv_counts && p.work_counter))) {
            /* Julia views are mostly short orbits with a heavy tail: two passes (see escape_first_kernel; the
             * host asks for it from 65 536 tiles up: fr_wants_two_pass) */
            name = first_pass_name<T>(p);
            return launch_two_pass<T>(p, out, stream);
        }
        if (tiles >= 262144) {
            /* patch refill: behind the periodicity shortcut, and for the COUNT / ESCAPE outputs of Julia images
             * (RGB renders of Julia images this large take the two-pass kernels above) */
            if ((p.algo == 2 && mode != FR_OUT_RGB) || p.cycle_shortcut) {
                name = FR_KNAME("escape_refill_kernel", "7x2-tile patches");
                return launch_refill<T, 7>(p, mode, out, stream);
            }
        }
        /* p.strip_tiles: the length the view's own statistics call for (fr_api.hip: choose_kernel); else by launch size */
        /* By size (round 4, tools/strip_length_study.py -> profiles/r04_strip_length_by_size.txt): ONE tile per workgroup up to
         * 4096^2 — below that a launch has too few workgroups for longer strips to fill and balance the chip's 8192 wave
         * slots (1920 x 1080, default view, f64: 0.122 ms against 0.158 / 0.244 / 0.294 for 2 / 4 / 7 tiles; still 13-23 %
         * at 4096^2) — and the longest strips from 8192 x 4096 up, where the per-workgroup costs they amortise are what is
         * left (8192^2: 7 tiles best on six views of eight).  Views of very short orbits prefer long strips at every size;
         * the view's statistics say so from the second frame on (strip_tiles). */
        const uint32_t k_len = (mode == FR_OUT_RGB && p.strip_tiles) ? p.strip_tiles : tiles >= 524288 ? 7u : 1u;
        if (k_len >= 7u) {
            name = FR_KNAME("escape_strip_kernel", "7 tiles");
            return launch_strips<T, 7>(p, mode, out, stream);
        }
        if (k_len >= 4u) {
            name = FR_KNAME("escape_strip_kernel", "4 tiles");
            return launch_strips<T, 4>(p, mode, out, stream);
        }
        if (k_len >= 2u) {
            name = FR_KNAME("escape_strip_kernel", "2 tiles");
            return launch_strips<T, 2>(p, mode, out, stream);
        }
        name = FR_KNAME("escape_strip_kernel", "1 tile");
        return launch_strips<T, 1>(p, mode, out, stream);
    }
    case 8:
        name = FR_KNAME("escape_strip_kernel", "7 tiles");
        return launch_strips<T, 7>(p, mode, out, stream);
    case 13: /* the first pass alone: no tile is handed over, no lists, no second kernel */
    case 16: /* ... in 4-tile strips */
    case 12: /* two passes with round 2's first pass (comparison only) */
        if (mode == FR_OUT_RGB && p.first_cap != 0 && (p.first_only || (p.surv_counts && p.work_counter))) {
            if (tile == 12) {
                name = FR_KNAME("escape_first_v1_kernel + escape_queue_kernel", "round 2's first pass, then persistent waves over the survivor lists");
                return launch_two_pass<T>(p, out, stream, true);
            }
            name = first_pass_name<T>(p);
            return launch_two_pass<T>(p, out, stream);
        }
        [[fallthrough]];
    case 14: /* two passes with round 2's second-pass kernel (comparison only) */
    case 15: /* two passes, the first in 4-tile strips */
    case 11: /* two passes: strips to first_cap, then persistent waves over the survivors (otherwise as 9) */
        if (mode == FR_OUT_RGB && p.first_cap != 0 && (p.first_only || (p.surv_counts && p.work_counter))) {
            name = tile == 14 ? FR_KNAME("escape_first_kernel + escape_queue_kernel", "7-tile strips, then round 2's persistent waves over the survivor lists")
                              : first_pass_name<T>(p);
            return launch_two_pass<T>(p, out, stream);
        }
        if (p.algo != 0 && p.algo != 2) {
            name = FR_KNAME("escape_strip_kernel", "7 tiles");
            return launch_strips<T, 7>(p, mode, out, stream);
        }
        name = FR_KNAME("escape_refill_kernel", "7x2-tile patches");
        return launch_refill<T, 7>(p, mode, out, stream);
    case 10: /* the work-queue kernel (RGB output of an escape-time algorithm; otherwise as 9) */
        if (mode == FR_OUT_RGB && p.work_counter && fr_wants_work_queue(p, 10)) {
            name = FR_KNAME("escape_queue_kernel", "persistent waves, 64x32-px patches");
            return launch_queue<T>(p, out, stream);
        }
        [[fallthrough]];
    case 9: /* refilling strips; only the escape-time algorithms have orbits to refill */
        if (p.algo != 0 && p.algo != 2) {
            name = FR_KNAME("escape_strip_kernel", "7 tiles");
            return launch_strips<T, 7>(p, mode, out, stream);
        }
        name = FR_KNAME("escape_refill_kernel", "7x2-tile patches");
        return launch_refill<T, 7>(p, mode, out, stream);
    case 1:
        name = FR_KNAME("escape_strip_kernel", "1 tile");
        return launch_strips<T, 1>(p, mode, out, stream);
    case 2:
        name = FR_KNAME("escape_strip_kernel", "2 tiles");
        return launch_strips<T, 2>(p, mode, out, stream);
    case 4:
        name = FR_KNAME("escape_strip_kernel", "4 tiles");
        return launch_strips<T, 4>(p, mode, out, stream);
    default:
        return hipErrorInvalidValue;
    }
}

/* ---- recursive() for arbitrary (start, c) pairs — calc/src/lib.rs:245-257 ------------------ */

template <typename T>
__global__ __launch_bounds__(256) void recursive_batch_kernel(uint32_t iterations, const double *start,
                                                             const double *c, size_t n, double limit,
                                                             double *out_pos, uint32_t *out_iters) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    T re = (T)start[2 * k], im = (T)start[2 * k + 1], r2, i2;
    const T lim = (T)limit;
    const uint32_t it = orbit<T>(iterations, re, im, (T)c[2 * k], (T)c[2 * k + 1], lim * lim, r2, i2);
    out_pos[2 * k] = (double)re;
    out_pos[2 * k + 1] = (double)im;
    out_iters[k] = it;
}

/* ---- view sample: which kernel suits this image? ------------------------------------------------------
 *
 * One wave renders ONE 8x8 tile of the launch — `side` x `side` of them on a regular grid over it — through
 * recursive()'s plain loop, capped at `cap_s` iterations, and adds to six device counters: executed iterations of
 * its 64 pixels, 64 x the largest of them (what a wave that keeps a tile to its end pays), tiles, lanes that hit
 * the cap, lanes the two-pass render would hand over after its first episode, and the lane-iterations that finishing
 * those in place would waste — what fr_api.hip: choose_kernel decides on.  The last wave to finish copies the
 * totals to `result` (host-mapped) and zeroes the counters for the next sample.  Colours nothing, stores nothing. */
template <typename T>
__global__ __launch_bounds__(64) void view_sample_kernel(const fr_kparams p, uint32_t side, uint32_t cap_s, uint32_t episode, uint32_t keep,
                                                         unsigned long long *counters, unsigned long long *result,
                                                         unsigned long long tag) {
    const uint32_t lane = threadIdx.x, lx = lane & 7u, ly = lane >> 3;
    const uint32_t ti = blockIdx.x % side, tj = blockIdx.x / side;
    /* tile origin: the centre of cell (ti, tj) of the grid, aligned down to a multiple of 8 */
    const uint32_t col0 = (uint32_t)(((uint64_t)(2u * ti + 1u) * p.ncols) / (2u * side)) & ~7u;
    const uint32_t row0 = (uint32_t)(((uint64_t)(2u * tj + 1u) * p.nrows) / (2u * side)) & ~7u;
    const uint32_t cx = col0 + lx, r = row0 + ly;
    const bool valid = cx < p.ncols && r < p.nrows;
    const double width = (double)p.width, height = (double)p.height;
    const uint32_t x = p.x_first + cx * p.x_stride;
    const uint32_t y = p.y_first + (r / p.block_rows) * p.y_stride + r % p.block_rows;
    const double sre = coord_to_space((double)x, height, (width / height) / 2.0, p.pos_re, p.scale_re);
    const double sim = coord_to_space((double)y, height, 0.5, p.pos_im, p.scale_im);
    uint32_t executed = 0;
    if (valid) {
        T re = (T)sre, im = (T)sim, r2, i2;
        const T cre = p.algo == 2 ? (T)p.julia_re : re, cim = p.algo == 2 ? (T)p.julia_im : im;
        const T lim = (T)p.limit;
        const uint32_t it = orbit<T>(cap_s, re, im, cre, cim, lim * lim, r2, i2);
        executed = it < cap_s ? it + 1u : cap_s;
    }
    const uint32_t mx = wave_max_u32(executed);
    unsigned long long sum = executed;
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    const unsigned long long capped = (unsigned long long)__builtin_popcountll(__ballot(valid && executed == cap_s));
    /* what the two-pass render would do with this tile: episodes (the first pass's own schedule: `episode` iterations,
     * doubling from the ninth on, up to 16 x) for as long as at least `keep` lanes are still running; at the first
     * boundary where fewer are, it hands them over.  Finishing them in place instead costs the wave (longest - boundary)
     * more iterations, of which only the running lanes' own are useful */
    uint32_t e = episode, len = episode, nrun = 0;
    bool hands_over = false;
    while (e < mx) { /* wave-uniform */
        nrun = (uint32_t)__builtin_popcountll(__ballot(executed > e));
        if (nrun < keep) {
            hands_over = nrun > 0u;
            break;
        }
        if (e >= 8u * episode && len < 16u * episode) len += len;
        e += len;
    }
    unsigned long long rest = executed > e ? executed - e : 0u;
    for (int off = 32; off > 0; off >>= 1) rest += __shfl_down(rest, off, 64);
    if (lane == 0) {
        atomicAdd(counters + 0, sum);
        atomicAdd(counters + 1, 64ull * mx);
        atomicAdd(counters + 2, 1ull);
        atomicAdd(counters + 3, capped);
        if (hands_over) {
            atomicAdd(counters + 4, (unsigned long long)nrun);
            atomicAdd(counters + 5, 64ull * (mx - e) - rest); /* lane-iterations wasted by finishing in place */
            atomicAdd(counters + 6, rest);                    /* iterations the handed-over lanes still have to run */
        }
        __threadfence();
        const unsigned long long done = atomicAdd(counters + 7, 1ull);
        if (done + 1ull == (unsigned long long)gridDim.x) { /* the last wave: publish, reset */
            __threadfence();
            for (int k = 0; k < 7; k++) {
                const unsigned long long v = atomicExch(counters + k, 0ull);
                __hip_atomic_store(result + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            atomicExch(counters + 7, 0ull);
            __threadfence_system();
            /* the tag LAST: a host that polls result[7] for it (the non-blocking sample of GUI-sized frames) reads
             * complete totals once it sees it */
            __hip_atomic_store(result + 7, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

/* palette[i] for i = 0 .. iterations: the outside colour of escape index i when smooth == false
 * (calc/src/lib.rs:228-229 with `iters` an integer) — exactly the per-pixel computation, done once */
__global__ __launch_bounds__(256) void palette_kernel(const fr_kparams p, uint32_t *palette) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > p.iterations) return;
    const ColourConsts cc = make_colour_consts(p);
    uint8_t rgb[3];
    colour_outside_flat(cc, i, rgb);
    palette[i] = (uint32_t)rgb[0] | ((uint32_t)rgb[1] << 8) | ((uint32_t)rgb[2] << 16);
}

/* The colour map alone, over stored recursive() results — what the GUI's exposure / colour sliders
 * need (src/gui.rs:183-203 change only inputs of calc/src/lib.rs:214-234): no orbit is re-run. */
__global__ __launch_bounds__(256) void colour_kernel(const fr_kparams p, const double *z, const uint32_t *iters, size_t n,
                                                   uint8_t *rgb) {
    __shared__ double s_tab[FR_LOG2_N * 3];
    const double *gt = &g_log2_tab[0][0];
    for (uint32_t k = threadIdx.x; k < FR_LOG2_N * 3; k += 256) s_tab[k] = gt[k];
    __syncthreads();
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    uint8_t out[3] = {0, 0, 0};
    if (p.algo == 0 || p.algo == 2) {
        const ColourConsts cc = make_colour_consts(p);
        const double re = z[2 * k], im = z[2 * k + 1];
        colour_of(cc, re * re + im * im, iters[k], s_tab, nullptr, out); /* pos.squared_distance(), :214 */
    }
    rgb[3 * k + 0] = out[0];
    rgb[3 * k + 1] = out[1];
    rgb[3 * k + 2] = out[2];
}

__global__ __launch_bounds__(256) void math_probe_kernel(int which, const double *in, double *out, size_t n) {
    __shared__ double s_tab[FR_LOG2_N * 3];
    const double *gt = &g_log2_tab[0][0];
    for (uint32_t k = threadIdx.x; k < FR_LOG2_N * 3; k += 256) s_tab[k] = gt[k];
    __syncthreads();
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double x = in[k];
    double y;
    if (which == 0)
        y = fr_log2_tab(x, s_tab);
    else if (which == 1)
        y = __builtin_sqrt(x);
    else if (which == 3)
        y = (double)sat_u8_dev(x);
    else if (which == 5) /* the packed cast: (float)x into byte 1 of a word whose other bytes must survive */
        y = (double)sat_u8_pack<1>((float)x, 0xAABBCCDDu);
    else
        y = x / in[(k + 1) % n];
    out[k] = y;
}

/* Test hook: the number of f32 bit patterns in [lo, hi] on which the packed cast (sat_u8_pack) and the plain one
 * (sat_u8_dev) disagree in any byte position. */
__global__ __launch_bounds__(256) void cast_scan_kernel(uint32_t lo, uint32_t hi, unsigned long long *count) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long bad = 0;
    for (uint64_t b = (uint64_t)lo + (uint64_t)blockIdx.x * 256 + threadIdx.x; b <= hi; b += stride) {
        const float f = __builtin_bit_cast(float, (uint32_t)b);
        const uint32_t want = sat_u8_dev(f);
        const uint32_t got = sat_u8_pack<2>(f, sat_u8_pack<1>(f, sat_u8_pack<0>(f, 0xFF000000u)));
        bad += got != (0xFF000000u | want | (want << 8) | (want << 16));
    }
    if (bad) atomicAdd(count, bad);
}

/* Test hook: the colour filter's bracket centre against the f64 path's nu, for EVERY f32 with bit pattern in
 * [lo, hi] (as the (float)dist of colour_outside_filtered): out[0] = the largest |nu32 - nu| seen. */
__global__ __launch_bounds__(256) void nu_scan_kernel(uint32_t lo, uint32_t hi, unsigned long long *worst_bits) {
    __shared__ double s_tab[FR_LOG2_N * 3];
    const double *gt = &g_log2_tab[0][0];
    for (uint32_t k = threadIdx.x; k < FR_LOG2_N * 3; k += 256) s_tab[k] = gt[k];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    double worst = 0.0;
    for (uint64_t b = (uint64_t)lo + (uint64_t)blockIdx.x * 256 + threadIdx.x; b <= hi; b += stride) {
        const float d = __builtin_bit_cast(float, (uint32_t)b);
        const float nu32 = __builtin_amdgcn_logf(__builtin_amdgcn_logf(d) * 0.25f);
        const double nu = fr_log2_tab(fr_log2_tab(__builtin_sqrt((double)d), s_tab) * 0.5, s_tab);
        const double err = __builtin_fabs((double)nu32 - nu);
        worst = err > worst || err != err ? err : worst;
    }
    atomicMax(worst_bits, (unsigned long long)fr_bits_of(worst)); /* non-negative doubles order like their bits */
}

/* see fr_kernels.h: fr_launch_copy_out */
__global__ __launch_bounds__(256) void copy_out_kernel(const uint8_t *src, uint8_t *dst, size_t bytes, unsigned int *counter,
                                                      unsigned long long *flag, unsigned long long seq) {
    /* head: the bytes in front of the first 16-byte boundary of dst (src has the same alignment); tail: what is left */
    const size_t head = (16u - (reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
    const size_t h = head < bytes ? head : bytes;
    const size_t units = (bytes - h) / 16;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src + h);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst + h);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; k + 3 * stride < units; k += 4 * stride) { /* four loads in flight per lane, then four stores */
        const uint4 a = s4[k], b = s4[k + stride], c = s4[k + 2 * stride], d = s4[k + 3 * stride];
        d4[k] = a, d4[k + stride] = b, d4[k + 2 * stride] = c, d4[k + 3 * stride] = d;
    }
    for (; k < units; k += stride) d4[k] = s4[k];
    if (blockIdx.x == 0) {
        for (size_t k = threadIdx.x; k < h; k += 256) dst[k] = src[k];
        const size_t t0 = h + units * 16;
        for (size_t k = t0 + threadIdx.x; k < bytes; k += 256) dst[k] = src[k];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int done = atomicAdd(counter, 1u);
        if (done + 1u == gridDim.x) { /* the last workgroup: everyone's stores are fenced; publish */
            atomicExch(counter, 0u);
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

} /* namespace */

hipError_t fr_launch_copy_out(const void *src, void *dst, size_t bytes, unsigned int *counter, unsigned long long *flag,
                              unsigned long long seq, hipStream_t stream) {
    if (bytes == 0) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) != (reinterpret_cast<uintptr_t>(dst) & 15u)) return hipErrorInvalidValue;
    /* The PCIe link is the limit (~55 GB/s: ~100 KB in flight cover its latency), not the chip: at most 128 workgroups —
     * 512 waves, 64 bytes in flight per lane — so that the next band's render kernel keeps the compute units */
    size_t groups = (bytes / 16 + 256 * 4 - 1) / (256 * 4);
    if (groups < 1) groups = 1;
    if (groups > 128) groups = 128;
    hipLaunchKernelGGL(copy_out_kernel, dim3((uint32_t)groups), dim3(256), 0, stream, static_cast<const uint8_t *>(src),
                       static_cast<uint8_t *>(dst), bytes, counter, flag, seq);
    return hipGetLastError();
}

bool fr_wants_work_queue(const fr_kparams &p, int tile) {
    if (p.cycle_shortcut || (p.algo != 0 && p.algo != 2)) return false;
    /* its main loop is the scaled form in blocks of loop_mode iterations, counted in an f32 */
    if (p.loop_mode == 0 || p.iterations >= (1u << 24)) return false;
    /* only on request: over a whole image (BASELINE C4) it runs 3.5 ms (f32) / 4.9 ms (f64) against the patch-refill
     * kernel's 3.1 / 4.5; the default for large Julia images is the two-pass render, whose second pass this kernel
     * is (fr_wants_two_pass) */
    return tile == 10;
}

bool fr_wants_two_pass(fr_kparams &p, int precision, int tile, int hint) {
    p.first_cap = 0;
    p.first_only = 0;
    if (p.cycle_shortcut || (p.algo != 0 && p.algo != 2)) return false;
    if (p.loop_mode == 0 || p.iterations >= (1u << 24)) return false; /* as for the work-queue kernel */
    if (p.ncols == 0 || p.nrows == 0 || (uint64_t)p.ncols * p.nrows > 0xFFF00000ull) return false;
    if (p.algo == 2 && tile == 0) {
        /* a Julia constant the scaled loop may not run with (a component that is zero, tiny or huge — the dendrite c = i):
         * the first pass would take its plain-loop fallback on every strip; the strip kernel is the better plain loop */
        const double lo = precision == 1 ? 0x1p-30 : 0x1p-300, hi = precision == 1 ? 0x1p30 : 0x1p400;
        const double jr = precision == 1 ? std::fabs((double)(float)p.julia_re) : std::fabs(p.julia_re);
        const double ji = precision == 1 ? std::fabs((double)(float)p.julia_im) : std::fabs(p.julia_im);
        if (!(jr >= lo && jr <= hi && ji >= lo && ji <= hi)) return false;
    }
    if (tile == 0) {
        /* the default dispatch: Julia images from 2048^2 up.  Measured (tools/two_pass_sizes.py, C4's view, f32 /
         * f64, against the strips the default would otherwise pick): 65 536 tiles 0.17 / 0.21 ms against 0.18 / 0.30,
         * 131 072 tiles 0.17 / 0.22 against 0.20 / 0.33; at 32 768 tiles and below the strips win in f32 */
        const uint64_t tiles = (((uint64_t)p.ncols + 7) / 8) * (((uint64_t)p.nrows + 7) / 8);
        if (tiles < 262144 && hint < 1) return false; /* (a MEASURED view may ask for them from 4096 tiles up) */
        if (tiles < 4096) return false;
        /* which of the two suits the IMAGE is measured where that pays (hint: 1 two passes, 0 strips — fr_api.hip:
         * choose_kernel); without a measurement, by the algorithm: Julia views are mostly short orbits with a heavy tail */
        if (hint == 0 || (hint < 0 && p.algo != 2)) return false;
        /* ... and only where a tail can be long: under a cap of 512 the lists have nothing to save (a 256-iteration dust at
         * 2048^2: 0.055 ms in two passes, 0.039 in strips; profiles/r03_kernel_choice_views.txt, mid-size section) */
        if (hint < 0 && p.iterations < 512u) return false;
        p.first_only = hint == 2 ? 1u : 0u;
    } else if (tile == 13 || tile == 16) {
        p.first_only = 1u;
    } else if (tile != 11 && tile != 12 && tile != 14 && tile != 15) {
        return false;
    }
    if (tile == 15 || tile == 16) p.strip_tiles = 4u;
    /* first_cap: a multiple of the loop's block length; the second pass must have something left to do */
    uint32_t k1 = p.two_pass_cap ? p.two_pass_cap : 64u; /* measured on C4: 64 / 48 (tools/sweep_two_pass.py) */
    k1 = (k1 + 3u) & ~3u;
    if (k1 + 8u > p.iterations) return false;
    p.first_cap = k1;
    if (p.first_keep == 0 || p.first_keep > 64) p.first_keep = 48;
    return true;
}

fr_two_pass_layout fr_two_pass_bytes(const fr_kparams &p, int precision, uint32_t sub_capacity) {
    const size_t entries = (size_t)sub_capacity * FR_SURV_QUEUES;
    const size_t pair = precision == 1 ? 8 : 16;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    fr_two_pass_layout l{};
    l.z_off = 0;
    l.pos_off = up(entries * pair);
    l.cnt_off = l.pos_off + up(entries * 8);
    l.c_off = l.cnt_off + up(entries * 4);
    l.counts_off = l.c_off + (p.algo == 2 ? 0 : up(entries * pair));
    l.total = l.counts_off + FR_SURV_QUEUES * FR_SURV_COUNT_STRIDE * sizeof(uint32_t);
    return l;
}

hipError_t fr_launch_nu_scan(uint32_t lo_bits, uint32_t hi_bits, double *out, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nu_scan_kernel, dim3(256 * 16), dim3(256), 0, stream, lo_bits, hi_bits,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t fr_launch_cast_scan(uint32_t lo_bits, uint32_t hi_bits, double *out, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cast_scan_kernel, dim3(256 * 16), dim3(256), 0, stream, lo_bits, hi_bits,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

hipError_t fr_launch_escape(const fr_kparams &p, int precision, int mode, const fr_kout &out, int tile,
                            hipStream_t stream, const char **kernel_name) {
    const char *name = "";
    const hipError_t e = precision == 1 ? launch_precision<float>(p, mode, out, tile, stream, name)
                                        : launch_precision<double>(p, mode, out, tile, stream, name);
    if (kernel_name) *kernel_name = name;
    return e;
}

hipError_t fr_launch_recursive_batch(uint32_t iterations, const double *start, const double *c, size_t n,
                                     double limit, int precision, double *out_pos, uint32_t *out_iters,
                                     hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    if (precision == 1)
        hipLaunchKernelGGL(recursive_batch_kernel<float>, dim3((uint32_t)blocks), dim3(256), 0, stream, iterations,
                           start, c, n, limit, out_pos, out_iters);
    else
        hipLaunchKernelGGL(recursive_batch_kernel<double>, dim3((uint32_t)blocks), dim3(256), 0, stream, iterations,
                           start, c, n, limit, out_pos, out_iters);
    return hipGetLastError();
}

hipError_t fr_launch_colour(const fr_kparams &p, const double *z, const uint32_t *iters, size_t n, uint8_t *rgb,
                            hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(colour_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, p, z, iters, n, rgb);
    return hipGetLastError();
}

hipError_t fr_launch_view_sample(const fr_kparams &p, int precision, uint32_t side, uint32_t cap_s, uint32_t episode, uint32_t keep,
                                 unsigned long long *counters, unsigned long long *result, unsigned long long tag,
                                 hipStream_t stream) {
    if (side == 0 || p.ncols == 0 || p.nrows == 0) return hipErrorInvalidValue;
    if (precision == 1)
        hipLaunchKernelGGL(view_sample_kernel<float>, dim3(side * side), dim3(64), 0, stream, p, side, cap_s, episode, keep, counters, result, tag);
    else
        hipLaunchKernelGGL(view_sample_kernel<double>, dim3(side * side), dim3(64), 0, stream, p, side, cap_s, episode, keep, counters, result, tag);
    return hipGetLastError();
}

hipError_t fr_launch_palette(const fr_kparams &p, uint32_t *palette, hipStream_t stream) {
    const uint32_t entries = p.iterations + 1;
    hipLaunchKernelGGL(palette_kernel, dim3((entries + 255) / 256), dim3(256), 0, stream, p, palette);
    return hipGetLastError();
}

hipError_t fr_launch_math_probe(int which, const double *in, double *out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(math_probe_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, which, in, out, n);
    return hipGetLastError();
}
