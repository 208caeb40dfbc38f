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
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fractal_hip.h"
#include "fr_kernels.h"

static_assert(sizeof(fr_config) == 104 && offsetof(fr_config, limit) == 16 && offsetof(fr_config, inside) == 72 &&
                  offsetof(fr_config, primary_color) == 74 && offsetof(fr_config, color_weight) == 80 &&
                  offsetof(fr_config, julia_set) == 88,
              "fr_config must stay the #[repr(C)] image of calc::Config");

namespace {

thread_local std::string tl_error;

struct Profiling {
    bool enabled = false;
    bool have = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
thread_local Profiling tl_prof;

struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct State {
    std::mutex mu; /* serialises the host-buffer entry points (they share stream + scratch) */
    bool inited = false;
    int device = 0;
    hipStream_t stream = nullptr;      /* kernels of the host-buffer entry points */
    hipStream_t copy_stream = nullptr; /* their D2H copies, overlapped with the next band's kernel */
    std::vector<hipEvent_t> band_done;
    Scratch rgb, z, iters, misc;
};
State g;
std::atomic<int> g_tile{0};
std::atomic<int> g_palette_enabled{1};
std::atomic<int> g_cycle_shortcut{0};
std::atomic<int> g_refill_minrun{32}, g_refill_quit16{8};
std::atomic<int> g_loop_mode{-1}; /* -1 auto, 0 unscaled, 2 / 4 scaled with that check interval */

int fail(int code, const char *what) {
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

#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return fail_hip(e_, #expr);     \
    } while (0)

/* Palette scratch for smooth == false renders: a small ring of device buffers owned by the library
 * (no allocation on the render path, usable from any stream).  A slot is handed out only after the
 * event recorded behind its last user has completed. */
struct PaletteSlot {
    uint32_t *dev = nullptr;
    hipEvent_t done = nullptr;
    bool pending = false; /* `done` was recorded and not yet waited for */
    std::atomic<bool> busy{false};
};
constexpr int kPaletteSlots = 16;
PaletteSlot g_palette_slots[kPaletteSlots];
std::mutex g_palette_mu;
unsigned g_palette_next = 0;

int acquire_palette_slot(PaletteSlot **out) {
    std::lock_guard<std::mutex> lk(g_palette_mu);
    for (int tries = 0; tries < kPaletteSlots; tries++) {
        PaletteSlot &s = g_palette_slots[g_palette_next++ % kPaletteSlots];
        if (s.busy.load()) continue; /* another thread is between acquire and its event record */
        if (!s.dev) {
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s.dev), sizeof(uint32_t) * FR_MAX_PALETTE_ENTRIES));
            HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        }
        if (s.pending) {
            HIP_TRY(hipEventSynchronize(s.done)); /* blocks only if 16 renders are in flight */
            s.pending = false;
        }
        s.busy.store(true);
        *out = &s;
        return FR_OK;
    }
    return fail(FR_ERR_HIP, "no palette slot available");
}

/* called with g.mu held, when the library leaves a device */
void release_palette_slots() {
    std::lock_guard<std::mutex> pl(g_palette_mu);
    for (PaletteSlot &ps : g_palette_slots) {
        if (ps.done) (void)hipEventDestroy(ps.done);
        if (ps.dev) (void)hipFree(ps.dev);
        ps.dev = nullptr;
        ps.done = nullptr;
        ps.pending = false;
        ps.busy.store(false);
    }
}

/* caller holds g.mu */
void release_streams_locked() {
    for (hipEvent_t e : g.band_done) (void)hipEventDestroy(e);
    g.band_done.clear();
    if (g.stream) (void)hipStreamDestroy(g.stream);
    if (g.copy_stream) (void)hipStreamDestroy(g.copy_stream);
    g.stream = nullptr;
    g.copy_stream = nullptr;
}

/* caller holds g.mu */
int init_locked(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(FR_ERR_NO_DEVICE, "no HIP device available (libfractal_hip has no CPU fallback)");
    }
    if (device < 0) device = g.inited ? g.device : 0;
    if (device >= n) return fail(FR_ERR_NO_DEVICE, "device index out of range");
    if (g.inited && g.device == device) return FR_OK;
    if (g.inited) {
        /* switching device: drop state that lives on the old one */
        (void)hipSetDevice(g.device);
        for (Scratch *s : {&g.rgb, &g.z, &g.iters, &g.misc}) {
            if (s->ptr) (void)hipFree(s->ptr);
            *s = Scratch();
        }
        release_streams_locked();
        release_palette_slots();
        g.inited = false;
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
    g.device = device;
    g.inited = true;
    return FR_OK;
}

/* caller holds g.mu; hipSetDevice is per host thread, so every entry point re-asserts it */
int ensure_locked() {
    if (!g.inited) {
        int rc = init_locked(-1);
        if (rc != FR_OK) return rc;
    }
    HIP_TRY(hipSetDevice(g.device));
    return FR_OK;
}

int reserve_locked(Scratch &s, size_t bytes) {
    if (bytes <= s.cap) return FR_OK;
    if (s.ptr) {
        HIP_TRY(hipFree(s.ptr));
        s = Scratch();
    }
    HIP_TRY(hipMalloc(&s.ptr, bytes));
    s.cap = bytes;
    return FR_OK;
}

/* calc::Config -> kernel arguments; the local grid is filled in by the caller */
void fill_params(const fr_config *cfg, fr_kparams &p) {
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
    p.refill_minrun = (uint32_t)g_refill_minrun.load();
    p.refill_quit16 = (uint32_t)g_refill_quit16.load();
    /* the flag bit of the loop's return value needs iterations < 2^31; keep a margin */
    p.cycle_shortcut = (g_cycle_shortcut.load() && cfg->iterations < (1u << 30)) ? 1u : 0u;
}

/* coord_to_space — calc/src/lib.rs:182-184 — evaluated on the host ONLY to bound |c| over a launch
 * (the kernels compute every coordinate themselves). */
double host_coord(double coord, double max, double offset, double pos, double scale) {
    return ((coord / max) - offset) / scale + pos;
}

/* Choose the orbit-loop plan for a launch whose local grid is already set in `p`
 * (see fr_kernels.hip, "orbit loop, scaled form", for what the kernel does with it and why it is
 * exact).  loop_mode = M in {4, 2} and skip_t = T such that
 *     dist_k <= T  and  every |c| component <= Cmax   =>   dist_{k+1..k+M-1} <= limit^2,
 * using dist' <= g(dist) = 2 * (dist + Cmax)^2 * (1 + slack); T is found by inverting g M-1 times
 * from limit^2.  Falls back to the unscaled loop (0) whenever the bound is useless or any
 * parameter is outside the range the scaled form is proven for. */
void plan_loop(const fr_config *cfg, int precision, fr_kparams &p) {
    p.loop_mode = 0;
    p.skip_t = 0.0;
    const int forced = g_loop_mode.load();
    if (forced == 0) return;
    if (cfg->algo != FR_ALGO_MANDELBROT && cfg->algo != FR_ALGO_JULIA) return;
    if (p.ncols == 0 || p.nrows == 0) return;
    const bool f32 = precision == FR_PRECISION_F32;
    const double range_hi = f32 ? 0x1p30 : 0x1p400;
    const double slack = f32 ? 1.0 + 0x1p-18 : 1.0 + 0x1p-30;
    if (!(std::fabs(cfg->limit) <= range_hi)) return; /* also rejects NaN */
    double lim2;
    if (f32) {
        const float lf = (float)cfg->limit;
        lim2 = (double)(lf * lf);
    } else {
        lim2 = cfg->limit * cfg->limit;
    }
    auto mag = [f32](double v) { return std::fabs(f32 ? (double)(float)v : v); };
    double cmax;
    if (cfg->algo == FR_ALGO_JULIA) {
        cmax = std::fmax(mag(cfg->julia_set.re), mag(cfg->julia_set.im));
    } else {
        /* c = pixel coordinate; the map is monotone in x and in y, so the extremes are at the ends */
        const double w = (double)cfg->width, h = (double)cfg->height;
        const uint64_t x_last = (uint64_t)p.x_first + (uint64_t)(p.ncols - 1) * p.x_stride;
        const uint32_t r_last = p.nrows - 1;
        const uint64_t y_last = (uint64_t)p.y_first + (uint64_t)(r_last / p.block_rows) * p.y_stride + r_last % p.block_rows;
        if (x_last > 0xFFFFFFFFull || y_last > 0xFFFFFFFFull) return;
        cmax = 0.0;
        for (double x : {(double)p.x_first, (double)x_last})
            cmax = std::fmax(cmax, mag(host_coord(x, h, (w / h) / 2.0, cfg->pos.re, cfg->scale.re)));
        for (double y : {(double)p.y_first, (double)y_last})
            cmax = std::fmax(cmax, mag(host_coord(y, h, 0.5, cfg->pos.im, cfg->scale.im)));
    }
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
            return;
        }
    }
}

int check_rows(const fr_config *cfg, uint32_t y0, uint32_t y1) {
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

/* device-pointer render of an arbitrary local grid; no locking, no global scratch: re-entrant */
int render_device(const fr_config *cfg, fr_kparams &p, int precision, void *d_out, hipStream_t stream) {
    plan_loop(cfg, precision, p);
    /* smooth == false: the outside colour is a function of the escape index alone, so build the
     * (iterations + 1)-entry palette once (into a library-owned slot) and let every workgroup stage it
     * in LDS.  Larger palettes would cost occupancy; they are computed per pixel instead. */
    const bool escape_algo = cfg->algo == FR_ALGO_MANDELBROT || cfg->algo == FR_ALGO_JULIA;
    PaletteSlot *slot = nullptr;
    if (!cfg->smooth && escape_algo && g_palette_enabled.load() && cfg->iterations < FR_MAX_PALETTE_ENTRIES &&
        g_tile.load() <= 9) {
        int rc = acquire_palette_slot(&slot);
        if (rc != FR_OK) return rc;
        p.palette = slot->dev;
        p.palette_entries = cfg->iterations + 1;
        hipError_t e = fr_launch_palette(p, slot->dev, stream);
        if (e != hipSuccess) {
            slot->busy.store(false);
            return fail_hip(e, "fr_launch_palette");
        }
    }
    /* the slot may be reused once everything enqueued so far on `stream` has run */
    struct SlotGuard {
        PaletteSlot *slot;
        hipStream_t stream;
        ~SlotGuard() {
            if (!slot) return;
            slot->pending = hipEventRecord(slot->done, stream) == hipSuccess;
            slot->busy.store(false);
        }
    } guard{slot, stream};
    fr_kout out{};
    out.rgb = static_cast<uint8_t *>(d_out);
    Profiling &pr = tl_prof;
    if (pr.enabled) {
        if (!pr.e0) {
            HIP_TRY(hipEventCreate(&pr.e0));
            HIP_TRY(hipEventCreate(&pr.e1));
        }
        HIP_TRY(hipEventRecord(pr.e0, stream));
    }
    HIP_TRY(fr_launch_escape(p, precision, FR_OUT_RGB, out, g_tile.load(), stream));
    if (pr.enabled) {
        HIP_TRY(hipEventRecord(pr.e1, stream));
        pr.have = true;
    }
    return FR_OK;
}

} /* namespace */

extern "C" {

int fr_abi_version(void) { return FR_ABI_VERSION; }

const char *fr_last_error(void) { return tl_error.c_str(); }

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
    std::lock_guard<std::mutex> lk(g.mu);
    return init_locked(device);
}

int fr_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.inited) return FR_OK;
    (void)hipSetDevice(g.device);
    if (g.stream) (void)hipStreamSynchronize(g.stream);
    for (Scratch *s : {&g.rgb, &g.z, &g.iters, &g.misc}) {
        if (s->ptr) (void)hipFree(s->ptr);
        *s = Scratch();
    }
    release_streams_locked();
    release_palette_slots();
    g.inited = false;
    return FR_OK;
}

int fr_device_name(char *buf, size_t buf_len) {
    if (!buf || buf_len == 0) return fail(FR_ERR_INVALID_ARGUMENT, "buf is NULL or empty");
    std::lock_guard<std::mutex> lk(g.mu);
    int rc = ensure_locked();
    if (rc != FR_OK) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g.device));
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

static int render_rows_device(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                              size_t out_len, void *hip_stream, unsigned bytes_per_pixel) {
    int rc = check_rows(cfg, y0, y1);
    if (rc == FR_OK) rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    const size_t need = (size_t)bytes_per_pixel * cfg->width * (size_t)(y1 - y0);
    if (need == 0) return FR_OK;
    if (!d_out) return fail(FR_ERR_INVALID_ARGUMENT, "d_out is NULL");
    if (out_len < need) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < bytes_per_pixel*width*(y1-y0)");
    if (bytes_per_pixel == 4 && (reinterpret_cast<uintptr_t>(d_out) & 3u))
        return fail(FR_ERR_INVALID_ARGUMENT, "RGBA8 output must be 4-byte aligned");
    fr_kparams p;
    fill_params(cfg, p);
    p.nrows = y1 - y0;
    p.y_first = y0;
    p.block_rows = p.nrows;
    p.y_stride = 0;
    p.out_rgba = bytes_per_pixel == 4 ? 1u : 0u;
    return render_device(cfg, p, precision, d_out, static_cast<hipStream_t>(hip_stream));
}

int fr_render_rows_rgb8_device(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                               size_t out_len, void *hip_stream) {
    return render_rows_device(cfg, precision, y0, y1, d_out, out_len, hip_stream, 3);
}

int fr_render_rows_rgba8_device(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, void *d_out,
                                size_t out_len, void *hip_stream) {
    return render_rows_device(cfg, precision, y0, y1, d_out, out_len, hip_stream, 4);
}

int fr_render_rows_rgba8(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out, size_t out_len) {
    int rc = check_rows(cfg, y0, y1);
    if (rc == FR_OK) rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    const size_t need = (size_t)4 * cfg->width * (size_t)(y1 - y0);
    if (need == 0) return FR_OK;
    if (!out) return fail(FR_ERR_INVALID_ARGUMENT, "out is NULL");
    if (out_len < need) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 4*width*(y1-y0)");
    std::lock_guard<std::mutex> lk(g.mu);
    rc = ensure_locked();
    if (rc != FR_OK) return rc;
    rc = reserve_locked(g.rgb, need);
    if (rc != FR_OK) return rc;
    rc = fr_render_rows_rgba8_device(cfg, precision, y0, y1, g.rgb.ptr, need, g.stream);
    if (rc != FR_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, g.rgb.ptr, need, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return FR_OK;
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

int fr_render_block_cyclic_range_rgb8_device(const fr_config *cfg, int precision, uint32_t block_rows,
                                             uint32_t first_block, uint32_t block_stride, uint32_t max_blocks,
                                             int dest_is_image, void *d_out, size_t out_len, void *hip_stream,
                                             uint64_t *rows_written) {
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
    fill_params(cfg, p);
    p.nrows = (uint32_t)rows;
    p.block_rows = block_rows;
    p.y_first = first_block * block_rows;
    p.y_stride = block_rows * block_stride;
    p.out_in_place = dest_is_image ? 1u : 0u;
    return render_device(cfg, p, precision, d_out, static_cast<hipStream_t>(hip_stream));
}

int fr_render_block_cyclic_rgb8_device(const fr_config *cfg, int precision, uint32_t block_rows,
                                       uint32_t first_block, uint32_t block_stride, void *d_out, size_t out_len,
                                       void *hip_stream, uint64_t *rows_written) {
    return fr_render_block_cyclic_range_rgb8_device(cfg, precision, block_rows, first_block, block_stride, 0, 0,
                                                    d_out, out_len, hip_stream, rows_written);
}

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
    std::lock_guard<std::mutex> lk(g.mu);
    int rc = ensure_locked();
    if (rc != FR_OK) return rc;
    rc = reserve_locked(g.rgb, need);
    if (rc != FR_OK) return rc;
    rc = fr_render_block_cyclic_rgb8_device(cfg, precision, block_rows, first_block, block_stride, g.rgb.ptr, need,
                                            g.stream, nullptr);
    if (rc != FR_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, g.rgb.ptr, need, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return FR_OK;
}

int fr_render_rows_rgb8(const fr_config *cfg, int precision, uint32_t y0, uint32_t y1, uint8_t *out,
                        size_t out_len) {
    int rc = check_rows(cfg, y0, y1);
    if (rc == FR_OK) rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    const size_t row_bytes = (size_t)3 * cfg->width;
    const size_t need = row_bytes * (size_t)(y1 - y0);
    if (need == 0) return FR_OK;
    if (!out) return fail(FR_ERR_INVALID_ARGUMENT, "out is NULL");
    if (out_len < need) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 3*width*(y1-y0)");
    std::lock_guard<std::mutex> lk(g.mu);
    rc = ensure_locked();
    if (rc != FR_OK) return rc;
    rc = reserve_locked(g.rgb, need);
    if (rc != FR_OK) return rc;
    uint8_t *scratch = static_cast<uint8_t *>(g.rgb.ptr);
    /* Large images: pin the caller's buffer for the duration of the call and render in bands of
     * ~64 MiB, so band k's DMA to the host (PCIe, ~57 GB/s into pinned memory) runs on the copy
     * stream while band k+1 renders — the call costs about max(kernel, copy), not their sum.
     * (Async copies into PAGEABLE memory are staged by the runtime and measured slower than one
     * plain copy, so without the pin — or for small images — it is one kernel + one copy.) */
    const bool pinned = need >= ((size_t)16 << 20) && hipHostRegister(out, need, hipHostRegisterDefault) == hipSuccess;
    if (!pinned) {
        (void)hipGetLastError();
        rc = fr_render_rows_rgb8_device(cfg, precision, y0, y1, scratch, need, g.stream);
        if (rc != FR_OK) return rc;
        HIP_TRY(hipMemcpyAsync(out, scratch, need, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        return FR_OK;
    }
    const size_t band_target = (size_t)64 << 20;
    uint64_t bands = (need + band_target - 1) / band_target;
    if (bands > 64) bands = 64;
    uint64_t band_rows = ((uint64_t)(y1 - y0) + bands - 1) / bands;
    band_rows = (band_rows + 7) / 8 * 8; /* whole 8-row tiles */
    hipError_t err = hipSuccess;
    const char *what = "";
    size_t b = 0;
    for (uint64_t ya = y0; ya < y1 && rc == FR_OK && err == hipSuccess; ya += band_rows, b++) {
        const uint32_t yb = (uint32_t)(ya + band_rows < y1 ? ya + band_rows : y1);
        const size_t off = row_bytes * (size_t)(ya - y0), bytes = row_bytes * (size_t)(yb - ya);
        rc = fr_render_rows_rgb8_device(cfg, precision, (uint32_t)ya, yb, scratch + off, bytes, g.stream);
        if (rc != FR_OK) break;
        if (b >= g.band_done.size()) {
            hipEvent_t e;
            if ((err = hipEventCreateWithFlags(&e, hipEventDisableTiming)) != hipSuccess) {
                what = "hipEventCreateWithFlags";
                break;
            }
            g.band_done.push_back(e);
        }
        if ((err = hipEventRecord(g.band_done[b], g.stream)) != hipSuccess) what = "hipEventRecord";
        else if ((err = hipStreamWaitEvent(g.copy_stream, g.band_done[b], 0)) != hipSuccess) what = "hipStreamWaitEvent";
        else if ((err = hipMemcpyAsync(out + off, scratch + off, bytes, hipMemcpyDeviceToHost, g.copy_stream)) != hipSuccess)
            what = "hipMemcpyAsync";
    }
    /* always drain both streams and unpin before returning, error or not */
    hipError_t e1 = hipStreamSynchronize(g.stream);
    hipError_t e2 = hipStreamSynchronize(g.copy_stream);
    (void)hipHostUnregister(out);
    if (rc != FR_OK) return rc;
    if (err != hipSuccess) return fail_hip(err, what);
    if (e1 != hipSuccess) return fail_hip(e1, "hipStreamSynchronize(stream)");
    if (e2 != hipSuccess) return fail_hip(e2, "hipStreamSynchronize(copy_stream)");
    return FR_OK;
}

int fr_render_rgb8(const fr_config *cfg, uint8_t *out, size_t out_len) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    return fr_render_rows_rgb8(cfg, FR_PRECISION_F64, 0, cfg->height, out, out_len);
}

int fr_pixel_p(const fr_config *cfg, int precision, uint32_t x, uint32_t y, fr_rgb *out) {
    if (!cfg || !out) return fail(FR_ERR_INVALID_ARGUMENT, "cfg or out is NULL");
    int rc = check_precision(precision);
    if (rc != FR_OK) return rc;
    /* get_recursive_pixel takes any u32 x, y — it does not clamp to width/height */
    std::lock_guard<std::mutex> lk(g.mu);
    rc = ensure_locked();
    if (rc != FR_OK) return rc;
    rc = reserve_locked(g.misc, 256);
    if (rc != FR_OK) return rc;
    fr_kparams p;
    fill_params(cfg, p);
    p.ncols = 1;
    p.nrows = 1;
    p.x_first = x;
    p.y_first = y;
    plan_loop(cfg, precision, p);
    fr_kout o{};
    o.rgb = static_cast<uint8_t *>(g.misc.ptr);
    HIP_TRY(fr_launch_escape(p, precision, FR_OUT_RGB, o, g_tile.load(), g.stream));
    uint8_t rgb[3];
    HIP_TRY(hipMemcpyAsync(rgb, g.misc.ptr, 3, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
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
    std::lock_guard<std::mutex> lk(g.mu);
    rc = ensure_locked();
    if (rc != FR_OK) return rc;
    const size_t zb = n * sizeof(fr_imaginary);
    rc = reserve_locked(g.z, 3 * zb);
    if (rc == FR_OK) rc = reserve_locked(g.iters, n * sizeof(uint32_t));
    if (rc != FR_OK) return rc;
    double *d_start = static_cast<double *>(g.z.ptr);
    double *d_c = d_start + 2 * n;
    double *d_pos = d_c + 2 * n;
    HIP_TRY(hipMemcpyAsync(d_start, start, zb, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_c, c, zb, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(fr_launch_recursive_batch(iterations, d_start, d_c, n, limit, precision, d_pos,
                                      static_cast<uint32_t *>(g.iters.ptr), g.stream));
    HIP_TRY(hipMemcpyAsync(out_pos, d_pos, zb, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipMemcpyAsync(out_iters, g.iters.ptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
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
    std::lock_guard<std::mutex> lk(g.mu);
    rc = ensure_locked();
    if (rc != FR_OK) return rc;
    if (z_re_im) rc = reserve_locked(g.z, npx * 2 * sizeof(double));
    if (rc == FR_OK && iters) rc = reserve_locked(g.iters, npx * sizeof(uint32_t));
    if (rc != FR_OK) return rc;
    fr_kparams p;
    fill_params(cfg, p);
    p.nrows = y1 - y0;
    p.y_first = y0;
    p.block_rows = p.nrows;
    p.y_stride = 0;
    plan_loop(cfg, precision, p);
    fr_kout o{};
    o.z = z_re_im ? static_cast<double *>(g.z.ptr) : nullptr;
    o.iters = iters ? static_cast<uint32_t *>(g.iters.ptr) : nullptr;
    HIP_TRY(fr_launch_escape(p, precision, FR_OUT_ESCAPE, o, g_tile.load(), g.stream));
    if (z_re_im) HIP_TRY(hipMemcpyAsync(z_re_im, g.z.ptr, npx * 2 * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    if (iters) HIP_TRY(hipMemcpyAsync(iters, g.iters.ptr, npx * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return FR_OK;
}

int fr_colour_rgb8(const fr_config *cfg, const double *z_re_im, const uint32_t *iters, size_t n, uint8_t *out,
                   size_t out_len) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (n == 0) return FR_OK;
    if (!z_re_im || !iters || !out) return fail(FR_ERR_INVALID_ARGUMENT, "NULL array");
    if (out_len < 3 * n) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 3*n");
    std::lock_guard<std::mutex> lk(g.mu);
    int rc = ensure_locked();
    if (rc != FR_OK) return rc;
    rc = reserve_locked(g.z, n * 2 * sizeof(double));
    if (rc == FR_OK) rc = reserve_locked(g.iters, n * sizeof(uint32_t));
    if (rc == FR_OK) rc = reserve_locked(g.rgb, 3 * n);
    if (rc != FR_OK) return rc;
    fr_kparams p;
    fill_params(cfg, p);
    HIP_TRY(hipMemcpyAsync(g.z.ptr, z_re_im, n * 2 * sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(g.iters.ptr, iters, n * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
    HIP_TRY(fr_launch_colour(p, static_cast<const double *>(g.z.ptr), static_cast<const uint32_t *>(g.iters.ptr), n,
                             static_cast<uint8_t *>(g.rgb.ptr), g.stream));
    HIP_TRY(hipMemcpyAsync(out, g.rgb.ptr, 3 * n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return FR_OK;
}

int fr_colour_rgb8_device(const fr_config *cfg, const void *d_z_re_im, const void *d_iters, size_t n, void *d_out,
                          size_t out_len, void *hip_stream) {
    if (!cfg) return fail(FR_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (n == 0) return FR_OK;
    if (!d_z_re_im || !d_iters || !d_out) return fail(FR_ERR_INVALID_ARGUMENT, "NULL array");
    if (out_len < 3 * n) return fail(FR_ERR_BUFFER_TOO_SMALL, "out_len < 3*n");
    fr_kparams p;
    fill_params(cfg, p);
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
    fr_kparams p;
    fill_params(cfg, p);
    p.ncols = (uint32_t)(((uint64_t)cfg->width + sx - 1) / sx);
    p.x_stride = sx;
    const uint64_t yf = ((uint64_t)y0 + sy - 1) / sy * sy; /* first sampled row >= y0 */
    p.nrows = yf < y1 ? (uint32_t)((y1 - 1 - yf) / sy + 1) : 0;
    p.y_first = (uint32_t)yf;
    p.block_rows = 1;
    p.y_stride = sy;
    if (pixels) *pixels = (uint64_t)p.ncols * p.nrows;
    if (p.ncols == 0 || p.nrows == 0) return FR_OK;
    std::lock_guard<std::mutex> lk(g.mu);
    rc = ensure_locked();
    if (rc != FR_OK) return rc;
    const size_t slot_bytes = sizeof(unsigned long long) * FR_COUNT_SLOTS;
    rc = reserve_locked(g.misc, slot_bytes);
    if (rc != FR_OK) return rc;
    HIP_TRY(hipMemsetAsync(g.misc.ptr, 0, slot_bytes, g.stream));
    plan_loop(cfg, precision, p);
    fr_kout o{};
    o.count = static_cast<unsigned long long *>(g.misc.ptr);
    HIP_TRY(fr_launch_escape(p, precision, FR_OUT_COUNT, o, g_tile.load(), g.stream));
    std::vector<unsigned long long> host(FR_COUNT_SLOTS);
    HIP_TRY(hipMemcpyAsync(host.data(), g.misc.ptr, slot_bytes, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    unsigned long long sum = 0;
    for (unsigned long long v : host) sum += v;
    *total = sum;
    return FR_OK;
}

int fr_set_profiling(int enabled) {
    tl_prof.enabled = enabled != 0;
    if (!enabled) tl_prof.have = false;
    return FR_OK;
}

int fr_last_kernel_ms(float *ms) {
    if (!ms) return fail(FR_ERR_INVALID_ARGUMENT, "ms is NULL");
    if (!tl_prof.have) return fail(FR_ERR_INVALID_ARGUMENT, "no profiled kernel on this thread");
    HIP_TRY(hipEventSynchronize(tl_prof.e1));
    HIP_TRY(hipEventElapsedTime(ms, tl_prof.e0, tl_prof.e1));
    return FR_OK;
}

int fr_set_tile(int tile) {
    switch (tile) {
    case 0:
    case 1:
    case 2:
    case 4:
    case 8:
    case 9:
    case 6401:
    case 3202:
    case 1604:
    case 808:
        g_tile.store(tile);
        return FR_OK;
    default:
        return fail(FR_ERR_INVALID_ARGUMENT, "tile must be 0, 1, 2, 4, 8, 9, 6401, 3202, 1604 or 808");
    }
}

int fr_set_refill_policy(int minrun, int quit16) {
    if (minrun < 0 || quit16 < 1 || quit16 > 16) return fail(FR_ERR_INVALID_ARGUMENT, "minrun >= 0, 1 <= quit16 <= 16");
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

int fr_set_loop_mode(int mode) {
    if (mode != -1 && mode != 0 && mode != 2 && mode != 4)
        return fail(FR_ERR_INVALID_ARGUMENT, "loop mode must be -1 (auto), 0, 2 or 4");
    g_loop_mode.store(mode);
    return FR_OK;
}

/* test hook (not part of the reference surface): elementwise device log2 (which=0), sqrt (1) or
 * x[k]/x[k+1] (2) over host arrays, so tests can compare device arithmetic with the host's. */
int fr_debug_math(int which, const double *in, double *out, size_t n) {
    if (n == 0) return FR_OK;
    if (!in || !out) return fail(FR_ERR_INVALID_ARGUMENT, "NULL array");
    std::lock_guard<std::mutex> lk(g.mu);
    int rc = ensure_locked();
    if (rc != FR_OK) return rc;
    rc = reserve_locked(g.z, 2 * n * sizeof(double));
    if (rc != FR_OK) return rc;
    double *d_in = static_cast<double *>(g.z.ptr), *d_out = d_in + n;
    HIP_TRY(hipMemcpyAsync(d_in, in, n * sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIP_TRY(fr_launch_math_probe(which, d_in, d_out, n, g.stream));
    HIP_TRY(hipMemcpyAsync(out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return FR_OK;
}

} /* extern "C" */
