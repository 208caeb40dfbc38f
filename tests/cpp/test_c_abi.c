/* The public header must be plain C (it is what a Rust bindgen / cgo / JNI binding would consume).
 * Compiled with gcc -std=c99 -pedantic -Werror by tests/test_cpp_host.py; run on the GPU box. */
#include <stdio.h>
#include <string.h>

#include "fractal_hip.h"

int main(void) {
    fr_config cfg;
    fr_imaginary start = {2.0, 0.0}, c = {2.0, 0.0}, pos;
    uint32_t iters = 0;
    unsigned char image[4 * 4 * 3];
    int n = 0, rc;

    if (fr_abi_version() != FR_ABI_VERSION) return 10;
    if (sizeof(fr_config) != 104 || sizeof(fr_rgb) != 3) return 11;
    fr_config_new(&cfg, FR_ALGO_MANDELBROT);
    if (cfg.iterations != 50 || cfg.primary_color.g != 255) return 12;
    if (fr_device_count(&n) != FR_OK) return 13;
    { /* the loop plan is host arithmetic: it answers without a device (include/fractal_hip.h: fr_debug_loop_plan) */
        uint32_t mode = 99, quiet = 99;
        double skip = -1.0;
        fr_config view;
        fr_config_new(&view, FR_ALGO_MANDELBROT);
        view.width = 1920, view.height = 1080, view.iterations = 1024, view.pos.re = -0.6;
        if (fr_debug_loop_plan(&view, FR_PRECISION_F64, &mode, &skip, &quiet) != FR_OK) return 30;
        if (mode != 4 || !(skip > 4.5 && skip < 7.5) || quiet != 16) return 31; /* blocks of four, speculative after 16 quiet iterations */
        view.limit = 2.0; /* limit^2 = 4 < 16: nothing may be skipped, nothing speculated */
        if (fr_debug_loop_plan(&view, FR_PRECISION_F64, &mode, &skip, &quiet) != FR_OK || mode != 0 || quiet != 0) return 32;
        if (fr_debug_loop_plan(&view, FR_PRECISION_F64, 0, &skip, &quiet) == FR_OK) return 33;
    }
    if (n == 0) { /* no GPU: every compute call must fail loudly */
        rc = fr_recursive(50, start, c, 65536.0, &pos, &iters);
        if (rc != FR_ERR_NO_DEVICE || strlen(fr_last_error()) == 0) return 14;
        puts("c abi ok (no device)");
        return 0;
    }
    /* KAT-1 and the 4x4 image of SURVEY.md §8c */
    if (fr_recursive(50, start, c, 65536.0, &pos, &iters) != FR_OK || pos.re != 2090918.0 || iters != 3) return 15;
    cfg.width = cfg.height = 4;
    cfg.scale.re = cfg.scale.im = 0.25;
    if (fr_render_rgb8(&cfg, image, sizeof image) != FR_OK) return 16;
    if (image[3 * 8 + 0] != 83 || image[3 * 8 + 1] != 83 || image[3 * 8 + 2] != 255) return 17; /* pixel (0, 2) */
    if (fr_render_rgb8(&cfg, image, 5) != FR_ERR_BUFFER_TOO_SMALL) return 18;
    /* get_image across a device set from this one process: three logical devices on GPU 0, a C2-shaped
     * and a ragged image, byte-identical to the single render (and per-call options) */
    {
        static unsigned char one[3 * 257 * 193], many[3 * 257 * 193], sq[3 * 256 * 256], sq3[3 * 256 * 256];
        int devs[3] = {0, 0, 0};
        fr_multi_stats st;
        fr_render_opts opts;
        if (fr_init_devices(devs, 3) != FR_OK) return 20;
        if (fr_multi_device_count(&n) != FR_OK || n != 3) return 21;
        fr_config_new(&cfg, FR_ALGO_MANDELBROT);
        cfg.pos.re = -0.6;
        cfg.exposure = 5.0;
        cfg.iterations = 200;
        cfg.width = 257;
        cfg.height = 193;
        if (fr_render_rgb8(&cfg, one, sizeof one) != FR_OK) return 22;
        if (fr_render_rgb8_multi(&cfg, FR_PRECISION_F64, 8, many, sizeof many) != FR_OK) return 23;
        if (memcmp(one, many, sizeof one) != 0) return 24;
        if (fr_multi_last_stats(&st) != FR_OK || st.n_devices != 3 || st.rows[0] + st.rows[1] + st.rows[2] != 193) return 25;
        cfg.width = cfg.height = 256;
        if (fr_render_rgb8(&cfg, sq, sizeof sq) != FR_OK) return 26;
        if (fr_render_rgb8_multi(&cfg, FR_PRECISION_F64, 0, sq3, sizeof sq3) != FR_OK) return 27;
        if (memcmp(sq, sq3, sizeof sq) != 0) return 28;
        fr_render_opts_init(&opts);
        if (opts.size != sizeof opts || opts.tile != 0) return 29;
        opts.tile = 9;
        opts.cycle_shortcut = 1;
        memset(sq3, 0, sizeof sq3);
        if (fr_render_rows_rgb8_opts(&cfg, FR_PRECISION_F64, 0, 256, sq3, sizeof sq3, &opts) != FR_OK) return 30;
        if (memcmp(sq, sq3, sizeof sq) != 0) return 31;
        if (fr_render_rgb8_multi(&cfg, FR_PRECISION_F64, 12, sq3, sizeof sq3) != FR_ERR_INVALID_ARGUMENT) return 32;
        if (fr_shutdown() != FR_OK) return 33;
    }
    puts("c abi ok");
    return 0;
}
