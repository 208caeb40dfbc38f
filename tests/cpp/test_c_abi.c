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
    puts("c abi ok");
    return 0;
}
