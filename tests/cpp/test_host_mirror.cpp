// Known-answer tests through the C++ host mirror of the calc API.
// Built and run by tests/test_gpu_parity.py::test_cpp_host_mirror on the GPU box.
#include <cstdio>
#include <cstdlib>

#include "fractal.hpp"

#define EXPECT(cond)                                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                              \
        }                                                              \
    } while (0)

int main() {
    using namespace fractal;
    // KAT-6: RGB::new(r, b, g)
    EXPECT((RGB::make(40, 40, 255) == RGB{40, 255, 40}));
    Config cfg = Config::make(Algo::Mandelbrot);
    EXPECT(cfg.width == 2000 && cfg.height == 1000 && cfg.iterations == 50 && cfg.exposure == 2.0);
    EXPECT(cfg.primary_color.g == 255 && cfg.secondary_color.b == 170);
    // KAT-1..5: recursive
    auto r = recursive(50, {2, 0}, {2, 0}, 65536);
    EXPECT(r.first.re == 2090918.0 && r.first.im == 0.0 && r.second == 3);
    r = recursive(50, {2, 0}, {2, 0}, 2);
    EXPECT(r.first.re == 6.0 && r.second == 0);
    r = recursive(50, {-2, 0}, {-2, 0}, 65536);
    EXPECT(r.first.re == 2.0 && r.second == 50);
    r = recursive(51, {-1, 0}, {-1, 0}, 65536);
    EXPECT(r.first.re == 0.0 && r.second == 51);
    // KAT-7: the 4x4 image
    cfg.width = cfg.height = 4;
    cfg.scale.re = cfg.scale.im = 0.25;
    EXPECT((get_recursive_pixel(cfg, 0, 2) == RGB{83, 83, 255}));
    std::vector<RGB> img = get_image(cfg);
    EXPECT(img.size() == 16);
    const RGB row2[4] = {{83, 83, 255}, {240, 170, 0}, {0, 0, 0}, {2, 2, 18}};
    for (int x = 0; x < 4; x++) EXPECT(img[2 * 4 + x] == row2[x]);
    EXPECT((img[0] == RGB{0, 0, 5}) && (img[5] == RGB{3, 3, 22}));
    // errors surface as exceptions, never aborts
    bool threw = false;
    try {
        fr_rgb px;
        check(fr_pixel(nullptr, 0, 0, &px));
    } catch (const Error &e) {
        threw = e.code() == FR_ERR_INVALID_ARGUMENT;
    }
    EXPECT(threw);
    std::puts("cpp host mirror ok");
    return 0;
}
