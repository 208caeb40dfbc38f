// fractal.hpp — C++ host-side mirror of the reference's public surface for the escape-time path,
// over the C ABI of include/fractal_hip.h (libfractal_hip.so, gfx950 kernels).
//
// The reference is a Rust crate and this image has no Rust toolchain, so the host side above the
// C ABI is written in C++ with the reference's names, argument meaning and error behaviour:
//
//   calc::Algo                     calc/src/lib.rs:150-154   fractal::Algo
//   calc::Imaginary                calc/src/lib.rs:79-117    fractal::Imaginary
//   calc::RGB, RGB::new(r, b, g)   calc/src/lib.rs:121-131   fractal::RGB, RGB::make(r, b, g)
//   calc::Config, Config::new      calc/src/lib.rs:21-69     fractal::Config, Config::make(algo)
//   calc::recursive                calc/src/lib.rs:245-257   fractal::recursive
//   calc::get_recursive_pixel      calc/src/lib.rs:199-235   fractal::get_recursive_pixel
//   get_image                      src/lib.rs:253-270        fractal::get_image (one GPU, or the set chosen with use_devices)
//   get_image, BarnsleyFern arm    src/lib.rs:271-319,417-463  fractal::get_image_fern
//
// The reference's functions are infallible and panic on misuse; here a failing C-ABI call throws
// fractal::Error (there is no CPU fallback to fall back to).
#ifndef FRACTAL_HPP
#define FRACTAL_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "fractal_hip.h"

namespace fractal {

class Error : public std::runtime_error {
  public:
    Error(int code, const char *msg) : std::runtime_error(std::string(msg ? msg : "")), code_(code) {}
    int code() const { return code_; }

  private:
    int code_;
};

inline void check(int rc) {
    if (rc != FR_OK) throw Error(rc, fr_last_error());
}

enum class Algo : uint32_t { Mandelbrot = FR_ALGO_MANDELBROT, BarnsleyFern = FR_ALGO_BARNSLEY_FERN, Julia = FR_ALGO_JULIA };

// calc/src/lib.rs:79-82; layout-compatible with fr_imaginary
struct Imaginary {
    double re = 0.0, im = 0.0;
    static constexpr Imaginary zero() { return Imaginary{0.0, 0.0}; }  // Imaginary::ZERO
    fr_imaginary c() const { return fr_imaginary{re, im}; }
};

// calc/src/lib.rs:121-125; layout-compatible with fr_rgb (the STORED fields)
struct RGB {
    uint8_t r = 0, g = 0, b = 0;
    // RGB::new(r, b, g) — calc/src/lib.rs:129-131: the second parameter is BLUE
    static constexpr RGB make(uint8_t r, uint8_t b, uint8_t g) { return RGB{r, g, b}; }
    bool operator==(const RGB &o) const { return r == o.r && g == o.g && b == o.b; }
};
static_assert(sizeof(RGB) == 3 && sizeof(Imaginary) == 16, "must match the C ABI");

// calc::Config (calc/src/lib.rs:21-37); the C struct IS the representation
struct Config : fr_config {
    // Config::new(algo) — calc/src/lib.rs:39-69
    static Config make(Algo algo = Algo::Mandelbrot) {
        Config c;
        fr_config_new(&c, static_cast<uint32_t>(algo));
        return c;
    }
};

// calc::recursive(iterations, start, c, limit) -> (Imaginary, u32) — calc/src/lib.rs:245-257
inline std::pair<Imaginary, uint32_t> recursive(uint32_t iterations, Imaginary start, Imaginary c, double limit) {
    fr_imaginary pos{};
    uint32_t iters = 0;
    check(fr_recursive(iterations, start.c(), c.c(), limit, &pos, &iters));
    return {Imaginary{pos.re, pos.im}, iters};
}

// calc::get_recursive_pixel(&Config, x, y) -> RGB — calc/src/lib.rs:199-235
inline RGB get_recursive_pixel(const Config &config, uint32_t x, uint32_t y) {
    fr_rgb out{};
    check(fr_pixel(&config, x, y, &out));
    return RGB{out.r, out.g, out.b};
}

// Spread get_image over several GPUs (the reference spreads its rows over every core, src/lib.rs:256-258):
// HIP device indices; an index may repeat.  One device (or never calling this) = the single-GPU path.
inline int &multi_devices() {
    static int n = 0;
    return n;
}
inline void use_devices(const std::vector<int> &devices) {
    check(fr_init_devices(devices.data(), static_cast<int>(devices.size())));
    multi_devices() = static_cast<int>(devices.size());
}

// get_image(&Config) -> Vec<RGB> — src/lib.rs:253-270 (Mandelbrot | Julia arm)
inline std::vector<RGB> get_image(const Config &config, int precision = FR_PRECISION_F64) {
    std::vector<RGB> image(static_cast<size_t>(config.width) * config.height);
    uint8_t *out = reinterpret_cast<uint8_t *>(image.data());
    if (multi_devices() > 1)
        check(fr_render_rgb8_multi(&config, precision, 0, out, image.size() * sizeof(RGB)));
    else
        check(fr_render_rows_rgb8(&config, precision, 0, config.height, out, image.size() * sizeof(RGB)));
    return image;
}

// get_image's Algo::BarnsleyFern arm — src/lib.rs:271-319 + fern() :417-463.  threads = what
// rayon::current_num_threads() is on the machine being stood in for (the reference returns ONE thread's image of
// iterations / threads points); seed replaces SmallRng::from_entropy() (src/lib.rs:428).
inline std::vector<RGB> get_image_fern(const Config &config, uint32_t threads, uint64_t seed) {
    std::vector<RGB> image(static_cast<size_t>(config.width) * config.height);
    check(fr_render_fern_rgb8(&config, threads, seed, 0, reinterpret_cast<uint8_t *>(image.data()), image.size() * sizeof(RGB)));
    return image;
}

}  // namespace fractal
#endif
