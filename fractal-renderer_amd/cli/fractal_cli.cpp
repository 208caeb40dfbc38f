// fractal_cli — command line over libfractal_hip.so with the reference's flags and defaults
// (clap builder at src/lib.rs:32-164, extraction at :168-226), so a user of the reference binary can
// run the same command lines on an MI355X.  It writes a binary PPM (P6) of the Vec<RGB> bytes: the
// AVIF encoder (src/lib.rs:323-367) is out of scope and stays the reference's.
//
//   g++ -std=c++17 -Iinclude -Ifractal-renderer_amd/host fractal-renderer_amd/cli/fractal_cli.cpp
//       -Lfractal-renderer_amd -lfractal_hip -Wl,-rpath,$PWD/fractal-renderer_amd -o fractal_cli
//   ./fractal_cli 3000 3000 -i 1024 -s 1000000 -x -0.7436447860 -y 0.1318252536 -o zoom
//
// Extensions (no counterpart upstream): --f32; --devices 0,1,... (spread the image over several GPUs);
// for -a fern: --threads N (the rayon thread count being stood in for; default: this machine's hardware
// threads, what rayon would use) and --seed N (default: from the OS, as the reference seeds from entropy).
// Not handled here (by design): --gui, --open.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <optional>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "fractal.hpp"

namespace {

[[noreturn]] void die(const std::string &msg) {
    std::fprintf(stderr, "error: %s\n", msg.c_str());
    std::exit(2);
}

// parse_hex_rgb (src/lib.rs:22-29): "RRGGBB" -> RGB::new(r, g, b); RGB::new's parameters are
// (r, b, g) (calc/src/lib.rs:129-131), so that call stores {r: r, g: b, b: g}.
fractal::RGB parse_hex_rgb(const std::string &s) {
    if (s.size() != 6) die("failed to parse hex color");
    auto byte = [&](int i) {
        char *end = nullptr;
        const std::string part = s.substr(i, 2);
        long v = std::strtol(part.c_str(), &end, 16);
        if (*end != '\0') die("failed to parse hex color");
        return static_cast<uint8_t>(v);
    };
    return fractal::RGB::make(byte(0), byte(2), byte(4));
}

double to_f64(const std::string &s, const char *what) {
    char *end = nullptr;
    double v = std::strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') die(std::string("invalid value for ") + what + ": " + s);
    return v;
}

uint32_t to_u32(const std::string &s, const char *what) {
    char *end = nullptr;
    unsigned long long v = std::strtoull(s.c_str(), &end, 10);
    if (end == s.c_str() || *end != '\0' || v > 0xFFFFFFFFull) die(std::string("invalid value for ") + what + ": " + s);
    return static_cast<uint32_t>(v);
}

}  // namespace

int main(int argc, char **argv) {
    using namespace fractal;
    // defaults: src/lib.rs:34-164
    std::string width = "750", height = "500", limit = "65536", stable_limit = "2", pos_y = "0", scale = "0.4",
                exposure = "5", filename = "output", algo_s = "mandelbrot", color_weight = "0.01";
    std::optional<std::string> iterations, pos_x, scale_x, scale_y, primary, secondary, julia_re, julia_im, devices, threads_s,
        seed_s;
    bool disable_inside = false, unsmooth = false, f32 = false, quiet = false;
    std::vector<std::string> positionals;

    auto value = [&](int &i, const char *flag) -> std::string {
        if (i + 1 >= argc) die(std::string("missing value for ") + flag);
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-i" || a == "--iterations") iterations = value(i, "-i");
        else if (a == "-l" || a == "--limit") limit = value(i, "-l");
        else if (a == "--stable-limit") stable_limit = value(i, "--stable-limit");
        else if (a == "-x") pos_x = value(i, "-x");
        else if (a == "-y") pos_y = value(i, "-y");
        else if (a == "--scale-x") scale_x = value(i, "--scale-x");
        else if (a == "--scale-y") scale_y = value(i, "--scale-y");
        else if (a == "-s" || a == "--scale") scale = value(i, "-s");
        else if (a == "-e" || a == "--exposure") exposure = value(i, "-e");
        else if (a == "--primary-color") primary = value(i, "--primary-color");
        else if (a == "--secondary-color") secondary = value(i, "--secondary-color");
        else if (a == "-d" || a == "--disable-inside") disable_inside = true;
        else if (a == "-u" || a == "--unsmooth") unsmooth = true;
        else if (a == "-o" || a == "--output") filename = value(i, "-o");
        else if (a == "-a" || a == "--algorithm") algo_s = value(i, "-a");
        else if (a == "--julia-real") julia_re = value(i, "--julia-real");
        else if (a == "--julia-imaginary") julia_im = value(i, "--julia-imaginary");
        else if (a == "-w" || a == "--color-weight") color_weight = value(i, "-w");
        else if (a == "--f32") f32 = true;       // this build's extension (no counterpart upstream)
        else if (a == "--devices") devices = value(i, "--devices");
        else if (a == "--threads") threads_s = value(i, "--threads");
        else if (a == "--seed") seed_s = value(i, "--seed");
        else if (a == "--quiet") quiet = true;
        else if (a == "--open" || a == "-g" || a == "--gui") die(a + " is not supported by this front end");
        else if (a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9') && a[1] != '.') die("unknown flag " + a);
        else positionals.push_back(a);
    }
    if (positionals.size() > 2) die("too many positional arguments (expected <width> <height>)");
    if (!positionals.empty()) width = positionals[0];
    if (positionals.size() > 1) height = positionals[1];

    // Algo::from_str (calc/src/lib.rs:165-179): case-insensitive
    std::string al = algo_s;
    for (char &c : al) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    Algo algo;
    if (al == "mandelbrot") algo = Algo::Mandelbrot;
    else if (al == "julia") algo = Algo::Julia;
    else if (al == "fern" || al == "barnsleyfern") algo = Algo::BarnsleyFern;
    else die("invalid algorithm name");
    if (algo == Algo::Julia && (!julia_re || !julia_im)) die("--julia-real and --julia-imaginary are required with -a julia");
    if ((scale_x || scale_y) && scale != "0.4") die("--scale conflicts with --scale-x/--scale-y");

    // src/lib.rs:207-226
    Config cfg = Config::make(algo);
    cfg.width = to_u32(width, "width");
    cfg.height = to_u32(height, "height");
    if (iterations) cfg.iterations = to_u32(*iterations, "iterations");
    cfg.limit = to_f64(limit, "limit");
    cfg.stable_limit = to_f64(stable_limit, "stable-limit");
    cfg.pos.re = to_f64(pos_x ? *pos_x : (algo == Algo::Julia ? "0" : "-0.6"), "-x"); /* src/lib.rs:66-72: 0 only for julia, so -0.6 for the fern too */
    cfg.pos.im = to_f64(pos_y, "-y");
    cfg.scale.re = to_f64(scale_x ? *scale_x : scale, "scale");
    cfg.scale.im = to_f64(scale_y ? *scale_y : scale, "scale");
    cfg.exposure = to_f64(exposure, "exposure");
    cfg.inside = !disable_inside;
    cfg.smooth = !unsmooth;
    if (primary) {
        RGB c = parse_hex_rgb(*primary);
        cfg.primary_color = fr_rgb{c.r, c.g, c.b};
    }
    if (secondary) {
        RGB c = parse_hex_rgb(*secondary);
        cfg.secondary_color = fr_rgb{c.r, c.g, c.b};
    }
    cfg.color_weight = to_f64(color_weight, "color-weight");
    if (algo == Algo::Julia) {
        cfg.julia_set.re = to_f64(*julia_re, "--julia-real");
        cfg.julia_set.im = to_f64(*julia_im, "--julia-imaginary");
    }

    try {
        if (devices) {
            std::vector<int> list;
            size_t pos = 0;
            while (pos <= devices->size()) {
                const size_t comma = devices->find(',', pos);
                const std::string part = devices->substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                list.push_back(static_cast<int>(to_u32(part, "--devices")));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
            use_devices(list);
        }
        std::vector<RGB> image;
        const auto t0 = std::chrono::steady_clock::now();
        if (algo == Algo::BarnsleyFern) {
            uint32_t threads = threads_s ? to_u32(*threads_s, "--threads") : std::thread::hardware_concurrency();
            if (threads == 0) threads = 1;
            uint64_t seed = 0;
            if (seed_s) {
                char *end = nullptr;
                seed = std::strtoull(seed_s->c_str(), &end, 10);
                if (end == seed_s->c_str() || *end != '\0') die("invalid value for --seed: " + *seed_s);
            } else {
                std::random_device rd;
                seed = (static_cast<uint64_t>(rd()) << 32) | rd();
            }
            image = get_image_fern(cfg, threads, seed);
        } else {
            image = get_image(cfg, f32 ? FR_PRECISION_F32 : FR_PRECISION_F64);
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const std::string path = filename + ".ppm";  // the reference appends ".avif" (src/lib.rs:192-195)
        std::FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) die("cannot open " + path);
        std::fprintf(f, "P6\n%u %u\n255\n", cfg.width, cfg.height);
        std::fwrite(image.data(), 3, image.size(), f);
        std::fclose(f);
        if (!quiet) std::printf("rendered %ux%u in %.2f ms -> %s\n", cfg.width, cfg.height, ms, path.c_str());
    } catch (const Error &e) {
        std::fprintf(stderr, "fractal_hip error %d: %s\n", e.code(), e.what());
        return 1;
    }
    return 0;
}
