// What a maintainer would change in the reference (illustrative; not compiled here).
//
// calc/src/lib.rs:121 — make the layout the rest of the code already assumes explicit:
//     #[repr(C)]
//     pub struct RGB { pub r: u8, pub g: u8, pub b: u8 }
//
// src/lib.rs:253-270 — the Mandelbrot | Julia arm of get_image becomes one FFI call; the
// BarnsleyFern arm (src/lib.rs:271-319) either stays as it is or becomes one too (`fern_into`: same
// distribution, a seeded RNG instead of SmallRng::from_entropy).
//
// src/main.rs — once, before the first render, to spread images over every GPU of the node:
//     fractal_hip_sys::use_devices(&[0, 1, 2, 3, 4, 5, 6, 7]).expect("fractal_hip");

fn to_ffi(config: &Config) -> fractal_hip_sys::fr_config {
    use fractal_hip_sys::{fr_config, fr_imaginary, fr_rgb};
    let im = |v: &Imaginary| fr_imaginary { re: v.re, im: v.im };
    let rgb = |c: &RGB| fr_rgb { r: c.r, g: c.g, b: c.b }; // stored fields, verbatim
    fr_config {
        algo: match config.algo { Algo::Mandelbrot => 0, Algo::BarnsleyFern => 1, Algo::Julia => 2 },
        width: config.width,
        height: config.height,
        iterations: config.iterations,
        limit: config.limit,
        stable_limit: config.stable_limit,
        pos: im(&config.pos),
        scale: im(&config.scale),
        exposure: config.exposure,
        inside: config.inside as u8,
        smooth: config.smooth as u8,
        primary_color: rgb(&config.primary_color),
        secondary_color: rgb(&config.secondary_color),
        color_weight: config.color_weight,
        julia_set: im(&config.julia_set),
    }
}

pub fn get_image(config: &Config) -> Vec<RGB> {
    match config.algo {
        Algo::Mandelbrot | Algo::Julia => {
            let mut image: Vec<RGB> = Vec::new();
            // get_image is infallible in the reference; a GPU failure is a panic here (or call the
            // old rayon arm instead — the maintainer's choice).
            fractal_hip_sys::render_into(&to_ffi(config), &mut image)
                .unwrap_or_else(|e| panic!("fractal_hip: {}", e));
            image
        }
        Algo::BarnsleyFern => {
            // either keep the body of src/lib.rs:271-319 here unchanged, or:
            let mut image: Vec<RGB> = Vec::new();
            let seed = rand::random::<u64>(); // the reference seeds from entropy too (src/lib.rs:428)
            fractal_hip_sys::fern_into(&to_ffi(config), rayon::current_num_threads() as u32, seed, &mut image)
                .unwrap_or_else(|e| panic!("fractal_hip: {}", e));
            image
        }
    }
}
