//! `extern "C"` bindings to libfractal_hip.so (include/fractal_hip.h) plus the safe wrappers the
//! reference's `get_image` / `get_recursive_pixel` / `recursive` call sites need.
//!
//! The `#[repr(C)]` types are the C ABI's image of `calc::Config`, `calc::Imaginary` and
//! `calc::RGB` (calc/src/lib.rs:21-37, 79-82, 121-125).  `calc::RGB` is `repr(Rust)` today although
//! the reference already transmutes it to 3 packed bytes (src/lib.rs:13-15, src/gui.rs:62-64);
//! adding `#[repr(C)]` to it makes that assumption sound and lets `Vec<RGB>` be filled in place.
//!
//! Untested in this pipeline (no rustc); kept mechanical on purpose.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct fr_imaginary {
    pub re: f64,
    pub im: f64,
}

/// The STORED fields of `calc::RGB` (not `RGB::new`'s `(r, b, g)` parameter order).
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct fr_rgb {
    pub r: u8,
    pub g: u8,
    pub b: u8,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct fr_config {
    pub algo: u32, // 0 Mandelbrot, 1 BarnsleyFern, 2 Julia (calc/src/lib.rs:150-154)
    pub width: u32,
    pub height: u32,
    pub iterations: u32,
    pub limit: f64,
    pub stable_limit: f64,
    pub pos: fr_imaginary,
    pub scale: fr_imaginary,
    pub exposure: f64,
    pub inside: u8,
    pub smooth: u8,
    pub primary_color: fr_rgb,
    pub secondary_color: fr_rgb,
    pub color_weight: f64,
    pub julia_set: fr_imaginary,
}

pub const FR_OK: c_int = 0;
/// `FR_ABI_VERSION` of the header these declarations were written against; `check_abi()` compares it with the library's.
pub const FR_ABI_VERSION: c_int = 3;
pub const FR_PRECISION_F64: c_int = 0;
pub const FR_PRECISION_F32: c_int = 1;

extern "C" {
    pub fn fr_init(device: c_int) -> c_int;
    pub fn fr_shutdown() -> c_int;
    pub fn fr_device_count(count: *mut c_int) -> c_int;
    pub fn fr_last_error() -> *const c_char;
    pub fn fr_abi_version() -> c_int;
    pub fn fr_build_id() -> *const c_char;
    pub fn fr_device_name(buf: *mut c_char, buf_len: usize) -> c_int;
    pub fn fr_config_new(cfg: *mut fr_config, algo: u32);
    pub fn fr_render_rgb8(cfg: *const fr_config, out: *mut u8, out_len: usize) -> c_int;
    pub fn fr_render_rows_rgb8(cfg: *const fr_config, precision: c_int, y0: u32, y1: u32, out: *mut u8, out_len: usize) -> c_int;
    pub fn fr_render_rows_rgb8_device(cfg: *const fr_config, precision: c_int, y0: u32, y1: u32, d_out: *mut c_void, out_len: usize, hip_stream: *mut c_void) -> c_int;
    pub fn fr_render_rows_rgba8(cfg: *const fr_config, precision: c_int, y0: u32, y1: u32, out: *mut u8, out_len: usize) -> c_int;
    pub fn fr_render_rows_rgba8_device(cfg: *const fr_config, precision: c_int, y0: u32, y1: u32, d_out: *mut c_void, out_len: usize, hip_stream: *mut c_void) -> c_int;
    pub fn fr_pixel(cfg: *const fr_config, x: u32, y: u32, out: *mut fr_rgb) -> c_int;
    pub fn fr_pixel_p(cfg: *const fr_config, precision: c_int, x: u32, y: u32, out: *mut fr_rgb) -> c_int;
    pub fn fr_recursive(iterations: u32, start: fr_imaginary, c: fr_imaginary, limit: f64, out_pos: *mut fr_imaginary, out_iters: *mut u32) -> c_int;
    pub fn fr_recursive_batch(iterations: u32, start: *const fr_imaginary, c: *const fr_imaginary, n: usize, limit: f64, precision: c_int, out_pos: *mut fr_imaginary, out_iters: *mut u32) -> c_int;
    pub fn fr_escape_rows(cfg: *const fr_config, precision: c_int, y0: u32, y1: u32, z_re_im: *mut f64, iters: *mut u32) -> c_int;
    // the colour map alone over stored recursive() results: what the GUI's exposure / colour controls need (src/gui.rs:183-203)
    pub fn fr_colour_rgb8(cfg: *const fr_config, z_re_im: *const f64, iters: *const u32, n: usize, out: *mut u8, out_len: usize) -> c_int;
    // one process, several GPUs (include/fractal_hip.h, "get_image across several GPUs from ONE process")
    pub fn fr_init_devices(devices: *const c_int, n: c_int) -> c_int;
    pub fn fr_render_rgb8_multi(cfg: *const fr_config, precision: c_int, block_rows: u32, out: *mut u8, out_len: usize) -> c_int;
    // a frame buffer that is rendered into again and again: pin it once; unpin before freeing it
    pub fn fr_pin_host_buffer(ptr: *mut c_void, len: usize) -> c_int;
    pub fn fr_unpin_host_buffer(ptr: *mut c_void) -> c_int;
    // the default dispatch's view sample (blocking only in front of the first launch of a view of 4096 x 2048 pixels and more): 0 = off
    pub fn fr_set_dispatch_sampling(enabled: c_int) -> c_int;
    // Algo::BarnsleyFern (src/lib.rs:271-319, 417-463) on the GPU
    pub fn fr_render_fern_rgb8(cfg: *const fr_config, threads: u32, seed: u64, walkers: u32, out: *mut u8, out_len: usize) -> c_int;
}

/// Message of the last failing call on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = fr_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

/// Select the GPUs `render_into` spreads an image over (row-block-cyclically, one host thread and one
/// PCIe link per GPU).  Not calling it, or passing one device, renders on a single GPU.
pub fn use_devices(devices: &[c_int]) -> Result<(), String> {
    let rc = unsafe { fr_init_devices(devices.as_ptr(), devices.len() as c_int) };
    if rc != FR_OK { Err(last_error()) } else { MULTI.store(devices.len() > 1, std::sync::atomic::Ordering::Relaxed); Ok(()) }
}
static MULTI: std::sync::atomic::AtomicBool = std::sync::atomic::AtomicBool::new(false);

/// `get_image` for `Algo::Mandelbrot | Algo::Julia` (src/lib.rs:253-270): fills a caller-owned
/// pixel vector in place.  `P` must be a 3-byte `#[repr(C)]` pixel (`calc::RGB` once annotated).
pub fn render_into<P: Copy>(cfg: &fr_config, image: &mut Vec<P>) -> Result<(), String> {
    assert_eq!(std::mem::size_of::<P>(), 3, "pixel type must be 3 packed bytes");
    let n = cfg.width as usize * cfg.height as usize;
    image.clear();
    image.reserve_exact(n);
    let out = image.as_mut_ptr() as *mut u8;
    let rc = if MULTI.load(std::sync::atomic::Ordering::Relaxed) {
        unsafe { fr_render_rgb8_multi(cfg, FR_PRECISION_F64, 0, out, n * 3) }
    } else {
        unsafe { fr_render_rgb8(cfg, out, n * 3) }
    };
    if rc != FR_OK {
        return Err(last_error());
    }
    unsafe { image.set_len(n) }; // every byte was written by the library
    Ok(())
}

/// `get_image`'s `Algo::BarnsleyFern` arm (src/lib.rs:271-319): `threads` is what
/// `rayon::current_num_threads()` returns on this machine (the reference's result is ONE thread's image of
/// `iterations / threads` points), `seed` replaces `SmallRng::from_entropy()` (src/lib.rs:428).
pub fn fern_into<P: Copy>(cfg: &fr_config, threads: u32, seed: u64, image: &mut Vec<P>) -> Result<(), String> {
    assert_eq!(std::mem::size_of::<P>(), 3, "pixel type must be 3 packed bytes");
    let n = cfg.width as usize * cfg.height as usize;
    image.clear();
    image.reserve_exact(n);
    let rc = unsafe { fr_render_fern_rgb8(cfg, threads, seed, 0, image.as_mut_ptr() as *mut u8, n * 3) };
    if rc != FR_OK {
        return Err(last_error());
    }
    unsafe { image.set_len(n) };
    Ok(())
}

/// `calc::get_recursive_pixel(&Config, x, y) -> RGB` (calc/src/lib.rs:199-235) on the device: one pixel per call — the
/// reference's per-pixel API kept for callers that use it; images go through `render_into`.
pub fn get_recursive_pixel(cfg: &fr_config, x: u32, y: u32) -> Result<fr_rgb, String> {
    let mut out = fr_rgb { r: 0, g: 0, b: 0 };
    let rc = unsafe { fr_pixel(cfg, x, y, &mut out) };
    if rc != FR_OK { Err(last_error()) } else { Ok(out) }
}

/// `calc::recursive(iterations, start, c, limit) -> (Imaginary, u32)` (calc/src/lib.rs:245-257): the final position
/// and the escape index (== `iterations` when the cap is exhausted), bit for bit.
pub fn recursive(iterations: u32, start: fr_imaginary, c: fr_imaginary, limit: f64) -> Result<(fr_imaginary, u32), String> {
    let mut pos = fr_imaginary { re: 0.0, im: 0.0 };
    let mut iters = 0u32;
    let rc = unsafe { fr_recursive(iterations, start, c, limit, &mut pos, &mut iters) };
    if rc != FR_OK { Err(last_error()) } else { Ok((pos, iters)) }
}

/// `recursive()` of every pixel of rows `[y0, y1)`: `(z, iters)` with `z[2k], z[2k+1]` the final position and
/// `iters[k]` the escape index of pixel `k = (y - y0) * width + x` — the raw results a GUI keeps to re-colour
/// (`colour_into`) when only exposure / colours / smooth / inside change (src/gui.rs:183-203).
pub fn escape_rows(cfg: &fr_config, y0: u32, y1: u32) -> Result<(Vec<f64>, Vec<u32>), String> {
    let n = cfg.width as usize * (y1.saturating_sub(y0)) as usize;
    let mut z: Vec<f64> = Vec::with_capacity(2 * n);
    let mut iters: Vec<u32> = Vec::with_capacity(n);
    let rc = unsafe { fr_escape_rows(cfg, FR_PRECISION_F64, y0, y1, z.as_mut_ptr(), iters.as_mut_ptr()) };
    if rc != FR_OK {
        return Err(last_error());
    }
    unsafe { z.set_len(2 * n); iters.set_len(n); } // every element was written by the library
    Ok((z, iters))
}

/// The colour map alone (calc/src/lib.rs:214-234 + `color_multiply`) over stored `recursive()` results into a pixel
/// vector: re-colouring without re-iterating.
pub fn colour_into<P: Copy>(cfg: &fr_config, z: &[f64], iters: &[u32], image: &mut Vec<P>) -> Result<(), String> {
    assert_eq!(std::mem::size_of::<P>(), 3, "pixel type must be 3 packed bytes");
    assert_eq!(z.len(), 2 * iters.len());
    let n = iters.len();
    image.clear();
    image.reserve_exact(n);
    let rc = unsafe { fr_colour_rgb8(cfg, z.as_ptr(), iters.as_ptr(), n, image.as_mut_ptr() as *mut u8, n * 3) };
    if rc != FR_OK {
        return Err(last_error());
    }
    unsafe { image.set_len(n) };
    Ok(())
}

/// The whole frame as RGBA8 (r, g, b, 255) into a caller-owned byte buffer of `4 * width * height` bytes: what the GUI
/// builds on the CPU today (`to_rgba8`, src/gui.rs:71-72) produced on the device.  Keep the buffer across frames (and
/// `fr_pin_host_buffer` it once) — see rust/gui.patch.rs.
pub fn render_rgba_into(cfg: &fr_config, frame: &mut [u8]) -> Result<(), String> {
    let need = 4 * cfg.width as usize * cfg.height as usize;
    if frame.len() < need {
        return Err(format!("frame buffer holds {} bytes, the frame needs {}", frame.len(), need));
    }
    let rc = unsafe { fr_render_rows_rgba8(cfg, FR_PRECISION_F64, 0, cfg.height, frame.as_mut_ptr(), frame.len()) };
    if rc != FR_OK { Err(last_error()) } else { Ok(()) }
}

/// The loaded library must speak the ABI these declarations describe (include/fractal_hip.h: FR_ABI_VERSION).
pub fn check_abi() -> Result<(), String> {
    let v = unsafe { fr_abi_version() };
    if v == FR_ABI_VERSION { Ok(()) } else { Err(format!("libfractal_hip speaks ABI {}, this crate {}", v, FR_ABI_VERSION)) }
}
