// What a maintainer would change in src/gui.rs:56-82 (illustrative; not compiled here — no Rust toolchain in this image).
//
// Today the render thread calls get_image (a fresh Vec<RGB> per redraw, src/gui.rs:60), transmutes it to bytes (:62-64),
// wraps it in an image::RgbImage (:66-68), converts that to RGBA on the CPU (:70-72: a second full pass over the frame and
// a third allocation) and copies it into an egui::ColorImage (:75).  With the library the frame arrives as RGBA8 from the
// device, into ONE buffer the thread keeps: no transmute, no `image` crate round trip, no to_rgba8.
//
//     std::thread::spawn(move || {
//         let mut frame_rgba: Vec<u8> = Vec::new();           // kept across redraws
//         let mut pinned: (*mut u8, usize) = (std::ptr::null_mut(), 0);
//         while let Ok((config, frame)) = rx.recv() {
//             let need = 4 * config.width as usize * config.height as usize;
//             if frame_rgba.len() != need {
//                 // the window was resized: re-make (and re-pin) the buffer.  Pinning is optional — frames up to
//                 // 3840 x 2160 go through the library's own pinned staging buffer and never touch these pages by DMA —
//                 // but costs nothing for a buffer that lives as long as the window.
//                 if !pinned.0.is_null() { unsafe { fractal_hip_sys::fr_unpin_host_buffer(pinned.0 as *mut _) }; }
//                 frame_rgba = vec![0u8; need];
//                 pinned = (frame_rgba.as_mut_ptr(), need);
//                 unsafe { fractal_hip_sys::fr_pin_host_buffer(pinned.0 as *mut _, need) };
//             }
//             match config.algo {
//                 Algo::Mandelbrot | Algo::Julia => {
//                     fractal_hip_sys::render_rgba_into(&to_ffi(&config), &mut frame_rgba)
//                         .unwrap_or_else(|e| panic!("fractal_hip: {}", e));
//                 }
//                 Algo::BarnsleyFern => { /* the old path: get_image + to_rgba8, src/gui.rs:60-72 */ }
//             }
//             let size = [config.width as usize, config.height as usize];
//             let color_image = egui::ColorImage::from_rgba_unmultiplied(size, &frame_rgba);   // src/gui.rs:75, unchanged
//             { let mut lock = image_handle.lock().unwrap(); *lock = Some(color_image); }
//             working_handle.store(false, std::sync::atomic::Ordering::SeqCst);
//             frame.request_repaint();
//         }
//     });
//
// Exposure / colour / smooth / inside controls (src/gui.rs:183-203) change only inputs of the colour map
// (calc/src/lib.rs:214-234): keep `(z, iters) = fractal_hip_sys::escape_rows(..)` of the current view and call
// `colour_into` on a slider move instead of re-iterating; a view change (pos, scale, iterations, algo, julia_set, size)
// re-iterates.  The screenshot thread (src/gui.rs:322-326) calls get_image concurrently: the library is re-entrant.
