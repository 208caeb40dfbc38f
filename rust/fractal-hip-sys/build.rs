// Points rustc at libfractal_hip.so (built by `python -c 'import __graft_entry__ as g; g.build()'`
// or fractal-renderer_amd/build.py).  FRACTAL_HIP_LIB_DIR = directory holding the .so.
fn main() {
    let dir = std::env::var("FRACTAL_HIP_LIB_DIR")
        .expect("set FRACTAL_HIP_LIB_DIR to the directory containing libfractal_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=fractal_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=FRACTAL_HIP_LIB_DIR");
}
