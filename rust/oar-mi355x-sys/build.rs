//! Locates libOarMi355x.so.  The library is built by `python -m oar_ocr_amd.build` (hipcc, gfx950) into
//! `oar_ocr_amd/lib/`; point OAR_MI355X_LIB_DIR at that directory (or at wherever the .so was installed).
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=OAR_MI355X_LIB_DIR");
    let dir = env::var_os("OAR_MI355X_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        // default: this crate lives in <repo>/rust/oar-mi355x-sys
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").expect("CARGO_MANIFEST_DIR"))
            .join("..")
            .join("..")
            .join("oar_ocr_amd")
            .join("lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=OarMi355x");
    // so that `cargo test` / `cargo run` find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:lib_dir={}", dir.display());
}
