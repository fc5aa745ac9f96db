// UNVERIFIED SOURCE (never compiled here).  Points the linker at the in-tree build of the library:
// BEVYRAY_AMD_LIB_DIR=/path/to/repo/bevyray_amd  (libbevyray_amd.so; it finds libamdhip64.so.7 through its RUNPATH).
fn main() {
    let dir = std::env::var("BEVYRAY_AMD_LIB_DIR").unwrap_or_else(|_| "../../bevyray_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=bevyray_amd");
    println!("cargo:rerun-if-env-changed=BEVYRAY_AMD_LIB_DIR");
}
