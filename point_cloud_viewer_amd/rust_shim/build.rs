// Links against libpcv_hip.so built by `make -C point_cloud_viewer_amd/csrc`.
fn main() {
    let dir = std::env::var("PCV_HIP_LIB_DIR").unwrap_or_else(|_| "../".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=pcv_hip");
}
