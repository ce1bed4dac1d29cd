// JXL_HIP_LIB_DIR = directory holding libjxl_hip.so (built by `make -C jxl_rs_amd/csrc`)
fn main() {
    let dir = std::env::var("JXL_HIP_LIB_DIR").unwrap_or_else(|_| "../../../jxl_rs_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=jxl_hip");
    println!("cargo:rerun-if-env-changed=JXL_HIP_LIB_DIR");
}
