fn main() {
    println!("cargo:rustc-link-search=native={}", std::env::var("SYLPH_HIP_LIB_DIR").unwrap());
    println!("cargo:rustc-link-lib=dylib=sylph_hip");      // libsylph_hip.so, built by `make -C sylph_amd/csrc`
}
