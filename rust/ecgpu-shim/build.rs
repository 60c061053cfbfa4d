// Links libecgpu.so (built by `python -m ethereum_consensus_amd.build` into ethereum_consensus_amd/lib/).
fn main() {
    let dir = std::env::var("ECGPU_LIB_DIR").unwrap_or_else(|_| "../../ethereum_consensus_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=ecgpu");
    println!("cargo:rerun-if-env-changed=ECGPU_LIB_DIR");
}
