// Links against libce_metrics_hip.so.  CE_METRICS_HIP_DIR points at the directory that holds it (the
// `codec-eval_amd/` directory of the backend's checkout after `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    println!("cargo:rerun-if-env-changed=CE_METRICS_HIP_DIR");
    if let Ok(dir) = std::env::var("CE_METRICS_HIP_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=ce_metrics_hip");
}
