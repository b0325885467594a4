// Links the two shared libraries built by `make -C rust-ida_amd/csrc libidahip.so` and `make -C rust-ida_amd/host libidaens.so`.
// IDAHIP_LIB_DIR / IDAENS_LIB_DIR name their directories (default: the in-tree build locations relative to this crate).
use std::env;
use std::path::PathBuf;

fn main() {
    let here = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let root = here.join("..").join("..").join("..");
    let hip = env::var("IDAHIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| root.join("rust-ida_amd").join("csrc"));
    let ens = env::var("IDAENS_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| root.join("rust-ida_amd").join("host"));
    for dir in [&hip, &ens] {
        println!("cargo:rustc-link-search=native={}", dir.display());
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    }
    println!("cargo:rustc-link-lib=dylib=idahip");
    println!("cargo:rustc-link-lib=dylib=idaens");
    println!("cargo:rerun-if-env-changed=IDAHIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=IDAENS_LIB_DIR");
    println!("cargo:rerun-if-changed=../../../include/ida_hip.h");
    println!("cargo:rerun-if-changed=../../../include/ida_ensemble.h");
}
