// Builds libtriplets_amd.so from ../csrc with hipcc (through the Makefile there) and tells cargo to
// link it.  With TRIPLETS_AMD_LIB_DIR set, or the `prebuilt` feature, an existing build is used.
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let csrc = manifest.join("..").join("csrc");
    let lib_dir = match env::var("TRIPLETS_AMD_LIB_DIR") {
        Ok(dir) => PathBuf::from(dir),
        Err(_) => {
            if env::var("CARGO_FEATURE_PREBUILT").is_err() {
                let status = Command::new("make")
                    .arg("-C")
                    .arg(&csrc)
                    .arg("libtriplets_amd.so")
                    .status()
                    .expect("failed to run make (is hipcc on PATH?)");
                assert!(status.success(), "building libtriplets_amd.so failed");
            }
            csrc.clone()
        }
    };
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=triplets_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rerun-if-changed={}", csrc.join("trp_kernels.hip").display());
    println!("cargo:rerun-if-changed={}", csrc.join("trp_lane.h").display());
    println!("cargo:rerun-if-changed={}", manifest.join("../../include/triplets_amd.h").display());
    println!("cargo:rerun-if-env-changed=TRIPLETS_AMD_LIB_DIR");
}
