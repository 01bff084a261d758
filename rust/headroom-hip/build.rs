// build.rs -- builds (or finds) libheadroom_hip.so and tells cargo to link it.
// UNCOMPILED HERE (no Rust toolchain in the build image); the hipcc command lines are those of headroom_amd/csrc/Makefile.
use std::env;
use std::path::{Path, PathBuf};
use std::process::Command;

const HIP_SRCS: &[&str] = &["hr_api.hip", "hr_kernels_generic.hip", "hr_kernels_fused.hip", "hr_synth.hip", "hr_gain.hip"];
const CPP_SRCS: &[&str] = &["hr_tables.cpp", "hr_analyzer.cpp", "hr_ingest.cpp", "hr_gainfile.cpp"];

fn run(cmd: &mut Command) {
    let status = cmd.status().unwrap_or_else(|e| panic!("failed to start {:?}: {e}", cmd));
    assert!(status.success(), "{:?} failed with {status}", cmd);
}

fn main() {
    println!("cargo:rerun-if-env-changed=HEADROOM_HIP_DIR");
    println!("cargo:rerun-if-env-changed=HIPCC");
    let prebuilt = env::var_os("CARGO_FEATURE_PREBUILT").is_some();
    if let Some(dir) = env::var_os("HEADROOM_HIP_DIR") {
        // an existing libheadroom_hip.so (make -C headroom_amd/csrc)
        println!("cargo:rustc-link-search=native={}", Path::new(&dir).display());
        println!("cargo:rustc-link-lib=dylib=headroom_hip");
        return;
    }
    assert!(!prebuilt, "feature `prebuilt` needs HEADROOM_HIP_DIR to point at the directory of libheadroom_hip.so");

    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let csrc = manifest.join("../../headroom_amd/csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"];
    let mut objs = Vec::new();
    for src in HIP_SRCS {
        println!("cargo:rerun-if-changed={}", csrc.join(src).display());
        let obj = out.join(format!("{src}.o"));
        run(Command::new(&hipcc).arg("--offload-arch=gfx950").args(flags).arg("-c").arg(csrc.join(src)).arg("-o").arg(&obj));
        objs.push(obj);
    }
    for src in CPP_SRCS {
        println!("cargo:rerun-if-changed={}", csrc.join(src).display());
        let obj = out.join(format!("{src}.o"));
        run(Command::new(&hipcc).args(flags).args(["-x", "c++", "-c"]).arg(csrc.join(src)).arg("-o").arg(&obj));
        objs.push(obj);
    }
    for hdr in ["hr_tables.hpp", "hr_device.hpp", "hr_kernels.hpp", "../../include/headroom_hip.h"] {
        println!("cargo:rerun-if-changed={}", csrc.join(hdr).display());
    }
    let lib = out.join("libheadroom_hip.so");
    run(Command::new(&hipcc).args(["--offload-arch=gfx950", "-shared", "-fPIC", "-o"]).arg(&lib).args(&objs));
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=headroom_hip");
    // the HIP runtime the library itself links against
    println!("cargo:rustc-link-search=native=/opt/rocm/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
}
