// NOT RUN where it was written (no Rust toolchain in that image).
//
// Builds libqe_gpu.so with hipcc for gfx950 from the sources of this repository and links the crate against it.
//   QE_GPU_SRC      root of the repository holding include/qe_gpu.h and query-engine_amd/csrc (default: two levels up)
//   QE_GPU_LIB_DIR  use a library already built there instead of compiling (what `python query-engine_amd/build.py` leaves
//                   in query-engine_amd/)
use std::env;
use std::path::{Path, PathBuf};
use std::process::Command;

const SOURCES: &[&str] = &[
    "qe_kernels.hip", "qe_static.hip", "qe_parquet_kernels.hip", "qe_core.cpp", "qe_exec.cpp", "qe_jit.cpp",
    "qe_logical.cpp", "qe_parquet.cpp", "qe_comm.cpp",
];
// -ffp-contract=off: `a + b*c` rounds twice, like the reference's separate mul and add passes over arrow2 arrays.
// -munsafe-fp-atomics: native ds_add_f64 / global_atomic_add_f64.
const FLAGS: &[&str] = &[
    "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-x", "hip",
];

fn main() {
    println!("cargo:rerun-if-env-changed=QE_GPU_LIB_DIR");
    println!("cargo:rerun-if-env-changed=QE_GPU_SRC");
    if let Ok(dir) = env::var("QE_GPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=qe_gpu");
        return;
    }
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let root = env::var("QE_GPU_SRC").map(PathBuf::from).unwrap_or_else(|_| manifest.join("../.."));
    let csrc = root.join("query-engine_amd/csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    // qe_embedded.h (the device headers as strings, for the run-time compiler) is written by the Python build script
    let status = Command::new("python3").arg(root.join("query-engine_amd/build.py")).arg("--embedded-only").status();
    assert!(matches!(status, Ok(s) if s.success()), "writing qe_embedded.h failed");
    let mut objects = Vec::new();
    for src in SOURCES {
        let obj = out.join(Path::new(src).with_extension("o"));
        println!("cargo:rerun-if-changed={}", csrc.join(src).display());
        let ok = Command::new(&hipcc).args(FLAGS).arg("-c").arg(csrc.join(src)).arg("-o").arg(&obj).status();
        assert!(matches!(ok, Ok(s) if s.success()), "hipcc failed on {src}");
        objects.push(obj);
    }
    let lib = out.join("libqe_gpu.so");
    let ok = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-shared", "-fPIC", "-o"])
        .arg(&lib)
        .args(&objects)
        .args(["-lhiprtc", "-lz", "-ldl"])
        .status();
    assert!(matches!(ok, Ok(s) if s.success()), "linking libqe_gpu.so failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=qe_gpu");
    println!("cargo:rerun-if-changed={}", root.join("include/qe_gpu.h").display());
}
