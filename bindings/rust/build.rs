// Links libqoaudio_amd.so.  Set QOAUDIO_AMD_LIB_DIR to the directory that holds it
// (the repository's qoaudio_amd/ after `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    let dir = std::env::var("QOAUDIO_AMD_LIB_DIR").unwrap_or_else(|_| "../../qoaudio_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=qoaudio_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=QOAUDIO_AMD_LIB_DIR");
}
