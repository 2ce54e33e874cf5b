module github.com/pbs-plus/pbs-gpu-chunker/go

go 1.22

// Source-only in this repository (no Go toolchain in the build image). tools/golden additionally needs
// github.com/pbs-plus/pxar v0.34.0 (the version pinned by pbs-plus's go.mod:30).
