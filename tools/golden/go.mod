// Stand-alone module of the golden-vector harness: it needs nothing of this repository, only the REAL chunker.
// (pbs-plus itself pins the same version: /root/reference go.mod:30.)
module pbsgpu-golden

go 1.22

require github.com/pbs-plus/pxar v0.34.0
