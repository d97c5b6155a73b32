"""zolt_amd — MI355X (gfx950) backend for Zolt's data-parallel prover inner loops.

The product is `libzolt_gpu.so` (hand-written HIP kernels behind the C ABI in
include/zolt_gpu.h). This package is the thin Python host layer used by the tests and
bench.py: `zolt_amd.lib` binds the C ABI with ctypes, `zolt_amd.api` mirrors the reference's
module API (`MSM.compute`, `HyperKZG.commit`, `EqPolynomial.evals`, `DensePolynomial`,
`Sumcheck.Prover`, `run_sumcheck`). There is no CPU fallback: importing `zolt_amd.lib`
without the built library raises, and every compute call fails without a GPU.
"""
__all__ = ["lib", "api"]
