"""eagleeverything_amd -- MI355X (gfx950) backend for the Eagle/WMAM association-mapping hot path.

The product is libeaglehip.so (HIP kernels + C ABI, include/eagle_hip.h).  This package is the host-side
mirror of the reference interface used by tests, bench.py and the sharded multi-GPU driver:

  rcpp_api   -- ReadBlock, calculateMMt_rcpp, calculate_a_and_vara_rcpp, calculate_reduced_a_rcpp
  r_api      -- calculateMMt, calcMMt, calculate_a_and_vara, find_qtl (the R wrappers' marshalling rules)
  host_model -- dense n x n model algebra that stays on host LAPACK by design
  sharded    -- marker-sharded multi-GPU scan / MM^T (one process per GPU, torch.distributed over RCCL)
  synth      -- seeded synthetic genotypes of the benchmark shapes
"""
__all__ = ["rcpp_api", "r_api", "host_model", "synth"]
