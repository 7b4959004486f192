"""spmv-samples_amd — MI355X-native CSR SpMV engine behind the operator interface of
peakcrosser7/spmv-samples (`SpMV<...>(kind_str, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y)`).

The directory name carries a hyphen (it mirrors the reference's repository name), so
it is imported under the alias `spmv_samples_amd`:

    import __graft_entry__          # registers the alias
    import spmv_samples_amd as sp

Contents (only what the hot path needs):
    csrc/      HIP kernels (gfx950) + the C ABI of include/mi355_spmv.h
    lib/       built libmi355spmv.so (git-ignored)
    host/      C++ mirror of the reference's include/spmv.h boundary (SpMV<>, SPMV_KINDS, Timer)
    capi.py    ctypes binding used by tests and bench.py
    synth.py   seeded synthetic CSR matrices (stand-ins for the BASELINE configs)
    dist.py    row-block sharding + allgatherv(y) for one process per GPU
    load.py    ctypes binding of the Matrix Market loader (include/mi355_load.h, host/load.hpp)
"""
from . import capi, dist, load, synth  # noqa: F401
from .capi import DistPlan, Functor, Plan, PlanShape, spmv, spmv_genl, spmv_mixed  # noqa: F401
