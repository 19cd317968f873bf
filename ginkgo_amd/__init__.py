"""ginkgo_amd - MI355X (gfx950 / CDNA4) backend for Ginkgo's Krylov hot path.

Thin host-side mirror of the gko::Executor / gko::LinOp / solver-factory
interface over the C ABI of libgko_cdna4.so (include/gko_cdna4.h): CSR / ELL /
SELL-P SpMV, block-Jacobi, BLAS-1 and the fused CG steps as hand-written HIP
kernels.  Importing the package does not need a GPU; creating a
Cdna4Executor does, and there is no CPU fallback.
"""
from ._lib import (DimensionMismatch, GkoError, NotCompiled, NotSupported,
                   LIB_PATH)
from .executor import Cdna4Executor
from .matrix import (Coo, Csr, Dense, DeviceMatrixData, Ell, Hybrid, Sellp, entry_dtype, scalar,
                     stencil_csr)
from .preconditioner import Jacobi, compute_storage_scheme
from .solver import Cg, Gmres, Identity, ortho_method
from .krylov import Bicg, Bicgstab, Cgs, Chebyshev, Fcg, Gcr, Ir, Minres, PipeCg
from . import stop

__all__ = ["Coo", "DeviceMatrixData", "entry_dtype", "Hybrid", "Bicg", "Bicgstab", "Chebyshev", "Gcr", "Ir", "Minres", "Cgs", "Fcg", "PipeCg", "Cdna4Executor", "Csr", "Dense", "Ell", "Sellp", "scalar",
           "stencil_csr", "Jacobi", "compute_storage_scheme", "Cg", "Gmres", "ortho_method", "Identity",
           "stop", "GkoError", "NotCompiled", "NotSupported",
           "DimensionMismatch", "LIB_PATH"]
