"""Matrices whose diagonal blocks have prescribed condition numbers, constant within a
storage group and different across groups, so that the adaptive block-Jacobi chooses
every storage precision; seeded."""
import numpy as np
import scipy.sparse as sp

CLASSES = (1.2, 8.0, 100.0, 5e4, 1e6, 1e9, 3.0, 40.0)


def graded_block_matrix(n_groups, bs, seed, coupling=1e-3):
    group_size = 64 // (1 << (bs - 1).bit_length())
    rng = np.random.default_rng(seed)
    blocks = []
    for g in range(n_groups):
        cond = CLASSES[g % len(CLASSES)]
        for _ in range(group_size):
            q1, _ = np.linalg.qr(rng.standard_normal((bs, bs)))
            q2, _ = np.linalg.qr(rng.standard_normal((bs, bs)))
            sv = np.geomspace(1.0, 1.0 / cond, bs)
            # every third group: magnitudes outside the range of half
            span = 9 if g % 3 == 2 else 2
            blocks.append((q1 * sv) @ q2.T * 10.0 ** rng.uniform(-span, span))
    n = len(blocks) * bs
    a = sp.block_diag(blocks, format="csr")
    if coupling:
        a = a + sp.random(n, n, density=2.0 / n, random_state=seed, format="csr") * coupling
    a = sp.csr_matrix(a)
    a.sort_indices()
    # block detection needs the full block pattern: make sure no block entry is exactly zero
    return a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)
