/* TEST INFRASTRUCTURE -- NOT part of the product.
 *
 * gko_oracle: a sequential, plain-C restatement of the algorithms on Ginkgo's
 * Krylov hot path (ReferenceExecutor kernels + the Cg driver + the benchmark
 * stencil generators).  It is the checker the HIP backend is compared with.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product (ginkgo_amd/, libgko_cdna4.so) never does.
 *
 * Parity pinning: tests/test_oracle_*.py check this file against (a) the
 * known-answer vectors of the reference's own unit tests
 * (reference/test/matrix/csr_kernels.cpp:353-364, :505-534;
 * reference/test/solver/cg_kernels.cpp:215-226, :407-424;
 * reference/test/preconditioner/jacobi_kernels.cpp; ...), (b) fixtures under
 * tests/golden/ produced by the real reference (oracle/_ref, built by
 * oracle/build_ref.py from /root/reference), and (c) when oracle/_ref is
 * present, the real reference in-process on random inputs.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off => no FMA contraction,
 * matching the reference compiled for baseline x86-64). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- (value, index) instantiations ----------------------------------- */
#define T double
#define I int32_t
#define SUF f64_i32
#include "gko_oracle_impl.inc"
#undef T
#undef I
#undef SUF

#define T double
#define I int64_t
#define SUF f64_i64
#include "gko_oracle_impl.inc"
#undef T
#undef I
#undef SUF

#define T float
#define I int32_t
#define SUF f32_i32
#include "gko_oracle_impl.inc"
#undef T
#undef I
#undef SUF

#define T float
#define I int64_t
#define SUF f32_i64
#include "gko_oracle_impl.inc"
#undef T
#undef I
#undef SUF

/* ---- (matrix, input, output, index) triples of a GINKGO_MIXED_PRECISION core ---- */
#define MT double
#define IT double
#define OT float
#define AT double
#define I int32_t
#define SUF f64_f64_f32_i32
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT double
#define IT double
#define OT float
#define AT double
#define I int64_t
#define SUF f64_f64_f32_i64
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT double
#define IT float
#define OT double
#define AT double
#define I int32_t
#define SUF f64_f32_f64_i32
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT double
#define IT float
#define OT double
#define AT double
#define I int64_t
#define SUF f64_f32_f64_i64
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT double
#define IT float
#define OT float
#define AT double
#define I int32_t
#define SUF f64_f32_f32_i32
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT double
#define IT float
#define OT float
#define AT double
#define I int64_t
#define SUF f64_f32_f32_i64
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT float
#define IT double
#define OT double
#define AT double
#define I int32_t
#define SUF f32_f64_f64_i32
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT float
#define IT double
#define OT double
#define AT double
#define I int64_t
#define SUF f32_f64_f64_i64
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT float
#define IT double
#define OT float
#define AT double
#define I int32_t
#define SUF f32_f64_f32_i32
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT float
#define IT double
#define OT float
#define AT double
#define I int64_t
#define SUF f32_f64_f32_i64
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT float
#define IT float
#define OT double
#define AT double
#define I int32_t
#define SUF f32_f32_f64_i32
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define MT float
#define IT float
#define OT double
#define AT double
#define I int64_t
#define SUF f32_f32_f64_i64
#include "gko_oracle_mixed.inc"
#undef MT
#undef IT
#undef OT
#undef AT
#undef I
#undef SUF

#define T double
#define SUF f64
#include "gko_oracle_val.inc"
#undef T
#undef SUF

#define T float
#define SUF f32
#include "gko_oracle_val.inc"
#undef T
#undef SUF

/* ---- stencil generators ----------------------------------------------
 * benchmark/utils/stencil_matrix.hpp:68-238 (generate_2d_stencil_subdomain)
 * and :264-453 (generate_3d_stencil_subdomain).  `dims` = number of
 * subdomains per dimension (x, y[, z]), `pos` = this subdomain, global grid =
 * g points per dimension (the reference derives g from target_local_size via
 * closest_nth_root; callers pass g directly).  restricted != 0 => 5-pt / 7-pt,
 * else 9-pt / 27-pt.  Emits the COO entries of the rows OWNED by the
 * subdomain, in the reference's emission order, with GLOBAL (renumbered,
 * subdomain-contiguous) row / column indices.  Returns the number of entries;
 * pass rows == NULL to only count. */
static int in_range(int64_t i, int64_t bound) { return 0 <= i && i < bound; }

typedef struct {
    int nd;
    int64_t g;
    int64_t dims[3], pos[3], dmin[3], drest[3], dp[3];
} sgrid;

static int64_t sub_size(const sgrid* s, int dim, int64_t i)
{
    return s->dmin[dim] + (i < s->drest[dim] ? 1 : 0);
}

static int64_t sub_off1(const sgrid* s, int dim, int64_t i)
{
    return s->dmin[dim] * i + (i < s->drest[dim] ? i : s->drest[dim]);
}

static int64_t sub_offset(const sgrid* s, int64_t pz, int64_t py, int64_t px)
{
    if (s->nd == 2) {
        return s->g * sub_off1(s, 1, py) + sub_size(s, 1, py) * sub_off1(s, 0, px);
    }
    return s->g * s->g * sub_off1(s, 2, pz) +
           s->g * sub_size(s, 2, pz) * sub_off1(s, 1, py) +
           sub_size(s, 2, pz) * sub_size(s, 1, py) * sub_off1(s, 0, px);
}

static int64_t target_pos(const sgrid* s, int dim, int64_t i)
{
    return in_range(i, s->dp[dim]) ? s->pos[dim]
                                   : (i < 0 ? s->pos[dim] - 1 : s->pos[dim] + 1);
}

static int64_t target_local(const sgrid* s, int dim, int64_t tp, int64_t i)
{
    const int64_t sz = sub_size(s, dim, tp);
    return in_range(i, sz) ? i : (i < 0 ? i + sz : i - sub_size(s, dim, s->pos[dim]));
}

static int64_t flat_idx(const sgrid* s, int64_t iz, int64_t iy, int64_t ix)
{
    const int64_t tpx = target_pos(s, 0, ix);
    const int64_t tpy = target_pos(s, 1, iy);
    const int64_t tpz = s->nd == 3 ? target_pos(s, 2, iz) : 0;
    if (!in_range(tpx, s->dims[0]) || !in_range(tpy, s->dims[1]) ||
        (s->nd == 3 && !in_range(tpz, s->dims[2]))) {
        return -1;
    }
    /* target_local_idx is evaluated against the TARGET subdomain's size */
    if (s->nd == 2) {
        return sub_offset(s, 0, tpy, tpx) + target_local(s, 0, tpx, ix) +
               target_local(s, 1, tpy, iy) * sub_size(s, 0, tpx);
    }
    return sub_offset(s, tpz, tpy, tpx) + target_local(s, 0, tpx, ix) +
           target_local(s, 1, tpy, iy) * sub_size(s, 0, tpx) +
           target_local(s, 2, tpz, iz) * sub_size(s, 0, tpx) * sub_size(s, 1, tpy);
}

int64_t oracle_stencil_subdomain(int nd, const int64_t* dims, const int64_t* pos,
                                 int64_t g, int restricted, int64_t* rows,
                                 int64_t* cols, double* vals,
                                 int64_t* local_size_out)
{
    sgrid s;
    memset(&s, 0, sizeof(s));
    s.nd = nd;
    s.g = g;
    for (int d = 0; d < 3; ++d) {
        s.dims[d] = d < nd ? dims[d] : 1;
        s.pos[d] = d < nd ? pos[d] : 0;
        s.dmin[d] = g / s.dims[d];
        s.drest[d] = g % s.dims[d];
    }
    for (int d = 0; d < 3; ++d) s.dp[d] = d < nd ? sub_size(&s, d, s.pos[d]) : 1;
    const int64_t global_size = nd == 2 ? g * g : g * g * g;
    int nnz_in_row = 0;
    for (int dz = (nd == 3 ? -1 : 0); dz <= (nd == 3 ? 1 : 0); ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
                if (!restricted || ((dz == 0) + (dy == 0) + (dx == 0) >= 2)) ++nnz_in_row;
    const double diag = (double)(nnz_in_row - 1);
    if (local_size_out) *local_size_out = s.dp[0] * s.dp[1] * s.dp[2];
    int64_t count = 0;
    for (int64_t iz = 0; iz < s.dp[2]; ++iz) {
        for (int64_t iy = 0; iy < s.dp[1]; ++iy) {
            for (int64_t ix = 0; ix < s.dp[0]; ++ix) {
                const int64_t row = flat_idx(&s, iz, iy, ix);
                for (int dz = (nd == 3 ? -1 : 0); dz <= (nd == 3 ? 1 : 0); ++dz) {
                    for (int dy = -1; dy <= 1; ++dy) {
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (restricted && ((dz == 0) + (dy == 0) + (dx == 0) < 2)) continue;
                            const int64_t col = flat_idx(&s, iz + dz, iy + dy, ix + dx);
                            if (in_range(col, global_size)) {
                                if (rows) {
                                    rows[count] = row;
                                    cols[count] = col;
                                    vals[count] = col != row ? -1.0 : diag;
                                }
                                ++count;
                            }
                        }
                    }
                }
            }
        }
    }
    return count;
}

/* matrix_data::sort_row_major + Csr::read (core/matrix/csr.cpp:558-630):
 * stable counting sort by row, then per-row sort by column.  Output
 * row_ptrs has n_rows+1 int64 entries; rows are shifted by row_offset. */
typedef struct {
    int64_t c;
    double v;
} cv_pair;

static int cmp_cv(const void* a, const void* b)
{
    const int64_t x = ((const cv_pair*)a)->c, y = ((const cv_pair*)b)->c;
    return (x > y) - (x < y);
}

void oracle_coo_to_csr(int64_t nnz, const int64_t* rows, const int64_t* cols,
                       const double* vals, int64_t row_offset, int64_t n_rows,
                       int64_t* row_ptrs, int64_t* out_cols, double* out_vals)
{
    memset(row_ptrs, 0, sizeof(int64_t) * (size_t)(n_rows + 1));
    for (int64_t k = 0; k < nnz; ++k) row_ptrs[rows[k] - row_offset + 1]++;
    for (int64_t r = 0; r < n_rows; ++r) row_ptrs[r + 1] += row_ptrs[r];
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_rows > 0 ? n_rows : 1));
    memcpy(fill, row_ptrs, sizeof(int64_t) * (size_t)n_rows);
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t p = fill[rows[k] - row_offset]++;
        out_cols[p] = cols[k];
        out_vals[p] = vals[k];
    }
    free(fill);
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t a = row_ptrs[r], e = row_ptrs[r + 1];
        int sorted = 1;
        for (int64_t k = a + 1; k < e; ++k) {
            if (out_cols[k - 1] > out_cols[k]) {
                sorted = 0;
                break;
            }
        }
        if (!sorted) {
            cv_pair* tmp = (cv_pair*)malloc(sizeof(cv_pair) * (size_t)(e - a));
            for (int64_t k = a; k < e; ++k) {
                tmp[k - a].c = out_cols[k];
                tmp[k - a].v = out_vals[k];
            }
            qsort(tmp, (size_t)(e - a), sizeof(cv_pair), cmp_cv);
            for (int64_t k = a; k < e; ++k) {
                out_cols[k] = tmp[k - a].c;
                out_vals[k] = tmp[k - a].v;
            }
            free(tmp);
        }
    }
}

/* Direct CSR (int32) of the single-domain stencil, without the COO detour:
 * same entries / order as oracle_stencil_subdomain({1,1,1}) + coo_to_csr.
 * Used for the larger parity sizes and the cpu_baseline matrix. */
int64_t oracle_stencil_csr_i32(int nd, int64_t g, int restricted,
                               int32_t* row_ptrs, int32_t* cols, double* vals)
{
    const int64_t gz = nd == 3 ? g : 1;
    int nnz_in_row = 0;
    for (int dz = (nd == 3 ? -1 : 0); dz <= (nd == 3 ? 1 : 0); ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
                if (!restricted || ((dz == 0) + (dy == 0) + (dx == 0) >= 2)) ++nnz_in_row;
    const double diag = (double)(nnz_in_row - 1);
    int64_t count = 0;
    if (row_ptrs) row_ptrs[0] = 0;
    for (int64_t iz = 0; iz < gz; ++iz) {
        for (int64_t iy = 0; iy < g; ++iy) {
            for (int64_t ix = 0; ix < g; ++ix) {
                const int64_t row = ix + iy * g + iz * g * g;
                for (int dz = (nd == 3 ? -1 : 0); dz <= (nd == 3 ? 1 : 0); ++dz) {
                    for (int dy = -1; dy <= 1; ++dy) {
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (restricted && ((dz == 0) + (dy == 0) + (dx == 0) < 2)) continue;
                            const int64_t jx = ix + dx, jy = iy + dy, jz = iz + dz;
                            if (!in_range(jx, g) || !in_range(jy, g) || !in_range(jz, gz)) continue;
                            if (cols) {
                                const int64_t col = jx + jy * g + jz * g * g;
                                cols[count] = (int32_t)col;
                                vals[count] = col != row ? -1.0 : diag;
                            }
                            ++count;
                        }
                    }
                }
                if (row_ptrs) row_ptrs[row + 1] = (int32_t)count;
            }
        }
    }
    return count;
}

/* ---- block-Jacobi with a fixed reduced storage precision, value type double --------
 * Storage types of core/preconditioner/jacobi_utils.hpp:15-37; conversions of
 * include/ginkgo/core/base/half.hpp:405-450 (half: via float, round to nearest even,
 * exponents below the normal half range give signed zero, above give infinity) and
 * core/base/extended_float.hpp:52-104 (truncated<>: the upper bits of the float /
 * double).  prec = precision_reduction byte (preserving << 4 | nonpreserving). */
static uint16_t oracle_float_to_half(float v)
{
    uint32_t f;
    memcpy(&f, &v, 4);
    const uint16_t sign = (uint16_t)((f >> 16) & 0x8000u);
    const uint32_t e = (f >> 23) & 0xffu, m = f & 0x007fffffu;
    if (e == 0xffu) return (uint16_t)(sign | 0x7c00u | (m ? 0x03ffu : 0u));
    if (e <= 112u) return sign;
    if (e - 112u >= 31u) return (uint16_t)(sign | 0x7c00u);
    const uint16_t res = (uint16_t)(sign | ((e - 112u) << 10) | (m >> 13));
    const uint32_t tail = m & 0x1fffu;
    return (uint16_t)(res + ((tail > 0x1000u || (tail == 0x1000u && (res & 1u))) ? 1u : 0u));
}

static float oracle_half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t bits;
    if (e == 0x1fu) {
        bits = sign | 0x7f800000u | (m ? 0x007fffffu : 0u);
    } else if (e == 0) {
        bits = sign;
    } else {
        bits = sign | ((e + 112u) << 23) | (m << 13);
    }
    float out;
    memcpy(&out, &bits, 4);
    return out;
}

static int oracle_prec_bytes(int prec)
{
    return prec == 0x01 || prec == 0x10 ? 4 : (prec == 0 ? 8 : 2);
}

static void oracle_store_reduced(int prec, void* base, int64_t idx, double v)
{
    uint64_t d;
    uint32_t f;
    const float vf = (float)v;
    memcpy(&d, &v, 8);
    memcpy(&f, &vf, 4);
    switch (prec) {
    case 0x01: ((float*)base)[idx] = vf; break;
    case 0x02: ((uint16_t*)base)[idx] = oracle_float_to_half(vf); break;
    case 0x10: ((uint32_t*)base)[idx] = (uint32_t)(d >> 32); break;
    case 0x11: ((uint16_t*)base)[idx] = (uint16_t)(f >> 16); break;
    case 0x20: ((uint16_t*)base)[idx] = (uint16_t)(d >> 48); break;
    default: ((double*)base)[idx] = v; break;
    }
}

static double oracle_load_reduced(int prec, const void* base, int64_t idx)
{
    uint64_t d;
    uint32_t f;
    double out;
    float fo;
    switch (prec) {
    case 0x01: return (double)((const float*)base)[idx];
    case 0x02: return (double)oracle_half_to_float(((const uint16_t*)base)[idx]);
    case 0x10:
        d = (uint64_t)((const uint32_t*)base)[idx] << 32;
        memcpy(&out, &d, 8);
        return out;
    case 0x11:
        f = (uint32_t)((const uint16_t*)base)[idx] << 16;
        memcpy(&fo, &f, 4);
        return (double)fo;
    case 0x20:
        d = (uint64_t)((const uint16_t*)base)[idx] << 48;
        memcpy(&out, &d, 8);
        return out;
    default: return ((const double*)base)[idx];
    }
}

/* narrow full-precision block storage (as oracle_jacobi_generate wrote it) in place:
 * group by group, entry (r, c) of block b goes to element index
 * block_offset (b & mask) + r + c stride of the narrower type
 * (reference/preconditioner/jacobi_kernels.cpp:393-408) */
void oracle_jacobi_convert_storage_f64(int64_t num_blocks, int64_t block_offset,
                                       int64_t group_offset, uint32_t group_power,
                                       double* blocks, int prec)
{
    if (prec == 0) return;
    const int64_t gsize = (int64_t)1 << group_power;
    const int64_t groups = (num_blocks + gsize - 1) / gsize;
    double* tmp = (double*)malloc(sizeof(double) * (size_t)group_offset);
    for (int64_t g = 0; g < groups; ++g) {
        double* gp = blocks + group_offset * g;
        memcpy(tmp, gp, sizeof(double) * (size_t)group_offset);
        for (int64_t i = 0; i < group_offset; ++i) oracle_store_reduced(prec, gp, i, tmp[i]);
    }
    free(tmp);
    (void)block_offset;
}

/* reference/preconditioner/jacobi_kernels.cpp:413-531 (apply_block with a
 * BlockValueType and its default_converter; apply / simple_apply) */
void oracle_jacobi_apply_stored_f64_i32(int64_t num_blocks, int64_t block_offset,
                                        int64_t group_offset, uint32_t group_power,
                                        const int32_t* block_ptrs, const double* blocks,
                                        int prec, double alpha, const double* b, int64_t ldb,
                                        double beta, double* x, int64_t ldx, int64_t nrhs)
{
    const int64_t stride = block_offset << group_power;
    const int64_t gmask = ((int64_t)1 << group_power) - 1;
    for (int64_t blk = 0; blk < num_blocks; ++blk) {
        const void* group = blocks + group_offset * (blk >> group_power);
        const int64_t off = block_offset * (blk & gmask);
        const int64_t start = block_ptrs[blk];
        const int64_t bs = block_ptrs[blk + 1] - start;
        const double* bb = b + ldb * start;
        double* bx = x + ldx * start;
        for (int64_t row = 0; row < bs; ++row) {
            for (int64_t col = 0; col < nrhs; ++col) {
                if (beta != 0.0) {
                    bx[row * ldx + col] *= beta;
                } else {
                    bx[row * ldx + col] = 0.0;
                }
            }
        }
        for (int64_t inner = 0; inner < bs; ++inner) {
            for (int64_t row = 0; row < bs; ++row) {
                for (int64_t col = 0; col < nrhs; ++col) {
                    bx[row * ldx + col] += alpha * oracle_load_reduced(prec, group, off + row + inner * stride) *
                                           bb[inner * ldb + col];
                }
            }
        }
    }
    (void)oracle_prec_bytes;
}

/* ---- adaptive block-Jacobi: generate with per-block storage precisions ----------
 * reference/preconditioner/jacobi_kernels.cpp:313-411 (generate), :280-307
 * (validate_precision_reduction_feasibility), reference/components/
 * matrix_operations.hpp:22-37 (compute_inf_norm, indexes the row-major block as
 * i + j * stride), core/preconditioner/jacobi_utils.hpp:104-176
 * (get_supported_storage_reductions, get_optimal_storage_reduction).
 * prec[b] in: the requested precision_reduction byte of block b (0xff = autodetect);
 * out: the precision of its group.  cond[b] = ||B||.||B^-1|| as computed there. */
static double oracle_block_norm(int bs, const double* m, int stride)
{
    double result = 0.0;
    for (int i = 0; i < bs; ++i) {
        double tmp = 0.0;
        for (int j = 0; j < bs; ++j) tmp += fabs(m[i + j * stride]);
        result = result > tmp ? result : tmp;
    }
    return result;
}

static double oracle_round_to(int prec, double v)
{
    double buf[1];
    oracle_store_reduced(prec, buf, 0, v);
    return oracle_load_reduced(prec, buf, 0);
}

static int oracle_feasible(int prec, int bs, const double* block)
{
    double tmp[64 * 64];
    int32_t perm[64];
    for (int i = 0; i < bs; ++i) {
        perm[i] = i;
        for (int j = 0; j < bs; ++j) tmp[i * bs + j] = oracle_round_to(prec, block[i * bs + j]);
    }
    double cond = oracle_block_norm(bs, tmp, bs);
    if (!oracle_invert_block_f64_i32(bs, perm, tmp, bs)) return 0;
    cond *= oracle_block_norm(bs, tmp, bs);
    return cond >= 1.0 && cond * (1.0 / 9007199254740992.0) < 1e-3; /* eps(double) = 2^-53 */
}

enum { PRD_P0N2 = 0x01, PRD_P1N1 = 0x02, PRD_P2N0 = 0x04, PRD_P0N1 = 0x08, PRD_P1N0 = 0x10 };

static uint32_t oracle_singleton(uint8_t pr)
{
    switch (pr) {
    case 0x01: return PRD_P0N1;
    case 0x02: return PRD_P0N2;
    case 0x10: return PRD_P1N0;
    case 0x11: return PRD_P1N1;
    case 0x20: return PRD_P2N0;
    default: return 0;
    }
}

static uint32_t oracle_supported_reductions(double accuracy, double cond, int bs,
                                            const double* block)
{
    /* eps of truncated<double,4>, truncated<float,2>, half, truncated<double,2>, float */
    const double e_p2n0 = 1.0 / 16, e_p1n1 = 1.0 / 128, e_p0n2 = 1.0 / 2048,
                 e_p1n0 = 1.0 / 1048576, e_p0n1 = 1.0 / 16777216;
    int verified1 = 2;
    uint32_t supported = 0;
    if (cond * e_p2n0 < accuracy) supported |= PRD_P2N0;
    if (cond * e_p1n1 < accuracy && (verified1 = oracle_feasible(0x01, bs, block))) {
        supported |= PRD_P1N1;
    }
    if (cond * e_p0n2 < accuracy && verified1 != 0 && oracle_feasible(0x02, bs, block)) {
        supported |= PRD_P0N2;
    }
    if (cond * e_p1n0 < accuracy) supported |= PRD_P1N0;
    if (cond * e_p0n1 < accuracy &&
        (verified1 == 1 || (verified1 == 2 && (verified1 = oracle_feasible(0x01, bs, block))))) {
        supported |= PRD_P0N1;
    }
    return supported;
}

static uint8_t oracle_optimal_reduction(uint32_t supported)
{
    if (supported & PRD_P0N2) return 0x02;
    if (supported & PRD_P1N1) return 0x11;
    if (supported & PRD_P2N0) return 0x20;
    if (supported & PRD_P0N1) return 0x01;
    if (supported & PRD_P1N0) return 0x10;
    return 0x00;
}

void oracle_jacobi_generate_adaptive_f64_i32(const int32_t* row_ptrs, const int32_t* cols,
                                             const double* vals, int64_t num_blocks,
                                             int64_t block_offset, int64_t group_offset,
                                             uint32_t group_power, const int32_t* block_ptrs,
                                             double accuracy, uint8_t* prec, double* cond,
                                             double* blocks)
{
    const int64_t gsize = (int64_t)1 << group_power;
    const int64_t stride = block_offset << group_power;
    double* blk = (double*)malloc(sizeof(double) * 64 * 64 * (size_t)gsize);
    int32_t* perm = (int32_t*)malloc(sizeof(int32_t) * 64 * (size_t)gsize);
    for (int64_t g = 0; g < num_blocks; g += gsize) {
        uint32_t all = 0xffffffffu;
        for (int64_t b = 0; b < gsize && g + b < num_blocks; ++b) {
            double* block = blk + 64 * 64 * b;
            int32_t* pm = perm + 64 * b;
            const int64_t start = block_ptrs[g + b];
            const int bs = (int)(block_ptrs[g + b + 1] - start);
            for (int i = 0; i < bs; ++i) {
                pm[i] = i;
                for (int j = 0; j < bs; ++j) block[i * bs + j] = 0.0;
            }
            for (int row = 0; row < bs; ++row) {
                for (int64_t k = row_ptrs[start + row]; k < row_ptrs[start + row + 1]; ++k) {
                    const int64_t col = (int64_t)cols[k] - start;
                    if (0 <= col && col < bs) block[row * bs + col] = vals[k];
                }
            }
            cond[g + b] = oracle_block_norm(bs, block, bs);
            oracle_invert_block_f64_i32(bs, pm, block, bs);
            cond[g + b] *= oracle_block_norm(bs, block, bs);
            uint32_t d;
            if (prec[g + b] == 0xff) {
                d = oracle_supported_reductions(accuracy, cond[g + b], bs, block);
            } else {
                d = oracle_singleton(prec[g + b]);
            }
            all &= d;
        }
        const uint8_t p = oracle_optimal_reduction(all);
        for (int64_t b = 0; b < gsize && g + b < num_blocks; ++b) {
            const double* block = blk + 64 * 64 * b;
            const int32_t* pm = perm + 64 * b;
            const int bs = (int)(block_ptrs[g + b + 1] - block_ptrs[g + b]);
            prec[g + b] = p;
            void* group = blocks + group_offset * ((g + b) >> group_power);
            const int64_t off = block_offset * b;
            for (int i = 0; i < bs; ++i) {
                for (int j = 0; j < bs; ++j) {
                    oracle_store_reduced(p, group, off + i + (int64_t)pm[j] * stride, block[i * bs + j]);
                }
            }
        }
    }
    free(blk);
    free(perm);
}

/* apply with the per-block precisions the adaptive generate chose */
void oracle_jacobi_apply_adaptive_f64_i32(int64_t num_blocks, int64_t block_offset,
                                          int64_t group_offset, uint32_t group_power,
                                          const int32_t* block_ptrs, const double* blocks,
                                          const uint8_t* prec, double alpha, const double* b,
                                          int64_t ldb, double beta, double* x, int64_t ldx,
                                          int64_t nrhs)
{
    for (int64_t blk = 0; blk < num_blocks; ++blk) {
        /* one block at a time through the fixed-precision routine */
        const int64_t gmask = ((int64_t)1 << group_power) - 1;
        const int64_t stride = block_offset << group_power;
        const void* group = blocks + group_offset * (blk >> group_power);
        const int64_t off = block_offset * (blk & gmask);
        const int64_t start = block_ptrs[blk];
        const int64_t bs = block_ptrs[blk + 1] - start;
        const double* bb = b + ldb * start;
        double* bx = x + ldx * start;
        for (int64_t row = 0; row < bs; ++row) {
            for (int64_t col = 0; col < nrhs; ++col) {
                if (beta != 0.0) {
                    bx[row * ldx + col] *= beta;
                } else {
                    bx[row * ldx + col] = 0.0;
                }
            }
        }
        for (int64_t inner = 0; inner < bs; ++inner) {
            for (int64_t row = 0; row < bs; ++row) {
                for (int64_t col = 0; col < nrhs; ++col) {
                    bx[row * ldx + col] +=
                        alpha * oracle_load_reduced(prec[blk], group, off + row + inner * stride) *
                        bb[inner * ldb + col];
                }
            }
        }
    }
}

/* ---- the same for float, complex<float>, complex<double> (gko_oracle_jacobi_types.inc) ---------- */
#define KIND_OF_DOUBLE(p) ((p) == 0x01 || (p) == 0x02 || (p) == 0x10 || (p) == 0x11 || (p) == 0x20 ? (p) : 0)
#define KIND_OF_FLOAT(p) ((p) == 0x01 || (p) == 0x02 || (p) == 0x11 ? 0x02 : ((p) == 0x10 || (p) == 0x20 ? 0x11 : 0))

/* float components: eps of truncated<float,2>, half, half, truncated<float,2>, half; own 2^-24 */
#define EPS_P2N0 (1.0 / 128)
#define EPS_P1N1 (1.0 / 2048)
#define EPS_P0N2 (1.0 / 2048)
#define EPS_P1N0 (1.0 / 128)
#define EPS_P0N1 (1.0 / 2048)
#define EPS_OWN (1.0 / 16777216)
#define VERIFY1 0x02
#define VERIFY2 0x02
#define KIND(p) KIND_OF_FLOAT(p)
#define R float

#define T float
#define SUFT f32_i32
#define ABS_T(v) fabsf(v)
#define CPLX 0
#include "gko_oracle_jacobi_types.inc"
#undef T
#undef SUFT
#undef ABS_T
#undef CPLX

#define T float _Complex
#define SUFT c64_i32
#define ABS_T(v) __builtin_cabsf(v)
#define CPLX 1
#include "gko_oracle_jacobi_types.inc"
#undef T
#undef SUFT
#undef ABS_T
#undef CPLX

#undef EPS_P2N0
#undef EPS_P1N1
#undef EPS_P0N2
#undef EPS_P1N0
#undef EPS_P0N1
#undef EPS_OWN
#undef VERIFY1
#undef VERIFY2
#undef KIND
#undef R

/* double components: eps of truncated<double,4>, truncated<float,2>, half, truncated<double,2>, float */
#define EPS_P2N0 (1.0 / 16)
#define EPS_P1N1 (1.0 / 128)
#define EPS_P0N2 (1.0 / 2048)
#define EPS_P1N0 (1.0 / 1048576)
#define EPS_P0N1 (1.0 / 16777216)
#define EPS_OWN (1.0 / 9007199254740992.0)
#define VERIFY1 0x01
#define VERIFY2 0x02
#define KIND(p) KIND_OF_DOUBLE(p)
#define R double
#define T double _Complex
#define SUFT c128_i32
#define ABS_T(v) __builtin_cabs(v)
#define CPLX 1
#include "gko_oracle_jacobi_types.inc"
#undef T
#undef SUFT
#undef ABS_T
#undef CPLX
#undef EPS_P2N0
#undef EPS_P1N1
#undef EPS_P0N2
#undef EPS_P1N0
#undef EPS_P0N1
#undef EPS_OWN
#undef VERIFY1
#undef VERIFY2
#undef KIND
#undef R

/* ---- CG driver ---------------------------------------------------------
 * core/solver/cg.cpp:93-181 (Cg::apply_dense_impl) with
 *   - preconditioner: 0 = Identity (z = r), 1 = scalar Jacobi, 2 = block Jacobi
 *     (core/preconditioner/jacobi.cpp:150-165)
 *   - criterion: Combined(Iteration(max_iters), ResidualNorm(reduction,
 *     baseline)) in that order (core/stop/combined.cpp:33-51,
 *     core/stop/iteration.cpp:15-26, core/stop/residual_norm.cpp:75-205);
 *     baseline: 0 = rhs_norm, 1 = initial_resnorm, 2 = absolute.
 * One right-hand side (nrhs = 1), f64 / int32.  Returns the iteration count
 * at which the loop was left; *resnorm_out = ||r||_2 of the last check. */
typedef struct {
    int precond; /* 0 none, 1 scalar, 2 block */
    const double* inv_diag;
    int64_t num_blocks, block_offset, group_offset;
    uint32_t group_power;
    const int32_t* block_ptrs;
    const double* blocks;
} oracle_precond;

static void apply_precond(const oracle_precond* m, int64_t n, const double* r,
                          double* z)
{
    if (m->precond == 1) {
        oracle_jacobi_scalar_apply_f64(n, 1, m->inv_diag, 0, 1.0, r, 1, 0.0, z, 1);
    } else if (m->precond == 2) {
        oracle_jacobi_apply_f64_i32(m->num_blocks, m->block_offset,
                                    m->group_offset, m->group_power,
                                    m->block_ptrs, m->blocks, 1.0, r, 1, 0.0, z,
                                    1, 1);
    } else {
        memcpy(z, r, sizeof(double) * (size_t)n);
    }
}

int64_t oracle_cg_solve_f64_i32(int64_t n, const int32_t* row_ptrs,
                                const int32_t* cols, const double* vals,
                                const oracle_precond* m, const double* b,
                                double* x, int64_t max_iters, double reduction,
                                int baseline, double* resnorm_out,
                                double* resnorm_history /* may be NULL, max_iters+1 */)
{
    double* r = (double*)malloc(sizeof(double) * (size_t)n * 4);
    double *z = r + n, *p = z + n, *q = p + n;
    double beta, prev_rho, rho, tau, tau0;
    uint8_t stop = 0;
    oracle_cg_initialize_f64(n, 1, b, 1, r, 1, z, 1, p, 1, q, 1, &prev_rho, &rho, &stop);
    /* r = b - A x */
    oracle_csr_advanced_spmv_f64_i32(n, -1.0, row_ptrs, cols, vals, x, 1, 1.0, r, 1, 1);
    if (baseline == 0) {
        oracle_dense_compute_norm2_f64(n, 1, b, 1, &tau0, 0);
    } else if (baseline == 1) {
        oracle_dense_compute_norm2_f64(n, 1, r, 1, &tau0, 0);
    } else {
        tau0 = 1.0;
    }
    int64_t iter = -1;
    for (;;) {
        apply_precond(m, n, r, z);
        oracle_dense_compute_dot_f64(n, 1, r, 1, z, 1, &rho);
        ++iter;
        int one_changed = 0, all_stopped = 0;
        /* Combined: Iteration first, then ResidualNorm */
        if (iter >= max_iters) {
            if ((stop & 0x3f) == 0) stop |= (uint8_t)(1 & 0x3f) | 0x40;
            all_stopped = 1;
        }
        oracle_dense_compute_norm2_f64(n, 1, r, 1, &tau, 0);
        if (resnorm_history && iter <= max_iters) resnorm_history[iter] = tau;
        if (!all_stopped) {
            all_stopped = oracle_residual_norm_f64(1, &tau, &tau0, reduction, 2, 1,
                                                   &stop, 0, &one_changed);
        }
        if (all_stopped) break;
        oracle_cg_step_1_f64(n, 1, p, 1, z, 1, &rho, &prev_rho, &stop);
        oracle_csr_spmv_f64_i32(n, row_ptrs, cols, vals, p, 1, q, 1, 1);
        oracle_dense_compute_dot_f64(n, 1, p, 1, q, 1, &beta);
        oracle_cg_step_2_f64(n, 1, x, 1, r, 1, p, 1, q, 1, &beta, &rho, &stop);
        {
            const double t = prev_rho;
            prev_rho = rho;
            rho = t;
        }
    }
    if (resnorm_out) *resnorm_out = tau;
    free(r);
    return iter;
}

/* ---- the other Krylov drivers ------------------------------------------
 * One right-hand side, f64 / int32, same preconditioner and criterion
 * conventions as oracle_cg_solve (Combined(Iteration, ResidualNorm(baseline)),
 * ids 1 and 2).  kind:
 *   1 Bicgstab  core/solver/bicgstab.cpp:95-236
 *   2 Cgs       core/solver/cgs.cpp:96-201
 *   3 Fcg       core/solver/fcg.cpp:94-183
 *   4 PipeCg    core/solver/pipe_cg.cpp:95-297 (the reference interleaves r/w and
 *               z1/z2 as two columns to merge the two dots; the values are the
 *               same with separate vectors, which is what is restated here)
 * Returns the iteration count at exit; *resnorm_out = ||r||_2 as the reference's
 * Convergence logger reports it. */
/* Jacobi::transpose (core/preconditioner/jacobi.cpp; jacobi::transpose_jacobi,
 * reference/preconditioner/jacobi_kernels.cpp:597-627): the same scheme with every block
 * transposed; scalar Jacobi and Identity are their own transposes.  Full precision. */
static double* transposed_blocks(const oracle_precond* m)
{
    if (m->precond != 2) return NULL;
    const int64_t gsize = (int64_t)1 << m->group_power;
    const int64_t groups = (m->num_blocks + gsize - 1) / gsize;
    const int64_t stride = m->block_offset << m->group_power;
    double* out = (double*)calloc((size_t)(groups * m->group_offset), sizeof(double));
    for (int64_t blk = 0; blk < m->num_blocks; ++blk) {
        const int64_t base = m->group_offset * (blk >> m->group_power) +
                             m->block_offset * (blk & (gsize - 1));
        const int64_t bs = m->block_ptrs[blk + 1] - m->block_ptrs[blk];
        for (int64_t r = 0; r < bs; ++r) {
            for (int64_t c = 0; c < bs; ++c) {
                out[base + r + c * stride] = m->blocks[base + c + r * stride];
            }
        }
    }
    return out;
}

static int krylov_check(int64_t iter, int64_t max_iters, int64_t n,
                        const double* res, double tau0, double reduction,
                        int set_finalized, uint8_t* stop, int* one_changed,
                        double* tau)
{
    *one_changed = 0;
    /* (the logger computes this norm itself when the Iteration criterion ends the loop) */
    oracle_dense_compute_norm2_f64(n, 1, res, 1, tau, 0);
    if (iter >= max_iters) { /* stop::Iteration, id 1 (iteration.cpp:15-26) */
        if ((*stop & 0x3f) == 0) *stop |= (uint8_t)1 | (set_finalized ? 0x40 : 0);
        *one_changed = 1;
        return 1;
    }
    return oracle_residual_norm_f64(1, tau, &tau0, reduction, 2, set_finalized, stop,
                                    0, one_changed);
}

static int64_t oracle_stationary_solve(int kind, int64_t n, const int32_t* row_ptrs,
                                       const int32_t* cols, const double* vals,
                                       const oracle_precond* m, const double* b, double* x,
                                       int64_t max_iters, double reduction, int baseline,
                                       double p0, double p1, double* resnorm_out);

int64_t oracle_krylov_solve_f64_i32(int kind, int64_t n, const int32_t* row_ptrs,
                                    const int32_t* cols, const double* vals,
                                    const oracle_precond* m, const double* b,
                                    double* x, int64_t max_iters, double reduction,
                                    int baseline, double p0, double p1,
                                    double* resnorm_out)
{
    if (kind == 5 || kind == 6) {
        return oracle_stationary_solve(kind, n, row_ptrs, cols, vals, m, b, x, max_iters,
                                       reduction, baseline, p0, p1, resnorm_out);
    }
    const size_t N = (size_t)n;
    double* w = (double*)calloc(N * 12, sizeof(double));
    double* V[12];
    for (int k = 0; k < 12; ++k) V[k] = w + N * (size_t)k;
    double tau = 0.0, tau0 = 1.0;
    uint8_t stop = 0;
    int one_changed = 0;
    int64_t iter = -1;
#define SPMV(in, out) oracle_csr_spmv_f64_i32(n, row_ptrs, cols, vals, in, 1, out, 1, 1)
#define RESID(r) oracle_csr_advanced_spmv_f64_i32(n, -1.0, row_ptrs, cols, vals, x, 1, 1.0, r, 1, 1)
#define DOT(a, c, out) oracle_dense_compute_dot_f64(n, 1, a, 1, c, 1, out)
#define BASELINE(r)                                                     \
    do {                                                                \
        if (baseline == 0) oracle_dense_compute_norm2_f64(n, 1, b, 1, &tau0, 0); \
        else if (baseline == 1) oracle_dense_compute_norm2_f64(n, 1, r, 1, &tau0, 0); \
        else tau0 = 1.0;                                                \
    } while (0)
    if (kind == 1) {
        double *r = V[0], *z = V[1], *y = V[2], *v = V[3], *s = V[4], *t = V[5], *p = V[6],
               *rr = V[7];
        double alpha, beta, gamma, prev_rho, rho, omega;
        oracle_bicgstab_initialize_f64(n, 1, 1, b, r, rr, y, s, t, z, v, p, &prev_rho, &rho,
                                       &alpha, &beta, &gamma, &omega, &stop);
        RESID(r);
        BASELINE(r);
        memcpy(rr, r, sizeof(double) * N);
        for (;;) {
            ++iter;
            DOT(rr, r, &rho);
            if (krylov_check(iter, max_iters, n, r, tau0, reduction, 1, &stop, &one_changed, &tau)) break;
            oracle_bicgstab_step_1_f64(n, 1, 1, r, p, v, &rho, &prev_rho, &alpha, &omega, &stop);
            apply_precond(m, n, p, y);
            SPMV(y, v);
            DOT(rr, v, &beta);
            oracle_bicgstab_step_2_f64(n, 1, 1, r, s, v, &rho, &alpha, &beta, &stop);
            /* the Convergence logger is handed r, not s, after this check
             * (bicgstab.cpp:183-185): the reported norm stays ||r|| */
            double tau_s = 0.0;
            const int all = krylov_check(iter, max_iters, n, s, tau0, reduction, 0, &stop,
                                         &one_changed, &tau_s);
            if (one_changed) oracle_bicgstab_finalize_f64(n, 1, 1, x, y, &alpha, &stop);
            if (all) break;
            apply_precond(m, n, s, z);
            SPMV(z, t);
            DOT(s, t, &gamma);
            DOT(t, t, &beta);
            oracle_bicgstab_step_3_f64(n, 1, 1, x, r, s, t, y, z, &alpha, &beta, &gamma, &omega, &stop);
            { const double tmp = prev_rho; prev_rho = rho; rho = tmp; }
        }
    } else if (kind == 2) {
        double *r = V[0], *r_tld = V[1], *p = V[2], *q = V[3], *u = V[4], *u_hat = V[5],
               *v_hat = V[6], *t = V[7];
        double alpha, beta, gamma, prev_rho, rho;
        oracle_cgs_initialize_f64(n, 1, 1, b, r, r_tld, p, q, u, u_hat, v_hat, t, &alpha, &beta,
                                  &gamma, &prev_rho, &rho, &stop);
        RESID(r);
        BASELINE(r);
        memcpy(r_tld, r, sizeof(double) * N);
        for (;;) {
            DOT(r, r_tld, &rho);
            ++iter;
            if (krylov_check(iter, max_iters, n, r, tau0, reduction, 1, &stop, &one_changed, &tau)) break;
            oracle_cgs_step_1_f64(n, 1, 1, r, u, p, q, &beta, &rho, &prev_rho, &stop);
            apply_precond(m, n, p, t);
            SPMV(t, v_hat);
            DOT(r_tld, v_hat, &gamma);
            oracle_cgs_step_2_f64(n, 1, 1, u, v_hat, q, t, &alpha, &rho, &gamma, &stop);
            apply_precond(m, n, t, u_hat);
            SPMV(u_hat, t);
            oracle_cgs_step_3_f64(n, 1, 1, t, u_hat, r, x, &alpha, &stop);
            { const double tmp = prev_rho; prev_rho = rho; rho = tmp; }
        }
    } else if (kind == 3) {
        double *r = V[0], *z = V[1], *p = V[2], *q = V[3], *t = V[4];
        double beta, prev_rho, rho, rho_t;
        oracle_fcg_initialize_f64(n, 1, 1, b, r, z, p, q, t, &prev_rho, &rho, &rho_t, &stop);
        RESID(r);
        BASELINE(r);
        for (;;) {
            apply_precond(m, n, r, z);
            DOT(r, z, &rho);
            DOT(t, z, &rho_t);
            ++iter;
            if (krylov_check(iter, max_iters, n, r, tau0, reduction, 1, &stop, &one_changed, &tau)) break;
            oracle_fcg_step_1_f64(n, 1, 1, p, z, &rho_t, &prev_rho, &stop);
            SPMV(p, q);
            DOT(p, q, &beta);
            oracle_fcg_step_2_f64(n, 1, 1, x, r, t, p, q, &beta, &rho, &stop);
            { const double tmp = prev_rho; prev_rho = rho; rho = tmp; }
        }
    } else if (kind == 4) {
        double *r = V[0], *wv = V[1], *z1 = V[2], *z2 = V[3], *p = V[4], *mm = V[5], *nn = V[6],
               *q = V[7], *f = V[8], *g = V[9];
        double rho, delta, beta, prev_rho;
        oracle_pipe_cg_initialize_1_f64(n, 1, 1, b, r, &prev_rho, &stop);
        RESID(r);
        BASELINE(r);
        apply_precond(m, n, r, z1);
        memcpy(z2, z1, sizeof(double) * N);
        SPMV(z1, wv);
        apply_precond(m, n, wv, mm);
        SPMV(mm, nn);
        DOT(r, z1, &rho);
        DOT(wv, z2, &delta);
        iter = 0;
        if (!krylov_check(iter, max_iters, n, r, tau0, reduction, 1, &stop, &one_changed, &tau)) {
            oracle_pipe_cg_initialize_2_f64(n, 1, 1, p, q, f, g, &beta, z1, wv, mm, nn, &delta);
            for (;;) {
                oracle_pipe_cg_step_1_f64(n, 1, 1, x, r, z1, z2, wv, p, q, f, g, &rho, &beta, &stop);
                apply_precond(m, n, wv, mm);
                SPMV(mm, nn);
                prev_rho = rho;
                DOT(r, z1, &rho);
                DOT(wv, z2, &delta);
                ++iter;
                if (krylov_check(iter, max_iters, n, r, tau0, reduction, 1, &stop, &one_changed, &tau)) break;
                oracle_pipe_cg_step_2_f64(n, 1, 1, &beta, p, q, f, g, z1, wv, mm, nn, &prev_rho, &rho,
                                          &delta, &stop);
            }
        }
    } else if (kind == 7) { /* Bicg, core/solver/bicg.cpp:106-230 */
        double *r = V[0], *z = V[1], *p = V[2], *q = V[3], *r2 = V[4], *z2 = V[5], *p2 = V[6],
               *q2 = V[7];
        double beta, prev_rho, rho;
        /* A^T through csr::conj_transpose, M^T through Jacobi::conj_transpose */
        int32_t* t_ptrs = (int32_t*)malloc(sizeof(int32_t) * (N + 1));
        int32_t* t_cols = (int32_t*)malloc(sizeof(int32_t) * (size_t)row_ptrs[n]);
        double* t_vals = (double*)malloc(sizeof(double) * (size_t)row_ptrs[n]);
        oracle_csr_transpose_f64_i32(n, n, row_ptrs, cols, vals, t_ptrs, t_cols, t_vals);
        oracle_precond mt = *m;
        double* tb = transposed_blocks(m);
        if (tb) mt.blocks = tb;
        oracle_bicg_initialize_f64(n, 1, 1, b, r, z, p, q, &prev_rho, &rho, r2, z2, p2, q2, &stop);
        RESID(r);
        BASELINE(r);
        memcpy(r2, r, sizeof(double) * N);
        for (;;) {
            apply_precond(m, n, r, z);
            apply_precond(&mt, n, r2, z2);
            DOT(z, r2, &rho);
            ++iter;
            if (krylov_check(iter, max_iters, n, r, tau0, reduction, 1, &stop, &one_changed, &tau)) break;
            oracle_bicg_step_1_f64(n, 1, 1, p, z, p2, z2, &rho, &prev_rho, &stop);
            SPMV(p, q);
            oracle_csr_spmv_f64_i32(n, t_ptrs, t_cols, t_vals, p2, 1, q2, 1, 1);
            DOT(p2, q, &beta);
            oracle_bicg_step_2_f64(n, 1, 1, x, r, r2, p, q, q2, &beta, &rho, &stop);
            { const double tmp = prev_rho; prev_rho = rho; rho = tmp; }
        }
        free(t_ptrs);
        free(t_cols);
        free(t_vals);
        free(tb);
    } else if (kind == 8) { /* Gcr, core/solver/gcr.cpp:95-320; p0 = krylov_dim */
        const int64_t kd = (int64_t)p0 > 0 ? (int64_t)p0 : 100;
        double* base = (double*)calloc(N * (size_t)(2 * (kd + 1) + 3), sizeof(double));
        double *r = base, *pr = base + N, *apr = base + 2 * N;
        double* P = base + 3 * N;
        double* AP = P + N * (size_t)(kd + 1);
        double* ap_norms = (double*)calloc((size_t)kd + 1, sizeof(double));
        double rap, minus_beta, rnorm;
        int64_t restart_iter = 0;
        memcpy(r, b, sizeof(double) * N); /* gcr::initialize */
        stop = 0;
        RESID(r);
        BASELINE(r);
        apply_precond(m, n, r, pr);
        SPMV(pr, apr);
        memcpy(P, pr, sizeof(double) * N); /* gcr::restart */
        memcpy(AP, apr, sizeof(double) * N);
        for (;;) {
            ++iter;
            oracle_dense_compute_norm2_f64(n, 1, r, 1, &rnorm, 0);
            tau = rnorm;
            one_changed = 0;
            {
                int all = 0;
                if (iter >= max_iters) {
                    if ((stop & 0x3f) == 0) stop |= (uint8_t)1 | 0x40;
                    all = 1;
                } else {
                    all = oracle_residual_norm_f64(1, &rnorm, &tau0, reduction, 2, 1, &stop, 0,
                                                   &one_changed);
                }
                if (all) break;
            }
            if (restart_iter == kd) {
                memcpy(P, pr, sizeof(double) * N);
                memcpy(AP, apr, sizeof(double) * N);
                restart_iter = 0;
            }
            double* Ap = AP + N * (size_t)restart_iter;
            double* pp = P + N * (size_t)restart_iter;
            DOT(r, Ap, &rap);
            oracle_dense_compute_norm2_f64(n, 1, Ap, 1, &ap_norms[restart_iter], 1);
            if (!(stop & 0x3f) && ap_norms[restart_iter] != 0.0) { /* gcr::step_1 */
                const double t = rap / ap_norms[restart_iter];
                for (int64_t i = 0; i < n; ++i) {
                    x[i] += t * pp[i];
                    r[i] -= t * Ap[i];
                }
            }
            apply_precond(m, n, r, pr);
            SPMV(pr, apr);
            double* next_Ap = AP + N * (size_t)(restart_iter + 1);
            double* next_p = P + N * (size_t)(restart_iter + 1);
            memcpy(next_Ap, apr, sizeof(double) * N);
            memcpy(next_p, pr, sizeof(double) * N);
            for (int64_t k = 0; k <= restart_iter; ++k) {
                const double* Apk = AP + N * (size_t)k;
                const double* pk = P + N * (size_t)k;
                DOT(apr, Apk, &minus_beta);
                minus_beta = minus_beta / ap_norms[k]; /* dense::inv_scale */
                for (int64_t i = 0; i < n; ++i) next_Ap[i] -= minus_beta * Apk[i];
                for (int64_t i = 0; i < n; ++i) next_p[i] -= minus_beta * pk[i];
            }
            ++restart_iter;
        }
        free(base);
        free(ap_norms);
    } else if (kind == 9) { /* Minres, core/solver/minres.cpp:110-230 */
        double *r = V[0], *z = V[1], *p = V[2], *q = V[3], *v = V[4], *z_tilde = V[5],
               *p_prev = V[6], *q_prev = V[7], *rt = V[8];
        double alpha = 0, beta, gamma, delta, eta_next, eta, tau_sq, cos_prev, cosv, sin_prev, sinv;
        memcpy(r, b, sizeof(double) * N);
        RESID(r);
        BASELINE(r);
        apply_precond(m, n, r, z);
        DOT(r, z, &beta);
        DOT(z, z, &tau_sq);
        /* the driver hands v as q_tilde to initialize (minres.cpp:169-178) */
        oracle_minres_initialize_f64(n, 1, 1, r, z, p, p_prev, q, q_prev, v, &beta, &gamma, &delta,
                                     &cos_prev, &cosv, &sin_prev, &sinv, &eta_next, &eta, &stop);
        for (;;) {
            ++iter;
            /* ResidualNorm without a residual computes b - A x (residual_norm.cpp:120-160) */
            memcpy(rt, b, sizeof(double) * N);
            RESID(rt);
            if (krylov_check(iter, max_iters, n, rt, tau0, reduction, 1, &stop, &one_changed, &tau)) break;
            oracle_csr_advanced_spmv_f64_i32(n, 1.0, row_ptrs, cols, vals, z, 1, -1.0, v, 1, 1);
            DOT(v, z, &alpha);
            for (int64_t i = 0; i < n; ++i) v[i] -= alpha * q[i];
            apply_precond(m, n, v, z_tilde);
            DOT(v, z_tilde, &beta);
            oracle_minres_step_1_f64(1, &alpha, &beta, &gamma, &delta, &cos_prev, &cosv, &sin_prev,
                                     &sinv, &eta, &eta_next, &tau_sq, &stop);
            { double* t_ = p; p = p_prev; p_prev = t_; }
            oracle_minres_step_2_f64(n, 1, 1, x, p, p_prev, z, z_tilde, q, q_prev, v, &alpha, &beta,
                                     &gamma, &delta, &cosv, &eta, &stop);
            { const double t_ = gamma; gamma = beta; beta = t_; }
        }
        /* the Convergence logger is handed r, which Minres never updates: it reports the
         * norm of the INITIAL residual (minres.cpp:196-198) */
        oracle_dense_compute_norm2_f64(n, 1, r, 1, &tau, 0);
    } else {
        iter = -2;
    }
#undef SPMV
#undef RESID
#undef DOT
#undef BASELINE
    if (resnorm_out) *resnorm_out = tau;
    free(w);
    return iter;
}

/* Ir (kind 5; core/solver/ir.cpp:189-255, inner solver = the preconditioner argument,
 * Identity if none, p0 = relaxation factor) and Chebyshev (kind 6;
 * core/solver/chebyshev.cpp:203-296, p0 / p1 = the foci) with the residual handling of
 * core/solver/update_residual.hpp:25-75: iteration 0 checks the initial residual; later
 * iterations first ask the criteria that need no residual (Iteration), then recompute
 * r = b - A x and check it.  The initial guess is the given x. */
static int64_t oracle_stationary_solve(int kind, int64_t n, const int32_t* row_ptrs,
                                       const int32_t* cols, const double* vals,
                                       const oracle_precond* m, const double* b, double* x,
                                       int64_t max_iters, double reduction, int baseline,
                                       double p0, double p1, double* resnorm_out)
{
    const size_t N = (size_t)n;
    double* w = (double*)calloc(N * 3, sizeof(double));
    double *r = w, *inner = w + N, *upd = w + 2 * N;
    double tau = 0.0, tau0 = 1.0;
    uint8_t stop = 0;
    int one_changed = 0;
    const double center = (p0 + p1) / 2.0, foci_direction = (p1 - p0) / 2.0;
    double alpha = 1.0 / center;
    double beta = 0.5 * (foci_direction * alpha) * (foci_direction * alpha);
    memcpy(r, b, sizeof(double) * N);
    oracle_csr_advanced_spmv_f64_i32(n, -1.0, row_ptrs, cols, vals, x, 1, 1.0, r, 1, 1);
    if (baseline == 0) {
        oracle_dense_compute_norm2_f64(n, 1, b, 1, &tau0, 0);
    } else if (baseline == 1) {
        oracle_dense_compute_norm2_f64(n, 1, r, 1, &tau0, 0);
    }
    int64_t iter = -1;
    for (;;) {
        ++iter;
        if (iter > 0) {
            if (iter >= max_iters) { /* Iteration, residual check ignored; not finalized */
                if ((stop & 0x3f) == 0) stop |= (uint8_t)1;
                /* the Convergence logger gets no residual here and computes
                 * ||b - A x|| itself (core/log/convergence.cpp) */
                memcpy(r, b, sizeof(double) * N);
                oracle_csr_advanced_spmv_f64_i32(n, -1.0, row_ptrs, cols, vals, x, 1, 1.0, r, 1, 1);
                oracle_dense_compute_norm2_f64(n, 1, r, 1, &tau, 0);
                break;
            }
            memcpy(r, b, sizeof(double) * N);
            oracle_csr_advanced_spmv_f64_i32(n, -1.0, row_ptrs, cols, vals, x, 1, 1.0, r, 1, 1);
        }
        if (krylov_check(iter, max_iters, n, r, tau0, reduction, 1, &stop, &one_changed, &tau)) break;
        if (kind == 5) {
            /* solver_->apply(relaxation, r, one, x): the ADVANCED apply of the inner
             * operator, in its own operation order (ir.cpp:252-253) */
            if (m->precond == 1) {
                oracle_jacobi_scalar_apply_f64(n, 1, m->inv_diag, 1, p0, r, 1, 1.0, x, 1);
            } else if (m->precond == 2) {
                oracle_jacobi_apply_f64_i32(m->num_blocks, m->block_offset, m->group_offset,
                                            m->group_power, m->block_ptrs, m->blocks, p0, r, 1,
                                            1.0, x, 1, 1);
            } else { /* Identity: x->scale(one); x->add_scaled(relaxation, r) */
                for (int64_t i = 0; i < n; ++i) x[i] = x[i] * 1.0 + p0 * r[i];
            }
        } else {
            apply_precond(m, n, r, inner);
            if (iter == 0) {
                oracle_chebyshev_init_update_f64(n, 1, 1, alpha, inner, upd, x);
                continue;
            }
            if (iter > 1) {
                beta = (foci_direction * alpha / 2.0) * (foci_direction * alpha / 2.0);
            }
            alpha = 1.0 / (center - beta / alpha);
            oracle_chebyshev_update_f64(n, 1, 1, alpha, beta, inner, upd, x);
        }
    }
    if (resnorm_out) *resnorm_out = tau;
    free(w);
    return iter;
}

/* ---- GMRES driver --------------------------------------------------------
 * core/solver/gmres.cpp:321-621 (Gmres::apply_dense_impl), one right-hand
 * side, non-flexible, f64 / int32; ortho: 0 = mgs (:157-177), 1 = cgs
 * (:223-251), 2 = cgs2 (:254-300).  Criterion: Combined(Iteration,
 * ResidualNorm(rhs_norm)) checked with the Givens residual-norm estimate
 * (updater.residual_norm(residual_norm), gmres.cpp:452-458).  Returns the
 * total iteration count. */
int64_t oracle_gmres_solve_f64_i32(int64_t n, const int32_t* row_ptrs,
                                   const int32_t* cols, const double* vals,
                                   const oracle_precond* m, const double* b,
                                   double* x, int64_t krylov_dim, int ortho,
                                   int64_t max_iters, double reduction,
                                   double* resnorm_out)
{
    const int64_t kd = krylov_dim;
    double* residual = (double*)malloc(sizeof(double) * (size_t)n * 4);
    double *precv = residual + n, *before = precv + n, *after = before + n;
    double* krylov = (double*)malloc(sizeof(double) * (size_t)n * (size_t)(kd + 1));
    double* hess = (double*)calloc((size_t)(kd * (kd + 1)), sizeof(double));
    double* haux = (double*)calloc((size_t)(kd + 1), sizeof(double));
    double* gsin = (double*)calloc((size_t)kd, sizeof(double));
    double* gcos = (double*)calloc((size_t)kd, sizeof(double));
    double* rnc = (double*)calloc((size_t)(kd + 1), sizeof(double));
    double* y = (double*)calloc((size_t)kd, sizeof(double));
    double residual_norm = 0, tau0 = 0;
    uint64_t final_iter = 0;
    uint8_t stop = 0;
    const int64_t hld = kd + 1; /* row stride of hessenberg (nrhs = 1) */

    oracle_gmres_initialize_f64(n, 1, b, 1, residual, 1, gsin, gcos, kd, &stop);
    oracle_csr_advanced_spmv_f64_i32(n, -1.0, row_ptrs, cols, vals, x, 1, 1.0, residual, 1, 1);
    oracle_dense_compute_norm2_f64(n, 1, residual, 1, &residual_norm, 0);
    oracle_gmres_restart_f64(n, 1, residual, 1, &residual_norm, rnc, krylov, 1, &final_iter);
    oracle_dense_compute_norm2_f64(n, 1, b, 1, &tau0, 0);

    int64_t total_iter = -1, restart_iter = 0;
    for (;;) {
        ++total_iter;
        int all_stopped = 0, one_changed = 0;
        if (total_iter >= max_iters) {
            if ((stop & 0x3f) == 0) stop |= 1; /* Iteration: id 1, setFinalized = false */
            all_stopped = 1;
        } else {
            all_stopped = oracle_residual_norm_f64(1, &residual_norm, &tau0, reduction, 2, 0,
                                                   &stop, 0, &one_changed);
        }
        if (all_stopped) break;
        if (restart_iter == kd) {
            oracle_gmres_solve_krylov_f64(1, rnc, hess, hld, y, &final_iter, &stop);
            oracle_gmres_multi_axpy_f64(n, 1, krylov, 1, y, 1, before, 1, &final_iter, &stop);
            apply_precond(m, n, before, after);
            { const double one = 1.0; oracle_dense_add_scaled_f64(n, 1, &one, 1, after, 1, x, 1, 0); }
            memcpy(residual, b, sizeof(double) * (size_t)n);
            oracle_csr_advanced_spmv_f64_i32(n, -1.0, row_ptrs, cols, vals, x, 1, 1.0, residual, 1, 1);
            oracle_dense_compute_norm2_f64(n, 1, residual, 1, &residual_norm, 0);
            oracle_gmres_restart_f64(n, 1, residual, 1, &residual_norm, rnc, krylov, 1, &final_iter);
            restart_iter = 0;
        }
        double* this_k = krylov + n * restart_iter;
        double* next_k = krylov + n * (restart_iter + 1);
        double* hiter = hess + restart_iter * hld; /* (restart_iter + 2) x 1 */
        apply_precond(m, n, this_k, precv);
        oracle_csr_spmv_f64_i32(n, row_ptrs, cols, vals, precv, 1, next_k, 1, 1);
        if (ortho == 0) {
            for (int64_t i = 0; i <= restart_iter; ++i) {
                oracle_dense_compute_dot_f64(n, 1, krylov + n * i, 1, next_k, 1, &hiter[i]);
                oracle_dense_add_scaled_f64(n, 1, &hiter[i], 1, krylov + n * i, 1, next_k, 1, 1);
            }
        } else {
            oracle_gmres_multi_dot_f64(n, 1, restart_iter + 1, krylov, 1, next_k, 1, hiter, 1);
            for (int64_t i = 0; i <= restart_iter; ++i) {
                oracle_dense_add_scaled_f64(n, 1, &hiter[i], 1, krylov + n * i, 1, next_k, 1, 1);
            }
            if (ortho == 2) {
                oracle_gmres_multi_dot_f64(n, 1, restart_iter + 1, krylov, 1, next_k, 1, haux, 1);
                for (int64_t i = 0; i <= restart_iter; ++i) {
                    oracle_dense_add_scaled_f64(n, 1, &haux[i], 1, krylov + n * i, 1, next_k, 1, 1);
                }
                /* hessenberg_iter->add_scaled(one, hessenberg_aux_iter): rows 0..restart_iter+1;
                 * row restart_iter+1 of both still holds stale data that is overwritten by the
                 * norm below, exactly as in the reference */
                for (int64_t i = 0; i <= restart_iter + 1; ++i) hiter[i] += 1.0 * haux[i];
            }
        }
        oracle_dense_compute_norm2_f64(n, 1, next_k, 1, &hiter[restart_iter + 1], 0);
        oracle_dense_inv_scale_f64(n, 1, &hiter[restart_iter + 1], 1, next_k, 1);
        oracle_gmres_hessenberg_qr_f64(1, gsin, gcos, &residual_norm, rnc, hiter, restart_iter,
                                       &final_iter, &stop);
        restart_iter++;
    }
    oracle_gmres_solve_krylov_f64(1, rnc, hess, hld, y, &final_iter, &stop);
    oracle_gmres_multi_axpy_f64(n, 1, krylov, 1, y, 1, before, 1, &final_iter, &stop);
    apply_precond(m, n, before, after);
    { const double one = 1.0; oracle_dense_add_scaled_f64(n, 1, &one, 1, after, 1, x, 1, 0); }
    if (resnorm_out) *resnorm_out = residual_norm;
    free(residual); free(krylov); free(hess); free(haux); free(gsin); free(gcos); free(rnc); free(y);
    return total_iter;
}
