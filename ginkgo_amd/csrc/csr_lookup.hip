// csr::build_lookup_offsets / csr::build_lookup (core/matrix/csr_kernels.hpp; the format is defined in
// core/matrix/csr_lookup.hpp:26-85, the reference's construction in reference/matrix/
// csr_kernels.cpp:1425-1573): per row of a CSR matrix with sorted columns a 64-bit descriptor and
// a piece of int32 storage that let a kernel find the position of (row, col) without a search -
//   full   (1): the row holds every column of [min_col, min_col + len): position = col - min_col
//   bitmap (2): one 32-bit mask per block of 32 columns of the row's range + the number of entries in
//               front of each block (descriptor = blocks << 32 | 2; storage = [ranks | masks])
//   hash   (4): open addressing with linear probing over max(2 len, 1) slots, slot = entry number,
//               hash(col) = (col * p) mod slots with p = 1 | floor(slots * 0.61803398875)
//               (descriptor = p << 32 | 4); entries are inserted in row order, so the table is a
//               function of the row alone
//   none   (0): nothing stored.
// `allowed` is the bit set of kinds the caller accepts.  Ginkgo's factorisations are the consumers
// (out of scope here); the binding used to hand out all-zero descriptors, which a
// consumer would have read as "full rows of length zero" (ADVICE round 3).  Set-up code: one lane
// per row, rows are walked sequentially exactly as the reference walks them - the tables are
// bit-identical by construction.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "scan.hpp"

namespace gkoc {
namespace {

constexpr int lookup_full = 1, lookup_bitmap = 2, lookup_hash = 4;
constexpr int bitmap_block = 32;

template <typename I>
__device__ __forceinline__ void row_shape(const I* __restrict__ row_ptrs, const I* __restrict__ cols,
                                          int64_t row, I& begin, I& len, I& min_col, I& range)
{
    begin = row_ptrs[row];
    len = row_ptrs[row + 1] - begin;
    min_col = len > 0 ? cols[begin] : I(0);
    range = len > 0 ? I(cols[begin + len - 1] - min_col + 1) : I(0);
}

template <typename I>
__global__ __launch_bounds__(256) void lookup_sizes_kernel(int64_t n_rows, const I* __restrict__ row_ptrs,
                                                           const I* __restrict__ cols, int allowed,
                                                           I* __restrict__ sizes)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row > n_rows) return;
    if (row == n_rows) {
        sizes[row] = 0;
        return;
    }
    I begin, len, min_col, range;
    row_shape(row_ptrs, cols, row, begin, len, min_col, range);
    I need = 0;
    if (!((allowed & lookup_full) && len == range)) {
        const I hash_slots = 2 * len > 1 ? 2 * len : I(1);
        const I bitmap_words = 2 * I((int64_t(range) + bitmap_block - 1) / bitmap_block);
        if ((allowed & lookup_bitmap) && bitmap_words <= hash_slots) {
            need = bitmap_words;
        } else if (allowed & lookup_hash) {
            need = hash_slots;
        }
    }
    sizes[row] = need;
}

template <typename I>
__global__ __launch_bounds__(256) void lookup_build_kernel(int64_t n_rows, const I* __restrict__ row_ptrs,
                                                           const I* __restrict__ cols, int allowed,
                                                           const I* __restrict__ offsets,
                                                           int64_t* __restrict__ desc,
                                                           int32_t* __restrict__ storage)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_rows) return;
    I begin, len, min_col, range;
    row_shape(row_ptrs, cols, row, begin, len, min_col, range);
    const I avail = offsets[row + 1] - offsets[row];
    int32_t* __restrict__ st = storage + offsets[row];
    const I* __restrict__ c = cols + begin;
    if ((allowed & lookup_full) && len == range) {
        desc[row] = lookup_full;
        return;
    }
    const int32_t blocks = int32_t((int64_t(range) + bitmap_block - 1) / bitmap_block);
    if ((allowed & lookup_bitmap) && I(2) * I(blocks) <= avail) {
        desc[row] = (int64_t(blocks) << 32) | int64_t(lookup_bitmap);
        int32_t* ranks = st;
        uint32_t* masks = reinterpret_cast<uint32_t*>(st + blocks);
        for (int32_t b = 0; b < blocks; ++b) masks[b] = 0u;
        for (I k = 0; k < len; ++k) {
            const I rel = c[k] - min_col;
            masks[rel / bitmap_block] |= uint32_t(1) << (rel % bitmap_block);
        }
        int32_t seen = 0;
        for (int32_t b = 0; b < blocks; ++b) {
            ranks[b] = seen;
            seen += __popc(masks[b]);
        }
        return;
    }
    if (allowed & lookup_hash) {
        const uint32_t param = 1u | uint32_t(double(avail) * 0.61803398875);
        desc[row] = (int64_t(param) << 32) | int64_t(lookup_hash);
        for (I t = 0; t < avail; ++t) st[t] = -1;
        using U = typename std::make_unsigned<I>::type;
        for (int32_t k = 0; k < int32_t(len); ++k) {
            // (the product is formed in the index type's unsigned arithmetic, as the reference does)
            U h = (U(c[k]) * U(param)) % U(uint32_t(avail));
            while (st[h] != -1) {
                ++h;
                if (h >= U(avail)) h = 0;
            }
            st[h] = k;
        }
        return;
    }
    desc[row] = 0;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_LOOKUP(I, IN)                                                                          \
    extern "C" int gkoc_csr_build_lookup_offsets_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, \
                                                      const I* col_idxs, int allowed, I* storage_offsets) \
    {                                                                                                   \
        GKOC_REQUIRE(n_rows >= 0 && storage_offsets, GKOC_E_INVALID, "bad argument");                   \
        GKOC_REQUIRE(n_rows == 0 || (row_ptrs && col_idxs), GKOC_E_INVALID, "null pointer");            \
        lookup_sizes_kernel<I><<<dim3(unsigned(ceildiv(n_rows + 1, 256))), dim3(256), 0, as_stream(s)>>>( \
            n_rows, row_ptrs, col_idxs, allowed, storage_offsets);                                      \
        GKOC_LAUNCH_OK();                                                                               \
        return device_exclusive_scan<I>(as_stream(s), storage_offsets, n_rows + 1);                     \
    }                                                                                                   \
    extern "C" int gkoc_csr_build_lookup_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,       \
                                              const I* col_idxs, int allowed, const I* storage_offsets, \
                                              int64_t* row_desc, int32_t* storage)                      \
    {                                                                                                   \
        GKOC_REQUIRE(n_rows >= 0, GKOC_E_INVALID, "bad argument");                                      \
        if (n_rows == 0) return GKOC_OK;                                                                \
        GKOC_REQUIRE(row_ptrs && col_idxs && storage_offsets && row_desc, GKOC_E_INVALID, "null pointer"); \
        lookup_build_kernel<I><<<dim3(unsigned(ceildiv(n_rows, 256))), dim3(256), 0, as_stream(s)>>>(   \
            n_rows, row_ptrs, col_idxs, allowed, storage_offsets, row_desc, storage);                   \
        GKOC_LAUNCH_OK();                                                                               \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_LOOKUP(int32_t, i32)
GKOC_DEF_LOOKUP(int64_t, i64)
