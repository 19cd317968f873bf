// gko::kernels::hip::{distributed_matrix, distributed_vector, index_map, partition,
// partition_helpers, assembly}: the set-up kernels of Ginkgo's distributed classes, forwarded to
// the C ABI (csrc/dist_setup.hip).  With them experimental::distributed::Partition, index_map,
// Matrix::read_distributed, Vector::read_distributed and assemble_rows_from_neighbors work on a
// HipExecutor without a detour over the host.
//   declarations: core/distributed/{matrix,vector,index_map,partition,partition_helpers,
//                 assembly}_kernels.hpp
// All value types (float, double and their complex counterparts): values are only moved.
#include <complex>

#include <ginkgo/core/base/device_matrix_data.hpp>
#include <ginkgo/core/distributed/index_map.hpp>
#include <ginkgo/core/distributed/partition.hpp>

#include "core/base/segmented_array.hpp"
#include "core/distributed/assembly_kernels.hpp"
#include "core/distributed/device_partition.hpp"
#include "core/distributed/index_map_kernels.hpp"
#include "core/distributed/matrix_kernels.hpp"
#include "core/distributed/partition_helpers_kernels.hpp"
#include "core/distributed/partition_kernels.hpp"
#include "core/distributed/vector_kernels.hpp"
#include "shim_common.hpp"

namespace gko {
namespace kernels {
namespace hip {

using cdna4::stream_of;
using exec_t = std::shared_ptr<const HipExecutor>;
using experimental::distributed::comm_index_type;
using experimental::distributed::Partition;

namespace {

template <typename L, typename G>
gkoc_partition describe(const Partition<L, G>* p)
{
    gkoc_partition d;
    d.num_ranges = static_cast<int64_t>(p->get_num_ranges());
    d.num_parts = p->get_num_parts();
    d.range_bounds = p->get_range_bounds();
    d.part_ids = p->get_part_ids();
    d.range_starting_indices = p->get_range_starting_indices();
    d.part_sizes = p->get_part_sizes();
    return d;
}

int index_space_id(experimental::distributed::index_space is)
{
    using experimental::distributed::index_space;
    return is == index_space::local ? 0 : is == index_space::non_local ? 1 : 2;
}

}  // namespace

#define FOR_LG(M) M(int32, i32, int32, i32) M(int32, i32, int64, i64) M(int64, i64, int64, i64)
#define FOR_G(M) M(int32, i32) M(int64, i64)
#define FOR_V(M, L, LN, G, GN)                                           \
    M(float, L, LN, G, GN) M(double, L, LN, G, GN) M(std::complex<float>, L, LN, G, GN) \
        M(std::complex<double>, L, LN, G, GN)


// the scratch state of a count / fill pair goes back if the resizes between them throw
struct state_guard {
    gkoc_stream_t s;
    void* state;
    int (*give_back)(gkoc_stream_t, void*);
    ~state_guard()
    {
        if (state) give_back(s, state);
    }
};

// ====================================================================== distributed_matrix
namespace distributed_matrix {

#define DEF(V, L, LN, G, GN)                                                                        \
    template <>                                                                                     \
    void separate_local_nonlocal<V, L, G>(                                                          \
        exec_t exec, const device_matrix_data<V, G>& input, const Partition<L, G>* row_partition,   \
        const Partition<L, G>* col_partition, comm_index_type local_part, array<L>& local_row_idxs, \
        array<L>& local_col_idxs, array<V>& local_values, array<L>& non_local_row_idxs,             \
        array<G>& non_local_col_idxs, array<V>& non_local_values)                                   \
    {                                                                                               \
        const auto nnz = static_cast<int64_t>(input.get_num_stored_elements());                     \
        const auto rp = describe(row_partition);                                                    \
        const auto cp = describe(col_partition);                                                    \
        void* state = nullptr;                                                                      \
        int64_t nl = 0, nn = 0;                                                                     \
        GKOC_CALL(gkoc_dist_separate_local_nonlocal_count_##LN##_##GN(                              \
            stream_of(exec), nnz, input.get_const_row_idxs(), input.get_const_col_idxs(), &rp, &cp, \
            local_part, &state, &nl, &nn));                                                         \
        state_guard guard{stream_of(exec), state, gkoc_dist_separate_state_free};                   \
        local_row_idxs.resize_and_reset(nl);                                                        \
        local_col_idxs.resize_and_reset(nl);                                                        \
        local_values.resize_and_reset(nl);                                                          \
        non_local_row_idxs.resize_and_reset(nn);                                                    \
        non_local_col_idxs.resize_and_reset(nn);                                                    \
        non_local_values.resize_and_reset(nn);                                                      \
        guard.state = nullptr; /* the fill call frees it, also when it fails */                     \
        GKOC_CALL(gkoc_dist_separate_local_nonlocal_fill_##LN##_##GN(                               \
            stream_of(exec), nnz, input.get_const_row_idxs(), input.get_const_col_idxs(),           \
            input.get_const_values(), sizeof(V), &rp, &cp, state, local_row_idxs.get_data(),        \
            local_col_idxs.get_data(), local_values.get_data(), non_local_row_idxs.get_data(),      \
            non_local_col_idxs.get_data(), non_local_values.get_data()));                           \
    }
#define DEF_LG(L, LN, G, GN) FOR_V(DEF, L, LN, G, GN)
FOR_LG(DEF_LG)
#undef DEF_LG
#undef DEF

}  // namespace distributed_matrix


// ====================================================================== distributed_vector
namespace distributed_vector {

#define DEF(V, L, LN, G, GN)                                                                        \
    template <>                                                                                     \
    void build_local<V, L, G>(exec_t exec, const device_matrix_data<V, G>& input,                   \
                              const Partition<L, G>* partition, comm_index_type local_part,         \
                              matrix::Dense<V>* local_mtx)                                          \
    {                                                                                               \
        const auto p = describe(partition);                                                         \
        GKOC_CALL(gkoc_dist_vector_build_local_##LN##_##GN(                                         \
            stream_of(exec), static_cast<int64_t>(input.get_num_stored_elements()),                 \
            input.get_const_row_idxs(), input.get_const_col_idxs(), input.get_const_values(),       \
            sizeof(V), &p, local_part, local_mtx->get_values(),                                     \
            static_cast<int64_t>(local_mtx->get_stride())));                                        \
    }
#define DEF_LG(L, LN, G, GN) FOR_V(DEF, L, LN, G, GN)
FOR_LG(DEF_LG)
#undef DEF_LG
#undef DEF

}  // namespace distributed_vector


// ====================================================================== index_map
namespace index_map {

#define DEF(L, LN, G, GN)                                                                           \
    template <>                                                                                     \
    void build_mapping<L, G>(exec_t exec, const Partition<L, G>* part,                              \
                             const array<G>& recv_connections, array<comm_index_type>& part_ids,    \
                             array<L>& remote_local_idxs, array<G>& remote_global_idxs,             \
                             array<int64>& remote_sizes)                                            \
    {                                                                                               \
        const auto p = describe(part);                                                              \
        void* state = nullptr;                                                                      \
        int64_t nu = 0, np = 0;                                                                     \
        GKOC_CALL(gkoc_index_map_build_mapping_count_##LN##_##GN(                                   \
            stream_of(exec), static_cast<int64_t>(recv_connections.get_size()),                     \
            recv_connections.get_const_data(), &p, &state, &nu, &np));                              \
        state_guard guard{stream_of(exec), state, gkoc_index_map_mapping_state_free};               \
        remote_global_idxs.resize_and_reset(nu);                                                    \
        remote_local_idxs.resize_and_reset(nu);                                                     \
        part_ids.resize_and_reset(np);                                                              \
        remote_sizes.resize_and_reset(np);                                                          \
        guard.state = nullptr; /* the fill call frees it, also when it fails */                     \
        GKOC_CALL(gkoc_index_map_build_mapping_fill_##LN##_##GN(                                    \
            stream_of(exec), &p, state, part_ids.get_data(), remote_local_idxs.get_data(),          \
            remote_global_idxs.get_data(), remote_sizes.get_data()));                               \
    }                                                                                               \
    template <>                                                                                     \
    void map_to_local<L, G>(exec_t exec, const Partition<L, G>* partition,                          \
                            const array<comm_index_type>& remote_target_ids,                        \
                            device_segmented_array<const G> remote_global_idxs,                     \
                            comm_index_type rank, const array<G>& global_ids,                       \
                            experimental::distributed::index_space is, array<L>& local_ids)         \
    {                                                                                               \
        const auto p = describe(partition);                                                         \
        local_ids.resize_and_reset(global_ids.get_size());                                          \
        GKOC_CALL(gkoc_index_map_map_to_local_##LN##_##GN(                                          \
            stream_of(exec), static_cast<int64_t>(global_ids.get_size()), global_ids.get_const_data(), \
            &p, static_cast<int64_t>(remote_target_ids.get_size()),                                 \
            remote_target_ids.get_const_data(), remote_global_idxs.flat_begin,                      \
            remote_global_idxs.offsets_begin, rank, index_space_id(is), local_ids.get_data()));     \
    }                                                                                               \
    template <>                                                                                     \
    void map_to_global<L, G>(exec_t exec, device_partition<const L, const G> partition,             \
                             device_segmented_array<const G> remote_global_idxs,                    \
                             comm_index_type rank, const array<L>& local_idxs,                      \
                             experimental::distributed::index_space is, array<G>& global_idxs)      \
    {                                                                                               \
        global_idxs.resize_and_reset(local_idxs.get_size());                                        \
        /* the ranges of this rank: two host reads of the (device) segment offsets and its size */  \
        int64 seg[2] = {0, 0};                                                                      \
        exec->get_master()->copy_from(exec.get(), 2, partition.ranges_by_part.offsets_begin + rank, \
                                      seg);                                                         \
        L local_size{};                                                                             \
        exec->get_master()->copy_from(exec.get(), 1, partition.part_sizes_begin + rank,             \
                                      &local_size);                                                 \
        static_assert(sizeof(size_type) == sizeof(uint64_t), "size_type must be 64 bits");          \
        GKOC_CALL(gkoc_index_map_map_to_global_##LN##_##GN(                                         \
            stream_of(exec), static_cast<int64_t>(local_idxs.get_size()),                           \
            local_idxs.get_const_data(), partition.offsets_begin, partition.starting_indices_begin, \
            static_cast<int64_t>(local_size),                                                       \
            reinterpret_cast<const uint64_t*>(partition.ranges_by_part.flat_begin) + seg[0],        \
            seg[1] - seg[0], remote_global_idxs.flat_begin,                                         \
            static_cast<int64_t>(remote_global_idxs.flat_end - remote_global_idxs.flat_begin),      \
            index_space_id(is), global_idxs.get_data()));                                           \
    }
FOR_LG(DEF)
#undef DEF

}  // namespace index_map


// ====================================================================== partition
namespace partition {

void count_ranges(exec_t exec, const array<comm_index_type>& mapping, size_type& num_ranges)
{
    int64_t n = 0;
    GKOC_CALL(gkoc_partition_count_ranges(stream_of(exec), static_cast<int64_t>(mapping.get_size()),
                                          mapping.get_const_data(), &n));
    num_ranges = static_cast<size_type>(n);
}

void build_ranges_by_part(exec_t exec, const int* range_parts, size_type num_ranges, int num_parts,
                          array<size_type>& range_ids, array<int64>& sizes)
{
    range_ids.resize_and_reset(num_ranges);
    sizes.resize_and_reset(num_parts);
    GKOC_CALL(gkoc_partition_build_ranges_by_part(
        stream_of(exec), range_parts, static_cast<int64_t>(num_ranges), num_parts,
        reinterpret_cast<uint64_t*>(range_ids.get_data()), sizes.get_data()));
}

#define DEF(G, GN)                                                                                  \
    template <>                                                                                     \
    void build_from_contiguous<G>(exec_t exec, const array<G>& ranges,                              \
                                  const array<comm_index_type>& part_id_mapping, G* range_bounds,   \
                                  comm_index_type* part_ids)                                        \
    {                                                                                               \
        GKOC_CALL(gkoc_partition_build_from_contiguous_##GN(                                        \
            stream_of(exec), static_cast<int64_t>(ranges.get_size()) - 1, ranges.get_const_data(),  \
            part_id_mapping.get_size() > 0 ? part_id_mapping.get_const_data() : nullptr,            \
            range_bounds, part_ids));                                                               \
    }                                                                                               \
    template <>                                                                                     \
    void build_from_mapping<G>(exec_t exec, const array<comm_index_type>& mapping, G* range_bounds, \
                               comm_index_type* part_ids)                                           \
    {                                                                                               \
        GKOC_CALL(gkoc_partition_build_from_mapping_##GN(                                           \
            stream_of(exec), static_cast<int64_t>(mapping.get_size()), mapping.get_const_data(),    \
            range_bounds, part_ids));                                                               \
    }                                                                                               \
    template <>                                                                                     \
    void build_ranges_from_global_size<G>(exec_t exec, comm_index_type num_parts, G global_size,    \
                                          array<G>& ranges)                                         \
    {                                                                                               \
        GKOC_CALL(gkoc_partition_build_ranges_from_global_size_##GN(stream_of(exec), num_parts,     \
                                                                    global_size, ranges.get_data())); \
    }
FOR_G(DEF)
#undef DEF

#define DEF(L, LN, G, GN)                                                                           \
    template <>                                                                                     \
    void build_starting_indices<L, G>(exec_t exec, const G* range_offsets, const int* range_parts,  \
                                      size_type num_ranges, comm_index_type num_parts,              \
                                      comm_index_type& num_empty_parts, L* ranks, L* sizes)         \
    {                                                                                               \
        int32_t empty = 0;                                                                          \
        GKOC_CALL(gkoc_partition_build_starting_indices_##LN##_##GN(                                \
            stream_of(exec), range_offsets, range_parts, static_cast<int64_t>(num_ranges),          \
            num_parts, &empty, ranks, sizes));                                                      \
        num_empty_parts = empty;                                                                    \
    }                                                                                               \
    template <>                                                                                     \
    void has_ordered_parts<L, G>(exec_t exec, const Partition<L, G>* partition, bool* result)       \
    {                                                                                               \
        int ok = 1;                                                                                 \
        GKOC_CALL(gkoc_partition_has_ordered_parts(stream_of(exec),                                 \
                                                   static_cast<int64_t>(partition->get_num_ranges()), \
                                                   partition->get_part_ids(), &ok));                \
        *result = ok != 0;                                                                          \
    }
FOR_LG(DEF)
#undef DEF

}  // namespace partition


// ====================================================================== partition_helpers
namespace partition_helpers {

#define DEF(G, GN)                                                                                  \
    template <>                                                                                     \
    void sort_by_range_start<G>(exec_t exec, array<G>& range_start_ends,                            \
                                array<comm_index_type>& part_ids)                                   \
    {                                                                                               \
        GKOC_CALL(gkoc_partition_helpers_sort_by_range_start_##GN(                                  \
            stream_of(exec), static_cast<int64_t>(part_ids.get_size()), range_start_ends.get_data(), \
            part_ids.get_data()));                                                                  \
    }                                                                                               \
    template <>                                                                                     \
    void check_consecutive_ranges<G>(exec_t exec, const array<G>& range_start_ends, bool& result)   \
    {                                                                                               \
        int ok = 1;                                                                                 \
        GKOC_CALL(gkoc_partition_helpers_check_consecutive_ranges_##GN(                             \
            stream_of(exec), static_cast<int64_t>(range_start_ends.get_size() / 2),                 \
            range_start_ends.get_const_data(), &ok));                                               \
        result = ok != 0;                                                                           \
    }                                                                                               \
    template <>                                                                                     \
    void compress_ranges<G>(exec_t exec, const array<G>& range_start_ends, array<G>& range_offsets) \
    {                                                                                               \
        GKOC_CALL(gkoc_partition_helpers_compress_ranges_##GN(                                      \
            stream_of(exec), static_cast<int64_t>(range_offsets.get_size()),                        \
            range_start_ends.get_const_data(), range_offsets.get_data()));                          \
    }
FOR_G(DEF)
#undef DEF

}  // namespace partition_helpers


// ====================================================================== assembly
namespace assembly {

#define DEF(V, L, LN, G, GN)                                                                        \
    template <>                                                                                     \
    void count_non_owning_entries<V, L, G>(                                                         \
        exec_t exec, const device_matrix_data<V, G>& input, const Partition<L, G>* row_partition,   \
        comm_index_type local_part, array<comm_index_type>& send_count,                             \
        array<G>& send_positions, array<G>& original_positions)                                     \
    {                                                                                               \
        const auto p = describe(row_partition);                                                     \
        GKOC_CALL(gkoc_assembly_count_non_owning_entries_##LN##_##GN(                               \
            stream_of(exec), static_cast<int64_t>(input.get_num_stored_elements()),                 \
            input.get_const_row_idxs(), &p, local_part, send_count.get_data(),                      \
            send_positions.get_data(), original_positions.get_data()));                             \
    }                                                                                               \
    template <>                                                                                     \
    void fill_send_buffers<V, L, G>(                                                                \
        exec_t exec, const device_matrix_data<V, G>& input, const Partition<L, G>* row_partition,   \
        comm_index_type local_part, const array<G>& send_positions,                                 \
        const array<G>& original_positions, array<G>& send_row_idxs, array<G>& send_col_idxs,       \
        array<V>& send_values)                                                                      \
    {                                                                                               \
        GKOC_CALL(gkoc_assembly_fill_send_buffers_##GN(                                             \
            stream_of(exec), static_cast<int64_t>(input.get_num_stored_elements()),                 \
            input.get_const_row_idxs(), input.get_const_col_idxs(), input.get_const_values(),       \
            sizeof(V), send_positions.get_const_data(), original_positions.get_const_data(),        \
            send_row_idxs.get_data(), send_col_idxs.get_data(), send_values.get_data()));           \
    }
#define DEF_LG(L, LN, G, GN) FOR_V(DEF, L, LN, G, GN)
FOR_LG(DEF_LG)
#undef DEF_LG
#undef DEF

}  // namespace assembly


}  // namespace hip
}  // namespace kernels
}  // namespace gko
