// RcclCommunicator: the plug-in Ginkgo offers for the halo exchange of its distributed
// matrix - a gko::experimental::mpi::CollectiveCommunicator
// (include/ginkgo/core/distributed/collective_communicator.hpp:25-131), handed to
// distributed::Matrix through its RowGatherer template
// (include/ginkgo/core/distributed/matrix.hpp:481-492, core/distributed/matrix.cpp:40-54,375-380):
//
//     auto coll = std::make_shared<gko::cdna4::RcclCommunicator>(comm);
//     auto A = dist_mtx::create(exec, comm, gko::experimental::distributed::RowGatherer<int>::create(exec, coll));
//
// Device buffers (Ginkgo hands them over when its MPI is GPU-aware, mpi::requires_host_buffer,
// include/ginkgo/core/base/mpi.hpp) travel over RCCL / xGMI as grouped send/recv on the
// executor's stream (gkoc_comm_exchange_begin, csrc/comm.hip) and the returned request is
// already complete: the transfer is ordered on the stream the non-local SpMV runs on.  Host
// buffers (Ginkgo stages through the host when MPI is not GPU-aware, as with the MPICH of this
// image), executors other than HipExecutor, and process sets where two ranks share a GPU (RCCL
// refuses those) take the MPI path of DenseCommunicator (core/distributed/dense_communicator.cpp).
// Header-only; needs a Ginkgo built with GINKGO_BUILD_MPI and libgko_cdna4.so.
#pragma once
#include <cstring>
#include <memory>
#include <numeric>
#include <variant>
#include <vector>

#include <ginkgo/core/base/executor.hpp>
#include <ginkgo/core/base/mpi.hpp>
#include <ginkgo/core/distributed/collective_communicator.hpp>
#include <ginkgo/core/distributed/index_map.hpp>

#include "complex_abi.hpp"

// libginkgo_hip.so of this backend (gko_binding/fusion.cpp): launches kernels the binding holds back
extern "C" void gko_cdna4_launch_deferred();

namespace gko {
namespace cdna4 {

class RcclCommunicator final : public experimental::mpi::CollectiveCommunicator {
    using base_t = experimental::mpi::CollectiveCommunicator;
    using communicator = experimental::mpi::communicator;
    using request = experimental::mpi::request;
    using comm_index_type = experimental::distributed::comm_index_type;

public:
    using base_t::i_all_to_all_v;

    // one RCCL communicator per MPI communicator and process, shared by all patterns on it
    struct rccl_state {
        gkoc_comm_t comm = nullptr;
        ~rccl_state()
        {
            if (comm) gkoc_comm_destroy(comm);
        }
    };

    explicit RcclCommunicator(communicator base, int device_id = 0)
        : base_t(base), comm_(base), device_id_(device_id)
    {
        const auto n = base.get() == MPI_COMM_NULL ? 0 : base.size();
        recv_sizes_.assign(n, 0);
        send_sizes_.assign(n, 0);
        recv_offsets_.assign(n + 1, 0);
        send_offsets_.assign(n + 1, 0);
        if (n > 0) state_ = connect(base, device_id);
    }

    template <typename LocalIndexType, typename GlobalIndexType>
    RcclCommunicator(communicator base,
                     const experimental::distributed::index_map<LocalIndexType, GlobalIndexType>& imap,
                     std::shared_ptr<rccl_state> state, int device_id)
        : base_t(base), comm_(base), device_id_(device_id), state_(std::move(state))
    {
        const auto n = base.size();
        recv_sizes_.assign(n, 0);
        send_sizes_.assign(n, 0);
        recv_offsets_.assign(n + 1, 0);
        send_offsets_.assign(n + 1, 0);
        auto exec = imap.get_executor();
        if (!exec) return;
        auto host = exec->get_master();
        // how many values every other rank owes us: one segment of the map per neighbour
        auto targets = make_temporary_clone(host, &imap.get_remote_target_ids());
        auto offsets = make_temporary_clone(host, &imap.get_remote_global_idxs().get_offsets());
        for (size_type seg = 0; seg < imap.get_remote_global_idxs().get_segment_count(); ++seg) {
            recv_sizes_[targets->get_const_data()[seg]] =
                offsets->get_const_data()[seg + 1] - offsets->get_const_data()[seg];
        }
        comm_.all_to_all(host, recv_sizes_.data(), 1, send_sizes_.data(), 1);
        std::partial_sum(send_sizes_.begin(), send_sizes_.end(), send_offsets_.begin() + 1);
        std::partial_sum(recv_sizes_.begin(), recv_sizes_.end(), recv_offsets_.begin() + 1);
    }

    std::unique_ptr<base_t> create_with_same_type(communicator base, index_map_ptr imap) const override
    {
        auto state = state_;
        const int dev = device_id_;
        return std::visit(
            [&](const auto* map) -> std::unique_ptr<base_t> {
                return std::make_unique<RcclCommunicator>(base, *map, state, dev);
            },
            imap);
    }

    std::unique_ptr<base_t> create_inverse() const override
    {
        auto inv = std::make_unique<RcclCommunicator>(*this);
        std::swap(inv->send_sizes_, inv->recv_sizes_);
        std::swap(inv->send_offsets_, inv->recv_offsets_);
        return inv;
    }

    comm_index_type get_recv_size() const override { return recv_offsets_.back(); }

    comm_index_type get_send_size() const override { return send_offsets_.back(); }

    // true if device buffers go over RCCL (false: every exchange takes the MPI path)
    bool uses_rccl() const { return state_ && state_->comm; }

    RcclCommunicator(const RcclCommunicator&) = default;

protected:
    request i_all_to_all_v_impl(std::shared_ptr<const Executor> exec, const void* send_buffer,
                                MPI_Datatype send_type, void* recv_buffer,
                                MPI_Datatype recv_type) const override
    {
        auto hip = std::dynamic_pointer_cast<const HipExecutor>(exec);
        int send_bytes = 0, recv_bytes = 0;
        MPI_Type_size(send_type, &send_bytes);
        MPI_Type_size(recv_type, &recv_bytes);
        if (hip && uses_rccl() && send_bytes == recv_bytes) {
            std::vector<int64_t> sc(send_sizes_.begin(), send_sizes_.end());
            std::vector<int64_t> sd(send_offsets_.begin(), send_offsets_.end() - 1);
            std::vector<int64_t> rc(recv_sizes_.begin(), recv_sizes_.end());
            gko_cdna4_launch_deferred();
            auto stream = reinterpret_cast<gkoc_stream_t>(hip->get_stream());
            const int status = gkoc_comm_exchange_begin(state_->comm, stream, nullptr, send_buffer, sc.data(),
                                                        sd.data(), recv_buffer, rc.data(),
                                                        static_cast<size_t>(send_bytes));
            if (status != 0) {
                throw ::gko::Error(__FILE__, __LINE__, std::string("RCCL exchange: ") + gkoc_last_error());
            }
            return {};   // complete in stream order; request::wait() has nothing to wait for
        }
        return comm_.i_all_to_all_v(exec, send_buffer, send_sizes_.data(), send_offsets_.data(), send_type,
                                    recv_buffer, recv_sizes_.data(), recv_offsets_.data(), recv_type);
    }

private:
    // RCCL communicator over the ranks of `base`; none if two ranks use the same GPU of a host
    // or RCCL cannot be loaded
    static std::shared_ptr<rccl_state> connect(const communicator& base, int device_id)
    {
        auto state = std::make_shared<rccl_state>();
        const int n = base.size(), rank = base.rank();
        char name[MPI_MAX_PROCESSOR_NAME + 16] = {};
        int len = 0;
        MPI_Get_processor_name(name, &len);
        std::snprintf(name + len, 16, "#%d", device_id);
        std::vector<char> all(size_t(n) * sizeof(name));
        MPI_Allgather(name, sizeof(name), MPI_CHAR, all.data(), sizeof(name), MPI_CHAR, base.get());
        int usable = gkoc_comm_load_rccl(nullptr) == 0 ? 1 : 0;
        for (int a = 0; a < n && usable; ++a) {
            for (int b = a + 1; b < n; ++b) {
                if (std::strcmp(&all[a * sizeof(name)], &all[b * sizeof(name)]) == 0) usable = 0;
            }
        }
        MPI_Allreduce(MPI_IN_PLACE, &usable, 1, MPI_INT, MPI_MIN, base.get());
        if (!usable) return state;
        char id[GKOC_COMM_ID_BYTES] = {};
        int ok = 1;
        if (rank == 0) ok = gkoc_comm_unique_id(id) == 0;
        MPI_Bcast(&ok, 1, MPI_INT, 0, base.get());
        if (!ok) return state;
        MPI_Bcast(id, GKOC_COMM_ID_BYTES, MPI_CHAR, 0, base.get());
        gkoc_set_device(device_id);
        int created = gkoc_comm_create(&state->comm, n, rank, id) == 0 ? 1 : 0;
        MPI_Allreduce(MPI_IN_PLACE, &created, 1, MPI_INT, MPI_MIN, base.get());
        if (!created && state->comm) {
            gkoc_comm_destroy(state->comm);
            state->comm = nullptr;
        }
        return state;
    }

    communicator comm_;
    int device_id_;
    std::shared_ptr<rccl_state> state_;
    std::vector<comm_index_type> recv_sizes_, recv_offsets_, send_sizes_, send_offsets_;
};

}  // namespace cdna4
}  // namespace gko
