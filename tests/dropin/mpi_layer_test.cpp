// The MPI layer (ginkgo_amd/gko_binding/mpi_rccl.cpp) by itself: the collectives Ginkgo's distributed
// classes issue (include/ginkgo/core/base/mpi.hpp: all_reduce :838, all_to_all :1184, i_all_to_all_v :1441,
// the neighbourhood form of core/distributed/neighborhood_communicator.cpp), handed DEVICE buffers, against
// what MPI defines for them.  Ranks that share a GPU talk through the library's mailbox transport
// (gkoc_comm_ipc_*), so the device route - including the nonblocking agreement that travels in the request -
// runs on a one-GPU box.   mpiexec -n <ranks> ./mpi_layer_test
#include <mpi.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gko_cdna4.h"

static int failures = 0;
#define CHECK(cond, what)                                                        \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::printf("FAILED (rank %d): %s\n", rank, what);                   \
            ++failures;                                                          \
        }                                                                        \
    } while (0)

template <typename T>
struct dev_array {
    T* p = nullptr;
    size_t n = 0;
    explicit dev_array(size_t n_) : n(n_)
    {
        void* q = nullptr;
        if (gkoc_malloc(&q, (n ? n : 1) * sizeof(T)) != 0) std::abort();
        p = static_cast<T*>(q);
    }
    ~dev_array() { gkoc_free(p); }
    void put(const std::vector<T>& h) { gkoc_memcpy_h2d(p, h.data(), n * sizeof(T), nullptr); }
    std::vector<T> get() const
    {
        std::vector<T> h(n);
        gkoc_memcpy_d2h(h.data(), p, n * sizeof(T), nullptr);
        return h;
    }
};

int main(int argc, char** argv)
{
    MPI_Init(&argc, &argv);
    int rank = 0, size = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    gkoc_set_device(0);
    double t_last = MPI_Wtime();
    auto lap = [&](const char* what) {
        const double t = MPI_Wtime();
        if (rank == 0 && std::getenv("GKOC_MPI_VERBOSE")) std::fprintf(stderr, "[mpi_layer_test] %-40s %8.3f ms\n", what, (t - t_last) * 1e3);
        t_last = t;
    };

    // all_reduce: in place and out of place, doubles
    {
        dev_array<double> a(3), b(3);
        a.put({rank + 1.0, 0.5 * (rank + 1), -2.0 * rank});
        MPI_Allreduce(a.p, b.p, 3, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
        const double tot = size * (size + 1) / 2.0;
        auto h = b.get();
        CHECK(h[0] == tot && h[1] == 0.5 * tot && h[2] == -2.0 * (tot - size), "all-reduce, out of place");
        MPI_Allreduce(MPI_IN_PLACE, a.p, 3, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
        h = a.get();
        CHECK(h[0] == tot && h[1] == 0.5 * tot, "all-reduce, in place");
    }
    lap("all-reduce (incl. bringing the communicator up)");
    // all_to_all of one int per rank (the sizes of read_distributed)
    {
        dev_array<int> s(size), r(size);
        std::vector<int> hs(size);
        for (int p = 0; p < size; ++p) hs[p] = 100 * rank + p;
        s.put(hs);
        MPI_Alltoall(s.p, 1, MPI_INT, r.p, 1, MPI_INT, MPI_COMM_WORLD);
        auto h = r.get();
        bool ok = true;
        for (int p = 0; p < size; ++p) ok = ok && h[p] == 100 * p + rank;
        CHECK(ok, "all-to-all of one int per rank");
    }
    lap("all-to-all");
    // all_to_all_v, blocking and nonblocking, uneven counts with gaps, some of them zero
    for (int nonblocking = 0; nonblocking < 3; ++nonblocking) {
        std::vector<int> sc(size), sd(size), rc(size), rd(size);
        int spos = 0, rpos = 0;
        for (int p = 0; p < size; ++p) {
            sc[p] = ((rank + 2 * p) % 3 == 0) ? 0 : 5 + rank + 3 * p;       // rank -> p
            rc[p] = ((p + 2 * rank) % 3 == 0) ? 0 : 5 + p + 3 * rank;       // p -> rank
            sd[p] = spos + 2;
            rd[p] = rpos + 1;
            spos = sd[p] + sc[p];
            rpos = rd[p] + rc[p];
        }
        dev_array<double> s(spos + 4), r(rpos + 4);
        std::vector<double> hs(spos + 4, -1.0);
        for (int p = 0; p < size; ++p) {
            for (int i = 0; i < sc[p]; ++i) hs[sd[p] + i] = 1000.0 * rank + 10.0 * p + 0.001 * i + 1e5 * nonblocking;
        }
        s.put(hs);
        r.put(std::vector<double>(rpos + 4, -7.0));
        if (nonblocking == 0) {
            MPI_Alltoallv(s.p, sc.data(), sd.data(), MPI_DOUBLE, r.p, rc.data(), rd.data(), MPI_DOUBLE, MPI_COMM_WORLD);
        } else if (nonblocking == 1) {
            MPI_Request q;
            MPI_Ialltoallv(s.p, sc.data(), sd.data(), MPI_DOUBLE, r.p, rc.data(), rd.data(), MPI_DOUBLE, MPI_COMM_WORLD,
                           &q);
            MPI_Wait(&q, MPI_STATUS_IGNORE);
        } else {
            // two requests in flight, completed in the OPPOSITE order on odd ranks, the second by MPI_Test
            dev_array<double> r2(rpos + 4);
            MPI_Request q1, q2;
            MPI_Ialltoallv(s.p, sc.data(), sd.data(), MPI_DOUBLE, r.p, rc.data(), rd.data(), MPI_DOUBLE, MPI_COMM_WORLD,
                           &q1);
            MPI_Ialltoallv(s.p, sc.data(), sd.data(), MPI_DOUBLE, r2.p, rc.data(), rd.data(), MPI_DOUBLE,
                           MPI_COMM_WORLD, &q2);
            if (rank % 2) {
                int done = 0;
                while (!done) MPI_Test(&q2, &done, MPI_STATUS_IGNORE);
                MPI_Wait(&q1, MPI_STATUS_IGNORE);
            } else {
                MPI_Wait(&q1, MPI_STATUS_IGNORE);
                MPI_Wait(&q2, MPI_STATUS_IGNORE);
            }
            auto h1 = r.get(), h2 = r2.get();
            bool same = true;
            for (int p = 0; p < size; ++p) {
                for (int i = 0; i < rc[p]; ++i) same = same && h1[rd[p] + i] == h2[rd[p] + i];
            }
            CHECK(same, "two nonblocking all-to-all-v in flight, completed in different orders");
        }
        auto h = r.get();
        bool ok = true;
        for (int p = 0; p < size; ++p) {
            for (int i = 0; i < rc[p]; ++i) ok = ok && h[rd[p] + i] == 1000.0 * p + 10.0 * rank + 0.001 * i + 1e5 * nonblocking;
        }
        CHECK(ok, nonblocking == 0 ? "all-to-all-v" : nonblocking == 1 ? "nonblocking all-to-all-v" : "... data");
    }
    lap("all-to-all-v x 3");
    // MPI_Test does not wait for the peers.  Even ranks post the exchange and then POLL it with
    // MPI_Test while serving a synchronous point-to-point message of their odd neighbour; odd ranks post the
    // exchange, send that message and only then wait.  An MPI_Test that carried the exchange out to its end would
    // sit in the device exchange until the neighbour's kernel starts - which it does in the neighbour's MPI_Wait,
    // behind an MPI_Ssend that only returns once this rank has received: a deadlock in a legal program.
    {
        std::vector<int> c(size, 4), d(size);
        for (int p = 0; p < size; ++p) d[p] = 4 * p;
        dev_array<double> s(size_t(size) * 4), r(size_t(size) * 4);
        std::vector<double> hs(size_t(size) * 4);
        for (int p = 0; p < size; ++p) {
            for (int i = 0; i < 4; ++i) hs[4 * p + i] = 7.0 * rank + 0.5 * p + 0.01 * i;
        }
        s.put(hs);
        r.put(std::vector<double>(size_t(size) * 4, -3.0));
        MPI_Request q;
        MPI_Ialltoallv(s.p, c.data(), d.data(), MPI_DOUBLE, r.p, c.data(), d.data(), MPI_DOUBLE, MPI_COMM_WORLD, &q);
        long token = 0;
        int tests = 0;
        if (rank % 2 == 0) {
            const int partner = rank + 1 < size ? rank + 1 : -1;
            bool got = partner < 0;
            int done = 0;
            while (!done || !got) {
                if (!done) {
                    MPI_Test(&q, &done, MPI_STATUS_IGNORE);
                    ++tests;
                }
                if (!got) {
                    int there = 0;
                    MPI_Iprobe(partner, 77, MPI_COMM_WORLD, &there, MPI_STATUS_IGNORE);
                    if (there) {
                        MPI_Recv(&token, 1, MPI_LONG, partner, 77, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
                        got = true;
                    }
                }
            }
            CHECK(partner < 0 || token == 1000 + partner, "the point-to-point message served between two MPI_Test");
        } else {
            token = 1000 + rank;
            MPI_Ssend(&token, 1, MPI_LONG, rank - 1, 77, MPI_COMM_WORLD);
            MPI_Wait(&q, MPI_STATUS_IGNORE);
        }
        auto h = r.get();
        bool ok = true;
        for (int p = 0; p < size; ++p) {
            for (int i = 0; i < 4; ++i) ok = ok && h[4 * p + i] == 7.0 * p + 0.5 * rank + 0.01 * i;
        }
        CHECK(ok, "MPI_Test does not wait for the peers: the polled exchange delivered");
        if (rank == 0 && std::getenv("GKOC_MPI_VERBOSE")) std::fprintf(stderr, "[mpi_layer_test] rank 0 polled %d times\n", tests);
    }
    lap("polled all-to-all-v beside a synchronous send");
    // a derived datatype that is freed BEFORE the wait (core/distributed/row_gatherer.cpp apply_async: the
    // contiguous type of a multi-column vector lives in the scope that posts the exchange)
    {
        std::vector<int> c(size, 3), d(size);
        for (int p = 0; p < size; ++p) d[p] = 3 * p;
        dev_array<double> s(size_t(size) * 6), r(size_t(size) * 6);
        std::vector<double> hs(size_t(size) * 6);
        for (int p = 0; p < size; ++p) {
            for (int i = 0; i < 6; ++i) hs[6 * p + i] = 10.0 * rank + p + 0.1 * i;
        }
        s.put(hs);
        MPI_Request q;
        {
            MPI_Datatype pair;
            MPI_Type_contiguous(2, MPI_DOUBLE, &pair);
            MPI_Type_commit(&pair);
            MPI_Ialltoallv(s.p, c.data(), d.data(), pair, r.p, c.data(), d.data(), pair, MPI_COMM_WORLD, &q);
            MPI_Type_free(&pair);
        }
        MPI_Wait(&q, MPI_STATUS_IGNORE);
        auto h = r.get();
        bool ok = true;
        for (int p = 0; p < size; ++p) {
            for (int i = 0; i < 6; ++i) ok = ok && h[6 * p + i] == 10.0 * p + rank + 0.1 * i;
        }
        CHECK(ok, "nonblocking all-to-all-v with a datatype freed before the wait");
    }
    lap("freed datatype");
    // the neighbourhood form on a ring (every rank talks to rank - 1 and rank + 1)
    if (size > 1) {
        const int left = (rank + size - 1) % size, right = (rank + 1) % size;
        std::vector<int> nb = left == right ? std::vector<int>{left} : std::vector<int>{left, right};
        MPI_Comm ring;
        MPI_Dist_graph_create_adjacent(MPI_COMM_WORLD, int(nb.size()), nb.data(), MPI_UNWEIGHTED, int(nb.size()),
                                       nb.data(), MPI_UNWEIGHTED, MPI_INFO_NULL, 0, &ring);
        const int cnt = 7;
        std::vector<int> c(nb.size(), cnt), d(nb.size());
        for (size_t i = 0; i < nb.size(); ++i) d[i] = int(i) * cnt;
        dev_array<double> s(nb.size() * cnt), r(nb.size() * cnt);
        std::vector<double> hs(nb.size() * cnt);
        for (size_t i = 0; i < nb.size(); ++i) {
            for (int k = 0; k < cnt; ++k) hs[i * cnt + k] = 100.0 * rank + nb[i] + 0.01 * k;
        }
        s.put(hs);
        MPI_Request q;
        MPI_Ineighbor_alltoallv(s.p, c.data(), d.data(), MPI_DOUBLE, r.p, c.data(), d.data(), MPI_DOUBLE, ring, &q);
        MPI_Wait(&q, MPI_STATUS_IGNORE);
        auto h = r.get();
        bool ok = true;
        for (size_t i = 0; i < nb.size(); ++i) {
            for (int k = 0; k < cnt; ++k) ok = ok && h[i * cnt + k] == 100.0 * nb[i] + rank + 0.01 * k;
        }
        CHECK(ok, "nonblocking neighbourhood all-to-all-v on a ring");
        MPI_Comm_free(&ring);
    }
    lap("neighbourhood all-to-all-v + Comm_free");
    int all = 0;
    MPI_Allreduce(&failures, &all, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
    if (rank == 0) std::printf(all == 0 ? "MPI LAYER: ALL PASSED (%d ranks)\n" : "MPI LAYER: %d FAILED\n", all == 0 ? size : all);
    MPI_Finalize();
    return all == 0 ? 0 : 1;
}
