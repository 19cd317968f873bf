/*
 * gko_cdna4.h -- C ABI of libgko_cdna4.so, the MI355X (gfx950 / CDNA4) sparse
 * linear-algebra kernel backend for Ginkgo's Krylov hot path.
 *
 * Every entry point is the C restatement of ONE kernel symbol that Ginkgo's
 * core calls through `exec->run(ns::make_<op>(...))` on a HipExecutor, i.e. one
 * `gko::kernels::hip::<ns>::<op>` function declared by
 * GKO_DECLARE_FOR_ALL_EXECUTOR_NAMESPACES (core/base/kernel_declaration.hpp:10-39
 * of the reference).  The Ginkgo-side binding (ginkgo_amd/gko_binding/, see
 * INTEGRATION.md) unwraps matrix::Csr / matrix::Dense / array<T> objects into
 * the raw device pointers + sizes taken here.
 *
 * Conventions
 *  - all pointers are DEVICE pointers unless a parameter is documented `host`;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *    work is enqueued asynchronously on it (HipExecutor::get_stream(),
 *    include/ginkgo/core/base/executor.hpp:1940);
 *  - dense operands are row-major with a row stride `ld` counted in elements
 *    (matrix::Dense, include/ginkgo/core/matrix/dense.hpp:88); nrhs = columns;
 *  - scalars alpha / beta / rho ... are device-resident 1 x 1 (or 1 x nrhs)
 *    Dense buffers exactly as in Ginkgo - never dereferenced on the host;
 *  - suffix _f64/_f32 = value type, _i32/_i64 = index type;
 *  - return value: 0 on success, a positive hipError_t, or a negative
 *    GKOC_E_* code; gkoc_last_error() returns a thread-local message.  The
 *    Ginkgo binding turns non-zero into gko::HipError / gko::NotSupported.
 *  - nothing here allocates device memory except the gkoc_*_create handles;
 *    reductions take a caller-owned workspace (Ginkgo's `array<char>& tmp`).
 */
#ifndef GKO_CDNA4_H_
#define GKO_CDNA4_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GKOC_VERSION_MAJOR 0
#define GKOC_VERSION_MINOR 1

#define GKOC_OK 0
#define GKOC_E_INVALID (-1)       /* bad argument                       */
#define GKOC_E_NOT_SUPPORTED (-2) /* combination not implemented        */
#define GKOC_E_WORKSPACE (-3)     /* workspace too small                */
#define GKOC_E_NO_DEVICE (-4)     /* no gfx950 device / HIP unavailable */
#define GKOC_E_COMM (-5)          /* RCCL failure                       */
#define GKOC_E_OVERFLOW (-6)      /* index computation overflows (gko::OverflowError) */

typedef void* gkoc_stream_t;

/* complex<double> / complex<float> of the value-type lists (include/ginkgo/core/base/types.hpp:471,
 * 689): two reals, real part first (the layout of the C and C++ complex types).  Complex kernels follow the
 * reference's expressions with the textbook complex product and a scaled quotient: they agree with the
 * reference to rounding (its own tolerance r<value_type>), not bit for bit. */
#ifndef GKOC_COMPLEX_TYPES_DEFINED   /* (the library defines the same two layouts with arithmetic) */
typedef struct { double re, im; } gkoc_c128;
typedef struct { float re, im; } gkoc_c64;
#endif


/* ------------------------------------------------------------------ runtime
 * replaces HipExecutor::{raw_alloc,raw_free,raw_copy_to,synchronize,
 * get_num_devices,set_gpu_property} (core/device_hooks/hip_hooks.cpp:21-252,
 * hip/base/executor.hip.cpp) */
typedef struct gkoc_device_info {
    int32_t device_id;
    int32_t num_cu;           /* exec_info.num_computing_units           */
    int32_t wave_size;        /* exec_info.max_subgroup_size (64)        */
    int32_t num_xcd;          /* 8 on MI355X                             */
    int32_t max_threads_per_block;
    int32_t major, minor;     /* 9, 5 for gfx950                         */
    int32_t lds_bytes_per_cu; /* 163840                                  */
    int64_t hbm_bytes;
    char arch[64];            /* "gfx950..."                             */
} gkoc_device_info;

const char* gkoc_last_error(void);
int gkoc_version(void);
int gkoc_get_num_devices(int* count);
int gkoc_get_device_info(int device_id, gkoc_device_info* info);
int gkoc_set_device(int device_id);
int gkoc_get_device(int* device_id);
int gkoc_malloc(void** ptr, size_t bytes);
int gkoc_free(void* ptr);               /* device and managed memory */
/* Device-memory arena behind gkoc_malloc / gkoc_free (HipExecutor::raw_alloc /
 * raw_free, hip/base/executor.hip.cpp:95-112; HipAllocator, hip/base/memory.hip.cpp).
 * MI355X memory consists of three classes; a kernel whose output lives in the class
 * of its read streams is ~11 % slower.  mode 0: one hipMalloc per
 * request (the reference's behaviour); 1: large chunks from hipMalloc, requests
 * placed inside; 2 (default): one region per memory class, built from 1 GiB
 * physical granules whose class is measured, requests placed by role - matrix
 * values, index arrays and vectors in three different classes.  gkoc_malloc guesses
 * the role from the size (>= 1/4 of the largest live request: matrix array),
 * gkoc_malloc_role states it.  chunk_bytes (mode 1) 0 keeps the current size (8 GiB,
 * GKOC_ARENA_CHUNK_MB); sync_on_free != 0 keeps hipFree's implicit device
 * synchronisation.  Environment: GKOC_ARENA=<mode>, GKOC_ARENA_VERBOSE=1,
 * GKOC_ARENA_MAX_CLASSES=1|2 (use fewer classes: what a device whose free memory lacks a
 * class looks like), GKOC_ARENA_MAX_WALK=<granules> (bound of one search; default: three
 * quarters of the free memory).
 * Configure before the first gkoc_malloc. */
#define GKOC_MEM_AUTO 0
#define GKOC_MEM_VALUES 1   /* large read-only stream no. 1 (matrix values, Jacobi blocks) */
#define GKOC_MEM_INDICES 2  /* large read-only stream no. 2 (column indices, row pointers)  */
#define GKOC_MEM_VECTOR 3   /* everything kernels write: vectors, workspaces               */
typedef struct gkoc_arena_info {
    int32_t mode;
    int32_t num_classes;          /* memory classes found so far (mode 2)          */
    int64_t chunk_bytes;          /* chunk / granule size                          */
    int64_t num_chunks;
    int64_t reserved_bytes;
    int64_t used_bytes;
    int64_t num_allocations;
    int64_t probes;               /* probe launches so far                         */
    int64_t granules_walked;      /* physical granules created while searching     */
    int64_t spare_bytes;          /* classified granules waiting in the pools      */
    int64_t class_reserved_bytes[3];
    int64_t class_used_bytes[3];
    int64_t granules_classified;  /* ... of them mapped and probed (the walk gallops)  */
    int64_t search_ns;            /* host time spent in the searches                    */
    int64_t probe_retries;        /* classifications repeated: verdict not one-hot      */
    int64_t surveyed;             /* 1: the one search for all classes has run          */
    int64_t search_budget_ms;     /* GKOC_ARENA_SURVEY_MS (default 1500): wall-clock bound of all searches of
                                   * this device in this process; 0 = bounded by the free memory only        */
    int64_t search_budget_spent;  /* 1: a search stopped at the bound - the classes found so far are final   */
    int64_t granules_unclassified;/* granules mapped into a region without a probe after the bound was hit   */
} gkoc_arena_info;
int gkoc_malloc_role(void** ptr, size_t bytes, int role);
int gkoc_arena_configure(int mode, size_t chunk_bytes, int sync_on_free);
int gkoc_arena_stats(gkoc_arena_info* info);
/* *cls = memory class (0..2) of an address inside a class region, -1 otherwise */
int gkoc_arena_class_of(const void* ptr, int* cls);
/* Role feedback for gkoc_malloc (which, like Ginkgo's raw_alloc, is told no role): the array
 * that holds ptr has been WRITTEN as a vector by a kernel - later requests of that size or a
 * multiple of it (Krylov bases, multi-vectors) are placed with the vectors.  role_stats: how
 * many vector sizes are known and how many of them were found next to matrix arrays when first
 * seen (placed before anything was known about the system). */
int gkoc_arena_note_vector(const void* ptr);
/* ... and the counterpart: the array at ptr is a MATRIX array (values / column indices handed to an SpMV
 * entry): requests of its size are matrix arrays from now on, although k n values of a matrix with k
 * entries per row are a multiple of the n-vector */
int gkoc_arena_note_matrix(const void* ptr);
int gkoc_arena_role_stats(int64_t* n_vector_sizes, int64_t* misplaced_vectors);
int gkoc_arena_trim(void);              /* return empty chunks to the driver */
/* The arena's memory-class probe, exposed for diagnostics: every wavefront reads
 * read_kb_per_wave KiB of x (x_bytes in all, read only) and then writes
 * write_bytes_per_wave (multiple of 1024) of y; *ns = best time of `reps` launches.
 * y needs x_bytes / (read_kb_per_wave * 1024) * write_bytes_per_wave bytes.
 * x and y in the same memory class: ~10 % slower than in different ones. */
int gkoc_arena_probe(const void* x, size_t x_bytes, void* y, int read_kb_per_wave,
                     int write_bytes_per_wave, int reps, int64_t* ns);
/* Process-wide tuning switches (defaults chosen by measurement; the
 * environment variable GKOC_TUNE_<key> overrides the default).  Results never
 * depend on them. */
#define GKOC_TUNE_CSR_XCD_MAP 0    /* each XCD walks one contiguous eighth of the rows: 0 (default) for matrices with
                                      hub rows (flagged segments) only, 1 always, 2 never */
#define GKOC_TUNE_JACOBI_XCD_MAP 1 /* same for the block-Jacobi apply                     */
#define GKOC_TUNE_JACOBI_MFMA 3     /* block-Jacobi(8) apply, several right-hand sides, on the f64 matrix cores
                                      (fused multiply-adds: ~5e-16 off the reference's bits): 0 never,
                                      1 from two columns, 2 (default) from nine columns on (two to eight
                                      take the exact multi-column kernel), 3 from four columns on */
#define GKOC_TUNE_COO_FUSED 4       /* coo::spmv, one column: one pass over values, columns and rows with the row
                                      pointers derived while streaming (default 1); 0: row pointers in a pass of
                                      their own, then the CSR kernel */
#define GKOC_TUNE_DEFERRED_FUSION 5 /* binding for the unmodified Ginkgo core (gko_binding/fusion.cpp): cg::step_2 and
                                      the block-Jacobi application after it are held until the next call and run
                                      as one kernel with the dot product that follows.  OPT-IN (default 0: every
                                      call launches its own kernel when it returns, Ginkgo's contract): with 1, code
                                      that launches its OWN kernels on exec->get_stream() with raw pointers of a
                                      solver's internal vectors - e.g. a matrix-free preconditioner - would read
                                      them before the held kernels ran ("Fusion across calls") */
#define GKOC_TUNE_MULTI_XCD_CHUNK_ROWS 6 /* SpMV with several right-hand sides (CSR / ELL / SELL-P): rows per chunk
                                      that one XCD walks before the next XCD's chunk begins (csrc/common.hpp,
                                      xcd_chunked_block); 0: every 8th workgroup (plain dispatch order) */
#define GKOC_TUNE_CSR_LOAD_GROUPS 2 /* csr::spmv, one column: 0 (default) two entries per lane and load, three load
                                      groups in flight (float: four entries, two groups); 1: four (eight) entries,
                                      one group - the layout of rounds 1-2, kept for A/B measurements; 2: the
                                      default layout also where the size rule picks the other (small products);
                                      3: one entry per lane and load, eight groups - since round 6 what 0 picks for
                                      float values with 40 and more entries per row; 4: two entries, four groups */
#define GKOC_TUNE_GATE_FENCE 7      /* one-kernel distributed product (gkoc_csr_spmv_gated_*): 0 (default) only a
                                      boundary wave that had to wait for its halo pays an agent-scope acquire
                                      fence; 1: every boundary wave does (+8 us per product at 2048 waves) */
#define GKOC_TUNE_GATE_POS 8        /* one-kernel distributed product: the boundary waves start behind this many per
                                      cent of the interior waves (100: they are the last waves of the grid) */
#define GKOC_TUNE_REDUCE_ONE_KERNEL 9 /* dot / norm2 / squared_norm2 of one contiguous column: 1: the block that finishes
                                     * last folds the partial sums (one launch, the same bits as the two-launch
                                     * form: same threads, same tree); 0 (default): two launches.  Measured: the
                                     * agent-scope fences cost more than the launch they save (one rank's CG
                                     * iteration of 256^3 / 8: 209 -> 235 us, profiles/r04_experiments.txt) */
#define GKOC_TUNE_ANTICIPATE 10     /* binding for the unmodified Ginkgo core, by-product mode (gko_binding/fusion.cpp):
                                      1 (default): once a Cg solve has shown cg::step_2(x, r) followed directly by
                                      jacobi::simple_apply(M, r -> z), the next step_2 runs as ONE kernel with that
                                      application (x, r exist when step_2 returns as always; z = M r is written
                                      early, the application call that follows launches nothing); 0: off */
#define GKOC_TUNE_CSR_MULTI_VARIANT 11 /* csr::spmv with three to eight right-hand sides: layout variants kept for A/B
                                      measurements (csrc/csr_spmv.hip); 0 = the default chosen by measurement */
#define GKOC_TUNE_CSR_LONG_ROWS 12   /* csr::spmv, one right-hand side: 1 (default) the 64-row segments that hold a row
                                      longer than GKOC_CSR_LONG_ROW are found once per matrix (one scan of the row
                                      pointers, remembered per (row_ptrs, n_rows)) and multiplied by many workgroups
                                      each (csrc/csr_long_rows.hpp) instead of by one wave; 0: one wave, as before.
                                      Two products with the SAME matrix must not run at the same time on two streams
                                      when it has such rows (they share the chunk sums' scratch) */
#define GKOC_TUNE_CSR_SEGS_PER_WAVE 13 /* csr::spmv, one right-hand side: 64-row segments a wave walks.  0 (default): two
                                      from 4 M rows on, one below; 1, 2: that many (round 6 tried four and eight
                                      on rows of a dozen entries: never faster - profiles/r06/) */
#define GKOC_TUNE_CSR_SHORT_ROWS 14 /* (not read any more) round 6's experiments on matrices with a dozen entries per
                                      row - smaller rings and load groups, non-temporal / agent-scope gathers, two
                                      to eight waves per workgroup: none was faster than the launcher's rule, the
                                      variants are gone (profiles/r06/r06_segments_per_wave.txt,
                                      r06_waves_per_workgroup.txt) */
#define GKOC_TUNE_JACOBI_LANES 15    /* block-Jacobi apply for float / complex values and adaptive storage of those:
                                      0 (default) lane = (block, row) of a storage group, 1: the thread-per-row
                                      kernels of round 5 (A/B runs) */
#define GKOC_TUNE_CCSR_THREAD_PER_ROW 16 /* csr::spmv on complex values: 0 (default) the row-segment kernel of the real
                                      types, 1: one thread per row (round 5; A/B runs) */
#define GKOC_TUNE_JACOBI_REHOME 17   /* binding for the unmodified Ginkgo core: 1 (default) jacobi::generate re-allocates
                                      an owning block array that sits in a memory class with vectors in the class of
                                      the matrix' column indices before it fills it; 0: left where raw_alloc put it */
int gkoc_tune_set(int key, int64_t value);
int gkoc_tune_get(int key, int64_t* value);
/* HipHostAllocator (pinned host memory) and HipUnifiedAllocator (managed memory,
 * flags = hipMemAttachGlobal 1 / hipMemAttachHost 2): hip_hooks.cpp:60-100,
 * hip/base/memory.hip.cpp */
int gkoc_malloc_host(void** ptr, size_t bytes);
int gkoc_free_host(void* ptr);
int gkoc_malloc_managed(void** ptr, size_t bytes, unsigned int flags);
/* 1 if ptr is device memory (the arena's or any other HIP device allocation), 0 for host
 * memory (pageable, pinned or managed); the process' device identity "host/pci-bus-id"
 * (64 bytes): what the GPU-aware-MPI layer (gko_binding/mpi_rccl.cpp) needs to route buffers */
int gkoc_pointer_is_device(const void* ptr, int* is_device);
int gkoc_device_identity(char* out, size_t out_bytes);
int gkoc_memcpy_h2d(void* dst, const void* src_host, size_t bytes, gkoc_stream_t s);
int gkoc_memcpy_d2h(void* dst_host, const void* src, size_t bytes, gkoc_stream_t s);
int gkoc_memcpy_d2d(void* dst, const void* src, size_t bytes, gkoc_stream_t s);
int gkoc_memset(void* dst, int value, size_t bytes, gkoc_stream_t s);
int gkoc_stream_create(gkoc_stream_t* s);
int gkoc_stream_create_high_priority(gkoc_stream_t* s);   /* for the exchange / collective side stream */
int gkoc_stream_destroy(gkoc_stream_t s);
int gkoc_stream_synchronize(gkoc_stream_t s);
/* *done = 1 if everything enqueued on s has completed, 0 if not (hipStreamQuery); never waits */
int gkoc_stream_query(gkoc_stream_t s, int* done);
int gkoc_device_synchronize(void);
/* ROCTX ranges (log::begin_roctx / end_roctx, hip/base/roctx.hip.cpp:30-36; ProfilerHook::
 * create_roctx): librocprofiler-sdk-roctx / libroctx64 bound with dlopen at first use, no-ops
 * when neither is there.  `rocprofv3 --marker-trace` then shows Ginkgo's operation ranges. */
int gkoc_range_push(const char* name);
int gkoc_range_pop(void);
int gkoc_range_available(void);   /* 1 if a roctx library was found */
/* events (HipTimer, hip/base/timer.hip.cpp; RowGatherer's event::record_event) */
typedef void* gkoc_event_t;
int gkoc_event_create(gkoc_event_t* e);
int gkoc_event_destroy(gkoc_event_t e);
int gkoc_event_record(gkoc_event_t e, gkoc_stream_t s);
int gkoc_event_synchronize(gkoc_event_t e);
int gkoc_event_elapsed_ns(gkoc_event_t start, gkoc_event_t stop, int64_t* ns);
int gkoc_stream_wait_event(gkoc_stream_t s, gkoc_event_t e);
/* hipGraph: everything enqueued on `s` between begin and end (any gkoc_* kernel
 * call without a host result, async copies) becomes one replayable graph.  No
 * counterpart in Ginkgo (its solvers re-issue every kernel per iteration); used
 * by this repository's own CG drivers for launch-bound system sizes. */
typedef void* gkoc_graph_t;
int gkoc_stream_begin_capture(gkoc_stream_t s);
int gkoc_stream_end_capture(gkoc_stream_t s, gkoc_graph_t* graph);
int gkoc_graph_launch(gkoc_graph_t graph, gkoc_stream_t s);
int gkoc_graph_destroy(gkoc_graph_t graph);

/* --------------------------------------------------------------- CSR SpMV
 * csr::spmv            core/matrix/csr_kernels.hpp:29-34
 * csr::advanced_spmv   core/matrix/csr_kernels.hpp:36-43
 * semantics = reference/matrix/csr_kernels.cpp:49-78 / :86-118: per row the
 * products val[k]*b[col[k]] are accumulated in k order with separate
 * multiply and add roundings => results are BIT-IDENTICAL to the
 * ReferenceExecutor for rows up to GKOC_CSR_LONG_ROW nnz (longer rows are
 * summed by a whole wavefront: same value to ~1 ulp*log2(len)).
 * beta == 0 never reads c (NaN-safe, reference/test/matrix/csr_kernels.cpp:521-534). */
#define GKOC_CSR_LONG_ROW 4096
#define GKOC_DECL_CSR(T, TN, I, IN)                                            \
    int gkoc_csr_spmv_##TN##_##IN(gkoc_stream_t s, int64_t n_rows,             \
                                  int64_t n_cols, const I* row_ptrs,           \
                                  const I* col_idxs, const T* vals,            \
                                  const T* b, int64_t ldb, T* c, int64_t ldc,  \
                                  int64_t nrhs);                               \
    int gkoc_csr_advanced_spmv_##TN##_##IN(                                    \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const T* alpha,       \
        const I* row_ptrs, const I* col_idxs, const T* vals, const T* b,       \
        int64_t ldb, const T* beta, T* c, int64_t ldc, int64_t nrhs);          \
    /* csr::extract_diagonal core/matrix/csr_kernels.hpp (diag[i]=A(i,i)|0) */ \
    int gkoc_csr_extract_diagonal_##TN##_##IN(                                 \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const I* row_ptrs,    \
        const I* col_idxs, const T* vals, T* diag);                            \
    /* csr::is_sorted_by_column_index; *is_sorted is HOST memory */            \
    int gkoc_csr_is_sorted_by_column_index_##TN##_##IN(                        \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, int* is_sorted_host);                               \
    /* csr::sort_by_column_index (in place, per-row stable by column)  */      \
    int gkoc_csr_sort_by_column_index_##TN##_##IN(                             \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, I* col_idxs,       \
        T* vals);
GKOC_DECL_CSR(double, f64, int32_t, i32)
GKOC_DECL_CSR(double, f64, int64_t, i64)
GKOC_DECL_CSR(float, f32, int32_t, i32)
GKOC_DECL_CSR(float, f32, int64_t, i64)

/* --------------------------------------------------------------- ELL SpMV
 * ell::spmv / advanced_spmv  core/matrix/ell_kernels.hpp:20-34
 * column-major storage: entry (row, j) at row + j*stride; padding col = -1
 * (include/ginkgo/core/matrix/ell.hpp:385-388).  reference semantics:
 * reference/matrix/ell_kernels.cpp:29-69. Bit-identical (row-sequential). */
#define GKOC_DECL_ELL(T, TN, I, IN)                                            \
    int gkoc_ell_spmv_##TN##_##IN(                                             \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols,                       \
        int64_t num_stored_per_row, int64_t stride, const I* col_idxs,         \
        const T* vals, const T* b, int64_t ldb, T* c, int64_t ldc,             \
        int64_t nrhs);                                                         \
    int gkoc_ell_advanced_spmv_##TN##_##IN(                                    \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols,                       \
        int64_t num_stored_per_row, int64_t stride, const T* alpha,            \
        const I* col_idxs, const T* vals, const T* b, int64_t ldb,             \
        const T* beta, T* c, int64_t ldc, int64_t nrhs);
GKOC_DECL_ELL(double, f64, int32_t, i32)
GKOC_DECL_ELL(double, f64, int64_t, i64)
GKOC_DECL_ELL(float, f32, int32_t, i32)
GKOC_DECL_ELL(float, f32, int64_t, i64)
GKOC_DECL_ELL(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_ELL(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_ELL(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_ELL(gkoc_c64, c64, int64_t, i64)

/* ------------------------------------------------------------ SELL-P SpMV
 * sellp::spmv / advanced_spmv  core/matrix/sellp_kernels.hpp:20-31
 * slice_sets / slice_lengths are size_type (uint64) arrays of length
 * n_slices+1 / n_slices; entry (row, j) of slice s at
 * (slice_sets[s] + j) * slice_size + row_in_slice
 * (include/ginkgo/core/matrix/sellp.hpp:379-383).  reference semantics:
 * reference/matrix/sellp_kernels.cpp:27-100. Bit-identical. */
#define GKOC_DECL_SELLP(T, TN, I, IN)                                          \
    int gkoc_sellp_spmv_##TN##_##IN(                                           \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t slice_size,   \
        const uint64_t* slice_sets, const uint64_t* slice_lengths,             \
        const I* col_idxs, const T* vals, const T* b, int64_t ldb, T* c,       \
        int64_t ldc, int64_t nrhs);                                            \
    int gkoc_sellp_advanced_spmv_##TN##_##IN(                                  \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t slice_size,   \
        const T* alpha, const uint64_t* slice_sets,                            \
        const uint64_t* slice_lengths, const I* col_idxs, const T* vals,       \
        const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc,             \
        int64_t nrhs);
GKOC_DECL_SELLP(double, f64, int32_t, i32)
GKOC_DECL_SELLP(double, f64, int64_t, i64)
GKOC_DECL_SELLP(float, f32, int32_t, i32)
GKOC_DECL_SELLP(float, f32, int64_t, i64)
GKOC_DECL_SELLP(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_SELLP(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_SELLP(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_SELLP(gkoc_c64, c64, int64_t, i64)

/* ---------------------------------------------------- format conversions
 * csr::convert_to_ell / convert_to_sellp, ell::compute_max_row_nnz,
 * sellp::compute_slice_sets, components::convert_ptrs_to_sizes,
 * convert_idxs_to_ptrs, prefix_sum_nonnegative, fill_array, fill_seq_array
 * (core/matrix/csr_kernels.hpp:99-125, core/components/prefix_sum_kernels.hpp etc.).
 * All outputs are integer-exact vs the reference. */
#define GKOC_DECL_CONV(T, TN, I, IN)                                           \
    int gkoc_csr_convert_to_ell_##TN##_##IN(                                   \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, const T* vals, int64_t num_stored_per_row,          \
        int64_t stride, I* ell_cols, T* ell_vals);                             \
    int gkoc_csr_convert_to_sellp_##TN##_##IN(                                 \
        gkoc_stream_t s, int64_t n_rows, int64_t slice_size,                   \
        const I* row_ptrs, const I* col_idxs, const T* vals,                   \
        const uint64_t* slice_sets, I* sellp_cols, T* sellp_vals);
GKOC_DECL_CONV(double, f64, int32_t, i32)
GKOC_DECL_CONV(double, f64, int64_t, i64)
GKOC_DECL_CONV(float, f32, int32_t, i32)
GKOC_DECL_CONV(float, f32, int64_t, i64)
GKOC_DECL_CONV(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CONV(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CONV(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CONV(gkoc_c64, c64, int64_t, i64)

/* components::aos_to_soa (core/base/device_matrix_data_kernels.hpp:27-30):
 * entries is an array of Ginkgo matrix_data_entry<T,I> = struct { I row;
 * I column; T value; } with natural alignment (include/ginkgo/core/base/
 * matrix_data.hpp:60); dense::fill_in_matrix_data
 * (core/matrix/dense_kernels.hpp:122-125): out(row, col) = value. */
#define GKOC_DECL_MD(T, TN, I, IN)                                             \
    int gkoc_aos_to_soa_##TN##_##IN(gkoc_stream_t s, int64_t nnz,              \
                                    const void* entries, I* row_idxs,          \
                                    I* col_idxs, T* vals);                     \
    int gkoc_dense_fill_in_matrix_data_##TN##_##IN(                            \
        gkoc_stream_t s, int64_t nnz, const I* row_idxs, const I* col_idxs,    \
        const T* vals, T* out, int64_t ld_out);
GKOC_DECL_MD(double, f64, int32_t, i32)
GKOC_DECL_MD(double, f64, int64_t, i64)
GKOC_DECL_MD(float, f32, int32_t, i32)
GKOC_DECL_MD(float, f32, int64_t, i64)

/* Device-side assembly of device_matrix_data
 * (core/base/device_matrix_data_kernels.hpp:23-51; reference/base/
 * device_matrix_data_kernels.cpp:24-143), all bit-identical to the reference:
 *   components::soa_to_aos      entries[i] = {row, column, value}
 *   components::sort_row_major  in place, stable by (row, column); needs
 *       gkoc_sort_row_major_workspace_bytes(nnz, sizeof value, sizeof index)
 *   components::remove_zeros    = count + fill: count marks value != 0, scans the
 *       marks into the workspace (gkoc_compact_workspace_bytes(nnz)) and returns
 *       the number of kept entries in HOST memory; the caller allocates the
 *       compacted arrays (as the reference does, and only if count < nnz) and
 *       fill scatters the kept entries through the SAME workspace
 *   components::sum_duplicates  = count + fill on row-major sorted input: count
 *       marks the first entry of every (row, column) run; fill writes one entry
 *       per run whose value is 0 + v0 + v1 + ... in storage order. */
size_t gkoc_sort_row_major_workspace_bytes(int64_t nnz, size_t value_size,
                                           size_t index_size);
size_t gkoc_compact_workspace_bytes(int64_t nnz);
int gkoc_remove_zeros_count_f64(gkoc_stream_t s, int64_t nnz,
                                const double* vals, void* work,
                                size_t work_bytes, int64_t* count_host);
int gkoc_remove_zeros_count_f32(gkoc_stream_t s, int64_t nnz,
                                const float* vals, void* work,
                                size_t work_bytes, int64_t* count_host);
int gkoc_sum_duplicates_count_i32(gkoc_stream_t s, int64_t nnz,
                                  const int32_t* row_idxs,
                                  const int32_t* col_idxs, void* work,
                                  size_t work_bytes, int64_t* count_host);
int gkoc_sum_duplicates_count_i64(gkoc_stream_t s, int64_t nnz,
                                  const int64_t* row_idxs,
                                  const int64_t* col_idxs, void* work,
                                  size_t work_bytes, int64_t* count_host);
#define GKOC_DECL_ASSEMBLY(T, TN, I, IN)                                       \
    int gkoc_soa_to_aos_##TN##_##IN(gkoc_stream_t s, int64_t nnz,              \
                                    const I* row_idxs, const I* col_idxs,      \
                                    const T* vals, void* entries);             \
    int gkoc_sort_row_major_##TN##_##IN(gkoc_stream_t s, int64_t nnz,          \
                                        I* row_idxs, I* col_idxs, T* vals,     \
                                        void* work, size_t work_bytes);        \
    int gkoc_remove_zeros_fill_##TN##_##IN(                                    \
        gkoc_stream_t s, int64_t nnz, const I* row_idxs, const I* col_idxs,    \
        const T* vals, const void* work, I* out_rows, I* out_cols,             \
        T* out_vals);                                                          \
    int gkoc_sum_duplicates_fill_##TN##_##IN(                                  \
        gkoc_stream_t s, int64_t nnz, const I* row_idxs, const I* col_idxs,    \
        const T* vals, const void* work, I* out_rows, I* out_cols,             \
        T* out_vals);
GKOC_DECL_ASSEMBLY(double, f64, int32_t, i32)
GKOC_DECL_ASSEMBLY(double, f64, int64_t, i64)
GKOC_DECL_ASSEMBLY(float, f32, int32_t, i32)
GKOC_DECL_ASSEMBLY(float, f32, int64_t, i64)

#define GKOC_DECL_IDX(I, IN)                                                   \
    /* ell::compute_max_row_nnz: *max_nnz is HOST memory */                    \
    int gkoc_compute_max_row_nnz_##IN(gkoc_stream_t s, int64_t n_rows,         \
                                      const I* row_ptrs, int64_t* max_host);   \
    /* sellp::compute_slice_sets: slice_lengths[s] = max row nnz in slice   \
       rounded up to stride_factor; slice_sets = exclusive prefix sum */       \
    int gkoc_sellp_compute_slice_sets_##IN(                                    \
        gkoc_stream_t s, int64_t n_rows, int64_t slice_size,                   \
        int64_t stride_factor, const I* row_ptrs, uint64_t* slice_sets,        \
        uint64_t* slice_lengths);                                              \
    int gkoc_convert_ptrs_to_sizes_##IN(gkoc_stream_t s, int64_t n,            \
                                        const I* ptrs, uint64_t* sizes);       \
    int gkoc_convert_idxs_to_ptrs_##IN(gkoc_stream_t s, int64_t num_idxs,      \
                                       const I* idxs, int64_t n, I* ptrs);     \
    /* in-place exclusive scan over n entries (last entry = total) */          \
    int gkoc_prefix_sum_nonnegative_##IN(gkoc_stream_t s, I* counts,           \
                                         int64_t n);                           \
    /* the same with the reference's overflow check (GKOC_E_OVERFLOW when a */ \
    /* partial sum exceeds the type; synchronises the stream)              */ \
    int gkoc_prefix_sum_nonnegative_checked_##IN(gkoc_stream_t s, I* counts,   \
                                                 int64_t n);                   \
    int gkoc_fill_array_##IN(gkoc_stream_t s, I* data, int64_t n, I value);    \
    int gkoc_fill_seq_array_##IN(gkoc_stream_t s, I* data, int64_t n);
/* csr::build_lookup_offsets / csr::build_lookup (core/matrix/csr_kernels.hpp, format
 * core/matrix/csr_lookup.hpp:26-85, reference/matrix/csr_kernels.cpp:1425-1573): per row a 64-bit
 * descriptor and int32 storage (full / bitmap / hash, `allowed` = bit set 1 | 2 | 4 of the kinds the
 * caller accepts) that locate an entry without a search; sorted columns.  storage_offsets: n_rows + 1
 * entries (exclusive sums of the storage per row).  Tables bit-identical to the reference's. */
#define GKOC_DECL_LOOKUP(I, IN)                                                                        \
    int gkoc_csr_build_lookup_offsets_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,         \
                                           const I* col_idxs, int allowed, I* storage_offsets);        \
    int gkoc_csr_build_lookup_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                 \
                                   const I* col_idxs, int allowed, const I* storage_offsets,           \
                                   int64_t* row_desc, int32_t* storage);
/* what a *_count call of the distributed set-up handed out as `state`, when the matching *_fill
 * call is not going to happen (the fill frees it otherwise) */
int gkoc_dist_separate_state_free(gkoc_stream_t s, void* state);
int gkoc_index_map_mapping_state_free(gkoc_stream_t s, void* state);
GKOC_DECL_LOOKUP(int32_t, i32)
GKOC_DECL_LOOKUP(int64_t, i64)
GKOC_DECL_IDX(int32_t, i32)
GKOC_DECL_IDX(int64_t, i64)
/* components::fill_array for value types (core/components/fill_array_kernels.hpp) */
int gkoc_fill_array_f64(gkoc_stream_t s, double* data, int64_t n, double value);
int gkoc_fill_array_f32(gkoc_stream_t s, float* data, int64_t n, float value);
int gkoc_prefix_sum_nonnegative_u64(gkoc_stream_t s, uint64_t* counts,
                                    int64_t n);
int gkoc_prefix_sum_nonnegative_checked_u64(gkoc_stream_t s, uint64_t* counts,
                                            int64_t n);
/* out[i] = (int32) in[i]: the 64-bit row pointers Ell / Sellp / Hybrid::read compute for
 * device_matrix_data, narrowed for the converters of matrices with 32-bit indices */
int gkoc_narrow_i64_to_i32(gkoc_stream_t s, int64_t n, const int64_t* in, int32_t* out);
/* convert_idxs_to_ptrs with row pointers of the other width
 * (core/components/format_conversion_kernels.hpp:31-38) */
int gkoc_convert_idxs_to_ptrs_i32_i64(gkoc_stream_t s, int64_t num_idxs,
                                      const int32_t* idxs, int64_t n, int64_t* ptrs);
int gkoc_convert_idxs_to_ptrs_i64_i32(gkoc_stream_t s, int64_t num_idxs,
                                      const int64_t* idxs, int64_t n, int32_t* ptrs);

/* --------------------------------------------- Dense x Dense, precision conversion
 * dense::simple_apply / apply (core/matrix/dense_kernels.hpp:23-32, reference/matrix/
 * dense_kernels.cpp:38-92): C (m x n) = A (m x k) B (k x n) resp. alpha A B + beta C, row-major
 * with row strides; every entry is the reference's in-order sum (bit-identical).  dense::copy
 * between precisions (:95-106) = gkoc_dense_convert_<from>_<to>.  Off the hot path. */
#define GKOC_DECL_GEMM(T, TN)                                                   \
    int gkoc_dense_simple_apply_##TN(gkoc_stream_t s, int64_t m, int64_t n,     \
                                     int64_t k, const T* a, int64_t lda,        \
                                     const T* b, int64_t ldb, T* c, int64_t ldc); \
    int gkoc_dense_apply_##TN(gkoc_stream_t s, int64_t m, int64_t n, int64_t k,  \
                              const T* alpha, const T* a, int64_t lda,          \
                              const T* b, int64_t ldb, const T* beta, T* c,     \
                              int64_t ldc);
GKOC_DECL_GEMM(double, f64)
GKOC_DECL_GEMM(float, f32)
GKOC_DECL_GEMM(gkoc_c128, c128)
GKOC_DECL_GEMM(gkoc_c64, c64)
/* dense::compute_sqrt on complex values (the real ones: GKOC_DECL_DENSE) */
int gkoc_dense_compute_sqrt_c128(gkoc_stream_t s, int64_t cols, gkoc_c128* x);
int gkoc_dense_compute_sqrt_c64(gkoc_stream_t s, int64_t cols, gkoc_c64* x);
int gkoc_dense_convert_f64_f32(gkoc_stream_t s, int64_t rows, int64_t cols,
                               const double* x, int64_t ldx, float* y, int64_t ldy);
int gkoc_dense_convert_f32_f64(gkoc_stream_t s, int64_t rows, int64_t cols,
                               const float* x, int64_t ldx, double* y, int64_t ldy);

/* ------------------------------------------------------------ Dense BLAS-1
 * dense::{fill,copy,scale,inv_scale,add_scaled,sub_scaled}
 *   core/matrix/dense_kernels.hpp:34-61, reference/matrix/dense_kernels.cpp:96-225
 *   alpha is 1 x 1 (alpha_cols == 1) or 1 x nrhs (alpha_cols == nrhs).
 * dense::{compute_dot,compute_conj_dot,compute_norm2,compute_squared_norm2}
 *   core/matrix/dense_kernels.hpp:75-131, reference/matrix/dense_kernels.cpp:263-352
 *   result is a device 1 x nrhs row; `work` is the caller-owned scratch
 *   (Ginkgo's array<char>& tmp), at least gkoc_reduction_workspace_bytes().
 *   Reductions are deterministic (fixed tree for a given n, nrhs) but their
 *   summation order differs from the sequential reference: |err| <=
 *   1e-13 * sum|x_i y_i| in fp64 (tests/test_dense_gpu.py).
 * dense::row_gather  common/unified/matrix/dense_kernels.template.cpp:449-473 */
size_t gkoc_reduction_workspace_bytes(int64_t n_rows, int64_t nrhs,
                                      size_t value_size);
#define GKOC_DECL_DENSE(T, TN)                                                 \
    int gkoc_dense_fill_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,      \
                             T* x, int64_t ldx, T value);                      \
    int gkoc_dense_copy_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,      \
                             const T* x, int64_t ldx, T* y, int64_t ldy);      \
    /* y = |x| (dense::outplace_absolute_dense; x == y: inplace_absolute_dense) */ \
    int gkoc_dense_absolute_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,  \
                                 const T* x, int64_t ldx, T* y, int64_t ldy);  \
    int gkoc_dense_scale_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,     \
                              const T* alpha, int64_t alpha_cols, T* x,        \
                              int64_t ldx);                                    \
    int gkoc_dense_inv_scale_##TN(gkoc_stream_t s, int64_t rows,               \
                                  int64_t cols, const T* alpha,                \
                                  int64_t alpha_cols, T* x, int64_t ldx);      \
    int gkoc_dense_add_scaled_##TN(gkoc_stream_t s, int64_t rows,              \
                                   int64_t cols, const T* alpha,               \
                                   int64_t alpha_cols, const T* x,             \
                                   int64_t ldx, T* y, int64_t ldy);            \
    int gkoc_dense_sub_scaled_##TN(gkoc_stream_t s, int64_t rows,              \
                                   int64_t cols, const T* alpha,               \
                                   int64_t alpha_cols, const T* x,             \
                                   int64_t ldx, T* y, int64_t ldy);            \
    int gkoc_dense_compute_dot_##TN(gkoc_stream_t s, int64_t rows,             \
                                    int64_t cols, const T* x, int64_t ldx,     \
                                    const T* y, int64_t ldy, T* result,        \
                                    void* work, size_t work_bytes);            \
    int gkoc_dense_compute_norm2_##TN(gkoc_stream_t s, int64_t rows,           \
                                      int64_t cols, const T* x, int64_t ldx,   \
                                      T* result, void* work,                   \
                                      size_t work_bytes);                      \
    int gkoc_dense_compute_squared_norm2_##TN(                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* x, int64_t ldx,  \
        T* result, void* work, size_t work_bytes);                             \
    int gkoc_dense_compute_sqrt_##TN(gkoc_stream_t s, int64_t cols, T* x);     \
    int gkoc_dense_row_gather_##TN##_i32(                                      \
        gkoc_stream_t s, int64_t n_gather, int64_t cols, const int32_t* rows,  \
        const T* orig, int64_t ld_orig, T* gathered, int64_t ld_gathered);     \
    int gkoc_dense_row_gather_##TN##_i64(                                      \
        gkoc_stream_t s, int64_t n_gather, int64_t cols, const int64_t* rows,  \
        const T* orig, int64_t ld_orig, T* gathered, int64_t ld_gathered);
GKOC_DECL_DENSE(double, f64)
GKOC_DECL_DENSE(float, f32)

/* ------------------------------------------------------------- CG steps
 * cg::{initialize,step_1,step_2}  core/solver/cg_kernels.hpp:25-48,
 * reference/solver/cg_kernels.cpp:25-100.  stop_status: one byte per rhs
 * column, bit layout include/ginkgo/core/stop/stopping_status.hpp:117-121
 * (bit7 converged, bit6 finalized, low 6 bits = stopping id).  Element-wise,
 * bit-identical to the reference (separate divide, multiply, add). */
#define GKOC_DECL_CG(T, TN)                                                    \
    int gkoc_cg_initialize_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,   \
                                const T* b, int64_t ldb, T* r, int64_t ldr,    \
                                T* z, int64_t ldz, T* p, int64_t ldp, T* q,    \
                                int64_t ldq, T* prev_rho, T* rho,              \
                                uint8_t* stop_status);                         \
    int gkoc_cg_step_1_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,       \
                            T* p, int64_t ldp, const T* z, int64_t ldz,        \
                            const T* rho, const T* prev_rho,                   \
                            const uint8_t* stop_status);                       \
    int gkoc_cg_step_2_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,       \
                            T* x, int64_t ldx, T* r, int64_t ldr, const T* p,  \
                            int64_t ldp, const T* q, int64_t ldq,              \
                            const T* beta, const T* rho,                       \
                            const uint8_t* stop_status);
GKOC_DECL_CG(double, f64)
GKOC_DECL_CG(float, f32)
GKOC_DECL_CG(gkoc_c128, c128)
GKOC_DECL_CG(gkoc_c64, c64)

/* ----------------------------------------------------------------- GMRES
 * gmres::{restart,multi_axpy,multi_dot}  core/solver/gmres_kernels.hpp:23-45,
 * common_gmres::{initialize,hessenberg_qr,solve_krylov}
 * core/solver/common_gmres_kernels.hpp:23-48; semantics
 * reference/solver/gmres_kernels.cpp:26-100, common_gmres_kernels.cpp:28-193.
 * krylov_bases: ((krylov_dim+1)*rows) x nrhs, basis i in rows [i*rows,(i+1)*rows);
 * hessenberg entry H(i,j) of column k at hessenberg[j*ld_h + i*nrhs + k]
 * (core/solver/gmres.cpp:352-363, :540-545); final_iter_nums is a size_type
 * (uint64) array; residual_norm is 1 x nrhs.  multi_dot: hessenberg_col(d,k) =
 * <basis_d(:,k), next_krylov(:,k)> for d < num_dots, deterministic tree
 * (tolerance 1e-13); everything else is bit-identical to the reference. */
size_t gkoc_gmres_multi_dot_workspace_bytes(int64_t rows, int64_t nrhs,
                                            int64_t num_dots, size_t value_size);
/* (R = remove_complex<T>: residual_norm is real also for complex value types) */
#define GKOC_DECL_GMRES(T, TN, R)                                                 \
    int gkoc_common_gmres_initialize_##TN(                                     \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, const T* b, int64_t ldb,  \
        T* residual, int64_t ldr, T* givens_sin, int64_t ld_sin,               \
        T* givens_cos, int64_t ld_cos, int64_t krylov_dim,                     \
        uint8_t* stop_status);                                                 \
    int gkoc_gmres_restart_##TN(                                               \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, const T* residual,        \
        int64_t ldr, const R* residual_norm, T* residual_norm_collection,      \
        T* krylov_bases, int64_t ldk, uint64_t* final_iter_nums);              \
    int gkoc_gmres_multi_axpy_##TN(                                            \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, const T* krylov_bases,    \
        int64_t ldk, const T* y, int64_t ldy, T* before_preconditioner,        \
        int64_t ldo, const uint64_t* final_iter_nums, uint8_t* stop_status);   \
    int gkoc_gmres_multi_dot_##TN(                                             \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t num_dots,         \
        const T* krylov_bases, int64_t ldk, const T* next_krylov, int64_t ldn, \
        T* hessenberg_col, int64_t ldh, void* work, size_t work_bytes);        \
    int gkoc_common_gmres_hessenberg_qr_##TN(                                  \
        gkoc_stream_t s, int64_t nrhs, T* givens_sin, int64_t ld_sin,          \
        T* givens_cos, int64_t ld_cos, R* residual_norm,                       \
        T* residual_norm_collection, int64_t ld_rnc, T* hessenberg_iter,       \
        int64_t ld_h, int64_t iter, uint64_t* final_iter_nums,                 \
        const uint8_t* stop_status);                                           \
    int gkoc_common_gmres_solve_krylov_##TN(                                   \
        gkoc_stream_t s, int64_t nrhs, const T* residual_norm_collection,      \
        int64_t ld_rnc, const T* hessenberg, int64_t ld_h, T* y, int64_t ldy,  \
        const uint64_t* final_iter_nums, const uint8_t* stop_status);
GKOC_DECL_GMRES(double, f64, double)
GKOC_DECL_GMRES(float, f32, float)
GKOC_DECL_GMRES(gkoc_c128, c128, double)
GKOC_DECL_GMRES(gkoc_c64, c64, float)

/* ---------------------------------------------------------------- IDR(s)
 * idr::{initialize, step_1, step_2, step_3, compute_omega}  core/solver/idr_kernels.hpp:22-72,
 * reference/solver/idr_kernels.cpp:27-290 (driver core/solver/idr.cpp:150-300).  Row-major, row
 * strides ld*: p = subspace_vectors s x n (P^H), m  s x (s nrhs), f / c  s x nrhs,
 * g / u  n x (s nrhs), g_k / v / residual / x  n x nrhs.  Updates in the reference's term order;
 * the dots <p_j, g_k> use a fixed two-level tree.  deterministic == 0: shadow vectors drawn on the
 * host from N(0,1) with a random seed, like the reference. */
#define GKOC_DECL_IDR(T, TN, R)                                                    \
    int gkoc_idr_initialize_##TN(gkoc_stream_t s, int64_t nrhs,                 \
                                 int64_t subspace_dim, T* m, int64_t ldm,       \
                                 int64_t n, T* subspace_vectors, int64_t ldp,   \
                                 int deterministic, uint8_t* stop_status);      \
    int gkoc_idr_step_1_##TN(gkoc_stream_t s, int64_t n, int64_t nrhs,          \
                             int64_t subspace_dim, int64_t k, const T* m,       \
                             int64_t ldm, const T* f, int64_t ldf,              \
                             const T* residual, int64_t ldr, const T* g,        \
                             int64_t ldg, T* c, int64_t ldc, T* v, int64_t ldv, \
                             const uint8_t* stop_status);                       \
    int gkoc_idr_step_2_##TN(gkoc_stream_t s, int64_t n, int64_t nrhs,          \
                             int64_t subspace_dim, int64_t k, const T* omega,   \
                             const T* preconditioned_vector, int64_t ldpv,      \
                             const T* c, int64_t ldc, T* u, int64_t ldu,        \
                             const uint8_t* stop_status);                       \
    int gkoc_idr_step_3_##TN(gkoc_stream_t s, int64_t n, int64_t nrhs,          \
                             int64_t subspace_dim, int64_t k, const T* p,       \
                             int64_t ldp, T* g, int64_t ldg, T* g_k,            \
                             int64_t ldgk, T* u, int64_t ldu, T* m, int64_t ldm, \
                             T* f, int64_t ldf, T* residual, int64_t ldr, T* x, \
                             int64_t ldx, const uint8_t* stop_status);          \
    int gkoc_idr_compute_omega_##TN(gkoc_stream_t s, int64_t nrhs, R kappa,     \
                                    const T* tht, const R* residual_norm,       \
                                    T* omega, const uint8_t* stop_status);
GKOC_DECL_IDR(double, f64, double)
GKOC_DECL_IDR(float, f32, float)
GKOC_DECL_IDR(gkoc_c128, c128, double)   /* kappa and residual_norm are real */
GKOC_DECL_IDR(gkoc_c64, c64, float)

/* ------------------------------------------------------------- CB-GMRES
 * cb_gmres::{restart, arnoldi, solve_krylov}  core/solver/cb_gmres_kernels.hpp:101-142,
 * reference/solver/cb_gmres_kernels.cpp:31-420 (driver core/solver/cb_gmres.cpp:205-480);
 * cb_gmres::initialize is gkoc_common_gmres_initialize.  The Krylov basis is the 3-d accessor range
 * (krylov_dim+1) x rows x nrhs of accessor/reduced_row_major.hpp / scaled_reduced_row_major.hpp:
 * element (k, r, c) at bases[k*st0 + r*st1 + c] in the storage type named by storage_kind,
 * GKOC_CB_KEEP (= value type), GKOC_CB_F32 (f64 only), GKOC_CB_F16 (gko::half), or an integer type
 * GKOC_CB_I64 (f64 only) / I32 / I16 times scalars[k*sst + c] (value = storage * scalar, storage =
 * trunc(value / scalar)).  arnoldi_norm is 3 x nrhs (row 0: eta * old norm, 1: norm, 2: inf-norm).
 * arnoldi runs classical Gram-Schmidt with up to two re-orthogonalisation rounds; the decision is
 * taken on the device per column (no host synchronisation); buffer_iter may be NULL. */
#define GKOC_CB_KEEP 0
#define GKOC_CB_F32 1
#define GKOC_CB_F16 2
#define GKOC_CB_I64 3
#define GKOC_CB_I32 4
#define GKOC_CB_I16 5
#define GKOC_DECL_CB_GMRES(T, TN)                                                \
    int gkoc_cb_gmres_restart_##TN(                                              \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t krylov_dim,         \
        const T* residual, int64_t ldr, T* residual_norm,                        \
        T* residual_norm_collection, int64_t ld_rnc, T* arnoldi_norm,            \
        int64_t ld_an, int storage_kind, void* bases, int64_t st0, int64_t st1,  \
        T* scalars, int64_t sst, T* next_krylov, int64_t ldn,                    \
        uint64_t* final_iter_nums);                                              \
    int gkoc_cb_gmres_arnoldi_##TN(                                              \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t iter,               \
        T* next_krylov, int64_t ldn, T* givens_sin, int64_t ld_sin,              \
        T* givens_cos, int64_t ld_cos, T* residual_norm,                         \
        T* residual_norm_collection, int64_t ld_rnc, int storage_kind,           \
        void* bases, int64_t st0, int64_t st1, T* scalars, int64_t sst,          \
        T* hessenberg_iter, int64_t ld_h, T* buffer_iter, int64_t ld_buf,        \
        T* arnoldi_norm, int64_t ld_an, uint64_t* final_iter_nums,               \
        const uint8_t* stop_status);                                             \
    int gkoc_cb_gmres_solve_krylov_##TN(                                         \
        gkoc_stream_t s, int64_t rows, int64_t nrhs,                             \
        const T* residual_norm_collection, int64_t ld_rnc, int storage_kind,     \
        const void* bases, int64_t st0, int64_t st1, const T* scalars,           \
        int64_t sst, const T* hessenberg, int64_t ld_h, T* y, int64_t ldy,       \
        T* before_preconditioner, int64_t ldo, const uint64_t* final_iter_nums);
GKOC_DECL_CB_GMRES(double, f64)
GKOC_DECL_CB_GMRES(float, f32)
/* complex value types (csrc/cb_gmres_complex.hip): the basis is stored as complex<double> (GKOC_CB_KEEP) or,
 * for complex<double> arithmetic, as complex<float> (GKOC_CB_F32) - the accessor reduced_row_major of
 * GKO_INSTANTIATE_FOR_EACH_CB_GMRES_TYPE; no scaled integer storage for complex values; residual_norm and
 * arnoldi_norm are real (R = remove_complex<T>).  Re-orthogonalisation as in the stock device kernels
 * (common/cuda_hip/solver/cb_gmres_kernels.cpp update_next_krylov_kernel): no conjugate on the basis in the
 * update.  Agrees with the ReferenceExecutor to rounding. */
#define GKOC_DECL_CB_GMRES_CX(T, TN, R)                                          \
    int gkoc_cb_gmres_restart_##TN(                                              \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t krylov_dim,         \
        const T* residual, int64_t ldr, R* residual_norm,                        \
        T* residual_norm_collection, int64_t ld_rnc, int storage_kind,           \
        void* bases, int64_t st0, int64_t st1, T* next_krylov, int64_t ldn,      \
        uint64_t* final_iter_nums);                                              \
    int gkoc_cb_gmres_arnoldi_##TN(                                              \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t iter,               \
        T* next_krylov, int64_t ldn, T* givens_sin, int64_t ld_sin,              \
        T* givens_cos, int64_t ld_cos, R* residual_norm,                         \
        T* residual_norm_collection, int64_t ld_rnc, int storage_kind,           \
        void* bases, int64_t st0, int64_t st1, T* hessenberg_iter, int64_t ld_h, \
        T* buffer_iter, int64_t ld_buf, R* arnoldi_norm, int64_t ld_an,          \
        uint64_t* final_iter_nums, const uint8_t* stop_status);                  \
    int gkoc_cb_gmres_solve_krylov_##TN(                                         \
        gkoc_stream_t s, int64_t rows, int64_t nrhs,                             \
        const T* residual_norm_collection, int64_t ld_rnc, int storage_kind,     \
        const void* bases, int64_t st0, int64_t st1, const T* hessenberg,        \
        int64_t ld_h, T* y, int64_t ldy, T* before_preconditioner, int64_t ldo,  \
        const uint64_t* final_iter_nums);
GKOC_DECL_CB_GMRES_CX(gkoc_c128, c128, double)
GKOC_DECL_CB_GMRES_CX(gkoc_c64, c64, float)

/* ------------------------------------------------------- stopping criteria
 * residual_norm::residual_norm, implicit_residual_norm::implicit_residual_norm,
 * set_all_statuses  (core/stop/residual_norm_kernels.hpp,
 * core/stop/criterion_kernels.hpp; reference/stop/residual_norm_kernels.cpp:27-90).
 * flags_dev: 2 device bytes of scratch (Ginkgo's array<bool> device_storage);
 * *all_converged / *one_changed are HOST bools written after a stream sync
 * (this is the solver's one host sync point per iteration).  Passing NULL for
 * BOTH host pointers selects the asynchronous form: the kernel is enqueued,
 * flags_dev[0] = all_converged and flags_dev[1] = one_changed stay on the
 * device and nothing is synchronised (the caller copies them when it wants
 * to; the step kernels are masked by stop_status, so running ahead of the
 * check does not change the result). */
#define GKOC_DECL_STOP(T, TN)                                                  \
    int gkoc_residual_norm_##TN(gkoc_stream_t s, int64_t cols, const T* tau,   \
                                const T* orig_tau, T rel_residual_goal,        \
                                uint8_t stopping_id, int set_finalized,        \
                                uint8_t* stop_status, uint8_t* flags_dev,      \
                                int* all_converged_host,                       \
                                int* one_changed_host);                        \
    int gkoc_implicit_residual_norm_##TN(                                      \
        gkoc_stream_t s, int64_t cols, const T* tau, const T* orig_tau,        \
        T rel_residual_goal, uint8_t stopping_id, int set_finalized,           \
        uint8_t* stop_status, uint8_t* flags_dev, int* all_converged_host,     \
        int* one_changed_host);
GKOC_DECL_STOP(double, f64)
GKOC_DECL_STOP(float, f32)
int gkoc_set_all_statuses(gkoc_stream_t s, int64_t cols, uint8_t stopping_id,
                          int set_finalized, uint8_t* stop_status);

/* ------------------------------------------------------------ block-Jacobi
 * jacobi::{find_blocks,generate,simple_apply,apply,invert_diagonal,
 * simple_scalar_apply,scalar_apply}  core/preconditioner/jacobi_kernels.hpp:18-86,
 * reference/preconditioner/jacobi_kernels.cpp:47-118 (find_blocks),
 * :314-411 (generate), :419-531 (apply).  Storage = Ginkgo's
 * block_interleaved_storage_scheme (include/ginkgo/core/preconditioner/jacobi.hpp:37-140):
 * element (r,c) of block b at
 *   group_offset*(b >> group_power) + block_offset*(b & (2^group_power-1)) + r + c*stride,
 * stride = block_offset << group_power.  Full-precision blocks only
 * (block_precisions == NULL); adaptive precision returns GKOC_E_NOT_SUPPORTED.
 * block_pointers / num_blocks are integer-exact vs the reference; the inverse
 * blocks and apply results are bit-identical (same pivoting, same operation
 * order, no FMA contraction). */
typedef struct gkoc_jacobi_scheme {
    int64_t block_offset;
    int64_t group_offset;
    uint32_t group_power;
} gkoc_jacobi_scheme;

#define GKOC_DECL_JACOBI(T, TN, I, IN)                                         \
    /* *num_blocks_host is HOST memory; block_ptrs has n_rows+1 entries */     \
    int gkoc_jacobi_find_blocks_##TN##_##IN(                                   \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, uint32_t max_block_size,                            \
        int64_t* num_blocks_host, I* block_ptrs);                              \
    int gkoc_jacobi_generate_##TN##_##IN(                                      \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, const T* vals, int64_t num_blocks,                  \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, T* blocks, T* conditioning /* may be NULL */);    \
    int gkoc_jacobi_simple_apply_##TN##_##IN(                                  \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,          \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks,       \
        const T* b, int64_t ldb, T* x, int64_t ldx, int64_t nrhs);             \
    int gkoc_jacobi_apply_##TN##_##IN(                                         \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,          \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks,       \
        const T* alpha, const T* b, int64_t ldb, const T* beta, T* x,          \
        int64_t ldx, int64_t nrhs);
GKOC_DECL_JACOBI(double, f64, int32_t, i32)
GKOC_DECL_JACOBI(double, f64, int64_t, i64)
GKOC_DECL_JACOBI(float, f32, int32_t, i32)
GKOC_DECL_JACOBI(float, f32, int64_t, i64)
/* complex blocks, uniform storage precision (block-wise / adaptive: GKOC_DECL_JACOBI_ADAPTIVE_ANY
 * below): max_block_size <= 32, untuned apply (one lane per row) */
GKOC_DECL_JACOBI(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_JACOBI(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_JACOBI(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_JACOBI(gkoc_c64, c64, int64_t, i64)
/* jacobi::transpose_jacobi and conj_transpose_jacobi for real value types
 * (core/preconditioner/jacobi_kernels.hpp; reference/preconditioner/
 * jacobi_kernels.cpp:597-627): every block transposed into out_blocks (same storage
 * scheme; precisions may be NULL = full precision, else one byte per block and the
 * entries keep their storage type).  Needed by Jacobi::transpose(), i.e. by Bicg. */
#define GKOC_DECL_JACOBI_TRANSPOSE(T, TN, I, IN)                               \
    int gkoc_jacobi_transpose_##TN##_##IN(                                     \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,          \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks,       \
        const uint8_t* precisions, T* out_blocks);
GKOC_DECL_JACOBI_TRANSPOSE(double, f64, int32_t, i32)
GKOC_DECL_JACOBI_TRANSPOSE(double, f64, int64_t, i64)
GKOC_DECL_JACOBI_TRANSPOSE(float, f32, int32_t, i32)
GKOC_DECL_JACOBI_TRANSPOSE(float, f32, int64_t, i64)
/* ... of complex blocks (uniform storage precision); conj != 0: conj_transpose_jacobi */
#define GKOC_DECL_CJACOBI_TRANSPOSE(T, TN, I, IN)                              \
    int gkoc_cjacobi_transpose_##TN##_##IN(                                    \
        gkoc_stream_t s, int64_t num_blocks, gkoc_jacobi_scheme scheme,        \
        const I* block_ptrs, const T* blocks, int conj, T* out_blocks);
GKOC_DECL_CJACOBI_TRANSPOSE(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CJACOBI_TRANSPOSE(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CJACOBI_TRANSPOSE(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CJACOBI_TRANSPOSE(gkoc_c64, c64, int64_t, i64)

/* Block-Jacobi with a fixed reduced storage precision (Jacobi::storage_optimization
 * other than autodetect; include/ginkgo/core/preconditioner/jacobi.hpp, storage types
 * of core/preconditioner/jacobi_utils.hpp:15-37), value type double.  `precision` is
 * the precision_reduction byte (preserving << 4 | nonpreserving): 0x01 float,
 * 0x02 half, 0x10 / 0x20 upper 32 / 16 bits of the double, 0x11 upper 16 bits of the
 * float, 0 = full precision.  gkoc_jacobi_convert_storage narrows, in place, the
 * blocks gkoc_jacobi_generate produced (the reference converts while it stores,
 * reference/preconditioner/jacobi_kernels.cpp:393-408: same values); apply_stored
 * widens each entry on load and multiplies in double (apply_block with its
 * default_converter, :413-470).  alpha = beta = NULL: x = M b, else
 * x = alpha M b + beta x.  Bit-identical to the reference.  Layouts with 64-wide
 * groups (max_block_size <= 16), one right-hand side, unit strides. */
/* jacobi::initialize_precisions (core/preconditioner/jacobi_kernels.hpp:120-123):
 * precisions[i] = source[i % source_size], one precision_reduction byte per block */
int gkoc_jacobi_initialize_precisions(gkoc_stream_t s, const uint8_t* source,
                                      int64_t source_size, uint8_t* precisions,
                                      int64_t n);
int gkoc_jacobi_convert_storage_f64(gkoc_stream_t s, int64_t num_blocks,
                                    gkoc_jacobi_scheme scheme, double* blocks,
                                    uint8_t precision);
int gkoc_jacobi_apply_stored_f64_i32(
    gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,
    gkoc_jacobi_scheme scheme, const int32_t* block_ptrs, const double* blocks,
    uint8_t precision, const double* alpha, const double* b, int64_t ldb,
    const double* beta, double* x, int64_t ldx, int64_t nrhs);
int gkoc_jacobi_apply_stored_f64_i64(
    gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,
    gkoc_jacobi_scheme scheme, const int64_t* block_ptrs, const double* blocks,
    uint8_t precision, const double* alpha, const double* b, int64_t ldb,
    const double* beta, double* x, int64_t ldx, int64_t nrhs);

/* Adaptive block-Jacobi (Jacobi::storage_optimization block-wise or autodetect),
 * value type double: jacobi::generate with a precision array
 * (core/preconditioner/jacobi_kernels.hpp:30-47; reference/preconditioner/
 * jacobi_kernels.cpp:313-411).  precisions[b] in: the requested precision_reduction
 * byte of block b, 0xff = autodetect (condition number x unit round-off of the
 * storage type < accuracy, and - for float / half - the rounded inverse must still be
 * invertible with a sane condition number, :280-307); out: the precision of its storage
 * group (all blocks of a group share one: the best one every block supports,
 * core/preconditioner/jacobi_utils.hpp:104-176).  conditioning[b] (may be NULL) =
 * norm(block) * norm(inverse) as the reference computes it.  The blocks are stored in
 * the chosen types; apply_adaptive widens on load (alpha = beta = NULL: x = M b).
 * Decisions, condition numbers, stored blocks and apply results are bit-identical to
 * the reference.  64-wide storage groups (max_block_size in {1,2,4,8,16}), one
 * right-hand side with unit strides. */
#define GKOC_DECL_JACOBI_ADAPTIVE(I, IN)                                       \
    int gkoc_jacobi_generate_adaptive_f64_##IN(                                \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, const double* vals, int64_t num_blocks,             \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, double accuracy, uint8_t* precisions,             \
        double* conditioning, double* blocks);                                 \
    int gkoc_jacobi_apply_adaptive_f64_##IN(                                   \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,          \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const double* blocks,  \
        const uint8_t* precisions, const double* alpha, const double* b,       \
        int64_t ldb, const double* beta, double* x, int64_t ldx,               \
        int64_t nrhs);
GKOC_DECL_JACOBI_ADAPTIVE(int32_t, i32)
GKOC_DECL_JACOBI_ADAPTIVE(int64_t, i64)

/* ... for the value types float, complex<float>, complex<double> (the other instantiations of
 * jacobi::generate / simple_apply / apply / transpose_jacobi / conj_transpose_jacobi with a
 * precision array, core/preconditioner/jacobi_kernels.hpp:30-103).  Same meaning of the arguments;
 * accuracy and conditioning are of the component type.  The storage types follow the component
 * type (core/preconditioner/jacobi_utils.hpp:15-38 with include/ginkgo/core/base/math.hpp:365-383,
 * :546-582): for double components the five of the double path; for float components (0,1), (0,2),
 * (1,1) = half and (1,0), (2,0) = the upper 16 bits of the float; a complex entry is its two parts
 * in that type.  float: decisions, condition numbers, stored blocks and x = M b bit-identical to the
 * reference (tests/test_jacobi_types_gpu.py); complex: the same decisions, values to rounding (the
 * complex quotient, csrc/complex_type.hpp).  Any number of
 * right-hand sides and strides; 64-wide storage groups (max_block_size <= 32).
 * transpose_adaptive: out block = transpose (conj != 0: conjugate transpose) of the block, in
 * the storage type of its group (precisions == NULL: the value type). */
#define GKOC_DECL_JACOBI_ADAPTIVE_ANY(T, R, TN, I, IN)                         \
    int gkoc_jacobi_generate_adaptive_##TN##_##IN(                             \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, const T* vals, int64_t num_blocks,                  \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, R accuracy, uint8_t* precisions,                  \
        R* conditioning, T* blocks);                                           \
    int gkoc_jacobi_apply_adaptive_##TN##_##IN(                                \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,          \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks,       \
        const uint8_t* precisions, const T* alpha, const T* b, int64_t ldb,    \
        const T* beta, T* x, int64_t ldx, int64_t nrhs);                       \
    int gkoc_jacobi_transpose_adaptive_##TN##_##IN(                            \
        gkoc_stream_t s, int64_t num_blocks, gkoc_jacobi_scheme scheme,        \
        const I* block_ptrs, const T* blocks, const uint8_t* precisions,       \
        int conj, T* out_blocks);
GKOC_DECL_JACOBI_ADAPTIVE_ANY(float, float, f32, int32_t, i32)
GKOC_DECL_JACOBI_ADAPTIVE_ANY(float, float, f32, int64_t, i64)
GKOC_DECL_JACOBI_ADAPTIVE_ANY(gkoc_c128, double, c128, int32_t, i32)
GKOC_DECL_JACOBI_ADAPTIVE_ANY(gkoc_c128, double, c128, int64_t, i64)
GKOC_DECL_JACOBI_ADAPTIVE_ANY(gkoc_c64, float, c64, int32_t, i32)
GKOC_DECL_JACOBI_ADAPTIVE_ANY(gkoc_c64, float, c64, int64_t, i64)

#define GKOC_DECL_JACOBI_SCALAR(T, TN)                                         \
    /* inv_diag[i] = 1 / diag[i] */                                            \
    int gkoc_jacobi_invert_diagonal_##TN(gkoc_stream_t s, int64_t n,           \
                                         const T* diag, T* inv_diag);          \
    /* x = b .* inv_diag (row-wise) */                                         \
    int gkoc_jacobi_simple_scalar_apply_##TN(                                  \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* inv_diag,        \
        const T* b, int64_t ldb, T* x, int64_t ldx);                           \
    /* x = beta*x + alpha * b .* inv_diag */                                   \
    int gkoc_jacobi_scalar_apply_##TN(                                         \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* inv_diag,        \
        const T* alpha, const T* b, int64_t ldb, const T* beta, T* x,          \
        int64_t ldx);
GKOC_DECL_JACOBI_SCALAR(double, f64)
GKOC_DECL_JACOBI_SCALAR(float, f32)

/* ------------------------------------------------- benchmark stencil matrices
 * Device-side restatement of the reference's workload generator
 * benchmark/utils/stencil_matrix.hpp:68-238 (5/9-pt 2-D) and :264-453
 * (7/27-pt 3-D): x-fastest lexicographic numbering, diag = #points-1,
 * off-diagonal = -1, ascending columns.  Produces the CSR rows of the slab of
 * planes [z0, z0+nz) (nd == 2: grid rows) with GLOBAL column indices;
 * z0 = 0, nz = g gives the whole matrix.  Two calls: row_ptrs (+ nnz), then
 * fill.  Integer-exact vs the oracle generator. */
#define GKOC_DECL_STENCIL_PTRS(I, IN)                                          \
    int gkoc_stencil_row_ptrs_##IN(gkoc_stream_t s, int nd, int64_t g,         \
                                   int restricted, int64_t z0, int64_t nz,     \
                                   I* row_ptrs, int64_t* nnz_host);
GKOC_DECL_STENCIL_PTRS(int32_t, i32)
GKOC_DECL_STENCIL_PTRS(int64_t, i64)
#define GKOC_DECL_STENCIL_FILL(T, TN, I, IN)                                   \
    int gkoc_stencil_fill_##TN##_##IN(gkoc_stream_t s, int nd, int64_t g,      \
                                      int restricted, int64_t z0, int64_t nz,  \
                                      const I* row_ptrs, I* cols, T* vals);
GKOC_DECL_STENCIL_FILL(double, f64, int32_t, i32)
GKOC_DECL_STENCIL_FILL(double, f64, int64_t, i64)
GKOC_DECL_STENCIL_FILL(float, f32, int32_t, i32)
GKOC_DECL_STENCIL_FILL(float, f32, int64_t, i64)

/* ------------------------------------------------ row-partitioned matrices
 * What experimental::distributed::Matrix does on read and apply
 * (core/distributed/matrix.cpp:300-381 read_distributed ->
 * distributed_matrix::separate_local_nonlocal + index_map; :450-509 apply_impl).
 * A rank owns rows with GLOBAL column indices and the column range
 * [col_lo, col_hi).  split_count / split_fill produce
 *   - the local matrix (columns re-based to col_lo), CSR,
 *   - the non-local part as a ROW LIST: nl_rows[i] = local row, its entries
 *     nl_ptrs[i]..nl_ptrs[i+1] with columns in the dense halo index space,
 *   - recv_gidx[h] = global column of halo slot h (ascending), i.e. the rows
 *     this rank must receive (index_map's ordering for contiguous partitions).
 * col_map is caller-owned scratch of n_global_cols + 1 indices.  All integer
 * outputs are exact.  rowlist_spmv_add: y[nl_rows[i]] += A_nl(i,:) * halo,
 * accumulated in k order on top of y (== csr::advanced_spmv(1, A_nl, halo, 1, y),
 * matrix.cpp:498-507, restricted to the rows that have non-local entries). */
#define GKOC_DECL_DIST_IDX(I, IN)                                              \
    int gkoc_dist_split_count_##IN(                                            \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* cols,     \
        int64_t col_lo, int64_t col_hi, int64_t n_global_cols, I* col_map,     \
        I* local_row_ptrs, I* nl_row_ptrs_full, int64_t* n_halo_host,          \
        int64_t* nnz_local_host, int64_t* nnz_nl_host,                         \
        int64_t* n_nl_rows_host);
GKOC_DECL_DIST_IDX(int32_t, i32)
GKOC_DECL_DIST_IDX(int64_t, i64)
/* Boundary rows as COMPLETE rows (one column; the fast path of contiguous partitions): the rows
 * `rows` (= nl_rows of the split) of the owned block with ALL their entries in the original
 * column order, a column index < halo_base meaning the rank's own vector and halo_base + h halo
 * entry h (halo_base >= n_local = col_hi - col_lo: n_local itself, or n_local rounded up so that
 * a halo stored behind the local vector starts on a 128-byte boundary).  gkoc_csr_rowlist_spmv_full: y[rows[i]] = the k-ordered sum over
 * that row - the single-domain row sum bit for bit.  These rows need nothing from the local
 * SpMV: that one is launched over the interior row range only, and this kernel runs on the
 * exchange's stream right behind the halo, overlapped with it (gkoc_comm_exchange_join). */
/* The distributed product in ONE kernel (replaces the local SpMV + the non-local SpMV of
 * distributed::Matrix::apply, core/distributed/matrix.cpp:450-509, for contiguous partitions):
 * row_ptrs / col_idxs / vals = the rank's LOCAL block (n_rows rows, local column indices), whose
 * rows [head_rows, n_rows - tail_rows) read no halo entry; bnd_* = the first head_rows and the last
 * tail_rows rows as COMPLETE rows over [local columns | halo] in the original column order
 * (gkoc_dist_boundary_count_* / gkoc_dist_boundary_fill_*: head rows, then tail rows); b = the
 * local vector with the halo BEHIND it in the same array (unit stride; PRECONDITION of the cheap
 * gate: b itself on a 128-byte boundary and the halo at a multiple of 128 bytes behind it, so that no
 * cache line holds local entries and halo entries - a b that is not aligned makes every boundary wave
 * pay the agent-scope acquire, as GKOC_TUNE_GATE_FENCE=1 does); c the local result.  The boundary rows are computed
 * by the last waves of the grid, which wait for gkoc_gate_open(.., epoch) (enqueued on the
 * exchange's stream behind the transfer that fills b's halo part): no second kernel beside the
 * local SpMV, no event its stream waits for, no second copy of the matrix.  gate: two uint32 in
 * device memory, zero at the start (gate[0] the number of the last exchange that arrived, gate[1]
 * set to 1 by a wave that waited ~10 s in vain - the result is then undefined, check it); epoch:
 * the number of the exchange (1, 2, ... - the caller counts; the product with the same number
 * waits for it).  fork_word / fork_number (may be NULL / 0): the kernel's first wave stores the
 * number into the word - "b is final on this stream" - for an exchange stream that waits for it
 * with gkoc_stream_fork_wait (gkoc_comm_fork_deferred): the fork in front of the exchange then
 * costs the main queue neither an event nor a kernel.  The two streams must not share a hardware
 * queue (give the exchange's stream another priority).
 * Results: complete rows in the original column order = the single-domain bits.
 * gkoc_csr_spmv_gated_fits: 1 if the boundary rows are few enough (64 rows per wave, at most 8
 * waiting waves per compute unit) for the waiting waves to leave the exchange's kernels room on
 * the device; otherwise the entry refuses (GKOC_E_NOT_SUPPORTED) and the caller runs the
 * stream-ordered product (gkoc_csr_rowlist_spmv_full_* behind gkoc_comm_exchange_join). */
#define GKOC_DECL_CSR_GATED(T, TN, I, IN)                                                              \
    int gkoc_csr_spmv_gated_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,            \
                                        const I* col_idxs, const T* vals, const I* bnd_ptrs,           \
                                        const I* bnd_cols, const T* bnd_vals, const T* b, T* c,        \
                                        int64_t head_rows, int64_t tail_rows, const uint32_t* gate,    \
                                        uint32_t epoch, uint32_t* fork_word, uint32_t fork_number);
/* gkoc_x_csr_spmv_gated_dot: the same product and dot_out = <b[0 .. n_rows), c> (the LOCAL part \
 * of <p, A p>, core/solver/cg.cpp:163-166) from the registers that hold the row sums: every wave \
 * leaves one partial sum, ONE launch folds them (fixed tree: the value does not depend on timing). \
 * work: gkoc_x_workspace_bytes(n_rows + 128, sizeof(T)) bytes.  c has the bits of the plain product. */
#define GKOC_DECL_CSR_GATED_DOT(T, TN, I, IN)                                                          \
    int gkoc_x_csr_spmv_gated_dot_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,      \
                                              const I* col_idxs, const T* vals, const I* bnd_ptrs,     \
                                              const I* bnd_cols, const T* bnd_vals, const T* b, T* c,  \
                                              int64_t head_rows, int64_t tail_rows,                    \
                                              const uint32_t* gate, uint32_t epoch,                    \
                                              uint32_t* fork_word, uint32_t fork_number, T* dot_out,   \
                                              void* work, size_t work_bytes);
GKOC_DECL_CSR_GATED_DOT(double, f64, int32_t, i32)
GKOC_DECL_CSR_GATED_DOT(double, f64, int64_t, i64)
GKOC_DECL_CSR_GATED_DOT(float, f32, int32_t, i32)
GKOC_DECL_CSR_GATED_DOT(float, f32, int64_t, i64)
GKOC_DECL_CSR_GATED(double, f64, int32_t, i32)
GKOC_DECL_CSR_GATED(double, f64, int64_t, i64)
GKOC_DECL_CSR_GATED(float, f32, int32_t, i32)
GKOC_DECL_CSR_GATED(float, f32, int64_t, i64)
int gkoc_csr_spmv_gated_fits(int64_t n_rows, int64_t head_rows, int64_t tail_rows);
int gkoc_gate_open(gkoc_stream_t s, uint32_t* gate, uint32_t epoch);
/* Diagnostic: `blocks` workgroups of `threads` threads with `lds_bytes` of LDS each that stay on
 * the device for `usec` microseconds (tests of the gated product: a late exchange, an RCCL-sized
 * kernel that has to find room next to waiting waves). */
int gkoc_debug_delay(gkoc_stream_t s, int64_t usec, int blocks, int threads, int lds_bytes);
#define GKOC_DECL_DIST_BND_IDX(I, IN)                                          \
    int gkoc_dist_boundary_count_##IN(gkoc_stream_t s, int64_t n_list,         \
                                      const I* rows, const I* row_ptrs,        \
                                      I* out_ptrs, int64_t* nnz_host);
GKOC_DECL_DIST_BND_IDX(int32_t, i32)
GKOC_DECL_DIST_BND_IDX(int64_t, i64)
#define GKOC_DECL_DIST(T, TN, I, IN)                                           \
    int gkoc_dist_boundary_fill_##TN##_##IN(                                   \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* row_ptrs,     \
        const I* cols, const T* vals, int64_t col_lo, int64_t col_hi,          \
        int64_t halo_base, const I* col_map, const I* out_ptrs, I* out_cols,   \
        T* out_vals);                                                          \
    int gkoc_csr_rowlist_spmv_full_##TN##_##IN(                                \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* ptrs,         \
        const I* cols, const T* vals, int64_t halo_base, const T* x,           \
        const T* halo, T* y);                                                  \
    int gkoc_dist_split_fill_##TN##_##IN(                                      \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* cols,     \
        const T* vals, int64_t col_lo, int64_t col_hi, int64_t n_global_cols,  \
        const I* col_map, const I* local_row_ptrs, const I* nl_row_ptrs_full,  \
        I* local_cols, T* local_vals, I* nl_rows, I* nl_ptrs, I* nl_cols,      \
        T* nl_vals, I* recv_gidx);                                             \
    int gkoc_csr_rowlist_spmv_add_##TN##_##IN(                                 \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* ptrs,         \
        const I* cols, const T* vals, const T* halo, int64_t ld_halo, T* y,    \
        int64_t ldy, int64_t nrhs);                                            \
    /* the same for one column, plus dot_inout += sum_rows x[row] * (what was    \
     * added to y[row]): the non-local part of <x, A x>, to go with              \
     * gkoc_x_csr_spmv_dot on the local block.  work: 8 bytes per 64 listed rows */ \
    int gkoc_x_csr_rowlist_spmv_add_dot_##TN##_##IN(                           \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* ptrs,         \
        const I* cols, const T* vals, const T* halo, T* y, const T* x,         \
        T* dot_inout, void* work, size_t work_bytes);
GKOC_DECL_DIST(double, f64, int32_t, i32)
GKOC_DECL_DIST(double, f64, int64_t, i64)
GKOC_DECL_DIST(float, f32, int32_t, i32)
GKOC_DECL_DIST(float, f32, int64_t, i64)

/* ------------------------------------------------- Diagonal, SparsityCsr, components
 * matrix::Diagonal (core/matrix/diagonal_kernels.hpp; reference/matrix/diagonal_kernels.cpp:
 * 20-170): c = diag * b (or its inverse), c = b * diag, the same on the values of a Csr whose
 * structure the caller has copied, Diagonal -> Csr, device_matrix_data -> Diagonal.
 * SparsityCsr (core/matrix/sparsity_csr_kernels.hpp): the number of diagonal entries per row
 * (the caller scans it: diagonal_element_prefix_sum) and the adjacency structure without them.
 * components::reduce_add_array (val[0] += sum(arr)), components::convert_precision. */
#define GKOC_DECL_MISC_T(T, TN)                                                                  \
    int gkoc_diagonal_apply_to_dense_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,           \
                                          const T* diag, const T* b, int64_t ldb, T* c,          \
                                          int64_t ldc, int inverse);                             \
    int gkoc_diagonal_right_apply_to_dense_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,     \
                                                const T* diag, const T* b, int64_t ldb, T* c,    \
                                                int64_t ldc);                                    \
    int gkoc_reduce_add_array_##TN(gkoc_stream_t s, int64_t n, const T* arr, T* val);
GKOC_DECL_MISC_T(double, f64)
GKOC_DECL_MISC_T(float, f32)
GKOC_DECL_MISC_T(gkoc_c128, c128)
GKOC_DECL_MISC_T(gkoc_c64, c64)
int gkoc_reduce_add_array_i32(gkoc_stream_t s, int64_t n, const int32_t* arr, int32_t* val);
int gkoc_reduce_add_array_i64(gkoc_stream_t s, int64_t n, const int64_t* arr, int64_t* val);
int gkoc_reduce_add_array_u64(gkoc_stream_t s, int64_t n, const uint64_t* arr, uint64_t* val);
int gkoc_convert_precision_f32_f64(gkoc_stream_t s, int64_t n, const float* in, double* out);
int gkoc_convert_precision_f64_f32(gkoc_stream_t s, int64_t n, const double* in, float* out);
#define GKOC_DECL_MISC_TI(T, TN, I, IN)                                                          \
    int gkoc_diagonal_apply_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const T* diag,   \
                                               const I* row_ptrs, T* vals, int inverse);         \
    int gkoc_diagonal_right_apply_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t nnz,               \
                                                     const T* diag, const I* cols, T* vals);     \
    int gkoc_diagonal_convert_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n, const T* diag,      \
                                                 I* row_ptrs, I* cols, T* vals);                 \
    int gkoc_diagonal_fill_in_matrix_data_##TN##_##IN(gkoc_stream_t s, int64_t nnz,              \
                                                      const I* rows, const I* cols,              \
                                                      const T* vals, T* diag);
GKOC_DECL_MISC_TI(double, f64, int32_t, i32)
GKOC_DECL_MISC_TI(double, f64, int64_t, i64)
GKOC_DECL_MISC_TI(float, f32, int32_t, i32)
GKOC_DECL_MISC_TI(float, f32, int64_t, i64)
GKOC_DECL_MISC_TI(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_MISC_TI(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_MISC_TI(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_MISC_TI(gkoc_c64, c64, int64_t, i64)
#define GKOC_DECL_MISC_I(I, IN)                                                                  \
    int gkoc_sparsity_csr_count_diagonal_##IN(gkoc_stream_t s, int64_t n_rows,                   \
                                              const I* row_ptrs, const I* cols, I* counts);      \
    int gkoc_sparsity_csr_remove_diagonal_##IN(gkoc_stream_t s, int64_t n_rows,                  \
                                               const I* row_ptrs, const I* cols,                 \
                                               const I* diag_prefix_sum, I* adj_ptrs,            \
                                               I* adj_idxs);
GKOC_DECL_MISC_I(int32_t, i32)
GKOC_DECL_MISC_I(int64_t, i64)

/* ----------------------------------------------- distributed set-up kernels
 * What experimental::distributed::{Partition, index_map, Matrix / Vector::read_distributed,
 * assemble_rows_from_neighbors} run on their executor (core/distributed/{partition,
 * partition_helpers,index_map,matrix,vector,assembly}_kernels.hpp; semantics:
 * reference/distributed/ *_kernels.cpp and partition_helpers.hpp).  Index work and copies only:
 * values are moved as words of value_size = 4, 8 or 16 bytes (float, double, complex<float>,
 * complex<double>).  Suffixes: <L>_<G> = local / global index type (i32_i32, i32_i64, i64_i64).
 * Outputs whose length depends on the data come from a _count call (sizes to the host, an opaque
 * state that the matching _fill call consumes and releases). */
typedef struct gkoc_partition {
    int64_t num_ranges;
    int32_t num_parts;
    const void* range_bounds;           /* G[num_ranges + 1] */
    const int32_t* part_ids;            /* [num_ranges]: the part that owns each range */
    const void* range_starting_indices; /* L[num_ranges] */
    const void* part_sizes;             /* L[num_parts] */
} gkoc_partition;
#define GKOC_DECL_DIST_LG(L, LN, G, GN)                                                              \
    /* distributed_matrix::separate_local_nonlocal (matrix_kernels.hpp:22-42): the entries of the  \
     * rows of local_part, split by the owner of their column, input order kept */                 \
    int gkoc_dist_separate_local_nonlocal_count_##LN##_##GN(                                        \
        gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const gkoc_partition* row_part, \
        const gkoc_partition* col_part, int32_t local_part, void** state, int64_t* n_local,         \
        int64_t* n_non_local);                                                                      \
    int gkoc_dist_separate_local_nonlocal_fill_##LN##_##GN(                                         \
        gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const void* vals,               \
        size_t value_size, const gkoc_partition* row_part, const gkoc_partition* col_part,          \
        void* state, L* local_rows, L* local_cols, void* local_vals, L* non_local_rows,             \
        G* non_local_cols, void* non_local_vals);                                                   \
    /* distributed_vector::build_local (vector_kernels.hpp:23-33) */                                \
    int gkoc_dist_vector_build_local_##LN##_##GN(                                                   \
        gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const void* vals,               \
        size_t value_size, const gkoc_partition* part, int32_t local_part, void* local_values,      \
        int64_t ld);                                                                                \
    /* index_map::build_mapping / map_to_local / map_to_global (index_map_kernels.hpp:40-102);      \
     * index_space: 0 local, 1 non_local, 2 combined */                                             \
    int gkoc_index_map_build_mapping_count_##LN##_##GN(                                             \
        gkoc_stream_t s, int64_t n, const G* recv_connections, const gkoc_partition* part,          \
        void** state, int64_t* n_unique, int64_t* n_part_unique);                                   \
    int gkoc_index_map_build_mapping_fill_##LN##_##GN(                                              \
        gkoc_stream_t s, const gkoc_partition* part, void* state, int32_t* part_ids,                \
        L* remote_local_idxs, G* remote_global_idxs, int64_t* remote_sizes);                        \
    int gkoc_index_map_map_to_local_##LN##_##GN(                                                    \
        gkoc_stream_t s, int64_t n, const G* global_ids, const gkoc_partition* part,                \
        int64_t n_targets, const int32_t* remote_target_ids, const G* remote_global_flat,           \
        const int64_t* remote_offsets, int32_t rank, int index_space, L* local_ids);                \
    int gkoc_index_map_map_to_global_##LN##_##GN(                                                   \
        gkoc_stream_t s, int64_t n, const L* local_ids, const G* range_bounds,                      \
        const L* starting_indices, int64_t local_size, const uint64_t* local_ranges,                \
        int64_t n_local_ranges, const G* remote_global_flat, int64_t remote_size, int index_space,  \
        G* global_ids);                                                                             \
    /* partition::build_starting_indices (partition_kernels.hpp:42-51) */                           \
    int gkoc_partition_build_starting_indices_##LN##_##GN(                                          \
        gkoc_stream_t s, const G* range_offsets, const int32_t* range_parts, int64_t num_ranges,    \
        int32_t num_parts, int32_t* num_empty_parts, L* ranks, L* sizes);                           \
    /* assembly::count_non_owning_entries (assembly_kernels.hpp); send_count is added to */        \
    int gkoc_assembly_count_non_owning_entries_##LN##_##GN(                                         \
        gkoc_stream_t s, int64_t nnz, const G* rows, const gkoc_partition* part, int32_t local_part, \
        int32_t* send_count, G* send_positions, G* original_positions);
GKOC_DECL_DIST_LG(int32_t, i32, int32_t, i32)
GKOC_DECL_DIST_LG(int32_t, i32, int64_t, i64)
GKOC_DECL_DIST_LG(int64_t, i64, int64_t, i64)
#define GKOC_DECL_DIST_G(G, GN)                                                                      \
    int gkoc_partition_build_from_contiguous_##GN(gkoc_stream_t s, int64_t num_ranges,              \
                                                  const G* ranges, const int32_t* part_id_mapping,  \
                                                  G* range_bounds, int32_t* part_ids);              \
    int gkoc_partition_build_from_mapping_##GN(gkoc_stream_t s, int64_t n, const int32_t* mapping,  \
                                               G* range_bounds, int32_t* part_ids);                 \
    int gkoc_partition_build_ranges_from_global_size_##GN(gkoc_stream_t s, int32_t num_parts,       \
                                                          G global_size, G* ranges);                \
    /* partition_helpers (partition_helpers_kernels.hpp): (start, end) pairs per part */            \
    int gkoc_partition_helpers_sort_by_range_start_##GN(gkoc_stream_t s, int64_t num_parts,         \
                                                        G* range_start_ends, int32_t* part_ids);    \
    int gkoc_partition_helpers_check_consecutive_ranges_##GN(gkoc_stream_t s, int64_t num_parts,    \
                                                             const G* range_start_ends,             \
                                                             int* result);                          \
    int gkoc_partition_helpers_compress_ranges_##GN(gkoc_stream_t s, int64_t n_offsets,             \
                                                    const G* range_start_ends, G* range_offsets);   \
    int gkoc_assembly_fill_send_buffers_##GN(                                                       \
        gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const void* vals,               \
        size_t value_size, const G* send_positions, const G* original_positions, G* send_rows,      \
        G* send_cols, void* send_vals);
GKOC_DECL_DIST_G(int32_t, i32)
GKOC_DECL_DIST_G(int64_t, i64)
int gkoc_partition_count_ranges(gkoc_stream_t s, int64_t n, const int32_t* mapping,
                                int64_t* num_ranges);
int gkoc_partition_build_ranges_by_part(gkoc_stream_t s, const int32_t* range_parts,
                                        int64_t num_ranges, int32_t num_parts, uint64_t* range_ids,
                                        int64_t* sizes);
int gkoc_partition_has_ordered_parts(gkoc_stream_t s, int64_t num_ranges, const int32_t* part_ids,
                                     int* result);

/* ------------------------------------------------- extensions (gkoc_x_*)
 * NOT part of Ginkgo's kernel set (the shim never calls them): producer
 * kernels that also emit the reduction a Krylov loop needs next, so the
 * vector is not read again and two launches disappear per fused pair.  They
 * replace, inside this repository's own solver drivers, the pairs
 *   csr::spmv + dense::compute_dot            (cg.cpp:160-163: q = A p, beta = p.q)
 *   jacobi::simple_apply + dense::compute_dot (cg.cpp:133-136: z = M r, rho = r.z)
 *   cg::step_2 + dense::compute_norm2         (cg.cpp:167-171 + stop/residual_norm.cpp:120)
 * Vectors / c / x / r are bit-identical to the unfused kernels; the scalar is
 * a deterministic tree sum (tolerance 1e-13 as for compute_dot).  One column,
 * unit strides.  work: gkoc_x_workspace_bytes(n, sizeof(T)) device bytes. */
size_t gkoc_x_workspace_bytes(int64_t n, size_t value_size);
/* gkoc_x_gmres_multi_sub_scaled: next_krylov -= sum_{d<num} h(d,:) * basis_d, the
 * num dense::sub_scaled calls of the classical Gram-Schmidt update
 * (gmres.cpp:222-236) in one pass, term by term in d order => bit-identical. */
/* gkoc_x_gmres_mgs_step: one modified Gram-Schmidt step fused with the next dot
 * (gmres.cpp:176-190): next_krylov -= h_cur * basis_cur (bit-identical to
 * dense::sub_scaled), h_next = <basis_next, next_krylov> (tree sum). */
/* Optional last argument of the fused PipeCg step kernels (NULL: neither).
 * wait_word / wait_number: every workgroup waits until *wait_word has reached wait_number before
 * it reads the scalars - the word that gkoc_gate_open sets on the exchange's stream behind the
 * all-reduce that produces them (gkoc_comm_all_reduce_exchange_begin): the main stream then needs
 * no join in front of the step kernel (core/solver/pipe_cg.cpp:211-262 has the host wait for the
 * reduction there).  wait_word[1] is set to 1 by a workgroup that waited ~10 s in vain.
 * tau / orig_tau / goal / implicit / stopping_id / set_finalized / flags: the stopping criterion
 * gkoc_residual_norm_* (implicit = 0) or gkoc_implicit_residual_norm_* (1) of ONE column, evaluated
 * inside the step kernel before anything else and recorded in stop_status[0] (which must then be
 * writable) and flags[0..1] exactly as those entries do; a column that has converged is left
 * alone by the step.  tau and orig_tau have the kernel's value type. */
typedef struct gkoc_step_gate {
    const uint32_t* wait_word;
    uint32_t wait_number;
    int32_t implicit;
    const void* tau;
    const void* orig_tau;
    double goal;
    uint8_t* flags;
    uint8_t stopping_id;
    uint8_t set_finalized;
} gkoc_step_gate;
#define GKOC_DECL_X_GMRES(T, TN)                                               \
    int gkoc_x_gmres_mgs_step_##TN(                                            \
        gkoc_stream_t s, int64_t rows, T* next_krylov, const T* basis_cur,     \
        const T* h_cur, const T* basis_next, T* h_next, void* work,            \
        size_t work_bytes);                                                    \
    int gkoc_x_gmres_multi_sub_scaled_##TN(                                    \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t num,              \
        const T* krylov_bases, int64_t ldk, const T* h, int64_t ldh,           \
        T* next_krylov, int64_t ldn);
GKOC_DECL_X_GMRES(double, f64)
GKOC_DECL_X_GMRES(float, f32)
GKOC_DECL_X_GMRES(gkoc_c128, c128)
GKOC_DECL_X_GMRES(gkoc_c64, c64)
#define GKOC_DECL_X(T, TN)                                                     \
    int gkoc_x_cg_step_2_norm_##TN(                                            \
        gkoc_stream_t s, int64_t rows, T* x, T* r, const T* p, const T* q,     \
        const T* beta, const T* rho, const uint8_t* stop_status, T* norm_out,  \
        int take_sqrt, void* work, size_t work_bytes);                         \
    /* cg::step_1 (one column, unit strides) with the stopping criterion in    \
     * front of it: residual_norm (implicit = 0) / implicit_residual_norm      \
     * (implicit = 1: sqrt(|tau|)) <= goal * orig_tau is evaluated inside the  \
     * kernel, recorded in stop_status[0] and flags[0..1] exactly as           \
     * gkoc_*residual_norm_* does, and p is left alone if the column has       \
     * stopped - the criterion's kernel (core/solver/cg.cpp:150-160 between    \
     * the reductions and step_1) costs nothing of its own.  flags may be      \
     * pinned host memory. */                                                  \
    int gkoc_x_cg_step_1_check_##TN(                                           \
        gkoc_stream_t s, int64_t rows, T* p, const T* z, const T* rho,         \
        const T* prev_rho, const T* tau, const T* orig_tau, T goal,            \
        int implicit, uint8_t id, int set_finalized, uint8_t* stop_status,     \
        uint8_t* flags);                                                       \
    /* pipe_cg::step_2 of one iteration and step_1 of the next in one pass, with the partial sums \
     * of <r,z>, <w,z>, <r,r> (out3): ten vectors in, eight out; vectors bit-identical to the two \
     * kernels.  beta_in / beta_out: two different scalars (the caller alternates them). */       \
    int gkoc_x_pipe_cg_step_2_step_1_dots_##TN(                                \
        gkoc_stream_t s, int64_t rows, T* x, T* r, T* z, T* w, T* p, T* q,     \
        T* f, T* g, const T* m, const T* n, const T* prev_rho, const T* rho,   \
        const T* delta, const T* beta_in, T* beta_out,                         \
        const uint8_t* stop_status, T* out3, void* work, size_t work_bytes,    \
        const gkoc_step_gate* gate);                                           \
    /* pipe_cg::step_1 (one column, unit strides; vectors bit-identical) and     \
     * out3 = {<r,z>, <w,z>, <r,r>} of the updated vectors: the three values a   \
     * distributed PipeCg iteration all-reduces in one message */              \
    int gkoc_x_pipe_cg_step_1_dots_##TN(                                       \
        gkoc_stream_t s, int64_t rows, T* x, T* r, T* z, T* w, const T* p,     \
        const T* q, const T* f, const T* g, const T* rho, const T* beta,       \
        const uint8_t* stop_status, T* out3, void* work, size_t work_bytes);
GKOC_DECL_X(double, f64)
GKOC_DECL_X(float, f32)
/* The stopping criterion of one column in its SYNCHRONOUS form (gkoc_residual_norm_* / gkoc_implicit_residual_norm_*
 * with host results) and, enqueued right behind its kernel - BEFORE the host starts to wait for the answer -,
 * cg::step_1(p, z, rho, prev_rho) of the iteration that follows if the criterion lets the solve go on
 * (core/solver/cg.cpp:148-165: check, then step_1).  The step is masked by stop_status like every step kernel: a
 * column the criterion has just stopped is left alone.  For a caller that KNOWS step_1 comes next (the binding
 * for Ginkgo's core has seen it in this solve: gko_binding/fusion.cpp) - the device works through the 30 - 40 us
 * the host needs to read the answer and come back with the next launch. */
#define GKOC_DECL_X_CRIT(T, TN)                                                \
    int gkoc_x_residual_norm_then_cg_step_1_##TN(                              \
        gkoc_stream_t s, const T* tau, const T* orig_tau, T goal,              \
        uint8_t stopping_id, int set_finalized, int implicit,                  \
        uint8_t* stop_status, uint8_t* flags_dev, int* all_converged,          \
        int* one_changed, int64_t rows, T* p, const T* z, const T* rho,        \
        const T* prev_rho);
GKOC_DECL_X_CRIT(double, f64)
GKOC_DECL_X_CRIT(float, f32)
#define GKOC_DECL_XI(T, TN, I, IN)                                             \
    int gkoc_x_csr_spmv_dot_##TN##_##IN(                                       \
        gkoc_stream_t s, int64_t n, const I* row_ptrs, const I* col_idxs,      \
        const T* vals, const T* b, T* c, T* dot_out, void* work,               \
        size_t work_bytes);                                                    \
    int gkoc_x_jacobi_simple_apply_dot_##TN##_##IN(                            \
        gkoc_stream_t s, int64_t num_blocks, int64_t n_rows,                   \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, const T* blocks, const T* b, T* x, T* dot_out,    \
        void* work, size_t work_bytes);                                        \
    /* cg::step_2 and the preconditioner application of the NEXT iteration in    \
     * one kernel (cg.cpp:167-171 + :133-136): t = rho / beta, x += t p,          \
     * r -= t q, z = M r, rho_out = <r, z>, norm_out = ||r||^2 (or ||r|| with     \
     * take_sqrt).  x, r, z bit-identical to step_2 followed by simple_apply;     \
     * rho_out must not alias rho.  Fast-path block layout, one column. */        \
    int gkoc_x_cg_step_2_jacobi_apply_##TN##_##IN(                             \
        gkoc_stream_t s, int64_t num_blocks, int64_t n_rows,                   \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, const T* blocks, T* x, T* r, const T* p,          \
        const T* q, const T* beta, const T* rho, const uint8_t* stop_status,   \
        T* z, T* rho_out, T* norm_out, int take_sqrt, void* work,              \
        size_t work_bytes);
#define GKOC_DECL_XI2(T, TN, I, IN)                                            \
    /* PipeCg: pipe_cg::step_2 of one iteration, step_1 of the next, out3 =     \
     * {<r,z>, <w,z>, <r,r>} and m = M w (block-Jacobi, fast-path layout) in ONE \
     * kernel; vectors bit-identical to the separate kernels; beta_in / beta_out \
     * are two different scalars */                                            \
    int gkoc_x_pipe_cg_steps_jacobi_##TN##_##IN(                               \
        gkoc_stream_t s, int64_t num_blocks, int64_t n_rows,                   \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, const T* blocks, T* x, T* r, T* z, T* w, T* p,    \
        T* q, T* f, T* g, T* m, const T* n, const T* prev_rho, const T* rho,   \
        const T* delta, const T* beta_in, T* beta_out,                         \
        const uint8_t* stop_status, T* out3, void* work, size_t work_bytes,    \
        const gkoc_step_gate* gate);
GKOC_DECL_XI2(double, f64, int32_t, i32)
GKOC_DECL_XI2(double, f64, int64_t, i64)
GKOC_DECL_XI2(float, f32, int32_t, i32)
GKOC_DECL_XI2(float, f32, int64_t, i64)
GKOC_DECL_XI(double, f64, int32_t, i32)
GKOC_DECL_XI(double, f64, int64_t, i64)
GKOC_DECL_XI(float, f32, int32_t, i32)
GKOC_DECL_XI(float, f32, int64_t, i64)
/* 1 if gkoc_x_cg_step_2_jacobi_apply_* can run on this block layout: fast-path scheme and room for
 * its two rows of per-workgroup partial sums in gkoc_x_workspace_bytes(n_rows, value_size) (blocks
 * much smaller than max_block_size produce more partials than the workspace holds); 0 otherwise -
 * the caller then issues cg::step_2 and jacobi::simple_apply separately. */
int gkoc_x_cg_step_2_jacobi_apply_fits(int64_t num_blocks, int64_t n_rows,
                                       gkoc_jacobi_scheme scheme, size_t value_size);

/* ------------------------------------------------------ complex value types
 * complex<double> / complex<float> (the C++ standard library types Ginkgo uses) as plain pairs.  Only what moves or measures complex
 * data without multiplying it: device_matrix_data assembly (aos_to_soa / soa_to_aos / sort_row_major
 * / remove_zeros / sum_duplicates: a value is zero if both parts are, sums are component-wise),
 * fill_array / fill_seq_array, and the 2-norm of the columns of a complex Dense (what
 * stop::ResidualNorm needs).  SpMV, BLAS-1 with complex scalars and the solvers are real-valued. */
/* (gkoc_c128 / gkoc_c64: declared at the top of this file) */
int gkoc_remove_zeros_count_c128(gkoc_stream_t s, int64_t nnz, const gkoc_c128* vals, void* work,
                                 size_t work_bytes, int64_t* count_host);
int gkoc_remove_zeros_count_c64(gkoc_stream_t s, int64_t nnz, const gkoc_c64* vals, void* work,
                                size_t work_bytes, int64_t* count_host);
int gkoc_fill_array_c128(gkoc_stream_t s, gkoc_c128* data, int64_t n, gkoc_c128 value);
/* dense::fill of a complex matrix (the rows x cols part of a strided matrix) */
int gkoc_dense_fill_c128(gkoc_stream_t s, int64_t rows, int64_t cols, gkoc_c128* x, int64_t ldx,
                         gkoc_c128 value);
int gkoc_dense_fill_c64(gkoc_stream_t s, int64_t rows, int64_t cols, gkoc_c64* x, int64_t ldx,
                        gkoc_c64 value);
int gkoc_fill_array_c64(gkoc_stream_t s, gkoc_c64* data, int64_t n, gkoc_c64 value);
int gkoc_fill_seq_array_c128(gkoc_stream_t s, gkoc_c128* data, int64_t n);
int gkoc_fill_seq_array_c64(gkoc_stream_t s, gkoc_c64* data, int64_t n);
/* result[j] = sqrt(sum_i |x(i, j)|^2); ldx in complex elements; work: gkoc_reduction_workspace_bytes
 * (rows, cols, sizeof(real type)) */
int gkoc_dense_compute_norm2_c128(gkoc_stream_t s, int64_t rows, int64_t cols, const gkoc_c128* x,
                                  int64_t ldx, double* result, void* work, size_t work_bytes);
int gkoc_dense_compute_norm2_c64(gkoc_stream_t s, int64_t rows, int64_t cols, const gkoc_c64* x,
                                 int64_t ldx, float* result, void* work, size_t work_bytes);
/* Dense BLAS-1 on complex columns (csrc/complex_blas.hip) - the complex instantiations of
 * dense::{scale, inv_scale, add_scaled, sub_scaled, compute_dot, compute_conj_dot,
 * compute_squared_norm2, compute_mean, make_complex, get_real, get_imag, conj_transpose, row_gather,
 * fill_in_matrix_data} (core/matrix/dense_kernels.hpp:34-135, :236-262, :355-373; semantics
 * reference/matrix/dense_kernels.cpp:184-440, :832-841, :916-925, :1208-1250).  ld* in complex
 * elements.  alpha: 1 or `cols` scalars on the device, complex pairs, or reals when scalar_is_real
 * (Ginkgo's ScalarType = remove_complex<ValueType> instantiation). */
#define GKOC_DECL_CBLAS(P, TN, R)                                                                      \
    int gkoc_cdense_scale_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const void* alpha,         \
                               int64_t alpha_cols, int scalar_is_real, P* x, int64_t ldx);             \
    int gkoc_cdense_inv_scale_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const void* alpha,     \
                                   int64_t alpha_cols, int scalar_is_real, P* x, int64_t ldx);         \
    int gkoc_cdense_add_scaled_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const void* alpha,    \
                                    int64_t alpha_cols, int scalar_is_real, const P* x, int64_t ldx,   \
                                    P* y, int64_t ldy);                                                \
    int gkoc_cdense_sub_scaled_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const void* alpha,    \
                                    int64_t alpha_cols, int scalar_is_real, const P* x, int64_t ldx,   \
                                    P* y, int64_t ldy);                                                \
    /* result[j] = sum_i x(i,j) y(i,j), or sum_i conj(x(i,j)) y(i,j) when conjugate_x */               \
    int gkoc_cdense_compute_dot_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* x,          \
                                     int64_t ldx, const P* y, int64_t ldy, P* result,                  \
                                     int conjugate_x);                                                 \
    int gkoc_cdense_compute_squared_norm2_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,            \
                                               const P* x, int64_t ldx, R* result);                    \
    /* result[j] = sum_i x(i,j) (components::reduce_add_array adds it to its accumulator) */           \
    int gkoc_cdense_compute_sum_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* x,          \
                                     int64_t ldx, P* result);                                          \
    int gkoc_cdense_compute_mean_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* x,         \
                                      int64_t ldx, P* result);                                         \
    /* mode 0 make_complex (in: reals, out: pairs), 1 get_real, 2 get_imag (in: pairs, out: reals),    \
     * 3 conj_transpose (in rows x cols, out cols x rows, both pairs) */                               \
    int gkoc_cdense_convert_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const void* in,          \
                                 int64_t ld_in, void* out, int64_t ld_out, int mode);
GKOC_DECL_CBLAS(gkoc_c128, c128, double)
GKOC_DECL_CBLAS(gkoc_c64, c64, float)
#define GKOC_DECL_CBLAS_I(P, TN, I, IN)                                                                \
    int gkoc_cdense_row_gather_##TN##_##IN(gkoc_stream_t s, int64_t n_gather, int64_t cols,            \
                                           const I* rows, const P* orig, int64_t ld_orig, P* gathered, \
                                           int64_t ld_gathered);                                       \
    int gkoc_cdense_fill_in_matrix_data_##TN##_##IN(gkoc_stream_t s, int64_t nnz, const I* rows,       \
                                                    const I* cols, const P* vals, P* out, int64_t ld);
GKOC_DECL_CBLAS_I(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CBLAS_I(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CBLAS_I(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CBLAS_I(gkoc_c64, c64, int64_t, i64)
/* absolute values and 1-norms of complex columns (dense::inplace_absolute_dense,
 * outplace_absolute_dense, compute_norm1; |z| = hypot(re, im)) */
#define GKOC_DECL_CABS(P, TN, R)                                                                       \
    /* mode 0: x = |x| in place (imaginary parts 0), out unused; 1: out (reals, ld_out) = |x| */       \
    int gkoc_cdense_absolute_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, P* x, int64_t ldx,      \
                                  R* out, int64_t ld_out, int mode);                                   \
    int gkoc_cdense_compute_norm1_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* x,        \
                                       int64_t ldx, R* result);
GKOC_DECL_CABS(gkoc_c128, c128, double)
GKOC_DECL_CABS(gkoc_c64, c64, float)
/* CSR with complex values (csr::spmv / advanced_spmv / extract_diagonal / row_wise_absolute_sum of
 * core/matrix/csr_kernels.hpp for complex values): one thread per (row, right-hand side) - the complex
 * instantiations exist so that Ginkgo's distributed classes work for every value type its tests
 * instantiate, they are not a tuned path.  alpha == NULL: y = A x; else y = alpha[0] A x + beta[0] y.
 * row_scan mode 0: out[row] = a(row, row) where stored; 1: out[row] = sum_k |a(row, k)| */
#define GKOC_DECL_CCSR(P, TN, I, IN)                                                                   \
    int gkoc_ccsr_spmv_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t nrhs, const I* row_ptrs,     \
                                   const I* col_idxs, const P* vals, const P* alpha, const P* x,       \
                                   int64_t ldx, const P* beta, P* y, int64_t ldy);                     \
    int gkoc_ccsr_row_scan_##TN##_##IN(gkoc_stream_t s, int64_t rows, const I* row_ptrs,               \
                                       const I* col_idxs, const P* vals, P* out, int mode);
GKOC_DECL_CCSR(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CCSR(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CCSR(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CCSR(gkoc_c64, c64, int64_t, i64)
/* csr::row_wise_absolute_sum (core/matrix/csr_kernels.hpp:287-290) for real values: sums[row] =
 * sum_k |a(row, k)| (the L1 smoother of the Schwarz preconditioner) */
#define GKOC_DECL_RWAS(T, TN, I, IN)                                                                   \
    int gkoc_csr_row_wise_absolute_sum_##TN##_##IN(gkoc_stream_t s, int64_t rows, const I* row_ptrs,   \
                                                   const T* vals, T* sums);
GKOC_DECL_RWAS(double, f64, int32_t, i32)
GKOC_DECL_RWAS(double, f64, int64_t, i64)
GKOC_DECL_RWAS(float, f32, int32_t, i32)
GKOC_DECL_RWAS(float, f32, int64_t, i64)
/* scalar Jacobi and Diagonal products on complex values: jacobi::{invert_diagonal (a zero entry
 * inverts as one), simple_scalar_apply, scalar_apply}, diagonal::{apply_to_csr, right_apply_to_csr}
 * (core/preconditioner/jacobi_kernels.hpp:42-81, core/matrix/diagonal_kernels.hpp:33-43) */
#define GKOC_DECL_CJAC(P, TN)                                                                          \
    int gkoc_cjacobi_invert_diagonal_##TN(gkoc_stream_t s, int64_t n, const P* diag, P* inv);          \
    /* alpha == NULL: x = diag b (row-wise); else x = beta x + alpha b diag */                         \
    int gkoc_cjacobi_scalar_apply_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* diag,     \
                                       const P* alpha, const P* b, int64_t ldb, const P* beta, P* x,   \
                                       int64_t ldx);
GKOC_DECL_CJAC(gkoc_c128, c128)
GKOC_DECL_CJAC(gkoc_c64, c64)
#define GKOC_DECL_CCSR_SCALE(P, TN, I, IN)                                                             \
    /* mode 0: vals *= diag[row]; 1: vals *= 1 / diag[row]; 2: vals *= diag[col] */                    \
    int gkoc_ccsr_scale_by_diagonal_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,    \
                                                const I* col_idxs, const P* diag, int mode, P* vals);
GKOC_DECL_CCSR_SCALE(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CCSR_SCALE(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CCSR_SCALE(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CCSR_SCALE(gkoc_c64, c64, int64_t, i64)
/* Coo with complex values (coo::spmv2 / advanced_spmv2; spmv / advanced_spmv clear or scale c
 * first): c += [alpha] A b entry by entry with atomic adds - the one place where the summation
 * order is not fixed */
#define GKOC_DECL_CCOO(P, TN, I, IN)                                                                   \
    int gkoc_ccoo_spmv2_##TN##_##IN(gkoc_stream_t s, int64_t nnz, int64_t nrhs, const I* rows,         \
                                    const I* cols, const P* vals, const P* alpha, const P* b,          \
                                    int64_t ldb, P* c, int64_t ldc);
GKOC_DECL_CCOO(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CCOO(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CCOO(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CCOO(gkoc_c64, c64, int64_t, i64)
/* Dense -> Csr with complex values: dense::count_nonzeros_per_row (out: int32 / int64 / uint64 by
 * out_bytes) and dense::convert_to_csr (row pointers are an input, as for the real types) */
int gkoc_cdense_count_nonzeros_per_row_c128(gkoc_stream_t s, int64_t rows, int64_t cols,
                                            const gkoc_c128* in, int64_t ld, void* out, int out_bytes);
int gkoc_cdense_count_nonzeros_per_row_c64(gkoc_stream_t s, int64_t rows, int64_t cols,
                                           const gkoc_c64* in, int64_t ld, void* out, int out_bytes);
#define GKOC_DECL_CDENSE_CSR(P, TN, I, IN)                                                             \
    int gkoc_cdense_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* in,       \
                                       int64_t ld, const I* row_ptrs, I* out_cols, P* out_vals);
GKOC_DECL_CDENSE_CSR(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CDENSE_CSR(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CDENSE_CSR(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CDENSE_CSR(gkoc_c64, c64, int64_t, i64)
/* csr::sort_by_column_index and csr::transpose for complex values (pairs are only moved; a
 * conj_transpose conjugates them afterwards: gkoc_cdense_convert mode 3 on the n x 1 array) */
#define GKOC_DECL_CCSR_MOVE(P, TN, I, IN)                                                              \
    int gkoc_csr_sort_by_column_index_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,  \
                                                  I* col_idxs, P* vals);                               \
    int gkoc_csr_transpose_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,                \
                                       const I* row_ptrs, const I* col_idxs, const P* vals,            \
                                       int64_t nnz, I* t_row_ptrs, I* t_col_idxs, P* t_vals,           \
                                       void* work, size_t work_bytes);
GKOC_DECL_CCSR_MOVE(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CCSR_MOVE(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CCSR_MOVE(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CCSR_MOVE(gkoc_c64, c64, int64_t, i64)
/* dense::compute_mean for real columns (core/matrix/dense_kernels.hpp:98-102): result[j] =
 * (sum_i x(i,j)) / rows */
int gkoc_dense_compute_mean_f64(gkoc_stream_t s, int64_t rows, int64_t cols, const double* x,
                                int64_t ldx, double* result);
int gkoc_dense_compute_mean_f32(gkoc_stream_t s, int64_t rows, int64_t cols, const float* x,
                                int64_t ldx, float* result);
#define GKOC_DECL_COMPLEX_MD(T, TN, I, IN)                                                            \
    int gkoc_aos_to_soa_##TN##_##IN(gkoc_stream_t s, int64_t nnz, const void* entries, I* row_idxs,   \
                                    I* col_idxs, T* vals);
GKOC_DECL_COMPLEX_MD(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_COMPLEX_MD(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_COMPLEX_MD(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_COMPLEX_MD(gkoc_c64, c64, int64_t, i64)
GKOC_DECL_ASSEMBLY(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_ASSEMBLY(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_ASSEMBLY(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_ASSEMBLY(gkoc_c64, c64, int64_t, i64)

/* ---------------------------------------------------- mixed-precision SpMV
 * csr::spmv / ell::spmv<MatrixValueType = float, InputValueType = OutputValueType = double>
 * (core/matrix/csr_kernels.hpp:34-52, ell_kernels.hpp:24-41; arithmetic_type = highest_precision =
 * double): the matrix values are STORED in float and widened as they are loaded, vectors, scalars
 * and every product and sum are double - 8 instead of 12 bytes per stored entry.  The result has the
 * bits of the double kernel applied to the widened values.  Several columns: one after the other. */
#define GKOC_DECL_MIXED(I, IN)                                                                        \
    int gkoc_csr_spmv_f32_f64_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const I* row_ptrs, \
                                   const I* col_idxs, const float* vals, const double* b,             \
                                   int64_t ldb, double* c, int64_t ldc, int64_t nrhs);                \
    int gkoc_csr_advanced_spmv_f32_f64_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,          \
                                            const double* alpha, const I* row_ptrs,                   \
                                            const I* col_idxs, const float* vals, const double* b,    \
                                            int64_t ldb, const double* beta, double* c, int64_t ldc,  \
                                            int64_t nrhs);                                            \
    int gkoc_ell_spmv_f32_f64_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t k,        \
                                   int64_t stride, const I* cols, const float* vals,                  \
                                   const double* b, int64_t ldb, double* c, int64_t ldc,              \
                                   int64_t nrhs);                                                     \
    int gkoc_ell_advanced_spmv_f32_f64_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,          \
                                            int64_t k, int64_t stride, const double* alpha,           \
                                            const I* cols, const float* vals, const double* b,        \
                                            int64_t ldb, const double* beta, double* c, int64_t ldc,  \
                                            int64_t nrhs);
GKOC_DECL_MIXED(int32_t, i32)
GKOC_DECL_MIXED(int64_t, i64)

/* Every (matrix, input, output) value-type triple of a core built with GINKGO_MIXED_PRECISION
 * (core/base/mixed_precision_types.hpp:15-120; csr_kernels.hpp:34-52, ell_kernels.hpp:24-41):
 * mt / it / ot are GKOC_VT_* codes, all real or all complex.  alpha points to ONE value of the matrix
 * type, beta to one of the output type (Dense<MatrixValueType>* alpha, Dense<OutputValueType>* beta);
 * both NULL: c = A b, both set: c = alpha A b + beta c.  arithmetic_type = the widest of the three;
 * values are widened on load, the result narrowed on the store (reference/matrix/csr_kernels.cpp:
 * 45-118, ell_kernels.cpp:27-125).  Uniform triples and (float, double, double) run the tuned
 * kernels above; the others two plain streaming kernels (csrc/mixed_precision.hip).  Real triples are
 * bit-identical to the reference, complex ones agree to rounding. */
#define GKOC_VT_F64 0
#define GKOC_VT_F32 1
#define GKOC_VT_C128 2
#define GKOC_VT_C64 3
#define GKOC_DECL_SPMV_MIXED(I, IN)                                                                   \
    int gkoc_csr_spmv_mixed_##IN(gkoc_stream_t s, int mt, int it, int ot, int64_t n_rows,             \
                                 int64_t n_cols, const void* alpha, const I* row_ptrs,                \
                                 const I* col_idxs, const void* vals, const void* b, int64_t ldb,     \
                                 const void* beta, void* c, int64_t ldc, int64_t nrhs);               \
    int gkoc_ell_spmv_mixed_##IN(gkoc_stream_t s, int mt, int it, int ot, int64_t n_rows,             \
                                 int64_t n_cols, int64_t num_stored_per_row, int64_t stride,          \
                                 const void* alpha, const I* col_idxs, const void* vals,              \
                                 const void* b, int64_t ldb, const void* beta, void* c, int64_t ldc,  \
                                 int64_t nrhs);
GKOC_DECL_SPMV_MIXED(int32_t, i32)
GKOC_DECL_SPMV_MIXED(int64_t, i64)
/* dense::row_gather / advanced_row_gather<ValueType, OutputType, IndexType> for two DIFFERENT
 * precisions (core/matrix/dense_kernels.hpp:284-295; reference/matrix/dense_kernels.cpp:915-950):
 * alpha == beta == NULL: out(i, j) = orig(rows[i], j) converted; else (both point to one ValueType
 * value) out(i, j) = type(alpha orig(rows[i], j)) + type(beta) type(out(i, j)), type = the wider. */
#define GKOC_DECL_ROW_GATHER_MIXED(I, IN)                                                             \
    int gkoc_dense_row_gather_mixed_##IN(gkoc_stream_t s, int vt, int ot, int64_t n_gather,           \
                                         int64_t cols, const void* alpha, const I* rows,              \
                                         const void* orig, int64_t ld_orig, const void* beta,         \
                                         void* out, int64_t ld_out);
GKOC_DECL_ROW_GATHER_MIXED(int32_t, i32)
GKOC_DECL_ROW_GATHER_MIXED(int64_t, i64)
/* coo::conj_array (x[i] = conj(x[i]); Coo::conj_transpose) and dense::add_scaled_identity<complex,
 * real>: m = beta m + alpha I with REAL device scalars on a complex matrix */
int gkoc_conj_array_c128(gkoc_stream_t s, int64_t n, gkoc_c128* x);
int gkoc_conj_array_c64(gkoc_stream_t s, int64_t n, gkoc_c64* x);
int gkoc_dense_add_scaled_identity_real_c128(gkoc_stream_t s, int64_t rows, int64_t cols, const double* alpha,
                                             const double* beta, gkoc_c128* m, int64_t ld);
int gkoc_dense_add_scaled_identity_real_c64(gkoc_stream_t s, int64_t rows, int64_t cols, const float* alpha,
                                            const float* beta, gkoc_c64* m, int64_t ld);

/* ------------------------------------------- conversions and matrix utilities
 * Everything Ginkgo's matrix classes ask the device for when a matrix moves between formats, and the
 * diagonal / transpose / 1-norm helpers (csrc/conversions.hip; reference/matrix/{dense,csr,coo,ell,
 * sellp,hybrid}_kernels.cpp, cited per kernel there).  Entry order of every output = the
 * reference's.  Padding of Ell / Sellp: value 0, column -1 (invalid_index).  slice_sets are
 * Ginkgo's size_type (uint64_t).  Row pointers of the *_to_csr / dense_to_* functions are inputs:
 * the caller counts and scans first, as the classes in core/matrix do. */
#define GKOC_DECL_CV_DENSE(T, TN)                                                                     \
    int gkoc_fill_seq_array_##TN(gkoc_stream_t s, T* data, int64_t n);                                \
    int gkoc_dense_transpose_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in,           \
                                  int64_t ldi, T* out, int64_t ldo);                                  \
    int gkoc_dense_extract_diagonal_##TN(gkoc_stream_t s, int64_t n, const T* in, int64_t ld,         \
                                         T* diag);                                                    \
    /* m = beta m + alpha I */                                                                        \
    int gkoc_dense_add_scaled_identity_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,              \
                                            const T* alpha, const T* beta, T* m, int64_t ld);         \
    /* y(i,i) += alpha diag[i] (subtract != 0: -=); nothing for alpha == 0 */                         \
    int gkoc_dense_add_scaled_diag_##TN(gkoc_stream_t s, int64_t n, const T* alpha, const T* diag,    \
                                        T* y, int64_t ldy, int subtract);                             \
    /* out: int32 / int64 / uint64 counts (out_bytes 4 or 8) */                                       \
    int gkoc_dense_count_nonzeros_per_row_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,           \
                                               const T* in, int64_t ld, void* out, int out_bytes);    \
    int gkoc_dense_max_nnz_per_row_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in,     \
                                        int64_t ld, uint64_t* result_host);                           \
    int gkoc_dense_compute_slice_sets_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in,  \
                                           int64_t ld, int64_t slice_size, int64_t stride_factor,     \
                                           uint64_t* slice_sets, uint64_t* slice_lengths);
/* (1-norms per column; real value types - the complex ones are gkoc_cdense_compute_norm1_*) */
int gkoc_dense_compute_norm1_f64(gkoc_stream_t s, int64_t rows, int64_t cols, const double* x, int64_t ldx,
                                 double* result, void* work, size_t work_bytes);
int gkoc_dense_compute_norm1_f32(gkoc_stream_t s, int64_t rows, int64_t cols, const float* x, int64_t ldx,
                                 float* result, void* work, size_t work_bytes);
GKOC_DECL_CV_DENSE(double, f64)
GKOC_DECL_CV_DENSE(float, f32)
GKOC_DECL_CV_DENSE(gkoc_c128, c128)
GKOC_DECL_CV_DENSE(gkoc_c64, c64)
int gkoc_fill_seq_array_u64(gkoc_stream_t s, uint64_t* data, int64_t n);
/* csr::spgemm_reuse / advanced_spgemm_reuse (alpha != NULL: c = alpha a b + beta d) and
 * csr::spgeam_numeric (c = alpha a + beta b) - core/matrix/csr_kernels.hpp:60-92: the VALUES of a
 * product / sum whose pattern (c_ptrs, c_cols; rows sorted by column) exists already; entries are
 * added in the reference's order (reference/matrix/csr_kernels.cpp:304-436, :474-499) */
#define GKOC_DECL_REUSE(T, TN, I, IN)                                                                  \
    int gkoc_csr_spgemm_reuse_##TN##_##IN(                                                             \
        gkoc_stream_t s, int64_t n_rows, const I* a_ptrs, const I* a_cols, const T* a_vals,            \
        const I* b_ptrs, const I* b_cols, const T* b_vals, const T* alpha, const T* beta,              \
        const I* d_ptrs, const I* d_cols, const T* d_vals, const I* c_ptrs, const I* c_cols,           \
        T* c_vals);                                                                                    \
    int gkoc_csr_spgeam_numeric_##TN##_##IN(                                                           \
        gkoc_stream_t s, int64_t n_rows, const T* alpha, const I* a_ptrs, const I* a_cols,             \
        const T* a_vals, const T* beta, const I* b_ptrs, const I* b_cols, const T* b_vals,             \
        const I* c_ptrs, T* c_vals);
GKOC_DECL_REUSE(double, f64, int32_t, i32)
GKOC_DECL_REUSE(double, f64, int64_t, i64)
GKOC_DECL_REUSE(float, f32, int32_t, i32)
GKOC_DECL_REUSE(float, f32, int64_t, i64)
GKOC_DECL_REUSE(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_REUSE(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_REUSE(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_REUSE(gkoc_c64, c64, int64_t, i64)
/* components::fill_array for bool / char / uint16 / uint32 arrays
 * (core/components/fill_array_kernels.hpp:18-21): elem_bytes 1, 2 or 4, value = low bytes of pattern */
int gkoc_fill_array_small(gkoc_stream_t s, void* data, int64_t n, int elem_bytes, uint32_t pattern);
#define GKOC_DECL_CV(T, TN, I, IN)                                                                    \
    /* out_vals == NULL: pattern only (SparsityCsr) */                                                \
    int gkoc_dense_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in,       \
                                      int64_t ld, const I* row_ptrs, I* out_cols, T* out_vals);       \
    int gkoc_dense_to_coo_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in,       \
                                      int64_t ld, const int64_t* row_ptrs, I* out_rows,               \
                                      I* out_cols, T* out_vals);                                      \
    /* the first ell_lim non-zeros of a row go to Ell (ell_k columns of storage, padded over the      \
     * whole stride), the rest - Hybrid, coo_row_ptrs != NULL - to Coo */                             \
    int gkoc_dense_to_ell_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in,       \
                                      int64_t ld, int64_t ell_k, int64_t ell_lim, int64_t stride,     \
                                      I* ell_cols, T* ell_vals, const int64_t* coo_row_ptrs,          \
                                      I* coo_rows, I* coo_cols, T* coo_vals);                         \
    int gkoc_dense_to_sellp_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in,     \
                                        int64_t ld, int64_t slice_size, const uint64_t* slice_sets,   \
                                        I* out_cols, T* out_vals);                                    \
    /* fill_in_dense: out is zero on entry; Coo adds (duplicates sum up) */                           \
    int gkoc_csr_fill_in_dense_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,        \
                                           const I* cols, const T* vals, T* out, int64_t ld);         \
    int gkoc_coo_fill_in_dense_##TN##_##IN(gkoc_stream_t s, int64_t nnz, const I* rows,               \
                                           const I* cols, const T* vals, T* out, int64_t ld);         \
    int gkoc_ell_fill_in_dense_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t ell_k,            \
                                           int64_t stride, const I* cols, const T* vals, T* out,      \
                                           int64_t ld);                                               \
    int gkoc_sellp_fill_in_dense_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t slice_size,     \
                                             const uint64_t* slice_sets, const I* cols,               \
                                             const T* vals, T* out, int64_t ld);                      \
    /* diag[r] = first stored (r, r) for r < n; rows without one keep what diag held */               \
    int gkoc_ell_extract_diagonal_##TN##_##IN(gkoc_stream_t s, int64_t n, int64_t ell_k,              \
                                              int64_t stride, const I* cols, const T* vals,           \
                                              T* diag);                                               \
    int gkoc_sellp_extract_diagonal_##TN##_##IN(gkoc_stream_t s, int64_t n, int64_t slice_size,       \
                                                const uint64_t* slice_sets, const I* cols,            \
                                                const T* vals, T* diag);                              \
    int gkoc_coo_extract_diagonal_##TN##_##IN(gkoc_stream_t s, int64_t nnz, const I* rows,            \
                                              const I* cols, const T* vals, T* diag);                 \
    /* vals = beta vals, diagonal entries += alpha */                                                 \
    int gkoc_csr_add_scaled_identity_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,  \
                                                 const I* cols, T* vals, const T* alpha,              \
                                                 const T* beta);                                      \
    int gkoc_ell_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t ell_k, int64_t stride,   \
                                    const I* cols, const T* vals, const I* row_ptrs, I* out_cols,     \
                                    T* out_vals);                                                     \
    int gkoc_sellp_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t slice_size,            \
                                      const uint64_t* slice_sets, const I* cols, const T* vals,       \
                                      const I* row_ptrs, I* out_cols, T* out_vals);                   \
    /* a row's Ell entries, then its Coo entries; *_row_ptrs: exclusive sums of the two parts'        \
     * row counts (n_rows + 1 each), out_row_ptrs = their sum */                                      \
    int gkoc_hybrid_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t ell_k,                \
                                       int64_t stride, const I* ell_cols, const T* ell_vals,          \
                                       const I* coo_cols, const T* coo_vals, const I* ell_row_ptrs,   \
                                       const I* coo_row_ptrs, I* out_row_ptrs, I* out_cols,           \
                                       T* out_vals);
GKOC_DECL_CV(double, f64, int32_t, i32)
GKOC_DECL_CV(double, f64, int64_t, i64)
GKOC_DECL_CV(float, f32, int32_t, i32)
GKOC_DECL_CV(float, f32, int64_t, i64)
GKOC_DECL_CV(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_CV(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_CV(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_CV(gkoc_c64, c64, int64_t, i64)
#define GKOC_DECL_CV_INDEX(I, IN)                                                                     \
    int gkoc_ell_count_nonzeros_per_row_##IN(gkoc_stream_t s, int64_t n_rows, int64_t ell_k,          \
                                             int64_t stride, const I* cols, I* out);                  \
    int gkoc_sellp_count_nonzeros_per_row_##IN(gkoc_stream_t s, int64_t n_rows, int64_t slice_size,   \
                                               const uint64_t* slice_sets, const I* cols, I* out);    \
    /* csr::check_diagonal_entries_exist: *missing_host = 1 if a row r < n has no entry (r, r).       \
     * Synchronises the stream. */                                                                    \
    int gkoc_csr_missing_diagonal_##IN(gkoc_stream_t s, int64_t n, const I* row_ptrs, const I* cols,  \
                                       int* missing_host);
GKOC_DECL_CV_INDEX(int32_t, i32)
GKOC_DECL_CV_INDEX(int64_t, i64)

/* Permutations (Dense / Csr ::permute, ::scale_permute, Permutation / ScaledPermutation).
 * dense_permute: si = row_perm[i] (NULL: i), sj = col_perm[j] (NULL: j); forward out(i, j) =
 * scale * in(si, sj), inverse out(si, sj) = in(i, j) / scale with scale = row_scale[si] * col_scale[sj]
 * or whichever of the two is given.  csr_permute: row_inverse 0: out row i = in row row_perm[i],
 * 1: out row row_perm[i] = in row i; new column = col_perm[old column] (not re-sorted);
 * scale_mode 0 none, 1 value * row_scale[source row], 2 value / (row_scale[new row] * col_scale[new
 * column]) with absent factors left out. */
#define GKOC_DECL_PERMUTE(T, TN, I, IN)                                                               \
    int gkoc_dense_permute_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in,      \
                                       int64_t ldi, T* out, int64_t ldo, const I* row_perm,           \
                                       const I* col_perm, const T* row_scale, const T* col_scale,     \
                                       int inverse);                                                  \
    /* out(i, :) = alpha in(rows_idx[i], :) + beta out(i, :) */                                       \
    int gkoc_dense_advanced_row_gather_##TN##_##IN(gkoc_stream_t s, int64_t n_gather, int64_t cols,   \
                                                   const T* alpha, const I* rows_idx, const T* in,    \
                                                   int64_t ldi, const T* beta, T* out, int64_t ldo);  \
    int gkoc_csr_permute_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* in_rp, const I* in_ci, \
                                     const T* in_v, const I* row_perm, int row_inverse,               \
                                     const I* col_perm, const T* row_scale, const T* col_scale,       \
                                     int scale_mode, I* out_rp, I* out_ci, T* out_v);                 \
    int gkoc_scaled_permutation_invert_##TN##_##IN(gkoc_stream_t s, int64_t n, const T* in_scale,     \
                                                   const I* perm, T* out_scale, I* out_perm);         \
    int gkoc_scaled_permutation_compose_##TN##_##IN(gkoc_stream_t s, int64_t n, const T* first_scale, \
                                                    const I* first, const T* second_scale,            \
                                                    const I* second, T* out_scale, I* out_perm);      \
    /* csr::calculate_nonzeros_per_row_in_span / compute_submatrix: rows [row0, row0 + n), columns    \
     * [col0, col1); out_rp = exclusive sums of counts (the caller scans) */                          \
    int gkoc_csr_count_in_span_##TN##_##IN(gkoc_stream_t s, int64_t n, int64_t row0, int64_t col0,    \
                                           int64_t col1, const I* in_rp, const I* in_ci, I* counts);  \
    int gkoc_csr_submatrix_##TN##_##IN(gkoc_stream_t s, int64_t n, int64_t row0, int64_t col0,        \
                                       int64_t col1, const I* in_rp, const I* in_ci, const T* in_v,   \
                                       const I* out_rp, I* out_ci, T* out_v);
GKOC_DECL_PERMUTE(double, f64, int32_t, i32)
GKOC_DECL_PERMUTE(double, f64, int64_t, i64)
GKOC_DECL_PERMUTE(float, f32, int32_t, i32)
GKOC_DECL_PERMUTE(float, f32, int64_t, i64)
GKOC_DECL_PERMUTE(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_PERMUTE(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_PERMUTE(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_PERMUTE(gkoc_c64, c64, int64_t, i64)
/* csr::calculate_nonzeros_per_row_in_index_set / compute_submatrix_from_index_set
 * (core/matrix/csr_kernels.hpp; reference/matrix/csr_kernels.cpp:772-812, 853-904; the reference's
 * own CUDA / HIP backend leaves both unimplemented): the rows of the row index set (subset j =
 * [row_begin[j], row_end[j]), first result row row_superset[j] = index_set::get_superset_indices())
 * restricted to the columns of the column index set, renumbered to col_superset[subset] + offset.
 * counts has n_result_rows entries; out_rp = their exclusive sums (the caller scans). */
#define GKOC_DECL_INDEX_SET(T, TN, I, IN)                                                             \
    int gkoc_csr_count_in_index_set_##TN##_##IN(                                                      \
        gkoc_stream_t s, int64_t n_result_rows, int64_t n_row_subsets, const I* row_begin,            \
        const I* row_superset, int64_t n_col_subsets, const I* col_begin, const I* col_end,           \
        int64_t col_set_size, const I* in_rp, const I* in_ci, I* counts);                             \
    int gkoc_csr_submatrix_from_index_set_##TN##_##IN(                                                \
        gkoc_stream_t s, int64_t n_result_rows, int64_t n_row_subsets, const I* row_begin,            \
        const I* row_superset, int64_t n_col_subsets, const I* col_begin, const I* col_end,           \
        const I* col_superset, int64_t col_set_size, const I* in_rp, const I* in_ci, const T* in_v,   \
        const I* out_rp, I* out_ci, T* out_v);
GKOC_DECL_INDEX_SET(double, f64, int32_t, i32)
GKOC_DECL_INDEX_SET(double, f64, int64_t, i64)
GKOC_DECL_INDEX_SET(float, f32, int32_t, i32)
GKOC_DECL_INDEX_SET(float, f32, int64_t, i64)
GKOC_DECL_INDEX_SET(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_INDEX_SET(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_INDEX_SET(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_INDEX_SET(gkoc_c64, c64, int64_t, i64)
#define GKOC_DECL_PERMUTATION(I, IN)                                                                  \
    /* out[perm[i]] = i;  out[i] = first[second[i]] */                                                \
    int gkoc_permutation_invert_##IN(gkoc_stream_t s, int64_t n, const I* perm, I* out);              \
    int gkoc_permutation_compose_##IN(gkoc_stream_t s, int64_t n, const I* first, const I* second,    \
                                      I* out);
GKOC_DECL_PERMUTATION(int32_t, i32)
GKOC_DECL_PERMUTATION(int64_t, i64)

/* SpGEMM / SpGEAM (csr::spgemm, advanced_spgemm, spgeam; reference/matrix/csr_kernels.cpp:156-300,
 * 425-468) as triplets: count gives where each row's contributions start (offsets: n_rows + 1
 * int64 on the device) and their number on the host; expand writes them as (row, column, value) -
 * first beta * D's entries (d_rp != NULL), then (alpha * a_ik) * b_kj in storage order, or, with
 * b_rp == NULL, alpha * a_ik themselves (SpGEAM: A's entries are "D", the second matrix is "A").
 * gkoc_sort_row_major + gkoc_sum_duplicates_* + gkoc_convert_idxs_to_ptrs turn the triplets into the
 * result: the reference's values (0 + contributions in the order they are met), pattern and order.
 * alpha / beta NULL: 1. */
int gkoc_csr_spgemm_count_i32(gkoc_stream_t s, int64_t n_rows, const int32_t* a_rp, const int32_t* a_ci,
                              const int32_t* b_rp, const int32_t* d_rp, int64_t* offsets,
                              int64_t* total_host);
int gkoc_csr_spgemm_count_i64(gkoc_stream_t s, int64_t n_rows, const int64_t* a_rp, const int64_t* a_ci,
                              const int64_t* b_rp, const int64_t* d_rp, int64_t* offsets,
                              int64_t* total_host);
#define GKOC_DECL_SPGEMM(T, TN, I, IN)                                                                \
    int gkoc_csr_spgemm_expand_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const T* alpha,           \
                                           const I* a_rp, const I* a_ci, const T* a_v, const I* b_rp, \
                                           const I* b_ci, const T* b_v, const T* beta, const I* d_rp, \
                                           const I* d_ci, const T* d_v, const int64_t* offsets,       \
                                           I* t_rows, I* t_cols, T* t_vals);
GKOC_DECL_SPGEMM(double, f64, int32_t, i32)
GKOC_DECL_SPGEMM(double, f64, int64_t, i64)
GKOC_DECL_SPGEMM(float, f32, int32_t, i32)
GKOC_DECL_SPGEMM(float, f32, int64_t, i64)
GKOC_DECL_SPGEMM(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_SPGEMM(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_SPGEMM(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_SPGEMM(gkoc_c64, c64, int64_t, i64)

/* L1 block-Jacobi (Jacobi::with_aggregate_l1; reference/preconditioner/jacobi_kernels.cpp:728-780,
 * reference/factorization/factorization_kernels.cpp:55-128): scalar_l1 adds to diag[r] the sum of
 * |a_rj|, j != r; block_l1 adds to the stored diagonal entry of every row the sum of |a_rj| over the
 * entries outside the row's diagonal block (both in storage order); add_diagonal_elements = shift
 * (which rows lack a diagonal entry, scanned; their number on the host) + fill (the matrix with an
 * explicit zero inserted before the first larger column of such a row). */
#define GKOC_DECL_L1(T, TN, I, IN)                                                                    \
    int gkoc_jacobi_scalar_l1_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* rp, const I* ci,  \
                                          const T* v, T* diag);                                       \
    int gkoc_jacobi_block_l1_##TN##_##IN(gkoc_stream_t s, int64_t num_blocks, const I* block_ptrs,    \
                                         const I* rp, const I* ci, T* v);                             \
    int gkoc_csr_add_diagonal_fill_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* rp,          \
                                               const I* ci, const T* v, const I* shift, I* new_rp,    \
                                               I* new_ci, T* new_v);
GKOC_DECL_L1(double, f64, int32_t, i32)
GKOC_DECL_L1(double, f64, int64_t, i64)
GKOC_DECL_L1(float, f32, int32_t, i32)
GKOC_DECL_L1(float, f32, int64_t, i64)
GKOC_DECL_L1(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_L1(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_L1(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_L1(gkoc_c64, c64, int64_t, i64)
int gkoc_csr_missing_diagonal_shift_i32(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,
                                        const int32_t* rp, const int32_t* ci, int32_t* shift,
                                        int64_t* missing_host);
int gkoc_csr_missing_diagonal_shift_i64(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,
                                        const int64_t* rp, const int64_t* ci, int64_t* shift,
                                        int64_t* missing_host);

/* ------------------------------------------------- COO SpMV, CSR -> Hybrid
 * coo::{spmv, advanced_spmv, spmv2, advanced_spmv2} (core/matrix/coo_kernels.hpp:24-58;
 * reference/matrix/coo_kernels.cpp:33-100): c = A b, c = alpha A b + beta c,
 * c += A b, c += alpha A b for a COO matrix.  Row indices sorted ascending (what
 * Ginkgo's Coo holds): bit-identical to the reference's entry-by-entry
 * accumulation, 16 B per stored entry of traffic, no atomics (csrc/coo.hip);
 * unsorted rows are detected on the device and handled with atomics (tolerance).
 * work: gkoc_coo_workspace_bytes(n_rows, sizeof(I), sizeof(T)) device bytes.
 * Hybrid needs nothing else for apply (ell::spmv then coo::spmv2, hybrid.cpp).
 * hybrid::compute_coo_row_ptrs (reference/matrix/hybrid_kernels.cpp:30-43) and
 * csr::convert_to_hybrid (reference/matrix/csr_kernels.cpp:911-955): the first
 * ell_lim entries of each row go to the ELL part (padding: value 0, column -1),
 * the rest to COO at coo_row_ptrs[row]...; index arrays bit-exact. */
size_t gkoc_coo_workspace_bytes(int64_t n_rows, size_t index_size,
                                size_t value_size);
/* components::convert_ptrs_to_idxs (core/components/format_conversion_kernels.hpp):
 * idxs[k] = row for ptrs[row] <= k < ptrs[row + 1] (Csr -> Coo) */
int gkoc_convert_ptrs_to_idxs_i32(gkoc_stream_t s, const int32_t* ptrs,
                                  int64_t n_rows, int32_t* idxs);
int gkoc_convert_ptrs_to_idxs_i64(gkoc_stream_t s, const int64_t* ptrs,
                                  int64_t n_rows, int64_t* idxs);
int gkoc_hybrid_compute_coo_row_ptrs(gkoc_stream_t s, int64_t n_rows,
                                     const uint64_t* row_nnz, uint64_t ell_lim,
                                     int64_t* coo_row_ptrs /* n_rows + 1 */);
#define GKOC_DECL_COO(T, TN, I, IN)                                            \
    int gkoc_coo_spmv_##TN##_##IN(                                             \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t nnz,          \
        const I* row_idxs, const I* col_idxs, const T* vals, const T* b,       \
        int64_t ldb, T* c, int64_t ldc, int64_t nrhs, void* work,              \
        size_t work_bytes);                                                    \
    int gkoc_coo_advanced_spmv_##TN##_##IN(                                    \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t nnz,          \
        const T* alpha, const I* row_idxs, const I* col_idxs, const T* vals,   \
        const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc,             \
        int64_t nrhs, void* work, size_t work_bytes);                          \
    int gkoc_coo_spmv2_##TN##_##IN(                                            \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t nnz,          \
        const I* row_idxs, const I* col_idxs, const T* vals, const T* b,       \
        int64_t ldb, T* c, int64_t ldc, int64_t nrhs, void* work,              \
        size_t work_bytes);                                                    \
    int gkoc_coo_advanced_spmv2_##TN##_##IN(                                   \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t nnz,          \
        const T* alpha, const I* row_idxs, const I* col_idxs, const T* vals,   \
        const T* b, int64_t ldb, T* c, int64_t ldc, int64_t nrhs, void* work,  \
        size_t work_bytes);
/* ell::copy (core/matrix/ell_kernels.hpp:53-56): the stored entries of an Ell into one with another
 * stride (Ell / Hybrid assignment, resize); csr::convert_to_hybrid.  All four value types. */
#define GKOC_DECL_COO_CONVERT(T, TN, I, IN)                                    \
    int gkoc_ell_copy_##TN##_##IN(                                             \
        gkoc_stream_t s, int64_t n_rows, int64_t k, int64_t src_stride,        \
        const I* src_cols, const T* src_vals, int64_t dst_stride,              \
        I* dst_cols, T* dst_vals);                                             \
    int gkoc_csr_convert_to_hybrid_##TN##_##IN(                                \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* col_idxs, \
        const T* vals, int64_t ell_lim, int64_t ell_stride, I* ell_cols,       \
        T* ell_vals, const int64_t* coo_row_ptrs, I* coo_rows, I* coo_cols,    \
        T* coo_vals);
GKOC_DECL_COO(double, f64, int32_t, i32)
GKOC_DECL_COO(double, f64, int64_t, i64)
GKOC_DECL_COO(float, f32, int32_t, i32)
GKOC_DECL_COO(float, f32, int64_t, i64)
GKOC_DECL_COO_CONVERT(double, f64, int32_t, i32)
GKOC_DECL_COO_CONVERT(double, f64, int64_t, i64)
GKOC_DECL_COO_CONVERT(float, f32, int32_t, i32)
GKOC_DECL_COO_CONVERT(float, f32, int64_t, i64)
GKOC_DECL_COO_CONVERT(gkoc_c128, c128, int32_t, i32)
GKOC_DECL_COO_CONVERT(gkoc_c128, c128, int64_t, i64)
GKOC_DECL_COO_CONVERT(gkoc_c64, c64, int32_t, i32)
GKOC_DECL_COO_CONVERT(gkoc_c64, c64, int64_t, i64)

/* csr::transpose / conj_transpose (real types) (core/matrix/csr_kernels.hpp,
 * GKO_DECLARE_CSR_TRANSPOSE_KERNEL; reference/matrix/csr_kernels.cpp:693-731): the
 * transposed matrix in CSR, entries of a row ordered as the reference orders them
 * (stable by original row, then storage order) - index arrays and values
 * bit-identical.  work: gkoc_csr_transpose_workspace_bytes(nnz, n_cols, sizeof(I)). */
size_t gkoc_csr_transpose_workspace_bytes(int64_t nnz, int64_t n_cols,
                                          size_t index_size);
#define GKOC_DECL_TRANSPOSE(T, TN, I, IN)                                      \
    int gkoc_csr_transpose_##TN##_##IN(                                        \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const I* row_ptrs,    \
        const I* col_idxs, const T* vals, int64_t nnz, I* t_row_ptrs,          \
        I* t_col_idxs, T* t_vals, void* work, size_t work_bytes);
GKOC_DECL_TRANSPOSE(double, f64, int32_t, i32)
GKOC_DECL_TRANSPOSE(double, f64, int64_t, i64)
GKOC_DECL_TRANSPOSE(float, f32, int32_t, i32)
GKOC_DECL_TRANSPOSE(float, f32, int64_t, i64)

/* ------------------------------------------------- other Krylov solvers
 * The fused vector updates of Bicgstab, Cgs, Fcg and PipeCg - the kernels
 * core/solver/{bicgstab,cgs,fcg,pipe_cg}.cpp issue through exec->run:
 *   bicgstab::{initialize, step_1, step_2, step_3, finalize}
 *       (core/solver/bicgstab_kernels.hpp:23-75; reference/solver/bicgstab_kernels.cpp:24-180)
 *   cgs::{initialize, step_1, step_2, step_3}
 *       (core/solver/cgs_kernels.hpp; reference/solver/cgs_kernels.cpp:24-146)
 *   fcg::{initialize, step_1, step_2}
 *       (core/solver/fcg_kernels.hpp; reference/solver/fcg_kernels.cpp:24-106)
 *   pipe_cg::{initialize_1, initialize_2, step_1, step_2}
 *       (core/solver/pipe_cg_kernels.hpp; reference/solver/pipe_cg_kernels.cpp:24-164)
 * Argument order = the reference kernel's, every rows x cols operand followed by
 * its leading dimension; scalars are device arrays of `cols` values; columns whose
 * stop_status has stopped are left untouched.  Results bit-identical to the
 * reference (same expressions, no contraction). */
#define GKOC_DECL_KRYLOV(T, TN) \
    int gkoc_bicgstab_initialize_##TN(gkoc_stream_t s, int64_t rows, int64_t \
        cols, const T* b, int64_t ldb, T* r, int64_t ldr, T* rr, int64_t \
        ldrr, T* y, int64_t ldy, T* sv, int64_t lds, T* t, int64_t ldt, T* \
        z, int64_t ldz, T* v, int64_t ldv, T* p, int64_t ldp, T* prev_rho, \
        T* rho, T* alpha, T* beta, T* gamma, T* omega, uint8_t* \
        stop_status); \
    int gkoc_bicgstab_step_1_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, \
        const T* r, int64_t ldr, T* p, int64_t ldp, const T* v, int64_t ldv, \
        const T* rho, const T* prev_rho, const T* alpha, const T* omega, \
        const uint8_t* stop_status); \
    int gkoc_bicgstab_step_2_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, \
        const T* r, int64_t ldr, T* sv, int64_t lds, const T* v, int64_t \
        ldv, const T* rho, T* alpha, const T* beta, const uint8_t* \
        stop_status); \
    int gkoc_bicgstab_step_3_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, \
        T* x, int64_t ldx, T* r, int64_t ldr, const T* sv, int64_t lds, \
        const T* t, int64_t ldt, const T* y, int64_t ldy, const T* z, \
        int64_t ldz, const T* alpha, const T* beta, const T* gamma, T* \
        omega, const uint8_t* stop_status); \
    int gkoc_bicgstab_finalize_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, \
        T* x, int64_t ldx, const T* y, int64_t ldy, const T* alpha, uint8_t* \
        stop_status); \
    int gkoc_cgs_initialize_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, \
        const T* b, int64_t ldb, T* r, int64_t ldr, T* r_tld, int64_t ldrt, \
        T* p, int64_t ldp, T* q, int64_t ldq, T* u, int64_t ldu, T* u_hat, \
        int64_t lduh, T* v_hat, int64_t ldvh, T* t, int64_t ldt, T* alpha, \
        T* beta, T* gamma, T* rho_prev, T* rho, uint8_t* stop_status); \
    int gkoc_cgs_step_1_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const \
        T* r, int64_t ldr, T* u, int64_t ldu, T* p, int64_t ldp, const T* q, \
        int64_t ldq, T* beta, const T* rho, const T* rho_prev, const \
        uint8_t* stop_status); \
    int gkoc_cgs_step_2_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const \
        T* u, int64_t ldu, const T* v_hat, int64_t ldvh, T* q, int64_t ldq, \
        T* t, int64_t ldt, T* alpha, const T* rho, const T* gamma, const \
        uint8_t* stop_status); \
    int gkoc_cgs_step_3_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const \
        T* t, int64_t ldt, const T* u_hat, int64_t lduh, T* r, int64_t ldr, \
        T* x, int64_t ldx, const T* alpha, const uint8_t* stop_status); \
    int gkoc_fcg_initialize_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, \
        const T* b, int64_t ldb, T* r, int64_t ldr, T* z, int64_t ldz, T* p, \
        int64_t ldp, T* q, int64_t ldq, T* t, int64_t ldt, T* prev_rho, T* \
        rho, T* rho_t, uint8_t* stop_status); \
    int gkoc_fcg_step_1_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, T* p, \
        int64_t ldp, const T* z, int64_t ldz, const T* rho_t, const T* \
        prev_rho, const uint8_t* stop_status); \
    int gkoc_fcg_step_2_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, T* x, \
        int64_t ldx, T* r, int64_t ldr, T* t, int64_t ldt, const T* p, \
        int64_t ldp, const T* q, int64_t ldq, const T* beta, const T* rho, \
        const uint8_t* stop_status); \
    int gkoc_pipe_cg_initialize_1_##TN(gkoc_stream_t s, int64_t rows, int64_t \
        cols, const T* b, int64_t ldb, T* r, int64_t ldr, T* prev_rho, \
        uint8_t* stop_status); \
    int gkoc_pipe_cg_initialize_2_##TN(gkoc_stream_t s, int64_t rows, int64_t \
        cols, T* p, int64_t ldp, T* q, int64_t ldq, T* f, int64_t ldf, T* g, \
        int64_t ldg, T* beta, const T* z, int64_t ldz, const T* w, int64_t \
        ldw, const T* m, int64_t ldm, const T* n, int64_t ldn, const T* \
        delta); \
    int gkoc_pipe_cg_step_1_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, T* \
        x, int64_t ldx, T* r, int64_t ldr, T* z1, int64_t ldz1, T* z2, \
        int64_t ldz2, T* w, int64_t ldw, const T* p, int64_t ldp, const T* \
        q, int64_t ldq, const T* f, int64_t ldf, const T* g, int64_t ldg, \
        const T* rho, const T* beta, const uint8_t* stop_status); \
    int gkoc_pipe_cg_step_2_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, T* \
        beta, T* p, int64_t ldp, T* q, int64_t ldq, T* f, int64_t ldf, T* g, \
        int64_t ldg, const T* z, int64_t ldz, const T* w, int64_t ldw, const \
        T* m, int64_t ldm, const T* n, int64_t ldn, const T* prev_rho, const \
        T* rho, const T* delta, const uint8_t* stop_status);
GKOC_DECL_KRYLOV(double, f64)
GKOC_DECL_KRYLOV(float, f32)
GKOC_DECL_KRYLOV(gkoc_c128, c128)
GKOC_DECL_KRYLOV(gkoc_c64, c64)

/* bicg::{initialize, step_1, step_2} (core/solver/bicg_kernels.hpp;
 * reference/solver/bicg_kernels.cpp:24-110): the updates of the biconjugate gradient
 * method, for the system and its transposed shadow at once.  Same conventions as the
 * kernels above.  (core/solver/bicg.cpp applies A^T and M^T through
 * csr::conj_transpose and jacobi::transpose_jacobi.) */
#define GKOC_DECL_BICG(T, TN)                                                  \
    int gkoc_bicg_initialize_##TN(                                             \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* b, int64_t ldb,  \
        T* r, int64_t ldr, T* z, int64_t ldz, T* p, int64_t ldp, T* q,         \
        int64_t ldq, T* prev_rho, T* rho, T* r2, int64_t ldr2, T* z2,          \
        int64_t ldz2, T* p2, int64_t ldp2, T* q2, int64_t ldq2,                \
        uint8_t* stop_status);                                                 \
    int gkoc_bicg_step_1_##TN(                                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* p, int64_t ldp,        \
        const T* z, int64_t ldz, T* p2, int64_t ldp2, const T* z2,             \
        int64_t ldz2, const T* rho, const T* prev_rho,                         \
        const uint8_t* stop_status);                                           \
    int gkoc_bicg_step_2_##TN(                                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* r,  \
        int64_t ldr, T* r2, int64_t ldr2, const T* p, int64_t ldp, const T* q, \
        int64_t ldq, const T* q2, int64_t ldq2, const T* beta, const T* rho,   \
        const uint8_t* stop_status);
GKOC_DECL_BICG(double, f64)
GKOC_DECL_BICG(float, f32)
GKOC_DECL_BICG(gkoc_c128, c128)
GKOC_DECL_BICG(gkoc_c64, c64)

/* gcr::{initialize, restart, step_1} (core/solver/gcr_kernels.hpp;
 * reference/solver/gcr_kernels.cpp:24-88): the restarted generalised conjugate
 * residual method keeps its search directions p and A p in two tall Dense matrices of
 * (krylov_dim + 1) x rows rows; restart copies the preconditioned residual and its
 * image into the first slot, step_1 is the update with t = <r, Ap> / ||Ap||^2. */
#define GKOC_DECL_GCR(T, TN, R)                                                   \
    int gkoc_gcr_initialize_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,  \
                                 const T* b, int64_t ldb, T* residual,         \
                                 int64_t ldr, uint8_t* stop_status);           \
    int gkoc_gcr_restart_##TN(                                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* residual,        \
        int64_t ldr, const T* a_residual, int64_t ldar, T* p_bases,            \
        int64_t ldp, T* ap_bases, int64_t ldap, uint64_t* final_iter_nums);    \
    int gkoc_gcr_step_1_##TN(                                                  \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx,        \
        T* residual, int64_t ldr, const T* p, int64_t ldp, const T* ap,        \
        int64_t ldap, const R* ap_norm, const T* rap,                          \
        const uint8_t* stop_status);
GKOC_DECL_GCR(double, f64, double)
GKOC_DECL_GCR(float, f32, float)
GKOC_DECL_GCR(gkoc_c128, c128, double)
GKOC_DECL_GCR(gkoc_c64, c64, float)

/* minres::{initialize, step_1, step_2} (core/solver/minres_kernels.hpp;
 * reference/solver/minres_kernels.cpp:24-150): MINRES for symmetric (indefinite)
 * systems.  initialize: beta (holding <r, z>) becomes its square root, q = r / beta,
 * z /= beta, the other vectors and the rotation scalars are reset.  step_1: the Givens
 * update of the tridiagonal recurrence, scalars only.  step_2: the update of the search
 * direction p, the solution x and the Lanczos vectors. */
#define GKOC_DECL_MINRES(T, TN)                                                \
    int gkoc_minres_initialize_##TN(                                           \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* r, int64_t ldr,  \
        T* z, int64_t ldz, T* p, int64_t ldp, T* p_prev, int64_t ldpp, T* q,   \
        int64_t ldq, T* q_prev, int64_t ldqp, T* q_tilde, int64_t ldqt,        \
        T* beta, T* gamma, T* delta, T* cos_prev, T* cosv, T* sin_prev,        \
        T* sinv, T* eta_next, T* eta, uint8_t* stop_status);                   \
    int gkoc_minres_step_1_##TN(                                               \
        gkoc_stream_t s, int64_t cols, T* alpha, T* beta, T* gamma, T* delta,  \
        T* cos_prev, T* cosv, T* sin_prev, T* sinv, T* eta, T* eta_next,       \
        T* tau, const uint8_t* stop_status);                                   \
    int gkoc_minres_step_2_##TN(                                               \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* p,  \
        int64_t ldp, const T* p_prev, int64_t ldpp, T* z, int64_t ldz,         \
        const T* z_tilde, int64_t ldzt, T* q, int64_t ldq, T* q_prev,          \
        int64_t ldqp, T* v, int64_t ldv, const T* alpha, const T* beta,        \
        const T* gamma, const T* delta, const T* cosv, const T* eta,           \
        const uint8_t* stop_status);
GKOC_DECL_MINRES(double, f64)
GKOC_DECL_MINRES(float, f32)
GKOC_DECL_MINRES(gkoc_c128, c128)
GKOC_DECL_MINRES(gkoc_c64, c64)

/* ir::initialize (core/solver/ir_kernels.hpp:19-21; reference/solver/ir_kernels.cpp:20-27):
 * reset the stopping status; used by Ir and Chebyshev.
 * chebyshev::{init_update, update} (core/solver/chebyshev_kernels.hpp:21-40;
 * reference/solver/chebyshev_kernels.cpp:20-66): alpha, beta are host scalars of
 * solver::detail::coeff_type (double); elements are widened to double, updated,
 * narrowed back.  init_update: update = inner, output += alpha inner.
 * update: val = inner + beta update; inner = update = val; output += alpha val. */
int gkoc_ir_initialize(gkoc_stream_t s, int64_t cols, uint8_t* stop_status);
#define GKOC_DECL_CHEB(T, TN)                                                  \
    int gkoc_chebyshev_init_update_##TN(                                       \
        gkoc_stream_t s, int64_t rows, int64_t cols, double alpha,             \
        const T* inner_sol, int64_t ldi, T* update_sol, int64_t ldu,           \
        T* output, int64_t ldo);                                               \
    int gkoc_chebyshev_update_##TN(                                            \
        gkoc_stream_t s, int64_t rows, int64_t cols, double alpha,             \
        double beta, T* inner_sol, int64_t ldi, T* update_sol, int64_t ldu,    \
        T* output, int64_t ldo);
GKOC_DECL_CHEB(double, f64)
GKOC_DECL_CHEB(float, f32)
/* complex values: coeff_type = complex<double>; the two coefficients are HOST values passed by address */
#define GKOC_DECL_CCHEB(T, TN)                                                 \
    int gkoc_chebyshev_init_update_##TN(                                       \
        gkoc_stream_t s, int64_t rows, int64_t cols, const gkoc_c128* alpha_host, \
        const T* inner_sol, int64_t ldi, T* update_sol, int64_t ldu,           \
        T* output, int64_t ldo);                                               \
    int gkoc_chebyshev_update_##TN(                                            \
        gkoc_stream_t s, int64_t rows, int64_t cols, const gkoc_c128* alpha_host, \
        const gkoc_c128* beta_host, T* inner_sol, int64_t ldi, T* update_sol,  \
        int64_t ldu, T* output, int64_t ldo);
GKOC_DECL_CCHEB(gkoc_c128, c128)
GKOC_DECL_CCHEB(gkoc_c64, c64)

/* ------------------------------------------------- communicator (RCCL over xGMI)
 * Replaces, for device buffers, what the distributed path asks of
 * experimental::mpi::communicator (include/ginkgo/core/base/mpi.hpp):
 *   all_reduce (:838, in place, sum)          -> gkoc_comm_all_reduce_sum
 *   i_all_to_all_v (:1441) + request::wait    -> gkoc_comm_exchange_begin / _end
 * as used by distributed::Vector::compute_dot/norm2 (vector.cpp:473-592) and the
 * RowGatherer (row_gatherer.cpp:67-190).  One communicator per process (= per
 * GPU); the 128-byte id is created on rank 0 and handed to the other ranks by
 * whatever the host program already has (MPI_Bcast, a key-value store).
 * RCCL is bound at run time (gkoc_comm_load_rccl: path of the librccl the process
 * should use, NULL = default search), so this library loads on machines without it.
 * Every call enqueues on the given stream and returns; nothing synchronises. */
typedef struct gkoc_comm_s* gkoc_comm_t;
#define GKOC_COMM_ID_BYTES 128
int gkoc_comm_load_rccl(const char* librccl_path);
int gkoc_comm_unique_id(void* id_out /* GKOC_COMM_ID_BYTES */);
int gkoc_comm_create(gkoc_comm_t* comm, int n_ranks, int rank, const void* id);
int gkoc_comm_destroy(gkoc_comm_t comm);
int gkoc_comm_size(gkoc_comm_t comm, int* n_ranks, int* rank);
/* buf[0..n) <- sum over ranks, in place; value_size 8 (double) or 4 (float).
 * Every rank receives the same bits (one reduction order for all). */
int gkoc_comm_all_reduce_sum(gkoc_comm_t comm, gkoc_stream_t s, void* buf,
                             int64_t n, size_t value_size);
/* The same all-reduce travelling on `side` while kernels enqueued on `main` after begin run
 * (pipelined Krylov methods: PipeCg reduces its three scalars while the preconditioner and
 * the SpMV of the same iteration run, core/solver/pipe_cg.cpp:244-256).  begin: `side` waits
 * for what `main` has enqueued (the kernels that wrote buf); end: `main` waits for the
 * result.  An exchange started between begin and end must use the same `side` stream: the
 * communicator sees one order of operations on every rank (anything else is refused with
 * GKOC_E_INVALID, as is gkoc_comm_all_reduce_sum on another stream while something is
 * pending).  side == main (or NULL): plain all-reduce on main, end is a no-op. */
int gkoc_comm_all_reduce_begin(gkoc_comm_t comm, gkoc_stream_t main_stream,
                               gkoc_stream_t side, void* buf, int64_t n,
                               size_t value_size);
int gkoc_comm_all_reduce_end(gkoc_comm_t comm, gkoc_stream_t main_stream);
/* Sparse all-to-all of contiguous segments: send_counts[p] values leave
 * send_buf for rank p - from offset send_displs[p] (in values; the send_offsets
 * of mpi.hpp:1441), or packed in rank order when send_displs == NULL -,
 * recv_counts[p] values arrive into recv_buf from rank p, packed in rank order;
 * ranks with both counts 0 are not contacted.  With displacements a rank whose
 * peers want contiguous row ranges (slab partitions) sends straight out of the
 * vector, without a pack kernel.
 * begin: `side` waits for what `main` has enqueued so far (the pack kernel),
 * then carries the grouped send/recv, so kernels enqueued on `main` after
 * begin overlap the transfer.  end: `main` waits for the transfer.
 * side == main (or NULL): plain in-order exchange on main, end is a no-op. */
int gkoc_comm_exchange_begin(gkoc_comm_t comm, gkoc_stream_t main_stream,
                             gkoc_stream_t side, const void* send_buf,
                             const int64_t* send_counts,
                             const int64_t* send_displs, void* recv_buf,
                             const int64_t* recv_counts, size_t value_size);
int gkoc_comm_exchange_end(gkoc_comm_t comm, gkoc_stream_t main_stream);
/* gkoc_comm_all_reduce_begin AND gkoc_comm_exchange_begin behind one fork and one join (two
 * barrier packets on the main stream instead of four): the side stream reduces, then exchanges;
 * gkoc_comm_exchange_end / _join end both.  What a pipelined solver wants when the vector whose
 * halo travels is final at the point where its dot products are (PipeCg with the preconditioner
 * application inside its step kernel). */
int gkoc_comm_all_reduce_exchange_begin(gkoc_comm_t comm, gkoc_stream_t main_stream,
                                        gkoc_stream_t side, void* reduce_buf, int64_t reduce_n,
                                        size_t reduce_value_size, const void* send_buf,
                                        const int64_t* send_counts, const int64_t* send_displs,
                                        void* recv_buf, const int64_t* recv_counts,
                                        size_t value_size);
/* the same, but main_stream also waits for the kernels enqueued on the exchange's stream since
 * gkoc_comm_exchange_begin (the boundary rows, computed there as soon as the halo is in) */
int gkoc_comm_exchange_join(gkoc_comm_t comm, gkoc_stream_t main_stream);
/* A fork without an event: what has been enqueued on main_stream so far happens before what is
 * enqueued on `side` from now on.  An event record + hipStreamWaitEvent puts a barrier packet on
 * the MAIN queue (6-7 us of idle device in front of main's next kernel on MI355X); this is two
 * one-thread kernels - main stores `number` into *word (device memory, zero at the start), side
 * polls until the word has reached it.  The caller counts (1, 2, ...).  The communicator's forks
 * (gkoc_comm_exchange_begin, gkoc_comm_all_reduce_begin, gkoc_comm_all_reduce_exchange_begin)
 * are of this kind unless GKOC_COMM_FORK=event. */
int gkoc_stream_fork(gkoc_stream_t main_stream, gkoc_stream_t side, uint32_t* word, uint32_t number);
/* the polling half alone: the store is done by a kernel of the caller on the main stream
 * (gkoc_csr_spmv_gated_*: fork_word / fork_number) */
int gkoc_stream_fork_wait(gkoc_stream_t side, const uint32_t* word, uint32_t number);
/* The fork of the communicator's NEXT gkoc_comm_exchange_begin / _all_reduce_begin /
 * _all_reduce_exchange_begin (same streams) is opened by the caller: *word / *number are what the
 * caller's next kernel on main_stream has to store (gkoc_csr_spmv_gated_* does it as its first
 * action), the begin call only enqueues the poller.  *word = NULL: not available (forks are events,
 * or the two streams have the same priority and might share a hardware queue, where a poller in
 * front of the storing kernel would block it) - the begin call then forks by itself as usual. */
int gkoc_comm_fork_deferred(gkoc_comm_t comm, gkoc_stream_t main_stream, gkoc_stream_t side,
                            uint32_t** word, uint32_t* number);
/* *timed_out = 1 if a poller of one of this communicator's forks gave up (about a minute without the
 * store it waited for: what ran behind it was not ordered behind the main stream); synchronises */
int gkoc_comm_fork_timed_out(gkoc_comm_t comm, int* timed_out);
/* ends an exchange WITHOUT making the main stream wait: the kernel that reads the halo waits for
 * it itself (gkoc_gate_open on the side stream + gkoc_csr_spmv_gated_*) */
int gkoc_comm_exchange_forget(gkoc_comm_t comm);
/* MPI_Alltoallv (mpi.hpp all_to_all_v / i_all_to_all_v) in bytes: counts and offsets per peer on
 * both sides, enqueued on s as one grouped send / recv */
int gkoc_comm_all_to_all_v_bytes(gkoc_comm_t comm, gkoc_stream_t s, const void* send_buf,
                                 const int64_t* send_bytes, const int64_t* send_offsets,
                                 void* recv_buf, const int64_t* recv_bytes,
                                 const int64_t* recv_offsets);

/* ---- the communicator's second transport: mailboxes in peer-mapped device memory (hipIpc) -----
 * The same gkoc_comm_* operations without RCCL: every rank owns a WINDOW (one device allocation) that
 * all peers map; an all-reduce is every rank storing its values (8-byte words carrying 4 bytes of data
 * and the operation's number) into every peer's window and summing, IN RANK ORDER, what it finds in
 * its own - one hop over xGMI, one kernel, the same bits on every rank and in every run; an exchange
 * is the sender copying its segment into its slot of the receiver's window followed by a flag, and
 * the receiver copying it out (csrc/comm_ipc.hpp).  It answers the latency question of the Krylov
 * loop (two 8-byte all-reduces per Cg iteration) and stands where the reference lets a
 * collective_communicator be chosen (include/ginkgo/core/distributed/collective_communicator.hpp:31-71).
 * Several processes on ONE device can form such a communicator (RCCL refuses that), which is how the
 * whole N > 1 device path is tested on a one-GPU box.
 *   1. every rank: gkoc_comm_ipc_create -> its handle card (GKOC_COMM_IPC_HANDLE_BYTES);   slot_bytes: room per peer and
 *      direction for one message (0: GKOC_IPC_SLOT_MIB or 8 MiB); at most 16 ranks
 *   2. the host program gathers all handles in rank order (MPI_Allgather, a key-value store)
 *   3. every rank: gkoc_comm_ipc_connect(all handles)
 * after which every gkoc_comm_* call above works as documented (messages larger than slot_bytes:
 * GKOC_E_NOT_SUPPORTED - the check is per rank, and a rank that passes it counts the exchange: the ranks of a
 * communicator must AGREE, before they call, on whether an exchange fits - as the MPI layer does with the two-int
 * agreement it sends ahead of every all-to-all-v, gko_binding/mpi_rccl.cpp route_local; an exchange that one
 * rank refuses and another starts leaves their per-pair sequence numbers one apart).  Waiting kernels have a patience (GKOC_IPC_PATIENCE_MS, default 120000): a
 * wait that runs out sets a bit in the status word and the kernel ends; gkoc_comm_status reads it
 * (no synchronisation; 0 = nothing ever timed out; bit 0 all-reduce, bit 1 a message, bit 2 an
 * acknowledgement).  gkoc_comm_destroy must be entered by a rank only after its peers have completed
 * the operations it takes part in (a host barrier, or the end of the solve). */
#define GKOC_COMM_IPC_HANDLE_BYTES 128    /* the hipIpc handle of the window + where it lives: PCI bus id of the
                                           * device, whether the window is uncached memory, a version byte */
#define GKOC_COMM_BUS_ID_BYTES 32
int gkoc_comm_ipc_create(gkoc_comm_t* comm, int n_ranks, int rank, int64_t slot_bytes,
                         void* handle_out /* GKOC_COMM_IPC_HANDLE_BYTES */);
int gkoc_comm_ipc_connect(gkoc_comm_t comm, const void* handles /* n_ranks x GKOC_COMM_IPC_HANDLE_BYTES */);
int gkoc_comm_status(gkoc_comm_t comm, uint32_t* status);
/* the patience of the operations enqueued from now on (milliseconds; <= 0: back to GKOC_IPC_PATIENCE_MS /
 * the default).  A first known-answer collective right after the hand-shake - all ranks are there - can
 * be given seconds instead of minutes.  No effect on an RCCL communicator. */
int gkoc_comm_set_patience_ms(gkoc_comm_t comm, int64_t ms);
/* *transport: 0 = RCCL, 1 = mailboxes; *window_uncached (may be NULL): the window is uncached device memory */
int gkoc_comm_transport(gkoc_comm_t comm, int* transport, int* window_uncached);
/* Who is where - the answer to "did the communicator see N ranks, on which devices" (bench.py prints it in the
 * N > 1 line).  RCCL: ranks_seen = ncclCommCount, rccl_version = ncclGetVersion, bus ids from one small
 * all-gather at creation; mailboxes: from the cards.  cross_device = 1: at least one peer sits on another
 * device.  Two rules follow from it (replacing core/distributed/matrix.cpp:450-509's req.wait() by waits inside
 * kernels needs them): (1) gkoc_comm_ipc_connect REFUSES - on every rank alike, GKOC_E_NOT_SUPPORTED - windows
 * in plain (coarse-grained) memory between different devices; (2) the kernels that read behind a gate word
 * (gkoc_csr_spmv_gated_*, the gated PipeCg steps) pay a system-scope acquire per waiting wave from then on
 * (gate_fence = 2), until the caller lowers it with gkoc_gate_fence_policy after its own check on that
 * communicator (the one-kernel product against the join-based one) has passed on every rank. */
typedef struct gkoc_comm_topology {
    int32_t transport, n_ranks, rank;
    int32_t ranks_seen;       /* ncclCommCount (RCCL) / cards read (mailboxes); 0: not asked, -1: the call failed */
    int32_t rccl_version;     /* ncclGetVersion, 0 for mailboxes */
    int32_t cross_device;
    int32_t window_uncached;  /* mailboxes: every rank's window is uncached memory */
    int32_t gate_fence;       /* the current gkoc_gate_fence_policy */
    char bus_id[16][GKOC_COMM_BUS_ID_BYTES];   /* PCI bus id of every rank's device (first 16 ranks) */
} gkoc_comm_topology;
int gkoc_comm_topology_get(gkoc_comm_t comm, gkoc_comm_topology* out);
/* set: -1 = query only, 0 = the cheap gate (an agent-scope acquire only for a wave that had to wait), 1 = every
 * waiting wave an agent-scope acquire, 2 = a system-scope one; *now (may be NULL) = the policy afterwards.
 * Process-wide.  Raised to 2 by any communicator that comes up with a peer on another device. */
int gkoc_gate_fence_policy(int set, int* now);

#ifdef __cplusplus
}
#endif
#endif /* GKO_CDNA4_H_ */
