// sga_engine_impl.h -- what the translation units of the host side share: the engine record, the error / option
// plumbing and the helpers several of them call.  Internal to libsga.so; the ABI is include/sga.h.
//   sga_engine.cpp    handle, options, replicas, ladder, sweeps, single-site operators, exchange
//   sga_problem.cpp   sga_set_dense / sga_set_csr / sga_set_tsp: scans, packing, CSR layouts
//   sga_autotune.cpp  sga_autotune (measured launch geometry / sweep form)
//   sga_state.cpp     state access, checkpoint / resume, timing, sga_describe, checksum
//   sga_route.cpp     WHICH form runs: pure functions of the problem's traits and the options (no HIP calls)
#ifndef SGA_ENGINE_IMPL_H
#define SGA_ENGINE_IMPL_H
#include <hip/hip_runtime.h>
#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "sga.h"
#include "sga_kernels.h"
#include "sga_route.h"

namespace sga_impl {


#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(_e == hipErrorOutOfMemory ? SGA_ERR_MEMORY : SGA_ERR_DEVICE,       \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                \
    } while (0)

template <typename T>
void dev_free(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

inline bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof(at));
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear: plain host memory
        return false;
    }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// Grow-only device scratch slots owned by an engine: staging of host-side call arguments and
// outputs re-uses them, so the steady-state call path performs no hipMalloc / hipFree (which
// would synchronise the device).
struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 256;
        hipError_t e = hipMalloc(&ptr, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

// A read-only view of a user buffer on the device: borrowed if it already lives there,
// otherwise staged into a scratch slot.
template <typename T>
struct DevIn {
    const T *ptr = nullptr;
    bool staged = false;
    int init(Scratch &slot, const T *user, size_t count, hipStream_t st) {
        if (!user || count == 0) return SGA_OK;
        if (is_device_ptr(user)) {
            ptr = user;
            return SGA_OK;
        }
        HIPCHK(slot.reserve(count * sizeof(T)));
        HIPCHK(hipMemcpyAsync(slot.ptr, user, count * sizeof(T), hipMemcpyHostToDevice, st));
        ptr = static_cast<const T *>(slot.ptr);
        staged = true;
        return SGA_OK;
    }
};

// A device scratch buffer whose contents are copied to a user buffer (host or device).
template <typename T>
struct DevOut {
    T *ptr = nullptr;
    T *user = nullptr;
    size_t count = 0;
    int init(Scratch &slot, T *user_, size_t count_, hipStream_t st) {
        user = user_;
        count = count_;
        if (!user || count == 0) return SGA_OK;
        HIPCHK(slot.reserve(count * sizeof(T)));
        ptr = static_cast<T *>(slot.ptr);
        HIPCHK(hipMemsetAsync(ptr, 0, count * sizeof(T), st));
        return SGA_OK;
    }
    int flush(hipStream_t st) {
        if (!ptr) return SGA_OK;
        HIPCHK(hipMemcpyAsync(user, ptr, count * sizeof(T), hipMemcpyDefault, st));
        return SGA_OK;
    }
};


}  // namespace sga_impl

using namespace sga_impl;

struct sga_engine {
    int device = 0;
    int cus = 256;  // compute units of the device
    long long opt[OPT_COUNT];  // sga_set_option values (defaults: OPT_DEFS, the environment read once in sga_create)
    // an option latched at sga_set_* (bit 2) / sga_init_replicas (bit 1) changed afterwards: the next sweep says so
    // instead of silently running the form the old value chose
    int opt_stale = 0;
    const char *opt_stale_key = nullptr;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // problem
    int n = 0;
    int n_models = 1;  // dense batches: models stacked row-wise, replicas split evenly
    bool csr = false;
    bool want_i8 = false, acc64 = false;
    bool acc_canon = false;  // acc64 and the fp64 row sum is not provably exact: canonical summation order
    bool use_t2 = false;           // ternary J as two bit-planes for the production sweeps
    unsigned int *J_bits = nullptr;  // [2][n][ld/32]
    float *row_nnz = nullptr;        // [n]
    int waves_t2 = 0, cpw_t2 = 0;    // bit-plane geometry (waves/cpw then describe the int8 fallback)
    void *J_packed = nullptr;  // [n][ldj] float | int8
    long long ld = 0;   // spins per replica (whole chunks)
    long long ldj = 0;  // row stride of J_packed: n rounded up to 128 bytes
    int waves = 0, cpw = 0;
    int32_t *rowptr = nullptr, *colidx = nullptr;  // rowptr: only while the layout has < 2^31 entries
    long long *rowptr64 = nullptr;                  // always (energy / single-site kernels)
    int4 *rowinfo = nullptr;     // slotted layout, per row: first slot, slots, offset of a zero slot, h (wide sweep forms)
    bool slotted = false;        // rows padded to whole 64-entry slots (value-0 entries behind each row)
    bool csr_sorted = false;     // rows strictly sorted by column: no duplicate entries
    uint32_t *cvp = nullptr;     // slotted layout with packed entries (24-bit column | int8 value << 24), on demand
    bool cvp_tried = false;      // packing was attempted for this problem (values may not fit)
    int csr_storage = SGA_CSR_STORAGE_AUTO;         // what the caller asked for ...
    int csr_storage_latched = SGA_CSR_STORAGE_AUTO; // ... and what the current replicas were laid out for
    int table_scale = 1;         // CSR accept table: entry q stands for dE = 2 q / table_scale
    long long layout_entries = 0;  // entries of the layout the kernels read (nnz + padding)
    long long max_row_len = 0;     // entries of the longest row
    bool big = false;  // CSR sweeps with bit spins in LDS (decided per replica set)
    int big_form = 0;  // 0 int8 spins | 1 bits, one replica per workgroup, 64-bit extents | 2 bits, narrow
    float *val = nullptr;   // colidx / val: only while the structure is being checked
    int2 *cv = nullptr;     // [nnz] interleaved (column, value bits): what the kernels read
    long long nnz = 0;
    float *h = nullptr, *diag = nullptr;
    // TSP-structured couplings, never stored (sga_set_tsp): scaled distance tables + penalties
    bool tsp = false, tsp_exact = true;
    float *nd4 = nullptr, *nd4t = nullptr;
    sga::TspArgs tsp_args{};
    int tsp_waves = 0, tsp_passes = 0;
    double *epart = nullptr;  // per-slice energy sums (few replicas)
    size_t epart_bytes = 0;
    int tune_waves = 0, tune_spl = 0;
    std::string tune_table;  // what the last sga_autotune measured: "candidate=ms per sweep;..." (sga_get_autotune_table)
    int rule = SGA_RULE_METROPOLIS;
    bool consistent_dE = true;  // J symmetric with zero diagonal: dE of the rule == energy change
    int table_m = 0;  // integer problems: largest possible |dE| / 2 (0 = not integer / too big)
    // cached-local-field sweep (sweep_clf_impl.h)
    int field_cache = SGA_FIELD_CACHE_OFF;  // what the caller asked for
    bool from_dense = false;  // CSR problem built from a sparse matrix handed over dense (sga_set_dense, SGA_J_AUTO)
    bool clf_problem = false;  // dense, one model, J and h integer valued, symmetric, zero diagonal, sums < 2^24
    float row_abs_max = 0.0f;  // max_i(sum_j |J_ij| + |h_i|)
    int j_abs_max = 0;         // ceil(max |J_ij|): the most one flip moves another site's field (several accepts per round: sweep_clfb_impl.h)
    // ... of CSR problems (sweep_clf_csr.hip): integer J, rows strictly sorted, max_i sum_j |J_ij| < 2^15, the accept
    // table applies, dE of the rule == energy change; the fields are then D = J s as int16, h stays outside
    bool clf_csr_problem = false;
    float row_j_abs_max = 0.0f;  // max_i sum_j |J_ij|
    float csr_row_abs_max = 0.0f;  // CSR: max_i (sum_j |J_ij| + |h_i|): no |fk| of a move exceeds it
    int *hq = nullptr;           // [n] table_scale * h_i as integers (built with the first cached sweep)
    int clf_scale = 1, clf_bits = 16;
    void *fields = nullptr;    // [R][ldf] int16 | int32: clf_scale * (J s + h), valid while fields_valid
    long long ldf = 0;
    bool fields_valid = false;
    void *ybuf = nullptr;      // [count][ldj] int32 | float: scratch of the all-replica field pass
    size_t ybuf_bytes = 0;
    // SGA_FIELD_CACHE_AUTO looks at the acceptance of the last sweeps now and then (host read-back of the
    // per-replica counters): an accept costs ~2 us of its replica's chain, so the cached-field sweep only
    // pays while the HOTTEST replica accepts little
    bool auto_unavailable = false;      // the fields could not be allocated: AUTO stays on the row-per-proposal kernels
    std::vector<int> route;             // per local replica: 0 = cached-field kernel, 1 = row-per-proposal kernel (AUTO)
    int n_route_clf = 0;                // replicas routed to the cached-field kernel
    bool clf_wide = false;              // the cached-field launch runs at eight waves per replica (option "clf_tail_waves")
    bool clf_hot = true;                // its hottest replica accepts > ~1 %: several accepts per round (option "clf_batched" = 2)
    bool route_dirty = true;            // the device copy of the replica lists is stale
    int *d_rep_lists = nullptr;         // [2][R]: the cached-field kernel's replicas, then the row kernels'
    hipStream_t aux_stream = nullptr;   // the second launch of a mixed sweep
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    char last_mixed[448] = {0};
    char last_kernel[512] = {0};        // the instantiation this engine's last sweep launch ran (sga_get_last_kernel)
    long long auto_mark_attempted = 0;  // per-replica attempts at the last look
    int auto_interval = 4;              // sweeps until the next look (doubles up to 32)
    std::vector<unsigned long long> auto_mark_acc;
    int csr_acc = sga::CSR_ACC_F64_CANON;  // CSR: how the sweep kernels form a row sum (set time)

    // replicas
    int R = 0, Rg = 0, replica0 = 0;
    uint64_t seed = 0;
    int sstride = 0;
    int8_t *spins = nullptr, *best_spins = nullptr;
    double *energy = nullptr, *best_energy = nullptr, *rep_temp = nullptr;
    unsigned long long *n_acc = nullptr;
    long long attempted = 0;  // per replica
    uint32_t sweeps_done = 0, rounds = 0;

    // ladder
    int n_ladders = 0;
    double *slot_temps = nullptr;
    int32_t *slot_to_rep = nullptr;
    long long *ex_attempts = nullptr, *ex_accepts = nullptr;
    int *d_count = nullptr;
    float *wolff_u = nullptr;        // recorded uniforms of the Wolff rule [R][wolff_cap] (parity tests)
    long long *wolff_cursor = nullptr;  // [R]
    long long wolff_cap = 0;
    int *d_flags = nullptr;  // [16] value / structure scan results of the set_* calls (one per engine)

    // staging slots: 0 sched, 1 replay sites, 2 replay u, 3 energy trace, 4 accept trace,
    // 5 dE trace, 6 exchange energies, 7 exchange start, 8 exchange u
    Scratch scratch[9];
    Scratch point_sites, point_out;  // single-site operators
    Scratch csr_energy;              // transposed spin bits + partial sums of the all-replica CSR energy pass

    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    int64_t launches = 0;
    double total_ms = 0.0;

    void free_problem() {
        dev_free(J_packed);
        dev_free(J_bits);
        dev_free(row_nnz);
        use_t2 = false;
        dev_free(rowptr);
        dev_free(rowptr64);
        dev_free(rowinfo);
        dev_free(cvp);
        cvp_tried = false;
        slotted = false;
        dev_free(colidx);
        dev_free(val);
        dev_free(cv);
        dev_free(h);
        dev_free(diag);
        dev_free(nd4);
        dev_free(nd4t);
        dev_free(hq);
        clf_csr_problem = false;
        tsp = false;
        dev_free(epart);
        epart_bytes = 0;
        clf_problem = false;
        n = 0;
        ld = 0;
    }
    void free_replicas() {
        dev_free(spins);
        dev_free(best_spins);
        dev_free(energy);
        dev_free(best_energy);
        dev_free(rep_temp);
        dev_free(n_acc);
        dev_free(slot_temps);
        dev_free(slot_to_rep);
        dev_free(ex_attempts);
        dev_free(ex_accepts);
        dev_free(wolff_u);
        dev_free(wolff_cursor);
        wolff_cap = 0;
        dev_free(fields);
        fields_valid = false;
        dev_free(ybuf);
        ybuf_bytes = 0;
        auto_unavailable = false;
        route.clear();
        n_route_clf = 0;
        clf_wide = false;
        clf_hot = true;
        route_dirty = true;
        dev_free(d_rep_lists);
        auto_mark_attempted = 0;
        auto_interval = 4;
        auto_mark_acc.clear();
        R = Rg = 0;
        n_ladders = 0;
    }
};

namespace sga_impl {

inline int elems_per_chunk(bool i8) { return i8 ? 1024 : 256; }
// Zeroed (column 0, value 0) entries behind the CSR entry array.  The sweep kernels load a row's entries without
// a bounds test and mask what lies past the row's end when summing: the one-update forms reach up to 64 entries
// past the last row's first entry (also the wide forms' zero slot), the several-updates-per-step builds for rows of
// 65 ... 256 entries (sweep_csr_rows.hip: 16 lanes x 8 | 16 entries per lane) up to 256.
constexpr long long CSR_TAIL_PAD = 256;
constexpr int T2_ELEMS_PER_CHUNK = 8192;  // 1 KiB of one bit-plane
inline long long t2_row_bits(int n) { return ((long long)n + 127) / 128 * 128; }  // 16-byte granules

// ---- sga_engine.cpp
sga_route_query route_query_of(const sga_engine *e);  // the engine's own traits / replicas / options as a query
bool fields_pass_applies(const sga_engine *e, int count);
int fields_pass(sga_engine *e, int r0, int count, double *energy, void *fields);
bool clf_possible(const sga_engine *e, const char **why);
bool clf_active(const sga_engine *e);
int ensure_fields(sga_engine *e);
int recompute_energy_range(sga_engine *e, int r0, int count);
int ensure_packed(sga_engine *e);
// ---- sga_problem.cpp
bool csr_rows_medium(const sga_engine *e);
int csr_updates_per_step(const sga_engine *e);
int ensure_slotted(sga_engine *e);
int ensure_packed_entries(sga_engine *e);

}  // namespace sga_impl

#endif  // SGA_ENGINE_IMPL_H
