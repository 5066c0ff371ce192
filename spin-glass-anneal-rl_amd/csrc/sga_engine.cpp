// sga_engine.cpp -- host side of the C ABI declared in include/sga.h: owns the HBM buffers
// (packed couplings, replica spins / energies / bests, ladder state), picks the launch
// geometry, and drives the HIP kernels.  No torch, no exceptions across the ABI.
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

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

// ---- engine options (sga_set_option / sga_get_option, include/sga.h) ----------------------------------
// Form selection switches -- A/B measurements, parity tests that force the slower forms -- are per-engine
// values behind the C ABI.  The environment is consulted ONCE, in sga_create, for the defaults (the variable
// named here); nothing else in the library reads it.
enum Opt {
    OPT_LOOK_AHEAD, OPT_CLF_WAVES, OPT_SPARSE_ROUTE, OPT_BATCHED_ENERGY, OPT_FORCE_GENERAL, OPT_CSR_UPDATES_PER_STEP,
    OPT_TSP_PARALLEL, OPT_FORCE_CSR_BITS, OPT_CSR_BITS, OPT_CSR_SLOTS, OPT_HALF_TABLE, OPT_FORCE_CSR_ACC,
    OPT_FORCE_DENSE_CANON, OPT_ZERO_SLOT_EVERY, OPT_REPLICA_ROUTING, OPT_FIELDS_SCRATCH_MB, OPT_CLF_BATCHED,
    OPT_CLF_TAIL_WAVES, OPT_COUNT
};
struct OptDef {
    const char *key;
    const char *env;     // environment variable giving the default at sga_create (nullptr: none)
    int env_presence;    // 1: the variable being set means `env_value`; 0: its integer value is taken
    long long env_value;
    long long def, lo, hi;
};
constexpr OptDef OPT_DEFS[OPT_COUNT] = {
    {"look_ahead", "SGA_NO_LOOK_AHEAD", 1, 0, 1, 0, 1},
    {"clf_waves", "SGA_CLF_WAVES", 0, 0, 0, 0, 16},
    {"sparse_route", "SGA_NO_SPARSE_ROUTE", 1, 0, 1, 0, 1},
    {"batched_energy", "SGA_NO_MFMA_ENERGY", 1, 0, 1, 0, 2},
    {"force_general", "SGA_FORCE_GENERAL", 1, 1, 0, 0, 1},
    {"csr_updates_per_step", "SGA_CSR_PAIR_AHEAD", 0, 0, -1, -1, 8},
    {"tsp_updates_per_step", "SGA_TSP_PARALLEL", 0, 0, -1, -1, 8},
    {"force_csr_bits", "SGA_FORCE_CSR_BIG", 1, 1, 0, 0, 1},
    {"csr_bits", "SGA_NO_CSR_BITS", 1, 0, 1, 0, 1},
    {"csr_slots", "SGA_NO_CSR_SLOTS", 1, 0, 1, 0, 1},
    {"half_integer_table", "SGA_NO_HALF_TABLE", 1, 0, 1, 0, 1},
    {"force_csr_acc", "SGA_FORCE_CSR_ACC", 0, 0, 0, 0, 3},
    {"force_dense_canonical", "SGA_FORCE_DENSE_CANON", 1, 1, 0, 0, 1},
    {"zero_slot_every", "SGA_ZERO_SLOT_EVERY", 0, 0, 0, 0, 1ll << 21},
    {"replica_routing", "SGA_NO_REPLICA_ROUTING", 1, 0, 1, 0, 1},
    {"fields_scratch_mb", "SGA_FIELDS_SCRATCH_MB", 0, 0, 256, 1, 65536},
    {"clf_batched", "SGA_CLF_BATCHED", 0, 0, 2, 0, 2},
    {"clf_tail_waves", "SGA_NO_CLF_TAIL_WAVES", 1, 0, 1, 0, 1},
};
int find_option(const char *key) {
    if (!key) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(key, OPT_DEFS[i].key) == 0) return i;
    return -1;
}

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

bool is_device_ptr(const void *p) {
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

}  // namespace

struct sga_engine {
    int device = 0;
    int cus = 256;  // compute units of the device
    long long opt[OPT_COUNT];  // sga_set_option values (defaults: OPT_DEFS, the environment read once in sga_create)
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

namespace {

int elems_per_chunk(bool i8) { return i8 ? 1024 : 256; }
// Zeroed (column 0, value 0) entries behind the CSR entry array.  The sweep kernels load a row's entries without
// a bounds test and mask what lies past the row's end when summing: the one-update forms reach up to 64 entries
// past the last row's first entry (also the wide forms' zero slot), the several-updates-per-step builds for rows of
// 65 ... 256 entries (sweep_csr_rows.hip: 16 lanes x 8 | 16 entries per lane) up to 256.
constexpr long long CSR_TAIL_PAD = 256;
constexpr int T2_ELEMS_PER_CHUNK = 8192;  // 1 KiB of one bit-plane
long long t2_row_bits(int n) { return ((long long)n + 127) / 128 * 128; }  // 16-byte granules

// Pick waves-per-replica W and chunks-per-wave CPW for a dense row of C chunks.  Measured on
// MI355X at n = 10^4, R = 1024 (profiles/r01_geometry_sweep.md): full occupancy (R*W ~ 32
// waves per CU) is best as long as every wave keeps >= 4 KiB of the row in flight and W
// balances the four SIMDs; row padding is paid on every read, so it dominates the cost.
bool choose_geometry(int n, int epc, int R, int forced_waves, int &W, int &CPW,
                     int max_cpw = sga::MAX_CPW, int unit = 1 /* 1-KiB chunks per counted unit */) {
    const int C = (n + epc - 1) / epc;
    // A wave beyond the row's last chunk would hold nothing but pad lanes (every lane redirected
    // to the row's first granule against zero pad spins): a forced count is clamped to the chunk
    // count, so that no geometry the heuristic itself would refuse is reachable by tuning.
    if (forced_waves > C) forced_waves = C;
    double target = 8192.0 / std::max(R, 1);
    target = std::min(16.0, std::max(1.0, target));
    double best_cost = 1e30;
    W = CPW = 0;
    for (int w = 1; w <= sga::MAX_WAVES; ++w) {
        if (forced_waves > 0 && w != forced_waves) continue;
        const int cpw = (C + w - 1) / w;
        if (cpw > max_cpw) continue;
        if (w > C && w > 1) continue;
        const double pad = (double)(w * cpw - C) / C;
        double cost = 4.0 * pad + 0.05 * std::fabs(std::log2(w / target));
        if (cpw * unit < 4 && w > 1) cost += 0.5 * (4 - cpw * unit);   // too little in flight per wave
        if (w > 2 && (w % 4) != 0) cost += 0.03;          // uneven over the 4 SIMDs
        // 9-10 chunks: at the edge of the register file, no look-ahead form (n = 10^4 fp32,
        // 4096 replicas: 4 waves x 10 chunks 1.53e8 attempts/s, 5-16 waves 1.9-2.0e8)
        if (cpw * unit > 8) cost += 0.2;
        if (cost < best_cost) {
            best_cost = cost;
            W = w;
            CPW = cpw;
        }
    }
    if (W == 0) {
        // only reached when the row is too long for the register-resident form (more than max_cpw
        // chunks per wave at the forced / at 16 waves): streaming kernel.  forced_waves <= C here.
        W = forced_waves > 0 ? forced_waves : std::min(sga::MAX_WAVES, C);
        CPW = (C + W - 1) / W;
    }
    return true;
}

// All replicas' local fields in one pass over the couplings on the matrix cores (fields_dense.hip),
// then energies and / or the resident fields of the cached-field sweep from them.
bool fields_pass_applies(const sga_engine *e, int count) {
    // option "batched_energy": 0 = off (A/B switch), 1 = where the batched sums carry the same bits as the
    // per-replica kernels', 2 = always.  Real-valued couplings that need the canonical summation order keep the
    // per-replica kernels under 1: the matrix-core pass sums a row in k-order, and the fp32-rounded row sum could
    // differ in its last bit with the number of replicas recomputed together (sga_set_spins: one; a shard: R_local).
    const long long mode = e->opt[OPT_BATCHED_ENERGY];
    if (mode == 0 || (mode == 1 && e->acc_canon)) return false;
    return !e->csr && !e->tsp && e->n_models == 1 && count >= 32 && e->J_packed;
}
// The pass writes Y = S J^T for the replicas it is given into a scratch buffer ([tile][ldj] int32 | fp32) before
// the finish kernel reduces it.  The scratch is bounded: replica sets whose Y would exceed the cap go
// through in tiles of whole 128-replica blocks (option "fields_scratch_mb", 256; one more pass over J per tile: 16 384 replicas of 10 000 spins =
// three passes over 100 MB instead of 655 MB of scratch kept for the life of the replicas), and a failed
// allocation halves the tile before giving up with SGA_ERR_MEMORY -- which the callers treat as "this fast path is
// not available" (per-replica energy kernels; SGA_FIELD_CACHE_AUTO stays on the row-per-proposal kernels).
int fields_pass(sga_engine *e, int r0, int count, double *energy, void *fields) {
    const size_t row = sizeof(float) * (size_t)e->ldj;
    long long tile = count;
    const size_t scratch_cap = (size_t)e->opt[OPT_FIELDS_SCRATCH_MB] << 20;  // option "fields_scratch_mb" (256)
    const long long cap_rows = std::max<long long>(128, (long long)(scratch_cap / row) / 128 * 128);
    if (tile > cap_rows) tile = cap_rows;
    while (row * (size_t)tile > e->ybuf_bytes) {
        dev_free(e->ybuf);
        e->ybuf_bytes = 0;
        if (hipMalloc(&e->ybuf, row * (size_t)tile) == hipSuccess) {
            e->ybuf_bytes = row * (size_t)tile;
            break;
        }
        (void)hipGetLastError();  // (cleared: the caller may go on without this pass)
        e->ybuf = nullptr;
        if (tile <= 128) return fail(SGA_ERR_MEMORY, "no memory for the scratch of the all-replica field pass");
        tile = std::max<long long>(128, tile / 2 / 128 * 128);
    }
    const int mode = e->want_i8 ? 0 : (e->acc64 ? 2 : 1);
    const size_t fbytes = (size_t)(e->clf_bits / 8);
    for (long long t0 = 0; t0 < count; t0 += tile) {
        sga::FieldsArgs f{};
        f.J = e->J_packed;
        f.spins = e->spins + (long long)(r0 + t0) * e->sstride;
        f.Y = e->ybuf;
        f.h = e->h;
        f.energy = energy ? energy + t0 : nullptr;
        f.fields = fields ? static_cast<unsigned char *>(fields) + (size_t)t0 * (size_t)e->ldf * fbytes : nullptr;
        f.ldj = e->ldj;
        f.ldy = e->ldj;
        f.ldf = e->ldf;
        f.n = e->n;
        f.R = (int)std::min<long long>(tile, count - t0);
        f.sstride = e->sstride;
        f.field_bits = fields ? e->clf_bits : 0;
        f.field_scale = e->clf_scale;
        HIPCHK(sga::launch_fields_dense(f, mode, e->stream));
        HIPCHK(sga::launch_fields_finish(f, mode == 0, e->stream));
    }
    return SGA_OK;
}

// The cached-local-field sweep serves: dense integer-valued symmetric problems (one model) whose
// fields and spin bits fit LDS, any single-site rule.  why: the reason when it does not.
bool clf_possible(const sga_engine *e, const char **why) {
    const char *reason = nullptr;
    if (e->csr && !e->tsp) {
        // sparse couplings: the dynamic part of the fields as int16 in LDS (sweep_clf_csr.hip)
        const long long ldf = ((long long)e->n + 127) / 128 * 128;
        if (!e->clf_csr_problem)
            reason = e->from_dense
                         ? "cached local fields: this sparse matrix was kept as CSR because the field cache was OFF when "
                           "sga_set_dense ran (its dense source is released), and as CSR it does not qualify (integer J in "
                           "strictly sorted rows, sum_j |J_ij| < 2^15, h in multiples of 1/2) -- call sga_set_field_cache "
                           "before sga_set_dense"
                         : "cached local fields over CSR couplings need integer-valued symmetric J in strictly sorted rows "
                           "(no duplicates), zero diagonal, max_i sum_j |J_ij| < 2^15 and h in multiples of 1/2";
        else if (e->R > 0 && (sga::sweep_clf_csr_lds_bytes(ldf, e->sstride, e->table_m) > 160 * 1024 ||
                              (e->slotted ? (e->max_row_len + 63) / 64 * 64 : e->max_row_len) > 4 * 64 * 8))
            reason = "cached local fields: fields and spins of a replica do not fit LDS (or a row is longer than 2048 entries)";
    } else if (e->tsp) reason = "cached local fields: stored couplings only";
    else if (!e->clf_problem)
        reason = "cached local fields need one model with integer-valued symmetric J, zero diagonal, h in "
                 "multiples of 1/2 and row sums below 2^24";
    else if (e->R > 0 && sga::sweep_clf_lds_bytes((e->ldj + 127) / 128 * 128, e->clf_bits, e->sstride,
                                                  e->clf_scale == 2 ? 2048 : e->table_m) > 160 * 1024)
        reason = "cached local fields: fields and spins of a replica do not fit LDS";
    if (why) *why = reason;
    return reason == nullptr;
}
bool clf_active(const sga_engine *e) {
    return e->field_cache != SGA_FIELD_CACHE_OFF && e->rule != SGA_RULE_WOLFF && clf_possible(e, nullptr);
}
// resident fields of every replica, from the all-replica pass (the tracked energies are left alone)
int ensure_fields(sga_engine *e) {
    if (e->fields_valid && e->fields) return SGA_OK;
    if (e->csr) {  // D = J s of every replica (int16), eight replicas per pass over the entries; scale * h once
        e->ldf = ((long long)e->n + 127) / 128 * 128;
        if (!e->fields && hipMalloc(&e->fields, (size_t)e->R * (size_t)e->ldf * 2) != hipSuccess) {
            (void)hipGetLastError();
            e->fields = nullptr;
            return fail(SGA_ERR_MEMORY, "no memory for the resident local fields of the cached-field sweep");
        }
        if (!e->hq) {
            HIPCHK(hipMalloc(&e->hq, sizeof(int) * (size_t)e->n));
            HIPCHK(sga::launch_scaled_fields(e->h, e->n, e->table_scale, e->hq, e->stream));
        }
        HIPCHK(sga::launch_csr_fields_seed(e->rowptr64, e->cv, e->spins, e->sstride, e->n, e->R,
                                           static_cast<short *>(e->fields), e->ldf, e->stream));
        e->fields_valid = true;
        return SGA_OK;
    }
    e->ldf = (e->ldj + 127) / 128 * 128;
    if (!e->fields && hipMalloc(&e->fields, (size_t)e->R * (size_t)e->ldf * (size_t)(e->clf_bits / 8)) != hipSuccess) {
        (void)hipGetLastError();
        e->fields = nullptr;
        return fail(SGA_ERR_MEMORY, "no memory for the resident local fields of the cached-field sweep");
    }
    int rc = fields_pass(e, 0, e->R, nullptr, e->fields);  // (any replica count: short tiles are clamped)
    if (rc != SGA_OK) return rc;
    e->fields_valid = true;
    return SGA_OK;
}

int recompute_energy_range(sga_engine *e, int r0, int count) {
    if (fields_pass_applies(e, count)) {
        const int rc = fields_pass(e, r0, count, e->energy + r0, nullptr);
        if (rc != SGA_ERR_MEMORY) return rc;  // (no room for its scratch: the per-replica kernels below need none)
    }
    const long long batched = e->opt[OPT_BATCHED_ENERGY];  // (one switch for both passes)
    const bool csr_all = batched == 2 || (batched == 1 && e->csr_acc != sga::CSR_ACC_F64_CANON);
    if (e->csr && !e->tsp && count >= 64 && csr_all) {
        // all replicas in one pass over the entries: spins transposed to bits, 32 replicas per lane
        // row groups: enough (group, replica word) threads to fill the chip -- ~4 waves per SIMD -- whatever
        // the replica count (256 replicas = 8 words: 4096 groups left half the SIMDs without a wave)
        const int words = (count + 31) / 32;
        const int groups = std::max(1, std::min(e->n, std::max(1024, 262144 / words)));
        if (e->csr_energy.reserve(sga::csr_energy_scratch_bytes(e->n, count, groups)) == hipSuccess) {
            const bool exact32 = e->csr_acc == sga::CSR_ACC_F32_TABLE || e->csr_acc == sga::CSR_ACC_F32;
            HIPCHK(sga::launch_energy_csr_all(e->rowptr64, e->cv, e->h, e->spins + (long long)r0 * e->sstride, e->sstride,
                                              e->n, count, groups, exact32, e->csr_energy.ptr, e->energy + r0, e->stream));
            return SGA_OK;
        }
        (void)hipGetLastError();  // no room for the transposed spin bits / partial sums: one pass per replica instead
    }
    sga::EnergyArgs a{};
    a.J = e->J_packed;
    a.rowptr = e->rowptr64;
    a.cv = e->cv;
    a.h = e->h;
    a.spins = e->spins + (long long)r0 * e->sstride;
    a.energy = e->energy + r0;
    a.ld = e->ld;
    a.ldj = e->ldj;
    a.n = e->n;
    a.sstride = e->sstride;
    a.R = count;
    a.reps_per_model = e->n_models > 1 ? e->Rg / e->n_models : 0;
    a.replica_base = e->replica0 + r0;
    a.model_stride_j = (long long)e->n * e->ldj;
    // few replicas: spread each replica's rows over several workgroups (one workgroup reading all
    // of J took 47 ms at n = 10^4 -- longer than the reference's CPU mv)
    a.slices = count >= 512 ? 1 : std::max(1, std::min({256, (1024 + count - 1) / count, e->n / 8}));
    if (a.slices > 1) {
        const size_t need = sizeof(double) * 2 * (size_t)count * a.slices;
        if (need > e->epart_bytes) {
            dev_free(e->epart);
            HIPCHK(hipMalloc(&e->epart, need));
            e->epart_bytes = need;
        }
        a.partial = e->epart;
    }
    HIPCHK(e->tsp ? sga::launch_energy_tsp(a, e->tsp_args, e->stream)
           : e->csr ? sga::launch_energy_csr(a, e->stream)
                    : sga::launch_energy_dense(a, e->want_i8, e->stream));
    HIPCHK(sga::launch_energy_finish(a.partial, a.slices, a.energy, count, e->stream));
    return SGA_OK;
}

// Launch geometry of the dense kernels for the current replica count / tuning.  The packed
// matrices are laid out by n alone (pack_dense, at set time), so a change of geometry never
// touches them.
int ensure_packed(sga_engine *e) {
    if (e->csr || e->tsp) return SGA_OK;
    if (!e->J_packed) return fail(SGA_ERR_INVALID, "no couplings set");
    int W, CPW;
    long long ld;
    if (e->use_t2) {
        // bit-plane geometry first; the int8 layout (energy / single-site kernels, non-LEAN
        // sweeps) shares its row length: 8 waves x (Wb * CPWb) chunks of 1024 int8
        int Wb, Cb;
        // at most 4 chunks per wave: a bit-plane chunk costs 8 VGPRs per ring slot
        choose_geometry(e->n, T2_ELEMS_PER_CHUNK, std::max(e->R, 1), e->tune_waves, Wb, Cb,
                        sga::T2_MAX_CPW);
        ld = (long long)Wb * Cb * T2_ELEMS_PER_CHUNK;
        W = 8;
        CPW = Wb * Cb;
        if (e->ld == ld && e->waves_t2 == Wb && e->cpw_t2 == Cb) return SGA_OK;
        if (sga::sweep_dense_lds_bytes(ld, e->table_m, false) > 160 * 1024)
            return fail(SGA_ERR_UNSUPPORTED, "replica spins do not fit LDS (n too large)");
        e->waves_t2 = Wb;
        e->cpw_t2 = Cb;
    } else if (e->acc_canon) {
        // canonical summation order: a wave owns whole super-chunks of 4 chunks (1024 fp32 elements),
        // one or two of them in registers; longer rows take the streaming form on 16 waves
        int S;
        choose_geometry(e->n, 4 * elems_per_chunk(false), std::max(e->R, 1), e->tune_waves, W, S, 2, 4);
        CPW = 4 * S;
        ld = (long long)W * CPW * elems_per_chunk(false);
        if (e->waves == W && e->cpw == CPW && e->ld == ld) return SGA_OK;
        if (sga::sweep_dense_lds_bytes(ld, e->table_m, true) > 160 * 1024)
            return fail(SGA_ERR_UNSUPPORTED, "replica spins do not fit LDS (n too large)");
    } else {
        choose_geometry(e->n, elems_per_chunk(e->want_i8), std::max(e->R, 1), e->tune_waves, W, CPW);
        ld = (long long)W * CPW * elems_per_chunk(e->want_i8);
        if (e->waves == W && e->cpw == CPW && e->ld == ld) return SGA_OK;
        if (sga::sweep_dense_lds_bytes(ld, e->table_m, false) > 160 * 1024)
            return fail(SGA_ERR_UNSUPPORTED, "replica spins do not fit LDS (n too large)");
    }
    e->waves = W;
    e->cpw = CPW;
    e->ld = ld;
    return SGA_OK;
}

// Pack the caller's fp32 matrix (device pointer `src`, row stride ld_src) into the engine's
// layout(s): rows packed to 128 bytes, not padded to the kernel's whole chunks (2.4 % fewer bytes
// per attempt at n = 10^4); lanes past a row's end re-read its first granule.
int pack_dense(sga_engine *e, const float *src, long long ld_src) {
    const long long rows = (long long)e->n_models * e->n;
    const long long elem = e->want_i8 ? 1 : 4;
    const long long ldj = ((long long)e->n * elem + 127) / 128 * 128 / elem;
    const size_t bytes = (size_t)rows * ldj * elem;
    HIPCHK(hipMalloc(&e->J_packed, bytes));
    HIPCHK(sga::launch_repack_dense(src, ld_src, rows, e->n, e->J_packed, ldj, e->want_i8, e->diag,
                                    e->stream));
    e->ldj = ldj;
    if (e->use_t2) {
        // a plane's rows are packed at 16-byte granularity, not padded to the kernel's 1-KiB chunks
        // (n = 10^4: 1264 B instead of 2048 B per row and plane -- this form is bound by the bytes
        // it pulls through the cache hierarchy); the kernel masks the lanes past a row's end
        const long long row_bits = t2_row_bits(e->n);
        HIPCHK(hipMalloc(&e->J_bits, sizeof(unsigned int) * 2 * (size_t)e->n * (size_t)(row_bits / 32)));
        HIPCHK(hipMalloc(&e->row_nnz, sizeof(float) * (size_t)e->n));
        HIPCHK(sga::launch_repack_tern2(src, ld_src, e->n, e->J_bits, row_bits, e->row_nnz, e->stream));
    }
    return SGA_OK;
}

}  // namespace

extern "C" {

const char *sga_last_error(void) { return g_last_error.c_str(); }
int sga_version(void) { return 400; }  // round 4: + sga_set_option / sga_get_option / sga_option_name, per-replica routing

int sga_create(int device, sga_engine **out) {
    if (!out) return fail(SGA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(SGA_ERR_DEVICE, "no HIP device available (the engine has no CPU fallback)");
    if (device < 0 || device >= count)
        return fail(SGA_ERR_DEVICE, "device index " + std::to_string(device) + " out of range (" +
                                        std::to_string(count) + " visible)");
    HIPCHK(hipSetDevice(device));
    sga_engine *eng = new (std::nothrow) sga_engine();
    if (!eng) return fail(SGA_ERR_MEMORY, "host allocation failed");
    eng->device = device;
    for (int i = 0; i < OPT_COUNT; ++i) {  // the ONE place the library reads the environment
        const OptDef &d = OPT_DEFS[i];
        eng->opt[i] = d.def;
        const char *v = d.env ? std::getenv(d.env) : nullptr;
        if (v) eng->opt[i] = d.env_presence ? d.env_value : std::max(d.lo, std::min(d.hi, (long long)std::atoll(v)));
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            eng->cus = cus;
    }
    e = hipStreamCreateWithFlags(&eng->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete eng;
        return fail(SGA_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    eng->stream = eng->own_stream;
    e = hipMalloc(&eng->d_count, sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&eng->d_flags, 16 * sizeof(int));
    if (e != hipSuccess) {
        dev_free(eng->d_count);
        (void)hipStreamDestroy(eng->own_stream);
        delete eng;
        return fail(SGA_ERR_MEMORY, "hipMalloc failed");
    }
    *out = eng;
    return SGA_OK;
}

void sga_destroy(sga_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (auto &p : e->events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    e->free_replicas();
    e->free_problem();
    for (auto &sl : e->scratch) sl.release();
    e->point_sites.release();
    e->point_out.release();
    e->csr_energy.release();
    dev_free(e->d_count);
    dev_free(e->d_flags);
    if (e->fork_ev) (void)hipEventDestroy(e->fork_ev);
    if (e->join_ev) (void)hipEventDestroy(e->join_ev);
    if (e->aux_stream) (void)hipStreamDestroy(e->aux_stream);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int sga_set_stream(sga_engine *e, void *hip_stream) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    return SGA_OK;
}

int sga_set_option(sga_engine *e, const char *key, int64_t value) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    const int i = find_option(key);
    if (i < 0) return fail(SGA_ERR_INVALID, std::string("unknown option: ") + (key ? key : "(null)"));
    const OptDef &d = OPT_DEFS[i];
    if (value < d.lo || value > d.hi)
        return fail(SGA_ERR_INVALID, std::string("option ") + key + ": value outside [" + std::to_string(d.lo) + ", " +
                                         std::to_string(d.hi) + "]");
    e->opt[i] = (long long)value;
    return SGA_OK;
}

int sga_get_option(sga_engine *e, const char *key, int64_t *value) {
    if (!e || !value) return fail(SGA_ERR_INVALID, "NULL argument");
    const int i = find_option(key);
    if (i < 0) return fail(SGA_ERR_INVALID, std::string("unknown option: ") + (key ? key : "(null)"));
    *value = (int64_t)e->opt[i];
    return SGA_OK;
}

int sga_option_name(int index, char *buf, int buflen) {
    if (!buf || buflen <= 0) return fail(SGA_ERR_INVALID, "bad arguments");
    if (index < 0 || index >= OPT_COUNT) return fail(SGA_ERR_INVALID, "no such option");
    std::snprintf(buf, (size_t)buflen, "%s", OPT_DEFS[index].key);
    return SGA_OK;
}

int sga_set_csr_storage(sga_engine *e, int storage) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (storage != SGA_CSR_STORAGE_AUTO && storage != SGA_CSR_STORAGE_F32 && storage != SGA_CSR_STORAGE_PACKED)
        return fail(SGA_ERR_INVALID, "bad CSR storage");
    e->csr_storage = storage;
    return SGA_OK;
}

int sga_set_tuning(sga_engine *e, int waves_per_replica, int sweeps_per_launch) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (waves_per_replica < 0 || waves_per_replica > sga::MAX_WAVES || sweeps_per_launch < 0)
        return fail(SGA_ERR_INVALID, "bad tuning values");
    e->tune_waves = waves_per_replica;
    e->tune_spl = sweeps_per_launch;
    return SGA_OK;
}

int sga_set_field_cache(sga_engine *e, int mode) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (mode != SGA_FIELD_CACHE_OFF && mode != SGA_FIELD_CACHE_ON && mode != SGA_FIELD_CACHE_AUTO)
        return fail(SGA_ERR_INVALID, "bad field-cache mode");
    e->field_cache = mode;
    return SGA_OK;
}

// Measured choice of the sweep FORM of a CSR problem (round 4).  The forms of sga_init_replicas -- waves per
// replica (1, 2, 4, 8: a row dealt to several waves), spins as int8 or bits, several updates per step or one -- are
// picked by thresholds measured on a few instance families; here every candidate that the problem admits runs the
// real sweep kernel on the real replicas.  The state travels through the geometry-independent checkpoint blob
// (sga_export_state / sga_import_state), so the run continues exactly as if this call had not happened; the chain
// does not depend on the form.  The winner stays as sga_set_tuning / "csr_updates_per_step" would have set it.
static int autotune_csr(sga_engine *e, double *best_ms_per_sweep) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    uint64_t need = 0;
    int rc = sga_export_state(e, nullptr, 0, &need);
    if (rc != SGA_OK) return rc;
    std::vector<unsigned char> blob((size_t)need);
    rc = sga_export_state(e, blob.data(), need, nullptr);
    if (rc != SGA_OK) return rc;
    const int R = e->R, Rg = e->Rg, replica0 = e->replica0, n_ladders = e->n_ladders;
    const uint64_t seed = e->seed;
    std::vector<double> ladder;
    if (n_ladders > 0) {
        ladder.resize((size_t)Rg);
        HIPCHK(hipMemcpy(ladder.data(), e->slot_temps, sizeof(double) * (size_t)Rg, hipMemcpyDeviceToHost));
    }
    const int user_waves = e->tune_waves, user_spl = e->tune_spl, user_cache = e->field_cache;
    const long long user_ups = e->opt[OPT_CSR_UPDATES_PER_STEP];
    const bool was_timing = e->timing;
    e->field_cache = SGA_FIELD_CACHE_OFF;  // (the forms are the row-per-proposal kernels')
    // lay the replicas out for a candidate and put the saved state back
    auto layout = [&](int waves, long long ups) -> int {
        e->tune_waves = waves;
        e->opt[OPT_CSR_UPDATES_PER_STEP] = ups;
        int r2 = sga_init_replicas(e, R, Rg, replica0, seed, nullptr);
        if (r2 == SGA_OK && n_ladders > 0) r2 = sga_set_ladder(e, ladder.data(), n_ladders);
        if (r2 == SGA_OK) r2 = sga_import_state(e, blob.data(), need);
        return r2;
    };
    auto timed = [&](int k, double &ms) -> int {
        e->tune_spl = k;
        e->timing = true;
        int64_t launches = 0;
        double t = 0.0;
        (void)sga_get_kernel_time(e, &launches, &t, 1);
        int r2 = sga_sweep(e, k, SGA_SITE_RANDOM, SGA_ARITH_F64, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (r2 != SGA_OK) return r2;
        r2 = sga_get_kernel_time(e, &launches, &t, 1);
        ms = t;
        return r2;
    };
    struct Cand {
        int waves;
        long long ups;
    };
    std::vector<Cand> cands = {{0, -1}, {1, -1}, {2, -1}, {4, -1}, {8, -1}};
    if (e->max_row_len <= 256) cands.push_back({0, 0}), cands.push_back({1, 0});  // (one update at a time)
    double best = 1e300;
    int best_i = -1;
    char seen[16][96];
    int n_seen = 0;
    for (size_t i = 0; i < cands.size(); ++i) {
        if (layout(cands[i].waves, cands[i].ups) != SGA_OK) {
            (void)hipGetLastError();
            continue;  // (a form the problem does not admit)
        }
        double t1 = 0.0, t = 0.0;
        if (timed(1, t1) != SGA_OK) continue;
        // the same kernel form as an earlier candidate?  (the heuristic's choice is one of the explicit ones)
        bool dup = false;
        for (int q = 0; q < n_seen; ++q) dup = dup || std::strncmp(seen[q], sga::last_sweep_kernel(), 95) == 0;
        if (dup) continue;
        if (n_seen < 16) std::snprintf(seen[n_seen++], 96, "%s", sga::last_sweep_kernel());
        const int k = t1 > 0.0 ? (int)std::min(32.0, std::max(1.0, std::ceil(2.0 / t1))) : 1;
        if (layout(cands[i].waves, cands[i].ups) != SGA_OK || timed(k, t) != SGA_OK) continue;
        if (t / k < best * 0.995) {  // (ties go to the earlier, simpler candidate)
            best = t / k;
            best_i = (int)i;
        }
    }
    e->timing = was_timing;
    e->tune_spl = user_spl;
    e->field_cache = user_cache;
    e->fields_valid = false;
    dev_free(e->fields);
    rc = layout(best_i >= 0 ? cands[(size_t)best_i].waves : user_waves, best_i >= 0 ? cands[(size_t)best_i].ups : user_ups);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (rc != SGA_OK) return rc;
    if (best_i >= 0 && best_ms_per_sweep) *best_ms_per_sweep = best;
    return SGA_OK;
}

// Measured choice of the dense launch geometry.  Every candidate (waves per replica) runs the
// real sweep kernel on the real replicas for a trial; the chain does not depend on the geometry,
// and spins / energies / best states / counters are put back afterwards, so the run continues
// exactly as if this call had not happened.
int sga_autotune(sga_engine *e, double *best_ms_per_sweep) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas (call sga_init_replicas)");
    if (best_ms_per_sweep) *best_ms_per_sweep = 0.0;
    if (e->tsp) return SGA_OK;
    if (e->csr) return autotune_csr(e, best_ms_per_sweep);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int n = e->n, R = e->R;
    const size_t cb = (size_t)R * n;
    // the state, independent of the spin stride
    int8_t *spins_c = nullptr, *best_c = nullptr;
    double *en = nullptr, *ben = nullptr;
    unsigned long long *acc = nullptr;
    auto release = [&]() {
        dev_free(spins_c);
        dev_free(best_c);
        dev_free(en);
        dev_free(ben);
        dev_free(acc);
    };
    struct Guard {  // every exit path, the HIPCHK returns included, releases the saved state
        decltype(release) &fn;
        ~Guard() { fn(); }
    } guard{release};
    hipError_t he = hipMalloc(&spins_c, cb);
    if (he == hipSuccess) he = hipMalloc(&best_c, cb);
    if (he == hipSuccess) he = hipMalloc(&en, sizeof(double) * R);
    if (he == hipSuccess) he = hipMalloc(&ben, sizeof(double) * R);
    if (he == hipSuccess) he = hipMalloc(&acc, sizeof(unsigned long long) * R);
    if (he == hipSuccess) he = sga::launch_unpad_spins(e->spins, e->sstride, spins_c, n, R, e->stream);
    if (he == hipSuccess) he = sga::launch_unpad_spins(e->best_spins, e->sstride, best_c, n, R, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(en, e->energy, sizeof(double) * R, hipMemcpyDeviceToDevice, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(ben, e->best_energy, sizeof(double) * R, hipMemcpyDeviceToDevice, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(acc, e->n_acc, sizeof(unsigned long long) * R, hipMemcpyDeviceToDevice, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess) {
        release();
        return fail(SGA_ERR_DEVICE, hipGetErrorString(he));
    }
    const uint32_t sweeps_done = e->sweeps_done;
    const long long attempted = e->attempted;
    const int user_waves = e->tune_waves, user_spl = e->tune_spl;
    const bool was_timing = e->timing;
    const int user_cache = e->field_cache;  // the geometry belongs to the row-per-proposal kernels
    e->field_cache = SGA_FIELD_CACHE_OFF;

    // lay the replicas out for `waves` (0 = heuristic) and put the saved state back
    auto layout = [&](int waves) -> int {
        e->tune_waves = waves;
        int rc = ensure_packed(e);
        if (rc != SGA_OK) return rc;
        if (e->sstride != (int)e->ld) {
            dev_free(e->spins);
            dev_free(e->best_spins);
            e->sstride = (int)e->ld;
            hipError_t me = hipMalloc(&e->spins, (size_t)R * e->sstride);
            if (me == hipSuccess) me = hipMalloc(&e->best_spins, (size_t)R * e->sstride);
            if (me != hipSuccess) {  // no half-allocated replica set: the engine is back to "no replicas"
                e->free_replicas();
                return fail(SGA_ERR_MEMORY, std::string("autotune layout: ") + hipGetErrorString(me));
            }
        }
        HIPCHK(sga::launch_pad_spins(spins_c, n, e->spins, e->sstride, R, e->stream));
        HIPCHK(sga::launch_pad_spins(best_c, n, e->best_spins, e->sstride, R, e->stream));
        HIPCHK(hipMemcpyAsync(e->energy, en, sizeof(double) * R, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->best_energy, ben, sizeof(double) * R, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->n_acc, acc, sizeof(unsigned long long) * R, hipMemcpyDeviceToDevice, e->stream));
        e->sweeps_done = sweeps_done;
        e->attempted = attempted;
        return SGA_OK;
    };
    // kernel time of k sweeps in one launch, ms
    auto timed = [&](int k, double &ms) -> int {
        e->tune_spl = k;
        e->timing = true;
        int64_t launches = 0;
        double t = 0.0;
        (void)sga_get_kernel_time(e, &launches, &t, 1);
        int rc = sga_sweep(e, k, SGA_SITE_RANDOM, SGA_ARITH_F64, nullptr, 0, 0, nullptr, nullptr,
                           nullptr, nullptr, nullptr);
        if (rc != SGA_OK) return rc;
        rc = sga_get_kernel_time(e, &launches, &t, 1);
        ms = t;
        return rc;
    };

    const int epc = e->use_t2 ? T2_ELEMS_PER_CHUNK : (e->acc_canon ? 4 : 1) * elems_per_chunk(e->want_i8);
    const int max_cpw = e->use_t2 ? sga::T2_MAX_CPW : (e->acc_canon ? 2 : 8);
    const int C = (n + epc - 1) / epc;
    int best_w = -1;
    double best = 1e300;
    double per_w[sga::MAX_WAVES + 1];
    for (double &v : per_w) v = 1e300;
    int rc = SGA_OK;
    for (int w = 0; w <= sga::MAX_WAVES && rc == SGA_OK; ++w) {  // 0 = the heuristic's own choice
        if (w > 0) {
            const int cpw = (C + w - 1) / w;
            if (cpw > max_cpw || (w > C && w > 1)) continue;
        }
        rc = layout(w);
        if (rc != SGA_OK) break;
        double t1 = 0.0, t = 0.0;
        rc = timed(1, t1);  // warm-up and scale
        if (rc != SGA_OK) break;
        const int k = t1 > 0.0 ? (int)std::min(64.0, std::max(1.0, std::ceil(2.0 / t1))) : 1;
        rc = timed(k, t);
        if (rc != SGA_OK) break;
        const double per = t / k;
        per_w[w] = per;
        if (per < best) {
            best = per;
            best_w = w;
        }
    }
    // Several geometries usually lie within the timing noise of each other (n = 10^4 fp32: 9 x 5, 13 x 4 and
    // 14 x 3 within 0.5 %, and the winner changed from run to run on one box): among those within 0.5 % of the
    // fastest take the one with the fewest waves, so that repeated runs -- and a profile taken later -- see
    // the same instantiation.
    if (rc == SGA_OK && best_w >= 0)
        for (int w = 1; w <= sga::MAX_WAVES; ++w)
            if (per_w[w] <= best * 1.005) {
                best_w = w;
                break;
            }
    // leave with the winner (or the caller's setting if something failed) and the saved state
    e->timing = was_timing;
    e->tune_spl = user_spl;
    e->field_cache = user_cache;
    e->fields_valid = false;
    dev_free(e->fields);  // (the spin stride may have changed; rebuilt on demand)
    const int final_rc = e->R > 0 ? layout(rc == SGA_OK && best_w >= 0 ? best_w : user_waves) : SGA_ERR_MEMORY;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (rc != SGA_OK) return rc;
    if (final_rc != SGA_OK) return final_rc;
    if (best_ms_per_sweep) *best_ms_per_sweep = best;
    return SGA_OK;
}

int sga_probe_read_bandwidth(int device, int64_t bytes, int reps, double *gb_per_s) {
    if (!gb_per_s || bytes < (1 << 20) || reps < 1) return fail(SGA_ERR_INVALID, "bad probe arguments");
    HIPCHK(hipSetDevice(device));
    bytes &= ~(int64_t)15;
    void *buf = nullptr;
    float *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipMalloc(&buf, (size_t)bytes);
    if (he == hipSuccess) he = hipMalloc(&sink, sizeof(float));
    if (he == hipSuccess) he = hipMemset(buf, 0, (size_t)bytes);
    if (he == hipSuccess) he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    if (he == hipSuccess) he = sga::launch_probe_read(buf, bytes, sink, nullptr);  // warm-up
    if (he == hipSuccess) he = hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps && he == hipSuccess; ++i) he = sga::launch_probe_read(buf, bytes, sink, nullptr);
    if (he == hipSuccess) he = hipEventRecord(e1, nullptr);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    dev_free(buf);
    dev_free(sink);
    if (he != hipSuccess) return fail(SGA_ERR_DEVICE, hipGetErrorString(he));
    *gb_per_s = (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
    return SGA_OK;
}

int sga_set_dense(sga_engine *e, const float *J, int64_t ldJ, const float *h, int n, int storage) {
    return sga_set_dense_batch(e, J, ldJ, h, n, 1, storage);
}

static int set_csr_common(sga_engine *e, const void *rowptr, bool wide_extents, const int32_t *colidx, const float *val,
                          const float *h, int n, int64_t nnz);

// Sparse couplings handed over as a dense matrix (the reference's IsingModel is dense by default; its assignment
// and scheduling encoders fill 1-2 % of it): with SGA_J_AUTO, one model, n >= 4096, integer-valued J and no row of
// more than 256 non-zeros the problem is taken as CSR -- a proposal then reads its row's entries instead of n
// couplings, and the several-updates-per-step forms apply (sweep_csr_rows.hip).  Integer row sums are exact in
// either form, so the chain is the dense forms' bit for bit.  When: see the call (the cached-field sweep is a dense
// form); never with option "sparse_route" = 0 (A/B switch).
// Returns SGA_OK with *taken = true when the problem was set as CSR.
static int route_sparse_dense(sga_engine *e, const float *src, long long ld_src, const float *h, int n, bool *taken) {
    *taken = false;
    int *nnz_d = nullptr;
    HIPCHK(hipMalloc(&nnz_d, sizeof(int) * ((size_t)n + 1)));
    struct Guard {
        int *a = nullptr, *b = nullptr, *c = nullptr;
        float *v = nullptr;
        ~Guard() { dev_free(a), dev_free(b), dev_free(c), dev_free(v); }
    } g;
    g.a = nnz_d;
    HIPCHK(sga::launch_dense_row_nnz(src, ld_src, n, nnz_d, e->stream));
    std::vector<int> len((size_t)n), rp((size_t)n + 1);
    HIPCHK(hipMemcpyAsync(len.data(), nnz_d, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    long long total = 0;
    int longest = 0;
    for (int i = 0; i < n; ++i) {
        rp[(size_t)i] = (int)total;
        total += len[(size_t)i];
        longest = std::max(longest, len[(size_t)i]);
    }
    rp[(size_t)n] = (int)total;
    if (longest > 256 || total == 0 || total >= (long long)INT32_MAX) return SGA_OK;
    HIPCHK(hipMalloc(&g.b, sizeof(int) * ((size_t)n + 1)));
    HIPCHK(hipMalloc(&g.c, sizeof(int) * (size_t)total));
    HIPCHK(hipMalloc(&g.v, sizeof(float) * (size_t)total));
    HIPCHK(hipMemcpyAsync(g.b, rp.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, e->stream));
    HIPCHK(sga::launch_dense_to_csr(src, ld_src, n, g.b, g.c, g.v, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int rc = set_csr_common(e, g.b, false, g.c, g.v, h, n, total);
    if (rc == SGA_OK) {
        *taken = true;
        e->from_dense = true;
    }
    return rc;
}

int sga_set_dense_batch(sga_engine *e, const float *J, int64_t ldJ, const float *h, int n,
                        int n_models, int storage) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (!J || !h || n <= 0 || ldJ < n || n_models <= 0)
        return fail(SGA_ERR_INVALID, "bad dense problem arguments");
    if (storage < SGA_J_AUTO || storage > SGA_J_T2)
        return fail(SGA_ERR_INVALID, "bad storage selector");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->free_replicas();
    e->free_problem();
    e->csr = false;
    e->from_dense = false;
    e->table_m = 0;
    e->n = n;
    e->n_models = n_models;
    const long long rows = (long long)n_models * n;
    // A device matrix is scanned and packed where it lies; a host matrix is staged first.  Either
    // way nothing but the packed layout(s) stays resident (400 MB, not 800, at n = 10^4 fp32).
    const float *src = J;
    long long ld_src = ldJ;
    struct Staged {
        float *p = nullptr;
        ~Staged() { dev_free(p); }
    } staged;
    if (!is_device_ptr(J)) {
        HIPCHK(hipMalloc(&staged.p, sizeof(float) * (size_t)rows * n));
        HIPCHK(hipMemcpy2DAsync(staged.p, sizeof(float) * (size_t)n, J, sizeof(float) * (size_t)ldJ,
                                sizeof(float) * (size_t)n, (size_t)rows, hipMemcpyHostToDevice, e->stream));
        src = staged.p;
        ld_src = n;
    }
    HIPCHK(hipMalloc(&e->h, sizeof(float) * (size_t)rows));
    HIPCHK(hipMemcpyAsync(e->h, h, sizeof(float) * (size_t)rows, hipMemcpyDefault, e->stream));
    HIPCHK(hipMalloc(&e->diag, sizeof(float) * (size_t)rows));
    // value scans over all models: can J live in int8; is fp32 accumulation exact; is the
    // problem integer valued with few possible uphill moves (per-sweep accept table); is J
    // symmetric with a zero diagonal (dE of the rule == energy change)?
    int *flags = e->d_flags;  // [0..3] value scans, [4] symmetry / diagonal
    unsigned int *uflags = reinterpret_cast<unsigned int *>(flags) + 2;
    int hflags[8] = {1, 1, 0, 1, 1, 0, 0, 0};  // ([7]: max |J_ij| as float bits, launch_dense_row_abs_max)
    HIPCHK(hipMemsetAsync(flags, 0, 8 * sizeof(int), e->stream));
    HIPCHK(sga::launch_scan_values(src, rows, n, ld_src, flags, e->stream));
    HIPCHK(sga::launch_dense_row_abs_max(src, ld_src, e->h, rows, n, uflags, e->stream));
    HIPCHK(sga::launch_check_symmetric(src, ld_src, rows, n, flags + 4, e->stream));
    HIPCHK(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->consistent_dE = hflags[4] == 0;
    const bool fits_i8 = hflags[0] == 0;
    if (storage == SGA_J_I8 && !fits_i8)
        return fail(SGA_ERR_INVALID, "int8 storage requested but J is not integer in [-127,127]");
    const bool ternary = hflags[1] == 0 && n_models == 1;
    if (storage == SGA_J_T2 && !ternary)
        return fail(SGA_ERR_INVALID, "bit-plane storage needs one model with J in {-1, 0, +1}");
    e->use_t2 = storage == SGA_J_T2 || (storage == SGA_J_AUTO && ternary && n >= 4096);
    e->want_i8 = e->use_t2 || (storage == SGA_J_I8) || (storage == SGA_J_AUTO && fits_i8);
    float m;
    std::memcpy(&m, &hflags[2], sizeof(float));
    const unsigned nonint = (unsigned)hflags[3];  // bit 0: some J, bit 1: some h not an integer
    // fp32 partial sums are exact (any order) when J is integer valued and no row's sum of
    // |J| reaches 2^24; otherwise the row sum is accumulated in fp64
    e->acc64 = !e->want_i8 && !((nonint & 1u) == 0u && m < 16777216.0f);
    {
        // ... and the fp64 sum of a row's (exact) fp32 products is exact in ANY order when the set bits
        // of all J lie within 53 binary places of each other, the row's carries included; only
        // couplings of a wider dynamic range (e.g. Gaussian J: tiny values next to large ones) need the
        // canonical summation order and its one tree per 256-element chunk
        int carry = 0;
        while ((1ll << carry) < n) ++carry;
        const bool any = hflags[5] != 0;
        const int span = (hflags[5] - 1024) - (1024 - hflags[6]) + 1;
        e->acc_canon = e->acc64 && any && span + carry > 52;
        if (e->opt[OPT_FORCE_DENSE_CANON]) e->acc_canon = e->acc64;  // parity tests
    }
    // integer problem: tabulate exp(float32(-2k/T)) for the moves k <= min(M, 2048) per sweep
    if (nonint == 0u && m >= 1.0f && m < 16777216.0f) e->table_m = (int)std::min(m, 2048.0f);
    // cached-local-field sweep: exact integer fields, dE of the rule == energy change, one model
    // (h a multiple of 1/2 -- the penalty encodings of 0/1 variables -- keeps 2 F an integer: scale 2)
    e->row_abs_max = m;
    {
        float jm;
        std::memcpy(&jm, &hflags[7], sizeof(float));
        e->j_abs_max = (int)std::min(std::ceil((double)jm), 16777216.0);
    }
    e->clf_scale = (nonint & 2u) ? 2 : 1;
    e->clf_problem = (nonint & 5u) == 0u && (double)m * e->clf_scale < 16777216.0 && e->consistent_dE && n_models == 1;
    e->clf_bits = (double)m * e->clf_scale < 32768.0 ? 16 : 32;
    // Sparse matrix?  (route_sparse_dense above.)  Taken when the caller asked for one row read per proposal
    // (field cache OFF), or left the choice (AUTO) on a problem the cached-field sweep cannot serve: where that
    // sweep applies it is the better form while few proposals are accepted (C2b, 1024 replicas, acceptance 2 %:
    // dense int8 rows 7.7e8, as CSR four updates per step 4.3e9, cached fields 1.06e10 attempts/s).
    if (storage == SGA_J_AUTO && n_models == 1 && n >= 4096 && (nonint & 1u) == 0u &&
        (e->field_cache == SGA_FIELD_CACHE_OFF || (e->field_cache == SGA_FIELD_CACHE_AUTO && !e->clf_problem)) &&
        e->opt[OPT_SPARSE_ROUTE] != 0) {
        bool taken = false;
        const int rcr = route_sparse_dense(e, src, ld_src, h, n, &taken);  // (h: the caller's pointer)
        if (rcr != SGA_OK || taken) return rcr;
    }
    int rc = pack_dense(e, src, ld_src);
    if (rc == SGA_OK) rc = ensure_packed(e);
    // the source (the caller's buffer, or the staging copy about to be released) is done with
    HIPCHK(hipStreamSynchronize(e->stream));
    return rc;
}

// Row extents of the layout the kernels read: dst[i] = prefix sum of the rows' lengths, each rounded
// up to whole 64-entry slots when `slotted`.  n <= ~1.3e6 rows: done on the host at set time.
//
// Slotted layouts also get the wide forms' per-row record (rowinfo: first slot, slot count | entries in
// the last slot << 24, slots from the first slot to an all-zero slot, h): a wave asks for a fixed number of slots per row and
// the ones past the row's end read that zero slot (value 0: nothing to mask when the row is
// summed).  The zero slot is the 64 zeroed entries behind the array; layouts beyond 2^21 slots
// (1 GB) get one more inside after every 2^21 slots -- it rides at the end of the row before it,
// like slot padding -- so that the offset always fits the 32-bit lane offset of a load.
static int build_layout(sga_engine *e, const std::vector<long long> &src, bool slotted) {
    const int n = e->n;
    long long ZERO_SLOT_EVERY = 1ll << 21;
    if (e->opt[OPT_ZERO_SLOT_EVERY] > 0)  // parity tests: zero slots inside small layouts
        ZERO_SLOT_EVERY = std::max(1ll, std::min(ZERO_SLOT_EVERY, e->opt[OPT_ZERO_SLOT_EVERY]));
    std::vector<long long> dst((size_t)n + 1);
    std::vector<int4> info(slotted ? (size_t)n : 0);
    std::vector<int32_t> narrow;
    std::vector<std::pair<int, long long>> zero_after;  // (row, slot number) of the zero slots inside
    long long at = 0, since = 0;
    for (int i = 0; i < n; ++i) {
        dst[(size_t)i] = at;
        const long long len = src[(size_t)i + 1] - src[(size_t)i];
        // (src may be a padded layout being re-padded: slot padding never adds a slot)
        e->max_row_len = i == 0 ? len : std::max(e->max_row_len, len);
        if (!slotted) {
            at += len;
            continue;
        }
        const long long slots = (len + 63) / 64;
        // .y: slot count | entries in the last slot << 24 (lanes beyond them read the zero slot: no HBM
        // traffic for the padding's cache lines)
        if (slots >= (1 << 24)) return fail(SGA_ERR_UNSUPPORTED, "CSR row too long for the slot addressing");
        info[(size_t)i] = make_int4((int)(at >> 6), (int)(slots | ((len - 64 * (slots - 1)) << 24)), 0, 0);
        if (slots == 0) info[(size_t)i].y = 0;
        at += slots * 64;
        since += slots;
        if (since >= ZERO_SLOT_EVERY && i + 1 < n) {
            zero_after.emplace_back(i, at >> 6);
            at += 64;
            since = 0;
        }
    }
    dst[(size_t)n] = at;
    if (slotted) {
        if ((at >> 6) >= (long long)INT32_MAX) return fail(SGA_ERR_UNSUPPORTED, "CSR problem too large");
        zero_after.emplace_back(n - 1, at >> 6);  // the zeroed entries behind the array
        size_t z = 0;
        for (int i = 0; i < n; ++i) {
            while (zero_after[z].first < i) ++z;
            info[(size_t)i].z = (int)(zero_after[z].second - info[(size_t)i].x);
            if (info[(size_t)i].z >= (1 << 23)) return fail(SGA_ERR_UNSUPPORTED, "CSR row too long for the slot addressing");
        }
    }
    dev_free(e->rowptr);
    dev_free(e->rowinfo);
    const size_t np1 = (size_t)n + 1;
    HIPCHK(hipMemcpyAsync(e->rowptr64, dst.data(), sizeof(long long) * np1, hipMemcpyHostToDevice, e->stream));
    if (at < (long long)INT32_MAX) {
        narrow.assign(dst.begin(), dst.end());
        HIPCHK(hipMalloc(&e->rowptr, sizeof(int32_t) * np1));
        HIPCHK(hipMemcpyAsync(e->rowptr, narrow.data(), sizeof(int32_t) * np1, hipMemcpyHostToDevice, e->stream));
    }
    if (slotted) {
        HIPCHK(hipMalloc(&e->rowinfo, sizeof(int4) * (size_t)n));
        HIPCHK(hipMemcpyAsync(e->rowinfo, info.data(), sizeof(int4) * (size_t)n, hipMemcpyHostToDevice, e->stream));
        HIPCHK(sga::launch_rowinfo_fields(e->rowinfo, e->h, n, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));  // the host vectors go out of scope
    e->slotted = slotted;
    e->layout_entries = at;
    return SGA_OK;
}

// Narrow CSR forms of integer problems whose longest row has <= 64 entries: 0 = one update at a time,
// 1 | 2 = the pair look-ahead (opt-in, option "csr_updates_per_step": round 3 measured -1 ... +3 % on BASELINE configs[2]),
// 4 | 8 = that many updates per step, one per row of 16 | 8 lanes (sweep_csr_rows.hip; the launcher takes it
// for production arguments -- Philox sites, Metropolis with the accept table): the default where it applies
// (profiles/r03_experiments.md 4b: C3, rows of up to 50 entries, 1.0e10 | 2.87e10 | 2.47e10 attempts/s for
// 1 | 4 | 8 updates per step; degree ~16: 1.0e10 | 3.7e10 | 5.1e10).  Option value 0 turns it off.
// Rows of 65 ... 256 entries (assignment / small scheduling problems: degree 100-250, cache resident, bound by the
// one-update chain): four per step with 8 | 16 entries per lane, integer problems with the accept table only.
static bool csr_rows_medium(const sga_engine *e) {
    return e->csr && e->max_row_len > 64 && e->max_row_len <= 256 && e->csr_acc == sga::CSR_ACC_F32_TABLE && e->table_m > 0;
}
static int csr_updates_per_step(const sga_engine *e) {
    if (!e->csr || e->max_row_len > 256) return 0;
    const bool medium = e->max_row_len > 64;
    if (medium && !csr_rows_medium(e)) return 0;
    int v = e->max_row_len <= 32 ? 8 : 4;
    if (e->opt[OPT_CSR_UPDATES_PER_STEP] >= 0) v = (int)e->opt[OPT_CSR_UPDATES_PER_STEP];
    if (v != 1 && v != 2 && v != 4 && v != 8) return 0;
    if (medium) v = v >= 4 ? 4 : 0;  // (the pair look-ahead holds one wave-load per row)
    if (v >= 4 && (e->layout_entries + CSR_TAIL_PAD) * 8 >= (1ll << 32)) return 0;  // (32-bit byte offsets of the entries)
    return v;
}

// The wide sweep forms (a row dealt to several waves) address rows by 64-entry slots: re-pad an
// unpadded layout on demand (short-row problems run wide only when tuning asks for it).
static int ensure_slotted(sga_engine *e) {
    if (!e->csr || e->slotted) return SGA_OK;
    const size_t np1 = (size_t)e->n + 1;
    std::vector<long long> src(np1);
    HIPCHK(hipMemcpy(src.data(), e->rowptr64, sizeof(long long) * np1, hipMemcpyDeviceToHost));
    long long *old_ptr = nullptr;
    HIPCHK(hipMalloc(&old_ptr, sizeof(long long) * np1));
    hipError_t he = hipMemcpy(old_ptr, e->rowptr64, sizeof(long long) * np1, hipMemcpyDeviceToDevice);
    int2 *old_cv = e->cv;
    int rc = he == hipSuccess ? build_layout(e, src, true) : fail(SGA_ERR_DEVICE, hipGetErrorString(he));
    if (rc == SGA_OK) {
        e->cv = nullptr;
        he = hipMalloc(&e->cv, sizeof(int2) * (size_t)(e->layout_entries + CSR_TAIL_PAD));
        if (he == hipSuccess) he = hipMemsetAsync(e->cv + e->layout_entries, 0, sizeof(int2) * CSR_TAIL_PAD, e->stream);
        if (he == hipSuccess)
            he = sga::launch_pack_cv_rows(old_ptr, e->rowptr64, nullptr, nullptr, old_cv, e->cv, e->n, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (he != hipSuccess) rc = fail(he == hipErrorOutOfMemory ? SGA_ERR_MEMORY : SGA_ERR_DEVICE, hipGetErrorString(he));
        dev_free(old_cv);
    }
    dev_free(old_ptr);
    if (rc != SGA_OK) {
        // build_layout may already have overwritten the extents while the entries are still the old
        // ones (or gone): no half-converted layout survives -- the engine is back to "no couplings set"
        const std::string msg = g_last_error;
        (void)hipStreamSynchronize(e->stream);
        e->free_replicas();
        e->free_problem();
        return fail(rc, msg + " (re-padding the CSR layout failed: set the couplings again)");
    }
    return rc;
}

// Packed entries for the bit-spin wide forms of integer-valued problems (|J| <= 127, n < 2^24): one
// dword per entry, the same slots (256 bytes each) -- half the bytes of a row.  Built on demand from the
// slotted layout; the (column, value) layout stays (energy kernels, traced sweeps).
static int ensure_packed_entries(sga_engine *e) {
    if (e->cvp || e->cvp_tried) return SGA_OK;
    e->cvp_tried = true;
    if (!e->csr || !e->slotted || e->n >= (1 << 24) ||
        (e->csr_acc != sga::CSR_ACC_F32 && e->csr_acc != sga::CSR_ACC_F32_TABLE))
        return SGA_OK;
    const size_t count = (size_t)e->layout_entries + 64;  // the zero slot behind the array included
    hipError_t he = hipMalloc(&e->cvp, sizeof(uint32_t) * count);
    if (he != hipSuccess) {
        e->cvp = nullptr;
        (void)hipGetLastError();
        return SGA_OK;  // no room: the unpacked layout serves
    }
    int bad = 0;
    he = hipMemsetAsync(e->d_flags, 0, sizeof(int), e->stream);
    if (he == hipSuccess) he = sga::launch_pack_entries(e->cv, e->cvp, (long long)count, e->d_flags, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(&bad, e->d_flags, sizeof(int), hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess || bad) dev_free(e->cvp);
    if (he != hipSuccess) return fail(SGA_ERR_DEVICE, hipGetErrorString(he));
    return SGA_OK;
}

// CSR problem from 32- or 64-bit row extents (host or device pointers).  The structure is
// checked on the device -- a bad extent or column would fault in the sweep kernels -- and the
// same pass classifies the problem: integer valued (accept table, fp32-exact row sums),
// symmetric with zero diagonal (dE of the rule == energy change).  Device arrays are read where
// they lie, host arrays are staged; only the interleaved layout stays resident.
static int set_csr_common(sga_engine *e, const void *rowptr, bool wide_extents, const int32_t *colidx,
                          const float *val, const float *h, int n, int64_t nnz) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (!rowptr || !h || n <= 0 || nnz < 0 || (nnz > 0 && (!colidx || !val)))
        return fail(SGA_ERR_INVALID, "bad CSR problem arguments");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->free_replicas();
    e->free_problem();
    e->csr = true;
    e->from_dense = false;
    e->n = n;
    e->n_models = 1;
    e->nnz = nnz;
    const size_t np1 = (size_t)n + 1;
    if (!wide_extents && nnz >= (int64_t)INT32_MAX) {
        e->free_problem();
        return fail(SGA_ERR_INVALID, "nnz >= 2^31 needs 64-bit row extents (sga_set_csr64)");
    }
    HIPCHK(hipMalloc(&e->rowptr64, sizeof(long long) * np1));
    if (wide_extents) {
        HIPCHK(hipMemcpyAsync(e->rowptr64, rowptr, sizeof(long long) * np1, hipMemcpyDefault, e->stream));
    } else {
        HIPCHK(e->scratch[1].reserve(sizeof(int32_t) * np1));
        int32_t *tmp = static_cast<int32_t *>(e->scratch[1].ptr);
        HIPCHK(hipMemcpyAsync(tmp, rowptr, sizeof(int32_t) * np1, hipMemcpyDefault, e->stream));
        HIPCHK(sga::launch_widen_rowptr(tmp, e->rowptr64, (long long)np1, e->stream));
    }
    const size_t nz = (size_t)std::max<int64_t>(nnz, 1);
    // the caller's arrays: borrowed when they are device memory, staged otherwise (freed below)
    const int32_t *ci = colidx;
    const float *vv = val;
    if (nnz > 0 && !is_device_ptr(colidx)) {
        HIPCHK(hipMalloc(&e->colidx, sizeof(int32_t) * nz));
        HIPCHK(hipMemcpyAsync(e->colidx, colidx, sizeof(int32_t) * nz, hipMemcpyHostToDevice, e->stream));
        ci = e->colidx;
    }
    if (nnz > 0 && !is_device_ptr(val)) {
        HIPCHK(hipMalloc(&e->val, sizeof(float) * nz));
        HIPCHK(hipMemcpyAsync(e->val, val, sizeof(float) * nz, hipMemcpyHostToDevice, e->stream));
        vv = e->val;
    }
    HIPCHK(hipMalloc(&e->h, sizeof(float) * (size_t)n));
    HIPCHK(hipMemcpyAsync(e->h, h, sizeof(float) * (size_t)n, hipMemcpyDefault, e->stream));
    HIPCHK(hipMalloc(&e->diag, sizeof(float) * (size_t)n));

    int *d_flags = e->d_flags;
    int flags[sga::CSR_FLAG_COUNT] = {0};
    static_assert(sga::CSR_FLAG_COUNT <= 16, "engine flag words");
    auto read_flags = [&]() -> hipError_t {
        hipError_t he = hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, e->stream);
        return he == hipSuccess ? hipStreamSynchronize(e->stream) : he;
    };
    auto bail = [&](int code, const char *msg) {
        e->free_problem();
        return fail(code, msg);
    };
    hipError_t he = hipMemsetAsync(d_flags, 0, sizeof(flags), e->stream);
    if (he == hipSuccess) he = sga::launch_csr_check_rowptr(e->rowptr64, n, nnz, d_flags, e->stream);
    if (he == hipSuccess) he = read_flags();
    if (he != hipSuccess) return bail(SGA_ERR_DEVICE, hipGetErrorString(he));
    if (flags[sga::CSR_BAD_ROWPTR])
        return bail(SGA_ERR_INVALID, "CSR rowptr is not monotone or does not span [0, nnz]");
    he = sga::launch_csr_scan(e->rowptr64, ci, vv, e->h, n, d_flags, e->stream);
    if (he == hipSuccess) he = read_flags();
    if (he != hipSuccess) return bail(SGA_ERR_DEVICE, hipGetErrorString(he));
    if (flags[sga::CSR_BAD_COLUMN]) return bail(SGA_ERR_INVALID, "CSR column index out of range");
    // symmetric with zero diagonal?  Sorted rows: one binary search per entry; unsorted rows are
    // compared by linear scans while that stays cheap, else treated as asymmetric (exact-energy
    // mode: slower, never wrong)
    const bool sorted = !flags[sga::CSR_UNSORTED];
    e->csr_sorted = sorted;
    const double avg_deg = (double)nnz / n;
    if (sorted || (double)nnz * avg_deg <= 4.0e10) {
        he = sga::launch_csr_symmetry(e->rowptr64, ci, vv, n, sorted, d_flags, e->stream);
        if (he == hipSuccess) he = read_flags();
        if (he != hipSuccess) return bail(SGA_ERR_DEVICE, hipGetErrorString(he));
    } else {
        flags[sga::CSR_ASYMMETRIC] = 1;
    }
    e->consistent_dE = !flags[sga::CSR_ASYMMETRIC] && !flags[sga::CSR_DIAGONAL];
    // integer-valued problem?  then dE takes at most M = max_i(sum_j |J_ij| + |h_i|) even values
    float m;
    std::memcpy(&m, &flags[sga::CSR_ROW_ABS_MAX], sizeof(m));
    // (J integer, h a multiple of 1/2 -- penalty encodings of 0/1 variables: dE takes integer values,
    // tabulated at twice the resolution)
    e->table_m = 0;
    e->table_scale = 1;
    if (!flags[sga::CSR_NOT_INTEGRAL] && m >= 1.0f && m < 16777216.0f) {
        e->table_m = (int)std::min(m, 2048.0f);
    } else if ((flags[sga::CSR_NOT_INTEGRAL] & 5) == 0 && m >= 1.0f && m < 8388608.0f &&
               e->opt[OPT_HALF_TABLE] != 0) {
        e->table_m = (int)std::min(2.0f * m, 2048.0f);
        e->table_scale = 2;
    }
    {
        float mj;
        std::memcpy(&mj, &flags[sga::CSR_ROW_J_ABS_MAX], sizeof(mj));
        e->row_j_abs_max = mj;
        // cached-field sweep over CSR: exact int16 dynamic fields, table arithmetic, every entry its own column
        e->clf_csr_problem = (flags[sga::CSR_NOT_INTEGRAL] & 5) == 0 && e->table_m > 0 && e->consistent_dE && sorted &&
                             mj < 32768.0f && n <= (1 << 30);
    }
    HIPCHK(sga::launch_gather_diag_csr(e->rowptr64, ci, vv, n, e->diag, e->stream));
    std::vector<long long> src(np1);
    HIPCHK(hipMemcpyAsync(src.data(), e->rowptr64, sizeof(long long) * np1, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    {
        // How exact is a row sum?  Integer J with sum |J| < 2^24: fp32 accumulation is exact.  Else,
        // if every J's set bits lie within 53 binary places of each other once the carries of
        // the longest row are counted, the fp64 sum of the (exact) fp32 products is exact in any
        // order.  Only couplings of a wider dynamic range need the canonical summation order.
        long long max_len = 0;
        for (int i = 0; i < n; ++i) max_len = std::max(max_len, src[(size_t)i + 1] - src[(size_t)i]);
        int carry = 0;
        while ((1ll << carry) < std::max<long long>(max_len, 1)) ++carry;
        const int e_hi = flags[sga::CSR_EXP_HI] - 1024, e_lo = 1024 - flags[sga::CSR_EXP_LO];
        const bool any = flags[sga::CSR_EXP_HI] != 0;
        const bool j_int = (flags[sga::CSR_NOT_INTEGRAL] & 1) == 0;
        if (j_int && m < 16777216.0f)
            e->csr_acc = e->table_m > 0 ? sga::CSR_ACC_F32_TABLE : sga::CSR_ACC_F32;
        else if (!any || (e_hi - e_lo + 1 + carry) <= 52)
            e->csr_acc = sga::CSR_ACC_F64;
        else
            e->csr_acc = sga::CSR_ACC_F64_CANON;
        if (e->opt[OPT_FORCE_CSR_ACC] > 0)  // parity tests: the slower forms
            e->csr_acc = std::max(e->csr_acc, std::min(3, (int)e->opt[OPT_FORCE_CSR_ACC]));
    }
    // The layout the kernels read: (column, value) interleaved, one 8-byte load per entry.  Long
    // rows (mean degree >= 192: the problems that run the wide forms) are padded to whole 64-entry
    // slots; CSR_TAIL_PAD zeroed entries behind the array (an empty last row's slot 0; unmasked row loads).
    long long *src_ptr = nullptr;  // the caller's extents, on the device, while rows are packed
    HIPCHK(hipMalloc(&src_ptr, sizeof(long long) * np1));
    he = hipMemcpyAsync(src_ptr, e->rowptr64, sizeof(long long) * np1, hipMemcpyDeviceToDevice, e->stream);
    int rc = he == hipSuccess ? build_layout(e, src, avg_deg >= 192.0 && e->opt[OPT_CSR_SLOTS] != 0)
                              : fail(SGA_ERR_DEVICE, hipGetErrorString(he));
    if (rc == SGA_OK) {
        he = hipMalloc(&e->cv, sizeof(int2) * (size_t)(e->layout_entries + CSR_TAIL_PAD));
        if (he == hipSuccess) he = hipMemsetAsync(e->cv + e->layout_entries, 0, sizeof(int2) * CSR_TAIL_PAD, e->stream);
        if (he == hipSuccess) he = sga::launch_pack_cv_rows(src_ptr, e->rowptr64, ci, vv, nullptr, e->cv, n, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (he != hipSuccess) rc = fail(he == hipErrorOutOfMemory ? SGA_ERR_MEMORY : SGA_ERR_DEVICE, hipGetErrorString(he));
    }
    dev_free(src_ptr);
    dev_free(e->colidx);  // staging copies of host arrays (null when the caller's were device memory)
    dev_free(e->val);
    if (rc != SGA_OK) e->free_problem();
    return rc;
}

int sga_set_csr(sga_engine *e, const int32_t *rowptr, const int32_t *colidx, const float *val,
                const float *h, int n, int64_t nnz) {
    return set_csr_common(e, rowptr, false, colidx, val, h, n, nnz);
}

int sga_set_csr64(sga_engine *e, const int64_t *rowptr, const int32_t *colidx, const float *val,
                  const float *h, int n, int64_t nnz) {
    return set_csr_common(e, rowptr, true, colidx, val, h, n, nnz);
}

int sga_set_tsp(sga_engine *e, const float *dist, int64_t ld, int n_cities, float city_visit,
                float position_fill, const float *h) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (!dist || !h || n_cities < 3 || ld < n_cities) return fail(SGA_ERR_INVALID, "bad TSP problem arguments");
    if (n_cities > 2048) return fail(SGA_ERR_UNSUPPORTED, "more than 2048 cities");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->free_replicas();
    e->free_problem();
    const int n = n_cities;
    const long long N = (long long)n * n;
    const int waves = (n + 255) / 256, npad = 256 * waves;
    if (sga::tsp_lds_bytes(n, npad) > 160 * 1024 - 256)
        return fail(SGA_ERR_UNSUPPORTED, "replica spins do not fit LDS (too many cities)");
    // the distances on the host (4 MB at 1000 cities): classification of the arithmetic
    std::vector<float> dh((size_t)N), hh((size_t)N);
    HIPCHK(hipMemcpy2D(dh.data(), sizeof(float) * (size_t)n, dist, sizeof(float) * (size_t)ld,
                       sizeof(float) * (size_t)n, (size_t)n, hipMemcpyDefault));
    HIPCHK(hipMemcpy(hh.data(), h, sizeof(float) * (size_t)N, hipMemcpyDefault));
    const float a2 = -(city_visit / 2.0f), b2 = -(position_fill / 2.0f);
    bool integral = a2 == std::rint(a2) && b2 == std::rint(b2);
    int e_hi = -10000, e_lo = 10000;
    auto span = [&](float v) {  // binary exponents of the highest and the lowest set bit
        if (v == 0.0f || !std::isfinite(v)) return;
        int ex;
        const float m = std::frexp(std::fabs(v), &ex);  // v = m 2^ex, m in [0.5, 1)
        uint32_t mant = (uint32_t)std::ldexp(m, 24);    // 24-bit integer mantissa
        int low = 0;
        while (!(mant & 1u)) {
            mant >>= 1;
            ++low;
        }
        e_hi = std::max(e_hi, ex - 1);
        e_lo = std::min(e_lo, ex - 24 + low);
    };
    span(a2);
    span(b2);
    double worst_row = 0.0;
    for (int c = 0; c < n; ++c) {
        double row = 0.0;
        for (int q = 0; q < n; ++q) {
            if (q == c) continue;
            const float v1 = dh[(size_t)c * n + q] / 4.0f, v2 = dh[(size_t)q * n + c] / 4.0f;
            if (!std::isfinite(v1)) return fail(SGA_ERR_INVALID, "distance matrix holds a non-finite value");
            integral = integral && v1 == std::rint(v1);
            span(v1);
            row += std::fabs((double)v1) + std::fabs((double)v2);
        }
        worst_row = std::max(worst_row, row);
    }
    for (long long i = 0; i < N && integral; ++i) integral = hh[(size_t)i] == std::rint(hh[(size_t)i]);
    worst_row += (double)(n - 1) * (std::fabs((double)a2) + std::fabs((double)b2));
    int carry = 0;
    while ((1ll << carry) < 4ll * n) ++carry;
    const bool exact32 = integral && worst_row < 16777216.0;
    e->tsp_exact = exact32 || e_hi < e_lo || (e_hi - e_lo + 1 + carry) <= 52;
    // site / n by multiply-shift, verified for every site
    const unsigned int magic = (unsigned int)((0x100000000ull + (unsigned long long)n - 1) / (unsigned long long)n);
    for (long long sidx = 0; sidx < N; ++sidx)
        if ((long long)(((unsigned long long)sidx * magic) >> 32) != sidx / n)
            return fail(SGA_ERR_UNSUPPORTED, "internal: site decomposition does not hold for this size");
    // tables on the device
    const float *src = dist;
    long long ld_src = ld;
    struct Staged {
        float *p = nullptr;
        ~Staged() { dev_free(p); }
    } staged;
    if (!is_device_ptr(dist)) {
        HIPCHK(hipMalloc(&staged.p, sizeof(float) * (size_t)N));
        HIPCHK(hipMemcpyAsync(staged.p, dh.data(), sizeof(float) * (size_t)N, hipMemcpyHostToDevice, e->stream));
        src = staged.p;
        ld_src = n;
    }
    HIPCHK(hipMalloc(&e->nd4, sizeof(float) * (size_t)n * npad));
    HIPCHK(hipMalloc(&e->nd4t, sizeof(float) * (size_t)n * npad));
    HIPCHK(sga::launch_tsp_tables(src, ld_src, n, npad, e->nd4, e->nd4t, e->stream));
    HIPCHK(hipMalloc(&e->h, sizeof(float) * (size_t)N));
    HIPCHK(hipMemcpyAsync(e->h, hh.data(), sizeof(float) * (size_t)N, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->tsp = true;
    e->csr = false;
    e->n = (int)N;
    e->n_models = 1;
    e->nnz = 4ll * (n - 1) * N;
    e->consistent_dE = true;  // symmetric with a zero diagonal by construction
    e->table_m = 0;
    e->tsp_waves = waves;
    e->tsp_passes = 1;
    e->tsp_args = sga::TspArgs{e->nd4, e->nd4t, n, npad, magic, (unsigned int)(4 * npad), a2, b2, exact32 ? 0 : 1};
    return SGA_OK;
}

static int init_replicas_body(sga_engine *e, int R_local, int R_global, int replica0, uint64_t seed,
                              const int8_t *s0);

int sga_init_replicas(sga_engine *e, int R_local, int R_global, int replica0, uint64_t seed,
                      const int8_t *s0) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    // whatever fails in there, the engine is left with NO replicas (R == 0): a later sweep / snapshot /
    // export then reports SGA_ERR_INVALID instead of launching kernels on half-allocated buffers
    const int rc = init_replicas_body(e, R_local, R_global, replica0, seed, s0);
    if (rc != SGA_OK) {
        (void)hipStreamSynchronize(e->stream);
        e->free_replicas();
    }
    return rc;
}

static int init_replicas_body(sga_engine *e, int R_local, int R_global, int replica0, uint64_t seed,
                              const int8_t *s0) {
    if (e->n <= 0) return fail(SGA_ERR_INVALID, "set the couplings before the replicas");
    if (R_local <= 0 || R_global < R_local || replica0 < 0 || replica0 + R_local > R_global)
        return fail(SGA_ERR_INVALID, "bad replica partition");
    if (e->n_models > 1 && R_global % e->n_models != 0)
        return fail(SGA_ERR_INVALID, "R_global must be a multiple of the number of models");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->free_replicas();
    e->R = R_local;
    e->Rg = R_global;
    e->replica0 = replica0;
    e->seed = seed;
    e->sweeps_done = 0;
    e->rounds = 0;
    e->attempted = 0;
    if (e->tsp) {
        e->sstride = (e->n + 15) / 16 * 16;
        // 256 cities per wave and pass.  Four or more waves at one pass: half the waves with two
        // passes each do better (1000 cities, same box: 4 x 1 711 ms, 2 x 2 677 ms, 1 x 4 701 ms per
        // sweep -- fewer barrier participants against a longer row sum); tuning may ask otherwise.
        {
            const int full = e->tsp_args.npad / 256;  // waves at one pass
            int w = (full >= 4 && full % 2 == 0) ? full / 2 : full;
            if (e->tune_waves == full) w = full;
            if (e->tune_waves > 0 && e->tune_waves < full && full % e->tune_waves == 0 &&
                (full / e->tune_waves == 2 || full / e->tune_waves == 4))
                w = e->tune_waves;
            e->tsp_waves = w;
            e->tsp_passes = full / w;
        }
        e->waves = e->tsp_waves;
        e->cpw = 0;
    } else if (!e->csr) {
        int rc = ensure_packed(e);  // geometry depends on the replica count
        if (rc != SGA_OK) return rc;
        e->sstride = (int)e->ld;
    } else {
        e->sstride = (e->n + 15) / 16 * 16;
        const double deg = (double)e->nnz / e->n;
        const int bits_stride = (e->n + 127) / 128 * 128;
        const bool bits_fit = sga::csr_big_fits(bits_stride, 0);
        // (rows the several-updates-per-step form covers are "short": one wave per replica, several replicas per workgroup)
        // -- while the structure is L2 resident or the replicas are few: beyond that the form is bound by the cache
        // fabric (8-byte entries), where the one-wave bit-spin form with packed 4-byte entries stays ahead (assignment
        // 100 x 100, degree 198, 20 MB: 1024 replicas 1.3e9 -> 3.7e9 attempts/s, 4096 replicas 6.3e9 -> 4.0e9)
        const bool rows_medium = csr_rows_medium(e) && csr_updates_per_step(e) >= 4 && e->tune_waves <= 1 &&
                                 (e->layout_entries * 8 <= (6ll << 20) || R_local <= 1024);
        const bool long_rows = deg >= 192.0 && !rows_medium;
        // the bit-spin form that would be used: narrow (several replicas per workgroup, 32-bit
        // extents) on short rows, else one replica per workgroup with its row dealt to waves
        const int rpb_bits = (e->rowptr && !long_rows && e->tune_waves <= 1)
                                 ? sga::csr_bits_waves_per_block(bits_stride, e->table_m) : 0;
        const bool narrow_bits = rpb_bits >= 2;
        // Spins as bits in LDS: beyond the int8 capacity or the 32-bit extents
        // (option "force_csr_bits": parity tests run the small cases through the same forms) ...
        bool bits = sga::csr_waves_per_block(e->sstride, 0) < 1 || !e->rowptr ||
                    e->opt[OPT_FORCE_CSR_BITS] != 0;
        // ... or when the int8 spins fit, but not for all replicas at once: workgroups beyond the
        // LDS-resident set run as a second, mostly empty round (C4: 50 KB per replica = 3 per CU
        // = 768 of 1024 replicas resident, 4.7e8 attempts/s; as bits all are resident: 6.8e8)
        if (!bits && bits_fit && e->opt[OPT_CSR_BITS] != 0) {
            const bool wide_i8 = e->tune_waves > 1 || (e->tune_waves == 0 && long_rows && R_local <= 1024);
            const int rpb = wide_i8 ? 1 : std::max(1, sga::csr_waves_per_block(e->sstride, e->table_m));
            const long long budget = 160 * 1024 - 256;
            const long long wg_i8 = (long long)sga::csr_lds_bytes(e->sstride, e->table_m, false) * rpb;
            const long long one_bits = (long long)sga::csr_lds_bytes(bits_stride, e->table_m, true);
            const long long res_i8 = (long long)e->cus * rpb * std::min<long long>(8, budget / wg_i8);
            const long long res_bits =
                narrow_bits ? (long long)e->cus * rpb_bits * std::min<long long>(8, budget / (one_bits * rpb_bits))
                            : (long long)e->cus * std::min<long long>(16, budget / one_bits);
            // Against the one-replica-per-workgroup bit form the barrier-free narrow int8 form with
            // 3-4 replicas per workgroup stays ahead (degree 32, 4096 replicas, n = 40k: 2.35e9 vs
            // 1.77e9 attempts/s with a quarter of the replicas resident; n = 60k, 2 per workgroup:
            // 1.17e9 vs 1.74e9); against the narrow bit form residency decides.
            if (R_local > res_i8 && res_bits > res_i8 && (narrow_bits || wide_i8 || rpb <= 2)) bits = true;
            // (the several-updates-per-step form, sweep_csr_rows.hip, runs on either: 3-D lattice, n = 10 648, 4096
            //  replicas: 3.55e10 attempts/s on int8 spins with 3072 replicas resident, 5.5e10 on bits with all)
        }
        // Long rows with MANY replicas (C5 at 100 cities: degree 396, 2048 replicas): one wave per
        // replica either way, but the slot-addressed bit form (one replica per workgroup, scalar
        // addressing, no per-lane bounds tests) beats the entry-addressed int8 form with four replicas
        // per workgroup: 10.6 vs 11.9 ms per sweep.
        const bool many_long = long_rows && R_local > 1024 && e->tune_waves == 0 && bits_fit && e->slotted &&
                               e->opt[OPT_CSR_BITS] != 0;
        if (many_long) bits = true;
        e->big = bits;
        e->big_form = !bits ? 0 : (narrow_bits ? 2 : 1);
        if (bits) {
            e->sstride = bits_stride;
            if (!bits_fit)
                return fail(SGA_ERR_UNSUPPORTED, "CSR problem too large for the LDS-resident spins");
            if (!sga::csr_big_fits(e->sstride, e->table_m)) e->table_m = 0;
            if (e->big_form == 2) {
                e->waves = 1;
            } else {
                // one workgroup per replica: deal a long row to as many waves as the 8 entries per
                // lane requested ahead need to cover it (profiles/r01_experiments.md: 500 cities,
                // degree 1996: 4 waves; 1000 cities, 3996: 8)
                const int wpr = e->tune_waves > 0 ? e->tune_waves : (many_long ? 1 : (int)std::ceil(deg / 512.0));
                e->waves = std::max(1, std::min(wpr, 8));
            }
        } else {
            if (sga::csr_waves_per_block(e->sstride, e->table_m) < 1)
                e->table_m = 0;  // no room for the probability table: general path
            // Long rows AND too few replicas to give every SIMD a wave: deal each row to two waves
            // (one replica per workgroup).  The kernel is issue bound, so with >= 2048 replicas the
            // extra waves only repeat the per-update work (measured: C4, R = 1024: 1 / 2 / 4 / 8
            // waves -> 2.98 / 3.69 / 3.67 / 3.34 e8 attempts/s; C5, R = 2048: 1.55 vs 1.19 e9).
            const int wpr = e->tune_waves > 0 ? e->tune_waves : ((long_rows && R_local <= 1024) ? 2 : 1);
            e->waves = std::min(wpr, 8);
        }
        // the wide builds exist for 1, 2, 4 and 8 waves per replica (slot arithmetic on constants; the
        // canonical summation order of real-valued rows is defined on that grid)
        {
            int p2 = 1;
            while (p2 < e->waves) p2 *= 2;
            e->waves = std::min(p2, 8);
        }
        e->cpw = 0;
        // one replica per workgroup (row dealt to its waves): rows are addressed by 64-entry slots
        if (e->waves > 1 || e->big_form == 1) {
            int rc = ensure_slotted(e);
            if (rc != SGA_OK) return rc;
        }
        e->csr_storage_latched = e->csr_storage;
        if (e->big_form == 1 && e->csr_storage != SGA_CSR_STORAGE_F32) {
            int rc = ensure_packed_entries(e);
            if (rc != SGA_OK) return rc;
        }
        if (e->csr_storage == SGA_CSR_STORAGE_PACKED && !(e->big_form == 1 && e->cvp))
            return fail(SGA_ERR_UNSUPPORTED, "packed CSR entries need integer couplings with |J| <= 127, n < 2^24 "
                                             "and the one-replica-per-workgroup bit-spin form");
    }
    const size_t sb = (size_t)R_local * e->sstride;
    HIPCHK(hipMalloc(&e->spins, sb));
    HIPCHK(hipMalloc(&e->best_spins, sb));
    HIPCHK(hipMalloc(&e->energy, sizeof(double) * R_local));
    HIPCHK(hipMalloc(&e->best_energy, sizeof(double) * R_local));
    HIPCHK(hipMalloc(&e->rep_temp, sizeof(double) * R_local));
    HIPCHK(hipMalloc(&e->n_acc, sizeof(unsigned long long) * R_local));
    HIPCHK(hipMemsetAsync(e->n_acc, 0, sizeof(unsigned long long) * R_local, e->stream));
    {
        std::vector<double> ones((size_t)R_local, 1.0);
        HIPCHK(hipMemcpyAsync(e->rep_temp, ones.data(), sizeof(double) * R_local,
                              hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (s0) {
        DevIn<int8_t> in;
        int rc = in.init(e->scratch[1], s0, (size_t)R_local * e->n, e->stream);
        if (rc != SGA_OK) return rc;
        HIPCHK(sga::launch_pad_spins(in.ptr, e->n, e->spins, e->sstride, R_local, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    } else {
        HIPCHK(sga::launch_init_spins(e->spins, e->n, e->sstride, R_local, (uint32_t)seed,
                                      (uint32_t)(seed >> 32), (uint32_t)replica0, e->stream));
    }
    int rc = recompute_energy_range(e, 0, R_local);
    if (rc != SGA_OK) return rc;
    HIPCHK(sga::launch_copy_best(e->energy, e->spins, e->best_energy, e->best_spins, e->sstride,
                                 R_local, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_set_temperatures(sga_engine *e, const double *T) {
    if (!e || !T) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->rep_temp, T, sizeof(double) * e->R, hipMemcpyDefault, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_set_ladder(sga_engine *e, const double *slot_temps, int n_ladders) {
    if (!e || !slot_temps) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    if (n_ladders <= 0 || e->Rg % n_ladders != 0)
        return fail(SGA_ERR_INVALID, "R_global must be a multiple of n_ladders");
    HIPCHK(hipSetDevice(e->device));
    dev_free(e->slot_temps);
    dev_free(e->slot_to_rep);
    dev_free(e->ex_attempts);
    dev_free(e->ex_accepts);
    const size_t Rg = (size_t)e->Rg;
    HIPCHK(hipMalloc(&e->slot_temps, sizeof(double) * Rg));
    HIPCHK(hipMalloc(&e->slot_to_rep, sizeof(int32_t) * Rg));
    HIPCHK(hipMalloc(&e->ex_attempts, sizeof(long long) * Rg));
    HIPCHK(hipMalloc(&e->ex_accepts, sizeof(long long) * Rg));
    HIPCHK(hipMemcpyAsync(e->slot_temps, slot_temps, sizeof(double) * Rg, hipMemcpyDefault,
                          e->stream));
    std::vector<int32_t> ident(Rg);
    for (size_t i = 0; i < Rg; ++i) ident[i] = (int32_t)i;
    HIPCHK(hipMemcpyAsync(e->slot_to_rep, ident.data(), sizeof(int32_t) * Rg, hipMemcpyHostToDevice,
                          e->stream));
    HIPCHK(hipMemsetAsync(e->ex_attempts, 0, sizeof(long long) * Rg, e->stream));
    HIPCHK(hipMemsetAsync(e->ex_accepts, 0, sizeof(long long) * Rg, e->stream));
    // slot i initially holds replica i: local temperatures are the matching slice
    HIPCHK(hipMemcpyAsync(e->rep_temp, e->slot_temps + e->replica0, sizeof(double) * e->R,
                          hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->n_ladders = n_ladders;
    e->rounds = 0;
    return SGA_OK;
}

int sga_sweep(sga_engine *e, int n_sweeps, int site_mode, int arith, const double *sched,
              int64_t sched_sweep_stride, int64_t sched_replica_stride,
              const int32_t *replay_site, const float *replay_u, double *energy_trace,
              uint8_t *accept_trace, double *dE_trace) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas (call sga_init_replicas)");
    if (n_sweeps < 0) return fail(SGA_ERR_INVALID, "n_sweeps < 0");
    if (site_mode < SGA_SITE_RANDOM || site_mode > SGA_SITE_REPLAY)
        return fail(SGA_ERR_INVALID, "bad site_mode");
    if (arith != SGA_ARITH_F64 && arith != SGA_ARITH_F32) return fail(SGA_ERR_INVALID, "bad arith");
    if (e->rule != SGA_RULE_METROPOLIS && arith != SGA_ARITH_F64)
        return fail(SGA_ERR_INVALID, "Glauber / heat-bath rules need SGA_ARITH_F64");
    if (site_mode == SGA_SITE_REPLAY && (!replay_site || !replay_u))
        return fail(SGA_ERR_INVALID, "SITE_REPLAY needs replay_site and replay_u");
    if (n_sweeps == 0) return SGA_OK;
    HIPCHK(hipSetDevice(e->device));
    int rc = ensure_packed(e);
    if (rc != SGA_OK) return rc;
    if (!e->csr && !e->tsp && e->sstride != (int)e->ld)
        return fail(SGA_ERR_INVALID, "tuning changed after sga_init_replicas; re-initialise");
    // The cached-field modes pick their kernel form by the acceptance counters, looked at when a call starts: a long
    // production call is walked in pieces of 16 sweeps so that the form follows the run (the chain does not depend on
    // how a run is cut into calls).
    constexpr int PIECE = 16;
    if (n_sweeps > PIECE && e->field_cache != SGA_FIELD_CACHE_OFF && e->rule != SGA_RULE_WOLFF && site_mode == SGA_SITE_RANDOM &&
        !replay_site && !replay_u && !accept_trace && !dE_trace) {
        const char *why = nullptr;
        if (clf_possible(e, &why)) {
            for (int k = 0; k < n_sweeps; k += PIECE) {
                rc = sga_sweep(e, std::min(PIECE, n_sweeps - k), site_mode, arith, sched ? sched + (long long)k * sched_sweep_stride : nullptr,
                               sched_sweep_stride, sched_replica_stride, nullptr, nullptr,
                               energy_trace ? energy_trace + (size_t)k * (size_t)e->R : nullptr, nullptr, nullptr);
                if (rc != SGA_OK) return rc;
            }
            return SGA_OK;
        }
    }

    const int n = e->n, R = e->R;
    const long long per = (long long)n_sweeps * n;
    hipStream_t st = e->stream;

    // schedule table: find its extent from the strides
    DevIn<double> d_sched;
    if (sched) {
        if (sched_sweep_stride < 0 || sched_replica_stride < 0)
            return fail(SGA_ERR_INVALID, "negative schedule stride");
        const size_t extent =
            (size_t)((n_sweeps - 1) * sched_sweep_stride + (R - 1) * sched_replica_stride + 1);
        rc = d_sched.init(e->scratch[0], sched, extent, st);
        if (rc != SGA_OK) return rc;
    }
    DevIn<int32_t> d_site;
    DevIn<float> d_u;
    if (site_mode == SGA_SITE_REPLAY) {
        rc = d_site.init(e->scratch[1], replay_site, (size_t)R * per, st);
        if (rc != SGA_OK) return rc;
    }
    if (site_mode != SGA_SITE_RANDOM && replay_u) {
        rc = d_u.init(e->scratch[2], replay_u, (size_t)R * per, st);
        if (rc != SGA_OK) return rc;
    }
    DevOut<double> d_etrace, d_dE;
    DevOut<uint8_t> d_acc;
    rc = d_etrace.init(e->scratch[3], energy_trace, (size_t)n_sweeps * R, st);
    if (rc != SGA_OK) return rc;
    rc = d_acc.init(e->scratch[4], accept_trace, (size_t)R * per, st);
    if (rc != SGA_OK) return rc;
    rc = d_dE.init(e->scratch[5], dE_trace, (size_t)R * per, st);
    if (rc != SGA_OK) return rc;

    // sweeps per launch: aim for ~50 ms of estimated work per launch
    int spl = e->tune_spl;
    if (spl <= 0) {
        const double row_bytes = e->tsp ? 8.0 * e->tsp_args.npad
                                 : e->csr ? 264.0 : (double)e->ldj * (e->want_i8 ? 1 : 4);
        const double per_update = std::max(row_bytes * R / 4.0e12, 1.0e-6);
        const double per_sweep = per_update * n;
        spl = (int)std::min<double>(n_sweeps, std::max(1.0, std::floor(0.05 / per_sweep)));
    }
    spl = std::max(1, std::min(spl, n_sweeps));
    // Asymmetric J or a non-zero diagonal: the rule's dE (row i only, as the reference computes
    // it) is not the energy change, so E += dE would drift from compute_energy().  Then every
    // sweep is its own launch, followed by a from-scratch energy evaluation and the best update
    // (exactly the reference's sequence, core/spin_dynamics.py:87, gpu_annealer.py:151-153).
    // (the Wolff rule reports compute_energy() after every sweep as well, spin_dynamics.py:87)
    const bool wolff = e->rule == SGA_RULE_WOLFF;
    if (wolff) {
        if (e->tsp) return fail(SGA_ERR_UNSUPPORTED, "the Wolff rule is not implemented for sga_set_tsp problems");
        if (sga::wolff_lds_bytes(n) > 160 * 1024 - 256)
            return fail(SGA_ERR_UNSUPPORTED, "the Wolff rule keeps spins, cluster and queue in LDS: n <= ~31 000");
        if (site_mode == SGA_SITE_SEQUENTIAL && !replay_u)
            return fail(SGA_ERR_INVALID, "sequential Wolff sweeps need replay_u (unused values are fine)");
        // the cluster growth treats every stored entry as one bond: duplicate columns of a row (which the
        // other rules add up) would be drawn twice and could overrun the cluster queue
        if (e->csr && !e->csr_sorted)
            return fail(SGA_ERR_UNSUPPORTED, "the Wolff rule over CSR couplings needs rows strictly sorted by "
                                             "column (no duplicate entries)");
    }
    const bool exact_mode = !e->consistent_dE || wolff;
    if (exact_mode) spl = 1;
    // cached local fields (sga_set_field_cache): a row is read only when a proposal is accepted
    bool clf = false;
    if (e->field_cache != SGA_FIELD_CACHE_OFF && !wolff) {
        const char *why = nullptr;
        clf = clf_possible(e, &why);
        if (!clf && e->field_cache == SGA_FIELD_CACHE_ON) return fail(SGA_ERR_UNSUPPORTED, why);
    }
    // AUTO routes every replica by ITS OWN acceptance: the cached-field kernel costs a replica ~1.1 - 1.7 us of its
    // serial chain per ACCEPTED proposal and next to nothing per rejected one, the row-per-proposal kernels cost
    // every replica the same per update whatever happens (~0.4 us on bit-planes, ~1.3 us on int8 rows, ~5 us on
    // fp32 rows at n = 10^4) -- so a replica belongs on the row kernels only while its acceptance exceeds the ratio
    // of the two (profiles/r04_experiments.md 2), and a ladder with a hot end runs as TWO concurrent launches (two
    // streams) over disjoint replica lists.  The chain of a replica does not depend on the kernel that walks it.
    // The run starts on the kernel that loses least if the guess is wrong (below); the per-replica counters are read back every
    // 4 ... 16 sweeps.  Option "replica_routing" = 0: one launch, decided by the hottest replica (round 3).
    // Both cached-field modes (ON and AUTO) look at the per-replica acceptance now and then.  A launch of the cached-field
    // kernel ends with its hottest replica's serial chain; once most replicas accept next to nothing -- their workgroups
    // are gone early and the chip idles behind that one chain -- EVERY replica gets eight waves (option "clf_tail_waves"):
    // the workgroups then run as two rounds, which costs where the replicas are busy (sweeps 5-25 of the 10 000-spin ladder:
    // 0.42 against 0.27 ms per sweep) and pays in the tail (after 100 sweeps 0.098 against 0.106;
    // profiles/r04_experiments.md 9).  Giving only the hottest replicas eight waves in a launch of their own did not:
    // beside the four-wave workgroups of the others their rounds took 1.6 us instead of 0.9.
    int n_clf = clf ? R : 0;  // replicas on the cached-field kernel in this call
    const bool is_auto = e->field_cache == SGA_FIELD_CACHE_AUTO;
    const int clf_waves_std = (clf && !e->csr) ? sga::sweep_clf_waves(e->ldj, e->want_i8, e->R, e->cus, (int)e->opt[OPT_CLF_WAVES]) : 0;
    const bool tail_opt = clf && !e->csr && e->opt[OPT_CLF_TAIL_WAVES] != 0 && e->opt[OPT_CLF_WAVES] == 0 &&
                         clf_waves_std < 8 && e->ldj >= 6 * (e->want_i8 ? 1024 : 256) && e->R >= 16;
    // Option "clf_batched" = 2 (default): the form that commits several accepts per round (sweep_clfb_impl.h) while the
    // hottest replica accepts more than ~1 % of its proposals -- 16 % ahead on the first sweeps from random spins, 10 %
    // at sweeps 5-25 of the 10 000-spin ladder -- and one accept per round below (7 % ahead after 100 sweeps).
    const bool adaptive = clf && !e->csr && e->opt[OPT_CLF_BATCHED] == 2;
    auto routing_theta = [&]() -> double {  // break-even acceptance of one replica: t_update (row kernel) / t_accept (cached)
        const double kn = (double)n / 1000.0;
        const double t_upd = e->csr ? 0.20 + 0.0008 * (double)e->nnz / (double)n  // (C4: 0.68, C2b as CSR: 0.36)
                             : e->use_t2 ? 0.29 + 0.009 * kn : (e->want_i8 ? 0.27 + 0.031 * kn : 0.30 + 0.12 * kn);
        return t_upd / 1.5;
    };
    if (clf && (is_auto || tail_opt || adaptive)) {
        if (is_auto && e->auto_unavailable) {
            n_clf = 0;
        } else {
            if ((long long)e->auto_mark_acc.size() != e->R) {
                e->auto_mark_acc.assign((size_t)e->R, 0ull);
                // ON: 0 = cached-field kernel.  AUTO: nothing is known yet -- the run starts on the kernel that loses least
                // if the guess is wrong: the cached-field kernel where a replica would have to accept more than ~30 % of
                // its proposals for the row kernels to win (int8 and fp32 rows at n = 10^4: the first four sweeps of the
                // bench ladder 55 / 218 ms on the row kernels against 8 / 30 ms cached, and 36 against 56 / 208 ms on a
                // ladder that stays hot), the row-per-proposal kernel otherwise (bit-planes, small n).
                const bool start_cached = !is_auto || (!e->csr && 0.8 * routing_theta() >= 0.3);
                e->route.assign((size_t)e->R, start_cached ? 0 : 1);
                e->n_route_clf = start_cached ? e->R : 0;
                e->clf_wide = false;
                e->clf_hot = true;   // (nothing known yet: a run starts hot)
                e->auto_mark_attempted = 0;
                e->auto_interval = 4;
            }
            const long long since = e->attempted - e->auto_mark_attempted;
            if (since >= (long long)e->auto_interval * n || since < 0) {
                std::vector<unsigned long long> now((size_t)e->R);
                HIPCHK(hipMemcpyAsync(now.data(), e->n_acc, sizeof(unsigned long long) * e->R, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (since > 0) {
                    bool back = false;
                    const bool was_wide = e->clf_wide;
                    if (is_auto) {
                        // (enter, leave): acceptance below which a replica is taken onto the cached-field kernel, above
                        // which it is given back (hysteresis)
                        // Break-even acceptance of ONE replica = (what an update costs its chain on the row kernel) / (what
                        // an accept costs it on the cached-field kernel).  Both kernels are paced by a replica's serial
                        // chain, not by the chip, whenever only part of the replicas is hot: ~1.5 us per accept (1.15 alone
                        // on its CU ... 1.7 with busy neighbours), and per update 0.38 us on bit-planes / 0.58 us on int8
                        // rows at n = 10^4, ~0.3 us on short rows (profiles/r04_routing.py; fp32 rows: estimate).
                        const double theta = routing_theta();
                        const double enter = 0.8 * theta, leave = 1.2 * theta;
                        if (e->opt[OPT_REPLICA_ROUTING] != 0 && !e->csr) {  // (the CSR row kernels take no replica lists)
                            for (int r2 = 0; r2 < e->R; ++r2) {
                                const double acc = (double)(now[(size_t)r2] - e->auto_mark_acc[(size_t)r2]) / (double)since;
                                int &rt = e->route[(size_t)r2];
                                if (rt == 0 && acc > leave) rt = 1;
                                else if (rt == 1 && acc < enter) rt = 0, back = true;
                            }
                        } else {
                            unsigned long long top = 0;
                            for (int r2 = 0; r2 < e->R; ++r2) top = std::max(top, now[(size_t)r2] - e->auto_mark_acc[(size_t)r2]);
                            const double hottest = (double)top / (double)since;
                            const bool was = e->n_route_clf > 0;
                            const bool use = was ? hottest < leave : hottest < enter;
                            back = use && !was;
                            e->route.assign((size_t)e->R, use ? 0 : 1);
                        }
                    }
                    int cnt = 0;
                    for (int v : e->route) cnt += v == 0;
                    e->n_route_clf = cnt;
                    e->clf_wide = false;
                    if (adaptive && cnt > 0) {  // (hysteresis: in above 1.5 % of the hottest replica's proposals, out below 1 %)
                        unsigned long long top = 0;
                        for (int r2 = 0; r2 < e->R; ++r2)
                            if (e->route[(size_t)r2] == 0) top = std::max(top, now[(size_t)r2] - e->auto_mark_acc[(size_t)r2]);
                        const double hottest = (double)top / (double)since;
                        e->clf_hot = hottest > (e->clf_hot ? 0.010 : 0.015);
                    }
                    if (tail_opt && cnt > 0 && !(adaptive && e->clf_hot)) {
                        // accepts per sweep of the replicas on the cached-field kernel: the hottest one's, and the mean
                        const double per_sweep = (double)n / (double)since;  // counter difference -> accepts per sweep
                        double amax = 0.0, asum = 0.0;
                        for (int r2 = 0; r2 < e->R; ++r2) {
                            if (e->route[(size_t)r2] != 0) continue;
                            const double ar = (double)(now[(size_t)r2] - e->auto_mark_acc[(size_t)r2]) * per_sweep;
                            amax = std::max(amax, ar);
                            asum += ar;
                        }
                        // (mean / hottest ~ the share of the launch during which the chip is busy: measured ahead at 0.21,
                        //  behind at 0.38 -- in below 0.28, out above 0.36; a chain of two dozen accepts per sweep is the
                        //  least that matters against the windows of a sweep)
                        const double ratio = asum / (double)cnt / std::max(amax, 1.0);
                        e->clf_wide = amax >= 24.0 && ratio < (was_wide ? 0.36 : 0.28);
                    }
                    if (back) e->fields_valid = false;  // somebody returns from the row kernels: fields are seeded anew
                    e->auto_interval = std::min(16, e->auto_interval * 2);
                    e->route_dirty = true;
                }
                e->auto_mark_acc.swap(now);
                e->auto_mark_attempted = e->attempted;
            }
            n_clf = e->n_route_clf;
        }
        clf = n_clf > 0;
    }
    if (clf) {
        rc = ensure_fields(e);
        if (rc == SGA_ERR_MEMORY && e->field_cache == SGA_FIELD_CACHE_AUTO) {
            // AUTO promises a faster form where it is available, not a failure where the row-per-proposal kernels
            // (which need none of this memory) would have run: the cache is "not available" for these replicas
            e->auto_unavailable = true;
            dev_free(e->fields);
            e->fields_valid = false;
            clf = false;
            n_clf = 0;
        } else if (rc != SGA_OK) {
            return rc;
        }
    }
    const int n_rows = clf ? R - n_clf : 0;  // replicas of this call on the row-per-proposal kernels beside the cached ones
    const bool mixed = clf && n_rows > 0;
    if (mixed) {
        // replica lists on the device ([0, n_clf): cached-field kernel, [R, R + R - n_clf): row kernels), the second
        // stream and the two events that fork / join it
        if (!e->d_rep_lists) HIPCHK(hipMalloc(&e->d_rep_lists, sizeof(int) * 2 * (size_t)R));
        if (e->route_dirty) {
            std::vector<int> lists(2 * (size_t)R, 0);
            int ia = 0, ib = 0;
            for (int r2 = 0; r2 < R; ++r2) {
                if (e->route[(size_t)r2] == 0) lists[(size_t)ia++] = r2;
                else lists[(size_t)R + (size_t)ib++] = r2;
            }
            HIPCHK(hipMemcpyAsync(e->d_rep_lists, lists.data(), sizeof(int) * lists.size(), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
            e->route_dirty = false;
        }
        if (!e->aux_stream) HIPCHK(hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking));
        if (!e->fork_ev) HIPCHK(hipEventCreateWithFlags(&e->fork_ev, hipEventDisableTiming));
        if (!e->join_ev) HIPCHK(hipEventCreateWithFlags(&e->join_ev, hipEventDisableTiming));
    }
    if (clf && n_rows == 0) {
        // no row streaming to bound the launch by: many sweeps per launch (a sweep is 0.1 ... 10 ms here:
        // at most 256 of them, so that a launch stays well under a few seconds)
        if (e->tune_spl <= 0) spl = std::min(n_sweeps, 256);
    } else if (!clf) {
        e->fields_valid = false;  // the row-per-proposal kernels move the spins only
    }
    e->last_mixed[0] = '\0';

    for (int k0 = 0; k0 < n_sweeps; k0 += spl) {
        const int ks = std::min(spl, n_sweeps - k0);
        sga::SweepArgs a{};
        a.J = e->J_packed;
        a.rowptr = e->rowptr;
        a.rowptr64 = e->rowptr64;
        a.rowinfo = e->rowinfo;
        a.cvp = (e->big_form == 1 && e->csr_storage_latched != SGA_CSR_STORAGE_F32) ? e->cvp : nullptr;
        a.csr_acc = e->csr_acc;  // (the table form needs its table: set below once table_m is final)
        {   // head slots per wave that the longest row needs (the wide bit forms are built per count)
            const long long slots = (e->max_row_len + 63) / 64;
            const long long need = (slots + std::max(e->waves, 1) - 1) / std::max(e->waves, 1);
            a.csr_head = (int)std::min<long long>(std::max<long long>(need, 1), 10);
        }
        a.big = e->big_form;
        // (a slotted layout's row extents include the padding to whole 64-entry slots)
        a.csr_row_cap = (e->csr && e->max_row_len <= 256)
                            ? (int)std::max<long long>(e->slotted ? (e->max_row_len + 63) / 64 * 64 : e->max_row_len, 1) : 0;
        a.csr_pair_ahead = csr_updates_per_step(e);
        // (option "look_ahead" = 0: A/B switch and the parity tests' cross-check)
        a.look_ahead = e->opt[OPT_LOOK_AHEAD] != 0 ? 1 : 0;
        a.force_general = e->opt[OPT_FORCE_GENERAL] != 0 ? 1 : 0;
        a.tsp_parallel = (int)e->opt[OPT_TSP_PARALLEL];
        a.cv = e->cv;
        a.h = e->h;
        a.diag = e->diag;
        a.spins = e->spins;
        a.energy = e->energy;
        a.best_energy = e->best_energy;
        a.best_spins = e->best_spins;
        a.n_accepted = e->n_acc;
        a.rep_temp = e->rep_temp;
        a.sched = d_sched.ptr ? d_sched.ptr + (long long)k0 * sched_sweep_stride : nullptr;
        a.sched_ss = sched_sweep_stride;
        a.sched_rs = sched_replica_stride;
        const long long off = (long long)k0 * n;
        a.replay_site = d_site.ptr ? d_site.ptr + off : nullptr;
        a.replay_u = d_u.ptr ? d_u.ptr + off : nullptr;
        a.replay_stride = per;
        a.energy_trace = (d_etrace.ptr && !exact_mode) ? d_etrace.ptr + (long long)k0 * R : nullptr;
        a.accept_trace = d_acc.ptr ? d_acc.ptr + off : nullptr;
        a.dE_trace = d_dE.ptr ? d_dE.ptr + off : nullptr;
        a.ld = e->ld;
        a.ldj = e->ldj;
        a.n = n;
        a.sstride = e->sstride;
        a.R = R;
        a.n_sweeps = ks;
        a.site_mode = site_mode;
        a.arith = arith;
        a.rule = e->rule;
        a.table_m = exact_mode ? 0 : e->table_m;
        a.table_scale = e->csr ? e->table_scale : 1;
        if (a.csr_acc == sga::CSR_ACC_F32_TABLE && a.table_m == 0) a.csr_acc = sga::CSR_ACC_F32;
        a.no_best = exact_mode ? 1 : 0;
        a.reps_per_model = e->n_models > 1 ? e->Rg / e->n_models : 0;
        a.model_stride_j = (long long)e->n * e->ldj;
        a.seed_lo = (uint32_t)e->seed;
        a.seed_hi = (uint32_t)(e->seed >> 32);
        a.sweep0 = e->sweeps_done + (uint32_t)k0;
        a.replica0 = (uint32_t)e->replica0;

        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (e->timing) {
            HIPCHK(hipEventCreate(&ev0));
            const hipError_t ce = hipEventCreate(&ev1);
            if (ce != hipSuccess) {
                (void)hipEventDestroy(ev0);
                HIPCHK(ce);
            }
            HIPCHK(hipEventRecord(ev0, st));
        }
        const bool lean = !a.force_general && site_mode == SGA_SITE_RANDOM && arith == SGA_ARITH_F64 &&
                          e->rule == SGA_RULE_METROPOLIS && !a.accept_trace && !a.dE_trace;
        // the row-per-proposal kernel of a dense problem (all replicas, or the list in aa.rep_list)
        auto launch_dense_rows = [&](sga::SweepArgs aa, hipStream_t s2) -> hipError_t {
            if (e->use_t2 && lean) {  // production sweeps read the two bit-planes
                aa.J = e->J_bits;
                aa.J_aux = e->J_packed;
                aa.plane_row_bytes = t2_row_bits(e->n) / 8;
                aa.plane_bytes = (long long)e->n * aa.plane_row_bytes;
                aa.diag = e->row_nnz;
                return sga::launch_sweep_dense_t2(aa, e->waves_t2, e->cpw_t2 > sga::T2_MAX_CPW ? 0 : e->cpw_t2, s2);
            }
            return sga::launch_sweep_dense(aa, e->want_i8, e->acc64 ? (e->acc_canon ? 2 : 1) : 0, e->waves,
                                           e->cpw > sga::MAX_CPW ? 0 : e->cpw, s2);
        };
        hipError_t le;
        bool clf_now = clf;
        if (clf && e->csr) {
            // sparse couplings: D = J s as int16 in LDS, the row's entries read on accept (sweep_clf_csr.hip);
            // production arguments only -- traced / replayed / sequential sweeps take the row-per-proposal kernels
            // (the same chain) and the fields are seeded anew afterwards
            sga::SweepArgs ac = a;
            ac.fields = e->fields;
            ac.ldf = e->ldf;
            ac.clf_hq = e->hq;
            ac.clf_row_max = (int)std::min<long long>(e->slotted ? (e->max_row_len + 63) / 64 * 64 : e->max_row_len, 1 << 20);
            const int cw = ac.clf_row_max > 256 || e->n > 20000 ? 8 : 4;
            if (sga::sweep_clf_csr_applies(ac, cw)) {
                le = sga::launch_sweep_clf_csr(ac, cw, st);
            } else {
                clf_now = false;
                e->fields_valid = false;
            }
        }
        if (clf_now && e->csr) {
            // (launched above)
        } else if (clf_now) {
            sga::SweepArgs ac = a;
            if (e->clf_scale == 2)  // half-integer fields: dE = q for q <= 2 M, tabulated at twice the resolution
                ac.table_m = (int)std::min(2.0 * (double)e->row_abs_max, 2048.0);
            ac.fields = e->fields;
            ac.ldf = e->ldf;
            ac.field_bits = e->clf_bits;
            ac.field_scale = e->clf_scale;
            ac.clf_jmax = e->j_abs_max;
            const int cw = (tail_opt && e->clf_wide)
                               ? 8
                               : sga::sweep_clf_waves(e->ldj, e->want_i8, mixed ? n_clf : e->R, e->cus, (int)e->opt[OPT_CLF_WAVES]);
            // option "clf_batched": production arguments commit several accepts per round -- every decision of a
            // super-window guessed at once, the guess checked against the few couplings between the accepting sites
            // (sweep_clfb_impl.h); the same chain.  Ahead while the hottest replica accepts more than ~1 % (first sweeps
            // from random spins 2.19 -> 1.72 ms, sweeps 5-25 0.272 -> 0.245), behind after 100 sweeps (0.105 -> 0.112):
            // 2 = by the hottest replica's acceptance (default), 1 = always, 0 = never (profiles/r04_experiments.md 9)
            ac.clf_batched = (e->opt[OPT_CLF_BATCHED] == 1 || (adaptive && e->clf_hot)) ? 1 : 0;
            const bool batched = sga::sweep_clfb_applies(ac, e->want_i8);
            auto launch_cached = [&](const sga::SweepArgs &aa, hipStream_t s2) -> hipError_t {
                return batched ? sga::launch_sweep_clfb(aa, e->want_i8, cw, s2) : sga::launch_sweep_clf(aa, e->want_i8, cw, s2);
            };
            if (!mixed) {
                le = launch_cached(ac, st);
            } else {
                // two launches over disjoint replica lists, side by side: the cached-field kernel on the engine's
                // stream, the row-per-proposal kernel on the second one, forked and joined by events
                ac.rep_list = e->d_rep_lists;
                ac.rep_count = n_clf;
                a.rep_list = e->d_rep_lists + R;
                a.rep_count = R - n_clf;
                le = hipEventRecord(e->fork_ev, st);
                if (le == hipSuccess) le = hipStreamWaitEvent(e->aux_stream, e->fork_ev, 0);
                if (le == hipSuccess) le = launch_cached(ac, st);
                char first[200];
                std::snprintf(first, sizeof(first), "%s", sga::last_sweep_kernel());
                if (le == hipSuccess) le = launch_dense_rows(a, e->aux_stream);
                if (le == hipSuccess) le = hipEventRecord(e->join_ev, e->aux_stream);
                if (le == hipSuccess) le = hipStreamWaitEvent(st, e->join_ev, 0);
                if (le != hipSuccess) (void)hipStreamSynchronize(e->aux_stream);
                std::snprintf(e->last_mixed, sizeof(e->last_mixed), "mixed launch: %d replica(s) on %s || %d on %s", n_clf,
                              first, R - n_clf, sga::last_sweep_kernel());
                sga::note_sweep_kernel("%s", e->last_mixed);
            }
        } else if (wolff) {
            const sga::WolffArgs wa{e->wolff_u, e->wolff_cap, e->wolff_cursor};
            le = sga::launch_sweep_wolff(a, wa, e->csr, e->want_i8, st);
        } else if (e->tsp) {
            a.table_m = 0;
            le = sga::launch_sweep_tsp(a, e->tsp_args, e->tsp_waves, e->tsp_passes, st);
        } else if (e->csr) {
            le = sga::launch_sweep_csr(a, e->waves, st);
        } else {
            le = launch_dense_rows(a, st);
        }
        if (e->timing) {
            (void)hipEventRecord(ev1, st);
            e->events.emplace_back(ev0, ev1);
        }
        if (le != hipSuccess) (void)hipStreamSynchronize(st);  // staged inputs / scratch slots are reusable again
        HIPCHK(le);
        if (exact_mode) {
            int rc2 = recompute_energy_range(e, 0, R);
            if (rc2 != SGA_OK) return rc2;
            HIPCHK(sga::launch_update_best(e->energy, e->spins, e->best_energy, e->best_spins,
                                           e->sstride, R, st));
            if (d_etrace.ptr)
                HIPCHK(hipMemcpyAsync(d_etrace.ptr + (long long)k0 * R, e->energy, sizeof(double) * R,
                                      hipMemcpyDeviceToDevice, st));
        }
    }
    e->sweeps_done += (uint32_t)n_sweeps;
    e->attempted += per;

    rc = d_etrace.flush(st);
    if (rc != SGA_OK) return rc;
    rc = d_acc.flush(st);
    if (rc != SGA_OK) return rc;
    rc = d_dE.flush(st);
    if (rc != SGA_OK) return rc;
    // host-side outputs must be complete, and staged host inputs consumed, before returning
    if (d_sched.staged || d_site.staged || d_u.staged || d_etrace.ptr || d_acc.ptr || d_dE.ptr)
        HIPCHK(hipStreamSynchronize(st));
    return SGA_OK;
}

int sga_set_update_rule(sga_engine *e, int rule) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (rule < SGA_RULE_METROPOLIS || rule > SGA_RULE_WOLFF)
        return fail(SGA_ERR_UNSUPPORTED, "update rule not implemented by the engine");
    if (rule == SGA_RULE_WOLFF && e->tsp)
        return fail(SGA_ERR_UNSUPPORTED, "the Wolff rule is not implemented for sga_set_tsp problems");
    e->rule = rule;
    return SGA_OK;
}

int sga_set_wolff_replay(sga_engine *e, const float *u, int64_t capacity) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    dev_free(e->wolff_u);
    dev_free(e->wolff_cursor);
    e->wolff_cap = 0;
    if (!u || capacity <= 0) return SGA_OK;
    HIPCHK(hipMalloc(&e->wolff_u, sizeof(float) * (size_t)e->R * (size_t)capacity));
    HIPCHK(hipMalloc(&e->wolff_cursor, sizeof(long long) * (size_t)e->R));
    HIPCHK(hipMemcpy(e->wolff_u, u, sizeof(float) * (size_t)e->R * (size_t)capacity, hipMemcpyDefault));
    HIPCHK(hipMemset(e->wolff_cursor, 0, sizeof(long long) * (size_t)e->R));
    e->wolff_cap = capacity;
    return SGA_OK;
}

int sga_recompute_energies(sga_engine *e) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    HIPCHK(hipSetDevice(e->device));
    return recompute_energy_range(e, 0, e->R);
}

static int point_op(sga_engine *e, int r, const int32_t *sites, int count, int op, double T,
                    float u, int arith, double *out_host, int out_count) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0 || r < 0 || r >= e->R) return fail(SGA_ERR_INVALID, "bad replica index");
    if (!sites || count <= 0) return fail(SGA_ERR_INVALID, "no sites");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    std::vector<int32_t> hs((size_t)count);
    if (is_device_ptr(sites))
        HIPCHK(hipMemcpy(hs.data(), sites, sizeof(int32_t) * hs.size(), hipMemcpyDeviceToHost));
    else
        std::memcpy(hs.data(), sites, sizeof(int32_t) * hs.size());
    for (int32_t v : hs)
        if (v < 0 || v >= e->n) return fail(SGA_ERR_INVALID, "site index out of range");
    if (op == 2 && e->rule == SGA_RULE_WOLFF)
        return fail(SGA_ERR_UNSUPPORTED, "sga_update applies single-site rules; Wolff moves run through sga_sweep");
    if (e->tsp) {  // structured couplings: local fields only (flip / update go through sweeps)
        if (op != 0)
            return fail(SGA_ERR_UNSUPPORTED, "single-site flip / update are not implemented for sga_set_tsp problems");
        HIPCHK(e->point_sites.reserve(sizeof(int32_t) * hs.size()));
        HIPCHK(e->point_out.reserve(sizeof(double) * (size_t)out_count));
        HIPCHK(hipMemcpyAsync(e->point_sites.ptr, hs.data(), sizeof(int32_t) * hs.size(), hipMemcpyHostToDevice, st));
        HIPCHK(sga::launch_fields_tsp(e->tsp_args, e->spins + (long long)r * e->sstride, e->h,
                                      static_cast<const int32_t *>(e->point_sites.ptr), count,
                                      static_cast<double *>(e->point_out.ptr), st));
        HIPCHK(hipMemcpyAsync(out_host, e->point_out.ptr, sizeof(double) * (size_t)out_count, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return SGA_OK;
    }
    // staging in grow-only scratch slots (no allocation / free per call)
    HIPCHK(e->point_sites.reserve(sizeof(int32_t) * hs.size()));
    HIPCHK(e->point_out.reserve(sizeof(double) * (size_t)std::max(out_count, 2)));
    int32_t *d_sites = static_cast<int32_t *>(e->point_sites.ptr);
    double *d_out = static_cast<double *>(e->point_out.ptr);
    hipError_t he = hipMemcpyAsync(d_sites, hs.data(), sizeof(int32_t) * hs.size(), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) {
        sga::PointArgs a{};
        const long long model = e->n_models > 1 ? (e->replica0 + r) / (e->Rg / e->n_models) : 0;
        a.J = e->J_packed;
        a.model_offset_j = model * e->n * e->ldj;
        a.rowptr = e->rowptr64;
        a.cv = e->cv;
        a.h = e->h + model * e->n;
        a.diag = e->diag + model * e->n;
        a.spins = e->spins + (long long)r * e->sstride;
        a.energy = e->energy + r;
        a.n_accepted = e->n_acc + r;
        a.sites = d_sites;
        a.out = d_out;
        a.ld = e->ld;
        a.ldj = e->ldj;
        a.n = e->n;
        a.count = count;
        a.op = op;
        a.arith = arith;
        a.rule = e->rule;
        a.T = T;
        a.u = u;
        he = sga::launch_point_op(a, e->csr, e->want_i8, st);
    }
    if (he == hipSuccess)
        he = hipMemcpyAsync(out_host, d_out, sizeof(double) * (size_t)out_count, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    HIPCHK(he);
    if (op != 0) e->fields_valid = false;
    if (op != 0 && !e->consistent_dE) {  // the rule's dE is not the energy change here
        int rc = recompute_energy_range(e, r, 1);
        if (rc != SGA_OK) return rc;
        HIPCHK(hipStreamSynchronize(st));
    }
    return SGA_OK;
}

int sga_local_fields(sga_engine *e, int r, const int32_t *sites, int count, double *out) {
    if (!out) return fail(SGA_ERR_INVALID, "out is NULL");
    if (is_device_ptr(out)) return fail(SGA_ERR_INVALID, "out must be a host buffer");
    return point_op(e, r, sites, count, 0, 1.0, 0.0f, SGA_ARITH_F64, out, count);
}

int sga_flip(sga_engine *e, int r, int site, double *dE) {
    double o[2] = {0.0, 0.0};
    const int32_t s = site;
    int rc = point_op(e, r, &s, 1, 1, 1.0, 0.0f, SGA_ARITH_F64, o, 2);
    if (rc == SGA_OK && dE) *dE = o[0];
    return rc;
}

int sga_update(sga_engine *e, int r, int site, double T, float u, int arith, int *accepted,
               double *dE) {
    if (arith != SGA_ARITH_F64 && arith != SGA_ARITH_F32) return fail(SGA_ERR_INVALID, "bad arith");
    double o[2] = {0.0, 0.0};
    const int32_t s = site;
    int rc = point_op(e, r, &s, 1, 2, T, u, arith, o, 2);
    if (rc == SGA_OK) {
        if (accepted) *accepted = o[1] != 0.0;
        if (dE) *dE = o[0];
    }
    return rc;
}

int sga_exchange(sga_engine *e, const double *energies_global, const int32_t *start,
                 const double *u, int *n_accepted) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->n_ladders <= 0) return fail(SGA_ERR_INVALID, "no ladder (call sga_set_ladder)");
    const int L = e->Rg / e->n_ladders;
    // whole ladders on this rank: their rounds need nobody else's energies (zero exchange traffic)
    const bool ladders_local = !energies_global && e->R != e->Rg && e->replica0 % L == 0 && e->R % L == 0;
    if (!energies_global && e->R != e->Rg && !ladders_local)
        return fail(SGA_ERR_INVALID, "sharded replicas need the all-gathered energies (unless every ladder lies "
                                     "whole on one rank)");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    DevIn<double> d_e, d_u;
    DevIn<int32_t> d_start;
    int rc;
    if (energies_global) {
        rc = d_e.init(e->scratch[6], energies_global, (size_t)e->Rg, st);
        if (rc != SGA_OK) return rc;
    }
    if (start) {
        rc = d_start.init(e->scratch[7], start, (size_t)e->n_ladders, st);
        if (rc != SGA_OK) return rc;
    }
    if (u) {
        rc = d_u.init(e->scratch[8], u, (size_t)e->n_ladders * (L / 2), st);
        if (rc != SGA_OK) return rc;
    }
    HIPCHK(hipMemsetAsync(e->d_count, 0, sizeof(int), st));
    sga::ExchangeArgs a{};
    a.energies = energies_global ? d_e.ptr : e->energy;
    a.slot_temps = e->slot_temps;
    a.slot_to_rep = e->slot_to_rep;
    a.rep_temp = e->rep_temp;
    a.attempts = e->ex_attempts;
    a.accepts = e->ex_accepts;
    a.start = d_start.ptr;
    a.u = d_u.ptr;
    a.n_accepted = e->d_count;
    a.R_global = e->Rg;
    a.R_local = e->R;
    a.replica0 = e->replica0;
    a.n_ladders = e->n_ladders;
    a.seed_lo = (uint32_t)e->seed;
    a.seed_hi = (uint32_t)(e->seed >> 32);
    a.round = e->rounds;
    a.ladder0 = ladders_local ? e->replica0 / L : 0;
    a.n_ladders_local = ladders_local ? e->R / L : e->n_ladders;
    a.energy_base = ladders_local ? e->replica0 : 0;
    HIPCHK(sga::launch_exchange_neighbor(a, st));
    e->rounds += 1;
    if (n_accepted) {
        HIPCHK(hipMemcpyAsync(n_accepted, e->d_count, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    } else if (d_e.staged || d_u.staged || d_start.staged) {
        HIPCHK(hipStreamSynchronize(st));  // the host buffers may be reused by the caller
    }
    return SGA_OK;
}

int sga_exchange_pairs(sga_engine *e, const double *energies_global, const int32_t *pairs,
                       const double *u, int count, int *n_accepted) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->n_ladders <= 0) return fail(SGA_ERR_INVALID, "no ladder (call sga_set_ladder)");
    if (!energies_global && e->R != e->Rg)
        return fail(SGA_ERR_INVALID, "sharded replicas need the all-gathered energies");
    if (count < 0 || (count > 0 && !pairs)) return fail(SGA_ERR_INVALID, "bad pair list");
    if (is_device_ptr(pairs)) return fail(SGA_ERR_INVALID, "pairs must be a host buffer");
    for (int k = 0; k < 2 * count; ++k)
        if (pairs[k] < 0 || pairs[k] >= e->Rg) return fail(SGA_ERR_INVALID, "slot index out of range");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    DevIn<double> d_e, d_u;
    DevIn<int32_t> d_pairs;
    int rc;
    if (energies_global) {
        rc = d_e.init(e->scratch[6], energies_global, (size_t)e->Rg, st);
        if (rc != SGA_OK) return rc;
    }
    rc = d_pairs.init(e->scratch[7], pairs, (size_t)2 * count, st);
    if (rc != SGA_OK) return rc;
    if (u) {
        rc = d_u.init(e->scratch[8], u, (size_t)count, st);
        if (rc != SGA_OK) return rc;
    }
    sga::ExchangeArgs a{};
    a.energies = energies_global ? d_e.ptr : e->energy;
    a.slot_temps = e->slot_temps;
    a.slot_to_rep = e->slot_to_rep;
    a.rep_temp = e->rep_temp;
    a.attempts = e->ex_attempts;
    a.accepts = e->ex_accepts;
    a.u = d_u.ptr;
    a.n_accepted = e->d_count;
    a.R_global = e->Rg;
    a.R_local = e->R;
    a.replica0 = e->replica0;
    a.n_ladders = e->n_ladders;
    a.seed_lo = (uint32_t)e->seed;
    a.seed_hi = (uint32_t)(e->seed >> 32);
    a.round = e->rounds;
    HIPCHK(sga::launch_exchange_pairs(a, d_pairs.ptr, count, st));
    e->rounds += 1;
    int cnt = 0;  // (the pair list was staged from the host: synchronise in any case)
    HIPCHK(hipMemcpyAsync(&cnt, e->d_count, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_accepted) *n_accepted = cnt;
    return SGA_OK;
}

int sga_op_pt_exchange(int device, float *spins, float *energies, const float *temps,
                       const float *u, uint64_t seed, uint32_t round, int R, int n,
                       int *n_accepted) {
    if (!spins || !energies || !temps || R <= 0 || n <= 0)
        return fail(SGA_ERR_INVALID, "bad operator-exchange arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(SGA_ERR_DEVICE, "no such HIP device");
    HIPCHK(hipSetDevice(device));
    hipStream_t st = nullptr;  // default stream: ordered with the caller's legacy-stream work
    const size_t rows = (size_t)R * n;
    float *d_spins = nullptr, *d_tmp = nullptr, *d_en = nullptr;
    int32_t *d_src = nullptr;
    int *d_cnt = nullptr;
    DevIn<float> d_t, d_uu;
    Scratch tmp_t, tmp_u;  // stateless entry point: its own short-lived staging
    struct Release {
        Scratch &a, &b;
        ~Release() { a.release(); b.release(); }
    } release_on_exit{tmp_t, tmp_u};
    int rc = d_t.init(tmp_t, temps, (size_t)R, st);
    if (rc != SGA_OK) return rc;
    if (u && R > 1) {
        rc = d_uu.init(tmp_u, u, (size_t)(R - 1), st);
        if (rc != SGA_OK) return rc;
    }
    const bool spins_dev = is_device_ptr(spins), en_dev = is_device_ptr(energies);
    auto cleanup = [&]() {
        if (!spins_dev) dev_free(d_spins);
        if (!en_dev) dev_free(d_en);
        dev_free(d_tmp);
        dev_free(d_src);
        dev_free(d_cnt);
    };
    hipError_t he = hipSuccess;
    auto step = [&](hipError_t x) {
        if (he == hipSuccess) he = x;
    };
    if (spins_dev) d_spins = spins; else step(hipMalloc(&d_spins, rows * sizeof(float)));
    if (en_dev) d_en = energies; else step(hipMalloc(&d_en, (size_t)R * sizeof(float)));
    step(hipMalloc(&d_tmp, rows * sizeof(float)));
    step(hipMalloc(&d_src, (size_t)R * sizeof(int32_t)));
    step(hipMalloc(&d_cnt, sizeof(int)));
    if (he == hipSuccess && !spins_dev)
        step(hipMemcpyAsync(d_spins, spins, rows * sizeof(float), hipMemcpyHostToDevice, st));
    if (he == hipSuccess && !en_dev)
        step(hipMemcpyAsync(d_en, energies, (size_t)R * sizeof(float), hipMemcpyHostToDevice, st));
    if (he == hipSuccess)
        step(sga::launch_op_exchange(d_spins, d_tmp, d_en, d_t.ptr, d_uu.ptr, d_src, d_cnt,
                                     (uint32_t)seed, (uint32_t)(seed >> 32), round, R, n, st));
    if (he == hipSuccess && !spins_dev)
        step(hipMemcpyAsync(spins, d_spins, rows * sizeof(float), hipMemcpyDeviceToHost, st));
    if (he == hipSuccess && !en_dev)
        step(hipMemcpyAsync(energies, d_en, (size_t)R * sizeof(float), hipMemcpyDeviceToHost, st));
    int cnt = 0;
    if (he == hipSuccess) step(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st));
    if (he == hipSuccess) step(hipStreamSynchronize(st));
    cleanup();
    HIPCHK(he);
    if (n_accepted) *n_accepted = cnt;
    return SGA_OK;
}

// ---- state access -------------------------------------------------------------------------
int sga_get_energies(sga_engine *e, double *out) {
    if (!e || !out) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(out, e->energy, sizeof(double) * e->R, hipMemcpyDefault, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_get_temperatures(sga_engine *e, double *out) {
    if (!e || !out) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(out, e->rep_temp, sizeof(double) * e->R, hipMemcpyDefault, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_get_spins(sga_engine *e, int r, int8_t *out) {
    if (!e || !out) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->R <= 0 || r >= e->R) return fail(SGA_ERR_INVALID, "bad replica index");
    HIPCHK(hipSetDevice(e->device));
    if (r >= 0) {
        HIPCHK(hipMemcpyAsync(out, e->spins + (long long)r * e->sstride, (size_t)e->n,
                              hipMemcpyDefault, e->stream));
    } else {
        HIPCHK(hipMemcpy2DAsync(out, (size_t)e->n, e->spins, (size_t)e->sstride, (size_t)e->n,
                                (size_t)e->R, hipMemcpyDefault, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_set_spins(sga_engine *e, int r, const int8_t *s) {
    if (!e || !s) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->R <= 0 || r < 0 || r >= e->R) return fail(SGA_ERR_INVALID, "bad replica index");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemsetAsync(e->spins + (long long)r * e->sstride, 0, (size_t)e->sstride, e->stream));
    HIPCHK(hipMemcpyAsync(e->spins + (long long)r * e->sstride, s, (size_t)e->n, hipMemcpyDefault,
                          e->stream));
    e->fields_valid = false;
    int rc = recompute_energy_range(e, r, 1);
    if (rc != SGA_OK) return rc;
    HIPCHK(sga::launch_copy_best(e->energy + r, e->spins + (long long)r * e->sstride,
                                 e->best_energy + r, e->best_spins + (long long)r * e->sstride,
                                 e->sstride, 1, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_get_best(sga_engine *e, int r, double *energy, int8_t *spins, int *r_out) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0 || r >= e->R) return fail(SGA_ERR_INVALID, "bad replica index");
    HIPCHK(hipSetDevice(e->device));
    std::vector<double> be((size_t)e->R);
    HIPCHK(hipMemcpyAsync(be.data(), e->best_energy, sizeof(double) * e->R, hipMemcpyDeviceToHost,
                          e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (r < 0) {
        r = 0;
        for (int i = 1; i < e->R; ++i)
            if (be[i] < be[r]) r = i;
    }
    if (energy) *energy = be[r];
    if (r_out) *r_out = r;
    if (spins) {
        HIPCHK(hipMemcpyAsync(spins, e->best_spins + (long long)r * e->sstride, (size_t)e->n,
                              hipMemcpyDefault, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return SGA_OK;
}

int sga_reset_best(sga_engine *e) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(sga::launch_copy_best(e->energy, e->spins, e->best_energy, e->best_spins, e->sstride,
                                 e->R, e->stream));
    return SGA_OK;
}

int sga_get_stats(sga_engine *e, int64_t *accepted, int64_t *attempted) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    HIPCHK(hipSetDevice(e->device));
    if (accepted) {
        HIPCHK(hipMemcpyAsync(accepted, e->n_acc, sizeof(int64_t) * e->R, hipMemcpyDefault,
                              e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (attempted) {
        if (is_device_ptr(attempted)) return fail(SGA_ERR_INVALID, "attempted must be a host buffer");
        for (int i = 0; i < e->R; ++i) attempted[i] = e->attempted;
    }
    return SGA_OK;
}

int sga_get_slot_map(sga_engine *e, int32_t *slot_to_rep) {
    if (!e || !slot_to_rep) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->n_ladders <= 0) return fail(SGA_ERR_INVALID, "no ladder");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(slot_to_rep, e->slot_to_rep, sizeof(int32_t) * e->Rg, hipMemcpyDefault,
                          e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_get_exchange_stats(sga_engine *e, int64_t *attempts, int64_t *accepts) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->n_ladders <= 0) return fail(SGA_ERR_INVALID, "no ladder");
    HIPCHK(hipSetDevice(e->device));
    if (attempts)
        HIPCHK(hipMemcpyAsync(attempts, e->ex_attempts, sizeof(int64_t) * e->Rg, hipMemcpyDefault,
                              e->stream));
    if (accepts)
        HIPCHK(hipMemcpyAsync(accepts, e->ex_accepts, sizeof(int64_t) * e->Rg, hipMemcpyDefault,
                              e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_snapshot(sga_engine *e, double *energies, int64_t *accepted, int32_t *slot_to_rep) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    if (slot_to_rep && e->n_ladders <= 0) return fail(SGA_ERR_INVALID, "no ladder");
    HIPCHK(hipSetDevice(e->device));
    if (energies)
        HIPCHK(hipMemcpyAsync(energies, e->energy, sizeof(double) * e->R, hipMemcpyDefault, e->stream));
    if (accepted)
        HIPCHK(hipMemcpyAsync(accepted, e->n_acc, sizeof(int64_t) * e->R, hipMemcpyDefault, e->stream));
    if (slot_to_rep)
        HIPCHK(hipMemcpyAsync(slot_to_rep, e->slot_to_rep, sizeof(int32_t) * e->Rg, hipMemcpyDefault, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGA_OK;
}

int sga_set_seed(sga_engine *e, uint64_t seed) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    e->seed = seed;
    return SGA_OK;
}

int sga_get_sweep_counter(sga_engine *e, uint32_t *sweeps_done, uint32_t *exchange_rounds) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (sweeps_done) *sweeps_done = e->sweeps_done;
    if (exchange_rounds) *exchange_rounds = e->rounds;
    return SGA_OK;
}

int sga_set_sweep_counter(sga_engine *e, uint32_t sweeps_done, uint32_t exchange_rounds) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    e->sweeps_done = sweeps_done;
    e->rounds = exchange_rounds;
    return SGA_OK;
}

// ---- checkpoint / resume -------------------------------------------------------------------
// The blob is independent of the launch geometry: spins travel unpadded ([R][n]), so a state
// exported after sga_autotune / sga_set_tuning imports into an engine laid out for any other
// waves-per-replica.  (The chain itself does not depend on the geometry either: integer problems
// sum exactly, real-valued ones in the canonical chunk order of sweep_dense_impl.h.)
namespace {
struct StateHeader {
    uint64_t magic;
    int32_t version, n, R, Rg, replica0, n_ladders;
    uint32_t sweeps_done, rounds;
    uint64_t seed;
    int64_t attempted;
};
constexpr uint64_t STATE_MAGIC = 0x5347415354415445ull;  // "SGASTATE"
constexpr int32_t STATE_VERSION = 2;

uint64_t state_bytes(const sga_engine *e) {
    const uint64_t R = (uint64_t)e->R, Rg = (uint64_t)e->Rg, sb = R * (uint64_t)e->n;
    uint64_t total = sizeof(StateHeader) + 2 * sb + 3 * R * sizeof(double) + R * sizeof(uint64_t);
    if (e->n_ladders > 0) total += Rg * (sizeof(int32_t) + 2 * sizeof(int64_t));
    return total;
}
}  // namespace

int sga_export_state(sga_engine *e, void *buf, uint64_t capacity, uint64_t *needed) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    const uint64_t total = state_bytes(e);
    if (needed) *needed = total;
    if (!buf) return SGA_OK;
    if (capacity < total) return fail(SGA_ERR_INVALID, "state buffer too small");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    unsigned char *p = static_cast<unsigned char *>(buf);
    StateHeader h{STATE_MAGIC, STATE_VERSION, e->n, e->R, e->Rg, e->replica0, e->n_ladders,
                  e->sweeps_done, e->rounds, e->seed, (int64_t)e->attempted};
    std::memcpy(p, &h, sizeof(h));
    p += sizeof(h);
    auto pull = [&](const void *dev, size_t bytes) -> hipError_t {
        hipError_t r = hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost);
        p += bytes;
        return r;
    };
    const size_t R = (size_t)e->R, Rg = (size_t)e->Rg, sb = R * (size_t)e->n;
    // spins leave the padded device layout through a staging slot
    HIPCHK(e->scratch[1].reserve(sb));
    int8_t *stage = static_cast<int8_t *>(e->scratch[1].ptr);
    for (const int8_t *src : {e->spins, e->best_spins}) {
        HIPCHK(sga::launch_unpad_spins(src, e->sstride, stage, e->n, e->R, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(pull(stage, sb));
    }
    HIPCHK(pull(e->energy, R * sizeof(double)));
    HIPCHK(pull(e->best_energy, R * sizeof(double)));
    HIPCHK(pull(e->rep_temp, R * sizeof(double)));
    HIPCHK(pull(e->n_acc, R * sizeof(uint64_t)));
    if (e->n_ladders > 0) {
        HIPCHK(pull(e->slot_to_rep, Rg * sizeof(int32_t)));
        HIPCHK(pull(e->ex_attempts, Rg * sizeof(int64_t)));
        HIPCHK(pull(e->ex_accepts, Rg * sizeof(int64_t)));
    }
    return SGA_OK;
}

int sga_import_state(sga_engine *e, const void *buf, uint64_t size) {
    if (!e || !buf) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "initialise the replicas before importing a state");
    if (size < sizeof(StateHeader)) return fail(SGA_ERR_INVALID, "state blob truncated");
    StateHeader h;
    std::memcpy(&h, buf, sizeof(h));
    if (h.magic != STATE_MAGIC) return fail(SGA_ERR_INVALID, "not an engine state blob");
    if (h.version != STATE_VERSION) return fail(SGA_ERR_INVALID, "state blob of another engine version");
    if (h.n != e->n || h.R != e->R || h.Rg != e->Rg || h.replica0 != e->replica0 ||
        h.n_ladders != e->n_ladders)
        return fail(SGA_ERR_INVALID, "state blob does not match this engine's problem / replicas / ladder");
    if (size != state_bytes(e)) return fail(SGA_ERR_INVALID, "state blob has the wrong size");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const unsigned char *p = static_cast<const unsigned char *>(buf) + sizeof(h);
    auto push = [&](void *dev, size_t bytes) -> hipError_t {
        hipError_t r = hipMemcpy(dev, p, bytes, hipMemcpyHostToDevice);
        p += bytes;
        return r;
    };
    const size_t R = (size_t)e->R, Rg = (size_t)e->Rg, sb = R * (size_t)e->n;
    HIPCHK(e->scratch[1].reserve(sb));
    int8_t *stage = static_cast<int8_t *>(e->scratch[1].ptr);
    for (int8_t *dst : {e->spins, e->best_spins}) {
        HIPCHK(push(stage, sb));
        HIPCHK(sga::launch_pad_spins(stage, e->n, dst, e->sstride, e->R, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    HIPCHK(push(e->energy, R * sizeof(double)));
    HIPCHK(push(e->best_energy, R * sizeof(double)));
    HIPCHK(push(e->rep_temp, R * sizeof(double)));
    HIPCHK(push(e->n_acc, R * sizeof(uint64_t)));
    if (e->n_ladders > 0) {
        HIPCHK(push(e->slot_to_rep, Rg * sizeof(int32_t)));
        HIPCHK(push(e->ex_attempts, Rg * sizeof(int64_t)));
        HIPCHK(push(e->ex_accepts, Rg * sizeof(int64_t)));
    }
    e->sweeps_done = h.sweeps_done;
    e->rounds = h.rounds;
    e->seed = h.seed;
    e->attempted = h.attempted;
    e->fields_valid = false;
    return SGA_OK;
}

// ---- measurement --------------------------------------------------------------------------
int sga_enable_timing(sga_engine *e, int on) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    e->timing = on != 0;
    return SGA_OK;
}

int sga_get_kernel_time(sga_engine *e, int64_t *n_launches, double *total_ms, int reset) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (auto &p : e->events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            e->total_ms += ms;
            e->launches += 1;
        }
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    e->events.clear();
    if (n_launches) *n_launches = e->launches;
    if (total_ms) *total_ms = e->total_ms;
    if (reset) {
        e->launches = 0;
        e->total_ms = 0.0;
    }
    return SGA_OK;
}

int sga_describe(sga_engine *e, char *buf, int buflen) {
    if (!e || !buf || buflen <= 0) return fail(SGA_ERR_INVALID, "bad arguments");
    char tmp[512];
    if (e->tsp)
        std::snprintf(tmp, sizeof(tmp),
                      "tsp n_cities=%d n=%d R=%d waves_per_replica=%d passes=%d couplings=implicit "
                      "(2 x %d-byte distance rows per update) acc=%s lds_bytes=%zu",
                      e->tsp_args.n_cities, e->n, e->R, e->tsp_waves, e->tsp_passes, 4 * e->tsp_args.n_cities,
                      !e->tsp_args.f64 ? "f32-exact" : (e->tsp_exact ? "f64-exact" : "f64"),
                      sga::tsp_lds_bytes(e->tsp_args.n_cities, e->tsp_args.npad));
    else if (e->csr)
        std::snprintf(tmp, sizeof(tmp),
                      "csr n=%d nnz=%lld R=%d waves_per_replica=%d replicas_per_block=%d sstride=%d "
                      "path=%s table_m=%d spins=%s",
                      e->n, e->nnz, e->R, e->waves,
                      e->big_form == 2 ? sga::csr_bits_waves_per_block(e->sstride, e->table_m)
                                       : ((e->waves > 1 || e->big) ? 1 : sga::csr_waves_per_block(e->sstride, e->table_m)),
                      e->sstride,
                      (e->csr_acc == sga::CSR_ACC_F32_TABLE && e->table_m > 0) ? (e->table_scale == 2 ? "half-integer-fast" : "integer-fast")
                      : e->csr_acc == sga::CSR_ACC_F32_TABLE ? "general acc=f32-exact"
                      : e->csr_acc == sga::CSR_ACC_F32     ? "general acc=f32-exact"
                      : e->csr_acc == sga::CSR_ACC_F64     ? "general acc=f64-exact"
                                                           : "general acc=f64-canonical",
                      e->table_m,
                      e->big ? "lds-bits" : "lds-int8");
    else
        std::snprintf(tmp, sizeof(tmp),
                      "dense n=%d models=%d storage=%s acc=%s R=%d waves_per_replica=%d "
                      "chunks_per_wave=%d%s ld=%lld row_bytes=%lld table_m=%d look_ahead=%d",
                      e->n, e->n_models, e->use_t2 ? "t2" : (e->want_i8 ? "i8" : "f32"),
                      e->want_i8 ? "i32" : (e->acc64 ? (e->acc_canon ? "f64-canonical" : "f64-exact") : "f32"), e->R,
                      e->use_t2 ? e->waves_t2 : e->waves, e->use_t2 ? e->cpw_t2 : e->cpw,
                      (e->use_t2 ? e->cpw_t2 > sga::T2_MAX_CPW : e->cpw > sga::MAX_CPW) ? "(streaming)" : "", e->ld,
                      e->use_t2 ? t2_row_bits(e->n) / 4 : e->ldj * (e->want_i8 ? 1 : 4), e->table_m,
                      (e->table_m > 0 && e->opt[OPT_LOOK_AHEAD] != 0)
                          ? sga::dense_look_ahead(e->use_t2, e->want_i8, e->acc64,
                                                  e->use_t2 ? e->cpw_t2 : e->cpw,
                                                  e->use_t2 ? e->waves_t2 : e->waves, e->R)
                          : 1);
    if (e->csr && csr_updates_per_step(e) >= 4 && e->waves <= 1 && (e->big_form == 0 || e->big_form == 2) && e->rowptr)
        std::snprintf(tmp + std::strlen(tmp), sizeof(tmp) - std::strlen(tmp), " updates_per_step=%d", csr_updates_per_step(e));
    if (e->csr && e->from_dense) std::strncat(tmp, " source=dense-matrix(sparse)", sizeof(tmp) - std::strlen(tmp) - 1);
    if (e->csr && e->slotted)
        std::snprintf(tmp + std::strlen(tmp), sizeof(tmp) - std::strlen(tmp),
                      " rows=64-entry-slots(+%.1f%%) longest_row_slots=%lld",
                      e->nnz > 0 ? 100.0 * (double)(e->layout_entries - e->nnz) / (double)e->nnz : 0.0,
                      (e->max_row_len + 63) / 64);
    if (e->csr && e->big_form == 1 && e->cvp && e->csr_storage_latched != SGA_CSR_STORAGE_F32)
        std::strncat(tmp, " entries=packed-32bit", sizeof(tmp) - std::strlen(tmp) - 1);
    if (!e->consistent_dE) std::strncat(tmp, " energy=recomputed-per-sweep", sizeof(tmp) - std::strlen(tmp) - 1);
    if (clf_active(e) && e->csr) {
        if (e->field_cache == SGA_FIELD_CACHE_ON)
            std::snprintf(tmp + std::strlen(tmp), sizeof(tmp) - std::strlen(tmp),
                          " sweep=cached-local-fields(int16 dynamic fields in LDS, row entries read on accept only)");
        else
            std::snprintf(tmp + std::strlen(tmp), sizeof(tmp) - std::strlen(tmp),
                          " sweep=auto(cached local fields while the hottest replica accepts little; now: %s)",
                          (!e->auto_unavailable && e->n_route_clf > 0) ? "cached" : "one row per proposal");
    } else if (clf_active(e)) {
        if (e->field_cache == SGA_FIELD_CACHE_ON)
            std::snprintf(tmp + std::strlen(tmp), sizeof(tmp) - std::strlen(tmp),
                          " sweep=cached-local-fields(int%d in LDS, %d wave(s) per replica%s, row read on accept only)",
                          e->clf_bits, sga::sweep_clf_waves(e->ldj, e->want_i8, e->R, e->cus, (int)e->opt[OPT_CLF_WAVES]),
                          e->clf_wide ? " -- now 8: the launch is its hottest replica's chain" : "");
        else
            std::snprintf(tmp + std::strlen(tmp), sizeof(tmp) - std::strlen(tmp),
                          " sweep=auto(cached local fields, int%d in LDS, per replica by its own acceptance; now: %d of %d "
                          "replica(s) cached%s, the rest one row per proposal)",
                          e->clf_bits, e->auto_unavailable ? 0 : e->n_route_clf, e->R,
                          (!e->auto_unavailable && e->clf_wide) ? " at 8 waves each" : "");
    }
    std::snprintf(buf, (size_t)buflen, "%s", tmp);
    return SGA_OK;
}

int sga_problem_checksum(sga_engine *e, uint64_t *out) {
    if (!e || !out) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->n <= 0) return fail(SGA_ERR_INVALID, "no couplings set");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(e->point_out.reserve(2 * sizeof(unsigned long long)));
    unsigned long long *d = static_cast<unsigned long long *>(e->point_out.ptr);
    HIPCHK(hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), e->stream));
    // what the sweep kernels read: the packed matrix | the entry layout | the distance tables; then h
    if (e->tsp) {
        const long long bytes = 4ll * e->tsp_args.n_cities * e->tsp_args.npad;
        HIPCHK(sga::launch_checksum(e->nd4, bytes, d, e->stream));
        HIPCHK(sga::launch_checksum(e->nd4t, bytes, d, e->stream));
    } else if (e->csr) {
        HIPCHK(sga::launch_checksum(e->cv, 8ll * e->layout_entries, d, e->stream));
        HIPCHK(sga::launch_checksum(e->rowptr64, 8ll * ((long long)e->n + 1), d, e->stream));
    } else {
        HIPCHK(sga::launch_checksum(e->J_packed, (long long)e->n_models * e->n * e->ldj * (e->want_i8 ? 1 : 4), d,
                                    e->stream));
    }
    HIPCHK(sga::launch_checksum(e->h, 4ll * e->n * (e->tsp ? 1 : e->n_models), d + 1, e->stream));
    unsigned long long host[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(host, d, sizeof(host), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *out = host[0] ^ ((host[1] << 17) | (host[1] >> 47)) ^ ((uint64_t)(uint32_t)e->n << 32);
    return SGA_OK;
}

int sga_get_geometry(sga_engine *e, int *waves_per_replica, int *chunks_per_wave) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (waves_per_replica) *waves_per_replica = (!e->csr && !e->tsp && e->use_t2) ? e->waves_t2 : e->waves;
    if (chunks_per_wave) *chunks_per_wave = (!e->csr && !e->tsp && e->use_t2) ? e->cpw_t2 : e->cpw;
    return SGA_OK;
}

int sga_get_energies_async(sga_engine *e, double *out_device) {
    if (!e || !out_device) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->R <= 0) return fail(SGA_ERR_INVALID, "no replicas");
    if (!is_device_ptr(out_device)) return fail(SGA_ERR_INVALID, "sga_get_energies_async needs a device buffer");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(out_device, e->energy, sizeof(double) * e->R, hipMemcpyDeviceToDevice, e->stream));
    return SGA_OK;
}

int sga_last_kernel(char *buf, int buflen) {
    if (!buf || buflen <= 0) return fail(SGA_ERR_INVALID, "bad arguments");
    std::snprintf(buf, (size_t)buflen, "%s", sga::last_sweep_kernel());
    return SGA_OK;
}

}  // extern "C"
