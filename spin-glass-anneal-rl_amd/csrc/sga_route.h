// sga_route.h -- WHICH kernel form sweeps a problem: pure functions of a sga_route_query (include/sga.h: the problem's
// traits as the set-time scans found them, replica count, tuning, options).  No device call anywhere in sga_route.cpp;
// the LDS-footprint helpers it uses (sga_kernels.h) are plain arithmetic on the kernels' layouts.  The engine asks
// these functions at the stage where each decision is latched (set / replicas / sweep); sga_explain_route strings the
// same calls together for tests.
#ifndef SGA_ROUTE_H
#define SGA_ROUTE_H
#include <cstring>
#include <string>

#include "sga.h"

namespace sga_impl {

inline thread_local std::string g_last_error;

inline int fail(int code, const std::string &msg) {
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
    int stage = 0;       // where the value is latched: 0 = every sga_sweep, 1 = sga_init_replicas, 2 = sga_set_dense / sga_set_csr
};
constexpr OptDef OPT_DEFS[OPT_COUNT] = {
    {"look_ahead", "SGA_NO_LOOK_AHEAD", 1, 0, 1, 0, 1, 0},
    {"clf_waves", "SGA_CLF_WAVES", 0, 0, 0, 0, 16, 0},
    {"sparse_route", "SGA_NO_SPARSE_ROUTE", 1, 0, 1, 0, 1, 2},
    {"batched_energy", "SGA_NO_MFMA_ENERGY", 1, 0, 1, 0, 2, 0},
    {"force_general", "SGA_FORCE_GENERAL", 1, 1, 0, 0, 1, 0},
    {"csr_updates_per_step", "SGA_CSR_PAIR_AHEAD", 0, 0, -1, -1, 8, 0},
    {"tsp_updates_per_step", "SGA_TSP_PARALLEL", 0, 0, -1, -1, 8, 0},
    {"force_csr_bits", "SGA_FORCE_CSR_BIG", 1, 1, 0, 0, 1, 1},
    {"csr_bits", "SGA_NO_CSR_BITS", 1, 0, 1, 0, 1, 1},
    {"csr_slots", "SGA_NO_CSR_SLOTS", 1, 0, 1, 0, 1, 2},
    {"half_integer_table", "SGA_NO_HALF_TABLE", 1, 0, 1, 0, 1, 2},
    {"force_csr_acc", "SGA_FORCE_CSR_ACC", 0, 0, 0, 0, 3, 2},
    {"force_dense_canonical", "SGA_FORCE_DENSE_CANON", 1, 1, 0, 0, 1, 2},
    {"zero_slot_every", "SGA_ZERO_SLOT_EVERY", 0, 0, 0, 0, 1ll << 21, 2},
    {"replica_routing", "SGA_NO_REPLICA_ROUTING", 1, 0, 1, 0, 1, 0},
    {"fields_scratch_mb", "SGA_FIELDS_SCRATCH_MB", 0, 0, 256, 1, 65536, 0},
    {"clf_batched", "SGA_CLF_BATCHED", 0, 0, 2, 0, 2, 0},
    {"clf_tail_waves", "SGA_NO_CLF_TAIL_WAVES", 1, 0, 1, 0, 1, 0},
};
inline int find_option(const char *key) {
    if (!key) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(key, OPT_DEFS[i].key) == 0) return i;
    return -1;
}
static_assert(OPT_COUNT <= SGA_ROUTE_MAX_OPTS, "sga_route_query::opt holds every option");

}  // namespace sga_impl

namespace sga_route {

using Query = sga_route_query;

// ---- set time ---------------------------------------------------------------------------------------------------
// A sparse matrix handed over dense (sga_set_dense, SGA_J_AUTO) is taken as CSR: asked before the rows are counted ...
bool sparse_route_wanted(int storage_requested, int n_models, int n, bool j_integer, int field_cache, bool clf_problem,
                         long long opt_sparse_route);
// ... and decided once they are (longest row, total entries)
bool sparse_route_taken(long long longest_row, long long total_entries);
// CSR rows padded to whole 64-entry slots at set time (the problems that run the wide forms)
bool csr_slots_at_set(long long nnz, int n, long long opt_csr_slots);

// ---- dense geometry (latched with the replicas; re-evaluated when the tuning changes) -----------------------------
struct DenseGeometry {
    int waves = 0, cpw = 0;        // what sga_describe reports (bit-plane form: the int8 fallback's 8 x (Wb * Cb))
    int waves_t2 = 0, cpw_t2 = 0;  // bit-plane form
    long long ld = 0;              // spins per replica (whole chunks)
    bool fits = true;              // replica spins (+ accept table) fit LDS
};
bool choose_geometry(int n, int epc, int R, int forced_waves, int &W, int &CPW, int max_cpw, int unit);
DenseGeometry dense_geometry(const Query &q);
long long dense_ldj(const Query &q);  // row stride of the packed couplings, elements

// ---- CSR forms ------------------------------------------------------------------------------------------------------
bool csr_rows_medium(const Query &q);       // rows of 65 ... 256 entries on the four-updates-per-step form
int csr_updates_per_step(const Query &q);   // 0 | 1 | 2 | 4 | 8
struct CsrForm {
    bool bits = false;        // spins as bits in LDS
    int big_form = 0;         // 0 int8 spins | 1 bits, one replica per workgroup (64-bit extents, slots) | 2 bits, narrow
    int waves = 1;            // waves per replica
    int sstride = 0;
    int table_m = 0;          // (dropped to 0 where the table does not fit LDS)
    bool needs_slots = false; // the form addresses rows by 64-entry slots
    bool wants_packed = false;// one dword per entry where the values allow
    const char *error = nullptr;  // the problem does not fit any form
};
CsrForm csr_replica_form(const Query &q);
int csr_replicas_per_block(const Query &q, const CsrForm &f);
// the kernel family launch_sweep_csr takes for production arguments: "rows" | "narrow" | "narrow-bits" | "wide-bits" | "wide-bytes"
const char *csr_kernel_family(const Query &q, const CsrForm &f, bool slotted_now);

// ---- TSP-structured couplings --------------------------------------------------------------------------------------
struct TspForm {
    int waves = 1, passes = 1;
};
TspForm tsp_form(int npad, int tune_waves);

// ---- cached local fields (sga_set_field_cache) ------------------------------------------------------------------------
// nullptr when the cached-field sweep can serve q (with q.sstride / q.ldj as laid out), else the reason
const char *clf_refusal(const Query &q);
double routing_theta(const Query &q);       // break-even acceptance of one replica between the two kernel families
bool auto_starts_cached(const Query &q);    // SGA_FIELD_CACHE_AUTO before any acceptance is known
int clf_csr_waves(const Query &q);          // waves per replica of the cached-field sweep over CSR couplings
// ---- sweep time ------------------------------------------------------------------------------------------------------
int sweeps_per_launch(const Query &q, int n_sweeps, int tune_spl, int npad_tsp);

std::string explain(const Query &q);

}  // namespace sga_route

#endif  // SGA_ROUTE_H
