// sga_route.cpp -- WHICH kernel form sweeps a problem (sga_route.h).  Everything here is arithmetic on a
// sga_route_query: no device call, no engine state.  The thresholds are measurements (profiles/r0*_experiments.md,
// cited where they stand); tests/test_host_logic.py pins the answers for the five BASELINE configs and the fuzz
// shapes, so an edit that reroutes one of them fails on the CPU.
// (No reference counterpart: the reference has one code path, core/spin_dynamics.py:61-152.)
#include "sga_route.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "sga_kernels.h"

namespace sga_route {

using namespace sga_impl;

namespace {
constexpr long long CSR_TAIL_PAD = 256;       // zeroed entries behind the CSR entry array (sga_engine_impl.h)
constexpr int T2_ELEMS_PER_CHUNK = 8192;      // 1 KiB of one bit-plane
int elems_per_chunk(bool i8) { return i8 ? 1024 : 256; }
bool is_i8(const Query &q) { return q.storage == SGA_J_I8 || q.storage == SGA_J_T2; }
}  // namespace

// ---- set time -------------------------------------------------------------------------------------------------------
// Sparse couplings handed over as a dense matrix (the reference's IsingModel is dense by default; its assignment
// and scheduling encoders fill 1-2 % of it): with SGA_J_AUTO, one model, n >= 4096, integer-valued J and no row of
// more than 256 non-zeros the problem is taken as CSR -- a proposal then reads its row's entries instead of n
// couplings, and the several-updates-per-step forms apply (sweep_csr_rows.hip).  Taken when the caller asked for one
// row read per proposal (field cache OFF), or left the choice (AUTO) on a problem the cached-field sweep cannot
// serve: where that sweep applies it is the better form while few proposals are accepted (C2b, 1024 replicas,
// acceptance 2 %: dense int8 rows 7.7e8, as CSR four updates per step 4.3e9, cached fields 1.06e10 attempts/s).
bool sparse_route_wanted(int storage_requested, int n_models, int n, bool j_integer, int field_cache, bool clf_problem,
                         long long opt_sparse_route) {
    return storage_requested == SGA_J_AUTO && n_models == 1 && n >= 4096 && j_integer &&
           (field_cache == SGA_FIELD_CACHE_OFF || (field_cache == SGA_FIELD_CACHE_AUTO && !clf_problem)) &&
           opt_sparse_route != 0;
}
bool sparse_route_taken(long long longest_row, long long total_entries) {
    return !(longest_row > 256 || total_entries == 0 || total_entries >= (long long)INT32_MAX);
}
// Long rows (mean degree >= 192: the problems that run the wide forms) are padded to whole 64-entry slots.
bool csr_slots_at_set(long long nnz, int n, long long opt_csr_slots) {
    return (double)nnz / std::max(n, 1) >= 192.0 && opt_csr_slots != 0;
}

// ---- dense geometry ---------------------------------------------------------------------------------------------------
// Pick waves-per-replica W and chunks-per-wave CPW for a dense row of C chunks.  Measured on
// MI355X at n = 10^4, R = 1024 (profiles/r01_geometry_sweep.md): full occupancy (R*W ~ 32
// waves per CU) is best as long as every wave keeps >= 4 KiB of the row in flight and W
// balances the four SIMDs; row padding is paid on every read, so it dominates the cost.
bool choose_geometry(int n, int epc, int R, int forced_waves, int &W, int &CPW, int max_cpw, int unit) {
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

long long dense_ldj(const Query &q) {
    if (q.ldj > 0) return q.ldj;
    const long long elem = is_i8(q) ? 1 : 4;
    return ((long long)q.n * elem + 127) / 128 * 128 / elem;
}

// Launch geometry of the dense kernels for the replica count / tuning.  The packed matrices are laid out by n alone
// (at set time), so a change of geometry never touches them.
DenseGeometry dense_geometry(const Query &q) {
    DenseGeometry g;
    const int R = std::max(q.R_local, 1);
    if (q.storage == SGA_J_T2) {
        // bit-plane geometry first; the int8 layout (energy / single-site kernels, non-LEAN sweeps) shares its row
        // length: 8 waves x (Wb * CPWb) chunks of 1024 int8.  At most 4 chunks per wave: a bit-plane chunk costs
        // 8 VGPRs per ring slot
        int Wb, Cb;
        choose_geometry(q.n, T2_ELEMS_PER_CHUNK, R, q.tune_waves, Wb, Cb, sga::T2_MAX_CPW, 1);
        g.ld = (long long)Wb * Cb * T2_ELEMS_PER_CHUNK;
        g.waves = 8;
        g.cpw = Wb * Cb;
        g.waves_t2 = Wb;
        g.cpw_t2 = Cb;
        g.fits = sga::sweep_dense_lds_bytes(g.ld, q.table_m, false) <= 160 * 1024;
    } else if (q.acc == 2) {
        // canonical summation order: a wave owns whole super-chunks of 4 chunks (1024 fp32 elements),
        // one or two of them in registers; longer rows take the streaming form on 16 waves
        int W, S;
        choose_geometry(q.n, 4 * elems_per_chunk(false), R, q.tune_waves, W, S, 2, 4);
        g.waves = W;
        g.cpw = 4 * S;
        g.ld = (long long)W * g.cpw * elems_per_chunk(false);
        g.fits = sga::sweep_dense_lds_bytes(g.ld, q.table_m, true) <= 160 * 1024;
    } else {
        const bool i8 = is_i8(q);
        int W, CPW;
        choose_geometry(q.n, elems_per_chunk(i8), R, q.tune_waves, W, CPW, sga::MAX_CPW, 1);
        g.waves = W;
        g.cpw = CPW;
        g.ld = (long long)W * CPW * elems_per_chunk(i8);
        g.fits = sga::sweep_dense_lds_bytes(g.ld, q.table_m, false) <= 160 * 1024;
    }
    return g;
}

// ---- CSR forms ----------------------------------------------------------------------------------------------------------
// Narrow CSR forms of integer problems whose longest row has <= 64 entries: 0 = one update at a time,
// 1 | 2 = the pair look-ahead (opt-in, option "csr_updates_per_step": round 3 measured -1 ... +3 % on BASELINE configs[2]),
// 4 | 8 = that many updates per step, one per row of 16 | 8 lanes (sweep_csr_rows.hip; the launcher takes it
// for production arguments -- Philox sites, Metropolis with the accept table): the default where it applies
// (profiles/r03_experiments.md 4b: C3, rows of up to 50 entries, 1.0e10 | 2.87e10 | 2.47e10 attempts/s for
// 1 | 4 | 8 updates per step; degree ~16: 1.0e10 | 3.7e10 | 5.1e10).  Option value 0 turns it off.
// Rows of 65 ... 256 entries (assignment / small scheduling problems: degree 100-250, cache resident, bound by the
// one-update chain): four per step with 8 | 16 entries per lane, integer problems with the accept table only.
bool csr_rows_medium(const Query &q) {
    return q.kind == SGA_ROUTE_CSR && q.max_row_len > 64 && q.max_row_len <= 256 && q.acc == sga::CSR_ACC_F32_TABLE &&
           q.table_m > 0;
}
int csr_updates_per_step(const Query &q) {
    if (q.kind != SGA_ROUTE_CSR || q.max_row_len > 256) return 0;
    const bool medium = q.max_row_len > 64;
    if (medium && !csr_rows_medium(q)) return 0;
    int v = q.max_row_len <= 32 ? 8 : 4;
    if (q.opt[OPT_CSR_UPDATES_PER_STEP] >= 0) v = (int)q.opt[OPT_CSR_UPDATES_PER_STEP];
    if (v != 1 && v != 2 && v != 4 && v != 8) return 0;
    if (medium) v = v >= 4 ? 4 : 0;  // (the pair look-ahead holds one wave-load per row)
    if (v >= 4 && (q.layout_entries + CSR_TAIL_PAD) * 8 >= (1ll << 32)) return 0;  // (32-bit byte offsets of the entries)
    return v;
}

CsrForm csr_replica_form(const Query &q) {
    CsrForm f;
    const int R_local = q.R_local;
    f.table_m = q.table_m;
    f.sstride = (q.n + 15) / 16 * 16;
    const double deg = (double)q.nnz / q.n;
    const int bits_stride = (q.n + 127) / 128 * 128;
    const bool bits_fit = sga::csr_big_fits(bits_stride, 0);
    // (rows the several-updates-per-step form covers are "short": one wave per replica, several replicas per workgroup)
    // -- while the structure is L2 resident or the replicas are few: beyond that the form is bound by the cache
    // fabric (8-byte entries), where the one-wave bit-spin form with packed 4-byte entries stays ahead (assignment
    // 100 x 100, degree 198, 20 MB: 1024 replicas 1.3e9 -> 3.7e9 attempts/s, 4096 replicas 6.3e9 -> 4.0e9)
    const bool rows_medium = csr_rows_medium(q) && csr_updates_per_step(q) >= 4 && q.tune_waves <= 1 &&
                             (q.layout_entries * 8 <= (6ll << 20) || R_local <= 1024);
    const bool long_rows = deg >= 192.0 && !rows_medium;
    // the bit-spin form that would be used: narrow (several replicas per workgroup, 32-bit
    // extents) on short rows, else one replica per workgroup with its row dealt to waves
    const int rpb_bits = (q.rowptr32 && !long_rows && q.tune_waves <= 1)
                             ? sga::csr_bits_waves_per_block(bits_stride, q.table_m) : 0;
    const bool narrow_bits = rpb_bits >= 2;
    // Spins as bits in LDS: beyond the int8 capacity or the 32-bit extents
    // (option "force_csr_bits": parity tests run the small cases through the same forms) ...
    bool bits = sga::csr_waves_per_block(f.sstride, 0) < 1 || !q.rowptr32 || q.opt[OPT_FORCE_CSR_BITS] != 0;
    // ... or when the int8 spins fit, but not for all replicas at once: workgroups beyond the
    // LDS-resident set run as a second, mostly empty round (C4: 50 KB per replica = 3 per CU
    // = 768 of 1024 replicas resident, 4.7e8 attempts/s; as bits all are resident: 6.8e8)
    if (!bits && bits_fit && q.opt[OPT_CSR_BITS] != 0) {
        const bool wide_i8 = q.tune_waves > 1 || (q.tune_waves == 0 && long_rows && R_local <= 1024);
        const int rpb = wide_i8 ? 1 : std::max(1, sga::csr_waves_per_block(f.sstride, q.table_m));
        const long long budget = 160 * 1024 - 256;
        const long long wg_i8 = (long long)sga::csr_lds_bytes(f.sstride, q.table_m, false) * rpb;
        const long long one_bits = (long long)sga::csr_lds_bytes(bits_stride, q.table_m, true);
        const long long res_i8 = (long long)q.cus * rpb * std::min<long long>(8, budget / wg_i8);
        const long long res_bits =
            narrow_bits ? (long long)q.cus * rpb_bits * std::min<long long>(8, budget / (one_bits * rpb_bits))
                        : (long long)q.cus * std::min<long long>(16, budget / one_bits);
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
    const bool many_long = long_rows && R_local > 1024 && q.tune_waves == 0 && bits_fit && q.slotted &&
                           q.opt[OPT_CSR_BITS] != 0;
    if (many_long) bits = true;
    f.bits = bits;
    f.big_form = !bits ? 0 : (narrow_bits ? 2 : 1);
    if (bits) {
        f.sstride = bits_stride;
        if (!bits_fit) {
            f.error = "CSR problem too large for the LDS-resident spins";
            return f;
        }
        if (!sga::csr_big_fits(f.sstride, f.table_m)) f.table_m = 0;
        if (f.big_form == 2) {
            f.waves = 1;
        } else {
            // one workgroup per replica: deal a long row to as many waves as the 8 entries per
            // lane requested ahead need to cover it (profiles/r01_experiments.md: 500 cities,
            // degree 1996: 4 waves; 1000 cities, 3996: 8)
            const int wpr = q.tune_waves > 0 ? q.tune_waves : (many_long ? 1 : (int)std::ceil(deg / 512.0));
            f.waves = std::max(1, std::min(wpr, 8));
        }
    } else {
        if (sga::csr_waves_per_block(f.sstride, f.table_m) < 1)
            f.table_m = 0;  // no room for the probability table: general path
        // Long rows AND too few replicas to give every SIMD a wave: deal each row to two waves
        // (one replica per workgroup).  The kernel is issue bound, so with >= 2048 replicas the
        // extra waves only repeat the per-update work (measured: C4, R = 1024: 1 / 2 / 4 / 8
        // waves -> 2.98 / 3.69 / 3.67 / 3.34 e8 attempts/s; C5, R = 2048: 1.55 vs 1.19 e9).
        const int wpr = q.tune_waves > 0 ? q.tune_waves : ((long_rows && R_local <= 1024) ? 2 : 1);
        f.waves = std::min(wpr, 8);
    }
    // the wide builds exist for 1, 2, 4 and 8 waves per replica (slot arithmetic on constants; the
    // canonical summation order of real-valued rows is defined on that grid)
    {
        int p2 = 1;
        while (p2 < f.waves) p2 *= 2;
        f.waves = std::min(p2, 8);
    }
    // one replica per workgroup (row dealt to its waves): rows are addressed by 64-entry slots
    f.needs_slots = f.waves > 1 || f.big_form == 1;
    f.wants_packed = f.big_form == 1 && q.storage != SGA_CSR_STORAGE_F32;
    return f;
}

int csr_replicas_per_block(const Query &q, const CsrForm &f) {
    (void)q;
    return f.big_form == 2 ? sga::csr_bits_waves_per_block(f.sstride, f.table_m)
                           : ((f.waves > 1 || f.bits) ? 1 : sga::csr_waves_per_block(f.sstride, f.table_m));
}

const char *csr_kernel_family(const Query &q, const CsrForm &f, bool slotted_now) {
    // the order of launch_sweep_csr (sweep_csr.hip) for production arguments: Philox sites, Metropolis, fp64, no traces
    sga::SweepArgs a{};
    a.rowptr = q.rowptr32 ? reinterpret_cast<const int32_t *>(8) : nullptr;  // (only tested against null)
    a.big = f.big_form;
    a.csr_row_cap = (q.max_row_len <= 256) ? (int)std::max<long long>(slotted_now ? (q.max_row_len + 63) / 64 * 64 : q.max_row_len, 1) : 0;
    a.csr_pair_ahead = csr_updates_per_step(q);
    a.csr_acc = q.acc;
    a.table_m = f.table_m;
    if (a.csr_acc == sga::CSR_ACC_F32_TABLE && a.table_m == 0) a.csr_acc = sga::CSR_ACC_F32;
    a.table_scale = q.table_scale > 0 ? q.table_scale : 1;
    a.site_mode = SGA_SITE_RANDOM;
    a.arith = SGA_ARITH_F64;
    a.rule = SGA_RULE_METROPOLIS;
    a.look_ahead = q.opt[OPT_LOOK_AHEAD] != 0 ? 1 : 0;
    a.force_general = q.opt[OPT_FORCE_GENERAL] != 0 ? 1 : 0;
    a.sstride = f.sstride;
    if (f.waves == 1 && sga::sweep_csr_rows_applies(a)) return "rows";
    if (f.bits) return (f.waves == 1 && q.rowptr32 && f.big_form == 2) ? "narrow-bits" : "wide-bits";
    return f.waves > 1 ? "wide-bytes" : "narrow";
}

// ---- TSP-structured couplings --------------------------------------------------------------------------------------------
// 256 cities per wave and pass.  Four or more waves at one pass: half the waves with two passes each do better (1000
// cities, same box: 4 x 1 711 ms, 2 x 2 677 ms, 1 x 4 701 ms per sweep -- fewer barrier participants against a longer
// row sum); tuning may ask otherwise.
TspForm tsp_form(int npad, int tune_waves) {
    TspForm t;
    const int full = npad / 256;  // waves at one pass
    int w = (full >= 4 && full % 2 == 0) ? full / 2 : full;
    if (tune_waves == full) w = full;
    if (tune_waves > 0 && tune_waves < full && full % tune_waves == 0 && (full / tune_waves == 2 || full / tune_waves == 4))
        w = tune_waves;
    t.waves = w;
    t.passes = full / w;
    return t;
}

// ---- cached local fields ----------------------------------------------------------------------------------------------------
// The cached-local-field sweep serves: dense integer-valued symmetric problems (one model) whose fields and spin bits
// fit LDS, and CSR problems with integer J in strictly sorted rows -- any single-site rule.
const char *clf_refusal(const Query &q) {
    if (q.kind == SGA_ROUTE_TSP) return "cached local fields: stored couplings only";
    if (q.kind == SGA_ROUTE_CSR) {
        // sparse couplings: the dynamic part of the fields as int16 in LDS (sweep_clf_csr.hip)
        const long long ldf = ((long long)q.n + 127) / 128 * 128;
        if (!q.clf_ok)
            return q.from_dense
                       ? "cached local fields: this sparse matrix was kept as CSR because the field cache was OFF when "
                         "sga_set_dense ran (its dense source is released), and as CSR it does not qualify (integer J in "
                         "strictly sorted rows, sum_j |J_ij| < 2^15, h in multiples of 1/2) -- call sga_set_field_cache "
                         "before sga_set_dense"
                       : "cached local fields over CSR couplings need integer-valued symmetric J in strictly sorted rows "
                         "(no duplicates), zero diagonal, max_i sum_j |J_ij| < 2^15 and h in multiples of 1/2";
        if (q.R_local > 0 && (sga::sweep_clf_csr_lds_bytes(ldf, q.sstride, q.table_m) > 160 * 1024 ||
                              (q.slotted ? (q.max_row_len + 63) / 64 * 64 : q.max_row_len) > 4 * 64 * 8))
            return "cached local fields: fields and spins of a replica do not fit LDS (or a row is longer than 2048 entries)";
        return nullptr;
    }
    if (!q.clf_ok)
        return "cached local fields need one model with integer-valued symmetric J, zero diagonal, h in "
               "multiples of 1/2 and row sums below 2^24";
    if (q.R_local > 0 && sga::sweep_clf_lds_bytes((dense_ldj(q) + 127) / 128 * 128, q.clf_bits, q.sstride,
                                                  q.clf_scale == 2 ? 2048 : q.table_m) > 160 * 1024)
        return "cached local fields: fields and spins of a replica do not fit LDS";
    return nullptr;
}

// Break-even acceptance of ONE replica = (what an update costs its chain on the row kernel) / (what an accept costs it
// on the cached-field kernel).  Both kernels are paced by a replica's serial chain, not by the chip, whenever only
// part of the replicas is hot: ~1.5 us per accept (1.15 alone on its CU ... 1.7 with busy neighbours), and per update
// 0.38 us on bit-planes / 0.58 us on int8 rows at n = 10^4, ~0.3 us on short rows (profiles/r04_routing.py; fp32
// rows: estimate).
double routing_theta(const Query &q) {
    const double kn = (double)q.n / 1000.0;
    const double t_upd = q.kind == SGA_ROUTE_CSR ? 0.20 + 0.0008 * (double)q.nnz / (double)q.n  // (C4: 0.68, C2b as CSR: 0.36)
                         : q.storage == SGA_J_T2 ? 0.29 + 0.009 * kn
                                                 : (q.storage == SGA_J_I8 ? 0.27 + 0.031 * kn : 0.30 + 0.12 * kn);
    return t_upd / 1.5;
}
// AUTO, nothing known yet: the run starts on the kernel that loses least if the guess is wrong: the cached-field
// kernel where a replica would have to accept more than ~30 % of its proposals for the row kernels to win (int8 and
// fp32 rows at n = 10^4: the first four sweeps of the bench ladder 55 / 218 ms on the row kernels against 8 / 30 ms
// cached, and 36 against 56 / 208 ms on a ladder that stays hot), the row-per-proposal kernel otherwise (bit-planes,
// small n, CSR).
bool auto_starts_cached(const Query &q) { return q.kind != SGA_ROUTE_CSR && 0.8 * routing_theta(q) >= 0.3; }

int clf_csr_waves(const Query &q) {
    const long long row_max = std::min<long long>(q.slotted ? (q.max_row_len + 63) / 64 * 64 : q.max_row_len, 1 << 20);
    return (row_max > 256 || q.n > 20000) ? 8 : 4;
}

// sweeps per launch: aim for ~50 ms of estimated work per launch
int sweeps_per_launch(const Query &q, int n_sweeps, int tune_spl, int npad_tsp) {
    int spl = tune_spl;
    if (spl <= 0) {
        const double row_bytes = q.kind == SGA_ROUTE_TSP ? 8.0 * npad_tsp
                                 : q.kind == SGA_ROUTE_CSR ? 264.0 : (double)dense_ldj(q) * (is_i8(q) ? 1 : 4);
        const double per_update = std::max(row_bytes * q.R_local / 4.0e12, 1.0e-6);
        const double per_sweep = per_update * q.n;
        spl = (int)std::min<double>(n_sweeps, std::max(1.0, std::floor(0.05 / per_sweep)));
    }
    return std::max(1, std::min(spl, n_sweeps));
}

// ---- the whole chain of decisions as one line ----------------------------------------------------------------------------
std::string explain(const Query &q0) {
    Query q = q0;
    char buf[640];
    std::string out;
    if (q.kind == SGA_ROUTE_TSP) {
        const int npad = 256 * ((q.n_cities + 255) / 256);
        const TspForm t = tsp_form(npad, q.tune_waves);
        std::snprintf(buf, sizeof(buf), "tsp n_cities=%d waves=%d passes=%d updates_per_step_option=%lld kernel=%s", q.n_cities,
                      t.waves, t.passes, (long long)q.opt[OPT_TSP_PARALLEL],
                      q.opt[OPT_TSP_PARALLEL] == 0 ? "sweep_tsp_kernel" : "sweep_tsp_par_kernel|sweep_tsp_kernel");
        out = buf;
    } else if (q.kind == SGA_ROUTE_CSR) {
        const CsrForm f = csr_replica_form(q);
        if (f.error) return std::string("csr error=") + f.error;
        const bool slotted_now = q.slotted || f.needs_slots;
        if (slotted_now && !q.slotted) {  // ensure_slotted re-pads the layout: at most 63 entries more per row
            q.layout_entries = q.layout_entries + 63ll * q.n;
            q.slotted = 1;
        }
        const char *fam = csr_kernel_family(q, f, slotted_now);
        const int ups = csr_updates_per_step(q);
        std::snprintf(buf, sizeof(buf),
                      "csr form=%s spins=%s waves=%d replicas_per_block=%d updates_per_step=%d slots=%d entries=%s table_m=%d "
                      "sstride=%d",
                      fam, f.bits ? "bits" : "int8", f.waves, csr_replicas_per_block(q, f),
                      std::strcmp(fam, "rows") == 0 ? ups : ((ups == 1 || ups == 2) && !f.bits && f.waves == 1 ? ups : 0),
                      slotted_now ? 1 : 0, (f.wants_packed && q.packed_ok) ? "packed" : "cv", f.table_m, f.sstride);
        out = buf;
        q.sstride = f.sstride;
    } else {
        const DenseGeometry g = dense_geometry(q);
        if (!g.fits) return "dense error=replica spins do not fit LDS (n too large)";
        const bool t2 = q.storage == SGA_J_T2, i8 = is_i8(q);
        const int W = t2 ? g.waves_t2 : g.waves, C = t2 ? g.cpw_t2 : g.cpw;
        const bool streaming = t2 ? C > sga::T2_MAX_CPW : C > sga::MAX_CPW;
        const int la = (q.table_m > 0 && q.opt[OPT_LOOK_AHEAD] != 0)
                           ? sga::dense_look_ahead(t2, i8, q.acc != 0 && !i8, C, W, std::max(q.R_local, 1)) : 1;
        std::snprintf(buf, sizeof(buf), "dense storage=%s acc=%s waves=%d chunks_per_wave=%d%s ld=%lld look_ahead=%d kernel=%s",
                      t2 ? "t2" : (i8 ? "i8" : "f32"), i8 ? "i32" : (q.acc == 0 ? "f32" : (q.acc == 2 ? "f64-canonical" : "f64-exact")),
                      W, C, streaming ? "(streaming)" : "", g.ld, la, t2 ? "sweep_dense_t2_kernel" : "sweep_dense_kernel");
        out = buf;
        q.sstride = (int)g.ld;
    }
    // what sga_set_field_cache would run
    if (q.field_cache == SGA_FIELD_CACHE_OFF) {
        out += " cached=off";
    } else {
        const char *why = clf_refusal(q);
        if (why) {
            out += q.field_cache == SGA_FIELD_CACHE_ON ? " cached=refused" : " cached=unavailable";
        } else if (q.field_cache == SGA_FIELD_CACHE_ON) {
            if (q.kind == SGA_ROUTE_CSR) std::snprintf(buf, sizeof(buf), " cached=on(waves=%d)", clf_csr_waves(q));
            else
                std::snprintf(buf, sizeof(buf), " cached=on(waves=%d fields=int%d)",
                              sga::sweep_clf_waves(dense_ldj(q), is_i8(q), std::max(q.R_local, 1), q.cus, (int)q.opt[OPT_CLF_WAVES]),
                              q.clf_bits);
            out += buf;
        } else {
            std::snprintf(buf, sizeof(buf), " cached=auto(start=%s theta=%.3f)", auto_starts_cached(q) ? "cached" : "rows",
                          routing_theta(q));
            out += buf;
        }
    }
    return out;
}

}  // namespace sga_route

extern "C" {

int sga_route_query_init(sga_route_query *q) {
    if (!q) return sga_impl::fail(SGA_ERR_INVALID, "query is NULL");
    std::memset(q, 0, sizeof(*q));
    q->n_models = 1;
    q->cus = 256;
    q->table_scale = 1;
    q->clf_bits = 16;
    q->clf_scale = 1;
    q->rowptr32 = 1;
    for (int i = 0; i < sga_impl::OPT_COUNT; ++i) q->opt[i] = sga_impl::OPT_DEFS[i].def;
    return SGA_OK;
}

int sga_explain_route(const sga_route_query *q, char *buf, int buflen) {
    if (!q || !buf || buflen <= 0) return sga_impl::fail(SGA_ERR_INVALID, "bad arguments");
    if (q->n <= 0 || q->kind < SGA_ROUTE_DENSE || q->kind > SGA_ROUTE_TSP || (q->kind == SGA_ROUTE_TSP && q->n_cities < 3))
        return sga_impl::fail(SGA_ERR_INVALID, "route query: kind / n out of range");
    std::snprintf(buf, (size_t)buflen, "%s", sga_route::explain(*q).c_str());
    return SGA_OK;
}

}  // extern "C"
