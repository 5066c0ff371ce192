// sga_engine.cpp -- host side of the C ABI declared in include/sga.h: owns the HBM buffers
// (packed couplings, replica spins / energies / bests, ladder state), picks the launch
// geometry, and drives the HIP kernels.  No torch, no exceptions across the ABI.
// (problem set-up: sga_problem.cpp; autotune: sga_autotune.cpp; state access / describe: sga_state.cpp; form selection:
// sga_route.cpp; shared internals: sga_engine_impl.h)
#include "sga_engine_impl.h"

namespace sga_impl {

sga_route_query route_query_of(const sga_engine *e) {
    sga_route_query q;
    (void)sga_route_query_init(&q);
    q.kind = e->tsp ? SGA_ROUTE_TSP : (e->csr ? SGA_ROUTE_CSR : SGA_ROUTE_DENSE);
    q.n = e->n;
    q.n_models = e->n_models;
    q.R_local = e->R;
    q.cus = e->cus;
    q.tune_waves = e->tune_waves;
    q.field_cache = e->field_cache;
    if (e->csr) {
        q.storage = e->csr_storage;
        q.acc = e->csr_acc;
        q.clf_ok = e->clf_csr_problem ? 1 : 0;
    } else {
        q.storage = e->use_t2 ? SGA_J_T2 : (e->want_i8 ? SGA_J_I8 : SGA_J_F32);
        q.acc = e->acc64 ? (e->acc_canon ? 2 : 1) : 0;
        q.clf_ok = e->clf_problem ? 1 : 0;
    }
    q.table_m = e->table_m;
    q.table_scale = e->table_scale;
    q.clf_bits = e->clf_bits;
    q.clf_scale = e->clf_scale;
    q.from_dense = e->from_dense ? 1 : 0;
    q.nnz = e->nnz;
    q.max_row_len = e->max_row_len;
    q.layout_entries = e->layout_entries;
    q.slotted = e->slotted ? 1 : 0;
    q.rowptr32 = e->rowptr ? 1 : 0;
    q.packed_ok = e->cvp ? 1 : 0;
    q.n_cities = e->tsp ? e->tsp_args.n_cities : 0;
    q.sstride = e->sstride;
    q.ldj = e->ldj;
    for (int i = 0; i < OPT_COUNT; ++i) q.opt[i] = e->opt[i];
    return q;
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

// The cached-local-field sweep serves this problem / these replicas?  why: the reason when it does not
// (sga_route.cpp, clf_refusal).
bool clf_possible(const sga_engine *e, const char **why) {
    const sga_route_query q = route_query_of(e);
    const char *reason = sga_route::clf_refusal(q);
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

// Launch geometry of the dense kernels for the current replica count / tuning (sga_route.cpp, dense_geometry).  The
// packed matrices are laid out by n alone (pack_dense, at set time), so a change of geometry never touches them.
int ensure_packed(sga_engine *e) {
    if (e->csr || e->tsp) return SGA_OK;
    if (!e->J_packed) return fail(SGA_ERR_INVALID, "no couplings set");
    const sga_route::DenseGeometry g = sga_route::dense_geometry(route_query_of(e));
    if (e->use_t2 ? (e->ld == g.ld && e->waves_t2 == g.waves_t2 && e->cpw_t2 == g.cpw_t2)
                  : (e->waves == g.waves && e->cpw == g.cpw && e->ld == g.ld))
        return SGA_OK;
    if (!g.fits) return fail(SGA_ERR_UNSUPPORTED, "replica spins do not fit LDS (n too large)");
    if (e->use_t2) {
        e->waves_t2 = g.waves_t2;
        e->cpw_t2 = g.cpw_t2;
    }
    e->waves = g.waves;
    e->cpw = g.cpw;
    e->ld = g.ld;
    return SGA_OK;
}

}  // namespace sga_impl

extern "C" {

const char *sga_last_error(void) { return g_last_error.c_str(); }
int sga_version(void) { return 500; }  // round 5: + sga_explain_route / sga_get_route_query, sga_get_last_kernel, sga_get_autotune_table, ladder-local sga_exchange

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
    if (e->opt[i] != (long long)value) {
        // where is this value read?  (include/sga.h: [set] | [init] | [sweep])
        if (d.stage == 2 && e->n > 0) e->opt_stale |= 2, e->opt_stale_key = d.key;
        if (d.stage == 1 && e->R > 0) e->opt_stale |= 1, e->opt_stale_key = d.key;
    }
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
    if (mode != e->field_cache) {
        // what an earlier mode learnt about these replicas does not carry over: ON runs every replica on the cached-field
        // kernel (AUTO's per-replica routes would leave some on the row kernels for good), a failed allocation under
        // AUTO is retried, the fields are seeded anew
        e->route.clear();
        e->auto_mark_acc.clear();
        e->n_route_clf = 0;
        e->clf_wide = false;
        e->clf_hot = true;
        e->auto_unavailable = false;
        e->auto_mark_attempted = 0;
        e->auto_interval = 4;
        e->route_dirty = true;
        e->fields_valid = false;
    }
    e->field_cache = mode;
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
    e->opt_stale &= ~1;
    e->R = R_local;
    e->Rg = R_global;
    e->replica0 = replica0;
    e->seed = seed;
    e->sweeps_done = 0;
    e->rounds = 0;
    e->attempted = 0;
    // WHICH form the replicas are laid out for: sga_route.cpp (pure functions of the problem's traits, the replica count,
    // the tuning and the options; tests/test_host_logic.py pins them)
    if (e->tsp) {
        e->sstride = (e->n + 15) / 16 * 16;
        const sga_route::TspForm t = sga_route::tsp_form(e->tsp_args.npad, e->tune_waves);
        e->tsp_waves = t.waves;
        e->tsp_passes = t.passes;
        e->waves = e->tsp_waves;
        e->cpw = 0;
    } else if (!e->csr) {
        int rc = ensure_packed(e);  // geometry depends on the replica count
        if (rc != SGA_OK) return rc;
        e->sstride = (int)e->ld;
    } else {
        const sga_route::CsrForm f = sga_route::csr_replica_form(route_query_of(e));
        if (f.error) return fail(SGA_ERR_UNSUPPORTED, f.error);
        e->big = f.bits;
        e->big_form = f.big_form;
        e->sstride = f.sstride;
        e->table_m = f.table_m;
        e->waves = f.waves;
        e->cpw = 0;
        if (f.needs_slots) {
            int rc = ensure_slotted(e);
            if (rc != SGA_OK) return rc;
        }
        e->csr_storage_latched = e->csr_storage;
        if (f.wants_packed) {
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
    if (e->opt_stale)
        return fail(SGA_ERR_INVALID, std::string("option \"") + (e->opt_stale_key ? e->opt_stale_key : "?") + "\" changed after " +
                                         ((e->opt_stale & 2) ? "the couplings were set (it is read by sga_set_dense / sga_set_csr): set them again"
                                                             : "sga_init_replicas (it is read there): initialise the replicas again"));
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
        // (only where the counters ARE looked at: AUTO, and ON over dense couplings with the tail / batched forms
        //  enabled -- ON over CSR couplings has one form: its call stays one piece, launches of up to 256 sweeps)
        const bool looks = e->field_cache == SGA_FIELD_CACHE_AUTO ||
                           (!e->csr && ((e->opt[OPT_CLF_TAIL_WAVES] != 0 && e->opt[OPT_CLF_WAVES] == 0) || e->opt[OPT_CLF_BATCHED] == 2));
        if (looks && clf_possible(e, &why)) {
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

    // sweeps per launch: aim for ~50 ms of estimated work per launch (sga_route.cpp)
    const sga_route_query rq = route_query_of(e);
    int spl = sga_route::sweeps_per_launch(rq, n_sweeps, e->tune_spl, e->tsp ? e->tsp_args.npad : 0);
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
    // break-even acceptance of one replica between the two kernel families: sga_route.cpp
    auto routing_theta = [&]() -> double { return sga_route::routing_theta(rq); };
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
                const bool start_cached = !is_auto || sga_route::auto_starts_cached(rq);
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
        a.table_covers = (e->csr && a.table_m > 0 && (double)e->table_scale * (double)e->csr_row_abs_max <= (double)a.table_m) ? 1 : 0;
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
            const int cw = sga_route::clf_csr_waves(rq);
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
        std::snprintf(e->last_kernel, sizeof(e->last_kernel), "%s", sga::last_sweep_kernel());  // (this thread just launched it)
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

}  // extern "C"
