// sga_state.cpp -- state access, checkpoint / resume, measurement and description entry points of the C ABI
// (include/sga.h): nothing here launches a sweep.
#include "sga_engine_impl.h"

extern "C" {

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

int sga_get_route_query(sga_engine *e, sga_route_query *out) {
    if (!e || !out) return fail(SGA_ERR_INVALID, "NULL argument");
    if (e->n <= 0) return fail(SGA_ERR_INVALID, "no couplings set");
    *out = route_query_of(e);
    return SGA_OK;
}

int sga_get_last_kernel(sga_engine *e, char *buf, int buflen) {
    if (!e || !buf || buflen <= 0) return fail(SGA_ERR_INVALID, "bad arguments");
    std::snprintf(buf, (size_t)buflen, "%s", e->last_kernel);
    return SGA_OK;
}

int sga_last_kernel(char *buf, int buflen) {
    if (!buf || buflen <= 0) return fail(SGA_ERR_INVALID, "bad arguments");
    std::snprintf(buf, (size_t)buflen, "%s", sga::last_sweep_kernel());
    return SGA_OK;
}

}  // extern "C"
