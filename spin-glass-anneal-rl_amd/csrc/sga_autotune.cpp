// sga_autotune.cpp -- sga_autotune (include/sga.h): the measured choice of the launch geometry (dense problems) or of
// the sweep form (CSR problems).  Every candidate runs the real sweep kernel on the real replicas; the state is put back.
#include "sga_engine_impl.h"

extern "C" {

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
    int64_t user_launches = 0;   // the caller's kernel-timing statistics survive the trials
    double user_ms = 0.0;
    (void)sga_get_kernel_time(e, &user_launches, &user_ms, 1);
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
    e->tune_table.clear();
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
        {
            char item[200];
            std::snprintf(item, sizeof(item), "%s%s=%.4f", e->tune_table.empty() ? "" : ";", seen[n_seen - 1], t / k);
            e->tune_table += item;
        }
        if (t / k < best * 0.99) {  // (ties within 1 % go to the earlier, simpler candidate: a fixed preference order)
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
    if (rc != SGA_OK && best_i >= 0) {
        // the winner's layout could not be set up again (memory): back to the caller's own form with the caller's state,
        // rather than leaving freshly seeded replicas behind
        (void)hipGetLastError();
        const int rc_user = layout(user_waves, user_ups);
        if (rc_user == SGA_OK) rc = SGA_OK, best_i = -1;
    }
    {
        int64_t l2 = 0;
        double t2 = 0.0;
        (void)sga_get_kernel_time(e, &l2, &t2, 1);  // drop the trials' events ...
        e->launches = user_launches;               // ... and put the caller's statistics back
        e->total_ms = user_ms;
    }
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
    int64_t user_launches = 0;   // the caller's kernel-timing statistics survive the trials
    double user_ms = 0.0;
    (void)sga_get_kernel_time(e, &user_launches, &user_ms, 1);
    e->tune_table.clear();
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
        {   // the candidate table (sga_get_autotune_table): "waves x chunks per wave = ms per sweep"
            char item[64];
            const int gw = e->use_t2 ? e->waves_t2 : e->waves, gc = e->use_t2 ? e->cpw_t2 : e->cpw;
            std::snprintf(item, sizeof(item), "%s%s%dx%d=%.4f", e->tune_table.empty() ? "" : ";", w == 0 ? "heuristic:" : "", gw, gc, per);
            e->tune_table += item;
        }
        if (per < best) {
            best = per;
            best_w = w;
        }
    }
    // Several geometries usually lie within the timing noise of each other (n = 10^4 fp32: 9 x 5, 13 x 4 and
    // 14 x 3 within 0.5 % on one box, and round 4's driver box and profile box disagreed on the winner): among those
    // within 1 % of the fastest take the one with the fewest waves -- a fixed preference order -- so that repeated runs,
    // other boxes and a profile taken later see the same instantiation.  The table goes out with the pick
    // (sga_get_autotune_table), so a reader sees what the tie cost.
    if (rc == SGA_OK && best_w >= 0)
        for (int w = 1; w <= sga::MAX_WAVES; ++w)
            if (per_w[w] <= best * 1.01) {
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
    {
        int64_t l2 = 0;
        double t2 = 0.0;
        (void)sga_get_kernel_time(e, &l2, &t2, 1);  // drop the trials' events ...
        e->launches = user_launches;               // ... and put the caller's statistics back
        e->total_ms = user_ms;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (rc != SGA_OK) return rc;
    if (final_rc != SGA_OK) return final_rc;
    if (best_ms_per_sweep) *best_ms_per_sweep = best;
    return SGA_OK;
}

int sga_get_autotune_table(sga_engine *e, char *buf, int buflen) {
    if (!e || !buf || buflen <= 0) return fail(SGA_ERR_INVALID, "bad arguments");
    std::snprintf(buf, (size_t)buflen, "%s", e->tune_table.c_str());
    return SGA_OK;
}

}  // extern "C"
