// sga_problem.cpp -- the problem side of the C ABI (include/sga.h): sga_set_dense / sga_set_dense_batch, sga_set_csr /
// sga_set_csr64, sga_set_tsp.  Value and structure scans on the device, the packed layouts the kernels read, CSR row
// layouts (plain / 64-entry slots / packed entries).  Which form a problem then runs in: sga_route.cpp.
#include "sga_engine_impl.h"

namespace {

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

// Row extents of the layout the kernels read: dst[i] = prefix sum of the rows' lengths, each rounded
// up to whole 64-entry slots when `slotted`.  n <= ~1.3e6 rows: done on the host at set time.
//
// Slotted layouts also get the wide forms' per-row record (rowinfo: first slot, slot count | entries in
// the last slot << 24, slots from the first slot to an all-zero slot, h): a wave asks for a fixed number of slots per row and
// the ones past the row's end read that zero slot (value 0: nothing to mask when the row is
// summed).  The zero slot is the 64 zeroed entries behind the array; layouts beyond 2^21 slots
// (1 GB) get one more inside after every 2^21 slots -- it rides at the end of the row before it,
// like slot padding -- so that the offset always fits the 32-bit lane offset of a load.
int build_layout(sga_engine *e, const std::vector<long long> &src, bool slotted) {
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

}  // namespace

namespace sga_impl {

// several updates per step (sweep_csr_rows.hip): which problems, how many -- sga_route.cpp
bool csr_rows_medium(const sga_engine *e) { return sga_route::csr_rows_medium(route_query_of(e)); }
int csr_updates_per_step(const sga_engine *e) { return sga_route::csr_updates_per_step(route_query_of(e)); }

// The wide sweep forms (a row dealt to several waves) address rows by 64-entry slots: re-pad an
// unpadded layout on demand (short-row problems run wide only when tuning asks for it).
int ensure_slotted(sga_engine *e) {
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
int ensure_packed_entries(sga_engine *e) {
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

}  // namespace sga_impl

namespace {

// CSR problem from 32- or 64-bit row extents (host or device pointers).  The structure is
// checked on the device -- a bad extent or column would fault in the sweep kernels -- and the
// same pass classifies the problem: integer valued (accept table, fp32-exact row sums),
// symmetric with zero diagonal (dE of the rule == energy change).  Device arrays are read where
// they lie, host arrays are staged; only the interleaved layout stays resident.
int set_csr_common(sga_engine *e, const void *rowptr, bool wide_extents, const int32_t *colidx,
                          const float *val, const float *h, int n, int64_t nnz) {
    if (!e) return fail(SGA_ERR_INVALID, "engine is NULL");
    if (!rowptr || !h || n <= 0 || nnz < 0 || (nnz > 0 && (!colidx || !val)))
        return fail(SGA_ERR_INVALID, "bad CSR problem arguments");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->free_replicas();
    e->free_problem();
    e->opt_stale = 0;
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
    e->csr_row_abs_max = m;
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
    int rc = he == hipSuccess ? build_layout(e, src, sga_route::csr_slots_at_set(nnz, n, e->opt[OPT_CSR_SLOTS]))
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

// Sparse couplings handed over as a dense matrix (the reference's IsingModel is dense by default; its assignment
// and scheduling encoders fill 1-2 % of it): with SGA_J_AUTO, one model, n >= 4096, integer-valued J and no row of
// more than 256 non-zeros the problem is taken as CSR -- a proposal then reads its row's entries instead of n
// couplings, and the several-updates-per-step forms apply (sweep_csr_rows.hip).  Integer row sums are exact in
// either form, so the chain is the dense forms' bit for bit.  When: see the call (the cached-field sweep is a dense
// form); never with option "sparse_route" = 0 (A/B switch).
// Returns SGA_OK with *taken = true when the problem was set as CSR.
int route_sparse_dense(sga_engine *e, const float *src, long long ld_src, const float *h, int n, bool *taken) {
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
    if (!sga_route::sparse_route_taken(longest, total)) return SGA_OK;
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

}  // namespace

extern "C" {

int sga_set_dense(sga_engine *e, const float *J, int64_t ldJ, const float *h, int n, int storage) {
    return sga_set_dense_batch(e, J, ldJ, h, n, 1, storage);
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
    e->opt_stale = 0;
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
    if (sga_route::sparse_route_wanted(storage, n_models, n, (nonint & 1u) == 0u, e->field_cache, e->clf_problem,
                                       e->opt[OPT_SPARSE_ROUTE])) {
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
    e->opt_stale = 0;
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

}  // extern "C"
