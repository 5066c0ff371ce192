// sweep_clf_csr.hip -- the cached-local-field sweep for SPARSE couplings (CSR): a coupling row is read only
// when a proposal is ACCEPTED (round 4; BASELINE configs[3]: the 50 000-spin scheduling instance, whose cold
// ladders reject 95 - 99 % of the proposals, and configs[1]'s assignment instance handed over sparse).
//
// Replaces the same reference code as sweep_csr_impl.h -- SpinDynamics.sweep / _metropolis_update
// (core/spin_dynamics.py:73-94,131-152) over IsingModel.get_local_field (core/ising_model.py:176-185) -- in the
// way the reference's incremental mode evaluates moves (core/energy_computer.py:166-173,262-265: dE from a
// maintained field, the field updated on a flip).  Same sites, uniforms and accept rule on exact integers: the
// chain is the row-per-proposal chain bit for bit (tests/test_cached_fields_gpu.py, against the oracle and the
// row-per-proposal CSR kernels).
//
// What is resident per replica (LDS): D_i = sum_j J_ij s_j as int16 -- the DYNAMIC part of the local field only:
// the penalty encodings' fields h_i reach 17 400 in steps of 1/2 at configs[3], far beyond 16 bits, but they
// never change; they are read (as integers scale * h_i, L2 resident) once per window with the candidates --,
// the spins as bits, the accept table.  n = 50 000: 100 KB + 6 KB + 8 KB: one replica per CU, eight waves.
// k = s_i (scale D_i + scale h_i), dE = 2 k / scale; an accept reads the row's (column, value) entries and
// moves D[column] by -2 J s_i with 16-bit LDS reads and writes (a row's columns are distinct: no two lanes
// meet; value-0 padding entries are skipped).
//
// Windows as in sweep_clf_impl.h: every wave evaluates its own 128 updates of a 128 W-update window, the waves
// meet in LDS slots, the earliest accept is applied by all, the rest is evaluated again; the entries of the
// PREDICTED next accept (the second accepting candidate) are requested together with the current row's.
//
// Byte model (its own, reported beside the graded one-row-per-proposal figure, never instead of it):
// B = acceptance rate x (deg x 8 + 8) bytes per attempt (last sentence).
#include "sweep_common.h"

namespace sga {

constexpr int CLFS_WINDOW = 128;     // updates a wave evaluates together: two per lane
constexpr int CLFS_MAX_WAVES = 8;
constexpr int CLFS_SLOT_INTS = 12;   // p, p2, site, site2 | k, s_old, len, len2 | beg lo, hi, beg2 lo, hi

__host__ __device__ constexpr long long clfs_bits_offset(long long ldf) { return (ldf * 2 + 15) & ~15ll; }
__host__ __device__ constexpr long long clfs_table_offset(long long ldf, int sstride) {
    return clfs_bits_offset(ldf) + (((sstride + 31) / 32 * 4 + 15) & ~15);
}
size_t sweep_clf_csr_lds_bytes(long long ldf, int sstride, int table_m) {
    return (size_t)clfs_table_offset(ldf, sstride) + sizeof(float) * (size_t)((table_m + 4) & ~3) +
           2 * 4 * CLFS_SLOT_INTS * CLFS_MAX_WAVES + 16;
}

// ---- seeding: D[r][i] = sum_j J_ij s_rj for eight replicas per pass over a slice of the rows -------------------
constexpr int CLFS_SEED_REPS = 8;
__global__ void __launch_bounds__(256) csr_fields_seed_kernel(const long long *__restrict__ rowptr, const int2 *__restrict__ cv,
                                                              const int8_t *__restrict__ spins, int sstride, int n, int R,
                                                              int slices, short *__restrict__ D, long long ldf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned int *sb = reinterpret_cast<unsigned int *>(smem);  // [8][words]: bit = spin down
    const int words = (n + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = blockIdx.x * CLFS_SEED_REPS;
    for (int q = tid; q < CLFS_SEED_REPS * words; q += 256) {
        const int rep = q / words, wd = q % words, r = r0 + rep;
        unsigned int b = 0;
        if (r < R)
            for (int t = 0; t < 32; ++t) {
                const int i = 32 * wd + t;
                if (i < n && spins[(long long)r * sstride + i] < 0) b |= 1u << t;
            }
        sb[q] = b;
    }
    __syncthreads();
    const int per = (n + slices - 1) / slices;
    const int i0 = blockIdx.y * per, i1 = min(n, i0 + per);
    for (int i = i0 + w; i < i1; i += 4) {
        const long long beg = rowptr[i], end = rowptr[i + 1];
        int acc[CLFS_SEED_REPS];
#pragma unroll
        for (int rep = 0; rep < CLFS_SEED_REPS; ++rep) acc[rep] = 0;
        for (long long e = beg + lane; e < end; e += 64) {
            const int2 ent = cv[e];
            const int J = (int)__int_as_float(ent.y);  // integer valued (engine: eligibility)
            const int wd = ent.x >> 5, bit = ent.x & 31;
#pragma unroll
            for (int rep = 0; rep < CLFS_SEED_REPS; ++rep) acc[rep] += ((sb[rep * words + wd] >> bit) & 1u) ? -J : J;
        }
#pragma unroll
        for (int rep = 0; rep < CLFS_SEED_REPS; ++rep) {
            const int tot = wave_sum(acc[rep]);
            if (lane == 0 && r0 + rep < R) D[(long long)(r0 + rep) * ldf + i] = (short)tot;
        }
    }
}
hipError_t launch_csr_fields_seed(const long long *rowptr, const int2 *cv, const int8_t *spins, int sstride, int n, int R,
                                  short *D, long long ldf, hipStream_t st) {
    const size_t lds = (size_t)CLFS_SEED_REPS * (size_t)((n + 31) / 32) * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(csr_fields_seed_kernel), lds);
    if (e != hipSuccess) return e;
    const int blocks = (R + CLFS_SEED_REPS - 1) / CLFS_SEED_REPS;
    const int slices = std::max(1, std::min(64, 2048 / std::max(blocks, 1)));
    hipLaunchKernelGGL(csr_fields_seed_kernel, dim3(blocks, slices), dim3(256), lds, st, rowptr, cv, spins, sstride, n, R,
                       slices, D, ldf);
    return hipGetLastError();
}
// hq[i] = scale * h_i as an integer (h is a multiple of 1 / scale: engine eligibility)
__global__ void scaled_fields_kernel(const float *h, int n, int scale, int *hq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) hq[i] = (int)__builtin_rintf((float)scale * h[i]);
}
hipError_t launch_scaled_fields(const float *h, int n, int scale, int *hq, hipStream_t st) {
    hipLaunchKernelGGL(scaled_fields_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h, n, scale, hq);
    return hipGetLastError();
}

// ---- the sweep ------------------------------------------------------------------------------------------------
// EPT: entries of a row per thread (the longest row <= EPT x threads of the workgroup)
template <int EPT>
__global__ void __launch_bounds__(64 * CLFS_MAX_WAVES) sweep_clf_csr_kernel(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    short *D = reinterpret_cast<short *>(smem);
    unsigned int *bits = reinterpret_cast<unsigned int *>(smem + clfs_bits_offset(a.ldf));
    float *ptab = reinterpret_cast<float *>(smem + clfs_table_offset(a.ldf, a.sstride));
    int *slots2 = reinterpret_cast<int *>(ptab + ((a.table_m + 4) & ~3));

    const int tid = threadIdx.x, lane = tid & 63;
    const int W = (int)(blockDim.x >> 6), nthreads = (int)blockDim.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = a.rep_list ? __builtin_amdgcn_readfirstlane(a.rep_list[blockIdx.x]) : (int)blockIdx.x, n = a.n;
    const int sc = a.table_scale;            // 1 | 2: k and the table are in units of 1 / scale
    const double inv_sc = 1.0 / (double)sc;  // exact
    const int *hq = a.clf_hq;
    const long long *rp = a.rowptr64;
    constexpr int NONE = 1 << 30;

    {   // resident state -> LDS
        const int4 *src = reinterpret_cast<const int4 *>(reinterpret_cast<const short *>(a.fields) + (long long)r * a.ldf);
        int4 *dst = reinterpret_cast<int4 *>(D);
        for (int i = tid; i < (int)(a.ldf / 8); i += nthreads) dst[i] = src[i];
        const int8_t *srow = a.spins + (long long)r * a.sstride;
        spins_to_bits(srow, bits, a.sstride, tid, nthreads);
        if ((a.sstride & 31) && tid == 0) {  // (int8 layouts are padded to 16: the last half word)
            unsigned int b = 0;
            for (int t = 0; t < (a.sstride & 31); ++t) b |= (srow[(a.sstride & ~31) + t] < 0 ? 1u : 0u) << t;
            bits[a.sstride / 32] = b;
        }
    }
    __syncthreads();

    double E = a.energy[r], bestE = a.best_energy[r];
    unsigned long long nacc = 0;
    long long ksum = 0;
    double T = 1.0;
    struct RowRegs {
        int2 e[EPT];
    };
    // a row's entries, EPT per thread (entries past the row's end: column 0, value 0 -- skipped when applied)
    auto row_request = [&](long long beg, int len) -> RowRegs {
        RowRegs o;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int off = tid + q * nthreads;
            o.e[q] = a.cv[beg + (off < len ? off : 0)];
            if (off >= len) o.e[q].y = 0;
        }
        return o;
    };
    auto apply_row = [&](const RowRegs &rr, int s_old) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int J = (int)__int_as_float(rr.e[q].y);
            if (J != 0) D[rr.e[q].x] = (short)((int)D[rr.e[q].x] - 2 * J * s_old);  // (distinct columns: nobody else's)
        }
    };
    int turn = 0;
    auto read_lane64 = [](long long v, int l) -> long long {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), l);
        return (long long)(((unsigned long long)hi << 32) | lo);
    };

    for (int k = 0; k < a.n_sweeps; ++k) {
        T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        __syncthreads();
        for (int q = tid; q <= a.table_m; q += nthreads) ptab[q] = expf_det((float)(-((double)(2 * q) * inv_sc) / T));
        __syncthreads();
        for (int t0 = 0; t0 < n; t0 += CLFS_WINDOW * W) {
            const int gA = w * CLFS_WINDOW + 2 * lane, gB = gA + 1;  // positions in the window
            const int tA = t0 + gA, tB = tA + 1;
            const bool vA = tA < n, vB = tB < n;
            uint32_t key_lo = a.seed_lo, key_hi = a.seed_hi;
            asm volatile("" : "+s"(key_lo), "+s"(key_hi));
            const u32x4 x = philox4x32_10((uint32_t)(tA >> 1), a.sweep0 + (uint32_t)k, a.replica0 + (uint32_t)r, DOMAIN_SWEEP,
                                          key_lo, key_hi);
            const int sA = (int)word_to_site(x.x, (uint32_t)n), sB = (int)word_to_site(x.z, (uint32_t)n);
            const float uA = word_to_u(x.y), uB = word_to_u(x.w);
            // what does not change during the window: the sites' static fields and row extents
            const int hA = hq[sA], hB = hq[sB];
            const long long begA = rp[sA], begB = rp[sB];
            const int lenA = (int)(rp[sA + 1] - begA), lenB = (int)(rp[sB + 1] - begB);
            int pos = 0;
            RowRegs buf0 = row_request(0, 0), buf1 = buf0;
            int held_pos = -1;
            auto first_of = [](unsigned long long mA, unsigned long long mB) -> int {
                const int pA = mA ? 2 * (int)__builtin_ctzll(mA) : NONE;
                const int pB = mB ? 2 * (int)__builtin_ctzll(mB) + 1 : NONE;
                return min(pA, pB);
            };
            auto round = [&](RowRegs &held, RowRegs &other) -> bool {
                int p = NONE, p2 = NONE, site = 0, kk = 0, s_old = 1, len = 0, len2 = 0;
                long long beg = 0, beg2 = 0;
                if ((w + 1) * CLFS_WINDOW > pos) {  // (a wave whose window is decided publishes "no accept")
                    const int fa = sc * (int)D[sA] + hA, fb = sc * (int)D[sB] + hB;
                    const int siA = ((bits[sA >> 5] >> (sA & 31)) & 1u) ? -1 : 1, siB = ((bits[sB >> 5] >> (sB & 31)) & 1u) ? -1 : 1;
                    const int kA = siA * fa, kB = siB * fb;
                    const bool liveA = vA && gA >= pos, liveB = vB && gB >= pos;
                    bool accA = liveA && uA < ptab[min(max(kA, 0), a.table_m)];
                    bool accB = liveB && uB < ptab[min(max(kB, 0), a.table_m)];
                    const bool beyondA = liveA && kA > a.table_m, beyondB = liveB && kB > a.table_m;
                    if (ballot64(beyondA || beyondB)) {  // rare: large uphill moves (p == 0 past -104, sweep_common.h)
                        const double dA = (double)(2 * kA) * inv_sc, dB = (double)(2 * kB) * inv_sc;
                        if (beyondA) accA = !(dA > T * 104.0) && uA < expf_det((float)(-dA / T));
                        if (beyondB) accB = !(dB > T * 104.0) && uB < expf_det((float)(-dB / T));
                    }
                    unsigned long long mA = ballot64(accA), mB = ballot64(accB);
                    p = first_of(mA, mB);
                    if (p < NONE) {
                        if (p & 1) mB &= mB - 1;
                        else mA &= mA - 1;
                        p2 = first_of(mA, mB);
                        const int l1 = p >> 1;
                        site = __builtin_amdgcn_readlane((p & 1) ? sB : sA, l1);
                        kk = __builtin_amdgcn_readlane((p & 1) ? kB : kA, l1);
                        s_old = __builtin_amdgcn_readlane((p & 1) ? siB : siA, l1);
                        beg = read_lane64((p & 1) ? begB : begA, l1);
                        len = __builtin_amdgcn_readlane((p & 1) ? lenB : lenA, l1);
                        if (p2 < NONE) {
                            const int l2 = p2 >> 1;
                            beg2 = read_lane64((p2 & 1) ? begB : begA, l2);
                            len2 = __builtin_amdgcn_readlane((p2 & 1) ? lenB : lenA, l2);
                            p2 += w * CLFS_WINDOW;
                        }
                        p += w * CLFS_WINDOW;
                    }
                }
                if (W > 1) {
                    int *slots = slots2 + turn * (CLFS_SLOT_INTS * CLFS_MAX_WAVES);
                    turn ^= 1;
                    if (lane == 0) {
                        int4 *mine = reinterpret_cast<int4 *>(slots + CLFS_SLOT_INTS * w);
                        mine[0] = make_int4(p, p2, site, 0);
                        mine[1] = make_int4(kk, s_old, len, len2);
                        mine[2] = make_int4((int)(unsigned int)beg, (int)(beg >> 32), (int)(unsigned int)beg2, (int)(beg2 >> 32));
                    }
                    __syncthreads();  // (A) every wave has evaluated against the old state and published
                    int4 q0 = make_int4(NONE, NONE, 0, 0), q1 = make_int4(0, 1, 0, 0), q2 = make_int4(0, 0, 0, 0);
                    if (lane < W) {
                        const int4 *theirs = reinterpret_cast<const int4 *>(slots + CLFS_SLOT_INTS * lane);
                        q0 = theirs[0], q1 = theirs[1], q2 = theirs[2];
                    }
                    const unsigned long long have = ballot64(q0.x < NONE);
                    if (have == 0ull) {
                        p = NONE;
                    } else {
                        const int win = (int)__builtin_ctzll(have);
                        p = __builtin_amdgcn_readlane(q0.x, win), p2 = __builtin_amdgcn_readlane(q0.y, win);
                        site = __builtin_amdgcn_readlane(q0.z, win);
                        kk = __builtin_amdgcn_readlane(q1.x, win), s_old = __builtin_amdgcn_readlane(q1.y, win);
                        len = __builtin_amdgcn_readlane(q1.z, win), len2 = __builtin_amdgcn_readlane(q1.w, win);
                        beg = (long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(q2.y, win) << 32) |
                                          (unsigned int)__builtin_amdgcn_readlane(q2.x, win));
                        beg2 = (long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(q2.w, win) << 32) |
                                           (unsigned int)__builtin_amdgcn_readlane(q2.z, win));
                        const unsigned long long later = have & (have - 1);
                        if (p2 >= NONE && later) {  // the predicted next accept: the first of a later wave
                            const int nx = (int)__builtin_ctzll(later);
                            p2 = __builtin_amdgcn_readlane(q0.x, nx);
                            len2 = __builtin_amdgcn_readlane(q1.z, nx);
                            beg2 = (long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(q2.y, nx) << 32) |
                                               (unsigned int)__builtin_amdgcn_readlane(q2.x, nx));
                        }
                    }
                }
                if (p >= NONE) return true;  // the rest of the window is rejected
                if (held_pos != p) held = row_request(beg, len);
                other = row_request(p2 < NONE ? beg2 : beg, p2 < NONE ? len2 : len);
                held_pos = p2 < NONE ? p2 : -1;
                ksum += (long long)kk;
                ++nacc;
                apply_row(held, s_old);
                if (tid == 0) bits[site >> 5] ^= 1u << (site & 31);
                pos = p + 1;
                __syncthreads();  // (B) fields and spin of the new state are visible
                return pos >= CLFS_WINDOW * W;
            };
            for (;;) {
                if (round(buf0, buf1)) break;
                if (round(buf1, buf0)) break;
            }
        }
        // sweep boundary: energy record, best tracking (annealing/gpu_annealer.py:151-153)
        E += (double)(2 * ksum) * inv_sc;  // (integers below 2^53: exact)
        ksum = 0;
        if (tid == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE && !a.no_best) {
            bestE = E;
            bits_to_spins(bits, a.best_spins + (long long)r * a.sstride, a.sstride, n, tid, nthreads);
        }
    }
    __syncthreads();
    {
        int4 *dst = reinterpret_cast<int4 *>(reinterpret_cast<short *>(a.fields) + (long long)r * a.ldf);
        const int4 *src = reinterpret_cast<const int4 *>(D);
        for (int i = tid; i < (int)(a.ldf / 8); i += nthreads) dst[i] = src[i];
        bits_to_spins(bits, a.spins + (long long)r * a.sstride, a.sstride, n, tid, nthreads);
    }
    if (tid == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

// production arguments only (Philox sites, Metropolis in the reference's fp64 / fp32-exp arithmetic, no per-update
// records); the engine checks the problem (integer J, sorted rows, |D| < 2^15, LDS)
bool sweep_clf_csr_applies(const SweepArgs &a, int waves) {
    return sweep_args_are_lean(a) && a.rule == SGA_RULE_METROPOLIS && a.table_m > 0 && a.clf_hq && a.fields && a.rowptr64 &&
           a.clf_row_max <= 4 * 64 * waves && a.ldf % 8 == 0 && a.sstride % 16 == 0 &&
           sweep_clf_csr_lds_bytes(a.ldf, a.sstride, a.table_m) <= 160 * 1024;
}

hipError_t launch_sweep_clf_csr(const SweepArgs &a, int waves, hipStream_t st) {
    if (waves < 1 || waves > CLFS_MAX_WAVES || !sweep_clf_csr_applies(a, waves)) return hipErrorInvalidValue;
    const int ept = (a.clf_row_max + 64 * waves - 1) / (64 * waves);
    void (*kern)(const SweepArgs) = ept <= 1 ? sweep_clf_csr_kernel<1> : ept <= 2 ? sweep_clf_csr_kernel<2> : sweep_clf_csr_kernel<4>;
    const size_t lds = sweep_clf_csr_lds_bytes(a.ldf, a.sstride, a.table_m);
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.rep_list ? a.rep_count : a.R), dim3(64 * waves), lds, st, a);
    note_sweep_kernel("sweep_clf_csr_kernel<%d entries per thread> x %d wave(s) (int16 fields in LDS, row read on accept only)",
                      ept <= 1 ? 1 : ept <= 2 ? 2 : 4, waves);
    return hipGetLastError();
}

}  // namespace sga
