// sweep_csr.hip -- single-spin sweep over CSR couplings (BASELINE config 3: N = 10k, degree
// ~32, 4096 replicas).
//
// Replaces the same reference functions as the dense kernel (core/spin_dynamics.py:73-94,
// core/ising_model.py:176-185) for IsingModelConfig(use_sparse=True) models; the reference's
// own sparse branch (ising_model.py:133-135) raises under the container's torch, so the math
// is the dense path's with the row restricted to its stored entries.
//
// Mapping: one replica per wavefront, up to CSR_WAVES_PER_BLOCK independent replicas per
// workgroup (as many as fit LDS: 4 up to n = 40k, 2 up to 80k, 1 up to 160k) (no workgroup barrier anywhere: each wave owns a private LDS slice holding its
// replica's spins).  A row has ~32 entries, i.e. one (colidx, val) wave-load each; the spin
// gather goes through LDS; the dot is a DPP wave sum.  The structure (2.6 MB at C3) is
// L2-resident, so the kernel is bound by instruction issue and the dependent-load chain,
// not by HBM; hence:
//   * the site sequence is known ahead of time (counter RNG): row extents are loaded one
//     PAIR of updates ahead and row entries one update ahead, so no update waits on a
//     rowptr -> colidx dependent load;
//   * FAST variant (integer-valued J and h, sum_j |J_ij| + |h_i| <= M small): the row sum is
//     accumulated in fp32 (exact), and the Metropolis probability exp(float32(-dE/T)) of the
//     M possible uphill moves dE = 2k is tabulated in LDS once per sweep -- the same function
//     of the same arguments, so decisions are bit-identical to the general path -- which
//     removes the fp64 divide and the exp from the per-update chain.
#include "sweep_common.h"

namespace sga {

constexpr int TAIL_UNROLL = 8;  // wave-loads of a long row kept in flight together
constexpr int CSR_MAX_WIDE = 8;  // most waves one replica's row is dealt to

// WIDE = several waves per replica (long rows, few replicas): one replica per workgroup, the
// row's entries are dealt to the waves in 64-entry slices, the per-wave sums meet in LDS with
// one barrier per update (double-buffered slots, as in the dense kernel).  Every wave applies
// an accepted flip to the shared spin byte itself before its next gather (same value from all
// waves), so no second barrier is needed.
template <bool FAST, bool LEAN, bool WIDE>
__global__ void __launch_bounds__(64 * (WIDE ? CSR_MAX_WIDE : CSR_WAVES_PER_BLOCK))
    sweep_csr_kernel(const SweepArgs a) {
    const int rule = LEAN ? SGA_RULE_METROPOLIS : a.rule;
    const int arith = LEAN ? SGA_ARITH_F64 : a.arith;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int nw = blockDim.x >> 6;                     // waves in the workgroup
    const int r = WIDE ? (int)blockIdx.x : (int)blockIdx.x * nw + w;
    if (!WIDE && r >= a.R) return;  // wave-uniform; the narrow form has no barriers
    const int n = a.n;
    const int slots = WIDE ? 1 : nw;                    // replicas sharing this workgroup's LDS
    const int me = WIDE ? 0 : w;
    const int stride_lanes = WIDE ? 64 * nw : 64;       // entries between a lane's row elements
    const int first_lane = WIDE ? w * 64 + lane : lane; // this lane's first entry of a row
    int8_t *s = reinterpret_cast<int8_t *>(smem) + (long long)me * a.sstride;
    float *ptab = reinterpret_cast<float *>(smem + (long long)slots * a.sstride) +
                  (long long)me * (a.table_m + 1);
    double *part = reinterpret_cast<double *>(smem + (long long)slots * a.sstride +
                                              sizeof(float) * ((a.table_m + 2) & ~1) * slots);
    int pp = 0;
    {
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (long long)r * a.sstride);
        int4 *dst = reinterpret_cast<int4 *>(s);
        const int step = WIDE ? (int)blockDim.x : 64, first = WIDE ? tid : lane;
        for (int i = first; i < a.sstride / 16; i += step) dst[i] = src[i];
    }
    if constexpr (WIDE) __syncthreads();
    const bool arith32 = arith == SGA_ARITH_F32;
    double E = a.energy[r], bestE = a.best_energy[r];
    unsigned long long nacc = 0;

    struct Extent {  // what is indexed by the site alone
        int beg, end;
        float h, d;
    };
    struct Head {  // first 64 stored entries of the row, one per lane
        int col;
        float val;
    };
    auto load_extent = [&](int site) {
        Extent o;
        o.beg = a.rowptr[site];
        o.end = a.rowptr[site + 1];
        o.h = a.h[site];
        o.d = arith32 ? a.diag[site] : 0.0f;
        return o;
    };
    auto load_head = [&](const Extent &x) {
        Head o;
        const int j = x.beg + first_lane;
        const bool in = j < x.end;
        o.col = in ? a.colidx[j] : 0;
        o.val = in ? a.val[j] : 0.0f;
        return o;
    };

    double T = 1.0;
    auto update = [&](int site, float u, const Extent &x, const Head &hd, long long upd) {
        // read s_i before any wave can have applied THIS update's flip (WIDE: before the barrier)
        const int si = s[site];
        // J[site,:].s over the stored entries; products val * (+-1) are exact
        float dot;
        if constexpr (FAST) {
            float acc = hd.val * (float)s[hd.col];
            // long rows: issue eight (colidx, val) wave-loads before the first gather so the
            // round trips overlap instead of serialising (degree ~600 at C4)
            for (int j0 = x.beg + stride_lanes + first_lane; j0 < x.end;
                 j0 += stride_lanes * TAIL_UNROLL) {
                int c[TAIL_UNROLL];
                float v[TAIL_UNROLL];
#pragma unroll
                for (int q = 0; q < TAIL_UNROLL; ++q) {
                    const int j = j0 + stride_lanes * q;
                    const bool in = j < x.end;
                    c[q] = in ? a.colidx[j] : 0;
                    v[q] = in ? a.val[j] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < TAIL_UNROLL; ++q) acc += v[q] * (float)s[c[q]];
            }
            dot = wave_sum(acc);
            if constexpr (WIDE) {
                float *slot = reinterpret_cast<float *>(part) + pp * CSR_MAX_WIDE;
                if (lane == 0) slot[w] = dot;
                __syncthreads();
                float t = slot[0];
                for (int i = 1; i < nw; ++i) t += slot[i];
                dot = t;
                pp ^= 1;
            }
        } else {  // fp64 sum rounded to fp32 once (core/ising_model.py:183)
            double acc = (double)(hd.val * (float)s[hd.col]);
            for (int j0 = x.beg + stride_lanes + first_lane; j0 < x.end;
                 j0 += stride_lanes * TAIL_UNROLL) {
                int c[TAIL_UNROLL];
                float v[TAIL_UNROLL];
#pragma unroll
                for (int q = 0; q < TAIL_UNROLL; ++q) {
                    const int j = j0 + stride_lanes * q;
                    const bool in = j < x.end;
                    c[q] = in ? a.colidx[j] : 0;
                    v[q] = in ? a.val[j] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < TAIL_UNROLL; ++q) acc += (double)(v[q] * (float)s[c[q]]);
            }
            double tot = wave_sum(acc);
            if constexpr (WIDE) {
                double *slot = part + pp * CSR_MAX_WIDE;
                if (lane == 0) slot[w] = tot;
                __syncthreads();
                double t = slot[0];
                for (int i = 1; i < nw; ++i) t += slot[i];
                tot = t;
                pp ^= 1;
            }
            dot = (float)tot;
        }
        double dE;
        bool flip;
        if (FAST && rule == SGA_RULE_METROPOLIS && arith == SGA_ARITH_F64) {
            // core/spin_dynamics.py:131-152 with every quantity an integer: dE = 2k exactly
            const float fk = (float)si * (dot + x.h);
            dE = (double)(2.0f * fk);
            if (fk <= 0.0f) flip = true;
            else if (fk <= (float)a.table_m) flip = u < ptab[(int)fk];
            else flip = u < expf_det((float)(-dE / T));  // beyond the table: evaluate
        } else {
            flip = metropolis_accept(rule, arith, dot, si, x.h, x.d, T, u, dE);
        }
        if (flip) {
            E += dE;
            ++nacc;
            if (lane == 0) s[site] = (int8_t)(-si);
        }
        if constexpr (!LEAN) {
            if (lane == 0) {
                if (a.accept_trace)
                    a.accept_trace[(long long)r * a.replay_stride + upd] = flip ? 1 : 0;
                if (a.dE_trace)
                    a.dE_trace[(long long)r * a.replay_stride + upd] =
                        flip ? (rule == SGA_RULE_HEAT_BATH ? -dE : dE) : 0.0;
            }
        }
    };

    const int nb = (n + 1) >> 1;
    PairSource<LEAN> rng;
    UpdatePair cur = rng.get(a, r, 0, 0, a.n_sweeps > 0, lane);
    Extent xA = load_extent(cur.sA), xB = load_extent(cur.sB);
    Head hA = load_head(xA);

    for (int k = 0; k < a.n_sweeps; ++k) {
        T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        if constexpr (FAST) {  // exp(float32(-dE / T)) for dE = 2k, k = 0..M
            if constexpr (WIDE) __syncthreads();  // nobody still reads last sweep's table
            for (int q = first_lane; q <= a.table_m; q += stride_lanes)
                ptab[q] = expf_det((float)(-(double)(2 * q) / T));
            if constexpr (WIDE) __syncthreads();
        }
        for (int b = 0; b < nb; ++b) {
            const bool last = (b + 1 == nb);
            const int kn = last ? k + 1 : k, bn = last ? 0 : b + 1;
            const UpdatePair nxt = rng.get(a, r, kn, bn, kn < a.n_sweeps, lane);
            const bool hasB = (2 * b + 1) < n;
            const Extent nA = load_extent(nxt.sA), nB = load_extent(nxt.sB);  // a pair ahead
            Head hB{0, 0.0f};
            if (hasB) hB = load_head(xB);  // in flight while A is reduced
            update(cur.sA, cur.uA, xA, hA, (long long)k * n + 2 * b);
            const Head hN = load_head(nA);  // in flight while B is reduced
            if (hasB) update(cur.sB, cur.uB, xB, hB, (long long)k * n + 2 * b + 1);
            cur = nxt;
            xA = nA;
            xB = nB;
            hA = hN;
        }
        const int cstep = WIDE ? (int)blockDim.x : 64, cfirst = WIDE ? tid : lane;
        if (lane == 0 && (!WIDE || w == 0) && a.energy_trace)
            a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE && !a.no_best) {  // annealing/gpu_annealer.py:151-153
            bestE = E;
            if constexpr (WIDE) __syncthreads();  // every wave has applied the last flip
            int4 *dst = reinterpret_cast<int4 *>(a.best_spins + (long long)r * a.sstride);
            const int4 *src = reinterpret_cast<const int4 *>(s);
            for (int i = cfirst; i < a.sstride / 16; i += cstep) dst[i] = src[i];
            if constexpr (WIDE) __syncthreads();
        }
    }
    if constexpr (WIDE) __syncthreads();
    {
        int4 *dst = reinterpret_cast<int4 *>(a.spins + (long long)r * a.sstride);
        const int4 *src = reinterpret_cast<const int4 *>(s);
        const int cstep = WIDE ? (int)blockDim.x : 64, cfirst = WIDE ? tid : lane;
        for (int i = cfirst; i < a.sstride / 16; i += cstep) dst[i] = src[i];
    }
    if (lane == 0 && (!WIDE || w == 0)) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

static size_t csr_lds_per_replica(int sstride, int table_m) {  // spins + accept table (>= 8 B)
    return (size_t)sstride + sizeof(float) * (size_t)((table_m + 2) & ~1);
}

int csr_waves_per_block(int sstride, int table_m) {
    const int wpb = (int)((160 * 1024 - 256) / csr_lds_per_replica(sstride, table_m));
    return wpb > CSR_WAVES_PER_BLOCK ? CSR_WAVES_PER_BLOCK : wpb;  // 0: does not fit
}

template <bool WIDE>
static hipError_t launch_csr(const SweepArgs &a, int waves, hipStream_t st) {
    const bool fast = a.table_m > 0, lean = sweep_args_are_lean(a);
    const int slots = WIDE ? 1 : waves;
    const size_t lds = csr_lds_per_replica(a.sstride, a.table_m) * slots +
                       2 * CSR_MAX_WIDE * sizeof(double);
    auto kern = fast ? (lean ? sweep_csr_kernel<true, true, WIDE> : sweep_csr_kernel<true, false, WIDE>)
                     : (lean ? sweep_csr_kernel<false, true, WIDE> : sweep_csr_kernel<false, false, WIDE>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int blocks = WIDE ? a.R : (a.R + waves - 1) / waves;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), lds, st, a);
    return hipGetLastError();
}

// waves_per_replica == 1: several replicas per workgroup (as many as fit LDS), no barriers;
// > 1: one replica per workgroup, its rows dealt to that many waves.
hipError_t launch_sweep_csr(const SweepArgs &a, int waves_per_replica, hipStream_t st) {
    if (waves_per_replica > 1) {
        if (waves_per_replica > CSR_MAX_WIDE || csr_waves_per_block(a.sstride, a.table_m) < 1)
            return hipErrorInvalidValue;
        return launch_csr<true>(a, waves_per_replica, st);
    }
    const int wpb = csr_waves_per_block(a.sstride, a.table_m);
    if (wpb < 1) return hipErrorInvalidValue;
    return launch_csr<false>(a, wpb, st);
}

}  // namespace sga
