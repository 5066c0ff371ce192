// sweep_csr.hip -- Metropolis sweep over CSR couplings (BASELINE config 3: N = 10k, degree
// ~32, 4096 replicas).
//
// Replaces the same reference functions as the dense kernel (core/spin_dynamics.py:73-94,
// core/ising_model.py:176-185) for IsingModelConfig(use_sparse=True) models; the reference's
// own sparse branch (ising_model.py:133-135) raises under the container's torch, so the math
// is the dense path's with the row restricted to its stored entries.
//
// Mapping: one replica per wavefront, CSR_WAVES_PER_BLOCK independent replicas per
// workgroup (no workgroup barrier anywhere: each wave owns a private LDS slice holding its
// replica's spins).  A row has ~32 entries, i.e. one (colidx, val) wave-load each; the spin
// gather goes through LDS; the dot is a DPP wave sum.  The site sequence is known ahead of
// time (counter RNG), so the next update's row extent and entries are loaded while the
// current one is reduced.
#include "sweep_common.h"

namespace sga {

__global__ void __launch_bounds__(64 * CSR_WAVES_PER_BLOCK) sweep_csr_kernel(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = blockIdx.x * CSR_WAVES_PER_BLOCK + w;
    if (r >= a.R) return;  // wave-uniform; no barriers below
    const int n = a.n;
    int8_t *s = reinterpret_cast<int8_t *>(smem) + (long long)w * a.sstride;
    {
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (long long)r * a.sstride);
        int4 *dst = reinterpret_cast<int4 *>(s);
        for (int i = lane; i < a.sstride / 16; i += 64) dst[i] = src[i];
    }
    const bool arith32 = a.arith == SGA_ARITH_F32;
    double E = a.energy[r], bestE = a.best_energy[r];
    unsigned long long nacc = 0;

    // prefetched head of a row: extent, and its first 64 entries (one per lane)
    struct RowHead {
        int beg, end, col;
        float val, h, d;
    };
    auto load_head = [&](int site) {
        RowHead o;
        o.beg = a.rowptr[site];
        o.end = a.rowptr[site + 1];
        const int j = o.beg + lane;
        const bool in = j < o.end;
        o.col = in ? a.colidx[j] : 0;
        o.val = in ? a.val[j] : 0.0f;
        o.h = a.h[site];
        o.d = arith32 ? a.diag[site] : 0.0f;
        return o;
    };

    const int nb = (n + 1) >> 1;
    UpdatePair cur = fetch_pair(a, r, 0, 0, a.n_sweeps > 0);
    RowHead head = load_head(cur.sA);

    for (int k = 0; k < a.n_sweeps; ++k) {
        const double T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        for (int b = 0; b < nb; ++b) {
            const bool last = (b + 1 == nb);
            const int kn = last ? k + 1 : k, bn = last ? 0 : b + 1;
            const UpdatePair nxt = fetch_pair(a, r, kn, bn, kn < a.n_sweeps);
            const bool hasB = (2 * b + 1) < n;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 1 && !hasB) break;
                const int site = half ? cur.sB : cur.sA;
                const float u = half ? cur.uB : cur.uA;
                const int site_next = (half == 0 && hasB) ? cur.sB : nxt.sA;
                const RowHead nh = load_head(site_next);  // in flight during the reduction
                // J[site,:].s over the stored entries: products are exact (val * +-1), the
                // sum is formed in fp64 and rounded to fp32 once (core/ising_model.py:183)
                double acc = (double)(head.val * (float)s[head.col]);
                for (int j = head.beg + 64 + lane; j < head.end; j += 64)
                    acc += (double)(a.val[j] * (float)s[a.colidx[j]]);
                const float dot = (float)wave_sum(acc);
                const int si = s[site];
                double dE;
                const bool acc_flip = metropolis_accept(a.rule, a.arith, dot, si, head.h, head.d, T, u, dE);
                if (acc_flip) {
                    E += dE;
                    ++nacc;
                    if (lane == 0) s[site] = (int8_t)(-si);
                }
                if (lane == 0) {
                    const long long upd = (long long)k * n + 2 * b + half;
                    if (a.accept_trace)
                        a.accept_trace[(long long)r * a.replay_stride + upd] = acc_flip ? 1 : 0;
                    if (a.dE_trace)
                        a.dE_trace[(long long)r * a.replay_stride + upd] =
                            acc_flip ? (a.rule == SGA_RULE_HEAT_BATH ? -dE : dE) : 0.0;
                }
                head = nh;
            }
            cur = nxt;
        }
        if (lane == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE) {  // annealing/gpu_annealer.py:151-153
            bestE = E;
            int4 *dst = reinterpret_cast<int4 *>(a.best_spins + (long long)r * a.sstride);
            const int4 *src = reinterpret_cast<const int4 *>(s);
            for (int i = lane; i < a.sstride / 16; i += 64) dst[i] = src[i];
        }
    }
    {
        int4 *dst = reinterpret_cast<int4 *>(a.spins + (long long)r * a.sstride);
        const int4 *src = reinterpret_cast<const int4 *>(s);
        for (int i = lane; i < a.sstride / 16; i += 64) dst[i] = src[i];
    }
    if (lane == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

hipError_t launch_sweep_csr(const SweepArgs &a, hipStream_t st) {
    const size_t lds = (size_t)a.sstride * CSR_WAVES_PER_BLOCK;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sweep_csr_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int blocks = (a.R + CSR_WAVES_PER_BLOCK - 1) / CSR_WAVES_PER_BLOCK;
    hipLaunchKernelGGL(sweep_csr_kernel, dim3(blocks), dim3(64 * CSR_WAVES_PER_BLOCK), lds, st, a);
    return hipGetLastError();
}

}  // namespace sga
