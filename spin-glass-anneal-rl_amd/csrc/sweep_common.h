// sweep_common.h -- pieces shared by the dense and CSR sweep kernels: where the site and the
// uniform of an update come from, and the Metropolis accept rule.
#pragma once
#include "sga.h"
#include "sga_device.h"
#include "sga_kernels.h"

namespace sga {

// Two consecutive updates (t = 2b, 2b+1) of one sweep: one Philox block serves both.
struct UpdatePair {
    int sA, sB;
    float uA, uB;
    // production (Philox) streams: the uniforms' 24 raw bits, u = r * 2^-24 -- a table of integer
    // thresholds ceil(p * 2^24) decides u < p without converting u (r < ceil(p 2^24) <=> u < p)
    uint32_t rA = 0, rB = 0;
};

// Kernels are compiled twice: LEAN = the production configuration (Philox random sites, the
// reference's fp64/fp32 rule arithmetic, no per-update traces) with the site / arithmetic / trace
// tests folded away at compile time -- fewer live scalars, no SGPR spills -- and the general
// variant that also serves the replay / sequential / fp32-operator / traced modes.  The accept
// table and the look-ahead form are Metropolis only.
inline bool sweep_args_are_lean(const SweepArgs &a) {
    return !a.force_general && a.site_mode == SGA_SITE_RANDOM && a.arith == SGA_ARITH_F64 &&
           !a.accept_trace && !a.dE_trace;  // any rule: it stays a wave-uniform run-time value
}

// Everything here is wave-uniform (blockIdx / loop counters / kernel arguments).
template <bool LEAN>
__device__ __forceinline__ UpdatePair fetch_pair(const SweepArgs &a, int r, int k, int b,
                                                 bool valid) {
    UpdatePair o{0, 0, 2.0f, 2.0f};
    if (!valid) return o;
    const int n = a.n, t0 = 2 * b, t1 = 2 * b + 1;
    const bool hasB = t1 < n;
    if constexpr (LEAN) {
        const u32x4 w = philox4x32_10((uint32_t)b, a.sweep0 + (uint32_t)k,
                                      a.replica0 + (uint32_t)r, DOMAIN_SWEEP, a.seed_lo, a.seed_hi);
        o.sA = (int)word_to_site(w.x, (uint32_t)n);
        o.sB = (int)word_to_site(w.z, (uint32_t)n);
        o.uA = word_to_u(w.y);
        o.uB = word_to_u(w.w);
        return o;
    }
    const long long base = (long long)r * a.replay_stride + (long long)k * n;
    if (a.site_mode == SGA_SITE_REPLAY) {
        // recorded stream of the reference: site = torch.randint(0, n, (1,)) at
        // core/spin_dynamics.py:69, u = torch.rand(1) at :146
        o.sA = a.replay_site[base + t0];
        o.uA = a.replay_u[base + t0];
        if (hasB) {
            o.sB = a.replay_site[base + t1];
            o.uB = a.replay_u[base + t1];
        }
        o.sA = min(max(o.sA, 0), n - 1);
        o.sB = min(max(o.sB, 0), n - 1);
        return o;
    }
    const u32x4 w = philox4x32_10((uint32_t)b, a.sweep0 + (uint32_t)k, a.replica0 + (uint32_t)r,
                                  DOMAIN_SWEEP, a.seed_lo, a.seed_hi);
    if (a.site_mode == SGA_SITE_RANDOM) {
        o.sA = (int)word_to_site(w.x, (uint32_t)n);
        o.sB = (int)word_to_site(w.z, (uint32_t)n);
    } else {  // SGA_SITE_SEQUENTIAL: annealing/cuda_kernels.py:381  for i in range(n_spins)
        o.sA = t0;
        o.sB = hasB ? t1 : 0;
    }
    if (a.site_mode == SGA_SITE_SEQUENTIAL && a.replay_u) {
        o.uA = a.replay_u[base + t0];
        if (hasB) o.uB = a.replay_u[base + t1];
    } else {
        o.uA = word_to_u(w.y);
        o.uB = word_to_u(w.w);
    }
    return o;
}

// Supplier of consecutive update pairs (k, b), b = 0, 1, 2, ... within a sweep.
//
// General variant: one scalar Philox block per pair (fetch_pair).  LEAN variant: the Philox
// counter is VECTORISED ACROSS THE WAVE -- lane l evaluates block (b & ~63) + l, so a single
// pass of the 10 rounds through the VALU yields the next 64 blocks = 128 updates, and each
// pair is then broadcast out of its lane with v_readlane.  Same counter -> output map as the
// scalar form (bit-identical streams), but ~55 SALU instructions per update become ~2: the
// CSR and small-dense kernels are issue-bound on exactly those.
template <bool LEAN>
struct PairSource {
    uint32_t vsa, vua, vsb, vub;  // this lane's block: site A, u-bits A, site B, u-bits B

    __device__ __forceinline__ UpdatePair get(const SweepArgs &a, int r, int k, int b, bool valid,
                                              int lane) {
        if constexpr (!LEAN) {
            return fetch_pair<false>(a, r, k, b, valid);
        } else {
            UpdatePair o{0, 0, 2.0f, 2.0f};
            if (!valid) return o;
            if ((b & 63) == 0) {  // wave-uniform: a new batch starts here
                const u32x4 w = philox4x32_10((uint32_t)(b + lane), a.sweep0 + (uint32_t)k,
                                              a.replica0 + (uint32_t)r, DOMAIN_SWEEP, a.seed_lo,
                                              a.seed_hi);
                vsa = word_to_site(w.x, (uint32_t)a.n);
                vua = w.y;
                vsb = word_to_site(w.z, (uint32_t)a.n);
                vub = w.w;
            }
            const int l = b & 63;
            o.sA = __builtin_amdgcn_readlane((int)vsa, l);
            o.sB = __builtin_amdgcn_readlane((int)vsb, l);
            o.rA = (uint32_t)__builtin_amdgcn_readlane((int)vua, l) >> 8;
            o.rB = (uint32_t)__builtin_amdgcn_readlane((int)vub, l) >> 8;
            o.uA = (float)o.rA * 0x1.0p-24f;
            o.uB = (float)o.rB * 0x1.0p-24f;
            return o;
        }
    }
};

// The accept rule.  dot = fp32 coupling dot product J[site,:].s (already rounded to fp32),
// si = s[site] (+-1).  Returns true if the spin flips; dE receives the energy change of the
// flip.
__device__ __forceinline__ bool metropolis_accept(int rule, int arith, float dot, int si,
                                                  float h_site, float diag_site, double T,
                                                  float u, double &dE) {
    if (rule != SGA_RULE_METROPOLIS) {
        // core/spin_dynamics.py:154-171 (Glauber) and :173-191 (heat bath):
        //   prob_up = 1.0 / (1.0 + exp(float32(x))),  x = -2.0*field/T  |  (-2.0*(1.0/T))*field
        //   new_spin = +1 if rand < prob_up else -1;  flip iff new_spin != s_i
        const double field = (double)dot + (double)h_site;
        const float x = (rule == SGA_RULE_GLAUBER) ? (float)(-2.0 * field / T)
                                                   : (float)((-2.0 * (1.0 / T)) * field);
        const float prob_up = 1.0f / (1.0f + expf_det(x));
        const int new_spin = (u < prob_up) ? 1 : -1;
        dE = 2.0 * (double)si * field;
        return new_spin != si;
    }
    if (arith == SGA_ARITH_F64) {
        // core/spin_dynamics.py:131-152 with core/ising_model.py:176-185:
        //   local_field = float(dot) + float(h[i])          (python doubles)
        //   delta_energy = 2.0 * s_i * local_field
        //   accept if delta_energy <= 0 else rand < exp(float32(-delta_energy / T))
        const double field = (double)dot + (double)h_site;
        dE = 2.0 * (double)si * field;
        if (dE <= 0.0) return true;
        // -dE/T below -104 rounds to an fp32 argument below expf_det's underflow bound (-103.97):
        // the probability is exactly 0 and no uniform in [0, 1) is below it -- same decision without
        // the fp64 divide and the exp (most uphill proposals of a cold replica)
        if (dE > T * 104.0) return false;
        const float p = expf_det((float)(-dE / T));
        return u < p;
    }
    // annealing/cuda_kernels.py:383-390 (fp32 tensors):
    //   local_field = h[i] + sum(J[i] * s) - J[i,i] * s[i]
    //   delta_energy = 2.0 * s[i] * local_field
    //   accept if delta_energy <= 0 or rand < exp(-delta_energy / T)
    const float sif = (float)si;
    const float field = (h_site + dot) - diag_site * sif;
    const float dEf = (2.0f * sif) * field;
    dE = (double)dEf;
    if (dEf <= 0.0f) return true;
    const float p = expf_det(-dEf / (float)T);
    return u < p;
}

// Replica spins between the int8 HBM layout ([sstride], +-1, pad = 0) and one bit per spin in
// LDS (1 = spin down).  `first`/`step`: this thread's share of the loop.  sstride % 32 == 0.
__device__ inline void spins_to_bits(const int8_t *src_row, unsigned int *bits, int sstride,
                                     int first, int step) {
    const int4 *src = reinterpret_cast<const int4 *>(src_row);
    for (int i = first; i < sstride / 32; i += step) {
        const int4 lo = src[2 * i], hi = src[2 * i + 1];
        const int wds[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        unsigned int b = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)  // sign bit of each of the 4 bytes of a dword
            b |= (((unsigned)wds[q] >> 7) & 1u) << (4 * q) | (((unsigned)wds[q] >> 15) & 1u) << (4 * q + 1) |
                 (((unsigned)wds[q] >> 23) & 1u) << (4 * q + 2) | (((unsigned)wds[q] >> 31) & 1u) << (4 * q + 3);
        bits[i] = b;
    }
}
__device__ inline void bits_to_spins(const unsigned int *bits, int8_t *dst_row, int sstride, int n,
                                     int first, int step) {
    int4 *dst = reinterpret_cast<int4 *>(dst_row);
    for (int i = first; i < sstride / 16; i += step) {
        const unsigned int half = (bits[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
        int out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned int v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = 16 * i + 4 * q + b;
                const unsigned int byte = j < n ? (((half >> (4 * q + b)) & 1u) ? 0xFFu : 0x01u) : 0u;
                v |= byte << (8 * b);
            }
            out[q] = (int)v;
        }
        dst[i] = make_int4(out[0], out[1], out[2], out[3]);
    }
}

}  // namespace sga
