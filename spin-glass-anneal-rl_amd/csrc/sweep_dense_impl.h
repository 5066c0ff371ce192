// sweep_dense_impl.h -- the dense Metropolis sweep kernel (the HBM-bound hot kernel).
//
// Replaces: CUDAKernelManager.metropolis_update_optimized (annealing/cuda_kernels.py:228-282,
// fallback :371-398) and SpinDynamics.sweep (core/spin_dynamics.py:73-94) batched over the
// replicas of ParallelTempering._parallel_sweeps (annealing/parallel_tempering.py:191-203).
//
// Mapping (DESIGN.md "Dense sweep kernel"):
//   * one workgroup per replica, W waves (W chosen on the host so R*W waves fill the chip);
//   * a coupling row J[site,:] is cut into 1-KiB chunks (64 lanes x 16 B, one
//     global_load_dwordx4 per wave); wave w owns chunks w, w+W, w+2W, ... -- CPW of them --
//     and keeps them in VGPRs: no LDS round trip for data that is streamed once;
//   * the replica's spins live in LDS as int8; each wave reads only the bytes under its
//     own chunks, and only the owning wave ever rewrites them (no cross-wave hazard);
//   * the single-spin Markov chain is serial, but the SITE sequence is known ahead of
//     time from the counter RNG, so row t+1 is prefetched into a second register buffer
//     while row t is reduced: CPW KiB per wave stay in flight across the per-update
//     barrier (plain loads survive s_barrier);
//   * per update: lane partial -> DPP wave sum -> W partials through LDS (one barrier,
//     double-buffered slots) -> every thread evaluates the same accept rule;
//   * CPW = 0 selects the STREAMING form for rows too long for the register buffers
//     (n > 40 960 fp32 / 163 840 int8 elements per 16 waves x 10 chunks): each wave walks its
//     chunks in batches of four loads and reduces them on the fly (no cross-update prefetch;
//     at those sizes 16 waves x 3+ workgroups per CU keep enough bytes in flight).
#pragma once
#include <type_traits>

#include "sweep_common.h"

namespace sga {

template <typename JT>
struct JTraits;
template <>
struct JTraits<float> {
    using vec_t = float4;
    static constexpr int EPL = 4;  // elements per lane per chunk (16 B)
};
template <>
struct JTraits<int8_t> {
    using vec_t = int4;
    static constexpr int EPL = 16;
};

// Ternary couplings J in {-1, 0, +1} as two bit-planes per row (sign, non-zero): 16x fewer
// bytes than fp32; the row dot becomes deg_i - 2 * popcount(nz & (sign ^ spin_bits)).
struct Tern2 {};
struct BitPair {
    int4 s, z;  // 128 sign bits (1 = negative) and 128 non-zero bits per lane per chunk
};
template <>
struct JTraits<Tern2> {
    using vec_t = BitPair;
    static constexpr int EPL = 128;
};

template <typename JT, bool ACC64>
struct AccType {
    using type = float;
};
template <bool ACC64>
struct AccType<Tern2, ACC64> {
    using type = int;
};
template <>
struct AccType<float, true> {
    using type = double;
};
template <bool ACC64>
struct AccType<int8_t, ACC64> {
    using type = int;
};

constexpr int PART_SLOT_BYTES = 8;
constexpr int DENSE_LDS_EXTRA = 2 * MAX_WAVES * PART_SLOT_BYTES + 16;

template <typename JT, int CPW, bool ACC64, bool LEAN>
__global__ void __launch_bounds__(1024) sweep_dense_kernel(const SweepArgs a) {
    const int rule = LEAN ? SGA_RULE_METROPOLIS : a.rule;
    const int arith = LEAN ? SGA_ARITH_F64 : a.arith;
    using TR = JTraits<JT>;
    using vec_t = typename TR::vec_t;
    using acc_t = typename AccType<JT, ACC64>::type;
    constexpr int EPL = TR::EPL, EPC = 64 * EPL;
    constexpr bool BITS = std::is_same<JT, Tern2>::value;  // spins are bits in LDS as well
    static_assert(!BITS || LEAN, "the bit-plane form serves the production configuration only");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const long long sbytes = BITS ? a.ld / 8 : a.ld;                       // LDS bytes of the spins
    int8_t *s_lds = reinterpret_cast<int8_t *>(smem);                      // [ld] int8 | [ld/8] bits
    unsigned int *s_bits = reinterpret_cast<unsigned int *>(smem);
    unsigned char *part_raw = smem + sbytes;                               // [2][MAX_WAVES] 8-B slots
    int *sislot = reinterpret_cast<int *>(smem + sbytes + 2 * MAX_WAVES * PART_SLOT_BYTES);  // [2]
    // integer problems with few distinct uphill moves: exp(float32(-dE/T)) tabulated per sweep
    // (bit-identical decisions, no fp64 divide / exp on the per-update chain); LEAN only
    float *ptab = reinterpret_cast<float *>(smem + sbytes + DENSE_LDS_EXTRA);  // [table_m + 1]
    const bool use_tab = LEAN && a.table_m > 0;

    const int tid = threadIdx.x;
    const int W = blockDim.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = blockIdx.x;
    const int n = a.n;

    if constexpr (BITS) {  // int8 spins in HBM -> one bit per spin in LDS (1 = spin down)
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (long long)r * a.sstride);
        for (int i = tid; i < a.sstride / 32; i += blockDim.x) {
            const int4 lo = src[2 * i], hi = src[2 * i + 1];
            const int wds[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            unsigned int bits = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q)  // sign bit of each of the 4 bytes of a dword
                bits |= (((unsigned)wds[q] >> 7) & 1u) << (4 * q) | (((unsigned)wds[q] >> 15) & 1u) << (4 * q + 1) |
                        (((unsigned)wds[q] >> 23) & 1u) << (4 * q + 2) | (((unsigned)wds[q] >> 31) & 1u) << (4 * q + 3);
            s_bits[i] = bits;
        }
    } else {  // replica spins -> LDS (pad bytes are zero in HBM)
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (long long)r * a.sstride);
        int4 *dst = reinterpret_cast<int4 *>(s_lds);
        for (int i = tid; i < a.sstride / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    // LDS spins back to the int8 HBM layout (bit form: pad spins beyond n stay 0)
    auto store_spins = [&](int8_t *dst_row) {
        if constexpr (BITS) {
            int4 *dst = reinterpret_cast<int4 *>(dst_row);
            for (int i = tid; i < a.sstride / 16; i += blockDim.x) {
                const unsigned int half = (s_bits[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
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
        } else {
            int4 *dst = reinterpret_cast<int4 *>(dst_row);
            const int4 *src = reinterpret_cast<const int4 *>(s_lds);
            for (int i = tid; i < a.sstride / 16; i += blockDim.x) dst[i] = src[i];
        }
    };

    const long long kstride = (long long)W * EPC;  // elements between a wave's chunks
    const int model = a.reps_per_model > 0 ? (int)((a.replica0 + (uint32_t)r) / a.reps_per_model) : 0;
    // element-addressed for fp32 / int8; the bit-plane form addresses bytes (two planes)
    using JE = typename std::conditional<BITS, unsigned char, JT>::type;
    constexpr int LANE_STEP = BITS ? 16 : EPL, CHUNK_STEP = BITS ? 1024 : EPC;
    const long long row_step = BITS ? a.ld / 8 : a.ld;
    const JE *Jlane = reinterpret_cast<const JE *>(a.J) + (BITS ? 0 : model * a.model_stride_j) +
                      (w * CHUNK_STEP + lane * LANE_STEP);
    const float *hvec = a.h + (long long)model * n;
    const float *dvec = a.diag + (long long)model * n;
    const long long kstep = (long long)W * CHUNK_STEP;  // address step between a wave's chunks
    const int8_t *slane = s_lds + (w * CHUNK_STEP + lane * LANE_STEP);
    const bool arith32 = arith == SGA_ARITH_F32;

    constexpr int NBUF = CPW > 0 ? CPW : 1;
    const int cpw_rt = (int)(a.ld / kstride);  // chunks per wave (runtime; = CPW when CPW > 0)

    auto load_chunk = [&](const JE *p, long long k) -> vec_t {
        // default cache policy on purpose: non-temporal loads measured 3-5 % slower here (part
        // of J is re-served by the 256 MB Infinity Cache; profiles/r01_experiments.md)
        if constexpr (BITS) {
            BitPair o;
            o.s = *reinterpret_cast<const int4 *>(p + k * kstep);
            o.z = *reinterpret_cast<const int4 *>(p + a.plane_bytes + k * kstep);
            return o;
        } else {
            return *reinterpret_cast<const vec_t *>(p + k * kstep);
        }
    };

    auto load_row = [&](vec_t(&buf)[NBUF], int site) {
        if constexpr (CPW == 0) return;  // streaming form loads inside the reduction
        const JE *p = Jlane + (long long)site * row_step;
#pragma unroll
        for (int k = 0; k < CPW; ++k) buf[k] = load_chunk(p, k);
    };

    auto accumulate = [&](acc_t &acc, const vec_t &x, long long k) {
        if constexpr (BITS) {  // count the stored couplings whose product with the spin is -1
            const int4 sv = *reinterpret_cast<const int4 *>(slane + k * kstep);
            acc += __builtin_popcount(x.z.x & (x.s.x ^ sv.x)) + __builtin_popcount(x.z.y & (x.s.y ^ sv.y)) +
                   __builtin_popcount(x.z.z & (x.s.z ^ sv.z)) + __builtin_popcount(x.z.w & (x.s.w ^ sv.w));
        } else if constexpr (sizeof(JT) == 4) {  // J * (+-1) is exact in fp32
            const int sw = *reinterpret_cast<const int *>(slane + k * kstride);
            const float s0 = (float)(int8_t)(sw), s1 = (float)(int8_t)(sw >> 8),
                        s2 = (float)(int8_t)(sw >> 16), s3 = (float)(sw >> 24);
            if constexpr (ACC64) {
                acc += (double)(x.x * s0);
                acc += (double)(x.y * s1);
                acc += (double)(x.z * s2);
                acc += (double)(x.w * s3);
            } else {
                acc = __builtin_fmaf(x.x, s0, acc);
                acc = __builtin_fmaf(x.y, s1, acc);
                acc = __builtin_fmaf(x.z, s2, acc);
                acc = __builtin_fmaf(x.w, s3, acc);
            }
        } else {
            const int4 sv = *reinterpret_cast<const int4 *>(slane + k * kstride);
            acc = __builtin_amdgcn_sdot4(x.x, sv.x, acc, false);
            acc = __builtin_amdgcn_sdot4(x.y, sv.y, acc, false);
            acc = __builtin_amdgcn_sdot4(x.z, sv.z, acc, false);
            acc = __builtin_amdgcn_sdot4(x.w, sv.w, acc, false);
        }
    };

    // streaming reduction of row `site`: batches of four 1-KiB chunks per wave
    auto dot_stream = [&](int site) -> acc_t {
        const JE *p = Jlane + (long long)site * row_step;
        acc_t acc = 0;
        for (int k0 = 0; k0 < cpw_rt; k0 += 4) {
            vec_t t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < cpw_rt) t[j] = load_chunk(p, k0 + j);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < cpw_rt) accumulate(acc, t[j], k0 + j);
        }
        return acc;
    };

    auto dot_row = [&](const vec_t(&buf)[NBUF]) -> acc_t {
        acc_t acc = 0;
#pragma unroll
        for (int k = 0; k < NBUF; ++k) accumulate(acc, buf[k], k);
        return acc;
    };

    double E = a.energy[r];
    double bestE = a.best_energy[r];
    unsigned long long nacc = 0;
    int pp = 0;
    double T = 1.0;

    // one Metropolis update at `site` using the row held in `buf`
    auto step = [&](const vec_t(&buf)[NBUF], int site, float u, float h_site, float d_site,
                    long long upd) {
        acc_t lane_sum;
        if constexpr (CPW == 0) lane_sum = dot_stream(site);
        else lane_sum = dot_row(buf);
        acc_t tot = wave_sum(lane_sum);
        const int owner = (site / EPC) % W;
        auto spin_at = [&](int i) -> int {
            if constexpr (BITS) return ((s_bits[i >> 5] >> (i & 31)) & 1u) ? -1 : 1;
            else return s_lds[i];
        };
        int si;
        if (W > 1) {
            acc_t *part = reinterpret_cast<acc_t *>(part_raw + pp * MAX_WAVES * PART_SLOT_BYTES);
            if (lane == 0) {
                *reinterpret_cast<acc_t *>(reinterpret_cast<unsigned char *>(part) +
                                           w * PART_SLOT_BYTES) = tot;
                if (w == owner) sislot[pp] = spin_at(site);
            }
            __syncthreads();
            acc_t s = *reinterpret_cast<acc_t *>(reinterpret_cast<unsigned char *>(part));
            for (int i = 1; i < W; ++i)
                s += *reinterpret_cast<acc_t *>(reinterpret_cast<unsigned char *>(part) +
                                                i * PART_SLOT_BYTES);
            tot = s;
            si = sislot[pp];
            pp ^= 1;
        } else {
            si = spin_at(site);
        }
        // bit-plane form: tot counts the -1 products, d_site carries the row's non-zero count
        const float dotf = BITS ? d_site - 2.0f * (float)tot : (float)tot;
        double dE;
        bool acc;
        if (use_tab) {  // every quantity is an integer: dE = 2 k exactly; k beyond the table
            const float fk = (float)si * (dotf + h_site);  // (rare, large moves) is evaluated
            dE = (double)(2.0f * fk);
            if (fk <= 0.0f) acc = true;
            else if (fk <= (float)a.table_m) acc = u < ptab[(int)fk];
            else acc = u < expf_det((float)(-dE / T));
        } else {
            acc = metropolis_accept(rule, arith, dotf, si, h_site, d_site, T, u, dE);
        }
        if (acc) {
            E += dE;
            ++nacc;
            if (w == owner && lane == 0) {
                if constexpr (BITS) s_bits[site >> 5] ^= 1u << (site & 31);
                else s_lds[site] = (int8_t)(-si);
            }
        }
        if constexpr (!LEAN) {
            if (tid == 0) {
                if (a.accept_trace)
                    a.accept_trace[(long long)r * a.replay_stride + upd] = acc ? 1 : 0;
                if (a.dE_trace)
                    a.dE_trace[(long long)r * a.replay_stride + upd] =
                        acc ? (rule == SGA_RULE_HEAT_BATH ? -dE : dE) : 0.0;
            }
        }
    };

    // Ring of NB row buffers: update g reads ring[g % NB]; its row was requested NB - 1 updates
    // earlier (sites come from the counter RNG, not from the chain's state).  Measured at depths
    // 2, 3 and 6 (profiles/r01_experiments.md): long rows are bandwidth bound and short rows are
    // bound by the update chain itself (LDS round trips + issue rate of one wave per SIMD), not
    // by the HBM round trip, so depth 2 -- the smallest code and register footprint -- is kept.
#ifndef SGA_RING_NB
#define SGA_RING_NB 2
#endif
    constexpr int NB = CPW == 0 ? 1 : SGA_RING_NB;
    struct Slot {
        vec_t row[NBUF];
        int site;
        float u, h, d;
    };
    Slot ring[NB];
    const bool need_d = arith32 || BITS;  // J_ii for the fp32 rule | row non-zero count (bits)

    PairSource<LEAN> rng;
    UpdatePair pairP{0, 0, 2.0f, 2.0f};
    int kP = 0, tP = 0;  // producer cursor: the next update whose row is requested
    auto produce = [&](Slot &sl) {
        // No early-out past the end of the launch: the last NB - 1 requests read rows nobody
        // uses, but a conditional request makes the compiler drain vmcnt at every update.
        const bool second = tP & 1;
        if (!second) pairP = rng.get(a, r, kP, tP >> 1, kP < a.n_sweeps, lane);  // past the end: site 0
        // values first, then the select: a select between the two members' addresses would
        // push the pair into scratch, and scratch loads drain vmcnt -- the whole prefetch ring
        const int sA = pairP.sA, sB = pairP.sB;
        const float uA = pairP.uA, uB = pairP.uB;
        sl.site = second ? sB : sA;
        sl.u = second ? uB : uA;
        load_row(sl.row, sl.site);
        sl.h = hvec[sl.site];
        sl.d = need_d ? dvec[sl.site] : 0.0f;
        if (++tP == n) {
            tP = 0;
            ++kP;
        }
    };
#pragma unroll
    for (int j = 0; j + 1 < NB; ++j) produce(ring[j]);

    const long long total = (long long)a.n_sweeps * n;
    int k = 0, t = 0;  // consumer cursor
    for (long long g0 = 0; g0 < total; g0 += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (g0 + j >= total) break;  // wave-uniform
            if (t == 0) {                // sweep start: temperature, accept table
                T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
                if (use_tab) {
                    __syncthreads();  // nobody still reads last sweep's table
                    for (int q = tid; q <= a.table_m; q += blockDim.x)
                        ptab[q] = expf_det((float)(-(double)(2 * q) / T));
                    __syncthreads();
                }
            }
            produce(ring[(j + NB - 1) % NB]);  // the buffer update g - 1 just released
            step(ring[j].row, ring[j].site, ring[j].u, ring[j].h, ring[j].d, g0 + j);
            if (++t == n) {
                // sweep boundary: energy record, best tracking (annealing/gpu_annealer.py:151-153)
                if (tid == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = E;
                if (E < bestE && !a.no_best) {
                    bestE = E;
                    __syncthreads();
                    store_spins(a.best_spins + (long long)r * a.sstride);
                    __syncthreads();
                }
                t = 0;
                ++k;
            }
        }
    }

    __syncthreads();
    store_spins(a.spins + (long long)r * a.sstride);
    if (tid == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

template <typename JT, bool ACC64, int CPW>
static hipError_t launch_one(const SweepArgs &a, int waves, hipStream_t st) {
    constexpr bool BITS = std::is_same<JT, Tern2>::value;
    const size_t lds = (size_t)(BITS ? a.ld / 8 : a.ld) + DENSE_LDS_EXTRA +
                       sizeof(float) * (size_t)(a.table_m + 1);
    void (*kern)(const SweepArgs);
    if constexpr (BITS) {
        if (!sweep_args_are_lean(a)) return hipErrorInvalidValue;  // engine falls back to int8
        kern = sweep_dense_kernel<JT, CPW, ACC64, true>;
    } else {
        kern = sweep_args_are_lean(a) ? sweep_dense_kernel<JT, CPW, ACC64, true>
                                      : sweep_dense_kernel<JT, CPW, ACC64, false>;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.R), dim3(64 * waves), lds, st, a);
    return hipGetLastError();
}

template <typename JT, bool ACC64>
static hipError_t launch_variant(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    switch (cpw) {
        case 0: return launch_one<JT, ACC64, 0>(a, waves, st);  // streaming form
        case 1: return launch_one<JT, ACC64, 1>(a, waves, st);
        case 2: return launch_one<JT, ACC64, 2>(a, waves, st);
        case 3: return launch_one<JT, ACC64, 3>(a, waves, st);
        case 4: return launch_one<JT, ACC64, 4>(a, waves, st);
        case 5: return launch_one<JT, ACC64, 5>(a, waves, st);
        case 6: return launch_one<JT, ACC64, 6>(a, waves, st);
        case 7: return launch_one<JT, ACC64, 7>(a, waves, st);
        case 8: return launch_one<JT, ACC64, 8>(a, waves, st);
        case 9: return launch_one<JT, ACC64, 9>(a, waves, st);
        case 10: return launch_one<JT, ACC64, 10>(a, waves, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sga
