// sweep_dense_impl.h -- the dense Metropolis sweep kernel (the HBM-bound hot kernel).
//
// Replaces: CUDAKernelManager.metropolis_update_optimized (annealing/cuda_kernels.py:228-282,
// fallback :371-398) and SpinDynamics.sweep (core/spin_dynamics.py:73-94) batched over the
// replicas of ParallelTempering._parallel_sweeps (annealing/parallel_tempering.py:191-203).
//
// Mapping:
//   * one workgroup per replica, W waves (W chosen on the host so R*W waves fill the chip);
//   * a coupling row J[site,:] is cut into 1-KiB chunks (64 lanes x 16 B, one
//     global_load_dwordx4 per wave); wave w owns chunks w, w+W, w+2W, ... -- CPW of them --
//     and keeps them in VGPRs: no LDS round trip for data that is streamed once.  Rows are
//     packed to 128 bytes in HBM; a lane past a row's end re-reads the row's first granule;
//   * the replica's spins live in LDS as int8 (bits in the bit-plane form), padded with zeros
//     to whole chunks; each wave reads only the bytes under its own chunks, and only the owning
//     wave ever rewrites them (no cross-wave hazard);
//   * the single-spin Markov chain is serial, but the SITE sequence is known ahead of time
//     from the counter RNG, so row t+1 is requested into the other slot of a two-slot register
//     ring while row t is reduced: CPW KiB per wave stay in flight across the per-update
//     barrier (plain loads survive s_barrier, the waits are counted vmcnt);
//   * per update: lane partial -> DPP wave sum -> W partials through LDS (one barrier,
//     double-buffered slots) -> every thread evaluates the same accept rule;
//   * BATCH = the look-ahead form for short rows of integer problems: four updates reduced
//     together, the chain replayed on scalars (see the comment at its loop);
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
// Look-ahead form: up to LOOK updates are reduced together, one barrier for all of them.
constexpr int LOOK = 4;
constexpr int LOOK_SLOT_BYTES = 4 * LOOK;  // one wave's LOOK partial sums (float | int)
// behind the spins: [2][MAX_WAVES] partial-sum slots (8 B, or 16 B in the look-ahead form), then
// the spin(s) at the update site(s) published by their owner waves
constexpr int DENSE_LDS_EXTRA = 2 * MAX_WAVES * LOOK_SLOT_BYTES + 2 * LOOK * 4;
// canonical-order kernels keep two buffers of per-super-chunk row sums behind the accept table
__host__ __device__ constexpr long long dense_canon_offset(long long sbytes, int table_m) {
    return (sbytes + DENSE_LDS_EXTRA + 4ll * (table_m + 1) + 7) & ~7ll;
}
__host__ __device__ constexpr long long dense_canon_bytes(long long ld, int epc) {
    return 2 * (ld / epc) * 8;
}

// Look-ahead kernels whose four rows do not fit the 128 VGPRs of a 1024-thread workgroup (4 chunks
// per wave, 2 for bit-planes) are built for at most 4 waves per replica instead.
template <typename JT, int CPW>
constexpr int look_updates() {  // updates per batch of the look-ahead form
    return CPW <= (std::is_same<JT, Tern2>::value ? 2 : 4) ? 4 : 2;
}
template <typename JT, int CPW, bool BATCH>
constexpr int dense_max_threads() {
    return (BATCH && CPW == (std::is_same<JT, Tern2>::value ? 2 : 4)) ? 256 : 1024;
}

// SINGLE = built for one wave per replica: the wave count, the owner-wave tests and the partial-sum
// exchange fold away (small problems run one wave per SIMD and are bound by the length of the
// instruction stream).
// CANON (real-valued fp32 couplings only): the fp64 row sum is formed in the canonical order; without
// it the engine has proved at set time that the fp64 sum is exact, so that the order is free.
template <typename JT, int CPW, bool ACC64, bool LEAN, bool BATCH = false, bool SINGLE = false, bool CANON = false>
__global__ void __launch_bounds__((SINGLE ? 64 : dense_max_threads<JT, CPW, BATCH>())) sweep_dense_kernel(const SweepArgs a) {
    const int rule = a.rule;
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
    const bool use_tab = LEAN && a.table_m > 0 && rule == SGA_RULE_METROPOLIS;
    // real-valued couplings (ACC64): per-chunk sums of the canonical summation order, [2][ld / EPC]
    double *canon = reinterpret_cast<double *>(smem + dense_canon_offset(sbytes, a.table_m));

    const int tid = threadIdx.x;
    const int W = SINGLE ? 1 : (int)(blockDim.x >> 6);
    const int w = SINGLE ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // (a launch over a subset of the replicas -- per-replica routing, sga_kernels.h -- names them in rep_list)
    const int r = a.rep_list ? __builtin_amdgcn_readfirstlane(a.rep_list[blockIdx.x]) : (int)blockIdx.x;
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
    const long long row_step = BITS ? a.plane_row_bytes : a.ldj;
    // A row is addressed as (uniform row base) + (32-bit lane offset): the scalar-base form of
    // global_load needs one shared offset VGPR instead of a 64-bit address pair per load.
    const JE *Jbase = reinterpret_cast<const JE *>(a.J) + (BITS ? 0 : model * a.model_stride_j);
    const unsigned int lane_off = (unsigned int)(w * CHUNK_STEP + lane * LANE_STEP);
    const float *hvec = a.h + (long long)model * n;
    const float *dvec = a.diag + (long long)model * n;
    const long long kstep = (long long)W * CHUNK_STEP;  // address step between a wave's chunks
    const int8_t *slane = s_lds + lane_off;
    // Which chunk is this wave's k-th?  Chunk w + k W everywhere -- except in the canonical-order
    // builds (real-valued J of a wide dynamic range), where a wave owns whole SUPER-CHUNKS of
    // SC = 4 consecutive chunks (1024 elements), so that one tree serves four chunks: its k-th
    // chunk is chunk ((k / 4) W + w) 4 + k % 4.  coff(k): that chunk's offset in address steps.
    constexpr bool SUPER = ACC64 && CANON;
    constexpr int SC = 4;
    static_assert(!SUPER || CPW % SC == 0, "canonical order: whole super-chunks per wave");
    auto coff = [&](unsigned int k) -> unsigned int {
        if constexpr (SUPER) return (((k / SC) * (unsigned int)W + (unsigned int)w) * SC + (k % SC)) * (unsigned int)CHUNK_STEP;
        else return ((unsigned int)w + k * (unsigned int)W) * (unsigned int)CHUNK_STEP;
    };
    const unsigned int lane_part = (unsigned int)(lane * LANE_STEP);
    const bool arith32 = arith == SGA_ARITH_F32;

    // the wave whose chunks hold `site` (chunk c belongs to wave c mod W).  A general modulo costs
    // ~20 instructions per update on a lone wave: multiply by a 16-bit reciprocal instead (exact
    // for c < 4096 chunks and W <= 16)
    const unsigned int w_recip = (65536u + (unsigned int)W - 1u) / (unsigned int)W;
    auto owner_of = [&](int site) -> int {
        if constexpr (SINGLE) return 0;
        const unsigned int c = (unsigned int)site / (unsigned int)(SUPER ? SC * EPC : EPC);
        return (int)(c - ((c * w_recip) >> 16) * (unsigned int)W);
    };

    constexpr int NBUF = CPW > 0 ? CPW : 1;
    const int cpw_rt = (int)(a.ld / kstride);  // chunks per wave (runtime; = CPW when CPW > 0)
    // canonical order: super-chunk slots of the layout / super-chunks that hold elements of a row
    const int n_chunks_ld = (int)(a.ld / (4 * EPC));
    const int n_chunks_row = (n + 4 * EPC - 1) / (4 * EPC);

    // The base of row `site`.  Up to 8 chunks per wave it is pinned to SGPRs and the loads take
    // the scalar-base form (one 32-bit offset VGPR per load instead of a 64-bit address pair);
    // with 9-10 chunks every register is spoken for and the plain per-lane pointer does better.
    constexpr bool SCALAR_BASE = CPW <= 8;
    auto row_base = [&](int site) -> const JE * {
        if constexpr (SCALAR_BASE) {
            // 32 x 32 -> 64 bit: two scalar multiplies (a 64-bit row_step costs five)
            const unsigned long long q = (unsigned long long)(unsigned int)site * (unsigned int)row_step;
            const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)q);
            const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(q >> 32));
            return Jbase + (((unsigned long long)hi << 32) | lo);  // (stays a global pointer)
        } else {
            return Jbase + (long long)site * row_step + lane_off;
        }
    };
    auto load_chunk = [&](const JE *p, int k) -> vec_t {  // p: row_base(site)
        // default cache policy on purpose: non-temporal loads measured 3-5 % slower here (part
        // of J is re-served by the 256 MB Infinity Cache; profiles/r01_experiments.md)
        const unsigned char *q = reinterpret_cast<const unsigned char *>(p);
        // Rows are packed to 128 bytes (16 for the bit-planes), not to whole chunks: a lane past
        // the row's end reads the row's first granule instead -- no extra traffic, and the product
        // with the zero pad spins is zero (bit-planes: its non-zero bits are cleared)
        bool in = true;
        unsigned long long off;
        if constexpr (SCALAR_BASE) {
            // byte offset in 32 bits, opaque to the optimiser (a hoisted 64-bit zero extension of
            // it loses the base + zext(VGPR) address form)
            unsigned int o32 = (lane_part + coff((unsigned int)k)) * (unsigned int)sizeof(JE);
            in = o32 < (unsigned int)row_step * (unsigned int)sizeof(JE);
            o32 = in ? o32 : 0u;
            asm volatile("" : "+v"(o32));
            off = o32;
        } else {
            off = (unsigned long long)((long long)k * kstep) * sizeof(JE);
            in = ((unsigned long long)lane_off * sizeof(JE)) + off < (unsigned long long)row_step * sizeof(JE);
            off = in ? off : 0ull - (unsigned long long)lane_off * sizeof(JE);
        }
        if constexpr (BITS) {
            const int keep = in ? -1 : 0;
            BitPair o;
            o.s = *reinterpret_cast<const int4 *>(q + off);
            o.z = *reinterpret_cast<const int4 *>(q + a.plane_bytes + off);
            o.z.x &= keep, o.z.y &= keep, o.z.z &= keep, o.z.w &= keep;
            return o;
        } else {
            return *reinterpret_cast<const vec_t *>(q + off);
        }
    };

    auto load_row = [&](vec_t(&buf)[NBUF], int site) {
        if constexpr (CPW == 0) return;  // streaming form loads inside the reduction
        const JE *p = row_base(site);
#pragma unroll
        for (int k = 0; k < CPW; ++k) buf[k] = load_chunk(p, k);
    };

    // the spins under chunk k of this wave, and one chunk's contribution to a row sum
    using spin_t = typename std::conditional<sizeof(JT) == 4, int, int4>::type;
    auto load_spins = [&](long long k) -> spin_t {
        const int8_t *sp = SUPER ? s_lds + lane_part + coff((unsigned int)k) : slane + k * kstep;
        if constexpr (BITS) return *reinterpret_cast<const int4 *>(sp);
        else if constexpr (sizeof(JT) == 4) return *reinterpret_cast<const int *>(sp);
        else return *reinterpret_cast<const int4 *>(sp);
    };
    auto accumulate_with = [&](acc_t &acc, const vec_t &x, const spin_t &sv) {
        if constexpr (BITS) {  // count the stored couplings whose product with the spin is -1
            acc += __builtin_popcount(x.z.x & (x.s.x ^ sv.x)) + __builtin_popcount(x.z.y & (x.s.y ^ sv.y)) +
                   __builtin_popcount(x.z.z & (x.s.z ^ sv.z)) + __builtin_popcount(x.z.w & (x.s.w ^ sv.w));
        } else if constexpr (sizeof(JT) == 4) {  // J * (+-1) is exact in fp32
            const int sw = sv;
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
            acc = __builtin_amdgcn_sdot4(x.x, sv.x, acc, false);
            acc = __builtin_amdgcn_sdot4(x.y, sv.y, acc, false);
            acc = __builtin_amdgcn_sdot4(x.z, sv.z, acc, false);
            acc = __builtin_amdgcn_sdot4(x.w, sv.w, acc, false);
        }
    };
    auto accumulate = [&](acc_t &acc, const vec_t &x, long long k) {
        accumulate_with(acc, x, load_spins(k));
    };

    // streaming reduction of row `site`: batches of four 1-KiB chunks per wave
    auto dot_stream = [&](int site) -> acc_t {
        const JE *p = row_base(site);
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
        constexpr bool CANON64 = ACC64 && CANON;
        acc_t tot;
        if constexpr (CANON64) {
            // CANONICAL ORDER for real-valued J: each 1024-element super-chunk is summed by itself --
            // lane l of 64 adds its 16 products (chunk j = 0..3 of the super-chunk, elements 4l..4l+3
            // of each, in that order) from +0, then the adjacent-pairs tree over the 64 lanes
            // (wave_sum) -- and the super-chunk sums are added in order.  Which wave holds a
            // super-chunk, and how many a wave holds, does not enter: the fp64 sum, and with it the
            // fp32 row sum and every decision, is the same for every launch geometry (the CPU
            // checker forms the same sum).
            double *slot = canon + pp * n_chunks_ld;
            if constexpr (CPW == 0) {
                const JE *p = row_base(site);
                for (int k0 = 0; k0 < cpw_rt; k0 += SC) {
                    vec_t t[SC];
#pragma unroll
                    for (int j = 0; j < SC; ++j) t[j] = load_chunk(p, k0 + j);
                    acc_t pl = 0;
#pragma unroll
                    for (int j = 0; j < SC; ++j) accumulate(pl, t[j], k0 + j);
                    const acc_t cs = wave_sum(pl);
                    if (lane == 0) slot[(k0 / SC) * W + w] = cs;
                }
                tot = 0;
            } else {
                constexpr int NS = NBUF / SC > 0 ? NBUF / SC : 1;
                acc_t cs[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    acc_t pl = 0;
#pragma unroll
                    for (int j = 0; j < SC; ++j) accumulate(pl, buf[(q * SC + j) % NBUF], q * SC + j);
                    cs[q] = wave_sum(pl);
                }
                if (W > 1) {
                    if (lane == 0) {
#pragma unroll
                        for (int q = 0; q < NS; ++q) slot[q * W + w] = cs[q];
                    }
                    tot = 0;
                } else {
                    tot = cs[0];
#pragma unroll
                    for (int q = 1; q < NS; ++q) tot += cs[q];
                }
            }
        } else {
            acc_t lane_sum;
            if constexpr (CPW == 0) lane_sum = dot_stream(site);
            else lane_sum = dot_row(buf);
            tot = wave_sum(lane_sum);
        }
        const int owner = owner_of(site);
        auto spin_at = [&](int i) -> int {
            if constexpr (BITS) return ((s_bits[i >> 5] >> (i & 31)) & 1u) ? -1 : 1;
            else return s_lds[i];
        };
        int si;
        if (W > 1) {
            acc_t *part = reinterpret_cast<acc_t *>(part_raw + pp * MAX_WAVES * PART_SLOT_BYTES);
            if (lane == 0) {
                if constexpr (!CANON64)
                    *reinterpret_cast<acc_t *>(reinterpret_cast<unsigned char *>(part) +
                                               w * PART_SLOT_BYTES) = tot;
                if (w == owner) sislot[pp] = spin_at(site);
            }
            __syncthreads();
            if constexpr (CANON64) {  // chunk order; chunks past the row's end hold +0
                const double *slot = canon + pp * n_chunks_ld;
                acc_t s = slot[0];
                for (int c = 1; c < n_chunks_row; ++c) s += slot[c];
                tot = s;
            } else {
                acc_t s = *reinterpret_cast<acc_t *>(reinterpret_cast<unsigned char *>(part));
                for (int i = 1; i < W; ++i)
                    s += *reinterpret_cast<acc_t *>(reinterpret_cast<unsigned char *>(part) +
                                                    i * PART_SLOT_BYTES);
                tot = s;
            }
            si = sislot[pp];
            pp ^= 1;
        } else {
            if constexpr (CANON64 && CPW == 0) {  // one streaming wave: its own LDS writes, in order
                acc_t s = canon[0];
                for (int c = 1; c < n_chunks_row; ++c) s += canon[c];
                tot = s;
            }
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
            else acc = (dE > T * 104.0) ? false : (u < expf_det((float)(-dE / T)));  // beyond the table (p == 0 past -104)
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

    if constexpr (BATCH) {
        // Look-ahead form (integer problems, LEAN).  Short rows are bound by the update chain
        // itself -- three dependent LDS round trips, a wave reduction and ~120 issued
        // instructions per update with one or two waves per SIMD -- not by memory.  The site
        // sequence is known ahead, so LOOK consecutive updates are reduced TOGETHER against the
        // spins as they stand before the first of them (one spin read, LOOK interleaved wave
        // sums, one barrier), and the chain is then replayed serially on scalars: update m's
        // row sum is corrected by -2 J[j_m][j_l] s_l for every accepted earlier update l of the
        // batch, its spin is negated if an accepted l sat on the same site.  All quantities are
        // integers below 2^24, so the corrected sums are exactly the sequential ones and every
        // decision, energy and spin is bit-identical to the one-update-at-a-time form.
        static_assert(LEAN && !ACC64 && CPW >= 1, "look-ahead: production form, exact fp32 / int sums");
        // four updates per batch while four rows fit the registers, two for longer rows
        constexpr int L = look_updates<JT, CPW>(), NX = L * (L - 1) / 2;
        static_assert(L == 2 || L == 4, "one or two RNG pairs per batch");
        unsigned char *lpart = part_raw;                                             // [2][MAX_WAVES][L]
        int *lsi = reinterpret_cast<int *>(part_raw + 2 * MAX_WAVES * LOOK_SLOT_BYTES);  // [2][L]
        struct Meta {  // what a batch needs besides its rows
            int site[L];
            float u[L], h[L], d[L];
            float x[NX];  // J[site[m]][site[l]], l < m, at m (m - 1) / 2 + l
            int cnt;      // updates in the batch (a batch never crosses a sweep boundary)
        };
        vec_t rows[L][NBUF];
        PairSource<LEAN> rng;
        int kP = 0, tP = 0;  // producer cursor: first update of the next batch to request
        auto cross = [&](int sm, int sl) -> float {  // J[sm][sl]: uniform address, one element
            if constexpr (BITS) {
                return (float)reinterpret_cast<const int8_t *>(a.J_aux)[(long long)sm * a.ldj + sl];
            } else {
                return (float)(Jbase + (long long)sm * a.ldj)[sl];
            }
        };
        auto request = [&](Meta &mt) {
            const bool live = kP < a.n_sweeps;
            mt.cnt = live ? (n - tP < L ? n - tP : L) : 0;
            const UpdatePair p0 = rng.get(a, r, kP, tP >> 1, mt.cnt > 0, lane);
            mt.site[0] = p0.sA, mt.site[1] = p0.sB;
            mt.u[0] = p0.uA, mt.u[1] = p0.uB;
            if constexpr (L == 4) {
                const UpdatePair p1 = rng.get(a, r, kP, (tP >> 1) + 1, mt.cnt > 2, lane);
                mt.site[2] = p1.sA, mt.site[3] = p1.sB;
                mt.u[2] = p1.uA, mt.u[3] = p1.uB;
            }
#pragma unroll
            for (int m = 0; m < L; ++m) {
                if (m >= mt.cnt) mt.site[m] = 0;  // the half-used last pair of an odd sweep
                load_row(rows[m], mt.site[m]);
                mt.h[m] = hvec[mt.site[m]];
                mt.d[m] = BITS ? dvec[mt.site[m]] : 0.0f;
            }
#pragma unroll
            for (int m = 1; m < L; ++m)
#pragma unroll
                for (int l = 0; l < m; ++l) mt.x[m * (m - 1) / 2 + l] = cross(mt.site[m], mt.site[l]);
            tP += mt.cnt;
            if (tP >= n) {
                tP = 0;
                ++kP;
            }
        };
        Meta cur;
        request(cur);
        int k = 0, t = 0, lp = 0;
        while (k < a.n_sweeps) {
            if (t == 0) {  // sweep start: temperature, accept table
                T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
                __syncthreads();  // nobody still reads last sweep's table
                for (int q = tid; q <= a.table_m; q += blockDim.x)
                    ptab[q] = expf_det((float)(-(double)(2 * q) / T));
                __syncthreads();
            }
            // spins at the batch's sites, read by their owner waves before anything flips
            int owner[L], si[L];
#pragma unroll
            for (int m = 0; m < L; ++m) {
                owner[m] = owner_of(cur.site[m]);
                si[m] = 0;
                if (w == owner[m]) {
                    if constexpr (BITS) si[m] = ((s_bits[cur.site[m] >> 5] >> (cur.site[m] & 31)) & 1u) ? -1 : 1;
                    else si[m] = s_lds[cur.site[m]];
                }
            }
            // LOOK row sums against the same spins
            acc_t ls[L];
#pragma unroll
            for (int m = 0; m < L; ++m) ls[m] = 0;
#pragma unroll
            for (int c = 0; c < NBUF; ++c) {
                const spin_t sv = load_spins(c);
#pragma unroll
                for (int m = 0; m < L; ++m) accumulate_with(ls[m], rows[m][c], sv);
            }
            // the rows are consumed: request the next batch into the same registers
            Meta nxt;
            request(nxt);
            acc_t tot[L];
#pragma unroll
            for (int m = 0; m < L; ++m) tot[m] = wave_sum(ls[m]);
            if (W > 1) {
                acc_t *mine = reinterpret_cast<acc_t *>(lpart + (lp * MAX_WAVES + w) * LOOK_SLOT_BYTES);
                if (lane == 0) {
#pragma unroll
                    for (int m = 0; m < L; ++m) {
                        mine[m] = tot[m];
                        if (w == owner[m]) lsi[lp * L + m] = si[m];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < L; ++m) tot[m] = 0;
                for (int i = 0; i < W; ++i) {
                    const acc_t *o = reinterpret_cast<const acc_t *>(lpart + (lp * MAX_WAVES + i) * LOOK_SLOT_BYTES);
#pragma unroll
                    for (int m = 0; m < L; ++m) tot[m] += o[m];
                }
#pragma unroll
                for (int m = 0; m < L; ++m) si[m] = lsi[lp * L + m];
                lp ^= 1;
            }
            // the chain, replayed on wave-uniform scalars
            bool took[L];
            float sused[L];
#pragma unroll
            for (int m = 0; m < L; ++m) {
                took[m] = false;
                sused[m] = 0.0f;
                if (m < cur.cnt) {
                    // bit-plane form: tot counts the -1 products, d carries the row's non-zero count
                    float dotf = BITS ? cur.d[m] - 2.0f * (float)tot[m] : (float)tot[m];
                    int sim = si[m];
#pragma unroll
                    for (int l = 0; l < m; ++l) {
                        if (took[l]) {
                            dotf -= 2.0f * cur.x[m * (m - 1) / 2 + l] * sused[l];
                            if (cur.site[l] == cur.site[m]) sim = -sim;
                        }
                    }
                    const float fk = (float)sim * (dotf + cur.h[m]);
                    const double dE = (double)(2.0f * fk);
                    bool acc;
                    if (fk <= 0.0f) acc = true;
                    else if (fk <= (float)a.table_m) acc = cur.u[m] < ptab[(int)fk];
                    else acc = (dE > T * 104.0) ? false : (cur.u[m] < expf_det((float)(-dE / T)));  // beyond the table (p == 0 past -104)
                    if (acc) {
                        E += dE;
                        ++nacc;
                        if (w == owner[m] && lane == 0) {
                            if constexpr (BITS) s_bits[cur.site[m] >> 5] ^= 1u << (cur.site[m] & 31);
                            else s_lds[cur.site[m]] = (int8_t)(-sim);
                        }
                    }
                    took[m] = acc;
                    sused[m] = (float)sim;
                }
            }
            t += cur.cnt;
            if (t >= n) {
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
            cur = nxt;
        }
    } else {
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
    }

    __syncthreads();
    store_spins(a.spins + (long long)r * a.sstride);
    if (tid == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

// look-ahead form: short rows only (longer ones are bandwidth bound, and LOOK of them would not
// fit the registers)
template <typename JT, bool ACC64, int CPW>
constexpr bool has_look_ahead() {  // keep dense_look_ahead() (sweep_dense_i8.hip) in step
    return !ACC64 && CPW >= 1 && CPW <= (std::is_same<JT, Tern2>::value ? 4 : 6);
}

template <typename JT, bool ACC64, int CPW, bool CANON = false>
static hipError_t launch_one(const SweepArgs &a, int waves, hipStream_t st) {
    if constexpr (ACC64 && CANON && CPW % 4 != 0) {
        return hipErrorInvalidValue;  // canonical order: whole 4-chunk super-chunks per wave (engine's geometry)
    } else {
    constexpr bool BITS = std::is_same<JT, Tern2>::value;
    const size_t lds = CANON ? (size_t)(dense_canon_offset(a.ld, a.table_m) +
                                        dense_canon_bytes(a.ld, 4 * 64 * JTraits<JT>::EPL))
                             : (size_t)(BITS ? a.ld / 8 : a.ld) + DENSE_LDS_EXTRA +
                                   sizeof(float) * (size_t)(a.table_m + 1);
    const bool lean = sweep_args_are_lean(a);
    void (*kern)(const SweepArgs) = nullptr;
    bool is_batch = false, is_single = false, is_lean = lean;  // what gets launched (note_sweep_kernel)
    if constexpr (has_look_ahead<JT, ACC64, CPW>()) {
        // the 256-thread builds use up to 170 VGPRs = 3 waves per SIMD: only while the launch does
        // not want more than that (n = 1024 fp32: +31 % at 1024 replicas, -5 % at 8192)
        constexpr bool fat = dense_max_threads<JT, CPW, true>() < 1024;
        if (lean && a.rule == SGA_RULE_METROPOLIS && a.table_m > 0 && a.look_ahead && (!BITS || a.J_aux) &&
            (!fat || (waves <= 4 && (long long)(a.rep_list ? a.rep_count : a.R) * waves <= 3 * 1024))) {
            kern = waves == 1 ? sweep_dense_kernel<JT, CPW, ACC64, true, true, true>
                              : sweep_dense_kernel<JT, CPW, ACC64, true, true>;
            is_batch = true;
            is_single = waves == 1;
        }
    }
    if (!kern) {
        if constexpr (BITS) {
            if (!lean || a.rule != SGA_RULE_METROPOLIS) return hipErrorInvalidValue;  // engine: int8 copy
            kern = sweep_dense_kernel<JT, CPW, ACC64, true>;
        } else {
            kern = lean ? sweep_dense_kernel<JT, CPW, ACC64, true, false, false, CANON>
                        : sweep_dense_kernel<JT, CPW, ACC64, false, false, false, CANON>;
            // real-valued small problems: the one-wave build of the one-update-at-a-time form
            if constexpr (CPW >= 1 && CPW <= 4) {
                if (lean && waves == 1) {
                    kern = sweep_dense_kernel<JT, CPW, ACC64, true, false, true, CANON>;
                    is_single = true;
                }
            }
        }
    }
    {
        hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.rep_list ? a.rep_count : a.R), dim3(64 * waves), lds, st, a);
    note_sweep_kernel("sweep_dense_kernel<%s, CPW=%d, ACC64=%d, LEAN=%d, BATCH=%d, SINGLE=%d, CANON=%d> x %d wave(s)",
                      BITS ? "Tern2" : (sizeof(JT) == 4 ? "float" : "int8_t"), CPW, (int)ACC64, (int)is_lean,
                      (int)is_batch, (int)is_single, (int)(CANON && !is_batch), waves);
    return hipGetLastError();
    }
}

template <typename JT, bool ACC64, bool CANON = false>
static hipError_t launch_variant(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    switch (cpw) {
        case 0: return launch_one<JT, ACC64, 0, CANON>(a, waves, st);  // streaming form
        case 1: return launch_one<JT, ACC64, 1, CANON>(a, waves, st);
        case 2: return launch_one<JT, ACC64, 2, CANON>(a, waves, st);
        case 3: return launch_one<JT, ACC64, 3, CANON>(a, waves, st);
        case 4: return launch_one<JT, ACC64, 4, CANON>(a, waves, st);
        case 5: return launch_one<JT, ACC64, 5, CANON>(a, waves, st);
        case 6: return launch_one<JT, ACC64, 6, CANON>(a, waves, st);
        case 7: return launch_one<JT, ACC64, 7, CANON>(a, waves, st);
        case 8: return launch_one<JT, ACC64, 8, CANON>(a, waves, st);
        case 9: return launch_one<JT, ACC64, 9, CANON>(a, waves, st);
        case 10: return launch_one<JT, ACC64, 10, CANON>(a, waves, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sga
