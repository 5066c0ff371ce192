// sweep_clf_impl.h -- the CACHED-LOCAL-FIELD sweep: the same single-spin chain as the dense sweep
// kernel (sweep_dense_impl.h), with a coupling row read only when a proposal is ACCEPTED.
//
// Replaces the same reference code as the dense sweep -- SpinDynamics.sweep /
// _metropolis_update (core/spin_dynamics.py:73-94,131-152) batched over replicas -- in the way
// the reference's own incremental mode evaluates moves (core/energy_computer.py:166-173,
// 262-265: dE from a maintained field, the field updated on a flip).
//
// For integer-valued symmetric couplings with a zero diagonal every replica keeps its local fields
//     F_i = scale * (sum_j J_ij s_j + h_i)        (int16 | int32, exact; scale = 2 for half-integer h)
// resident in LDS next to its spins (one bit each).  A proposal at site i needs F_i and s_i only:
// dE = 2 s_i F_i / scale.  The site and the uniform of every update come from the counter RNG, not
// from the chain's state, and a REJECTED proposal leaves the state untouched -- so a window of 128
// consecutive updates is evaluated at once, lane l taking updates 2l and 2l + 1 against the spins
// and fields as they stand; a wave ballot finds the first accepted one, everything before it is
// rejected for good, the flip is applied (row i of J streamed once:  F_j -= 2 scale J_ij s_i),
// and the candidates behind it are evaluated again.  Decisions, energies and spins are those of
// the one-update-at-a-time chain bit for bit (all quantities are integers below 2^24; the accept
// rule is the same function of the same arguments), for every site mode, rule and arithmetic the
// dense kernel serves -- the tests run it against the same oracle and reference fixtures.
//
// Byte model (its own, reported beside the graded one-row-per-proposal figure, never instead of
// it): B = acceptance rate x n x sizeof(J element) per attempt, SURVEY.md 8(d) last sentence.
//
// Mapping: one workgroup per replica, W waves (1 ... 8).  Wave w evaluates the 128 updates of ITS window
// of a W x 128-update super-window and publishes its first two accepting candidates (position, site, dE,
// the spin at the first site) in an LDS slot; after one barrier every wave reads the slots and takes the
// earliest accept -- the decision is the same in all waves without further communication.  The row of the
// accepted site is dealt to the waves in 1-KiB chunks (chunk c -> wave c mod W), each wave updating the
// fields under its chunks; a second barrier makes the new state visible and the candidates behind the
// accept are evaluated again.  One row request per round: the row of the PREDICTED next accept (the second
// accepting candidate) travels while the current accept is applied and the next round is evaluated; the two
// row buffers take turns, so a predicted row is never copied or waited for before the round that uses it.
// What must hold between the waves: nobody applies before everybody has evaluated (barrier A), nobody
// evaluates before everybody has applied and wave 0 has flipped the spin (barrier B), and the spin at
// the accepted site comes from the EVALUATION (before barrier A: wave 0 flips it right after its own share
// of the row).
// Cost (profiles/r03_experiments.md 1, profiles/clf_cold.py): a round = one accept is a serial chain of
// ~230 issued instructions per wave, two barriers and three dependent LDS round trips ~ 1.1 us for a
// replica alone on its CU, ~1.7 us with four busy replicas per CU; a launch lasts as long as its replica
// with the most accepts.  Proposals that are all rejected cost their Philox draws (18 v_mad_u64_u32 per
// two proposals, the bound of the accept-free sweep: 25 us per 10 000-spin sweep at 4 replicas per CU =
// 4.0e11 attempts/s).
#pragma once
#include <type_traits>

#include "sweep_common.h"

namespace sga {

constexpr int CLF_WINDOW = 128;  // updates evaluated together: two per lane
constexpr int CLF_BATCH_MAX = 5; // row chunks a wave requests together (kernel builds for 3 and 5)
constexpr int CLF_SLOT_INTS = 8;  // what a wave publishes per round: 2 positions, 2 sites, dE, the spin at the first site
constexpr int CLF_MAX_WAVES = 8; // waves per replica (512 threads: up to 256 VGPRs for the two row buffers)

// The accept rule as a function of the exact local field (sum + h): what metropolis_accept
// (sweep_common.h) computes from (double)dot + (double)h -- the diagonal term of the fp32 operator
// arithmetic is zero here (the cached-field form needs a zero diagonal).
__device__ __forceinline__ bool field_rule_accept(int rule, int arith, double field, int si, double T,
                                                  float u, double &dE) {
    if (rule != SGA_RULE_METROPOLIS) {  // core/spin_dynamics.py:154-171, :173-191
        const float x = (rule == SGA_RULE_GLAUBER) ? (float)(-2.0 * field / T)
                                                   : (float)((-2.0 * (1.0 / T)) * field);
        const float prob_up = 1.0f / (1.0f + expf_det(x));
        const int new_spin = (u < prob_up) ? 1 : -1;
        dE = 2.0 * (double)si * field;
        return new_spin != si;
    }
    if (arith == SGA_ARITH_F64) {  // core/spin_dynamics.py:131-152
        dE = 2.0 * (double)si * field;
        if (dE <= 0.0) return true;
        if (dE > T * 104.0) return false;
        return u < expf_det((float)(-dE / T));
    }
    // annealing/cuda_kernels.py:383-390, fp32 throughout
    const float sif = (float)si;
    const float dEf = (2.0f * sif) * (float)field;
    dE = (double)dEf;
    if (dEf <= 0.0f) return true;
    return u < expf_det(-dEf / (float)T);
}

// two int16 in a dword, each increased by its own (small) amount, wrapping separately
__device__ __forceinline__ int add_pair(int pair, int d_lo, int d_hi) {
    const unsigned int p = (unsigned int)pair;
    return (int)(((p + (unsigned int)d_lo) & 0xFFFFu) | ((p + ((unsigned int)d_hi << 16)) & 0xFFFF0000u));
}

// F_j -= 2 scale J_ij s_i for the EPL = 16 / sizeof(JT) couplings x = J[i][j0 .. j0 + EPL) of one lane, in three
// steps -- the lane's EPL fields LDS -> registers, the update, registers -> LDS -- so that a caller with several
// chunks per row can read all their fields first and pay the LDS round trip once per row (clf_apply_chunk does one
// chunk from end to end).  NEG = the amount is subtracted (mult < 0): a compile-time constant, the caller branches
// once per accept on the wave-uniform sign (left to the compiler the select costs two more VALU per int16 pair).
// mult = -2 scale s_i(old).
template <typename JT, typename FT>
struct ClfFields {
    static constexpr int N = (16 / (int)sizeof(JT)) * (int)sizeof(FT) / 4;  // dwords: 2 | 4 | 8 | 16
    int v[N];
};
template <typename JT, typename FT>
__device__ __forceinline__ ClfFields<JT, FT> clf_fields_load(const FT *F, long long j0) {
    ClfFields<JT, FT> f;
    if constexpr (ClfFields<JT, FT>::N == 2) {
        const int2 t = *reinterpret_cast<const int2 *>(F + j0);
        f.v[0] = t.x, f.v[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < ClfFields<JT, FT>::N / 4; ++i) {
            const int4 t = *reinterpret_cast<const int4 *>(reinterpret_cast<const unsigned char *>(F + j0) + 16 * i);
            f.v[4 * i] = t.x, f.v[4 * i + 1] = t.y, f.v[4 * i + 2] = t.z, f.v[4 * i + 3] = t.w;
        }
    }
    return f;
}
template <typename JT, typename FT>
__device__ __forceinline__ void clf_fields_store(FT *F, long long j0, const ClfFields<JT, FT> &f) {
    if constexpr (ClfFields<JT, FT>::N == 2) {
        *reinterpret_cast<int2 *>(F + j0) = make_int2(f.v[0], f.v[1]);
    } else {
#pragma unroll
        for (int i = 0; i < ClfFields<JT, FT>::N / 4; ++i)
            *reinterpret_cast<int4 *>(reinterpret_cast<unsigned char *>(F + j0) + 16 * i) =
                make_int4(f.v[4 * i], f.v[4 * i + 1], f.v[4 * i + 2], f.v[4 * i + 3]);
    }
}
template <typename JT, typename FT, bool NEG, typename VEC>
__device__ __forceinline__ void clf_fields_update(ClfFields<JT, FT> &f, const VEC &x, int mult, int sc) {
    constexpr int FB = (int)sizeof(FT);
    if constexpr (sizeof(JT) == 4) {
        const int d0 = mult * (int)x.x, d1 = mult * (int)x.y, d2 = mult * (int)x.z, d3 = mult * (int)x.w;
        if constexpr (FB == 2) {
            f.v[0] = add_pair(f.v[0], d0, d1);
            f.v[1] = add_pair(f.v[1], d2, d3);
        } else {
            f.v[0] += d0, f.v[1] += d1, f.v[2] += d2, f.v[3] += d3;
        }
    } else {
        const int wds[4] = {x.x, x.y, x.z, x.w};
        if constexpr (FB == 2) {
            // Four couplings per dword -> two dwords of int16 pairs, in packed 16-bit arithmetic:
            // v_perm_b32 puts a coupling into the HIGH byte of each half (b << 8), one packed
            // arithmetic shift right by 8 - log2 |mult| makes it |mult| * b, one packed add or subtract
            // applies it: 3 instructions per pair.  |mult| = 2 scale is 2 or 4.
            typedef short short2v __attribute__((ext_vector_type(2)));
            const short sh = (short)(sc == 1 ? 7 : 6);
            const short2v shv = {sh, sh};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned int t01 = __builtin_amdgcn_perm((unsigned int)wds[d], 0u, 0x050c040cu);
                const unsigned int t23 = __builtin_amdgcn_perm((unsigned int)wds[d], 0u, 0x070c060cu);
                const short2v v01 = __builtin_bit_cast(short2v, t01) >> shv;
                const short2v v23 = __builtin_bit_cast(short2v, t23) >> shv;
                short2v lo = __builtin_bit_cast(short2v, f.v[2 * d]), hi = __builtin_bit_cast(short2v, f.v[2 * d + 1]);
                if constexpr (NEG) lo -= v01, hi -= v23;
                else lo += v01, hi += v23;
                f.v[2 * d] = __builtin_bit_cast(int, lo);
                f.v[2 * d + 1] = __builtin_bit_cast(int, hi);
            }
        } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                f.v[4 * d] += mult * (int)(int8_t)(wds[d]);
                f.v[4 * d + 1] += mult * (int)(int8_t)(wds[d] >> 8);
                f.v[4 * d + 2] += mult * (int)(int8_t)(wds[d] >> 16);
                f.v[4 * d + 3] += mult * (wds[d] >> 24);
            }
        }
    }
}
// ... the same update with the sign taken from mult at run time (several rows into one set of fields): int8 couplings
// into int16 pairs as one packed multiply-add per pair (v_perm_b32 puts a coupling into the high byte of each half,
// one packed arithmetic shift right by 8 sign-extends it, v_pk_mad_i16 adds mult times it); the other types
// multiply by mult anyway
template <typename JT, typename FT, typename VEC>
__device__ __forceinline__ void clf_fields_update_signed(ClfFields<JT, FT> &f, const VEC &x, int mult, int sc) {
    if constexpr (sizeof(JT) == 1 && sizeof(FT) == 2) {
        typedef short short2v __attribute__((ext_vector_type(2)));
        const short2v sh8 = {8, 8};
        const int mm = (mult & 0xFFFF) | (mult << 16);  // mult in both halves
        const int wds[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned int t01 = __builtin_amdgcn_perm((unsigned int)wds[d], 0u, 0x050c040cu);
            const unsigned int t23 = __builtin_amdgcn_perm((unsigned int)wds[d], 0u, 0x070c060cu);
            const int v01 = __builtin_bit_cast(int, __builtin_bit_cast(short2v, t01) >> sh8);
            const int v23 = __builtin_bit_cast(int, __builtin_bit_cast(short2v, t23) >> sh8);
            // (the compiler splits a v2i16 multiply-add into v_pk_mul_lo_u16 + v_pk_add_u16)
            asm("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(f.v[2 * d]) : "v"(v01), "s"(mm));
            asm("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(f.v[2 * d + 1]) : "v"(v23), "s"(mm));
        }
    } else {
        clf_fields_update<JT, FT, false>(f, x, mult, sc);  // (these forms do not look at NEG)
    }
}
template <typename JT, typename FT, bool NEG, typename VEC>
__device__ __forceinline__ void clf_apply_chunk(FT *F, const VEC &x, long long j0, int mult, int sc) {
    ClfFields<JT, FT> f = clf_fields_load<JT, FT>(F, j0);
    clf_fields_update<JT, FT, NEG>(f, x, mult, sc);
    clf_fields_store<JT, FT>(F, j0, f);
}
// a wave's whole first batch of chunks, all of them full (no lane guard): fields of every chunk read first
template <typename JT, typename FT, bool NEG, int NCH, bool UPFRONT = true, typename VEC, typename ELEM0>
__device__ __forceinline__ void clf_apply_full_chunks(FT *F, const VEC *x, ELEM0 elem_of, int mult, int sc) {
    if constexpr (UPFRONT && ClfFields<JT, FT>::N * NCH <= 24) {
        ClfFields<JT, FT> f[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) f[q] = clf_fields_load<JT, FT>(F, elem_of(q));
#pragma unroll
        for (int q = 0; q < NCH; ++q) clf_fields_update<JT, FT, NEG>(f[q], x[q], mult, sc);
#pragma unroll
        for (int q = 0; q < NCH; ++q) clf_fields_store<JT, FT>(F, elem_of(q), f[q]);
    } else {  // (too many registers to hold at once: chunk by chunk)
#pragma unroll
        for (int q = 0; q < NCH; ++q) clf_apply_chunk<JT, FT, NEG>(F, x[q], elem_of(q), mult, sc);
    }
}

// LDS of one replica: fields [ldf] FT | spin bits [sstride / 8 bytes] | accept table [table_m + 1] floats
__host__ __device__ constexpr long long clf_bits_offset(long long ldf, int fbytes) {
    return (ldf * fbytes + 15) & ~15ll;
}
__host__ __device__ constexpr long long clf_table_offset(long long ldf, int fbytes, int sstride) {
    return clf_bits_offset(ldf, fbytes) + ((sstride / 8 + 15) & ~15);
}
inline size_t clf_lds_bytes(long long ldf, int fbytes, int sstride, int table_m) {  // + [2][CLF_MAX_WAVES][CLF_SLOT_INTS] decision slots
    return (size_t)clf_table_offset(ldf, fbytes, sstride) + sizeof(float) * (size_t)((table_m + 4) & ~3) + 2 * 4 * CLF_SLOT_INTS * CLF_MAX_WAVES + 16;
}

// TAIL: rows longer than CLF_BATCH chunks per wave (the rest is streamed inside the field update); built
// without it the round holds no other global load than the two row requests, and their waits stay counted
template <typename JT, typename FT, bool LEAN, int CLF_BATCH = CLF_BATCH_MAX, bool TAIL = true>
__global__ void __launch_bounds__(64 * CLF_MAX_WAVES) sweep_clf_kernel(const SweepArgs a) {
    constexpr int EPL = 16 / (int)sizeof(JT), EPC = 64 * EPL;  // elements per lane / per 1-KiB chunk
    constexpr int FB = (int)sizeof(FT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FT *F = reinterpret_cast<FT *>(smem);
    unsigned int *bits = reinterpret_cast<unsigned int *>(smem + clf_bits_offset(a.ldf, FB));
    float *ptab = reinterpret_cast<float *>(smem + clf_table_offset(a.ldf, FB, a.sstride));

    const int tid = threadIdx.x, lane = tid & 63;
    const int W = (int)(blockDim.x >> 6);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (a launch over a subset of the replicas -- per-replica routing, sga_kernels.h -- names them in rep_list)
    const int r = a.rep_list ? __builtin_amdgcn_readfirstlane(a.rep_list[blockIdx.x]) : (int)blockIdx.x, n = a.n;
    const int rule = LEAN ? SGA_RULE_METROPOLIS : a.rule;
    const int arith = LEAN ? SGA_ARITH_F64 : a.arith;
    const int sc = a.field_scale;
    const double inv_sc = 1.0 / (double)sc;  // 1 | 0.5: exact

    {   // resident state -> LDS
        const int4 *src = reinterpret_cast<const int4 *>(reinterpret_cast<const FT *>(a.fields) + (long long)r * a.ldf);
        int4 *dst = reinterpret_cast<int4 *>(F);
        for (int i = tid; i < (int)(a.ldf * FB / 16); i += blockDim.x) dst[i] = src[i];
        spins_to_bits(a.spins + (long long)r * a.sstride, bits, a.sstride, tid, blockDim.x);
    }
    __syncthreads();

    const JT *Jbase = reinterpret_cast<const JT *>(a.J);
    const int n_chunks = (int)((a.ldj + EPC - 1) / EPC);
    double E = a.energy[r], bestE = a.best_energy[r];
    unsigned long long nacc = 0;
    double T = 1.0;

    // A row is dealt to the waves in 1-KiB chunks: chunk c -> wave c mod W.  The first CLF_BATCH chunks of
    // a wave are requested together into registers (row_request: unconditional loads -- a lane past the
    // row's end reads the row's first granule -- so that requests for a PREDICTED accept can stay in
    // flight across the evaluation of the next round with counted waits); rows longer than
    // CLF_BATCH * W chunks stream the rest inside apply_row.
    using vec_t = typename std::conditional<sizeof(JT) == 4, float4, int4>::type;
    struct RowRegs {
        vec_t x[CLF_BATCH];
    };
    auto elem0 = [&](int c) -> long long { return ((long long)c * 64 + lane) * EPL; };
    // this lane's byte offset into a row for each chunk of the first batch (a lane past the row's end reads
    // the row's first granule): 32-bit, computed once -- a request is then the scalar-base form of
    // global_load (uniform row base in SGPRs + one offset VGPR), no address arithmetic per load
    unsigned int req_off[CLF_BATCH];
#pragma unroll
    for (int q = 0; q < CLF_BATCH; ++q) {
        const long long j0 = elem0(w + q * W);
        req_off[q] = (unsigned int)((j0 < a.ldj ? j0 : 0) * (long long)sizeof(JT));
    }
    const unsigned int pitch = (unsigned int)(a.ldj * (long long)sizeof(JT));
    auto row_request = [&](int site) -> RowRegs {
        RowRegs o;
        // wave-uniform; a row's pitch is below 4 GB, 32 x 32 -> 64 bit is the whole product
        const unsigned char *row = reinterpret_cast<const unsigned char *>(Jbase) + (unsigned long long)(unsigned int)site * pitch;
#pragma unroll
        for (int q = 0; q < CLF_BATCH; ++q) {
            unsigned int off = req_off[q];
            asm volatile("" : "+v"(off));  // (kept a 32-bit value: folded into 64-bit pointer arithmetic the form is lost)
            o.x[q] = *reinterpret_cast<const vec_t *>(row + off);
        }
        return o;
    };
    auto apply_chunk = [&](const vec_t &x, long long j0, int mult /* -2 scale s_i(old) */, auto neg) {
        clf_apply_chunk<JT, FT, decltype(neg)::value>(F, x, j0, mult, sc);
    };
    // chunks of this wave's first batch that lie fully inside the row (wave-uniform): their field update needs no lane
    // guard, and without the guards a row's updates are one straight line with the field reads up front
    int nfull = 0;
#pragma unroll
    for (int q = 0; q < CLF_BATCH; ++q)
        if ((long long)(w + q * W + 1) * EPC <= a.ldj) nfull = q + 1;
    auto first_chunk = [&](int q) -> long long { return elem0(w + q * W); };
    auto apply_row_signed = [&](const RowRegs &rr, int site, int mult, auto neg) {
        if (!TAIL && nfull == CLF_BATCH) {  // (the long-row build keeps the guarded form throughout)
            clf_apply_full_chunks<JT, FT, decltype(neg)::value, CLF_BATCH>(F, rr.x, first_chunk, mult, sc);
        } else if (!TAIL && nfull == CLF_BATCH - 1) {
            clf_apply_full_chunks<JT, FT, decltype(neg)::value, CLF_BATCH - 1>(F, rr.x, first_chunk, mult, sc);
            const long long j0 = elem0(w + (CLF_BATCH - 1) * W);
            if (j0 < a.ldj) apply_chunk(rr.x[CLF_BATCH - 1], j0, mult, neg);
        } else {
#pragma unroll
            for (int q = 0; q < CLF_BATCH; ++q) {
                const long long j0 = elem0(w + q * W);
                if (j0 < a.ldj) apply_chunk(rr.x[q], j0, mult, neg);
            }
        }
        if constexpr (!TAIL) return;
        const JT *row = Jbase + (long long)site * a.ldj;
        for (int c0 = w + CLF_BATCH * W; c0 < n_chunks; c0 += CLF_BATCH * W) {  // (long rows only)
            vec_t x[CLF_BATCH];
#pragma unroll
            for (int q = 0; q < CLF_BATCH; ++q) {
                const long long j0 = elem0(c0 + q * W);
                x[q] = *reinterpret_cast<const vec_t *>(row + (j0 < a.ldj ? j0 : 0));
            }
#pragma unroll
            for (int q = 0; q < CLF_BATCH; ++q) {
                const long long j0 = elem0(c0 + q * W);
                if (j0 < a.ldj) apply_chunk(x[q], j0, mult, neg);
            }
        }
    };
    auto apply_row = [&](const RowRegs &rr, int site, int mult) {
        if (mult < 0) apply_row_signed(rr, site, mult, std::true_type{});  // wave-uniform
        else apply_row_signed(rr, site, mult, std::false_type{});
    };

    // One candidate against the current state: does it flip?  The production build returns k = s_i F_i
    // (dE = 2 k / scale is formed once, for the accepted candidate) and is branch free -- the table covers
    // k <= table_m (entry 0 = 1 serves every downhill move: u < 1 always), the few moves beyond it are
    // evaluated behind a wave-uniform test; the general build returns dE from the reference's arithmetic.
    struct Verdict {
        int k;      // LEAN
        double dE;  // !LEAN
        int si;     // the spin at the candidate's site
    };
    // Both candidates of a lane at once: their field / spin-word reads travel together, then their two table
    // reads (two dependent LDS round trips per round instead of four).
    auto decide2 = [&](int sA, float uA, bool liveA, Verdict &vA, int sB, float uB, bool liveB, Verdict &vB, bool &fA,
                       bool &fB) {
        const int fa = (int)F[sA], fb = (int)F[sB];
        const unsigned int wa = bits[sA >> 5], wb = bits[sB >> 5];
        vA.si = ((wa >> (sA & 31)) & 1u) ? -1 : 1;
        vB.si = ((wb >> (sB & 31)) & 1u) ? -1 : 1;
        if constexpr (LEAN) {
            const int ka = vA.si * fa, kb = vB.si * fb;
            vA.k = ka, vB.k = kb;
            const float pa = ptab[min(max(ka, 0), a.table_m)], pb = ptab[min(max(kb, 0), a.table_m)];
            bool accA = uA < pa, accB = uB < pb;
            const bool beyondA = liveA && ka > a.table_m, beyondB = liveB && kb > a.table_m;
            if (ballot64(beyondA || beyondB)) {  // rare: large uphill moves (p == 0 past -104, sweep_common.h)
                const double dA = (double)(2 * ka) * inv_sc, dB = (double)(2 * kb) * inv_sc;
                if (beyondA) accA = !(dA > T * 104.0) && uA < expf_det((float)(-dA / T));
                if (beyondB) accB = !(dB > T * 104.0) && uB < expf_det((float)(-dB / T));
            }
            fA = liveA && accA, fB = liveB && accB;
        } else {
            fA = liveA && field_rule_accept(rule, arith, (double)fa * inv_sc, vA.si, T, uA, vA.dE);
            fB = liveB && field_rule_accept(rule, arith, (double)fb * inv_sc, vB.si, T, uB, vB.dE);
        }
    };
    // what a wave found in its window, for the other waves: [2][CLF_MAX_WAVES][CLF_SLOT_INTS] ints behind the accept
    // table, the two halves taking turns (a wave may run one round ahead of a wave still reading)
    int *slots2 = reinterpret_cast<int *>(ptab + ((a.table_m + 4) & ~3));  // (16-byte aligned: a slot is two int4)
    int turn = 0;
    constexpr int NONE = 1 << 20;

    for (int k = 0; k < a.n_sweeps; ++k) {
        T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        if constexpr (LEAN) {  // exp(float32(-dE/T)) of the moves dE = 2 q / scale, q <= table_m
            __syncthreads();
            for (int q = tid; q <= a.table_m; q += blockDim.x)
                ptab[q] = expf_det((float)(-((double)(2 * q) * inv_sc) / T));
            __syncthreads();
        }
        const long long base = (long long)r * a.replay_stride + (long long)k * n;
        // W consecutive windows at a time, one per wave: positions [0, 128 W) of the super-window
        for (int t0 = 0; t0 < n; t0 += CLF_WINDOW * W) {
            // this lane's two candidates: updates tA and tA + 1 of sweep k
            const int gA = w * CLF_WINDOW + 2 * lane, gB = gA + 1;  // positions in the super-window
            const int tA = t0 + gA, tB = tA + 1;
            const bool vA = tA < n, vB = tB < n;
            int sA = 0, sB = 0;
            float uA = 2.0f, uB = 2.0f;
            if (LEAN || a.site_mode != SGA_SITE_REPLAY) {
                // (the key made opaque here: its ten round values are then formed by scalar adds on the spot; hoisted
                //  out of the sweep loop they are spilled and come back through one v_readlane each)
                uint32_t key_lo = a.seed_lo, key_hi = a.seed_hi;
                asm volatile("" : "+s"(key_lo), "+s"(key_hi));
                const u32x4 x = philox4x32_10((uint32_t)(tA >> 1), a.sweep0 + (uint32_t)k, a.replica0 + (uint32_t)r,
                                              DOMAIN_SWEEP, key_lo, key_hi);
                sA = (int)word_to_site(x.x, (uint32_t)n);
                sB = (int)word_to_site(x.z, (uint32_t)n);
                uA = word_to_u(x.y);
                uB = word_to_u(x.w);
            }
            if constexpr (!LEAN) {
                if (a.site_mode == SGA_SITE_REPLAY) {  // recorded stream of the reference (sweep_common.h)
                    if (vA) sA = a.replay_site[base + tA], uA = a.replay_u[base + tA];
                    if (vB) sB = a.replay_site[base + tB], uB = a.replay_u[base + tB];
                    sA = min(max(sA, 0), n - 1);
                    sB = min(max(sB, 0), n - 1);
                } else if (a.site_mode == SGA_SITE_SEQUENTIAL) {  // annealing/cuda_kernels.py:381
                    sA = vA ? tA : 0;
                    sB = vB ? tB : 0;
                    if (a.replay_u) {
                        if (vA) uA = a.replay_u[base + tA];
                        if (vB) uB = a.replay_u[base + tB];
                    }
                }
            }
            int pos = 0;  // super-window positions below pos are decided
            // The row of the accept after this one is requested ahead: of the candidates that accept
            // against the current state, the second is very likely still the next accept once the
            // first has been applied (one flip moves a field by 2 |J|), so its row travels while
            // this accept is applied and the rest is evaluated again.  (Two rows ahead -- three
            // candidates per wave and round -- measured SLOWER: a round is ~400 issued instructions per
            // wave, not a memory round trip; profiles/r03_experiments.md 1.)
            // One row request per round, unconditionally (no predicted accept: the current row again, a
            // cache hit): the loads in flight at the end of a round are then the same on every path and the
            // compiler keeps counted waits.  The two row buffers take turns (the round body is expanded
            // twice): the predicted row is never copied, so nothing waits for it before the round that
            // uses it -- it travels during this round's field update, the barrier and the next evaluation.
            RowRegs buf0 = row_request(0), buf1 = buf0;
            int held_pos = -1;  // super-window position whose row the buffer `held` of the coming round holds (-1: none)
            auto first_of = [](unsigned long long mA, unsigned long long mB) -> int {
                const int pA = mA ? 2 * (int)__builtin_ctzll(mA) : NONE;
                const int pB = mB ? 2 * (int)__builtin_ctzll(mB) + 1 : NONE;
                return min(pA, pB);
            };
            // one round; true = the super-window is done
            auto round = [&](RowRegs &held, RowRegs &other) -> bool {
                // this wave's window: first and second candidate that flip against the current state
                int p = NONE, p2 = NONE, site = 0, site2 = 0;
                int s_old = 1;  // the spin at the first candidate's site, as evaluated (the state is stable here)
                double dE = 0.0;
                // (a wave whose whole window is decided already -- it lies before `pos` -- has nothing to evaluate:
                //  it publishes "no accept" and goes on to its share of the field update)
                Verdict vdA{0, 0.0, 1}, vdB{0, 0.0, 1};
                unsigned long long mA = 0ull, mB = 0ull;
                if ((w + 1) * CLF_WINDOW > pos) {  // wave-uniform
                    bool fA, fB;
                    decide2(sA, uA, vA && gA >= pos, vdA, sB, uB, vB && gB >= pos, vdB, fA, fB);
                    mA = ballot64(fA), mB = ballot64(fB);
                    p = first_of(mA, mB);
                }
                if (p < NONE) {
                    if (p & 1) mB &= mB - 1;
                    else mA &= mA - 1;
                    p2 = first_of(mA, mB);
                    site = __builtin_amdgcn_readlane((p & 1) ? sB : sA, p >> 1);
                    if constexpr (LEAN) dE = (double)(2 * read_lane((p & 1) ? vdB.k : vdA.k, p >> 1)) * inv_sc;
                    else dE = read_lane((p & 1) ? vdB.dE : vdA.dE, p >> 1);
                    s_old = __builtin_amdgcn_readlane((p & 1) ? vdB.si : vdA.si, p >> 1);
                    if (p2 < NONE) site2 = __builtin_amdgcn_readlane((p2 & 1) ? sB : sA, p2 >> 1);
                    p += w * CLF_WINDOW;
                    if (p2 < NONE) p2 += w * CLF_WINDOW;
                }
                if (W > 1) {
                    // the earliest window with an accept decides; the predicted next accept is that wave's
                    // second, else the first of a later wave.  Lane v reads wave v's whole slot (one LDS
                    // round trip); the winner's entries are then picked across lanes.
                    int *slots = slots2 + turn * (CLF_SLOT_INTS * CLF_MAX_WAVES);
                    turn ^= 1;
                    if (lane == 0) {
                        int4 *mine = reinterpret_cast<int4 *>(slots + CLF_SLOT_INTS * w);
                        const long long dbits = __double_as_longlong(dE);
                        mine[0] = make_int4(p, p2, site, site2);
                        // (s_old comes from the evaluation, BEFORE the barrier: wave 0 flips the spin right after its
                        //  share of the row, possibly before a slower wave would get to read it)
                        mine[1] = make_int4((int)(unsigned int)dbits, (int)(dbits >> 32), s_old, 0);
                    }
                    __syncthreads();  // (A) every wave has evaluated against the old state and published
                    int4 q0 = make_int4(NONE, NONE, 0, 0), q1 = make_int4(0, 0, 1, 0);
                    if (lane < W) {
                        const int4 *theirs = reinterpret_cast<const int4 *>(slots + CLF_SLOT_INTS * lane);
                        q0 = theirs[0], q1 = theirs[1];
                    }
                    const unsigned long long have = ballot64(q0.x < NONE);
                    if (have == 0ull) {
                        p = NONE;
                    } else {
                        const int win = (int)__builtin_ctzll(have);
                        p = __builtin_amdgcn_readlane(q0.x, win), p2 = __builtin_amdgcn_readlane(q0.y, win);
                        site = __builtin_amdgcn_readlane(q0.z, win), site2 = __builtin_amdgcn_readlane(q0.w, win);
                        const unsigned int dlo = (unsigned int)__builtin_amdgcn_readlane(q1.x, win);
                        const unsigned int dhi = (unsigned int)__builtin_amdgcn_readlane(q1.y, win);
                        dE = __longlong_as_double((long long)(((unsigned long long)dhi << 32) | dlo));
                        s_old = __builtin_amdgcn_readlane(q1.z, win);
                        const unsigned long long later = have & (have - 1);
                        if (p2 >= NONE && later) {
                            const int nx = (int)__builtin_ctzll(later);
                            p2 = __builtin_amdgcn_readlane(q0.x, nx), site2 = __builtin_amdgcn_readlane(q0.z, nx);
                        }
                    }
                }
                if (p >= NONE) return true;  // the rest of the super-window is rejected
                if (held_pos != p) held = row_request(site);
                other = row_request(p2 < NONE ? site2 : site);
                held_pos = p2 < NONE ? p2 : -1;
                E += dE;
                ++nacc;
                apply_row(held, site, -2 * sc * s_old);
                if (tid == 0) {
                    bits[site >> 5] ^= 1u << (site & 31);
                    if constexpr (!LEAN) {
                        const long long upd = base + t0 + p;  // (trace buffers arrive zeroed: rejected = 0)
                        if (a.accept_trace) a.accept_trace[upd] = 1;
                        if (a.dE_trace) a.dE_trace[upd] = rule == SGA_RULE_HEAT_BATH ? -dE : dE;
                    }
                }
                pos = p + 1;
                __syncthreads();  // (B) fields and spin of the new state are visible
                return pos >= CLF_WINDOW * W;
            };
            for (;;) {
                if (round(buf0, buf1)) break;
                if (round(buf1, buf0)) break;
            }
        }
        // sweep boundary: energy record, best tracking (annealing/gpu_annealer.py:151-153)
        if (tid == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE && !a.no_best) {
            bestE = E;
            bits_to_spins(bits, a.best_spins + (long long)r * a.sstride, a.sstride, n, tid, blockDim.x);
        }
    }

    __syncthreads();
    {
        int4 *dst = reinterpret_cast<int4 *>(reinterpret_cast<FT *>(a.fields) + (long long)r * a.ldf);
        const int4 *src = reinterpret_cast<const int4 *>(F);
        for (int i = tid; i < (int)(a.ldf * FB / 16); i += blockDim.x) dst[i] = src[i];
        bits_to_spins(bits, a.spins + (long long)r * a.sstride, a.sstride, n, tid, blockDim.x);
    }
    if (tid == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

}  // namespace sga
