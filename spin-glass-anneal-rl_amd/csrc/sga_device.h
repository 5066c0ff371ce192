// sga_device.h -- device-side building blocks of the gfx950 annealing kernels:
// Philox4x32-10 in registers, the deterministic exp used by the accept rule, and DPP
// wave64 reductions.  CDNA4 only (wave64, gfx9 DPP row_mirror controls, v_readlane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sga {

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  All operands are wave-uniform in the sweep
// kernels, so hipcc keeps the whole block in SGPRs (s_mul_i32 / s_mul_hi_u32).
// Stream layout (DESIGN.md "Random streams"): key = (seed lo, seed hi),
// ctr = (block, sweep | round, replica | ladder, domain).
// ---------------------------------------------------------------------------------------
struct u32x4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 bit multiply per product (v_mad_u64_u32: half the quarter-rate multiplier passes
        // of a separate high and low product)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

constexpr uint32_t DOMAIN_SWEEP = 0, DOMAIN_EXCHANGE = 1, DOMAIN_INIT = 2;

__device__ __forceinline__ uint32_t word_to_site(uint32_t w, uint32_t n) { return __umulhi(w, n); }
__device__ __forceinline__ float word_to_u(uint32_t w) { return (float)(w >> 8) * 0x1.0p-24f; }
__device__ __forceinline__ double words_to_u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1.0p-53;
}

// ---------------------------------------------------------------------------------------
// exp for accept probabilities: Cody-Waite reduction, Taylor/Horner in explicit fma,
// two-step power-of-two scaling.  Written so that every operation is a single IEEE
// operation (build with -ffp-contract=off): results are bit-reproducible and <= 1 ulp
// from a correctly rounded exp.  Stands in for torch.exp on fp32
// (reference core/spin_dynamics.py:145) and np.exp on fp64 (annealing/parallel_tempering.py:246).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float expf_det(float x) {
    if (x != x) return x;
    if (x > 88.72284f) return __builtin_inff();
    if (x < -103.972084f) return 0.0f;
    const float nf = __builtin_rintf(x * 0x1.715476p+0f);
    float r = __builtin_fmaf(nf, -0x1.62e400p-1f, x);
    r = __builtin_fmaf(nf, -0x1.7f7d1cp-20f, r);
    float p = 0x1.a01a02p-13f;
    p = __builtin_fmaf(p, r, 0x1.6c16c2p-10f);
    p = __builtin_fmaf(p, r, 0x1.111112p-7f);
    p = __builtin_fmaf(p, r, 0x1.555556p-5f);
    p = __builtin_fmaf(p, r, 0x1.555556p-3f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const int n = (int)nf;
    const int n1 = n >> 1, n2 = n - n1;
    const float s1 = __builtin_bit_cast(float, (uint32_t)(n1 + 127) << 23);
    const float s2 = __builtin_bit_cast(float, (uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}

__device__ __forceinline__ double exp_det(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return __builtin_inf();
    if (x < -745.1332191019412) return 0.0;
    const double nf = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(nf, -0x1.62e42fee00000p-1, x);
    r = __builtin_fma(nf, -0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = __builtin_fma(p, r, 0x1.1eed8eff8d898p-29);
    p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);
    p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);
    p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int n = (int)nf;
    const int n1 = n >> 1, n2 = n - n1;
    const double s1 = __builtin_bit_cast(double, (uint64_t)(n1 + 1023) << 52);
    const double s2 = __builtin_bit_cast(double, (uint64_t)(n2 + 1023) << 52);
    return (p * s1) * s2;
}

// ---------------------------------------------------------------------------------------
// wave64 sum.  Four DPP steps fold each 16-lane row onto every lane of the row
// (quad_perm xor-1, xor-2, row_half_mirror, row_mirror); two more carry the row totals
// across: row_bcast:15 into rows 1 and 3 (r0 + r1, r2 + r3), row_bcast:31 into row 3, and lane 63
// is read back -- the association order is fixed, ((r0 + r1) + (r2 + r3)), the result
// wave-uniform.  (The rows a masked step does not write hold garbage; nothing reads them.)
// Requires EXEC = all ones.
// ---------------------------------------------------------------------------------------
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, ROWS, 0xf, true);
}
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_BCAST15 = 0x142;     // lane 15 of each row -> the next row
constexpr int DPP_ROW_BCAST31 = 0x143;     // lane 31 -> rows 2 and 3

template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, dpp_i<CTRL, ROWS>(__builtin_bit_cast(int, v)));
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ int dpp_move(int v) {
    return dpp_i<CTRL, ROWS>(v);
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ double dpp_move(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)dpp_i<CTRL, ROWS>((int)(uint32_t)b);
    const uint32_t hi = (uint32_t)dpp_i<CTRL, ROWS>((int)(uint32_t)(b >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// The lane mask of a wave-wide predicate.  HIP's __ballot(int) compares an INTEGER against zero, so a predicate that is the
// result of a vector compare is first materialised in a VGPR (v_cndmask 0 / 1) and compared again; the w64 builtin
// takes the compare's own mask (two vector instructions less on a dependent chain, per use).
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ float read_lane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int read_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double read_lane(double v, int l) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    v += dpp_move<DPP_QUAD_XOR1>(v);
    v += dpp_move<DPP_QUAD_XOR2>(v);
    v += dpp_move<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_move<DPP_ROW_MIRROR>(v);
    v += dpp_move<DPP_ROW_BCAST15, 0xa>(v);
    v += dpp_move<DPP_ROW_BCAST31, 0xc>(v);
    return read_lane(v, 63);
}

// Two wave sums with their steps interleaved: a DPP step needs wait states after the instruction that
// wrote its source, which the other chain's step fills (the compiler schedules two wave_sum calls one
// after the other, s_nop and all).  Same association order as wave_sum.
template <typename T>
__device__ __forceinline__ void wave_sum2(T &a, T &b) {
#define SGA_STEP2(CTRL, ROWS)                      \
    {                                              \
        const T ta = dpp_move<CTRL, ROWS>(a);      \
        const T tb = dpp_move<CTRL, ROWS>(b);      \
        asm volatile("" : "+v"(a), "+v"(b));       \
        a += ta;                                   \
        b += tb;                                   \
    }
    SGA_STEP2(DPP_QUAD_XOR1, 0xf)
    SGA_STEP2(DPP_QUAD_XOR2, 0xf)
    SGA_STEP2(DPP_ROW_HALF_MIRROR, 0xf)
    SGA_STEP2(DPP_ROW_MIRROR, 0xf)
    SGA_STEP2(DPP_ROW_BCAST15, 0xa)
    SGA_STEP2(DPP_ROW_BCAST31, 0xc)
#undef SGA_STEP2
    a = read_lane(a, 63);
    b = read_lane(b, 63);
}

}  // namespace sga
