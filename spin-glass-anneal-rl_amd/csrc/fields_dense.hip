// fields_dense.hip -- local fields of EVERY replica in one pass over the couplings, on the matrix
// cores:  Y[r][i] = sum_j J[i][j] s[r][j]  =  (S J^T)[r][i],  S the R x n matrix of +-1 spins.
//
// Replaces, for batches of replicas: IsingModel.compute_energy (core/ising_model.py:149-174:
// -0.5 s.(J s) - h.s, one torch.mv per replica) and EnergyComputer.compute_batch_energies
// (core/energy_computer.py:142-158); it also seeds the resident local fields of the
// cached-field sweep (sweep_clf_impl.h; the reference's incremental mode,
// core/energy_computer.py:166-173,262-265).  The per-replica kernels of sga_misc.hip walk all
// of J once per replica (22x - 415x the matrix in HBM traffic at 1024 replicas); here J is read
// once per 128 replicas.
//
//   * int8 couplings   -> v_mfma_i32_32x32x32_i8: exact
//   * fp32 couplings whose row sums are exact in fp32 (integer J, sum |J| < 2^24)
//                      -> v_mfma_f32_32x32x2_f32: every product J * (+-1) is exact, every partial
//                         sum an integer below 2^24, so the k-ordered fma chain is exact
//   * real-valued fp32 couplings -> v_mfma_f64_16x16x4_f64: exact products, fp64 accumulation,
//                         one rounding to fp32 per row (torch.mv's result type)
//
// Both operands are read straight from their row-major HBM layouts -- the A fragment of lane
// (row, half) is 16 contiguous bytes of a spin row, the B fragment 16 contiguous bytes of a
// coupling row (A = S, B = J^T: no transpose is ever formed) -- so there is no LDS staging; a
// 128-byte line of a row is consumed by the same lane pair over four consecutive K steps and
// the four waves of a workgroup share their rows through the vector L1.  The K order inside a
// step only has to be the same for A and B.
#include <type_traits>

#include "sga_device.h"
#include "sga_kernels.h"

namespace sga {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int FIELDS_TILE = 128;  // replicas x sites per workgroup (2 x 2 waves of 64 x 64)

// Which (replica tile, site tile) a workgroup takes.  Workgroups are dealt to the 8 XCDs round robin by their linear
// id, and every XCD has its own L2: with the replica tiles fastest in the id (round 3) the 8 workgroups that read the
// SAME 128 coupling rows sat on 8 different XCDs and J left HBM 8 times per pass (3.55 GB for the 400 MB matrix of
// BASELINE configs[1], 94.8 GB for 4.3 GB).  Here XCD x owns the site tiles it = x, x + 8, ... and runs a tile's
// replica tiles back to back: the first of them pulls the rows through its L2, the others find them there.
__device__ __forceinline__ void fields_tile_of(int n_rt, int n_it, int &rt, int &it) {
    const int L = (int)blockIdx.x, x = L & 7, slot = L >> 3;  // (1-D grid of 8 * ceil(n_it / 8) * n_rt workgroups)
    rt = slot % n_rt;
    it = (slot / n_rt) * 8 + x;
    if (it >= n_it) rt = -1;  // (a remainder slot of the last group of 8 site tiles)
}

// MODE 0: int8 J -> int32 Y; MODE 1: fp32 J, exact fp32 sums -> fp32 Y
template <int MODE>
__global__ void __launch_bounds__(256) fields_mfma_kernel(const FieldsArgs a) {
    using acc_t = typename std::conditional<MODE == 0, v16i, v16f>::type;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 31, hh = lane >> 5;
    int rt, it;
    fields_tile_of((a.R + FIELDS_TILE - 1) / FIELDS_TILE, (a.n + FIELDS_TILE - 1) / FIELDS_TILE, rt, it);
    if (rt < 0) return;
    const int r0 = rt * FIELDS_TILE + (w >> 1) * 64;
    const int i0 = it * FIELDS_TILE + (w & 1) * 64;
    // rows past the end are clamped for the loads (their results are not stored)
    const int8_t *Sa[2];
    const unsigned char *Jb[2];
    constexpr int JB = MODE == 0 ? 1 : 4;  // bytes per coupling
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rr = min(r0 + 32 * t + q, a.R - 1), ii = min(i0 + 32 * t + q, a.n - 1);
        Sa[t] = a.spins + (long long)rr * a.sstride;
        Jb[t] = reinterpret_cast<const unsigned char *>(a.J) + (long long)ii * a.ldj * JB;
    }
    acc_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][nn][j] = 0;

    constexpr int U = 4;  // K steps whose loads are issued together
    if constexpr (MODE == 0) {
        // K step = 32 couplings: lane (row, half) holds bytes [k0 + 16 half, +16) of its row
        for (long long k0 = 0; k0 < a.ldj; k0 += 32 * U) {  // ldj % 128 == 0
            v4i fa[U][2], fb[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[u][t] = *reinterpret_cast<const v4i *>(Sa[t] + k0 + 32 * u + 16 * hh);
                    fb[u][t] = *reinterpret_cast<const v4i *>(Jb[t] + k0 + 32 * u + 16 * hh);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u][m], fb[u][nn], acc[m][nn], 0, 0, 0);
        }
    } else {
        // K step = 2 couplings per MFMA; a macro step of 8: lane (row, half) holds elements
        // [k0 + 4 half, +4) of its row and feeds them to four consecutive MFMAs
        for (long long k0 = 0; k0 < a.ldj; k0 += 8 * U) {  // ldj % 32 == 0
            int sa[U][2];
            float4 xb[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    sa[u][t] = *reinterpret_cast<const int *>(Sa[t] + k0 + 8 * u + 4 * hh);
                    xb[u][t] = *reinterpret_cast<const float4 *>(Jb[t] + (k0 + 8 * u + 4 * hh) * 4);
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float af[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[t][0] = (float)(int8_t)(sa[u][t]);
                    af[t][1] = (float)(int8_t)(sa[u][t] >> 8);
                    af[t][2] = (float)(int8_t)(sa[u][t] >> 16);
                    af[t][3] = (float)(sa[u][t] >> 24);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nn = 0; nn < 2; ++nn) {
                            const float bv = e == 0 ? xb[u][nn].x : e == 1 ? xb[u][nn].y : e == 2 ? xb[u][nn].z : xb[u][nn].w;
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][e], bv, acc[m][nn], 0, 0, 0);
                        }
                }
            }
        }
    }
    // C/D layout of the 32x32 forms: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    using out_t = typename std::conditional<MODE == 0, int, float>::type;
    out_t *Y = reinterpret_cast<out_t *>(a.Y);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            const int col = i0 + 32 * nn + q;
            if (col >= a.n) continue;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = r0 + 32 * m + (j & 3) + 8 * (j >> 2) + 4 * hh;
                if (row < a.R) Y[(long long)row * a.ldy + col] = acc[m][nn][j];
            }
        }
}

// Real-valued fp32 couplings: fp64 MFMA.  16 x 16 tiles, K step 4: lane l holds A[l & 15][k = l >> 4]
// and B[k = l >> 4][l & 15] as one double each; C/D: column = lane & 15, row = (lane >> 4) + 4 reg.
// A wave computes 32 replicas x 64 sites (2 x 4 tiles), a workgroup 64 x 128.
__global__ void __launch_bounds__(256) fields_mfma_f64_kernel(const FieldsArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, kk = lane >> 4;
    int rt, it;
    fields_tile_of((a.R + 63) / 64, (a.n + 127) / 128, rt, it);
    if (rt < 0) return;
    const int r0 = rt * 64 + (w >> 1) * 32;
    const int i0 = it * 128 + (w & 1) * 64;
    const int8_t *Sa[2];
    const float *Jb[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) Sa[t] = a.spins + (long long)min(r0 + 16 * t + q, a.R - 1) * a.sstride;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        Jb[t] = reinterpret_cast<const float *>(a.J) + (long long)min(i0 + 16 * t + q, a.n - 1) * a.ldj;
    v4d acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][nn][j] = 0.0;
    // macro step of 16 couplings: lane (row, kk) holds elements [k0 + 4 kk, +4) of its row and feeds
    // them to four consecutive MFMAs (the K order is the same for A and B)
    for (long long k0 = 0; k0 < a.ldj; k0 += 16) {  // ldj % 32 == 0
        int sa[2];
        float4 xb[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) sa[t] = *reinterpret_cast<const int *>(Sa[t] + k0 + 4 * kk);
#pragma unroll
        for (int t = 0; t < 4; ++t) xb[t] = *reinterpret_cast<const float4 *>(Jb[t] + k0 + 4 * kk);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const double av = (double)(int8_t)(sa[m] >> (8 * e));
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    const float bv = e == 0 ? xb[nn].x : e == 1 ? xb[nn].y : e == 2 ? xb[nn].z : xb[nn].w;
                    acc[m][nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, (double)bv, acc[m][nn], 0, 0, 0);
                }
            }
    }
    float *Y = reinterpret_cast<float *>(a.Y);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            const int col = i0 + 16 * nn + q;
            if (col >= a.n) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = r0 + 16 * m + kk + 4 * j;
                if (row < a.R) Y[(long long)row * a.ldy + col] = (float)acc[m][nn][j];  // torch.mv row, fp32
            }
        }
}

hipError_t launch_fields_dense(const FieldsArgs &a, int mode, hipStream_t st) {
    if (a.R <= 0 || a.n <= 0) return hipErrorInvalidValue;
    auto grid_of = [](int n_rt, int n_it) { return dim3((unsigned)(8 * ((n_it + 7) / 8) * n_rt)); };  // fields_tile_of
    if (mode == 2) {
        hipLaunchKernelGGL(fields_mfma_f64_kernel, grid_of((a.R + 63) / 64, (a.n + 127) / 128), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    const dim3 grid = grid_of((a.R + FIELDS_TILE - 1) / FIELDS_TILE, (a.n + FIELDS_TILE - 1) / FIELDS_TILE);
    if (mode == 0) hipLaunchKernelGGL(fields_mfma_kernel<0>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(fields_mfma_kernel<1>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// From the fields to the energies (and to the resident fields of the cached-field sweep):
//   E_r = -1/2 fp32(sum_i Y_ri s_ri) - fp32(sum_i h_i s_ri)      (core/ising_model.py:161-168)
//   F_ri = scale * (Y_ri + h_i)   as int16 | int32               (integer problems only)
// One workgroup per replica; sums in fp64 in a fixed order (lane-strided, tree, waves in order).
template <typename YT>
__global__ void __launch_bounds__(256) fields_finish_kernel(const FieldsArgs a) {
    __shared__ double red[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    const YT *y = reinterpret_cast<const YT *>(a.Y) + (long long)r * a.ldy;
    const int8_t *s = a.spins + (long long)r * a.sstride;
    int16_t *f16 = a.field_bits == 16 ? reinterpret_cast<int16_t *>(a.fields) + (long long)r * a.ldf : nullptr;
    int32_t *f32 = a.field_bits == 32 ? reinterpret_cast<int32_t *>(a.fields) + (long long)r * a.ldf : nullptr;
    double e = 0.0, hs = 0.0;
    for (int i = tid; i < a.n; i += 256) {
        const double yi = (double)y[i], si = (double)s[i];
        const float hi = a.h[i];
        e += yi * si;
        hs += (double)hi * si;
        if (f16 || f32) {
            const int v = (int)((float)a.field_scale * ((float)y[i] + hi));  // exact: integers / half-integers < 2^24
            if (f16) f16[i] = (int16_t)v;
            else f32[i] = v;
        }
    }
    for (int i = a.n + tid; i < a.ldf && (f16 || f32); i += 256) {  // pad entries: never read by a decision
        if (f16) f16[i] = 0;
        else f32[i] = 0;
    }
    e = wave_sum(e);
    hs = wave_sum(hs);
    if (lane == 0) {
        red[w] = e;
        red[4 + w] = hs;
    }
    __syncthreads();
    if (tid == 0 && a.energy) {
        const double et = (red[0] + red[1]) + (red[2] + red[3]);
        const double ht = (red[4] + red[5]) + (red[6] + red[7]);
        a.energy[r] = -0.5 * (double)(float)et + (-(double)(float)ht);
    }
}

hipError_t launch_fields_finish(const FieldsArgs &a, bool y_is_int, hipStream_t st) {
    if (y_is_int) hipLaunchKernelGGL(fields_finish_kernel<int>, dim3(a.R), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(fields_finish_kernel<float>, dim3(a.R), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace sga
