// sga_misc.hip -- the kernels around the sweep: coupling repack, spin init, full energy
// evaluation, replica exchange.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>

#include "sweep_common.h"

namespace sga {

static int grid_for(long long total);

// ---------------------------------------------------------------------------------------
// J repack: caller's fp32 [n][ldJ] -> engine layout [n][ld] (float | int8), rows zero padded
// to a whole number of 1-KiB chunks per wave, so the sweep kernel needs no tail masking.
// ---------------------------------------------------------------------------------------
template <typename OT>
__global__ void repack_dense_kernel(const float *__restrict__ J, long long ldJ, long long rows,
                                    int n, OT *__restrict__ out, long long ld,
                                    float *__restrict__ diag) {
    // `rows` = n_models * n stacked rows of n columns each
    const long long total = rows * ld;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / ld, col = i - row * ld;
        const float v = (col < n) ? J[row * ldJ + col] : 0.0f;
        out[i] = (OT)v;
        if (diag && col == row % n) diag[row] = v;
    }
}

hipError_t launch_repack_dense(const float *J, long long ldJ, long long rows, int n, void *out,
                               long long ld, bool to_i8, float *diag, hipStream_t st) {
    const long long total = rows * ld;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (to_i8)
        hipLaunchKernelGGL(repack_dense_kernel<int8_t>, dim3(blocks), dim3(256), 0, st, J, ldJ, rows,
                           n, (int8_t *)out, ld, diag);
    else
        hipLaunchKernelGGL(repack_dense_kernel<float>, dim3(blocks), dim3(256), 0, st, J, ldJ, rows,
                           n, (float *)out, ld, diag);
    return hipGetLastError();
}

__global__ void scan_values_kernel(const float *__restrict__ v, long long rows, long long cols,
                                   long long ld, int *flags) {
    int not_i8 = 0, not_small_int = 0;  // second flag: not ternary
    int exp_hi = 0, exp_lo = 0;
    const long long total = rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / cols, col = i - row * cols;
        const float x = v[row * ld + col];
        const bool is_int = (x == __builtin_rintf(x));
        if (!is_int || !(__builtin_fabsf(x) <= 127.0f)) not_i8 = 1;
        if (!is_int || !(__builtin_fabsf(x) <= 1.0f)) not_small_int = 1;
        if (x != 0.0f) {  // binary exponents of the value's highest and lowest set bits
            const unsigned int bits = __float_as_uint(x);
            const int ef = (int)((bits >> 23) & 255u);
            const unsigned int mant = (bits & 0x7FFFFFu) | (ef ? 0x800000u : 0u);
            exp_hi = max(exp_hi, (ef ? ef - 127 : -127) + 1024);
            exp_lo = max(exp_lo, 1024 - ((ef ? ef - 127 : -126) - 23 + __builtin_ctz(mant ? mant : 1u)));
        }
    }
    if (not_i8) atomicOr(&flags[0], 1);
    if (not_small_int) atomicOr(&flags[1], 1);  // some J outside {-1, 0, +1}
    if (exp_hi) atomicMax(&flags[5], exp_hi);
    if (exp_lo) atomicMax(&flags[6], exp_lo);
}

hipError_t launch_scan_values(const float *v, long long rows, long long cols, long long ld,
                              int *flags, hipStream_t st) {
    const long long total = rows * cols;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(scan_values_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, v,
                       rows, cols, ld, flags);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) dense_row_abs_max_kernel(const float *__restrict__ J,
                                                                long long ldJ,
                                                                const float *__restrict__ h, int n,
                                                                unsigned int *out) {
    // one workgroup per stacked row (gridDim.x = n_models * n), n columns each
    __shared__ double red[4];
    __shared__ int bad[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double acc = 0.0;
    int nonint = 0;
    float big = 0.0f;
    for (int j = tid; j < n; j += 256) {
        const float x = J[(long long)i * ldJ + j];
        acc += (double)__builtin_fabsf(x);
        nonint |= (x != __builtin_rintf(x));
        big = __builtin_fmaxf(big, __builtin_fabsf(x));
    }
    // out[5]: max |J_ij| over the matrix (non-negative floats: bit order = value order; one atomic per wave)
    for (int sft = 32; sft >= 1; sft >>= 1) big = __builtin_fmaxf(big, __shfl_xor(big, sft));
    if (lane == 0 && big > 0.0f) atomicMax(&out[5], __builtin_bit_cast(unsigned int, big));
    const double ws = wave_sum(acc);
    const int wb = wave_sum(nonint);
    if (lane == 0) {
        red[w] = ws;
        bad[w] = wb;
    }
    __syncthreads();
    if (tid == 0) {
        const float hi = h[i];
        const float row = (float)((red[0] + red[1]) + (red[2] + red[3]) + (double)__builtin_fabsf(hi));
        atomicMax(&out[0], __builtin_bit_cast(unsigned int, row));  // non-negative: bit order = value order
        if (bad[0] | bad[1] | bad[2] | bad[3]) atomicOr(&out[1], 1u);  // some J not an integer
        if (hi != __builtin_rintf(hi)) atomicOr(&out[1], 2u);           // some h not an integer
        if (2.0f * hi != __builtin_rintf(2.0f * hi)) atomicOr(&out[1], 4u);  // ... not even a multiple of 1/2
    }
}
hipError_t launch_dense_row_abs_max(const float *J, long long ldJ, const float *h, long long rows,
                                    int n, unsigned int *out, hipStream_t st) {
    hipLaunchKernelGGL(dense_row_abs_max_kernel, dim3((unsigned)rows), dim3(256), 0, st, J, ldJ, h,
                       n, out);
    return hipGetLastError();
}

// fp32 row -> sign plane / non-zero plane words (bit b of word q = coupling 32 q + b)
__global__ void __launch_bounds__(256) repack_tern2_kernel(const float *__restrict__ J, long long ldJ,
                                                            int n, unsigned int *__restrict__ planes,
                                                            long long row_bits, float *row_nnz) {
    __shared__ int cnt[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long words = row_bits / 32;
    unsigned int *S = planes + (long long)i * words;
    unsigned int *Z = planes + (long long)gridDim.x * words + (long long)i * words;
    int nnz = 0;
    for (long long q = tid; q < words; q += 256) {
        unsigned int sb = 0, zb = 0;
        for (int b = 0; b < 32; ++b) {
            const long long j = q * 32 + b;
            const float v = j < n ? J[(long long)i * ldJ + j] : 0.0f;
            sb |= (v < 0.0f ? 1u : 0u) << b;
            zb |= (v != 0.0f ? 1u : 0u) << b;
        }
        S[q] = sb;
        Z[q] = zb;
        nnz += __builtin_popcount(zb);
    }
    const int ws = wave_sum(nnz);
    if (lane == 0) cnt[w] = ws;
    __syncthreads();
    if (tid == 0) row_nnz[i] = (float)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
}
hipError_t launch_repack_tern2(const float *J, long long ldJ, int n, unsigned int *planes,
                               long long row_bits, float *row_nnz, hipStream_t st) {
    hipLaunchKernelGGL(repack_tern2_kernel, dim3(n), dim3(256), 0, st, J, ldJ, n, planes, row_bits, row_nnz);
    return hipGetLastError();
}

// J[i][j] == J[j][i] and J[i][i] == 0, per model block of n rows: 64 x 64 tiles, the tile (bi, bj) through
// LDS against the coalesced rows of tile (bj, bi) -- the matrix is read once (the element-wise form read
// the transposed operand one cache line per element: 12.6 GB for the 400 MB matrix of n = 10^4, 118 GB and
// 16 ms at n = 32 768, profiles/r03_experiments.md).
__global__ void __launch_bounds__(256) check_symmetric_kernel(const float *__restrict__ J, long long ldJ, long long rows,
                                                              int n, int *out) {
    __shared__ float tile[64][65];
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bi > bj) return;  // the pair (bi, bj) covers both triangles
    const float *M = J + (long long)blockIdx.z * n * ldJ;  // this model's block
    const int r0 = bi * 64, c0 = bj * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        tile[i][tx] = (r0 + i < n && c0 + tx < n) ? M[(long long)(r0 + i) * ldJ + c0 + tx] : 0.0f;
    __syncthreads();
    int bad = 0;
    for (int j = ty; j < 64; j += 4) {  // row c0 + j of the mirrored tile, columns r0 + tx
        if (c0 + j < n && r0 + tx < n) {
            const float b = M[(long long)(c0 + j) * ldJ + r0 + tx], a = tile[tx][j];
            if (a != b || (r0 + tx == c0 + j && b != 0.0f)) bad = 1;
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(out, 1);
    (void)rows;
}
hipError_t launch_check_symmetric(const float *J, long long ldJ, long long rows, int n, int *out,
                                  hipStream_t st) {
    const int nt = (n + 63) / 64;
    hipLaunchKernelGGL(check_symmetric_kernel, dim3(nt, nt, (unsigned)(rows / n)), dim3(256), 0, st, J, ldJ, rows, n,
                       out);
    return hipGetLastError();
}

__global__ void update_best_kernel(const double *energy, const int8_t *spins, double *best_energy,
                                   int8_t *best_spins, int sstride, int R) {
    const int r = blockIdx.x;
    if (r >= R || !(energy[r] < best_energy[r])) return;  // uniform per workgroup
    const int4 *src = reinterpret_cast<const int4 *>(spins + (long long)r * sstride);
    int4 *dst = reinterpret_cast<int4 *>(best_spins + (long long)r * sstride);
    for (int i = threadIdx.x; i < sstride / 16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) best_energy[r] = energy[r];
}
hipError_t launch_update_best(const double *energy, const int8_t *spins, double *best_energy,
                              int8_t *best_spins, int sstride, int R, hipStream_t st) {
    hipLaunchKernelGGL(update_best_kernel, dim3(R), dim3(256), 0, st, energy, spins, best_energy,
                       best_spins, sstride, R);
    return hipGetLastError();
}

__global__ void gather_diag_csr_kernel(const long long *rowptr, const int32_t *colidx,
                                       const float *val, int n, float *diag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d = 0.0f;
    for (long long j = rowptr[i]; j < rowptr[i + 1]; ++j)
        if (colidx[j] == i) d += val[j];
    diag[i] = d;
}
hipError_t launch_gather_diag_csr(const long long *rowptr, const int32_t *colidx, const float *val,
                                  int n, float *diag, hipStream_t st) {
    hipLaunchKernelGGL(gather_diag_csr_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rowptr,
                       colidx, val, n, diag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// CSR structure: row extents in both widths, validation without a round trip of the entries
// to the host (4e9 entries at the 1000-city TSP instance).
// ---------------------------------------------------------------------------------------
__global__ void pack_cv_kernel(const int32_t *__restrict__ colidx, const float *__restrict__ val,
                               int2 *__restrict__ cv, long long nnz) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz;
         i += (long long)gridDim.x * blockDim.x)
        cv[i] = make_int2(colidx[i], __float_as_int(val[i]));
}
hipError_t launch_pack_cv(const int32_t *colidx, const float *val, int2 *cv, long long nnz,
                          hipStream_t st) {
    if (nnz <= 0) return hipSuccess;
    const int blocks = (int)std::min<long long>((nnz + 255) / 256, 65536);
    hipLaunchKernelGGL(pack_cv_kernel, dim3(blocks), dim3(256), 0, st, colidx, val, cv, nnz);
    return hipGetLastError();
}

// Row-wise packing into a layout whose rows may be PADDED (dst extents >= src extents): long rows
// are padded to whole 64-entry slots (512 B) so that the wide sweep forms address a row by
// wave-uniform slot numbers -- scalar address arithmetic, no per-lane bounds tests.  Pad entries
// carry value 0 and the row's first column (an in-range gather).  One workgroup per row.
// SRC_CV: the source is already interleaved (re-padding an unpadded layout on demand).
template <bool SRC_CV>
__global__ void __launch_bounds__(256) pack_cv_rows_kernel(const long long *__restrict__ src_ptr,
                                                           const long long *__restrict__ dst_ptr,
                                                           const int32_t *__restrict__ colidx,
                                                           const float *__restrict__ val,
                                                           const int2 *__restrict__ cv_src,
                                                           int2 *__restrict__ cv, int n) {
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const long long sb = src_ptr[i], len = src_ptr[i + 1] - sb;
        const long long db = dst_ptr[i], plen = dst_ptr[i + 1] - db;
        int pad_col = 0;
        if (len > 0) pad_col = SRC_CV ? cv_src[sb].x : colidx[sb];
        for (long long j = threadIdx.x; j < plen; j += blockDim.x) {
            int2 e = make_int2(pad_col, 0);
            if (j < len) {
                if constexpr (SRC_CV) e = cv_src[sb + j];
                else e = make_int2(colidx[sb + j], __float_as_int(val[sb + j]));
            }
            cv[db + j] = e;
        }
    }
}
hipError_t launch_pack_cv_rows(const long long *src_ptr, const long long *dst_ptr, const int32_t *colidx,
                               const float *val, const int2 *cv_src, int2 *cv, int n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int blocks = std::min(n, 65536);
    if (cv_src)
        hipLaunchKernelGGL(pack_cv_rows_kernel<true>, dim3(blocks), dim3(256), 0, st, src_ptr, dst_ptr,
                           colidx, val, cv_src, cv, n);
    else
        hipLaunchKernelGGL(pack_cv_rows_kernel<false>, dim3(blocks), dim3(256), 0, st, src_ptr, dst_ptr,
                           colidx, val, cv_src, cv, n);
    return hipGetLastError();
}

// Sparse couplings handed over as a dense matrix (sga_set_dense with SGA_J_AUTO): non-zeros per row, then the
// CSR arrays (columns ascending), one wave per row.
__global__ void __launch_bounds__(256) dense_row_nnz_kernel(const float *__restrict__ J, long long ld, int n, int *__restrict__ nnz) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *row = J + (long long)i * ld;
    int count = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        count += __builtin_popcountll(__ballot(c < n && row[c] != 0.0f));
    }
    if (lane == 0) nnz[i] = count;
}
__global__ void __launch_bounds__(256) dense_to_csr_kernel(const float *__restrict__ J, long long ld, int n,
                                                           const int *__restrict__ rowptr, int *__restrict__ col,
                                                           float *__restrict__ val) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *row = J + (long long)i * ld;
    int at = rowptr[i];
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        const float v = c < n ? row[c] : 0.0f;
        const unsigned long long m = __ballot(v != 0.0f);
        if (v != 0.0f) {
            const int k = at + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            col[k] = c;
            val[k] = v;
        }
        at += __builtin_popcountll(m);
    }
}
hipError_t launch_dense_row_nnz(const float *J, long long ld, int n, int *nnz, hipStream_t st) {
    hipLaunchKernelGGL(dense_row_nnz_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, J, ld, n, nnz);
    return hipGetLastError();
}
hipError_t launch_dense_to_csr(const float *J, long long ld, int n, const int *rowptr, int *col, float *val, hipStream_t st) {
    hipLaunchKernelGGL(dense_to_csr_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, J, ld, n, rowptr, col, val);
    return hipGetLastError();
}

// (column, fp32 value) -> column | int8 value << 24; *bad is set when some value is not an integer in [-127, 127]
// or some column needs more than 24 bits
__global__ void __launch_bounds__(256) pack_entries_kernel(const int2 *cv, uint32_t *cvp, long long count, int *bad) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const int2 e = cv[i];
        const float v = __int_as_float(e.y);
        const int iv = (int)v;
        if ((float)iv != v || iv < -127 || iv > 127 || (unsigned int)e.x >= (1u << 24)) *bad = 1;
        cvp[i] = ((uint32_t)iv << 24) | ((uint32_t)e.x & 0xFFFFFFu);
    }
}
hipError_t launch_pack_entries(const int2 *cv, uint32_t *cvp, long long count, int *bad, hipStream_t st) {
    hipLaunchKernelGGL(pack_entries_kernel, dim3(4096), dim3(256), 0, st, cv, cvp, count, bad);
    return hipGetLastError();
}

// the wide forms' per-row record carries the row's field (one 16-byte load per extent)
__global__ void rowinfo_fields_kernel(int4 *rowinfo, const float *h, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rowinfo[i].w = __float_as_int(h[i]);
}
hipError_t launch_rowinfo_fields(int4 *rowinfo, const float *h, int n, hipStream_t st) {
    hipLaunchKernelGGL(rowinfo_fields_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rowinfo, h, n);
    return hipGetLastError();
}

__global__ void widen_rowptr_kernel(const int32_t *src, long long *dst, long long count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[i];
}
__global__ void narrow_rowptr_kernel(const long long *src, int32_t *dst, long long count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (int32_t)src[i];
}
hipError_t launch_widen_rowptr(const int32_t *src, long long *dst, long long count, hipStream_t st) {
    hipLaunchKernelGGL(widen_rowptr_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                       src, dst, count);
    return hipGetLastError();
}
hipError_t launch_narrow_rowptr(const long long *src, int32_t *dst, long long count, hipStream_t st) {
    hipLaunchKernelGGL(narrow_rowptr_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                       src, dst, count);
    return hipGetLastError();
}

__global__ void csr_check_rowptr_kernel(const long long *rowptr, int n, long long nnz, int *flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    bool bad = false;
    if (i == 0) bad = rowptr[0] != 0 || rowptr[n] != nnz;
    if (i < n) bad = bad || rowptr[i + 1] < rowptr[i] || rowptr[i] < 0 || rowptr[i + 1] > nnz;
    if (bad) flags[CSR_BAD_ROWPTR] = 1;
}
hipError_t launch_csr_check_rowptr(const long long *rowptr, int n, long long nnz, int *flags,
                                   hipStream_t st) {
    hipLaunchKernelGGL(csr_check_rowptr_kernel, dim3(n / 256 + 1), dim3(256), 0, st, rowptr, n, nnz,
                       flags);
    return hipGetLastError();
}

// one wave per row (grid-stride): ranges, integrality, ordering, diagonal, sum of |entries|
__global__ void __launch_bounds__(256) csr_scan_kernel(const long long *rowptr, const int32_t *colidx,
                                                       const float *val, const float *h, int n,
                                                       int *flags) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
    int bad_col = 0, non_int = 0, unsorted = 0, diag = 0;
    int exp_hi = 0, exp_lo = 0;  // 1024 + highest exponent | 1024 - lowest set bit's exponent (0: no value yet)
    float row_max = 0.0f, row_j_max = 0.0f;
    for (int i = wave; i < n; i += n_waves) {
        const long long beg = rowptr[i], end = rowptr[i + 1];
        double acc = 0.0;
        for (long long j = beg + lane; j < end; j += 64) {
            const int c = colidx[j];
            const float v = val[j];
            if (c < 0 || c >= n) bad_col = 1;
            if (v != rintf(v)) non_int |= 1;
            if (v != 0.0f) {  // binary exponents of the value's highest and lowest set bits
                const unsigned int bits = __float_as_uint(v);
                const int ef = (int)((bits >> 23) & 255u);
                const unsigned int mant = (bits & 0x7FFFFFu) | (ef ? 0x800000u : 0u);
                const int e_hi = ef ? ef - 127 : -127;
                const int e_lo = (ef ? ef - 127 : -126) - 23 + __builtin_ctz(mant ? mant : 1u);
                exp_hi = max(exp_hi, e_hi + 1024);
                exp_lo = max(exp_lo, 1024 - e_lo);
            }
            if (c == i && v != 0.0f) diag = 1;
            if (j > beg && colidx[j - 1] >= c) unsorted = 1;
            acc += (double)fabsf(v);
        }
        const float hi = h[i];
        if (hi != rintf(hi)) non_int |= 2;
        if (2.0f * hi != rintf(2.0f * hi)) non_int |= 4;
        // an upper bound is all the table needs; fp32 rounds it up or down by < 1 ulp
        const double jsum = wave_sum(acc);
        const float tot = (float)(jsum + (double)fabsf(hi));
        row_max = fmaxf(row_max, tot);
        row_j_max = fmaxf(row_j_max, (float)jsum);  // sum_j |J_ij| alone: the range of the dynamic part of a field
    }
    if (bad_col) flags[CSR_BAD_COLUMN] = 1;
    if (non_int) atomicOr(&flags[CSR_NOT_INTEGRAL], non_int);  // bit 0: some J, bit 1: some h, bit 2: some 2 h
    if (exp_hi) atomicMax(&flags[CSR_EXP_HI], exp_hi);
    if (exp_lo) atomicMax(&flags[CSR_EXP_LO], exp_lo);
    if (unsorted) flags[CSR_UNSORTED] = 1;
    if (diag) flags[CSR_DIAGONAL] = 1;
    if (lane == 0) atomicMax(&flags[CSR_ROW_ABS_MAX], __float_as_int(row_max));  // >= 0: bits order
    if (lane == 0) atomicMax(&flags[CSR_ROW_J_ABS_MAX], __float_as_int(row_j_max));
}
hipError_t launch_csr_scan(const long long *rowptr, const int32_t *colidx, const float *val,
                           const float *h, int n, int *flags, hipStream_t st) {
    const int blocks = (int)std::min<long long>(((long long)n + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(blocks), dim3(256), 0, st, rowptr, colidx, val, h, n,
                       flags);
    return hipGetLastError();
}

// J[i][j] == J[j][i] for every stored entry (duplicates count summed when rows are unsorted)
template <bool SORTED>
__global__ void __launch_bounds__(256) csr_symmetry_kernel(const long long *rowptr,
                                                           const int32_t *colidx, const float *val,
                                                           int n, int *flags) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
    int asym = 0;
    for (int i = wave; i < n; i += n_waves) {
        const long long beg = rowptr[i], end = rowptr[i + 1];
        for (long long j = beg + lane; j < end; j += 64) {
            const int c = colidx[j];
            if (c == i) continue;
            const long long cb = rowptr[c], ce = rowptr[c + 1];
            if constexpr (SORTED) {
                long long lo = cb, hi = ce;  // first entry of row c with column >= i
                while (lo < hi) {
                    const long long mid = (lo + hi) >> 1;
                    if (colidx[mid] < i) lo = mid + 1;
                    else hi = mid;
                }
                const float other = (lo < ce && colidx[lo] == i) ? val[lo] : 0.0f;
                if (other != val[j]) asym = 1;
            } else {
                float mine = 0.0f, other = 0.0f;
                for (long long q = beg; q < end; ++q)
                    if (colidx[q] == c) mine += val[q];
                for (long long q = cb; q < ce; ++q)
                    if (colidx[q] == i) other += val[q];
                if (mine != other) asym = 1;
            }
        }
    }
    if (asym) flags[CSR_ASYMMETRIC] = 1;
}
hipError_t launch_csr_symmetry(const long long *rowptr, const int32_t *colidx, const float *val,
                               int n, bool sorted, int *flags, hipStream_t st) {
    const int blocks = (int)std::min<long long>(((long long)n + 3) / 4, 256 * 32);
    if (sorted)
        hipLaunchKernelGGL(csr_symmetry_kernel<true>, dim3(blocks), dim3(256), 0, st, rowptr, colidx,
                           val, n, flags);
    else
        hipLaunchKernelGGL(csr_symmetry_kernel<false>, dim3(blocks), dim3(256), 0, st, rowptr,
                           colidx, val, n, flags);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// spins: random init (Philox domain 2), pad / unpad between [R][n] and [R][sstride]
// ---------------------------------------------------------------------------------------
__global__ void init_spins_kernel(int8_t *spins, int n, int sstride, int R, uint32_t seed_lo,
                                  uint32_t seed_hi, uint32_t replica0) {
    // one thread per 128-spin block: reference core/ising_model.py:67 draws randint(0,2)*2-1
    const int nblk = (sstride + 127) / 128;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * nblk) return;
    const int r = (int)(idx / nblk), blk = (int)(idx - (long long)r * nblk);
    const u32x4 w = philox4x32_10((uint32_t)blk, 0u, replica0 + (uint32_t)r, DOMAIN_INIT, seed_lo,
                                  seed_hi);
    const uint32_t words[4] = {w.x, w.y, w.z, w.w};
    int8_t *dst = spins + (long long)r * sstride;
    for (int q = 0; q < 128; ++q) {
        const int i = blk * 128 + q;
        if (i >= sstride) break;
        const uint32_t bit = (words[q >> 5] >> (q & 31)) & 1u;
        dst[i] = (i < n) ? (bit ? 1 : -1) : 0;
    }
}
hipError_t launch_init_spins(int8_t *spins, int n, int sstride, int R, uint32_t seed_lo,
                             uint32_t seed_hi, uint32_t replica0, hipStream_t st) {
    const long long total = (long long)R * ((sstride + 127) / 128);
    hipLaunchKernelGGL(init_spins_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       spins, n, sstride, R, seed_lo, seed_hi, replica0);
    return hipGetLastError();
}

__global__ void pad_spins_kernel(const int8_t *src, int n, int8_t *dst, int sstride, int R) {
    const long long total = (long long)R * sstride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / sstride, c = i - r * sstride;
        dst[i] = (c < n) ? src[r * n + c] : (int8_t)0;
    }
}
__global__ void unpad_spins_kernel(const int8_t *src, int sstride, int8_t *dst, int n, int R) {
    const long long total = (long long)R * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / n, c = i - r * n;
        dst[i] = src[r * sstride + c];
    }
}
static int grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
hipError_t launch_pad_spins(const int8_t *src, int n, int8_t *dst, int sstride, int R,
                            hipStream_t st) {
    hipLaunchKernelGGL(pad_spins_kernel, dim3(grid_for((long long)R * sstride)), dim3(256), 0, st,
                       src, n, dst, sstride, R);
    return hipGetLastError();
}
hipError_t launch_unpad_spins(const int8_t *src, int sstride, int8_t *dst, int n, int R,
                              hipStream_t st) {
    hipLaunchKernelGGL(unpad_spins_kernel, dim3(grid_for((long long)R * n)), dim3(256), 0, st, src,
                       sstride, dst, n, R);
    return hipGetLastError();
}

__global__ void copy_best_kernel(const double *energy, const int8_t *spins, double *best_energy,
                                 int8_t *best_spins, int sstride, int R) {
    const int r = blockIdx.x;
    if (r >= R) return;
    if (threadIdx.x == 0) best_energy[r] = energy[r];
    const int4 *src = reinterpret_cast<const int4 *>(spins + (long long)r * sstride);
    int4 *dst = reinterpret_cast<int4 *>(best_spins + (long long)r * sstride);
    for (int i = threadIdx.x; i < sstride / 16; i += blockDim.x) dst[i] = src[i];
}
hipError_t launch_copy_best(const double *energy, const int8_t *spins, double *best_energy,
                            int8_t *best_spins, int sstride, int R, hipStream_t st) {
    hipLaunchKernelGGL(copy_best_kernel, dim3(R), dim3(256), 0, st, energy, spins, best_energy,
                       best_spins, sstride, R);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Full energy  E = -0.5 * fp32(s.(J s)) - fp32(h.s)   (core/ising_model.py:149-174;
// operator form annealing/cuda_kernels.py:284-324, fallback :400-413).
// One workgroup (4 waves) per replica; spins in LDS; each wave takes rows w, w+4, ...,
// streams the row with 16-B loads, DPP-reduces J[i,:].s, rounds it to fp32 as torch.mv does.
// ---------------------------------------------------------------------------------------
// E = -1/2 fp32(sum_i mv_i s_i) - fp32(sum_i h_i s_i) (core/ising_model.py:161-168): written
// directly, or left as this slice's two sums for energy_finish_kernel
__device__ inline void energy_out(const EnergyArgs &a, int r, double e, double hs) {
    if (a.slices <= 1) {
        a.energy[r] = -0.5 * (double)(float)e + (-(double)(float)hs);
    } else {
        double *p = a.partial + ((long long)r * a.slices + blockIdx.y) * 2;
        p[0] = e;
        p[1] = hs;
    }
}
__global__ void energy_finish_kernel(const double *partial, int slices, double *energy, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double e = 0.0, hs = 0.0;
    for (int q = 0; q < slices; ++q) {  // fixed order: deterministic
        e += partial[((long long)r * slices + q) * 2];
        hs += partial[((long long)r * slices + q) * 2 + 1];
    }
    energy[r] = -0.5 * (double)(float)e + (-(double)(float)hs);
}
hipError_t launch_energy_finish(const double *partial, int slices, double *energy, int R,
                                hipStream_t st) {
    if (slices <= 1) return hipSuccess;
    hipLaunchKernelGGL(energy_finish_kernel, dim3((R + 255) / 256), dim3(256), 0, st, partial, slices,
                       energy, R);
    return hipGetLastError();
}

template <typename JT>
__global__ void __launch_bounds__(256) energy_dense_kernel(const EnergyArgs a) {
    constexpr int EPL = 16 / sizeof(JT), EPC = 64 * EPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int8_t *s = reinterpret_cast<int8_t *>(smem);
    double *red = reinterpret_cast<double *>(smem + a.sstride);  // [2][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    const int rows_per_slice = (a.n + a.slices - 1) / a.slices;
    const int row0 = blockIdx.y * rows_per_slice, row1 = min(a.n, row0 + rows_per_slice);
    {
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (long long)r * a.sstride);
        int4 *dst = reinterpret_cast<int4 *>(s);
        for (int i = tid; i < a.sstride / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int model = a.reps_per_model > 0 ? (a.replica_base + r) / a.reps_per_model : 0;
    const JT *J = reinterpret_cast<const JT *>(a.J) + model * a.model_stride_j;
    const float *hvec = a.h + (long long)model * a.n;
    double e_acc = 0.0, h_acc = 0.0;
    for (int i = row0 + w; i < row1; i += 4) {
        const JT *row = J + (long long)i * a.ldj;
        double acc = 0.0;
        for (long long c = lane * EPL; c < a.ldj; c += EPC) {
            if constexpr (sizeof(JT) == 4) {
                const float4 x = *reinterpret_cast<const float4 *>(row + c);
                const int sw = *reinterpret_cast<const int *>(s + c);
                acc += (double)(x.x * (float)(int8_t)(sw));
                acc += (double)(x.y * (float)(int8_t)(sw >> 8));
                acc += (double)(x.z * (float)(int8_t)(sw >> 16));
                acc += (double)(x.w * (float)(sw >> 24));
            } else {
                const int4 x = *reinterpret_cast<const int4 *>(row + c);
                const int4 sv = *reinterpret_cast<const int4 *>(s + c);
                int t = __builtin_amdgcn_sdot4(x.x, sv.x, 0, false);
                t = __builtin_amdgcn_sdot4(x.y, sv.y, t, false);
                t = __builtin_amdgcn_sdot4(x.z, sv.z, t, false);
                t = __builtin_amdgcn_sdot4(x.w, sv.w, t, false);
                acc += (double)t;
            }
        }
        const float mv_i = (float)wave_sum(acc);  // torch.mv row, fp32
        const double si = (double)s[i];
        e_acc += (double)mv_i * si;
        h_acc += (double)hvec[i] * si;
    }
    if (lane == 0) {
        red[w] = e_acc;
        red[4 + w] = h_acc;
    }
    __syncthreads();
    if (tid == 0) {
        const double e = (red[0] + red[1]) + (red[2] + red[3]);
        const double hs = (red[4] + red[5]) + (red[6] + red[7]);
        energy_out(a, r, e, hs);
    }
}

hipError_t launch_energy_dense(const EnergyArgs &a, bool j_is_i8, hipStream_t st) {
    const size_t lds = (size_t)a.sstride + 64;
    auto set = [&](const void *f) {
        return ensure_lds_limit(f, lds);
    };
    if (j_is_i8) {
        hipError_t e = set(reinterpret_cast<const void *>(energy_dense_kernel<int8_t>));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(energy_dense_kernel<int8_t>, dim3(a.R, a.slices), dim3(256), lds, st, a);
    } else {
        hipError_t e = set(reinterpret_cast<const void *>(energy_dense_kernel<float>));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(energy_dense_kernel<float>, dim3(a.R, a.slices), dim3(256), lds, st, a);
    }
    return hipGetLastError();
}

// LDS_SPINS = false: spins gathered from HBM / L2 (problems beyond the int8 LDS capacity)
template <bool LDS_SPINS>
__global__ void __launch_bounds__(256) energy_csr_kernel(const EnergyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    const int8_t *s = a.spins + (long long)r * a.sstride;
    double *red = reinterpret_cast<double *>(smem + (LDS_SPINS ? a.sstride : 0));
    if constexpr (LDS_SPINS) {
        const int4 *src = reinterpret_cast<const int4 *>(s);
        int4 *dst = reinterpret_cast<int4 *>(smem);
        for (int i = tid; i < a.sstride / 16; i += blockDim.x) dst[i] = src[i];
        s = reinterpret_cast<const int8_t *>(smem);
        __syncthreads();
    }
    const int rows_per_slice = (a.n + a.slices - 1) / a.slices;
    const int row0 = blockIdx.y * rows_per_slice, row1 = min(a.n, row0 + rows_per_slice);
    double e_acc = 0.0, h_acc = 0.0;
    for (int i = row0 + w; i < row1; i += 4) {
        double acc = 0.0;
        for (long long j = a.rowptr[i] + lane; j < a.rowptr[i + 1]; j += 64)
        {
            const int2 ent = a.cv[j];
            acc += (double)(__int_as_float(ent.y) * (float)s[ent.x]);
        }
        const float mv_i = (float)wave_sum(acc);
        const double si = (double)s[i];
        e_acc += (double)mv_i * si;
        h_acc += (double)a.h[i] * si;
    }
    if (lane == 0) {
        red[w] = e_acc;
        red[4 + w] = h_acc;
    }
    __syncthreads();
    if (tid == 0) {
        const double e = (red[0] + red[1]) + (red[2] + red[3]);
        const double hs = (red[4] + red[5]) + (red[6] + red[7]);
        energy_out(a, r, e, hs);
    }
}

// Problems beyond the int8 LDS capacity: the replica's spins as bits in LDS (as in the sweep
// kernel's BIG form), 16 waves per replica, one row per wave with eight (colidx, val) loads in
// flight per lane.  At the 1000-city TSP instance (4e9 entries, 256 replicas) the HBM-spin
// form above took 10.7 s for the initial energies.
constexpr int ENERGY_BIG_WAVES = 16, ENERGY_BIG_UNROLL = 8;
__global__ void __launch_bounds__(64 * ENERGY_BIG_WAVES) energy_csr_bits_kernel(const EnergyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned int *bits = reinterpret_cast<unsigned int *>(smem);
    double *red = reinterpret_cast<double *>(smem + a.sstride / 8);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    spins_to_bits(a.spins + (long long)r * a.sstride, bits, a.sstride, tid, blockDim.x);
    __syncthreads();
    auto spin_f = [&](int c) -> float { return ((bits[c >> 5] >> (c & 31)) & 1u) ? -1.0f : 1.0f; };
    const int rows_per_slice = (a.n + a.slices - 1) / a.slices;
    const int row0 = blockIdx.y * rows_per_slice, row1 = min(a.n, row0 + rows_per_slice);
    double e_acc = 0.0, h_acc = 0.0;
    for (int i = row0 + w; i < row1; i += ENERGY_BIG_WAVES) {
        const long long beg = a.rowptr[i], end = a.rowptr[i + 1];
        double acc = 0.0;
        for (long long j0 = beg + lane; j0 < end; j0 += 64 * ENERGY_BIG_UNROLL) {
            int c[ENERGY_BIG_UNROLL];
            float v[ENERGY_BIG_UNROLL];
#pragma unroll
            for (int q = 0; q < ENERGY_BIG_UNROLL; ++q) {
                const long long j = j0 + 64 * q;
                const bool in = j < end;
                const int2 ent = in ? a.cv[j] : make_int2(0, 0);
                c[q] = ent.x;
                v[q] = __int_as_float(ent.y);
            }
#pragma unroll
            for (int q = 0; q < ENERGY_BIG_UNROLL; ++q) acc += (double)(v[q] * spin_f(c[q]));
        }
        const float mv_i = (float)wave_sum(acc);
        const double si = (double)spin_f(i);
        e_acc += (double)mv_i * si;
        h_acc += (double)a.h[i] * si;
    }
    if (lane == 0) {
        red[w] = e_acc;
        red[ENERGY_BIG_WAVES + w] = h_acc;
    }
    __syncthreads();
    if (tid == 0) {
        double e = 0.0, hs = 0.0;
        for (int q = 0; q < ENERGY_BIG_WAVES; ++q) {
            e += red[q];
            hs += red[ENERGY_BIG_WAVES + q];
        }
        energy_out(a, r, e, hs);
    }
}

hipError_t launch_energy_csr(const EnergyArgs &a, hipStream_t st) {
    const size_t lds = (size_t)a.sstride + 64;
    if (lds > 160 * 1024 - 256) {
        const size_t lds_bits = (size_t)a.sstride / 8 + 2 * ENERGY_BIG_WAVES * sizeof(double);
        if (a.sstride % 128 == 0 && lds_bits <= 160 * 1024 - 256) {
            hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(energy_csr_bits_kernel), lds_bits);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(energy_csr_bits_kernel, dim3(a.R, a.slices), dim3(64 * ENERGY_BIG_WAVES), lds_bits,
                               st, a);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(energy_csr_kernel<false>, dim3(a.R, a.slices), dim3(256), 64, st, a);
        return hipGetLastError();
    }
    {
        hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(energy_csr_kernel<true>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(energy_csr_kernel<true>, dim3(a.R, a.slices), dim3(256), lds, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Replica exchange, nearest neighbours (annealing/parallel_tempering.py:214-258).
// Slots carry temperatures; slot_to_rep names the configuration in each slot.  All pairs of
// one parity are disjoint, so one thread per pair; an accepted swap exchanges the two
// slot_to_rep entries (== the reference swapping the spin tensors) and rewrites the two
// replicas' temperatures.  Decisions are a pure function of (energies, seed, round): every
// rank of a multi-GPU run evaluates the same ones on the all-gathered energies.
// ---------------------------------------------------------------------------------------
__global__ void exchange_neighbor_kernel(const ExchangeArgs a) {
    const int L = a.R_global / a.n_ladders;
    const int half = L / 2;
    const int total = a.n_ladders_local * half;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += gridDim.x * blockDim.x) {
        const int ll = idx / half, q = idx - ll * half;
        const int l = a.ladder0 + ll;  // global ladder: keys the Philox draws and indexes start / u
        int start;
        if (a.start) {
            start = a.start[l] & 1;
        } else {  // np.random.randint(0, 2), parallel_tempering.py:217
            const u32x4 w = philox4x32_10(0xFFFFFFFFu, a.round, (uint32_t)l, DOMAIN_EXCHANGE,
                                          a.seed_lo, a.seed_hi);
            start = (int)(w.x & 1u);
        }
        const int j = start + 2 * q;
        if (j + 1 >= L) continue;
        const int i = l * L + j;  // global lower slot of the pair
        // _attempt_single_exchange, parallel_tempering.py:234-258
        const double beta_i = 1.0 / a.slot_temps[i], beta_j = 1.0 / a.slot_temps[i + 1];
        const int ri = a.slot_to_rep[i], rj = a.slot_to_rep[i + 1];
        const double x = (beta_j - beta_i) * (a.energies[rj - a.energy_base] - a.energies[ri - a.energy_base]);
        const double prob = (x >= 0.0) ? 1.0 : exp_det(x);
        double uu;
        if (a.u) {
            uu = a.u[l * half + q];
        } else {
            const u32x4 w = philox4x32_10((uint32_t)j, a.round, (uint32_t)l, DOMAIN_EXCHANGE,
                                          a.seed_lo, a.seed_hi);
            uu = words_to_u53(w.x, w.y);
        }
        a.attempts[i] += 1;
        int new_i = ri, new_j = rj;
        if (uu < prob) {
            new_i = rj;
            new_j = ri;
            a.slot_to_rep[i] = new_i;
            a.slot_to_rep[i + 1] = new_j;
            a.accepts[i] += 1;
            atomicAdd(a.n_accepted, 1);
        }
        const int li = new_i - a.replica0, lj = new_j - a.replica0;
        if (li >= 0 && li < a.R_local) a.rep_temp[li] = a.slot_temps[i];
        if (lj >= 0 && lj < a.R_local) a.rep_temp[lj] = a.slot_temps[i + 1];
    }
}

hipError_t launch_exchange_neighbor(const ExchangeArgs &a, hipStream_t st) {
    const int L = a.R_global / a.n_ladders;
    const int total = a.n_ladders_local * (L / 2);
    if (total <= 0) return hipSuccess;
    const int blocks = (total + 255) / 256;
    hipLaunchKernelGGL(exchange_neighbor_kernel, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

// Exchange attempts over an explicit, ordered list of slot pairs -- the reference's
// exchange_method="all_pairs" (annealing/parallel_tempering.py:222-232: for i < j, gated by
// rand() < 0.1, _attempt_single_exchange(i, j)): every attempt sees the swaps before it, so the
// list is one serial chain (thread 0); the local temperatures are rewritten by the workgroup.
__global__ void __launch_bounds__(256) exchange_pairs_kernel(const ExchangeArgs a, const int32_t *pairs,
                                                             int count) {
    if (threadIdx.x == 0) {
        int cnt = 0;
        for (int k = 0; k < count; ++k) {
            const int i = pairs[2 * k], j = pairs[2 * k + 1];
            // _attempt_single_exchange, parallel_tempering.py:234-258
            const double beta_i = 1.0 / a.slot_temps[i], beta_j = 1.0 / a.slot_temps[j];
            const int ri = a.slot_to_rep[i], rj = a.slot_to_rep[j];
            const double x = (beta_j - beta_i) * (a.energies[rj] - a.energies[ri]);
            const double prob = (x >= 0.0) ? 1.0 : exp_det(x);
            double uu;
            if (a.u) {
                uu = a.u[k];
            } else {
                const u32x4 w = philox4x32_10(0x40000000u | (uint32_t)k, a.round, 0u, DOMAIN_EXCHANGE,
                                              a.seed_lo, a.seed_hi);
                uu = words_to_u53(w.x, w.y);
            }
            const int lo = i < j ? i : j;  // pair_idx = min(i, j), parallel_tempering.py:249
            a.attempts[lo] += 1;
            if (uu < prob) {
                a.slot_to_rep[i] = rj;
                a.slot_to_rep[j] = ri;
                a.accepts[lo] += 1;
                ++cnt;
            }
        }
        *a.n_accepted = cnt;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < a.R_global; s += blockDim.x) {
        const int l = a.slot_to_rep[s] - a.replica0;
        if (l >= 0 && l < a.R_local) a.rep_temp[l] = a.slot_temps[s];
    }
}
hipError_t launch_exchange_pairs(const ExchangeArgs &a, const int32_t *pairs, int count, hipStream_t st) {
    hipLaunchKernelGGL(exchange_pairs_kernel, dim3(1), dim3(256), 0, st, a, pairs, count);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Single-site operators: one workgroup, one coupling row per site, spins read from HBM.
// ---------------------------------------------------------------------------------------
template <typename JT, bool CSR>
__global__ void __launch_bounds__(256) point_op_kernel(const PointArgs a) {
    constexpr int EPL = 16 / sizeof(JT), EPC = 64 * EPL;
    __shared__ double red[8];
    // dense fp32: per-super-chunk sums of the canonical summation order (n <= 163 840: 160 of them)
    __shared__ double csum[(!CSR && sizeof(JT) == 4) ? 640 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fields (op 0) are independent: one workgroup per requested site; flips / updates mutate the
    // replica and stay one serial chain in one workgroup
    const int k_begin = a.op == 0 ? (int)blockIdx.x : 0, k_end = a.op == 0 ? k_begin + 1 : a.count;
    for (int k = k_begin; k < k_end; ++k) {
        const int site = a.sites[k];
        float dot;
        __syncthreads();  // the previous site's sums have been read
        if constexpr (CSR) {
            // canonical order of sweep_csr.hip: entry e -> lane e % 64 of virtual wave (e / 64) % 8;
            // thread tid takes e = tid + 256 m, i.e. virtual waves w (m even) and w + 4 (m odd)
            const long long beg = a.rowptr[site];
            const int len = (int)(a.rowptr[site + 1] - beg);
            double acc0 = 0.0, acc1 = 0.0;
            for (int e0 = tid; e0 < len; e0 += 512) {
                const int2 ent = a.cv[beg + e0];
                acc0 += (double)(__int_as_float(ent.y) * (float)a.spins[ent.x]);
                if (e0 + 256 < len) {
                    const int2 en2 = a.cv[beg + e0 + 256];
                    acc1 += (double)(__int_as_float(en2.y) * (float)a.spins[en2.x]);
                }
            }
            const double s0 = wave_sum(acc0), s1 = wave_sum(acc1);
            if (lane == 0) {
                red[w] = s0;
                red[w + 4] = s1;
            }
            __syncthreads();
            double t = red[0];
            for (int v = 1; v < 8; ++v) t += red[v];
            dot = (float)t;
        } else if constexpr (sizeof(JT) == 4) {
            // canonical order of sweep_dense_impl.h: 1024-element super-chunks; lane l adds its 16
            // products (chunk j = 0..3, elements 4l..4l+3 of each) from +0, adjacent-pairs tree,
            // super-chunk sums added in order
            const JT *row = reinterpret_cast<const JT *>(a.J) + a.model_offset_j + (long long)site * a.ldj;
            const int C = (a.n + 4 * EPC - 1) / (4 * EPC);
            for (int c = w; c < C; c += 4) {
                double p = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long col = ((long long)c * 4 + j) * EPC + lane * EPL;
                    if (col < a.ldj) {  // J's row pad and the spins' pad are zero
                        const float4 x = *reinterpret_cast<const float4 *>(row + col);
                        const int sw = *reinterpret_cast<const int *>(a.spins + col);
                        p += (double)(x.x * (float)(int8_t)(sw));
                        p += (double)(x.y * (float)(int8_t)(sw >> 8));
                        p += (double)(x.z * (float)(int8_t)(sw >> 16));
                        p += (double)(x.w * (float)(sw >> 24));
                    }
                }
                const double cs = wave_sum(p);
                if (lane == 0) csum[c] = cs;
            }
            __syncthreads();
            double t = csum[0];
            for (int c = 1; c < C; ++c) t += csum[c];
            dot = (float)t;
        } else {  // int8: exact in any order
            const JT *row = reinterpret_cast<const JT *>(a.J) + a.model_offset_j + (long long)site * a.ldj;
            int acc = 0;
            for (long long c = (long long)tid * EPL; c < a.ldj; c += 4 * EPC) {
                const int4 x = *reinterpret_cast<const int4 *>(row + c);
                const int4 sv = *reinterpret_cast<const int4 *>(a.spins + c);
                acc = __builtin_amdgcn_sdot4(x.x, sv.x, acc, false);
                acc = __builtin_amdgcn_sdot4(x.y, sv.y, acc, false);
                acc = __builtin_amdgcn_sdot4(x.z, sv.z, acc, false);
                acc = __builtin_amdgcn_sdot4(x.w, sv.w, acc, false);
            }
            const int ws = wave_sum(acc);
            if (lane == 0) red[w] = (double)ws;
            __syncthreads();
            dot = (float)((red[0] + red[1]) + (red[2] + red[3]));
        }
        if (a.op == 0) {
            if (tid == 0) a.out[k] = (double)dot + (double)a.h[site];  // ising_model.py:176-185
            continue;
        }
        const int si = a.spins[site];
        double dE;
        bool flip;
        if (a.op == 1) {  // IsingModel.flip_spin, ising_model.py:125-147
            dE = 2.0 * (double)si * ((double)dot + (double)a.h[site]);
            flip = true;
        } else {
            flip = metropolis_accept(a.rule, a.arith, dot, si, a.h[site], a.diag[site], a.T, a.u, dE);
        }
        __syncthreads();  // every thread has read s[site]
        if (tid == 0) {
            if (flip) {
                a.spins[site] = (int8_t)(-si);
                *a.energy += dE;
                if (a.op == 2) *a.n_accepted += 1;
            }
            a.out[0] = (a.op == 2 && a.rule == SGA_RULE_HEAT_BATH) ? -dE : dE;
            a.out[1] = flip ? 1.0 : 0.0;
        }
    }
}

hipError_t launch_point_op(const PointArgs &a, bool csr, bool j_is_i8, hipStream_t st) {
    const dim3 grid(a.op == 0 ? (unsigned)(a.count > 0 ? a.count : 1) : 1u);
    if (csr)
        hipLaunchKernelGGL((point_op_kernel<float, true>), grid, dim3(256), 0, st, a);
    else if (j_is_i8)
        hipLaunchKernelGGL((point_op_kernel<int8_t, false>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((point_op_kernel<float, false>), grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Operator-form exchange, annealing/cuda_kernels.py:415-443: the pairs are visited in order
// and each decision sees the swaps before it, so the decisions are one serial chain (one
// thread); the row permutation they compose is then applied by the whole grid.
// ---------------------------------------------------------------------------------------
__global__ void op_exchange_decide_kernel(float *energies, const float *temps, const float *u,
                                          int32_t *src_of_pos, int *n_accepted, uint32_t seed_lo,
                                          uint32_t seed_hi, uint32_t round, int R) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int i = 0; i < R; ++i) src_of_pos[i] = i;
    int cnt = 0;
    for (int i = 0; i + 1 < R; ++i) {
        const float beta1 = 1.0f / temps[i], beta2 = 1.0f / temps[i + 1];
        const float db = beta2 - beta1;
        const float de = energies[i] - energies[i + 1];
        const float prob = expf_det(db * de);
        float uu;
        if (u) {
            uu = u[i];
        } else {
            const u32x4 w = philox4x32_10((uint32_t)i, round, 0u, DOMAIN_EXCHANGE, seed_lo, seed_hi);
            uu = word_to_u(w.x);
        }
        if (uu < prob) {
            const float e = energies[i];
            energies[i] = energies[i + 1];
            energies[i + 1] = e;
            const int32_t t = src_of_pos[i];
            src_of_pos[i] = src_of_pos[i + 1];
            src_of_pos[i + 1] = t;
            ++cnt;
        }
    }
    *n_accepted = cnt;
}
__global__ void op_exchange_gather_kernel(const float *spins, float *tmp, const int32_t *src_of_pos,
                                          int R, int n) {
    const long long total = (long long)R * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pos = i / n, c = i - pos * n;
        tmp[i] = spins[(long long)src_of_pos[pos] * n + c];
    }
}
hipError_t launch_op_exchange(float *spins, float *tmp_rows, float *energies, const float *temps,
                              const float *u, int32_t *src_of_pos, int *n_accepted,
                              uint32_t seed_lo, uint32_t seed_hi, uint32_t round, int R, int n,
                              hipStream_t st) {
    hipLaunchKernelGGL(op_exchange_decide_kernel, dim3(1), dim3(64), 0, st, energies, temps, u,
                       src_of_pos, n_accepted, seed_lo, seed_hi, round, R);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(op_exchange_gather_kernel, dim3(grid_for((long long)R * n)), dim3(256), 0,
                       st, spins, tmp_rows, src_of_pos, R, n);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(spins, tmp_rows, sizeof(float) * (size_t)R * n, hipMemcpyDeviceToDevice,
                          st);
}

}  // namespace sga

namespace sga {
// ---------------------------------------------------------------------------------------
// Streaming-read probe: what this box delivers to a plain 16-byte-per-lane read of a buffer far
// larger than the caches -- the practical denominator beside the 8 TB/s spec figure.
// ---------------------------------------------------------------------------------------
// Round 4 (profiles/src/probe_bw.hip, profiles/r04_probe_bw.txt: 56 patterns on one box): every workgroup streams
// its own contiguous segment with eight independent NON-TEMPORAL 16-byte loads in flight per lane -- 7.1-7.2 TB/s;
// the same with plain loads 6.4-6.5 TB/s; round 3's grid-stride pattern with four plain loads 5.8 TB/s, below what
// the dense sweep kernel itself sustains beyond the caches (6.65 TB/s), i.e. not a ceiling.
__global__ void __launch_bounds__(256) probe_read_kernel(const float4 *__restrict__ x, long long n4, long long seg4,
                                                         float *sink) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const long long base = (long long)blockIdx.x * seg4;
    const long long end = base + seg4 < n4 ? base + seg4 : n4;
    const long long step = blockDim.x;
    float acc = 0.0f;
    long long i = base + threadIdx.x;
    for (; i + 7 * step < end; i += 8 * step) {
        f4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&x[i + q * step]));
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
    for (; i < end; i += step) {
        const float4 a = x[i];
        acc += a.x + a.y + a.z + a.w;
    }
    if (acc == 123456.789f) *sink = acc;  // keeps the loads alive
}
// Position-weighted 64-bit checksum of a word array (commutative accumulation: deterministic):
// sum_i (w_i + 1) * (2 i + 1) * 0x9E3779B97F4A7C15 mod 2^64, added to *out.
__global__ void __launch_bounds__(256) checksum_kernel(const unsigned int *w, long long count, unsigned long long *out) {
    unsigned long long acc = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        acc += ((unsigned long long)w[i] + 1ull) * (2ull * (unsigned long long)i + 1ull) * 0x9E3779B97F4A7C15ull;
    // wave sum through two 32-bit halves would lose the carries: add lane by lane with readlane
    unsigned long long tot = 0;
    for (int l = 0; l < 64; ++l) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)acc, l);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(acc >> 32), l);
        tot += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(out, tot);
}
hipError_t launch_checksum(const void *buf, long long bytes, unsigned long long *out, hipStream_t st) {
    const long long words = bytes / 4;
    if (words <= 0) return hipSuccess;
    const int blocks = (int)std::min<long long>(4096, (words + 255) / 256);
    hipLaunchKernelGGL(checksum_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned int *>(buf), words,
                       out);
    return hipGetLastError();
}

hipError_t launch_probe_read(const void *buf, long long bytes, float *sink, hipStream_t st) {
    const long long n4 = bytes / 16;
    const int blocks = 256 * 128;  // 128 segments per CU
    hipLaunchKernelGGL(probe_read_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float4 *>(buf), n4,
                       (n4 + blocks - 1) / blocks, sink);
    return hipGetLastError();
}
}  // namespace sga

namespace sga {
static thread_local char g_sweep_kernel[448] = "";
void note_sweep_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_sweep_kernel, sizeof(g_sweep_kernel), fmt, ap);
    va_end(ap);
}
const char *last_sweep_kernel() { return g_sweep_kernel; }

hipError_t ensure_lds_limit(const void *kernel, size_t lds_bytes) {
    if (lds_bytes <= 48 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[{dev, kernel}];
    if (have >= lds_bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) have = lds_bytes;
    return e;
}
}  // namespace sga
