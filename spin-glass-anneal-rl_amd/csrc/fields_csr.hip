// fields_csr.hip -- energies of ALL replicas of a CSR problem in one pass over the entries.
//
// Replaces, for batches of replicas: IsingModel.compute_energy (core/ising_model.py:149-174) /
// EnergyComputer.compute_batch_energies (core/energy_computer.py:142-158) on sparse couplings.  The
// per-replica kernels of sga_misc.hip read every entry once PER REPLICA (BASELINE configs[3]: 97.8 ms for
// the 1024 initial energies of the 239 MB layout; configs[4] at 1000 cities: ~1 s for 256 replicas of
// the 32 GB layout).  Here the spins are first transposed into a bit matrix Sb[site][replica / 32]
// (bit = spin down), so that ONE read of an entry (column, value) serves 32 replicas per lane:
//     acc_b += value * s_b,   s_b from bit b of Sb[column][word],   b = 0..31
// -- 3 VALU operations per (entry, replica), the entries read once per group of rows, the bit matrix
// (n x R / 8 bytes) cache resident.  Thread (g, q) walks the rows i = g, g + G, ... for the replica
// word q; a row's sum is rounded to fp32 as torch.mv rounds it, multiplied by the row's own spin and
// accumulated in fp64; the per-group partial sums are added in group order by the finish pass
// (deterministic), which also forms  E_r = -1/2 fp32(sum_i mv_ri s_ri) - fp32(h . s_r).
// EXACT32: integer couplings with row sums below 2^24 accumulate in fp32 (exact); otherwise in fp64.
#include <type_traits>

#include "sga_device.h"
#include "sga_kernels.h"

namespace sga {

// Sb[i][q] bit b = 1 iff spin (32 q + b) at site i is -1; replicas beyond R read as +1
__global__ void __launch_bounds__(256) transpose_spin_bits_kernel(const int8_t *__restrict__ spins, int sstride, int n,
                                                                  int R, int RW, unsigned int *__restrict__ sb) {
    const int i = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
    if (i >= n) return;
    unsigned int w = 0;
#pragma unroll 8
    for (int b = 0; b < 32; ++b) {
        const int r = 32 * q + b;
        if (r < R) w |= (unsigned int)(spins[(long long)r * sstride + i] < 0) << b;
    }
    sb[(long long)i * RW + q] = w;
}

// hs[r] = sum_i h_i s_ri in fp64, fixed order (lane-strided, tree, waves in order)
__global__ void __launch_bounds__(256) field_dot_kernel(const float *__restrict__ h, const int8_t *__restrict__ spins,
                                                        int sstride, int n, double *__restrict__ hs) {
    __shared__ double red[4];
    const int tid = threadIdx.x, r = blockIdx.x;
    const int8_t *s = spins + (long long)r * sstride;
    double acc = 0.0;
    for (int i = tid; i < n; i += 256) acc += (double)h[i] * (double)s[i];
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) hs[r] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <bool EXACT32>
__global__ void __launch_bounds__(256) energy_csr_all_kernel(const CsrEnergyArgs a) {
    using acc_t = typename std::conditional<EXACT32, float, double>::type;
    const long long T = (long long)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(T / a.RW), q = (int)(T % a.RW);
    if (g >= a.groups) return;
    double e[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) e[b] = 0.0;
    for (int i = g; i < a.n; i += a.groups) {
        const long long beg = a.rowptr[i], end = a.rowptr[i + 1];
        acc_t acc[32];
#pragma unroll
        for (int b = 0; b < 32; ++b) acc[b] = 0;
        for (long long j = beg; j < end; j += 4) {  // four entries' loads in flight
            int2 ent[4];
            unsigned int w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ent[u] = j + u < end ? a.cv[j + u] : make_int2(0, 0);  // (value 0: adds nothing)
                w[u] = a.sb[(long long)ent[u].x * a.RW + q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int vbits = ent[u].y;
#pragma unroll
                for (int b = 0; b < 32; ++b) {
                    // value * (+-1): the spin bit goes into the value's sign bit (exact)
                    const int sv = vbits ^ (int)((w[u] >> b) << 31);
                    acc[b] += (acc_t)__int_as_float(sv);
                }
            }
        }
        const unsigned int own = a.sb[(long long)i * a.RW + q];
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const float mv = (float)acc[b];  // torch.mv row, fp32
            e[b] += ((own >> b) & 1u) ? -(double)mv : (double)mv;
        }
    }
    double *out = a.partial + (long long)g * (32ll * a.RW) + 32ll * q;
#pragma unroll
    for (int b = 0; b < 32; ++b) out[b] = e[b];
}

__global__ void energy_csr_all_finish_kernel(const double *__restrict__ partial, const double *__restrict__ hs, int groups,
                                             int RW, int R, double *__restrict__ energy) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double e = 0.0;
    for (int g = 0; g < groups; ++g) e += partial[(long long)g * (32ll * RW) + r];  // group order: deterministic
    energy[r] = -0.5 * (double)(float)e + (-(double)(float)hs[r]);
}

size_t csr_energy_scratch_bytes(int n, int R, int groups) {
    const size_t RW = ((size_t)R + 31) / 32;
    return sizeof(unsigned int) * (((size_t)n * RW + 1) & ~(size_t)1) + sizeof(double) * (size_t)groups * 32 * RW +
           sizeof(double) * (size_t)R;
}

// scratch: [n][RW] spin-bit words | [groups][32 RW] partial sums | [R] field dots
hipError_t launch_energy_csr_all(const long long *rowptr, const int2 *cv, const float *h, const int8_t *spins, int sstride,
                                 int n, int R, int groups, bool exact32, void *scratch, double *energy, hipStream_t st) {
    const int RW = (R + 31) / 32;
    unsigned int *sb = static_cast<unsigned int *>(scratch);
    double *partial = reinterpret_cast<double *>(sb + (((size_t)n * RW + 1) & ~(size_t)1));
    double *hs = partial + (size_t)groups * 32 * RW;
    hipLaunchKernelGGL(transpose_spin_bits_kernel, dim3((n + 255) / 256, RW), dim3(256), 0, st, spins, sstride, n, R, RW, sb);
    hipLaunchKernelGGL(field_dot_kernel, dim3(R), dim3(256), 0, st, h, spins, sstride, n, hs);
    CsrEnergyArgs a{rowptr, cv, sb, partial, n, R, RW, groups};
    const long long threads = (long long)groups * RW;
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (exact32) hipLaunchKernelGGL(energy_csr_all_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(energy_csr_all_kernel<false>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(energy_csr_all_finish_kernel, dim3((R + 255) / 256), dim3(256), 0, st, partial, hs, groups, RW, R,
                       energy);
    return hipGetLastError();
}

}  // namespace sga
