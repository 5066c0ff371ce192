// sweep_wolff.hip -- the Wolff cluster rule (UpdateRule.WOLFF, core/spin_dynamics.py:193-323) on the
// device: SpinDynamics._wolff_cluster_dense restated for one wavefront per replica.
//
// Reference semantics kept (spin_dynamics.py:210-255):
//   * a sweep is n cluster moves, each grown from a uniformly drawn start site (:69, :82-85);
//   * breadth-first growth in FIFO order; the neighbours of the current site are visited in index
//     order; a neighbour joins with probability 1 - exp(float32(2 J / T)) iff J < 0 (the reference's
//     sign: its "ferromagnetic" bond is J < 0 although H = -1/2 sum J s s), the two spins are equal
//     and it is not in the cluster yet; ONE uniform is drawn per such candidate, none otherwise;
//   * the whole cluster is flipped, the move is always accepted, n_accepted grows by the cluster size;
//   * the energy change it reports is compute_energy() after minus before (both rounded as
//     core/ising_model.py:161-168 rounds them); the sweep's energy is compute_energy().
// For CSR couplings the neighbours are the row's stored entries in storage order (the dense rule
// restricted to the non-zero couplings; the reference's own sparse branch does not run).
//
// Mapping: the replica's spins, the cluster bitmap and the FIFO queue live in LDS (5.2 n bytes:
// n <= ~31 000).  64 candidates are tested per step; their draw indices and queue positions are
// prefix counts of wave ballots, so the draw order is the reference's.  Uniforms: Philox domain 3,
// counter (draw >> 2, sweep, replica, 3 | update << 2), or a recorded stream (sga_set_wolff_replay).
// This rule is here for completeness of the reference's API, not for throughput.
#include "sweep_common.h"

namespace sga {

constexpr uint32_t DOMAIN_WOLFF = 3;

template <typename JT, bool CSR>
__global__ void __launch_bounds__(64) sweep_wolff_kernel(const SweepArgs a, const WolffArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = a.n, lane = threadIdx.x, r = blockIdx.x;
    int8_t *s = reinterpret_cast<int8_t *>(smem);                                  // [n]
    const int words = (n + 31) / 32;
    unsigned int *inc = reinterpret_cast<unsigned int *>(smem + ((n + 15) & ~15)); // cluster bitmap
    int *queue = reinterpret_cast<int *>(inc + ((words + 3) & ~3));                // FIFO [n]
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int model = a.reps_per_model > 0 ? (int)((a.replica0 + (uint32_t)r) / a.reps_per_model) : 0;
    const JT *Jm = reinterpret_cast<const JT *>(a.J) + (CSR ? 0 : model * a.model_stride_j);
    const float *hvec = a.h + (long long)model * n;
    int8_t *hbm_spins = a.spins + (long long)r * a.sstride;
    for (int i = lane; i < n; i += 64) s[i] = hbm_spins[i];
    __syncthreads();

    // J[i,:] . s rounded to fp32 in the canonical order of the sweep kernels (one wave does all of it)
    auto row_dot = [&](int i) -> float {
        if constexpr (CSR) {
            const long long beg = a.rowptr64[i];
            const int len = (int)(a.rowptr64[i + 1] - beg);
            double acc[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) acc[v] = 0.0;
            for (int e0 = 0; e0 < len; e0 += 512) {
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const int e = e0 + 64 * v + lane;
                    if (e < len) {
                        const int2 ent = a.cv[beg + e];
                        acc[v] += (double)(__int_as_float(ent.y) * (float)s[ent.x]);
                    }
                }
            }
            double t = wave_sum(acc[0]);
#pragma unroll
            for (int v = 1; v < 8; ++v) t += wave_sum(acc[v]);
            return (float)t;
        } else {
            const JT *row = Jm + (long long)i * a.ldj;
            double t = 0.0;
            for (int c0 = 0; c0 < n; c0 += 1024) {  // 1024-element super-chunks (sweep_dense_impl.h)
                double p = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int j = c0 + 256 * (q >> 2) + 4 * lane + (q & 3);
                    if (j < n) p += (double)((float)row[j] * (float)s[j]);
                }
                const double cs = wave_sum(p);
                t = c0 == 0 ? cs : t + cs;
            }
            return (float)t;
        }
    };
    // IsingModel.compute_energy, core/ising_model.py:149-174
    auto full_energy = [&]() -> double {
        double e = 0.0, hs = 0.0;
        for (int i = 0; i < n; ++i) {
            const float mv = row_dot(i);
            e += (double)mv * (double)s[i];
            hs += (double)hvec[i] * (double)s[i];
        }
        return -0.5 * (double)(float)e + (-(double)(float)hs);
    };

    const bool traced = a.dE_trace != nullptr;
    double E_prev = traced ? full_energy() : 0.0;
    unsigned long long nacc = 0;
    long long cursor = wa.replay_u ? wa.cursor[r] : 0;
    PairSource<false> rng;  // sites from the same streams as every other rule

    for (int k = 0; k < a.n_sweeps; ++k) {
        const double T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        UpdatePair pair{0, 0, 2.0f, 2.0f};
        for (int t = 0; t < n; ++t) {
            if ((t & 1) == 0) pair = rng.get(a, r, k, t >> 1, true, lane);
            const int start = (t & 1) ? pair.sB : pair.sA;
            for (int i = lane; i < words; i += 64) inc[i] = 0u;
            __syncthreads();
            if (lane == 0) {
                queue[0] = start;
                inc[start >> 5] = 1u << (start & 31);
            }
            __syncthreads();
            int head = 0, tail = 1;
            long long drawn = 0;
            while (head < tail) {
                const int cur = queue[head];
                ++head;
                const int sc = s[cur];
                long long beg = 0;
                int len = n;
                if constexpr (CSR) {
                    beg = a.rowptr64[cur];
                    len = (int)(a.rowptr64[cur + 1] - beg);
                }
                for (int b0 = 0; b0 < len; b0 += 64) {
                    const int idx = b0 + lane;
                    int j = 0;
                    float Jv = 0.0f;
                    if (idx < len) {
                        if constexpr (CSR) {
                            const int2 ent = a.cv[beg + idx];
                            j = ent.x;
                            Jv = __int_as_float(ent.y);
                        } else {
                            j = idx;
                            Jv = (float)(Jm + (long long)cur * a.ldj)[idx];
                        }
                    }
                    const bool cand = idx < len && j != cur && Jv < 0.0f && s[j] == sc &&
                                      !((inc[j >> 5] >> (j & 31)) & 1u);
                    const unsigned long long m = __ballot(cand);
                    if (m == 0ull) continue;  // wave-uniform
                    const long long kd = drawn + __popcll(m & lt);  // this candidate's draw index
                    float u = 2.0f;
                    if (cand) {
                        if (wa.replay_u) {
                            const long long at = cursor + kd;
                            u = at < wa.capacity ? wa.replay_u[(long long)r * wa.capacity + at] : 2.0f;
                        } else {
                            const u32x4 w = philox4x32_10((uint32_t)(kd >> 2), a.sweep0 + (uint32_t)k,
                                                          a.replica0 + (uint32_t)r, DOMAIN_WOLFF | ((uint32_t)t << 2),
                                                          a.seed_lo, a.seed_hi);
                            const uint32_t wd = (kd & 3) == 0 ? w.x : (kd & 3) == 1 ? w.y : (kd & 3) == 2 ? w.z : w.w;
                            u = word_to_u(wd);
                        }
                    }
                    // prob_add = 1.0 - torch.exp(torch.tensor(2.0 * coupling / T))   (fp32 tensor)
                    const float p_add = 1.0f - expf_det((float)(2.0 * (double)Jv / T));
                    const bool add = cand && u < p_add;
                    const unsigned long long am = __ballot(add);
                    if (add) {
                        queue[tail + __popcll(am & lt)] = j;
                        atomicOr(&inc[j >> 5], 1u << (j & 31));
                    }
                    tail += __popcll(am);
                    drawn += __popcll(m);
                    __syncthreads();  // (one wave: orders the LDS writes before the next reads)
                }
            }
            for (int i = lane; i < tail; i += 64) {
                const int j = queue[i];
                s[j] = (int8_t)(-s[j]);
            }
            __syncthreads();
            nacc += (unsigned long long)tail;
            cursor += drawn;
            if constexpr (true) {
                const long long upd = (long long)k * n + t;
                if (traced) {
                    const double E_now = full_energy();
                    if (lane == 0) a.dE_trace[(long long)r * a.replay_stride + upd] = E_now - E_prev;
                    E_prev = E_now;
                }
                if (lane == 0 && a.accept_trace) a.accept_trace[(long long)r * a.replay_stride + upd] = 1;
            }
        }
    }
    for (int i = lane; i < a.sstride; i += 64) hbm_spins[i] = i < n ? s[i] : (int8_t)0;
    if (lane == 0) {
        a.n_accepted[r] += nacc;
        if (wa.replay_u) wa.cursor[r] = cursor;
    }
}

size_t wolff_lds_bytes(int n) {
    const size_t words = ((size_t)n + 31) / 32;
    return (((size_t)n + 15) & ~(size_t)15) + 4 * ((words + 3) & ~(size_t)3) + 4 * (size_t)n;
}

hipError_t launch_sweep_wolff(const SweepArgs &a, const WolffArgs &wa, bool csr, bool j_is_i8, hipStream_t st) {
    const size_t lds = wolff_lds_bytes(a.n);
    void (*kern)(const SweepArgs, const WolffArgs) =
        csr ? sweep_wolff_kernel<float, true>
            : (j_is_i8 ? sweep_wolff_kernel<int8_t, false> : sweep_wolff_kernel<float, false>);
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.R), dim3(64), lds, st, a, wa);
    return hipGetLastError();
}

}  // namespace sga
