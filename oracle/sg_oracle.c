/*
 * sg_oracle.c -- CPU restatement of the reference's Ising spin-sweep hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP kernels and the reported CPU baseline.
 * The product (spin-glass-anneal-rl_amd/) never includes, links or calls this file.
 * Parity: PINNED against tests/golden/ npz files captured from the imported reference.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: every fused multiply-add below is an
 * explicit fma()/fmaf() so the arithmetic is the same on every compiler and on the GPU).
 */
#include "sg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------
 * Philox4x32-10, Salmon/Moraes/Dror/Shaw, "Parallel random numbers: as easy as 1, 2, 3"
 * (SC'11).  Third-party algorithm restated from the paper; pinned by the Random123
 * known-answer vectors in tests/test_oracle_golden.py.
 * ---------------------------------------------------------------------------------- */
void sgo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Stream layout of the build's production RNG (DESIGN.md "Random streams"):
 *   key = (seed lo, seed hi); ctr = (block, sweep|round, replica, domain)
 *   domain 0: sweep updates, block = t>>1; update t uses words (2*(t&1), 2*(t&1)+1)
 *   domain 1: exchange decisions, block = lower slot of the pair within its ladder
 *             (0xFFFFFFFF = parity draw), replica field = ladder index
 *   domain 2: initial spins, block = i>>7, bit i&127 of the 128-bit block          */
static inline void stream_block(uint64_t seed, uint32_t block, uint32_t sweep, uint32_t replica,
                                uint32_t domain, uint32_t out[4]) {
    uint32_t ctr[4] = {block, sweep, replica, domain};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    sgo_philox4x32_10(ctr, key, out);
}
static inline uint32_t word_to_site(uint32_t w, uint32_t n) {
    return (uint32_t)(((uint64_t)w * n) >> 32);
}
static inline float word_to_u(uint32_t w) { return (float)(w >> 8) * 0x1.0p-24f; }

uint32_t sgo_stream_site(uint64_t seed, uint32_t replica, uint32_t sweep, uint32_t t, uint32_t n) {
    uint32_t o[4];
    stream_block(seed, t >> 1, sweep, replica, 0, o);
    return word_to_site(o[2 * (t & 1)], n);
}
float sgo_stream_u(uint64_t seed, uint32_t replica, uint32_t sweep, uint32_t t) {
    uint32_t o[4];
    stream_block(seed, t >> 1, sweep, replica, 0, o);
    return word_to_u(o[2 * (t & 1) + 1]);
}

void sgo_init_spins(int n, int R, uint64_t seed, uint32_t replica0, int8_t *spins) {
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < n; ++i) {
            uint32_t o[4];
            stream_block(seed, (uint32_t)i >> 7, 0, replica0 + (uint32_t)r, 2, o);
            uint32_t bit = (o[(i >> 5) & 3] >> (i & 31)) & 1u;
            spins[(int64_t)r * n + i] = bit ? 1 : -1;
        }
}

/* ------------------------------------------------------------------------------------
 * exp: stands in for torch.exp(float32 tensor) (spin_dynamics.py:145) and np.exp(double)
 * (parallel_tempering.py:246).  Cody-Waite reduction + Taylor/Horner in explicit fma,
 * two-step power-of-two scaling (gradual underflow preserved).  <= 1 ulp from libm; the
 * HIP kernels carry an independent copy of the same recipe.
 * ---------------------------------------------------------------------------------- */
static inline float f32_pow2(int e) { /* 2^e, -126 <= e <= 127 */
    union { uint32_t u; float f; } v;
    v.u = (uint32_t)(e + 127) << 23;
    return v.f;
}
float sgo_expf(float x) {
    if (x != x) return x;
    if (x > 88.72284f) return INFINITY;
    if (x < -103.972084f) return 0.0f;
    float nf = rintf(x * 0x1.715476p+0f);
    float r = fmaf(nf, -0x1.62e400p-1f, x);
    r = fmaf(nf, -0x1.7f7d1cp-20f, r);
    float p = 0x1.a01a02p-13f;
    p = fmaf(p, r, 0x1.6c16c2p-10f);
    p = fmaf(p, r, 0x1.111112p-7f);
    p = fmaf(p, r, 0x1.555556p-5f);
    p = fmaf(p, r, 0x1.555556p-3f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int n = (int)nf;
    int n1 = n >> 1, n2 = n - n1;
    return (p * f32_pow2(n1)) * f32_pow2(n2);
}
static inline double f64_pow2(int e) { /* 2^e, -1022 <= e <= 1023 */
    union { uint64_t u; double f; } v;
    v.u = (uint64_t)(e + 1023) << 52;
    return v.f;
}
double sgo_exp(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.1332191019412) return 0.0;
    double nf = rint(x * 0x1.71547652b82fep+0);
    double r = fma(nf, -0x1.62e42fee00000p-1, x);
    r = fma(nf, -0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = fma(p, r, 0x1.1eed8eff8d898p-29);
    p = fma(p, r, 0x1.ae64567f544e4p-26);
    p = fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = fma(p, r, 0x1.71de3a556c734p-19);
    p = fma(p, r, 0x1.a01a01a01a01ap-16);
    p = fma(p, r, 0x1.a01a01a01a01ap-13);
    p = fma(p, r, 0x1.6c16c16c16c17p-10);
    p = fma(p, r, 0x1.1111111111111p-7);
    p = fma(p, r, 0x1.5555555555555p-5);
    p = fma(p, r, 0x1.5555555555555p-3);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int n = (int)nf;
    int n1 = n >> 1, n2 = n - n1;
    return (p * f64_pow2(n1)) * f64_pow2(n2);
}

/* ------------------------------------------------------------------------------------
 * local field and energy
 * ---------------------------------------------------------------------------------- */
/* torch.dot(couplings[i], spins) in fp32 (ising_model.py:183): products J*(+-1) are exact;
 * the sum is formed in double and rounded once to fp32, i.e. the correctly rounded fp32 dot
 * (exactly the reference's value whenever its own fp32 summation is exact, e.g. integer J). */
/* When the caller asserts that J is integer valued with |row sums| < 2^24 (sgo_set_exact_f32),
 * fp32 accumulation in any order is exact and equals the double-accumulated value: the loop
 * below then keeps 16 independent fp32 lanes (what MKL's sdot does for the reference). */
static int g_exact_f32 = 0;
void sgo_set_exact_f32(int on) { g_exact_f32 = on; }

static inline float row_dot_f32(int n, const float *J, int64_t ld, const int32_t *rowptr,
                                const int32_t *colidx, const float *val, const int8_t *s, int i) {
    if (J) {
        const float *row = J + (int64_t)i * ld;
        if (g_exact_f32) {
            float lane[16] = {0};
            int j = 0;
            for (; j + 16 <= n; j += 16)
                for (int q = 0; q < 16; ++q) lane[q] += row[j + q] * (float)s[j + q];
            float acc = 0.0f;
            for (int q = 0; q < 16; ++q) acc += lane[q];
            for (; j < n; ++j) acc += row[j] * (float)s[j];
            return acc;
        }
        /* Real-valued J: fp32 products (exact), summed in double in the CANONICAL ORDER the HIP
         * kernels use for every launch geometry (sweep_dense_impl.h): super-chunks of 1024 elements;
         * lane l of 64 adds its sixteen products -- chunk j = 0..3 of the super-chunk, elements
         * 4l .. 4l+3 of each, in that order -- starting from +0, the 64 lane sums are folded by an
         * adjacent-pairs tree, and the super-chunk sums are added in order.  The double sum is
         * rounded to fp32 once (torch.dot returns fp32). */
        double total = 0.0;
        for (int c0 = 0; c0 < n; c0 += 1024) {
            double lane[64];
            for (int l = 0; l < 64; ++l) {
                double p = 0.0;
                for (int q = 0; q < 16; ++q) {
                    int j = c0 + 256 * (q >> 2) + 4 * l + (q & 3);
                    if (j < n) p += (double)(row[j] * (float)s[j]);
                }
                lane[l] = p;
            }
            for (int stride = 1; stride < 64; stride *= 2)
                for (int l = 0; l < 64; l += 2 * stride) lane[l] = lane[l] + lane[l + stride];
            total = (c0 == 0) ? lane[0] : total + lane[0];
        }
        return (float)total;
    }
    /* CSR, same idea with the entries of the row in storage order: entry e belongs to lane e % 64
     * of virtual wave (e / 64) % 8; a virtual lane adds its entries in storage order, each
     * virtual wave folds its 64 lanes by the adjacent-pairs tree, and the 8 wave sums are added
     * in order (sweep_csr.hip: 1, 2, 4 or 8 real waves per replica all reproduce it). */
    double lanes[8][64];
    int used = 0;
    const int32_t beg = rowptr[i], len = rowptr[i + 1] - rowptr[i];
    for (int e = 0; e < len; ++e) {
        int v = (e >> 6) & 7, l = e & 63;
        double t = (double)(val[beg + e] * (float)s[colidx[beg + e]]);
        if ((e >> 9) == 0) {
            lanes[v][l] = t;
            if (v + 1 > used) used = v + 1;
        } else {
            lanes[v][l] += t;
        }
    }
    double total = 0.0;
    for (int v = 0; v < used; ++v) {
        int filled = len - 64 * v; /* lanes of the first pass that hold an entry */
        for (int l = (filled > 64 ? 64 : filled); l < 64; ++l) lanes[v][l] = 0.0;
        for (int stride = 1; stride < 64; stride *= 2)
            for (int l = 0; l < 64; l += 2 * stride) lanes[v][l] = lanes[v][l] + lanes[v][l + stride];
        total = (v == 0) ? lanes[v][0] : total + lanes[v][0];
    }
    return (float)total;
}
static inline float diag_elem(int n, const float *J, int64_t ld, const int32_t *rowptr,
                              const int32_t *colidx, const float *val, int i) {
    (void)n;
    if (J) return J[(int64_t)i * ld + i];
    float d = 0.0f;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if (colidx[k] == i) d += val[k];
    return d;
}

/* IsingModel.get_local_field, ising_model.py:176-185:
 *   coupling_field = torch.dot(couplings[i], spins).item()   (fp32 -> python float)
 *   return coupling_field + external_fields[i].item()         (double add)            */
double sgo_local_field(int n, const float *J, int64_t ld, const int32_t *rowptr,
                       const int32_t *colidx, const float *val, const float *h,
                       const int8_t *s, int i) {
    return (double)row_dot_f32(n, J, ld, rowptr, colidx, val, s, i) + (double)h[i];
}

/* IsingModel.compute_energy, ising_model.py:149-174:
 *   interaction = -0.5 * torch.dot(spins, torch.mv(couplings, spins)).item()
 *   field       = -torch.dot(external_fields, spins).item()
 *   total       = interaction + field                        (python doubles)         */
double sgo_energy(int n, const float *J, int64_t ld, const int32_t *rowptr,
                  const int32_t *colidx, const float *val, const float *h, const int8_t *s) {
    double acc = 0.0, hs = 0.0;
    for (int i = 0; i < n; ++i) {
        float mv_i = row_dot_f32(n, J, ld, rowptr, colidx, val, s, i); /* torch.mv row, fp32 */
        acc += (double)mv_i * (double)s[i];
        hs += (double)h[i] * (double)s[i];
    }
    double interaction = -0.5 * (double)(float)acc;
    double field = -(double)(float)hs;
    return interaction + field;
}

/* ------------------------------------------------------------------------------------
 * single Metropolis update
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int accepted;
    int used_u; /* the reference draws torch.rand(1) only on this path */
    double dE;  /* proposed delta energy (before the accept test) */
} upd_t;

static inline upd_t metropolis_core(int n, const float *J, int64_t ld, const int32_t *rowptr,
                                    const int32_t *colidx, const float *val, const float *h,
                                    int8_t *s, int site, double T, float u, int arith, int rule) {
    upd_t o = {0, 0, 0.0};
    if (rule != SGO_RULE_METROPOLIS) {
        /* SpinDynamics._glauber_update, spin_dynamics.py:154-171:
         *   prob_up = 1.0 / (1.0 + torch.exp(torch.tensor(-2.0 * local_field / T)))
         * SpinDynamics._heat_bath_update, :173-191:
         *   beta = 1.0 / T;  prob_up = 1.0 / (1.0 + torch.exp(torch.tensor(-2.0 * beta * field)))
         * (double argument rounded to fp32, fp32 exp / add / divide), then
         *   new_spin = 1 if torch.rand(1).item() < prob_up else -1;  flip iff it differs.   */
        double field = sgo_local_field(n, J, ld, rowptr, colidx, val, h, s, site);
        float x = (rule == SGO_RULE_GLAUBER) ? (float)(-2.0 * field / T)
                                             : (float)((-2.0 * (1.0 / T)) * field);
        float prob_up = 1.0f / (1.0f + sgo_expf(x));
        int new_spin = (u < prob_up) ? 1 : -1;
        o.used_u = 1;
        o.dE = 2.0 * (double)s[site] * field; /* the true energy change of a flip */
        if (new_spin != (int)s[site]) {
            s[site] = (int8_t)new_spin;
            o.accepted = 1;
        }
        return o;
    }
    if (arith == SGO_ARITH_F64) {
        /* SpinDynamics._metropolis_update, spin_dynamics.py:131-152 */
        double field = sgo_local_field(n, J, ld, rowptr, colidx, val, h, s, site); /* :134 */
        double dE = 2.0 * (double)s[site] * field;                                  /* :135 */
        o.dE = dE;
        if (dE <= 0.0) { /* :138-142 */
            s[site] = (int8_t)-s[site]; /* IsingModel.flip_spin, ising_model.py:144 */
            o.accepted = 1;
        } else {
            /* :145  torch.exp(torch.tensor(-delta_energy / T)): double divide, round to
             * fp32, fp32 exp */
            float p = sgo_expf((float)(-dE / T));
            o.used_u = 1;
            if ((double)u < (double)p) { /* :146  torch.rand(1).item() < acceptance_prob */
                s[site] = (int8_t)-s[site];
                o.accepted = 1;
            }
        }
    } else {
        /* CUDAKernelManager._metropolis_update_fallback, cuda_kernels.py:381-396 (fp32
         * tensors):  local_field = h[i] + sum(J[i]*s) - J[i,i]*s[i]
         *            delta_energy = 2.0 * s[i] * local_field
         *            accept if dE <= 0 or rand < exp(-dE / T)                          */
        float dot = row_dot_f32(n, J, ld, rowptr, colidx, val, s, site);
        float si = (float)s[site];
        float field = (h[site] + dot) - diag_elem(n, J, ld, rowptr, colidx, val, site) * si;
        float dE = (2.0f * si) * field;
        o.dE = (double)dE;
        if (dE <= 0.0f) {
            s[site] = (int8_t)-s[site];
            o.accepted = 1;
        } else {
            float p = sgo_expf(-dE / (float)T);
            o.used_u = 1;
            if (u < p) {
                s[site] = (int8_t)-s[site];
                o.accepted = 1;
            }
        }
    }
    return o;
}

/* ------------------------------------------------------------------------------------
 * Wolff cluster move: SpinDynamics._wolff_cluster_dense, spin_dynamics.py:210-255 (for CSR input the
 * same rule over the row's stored entries in storage order).  One uniform per candidate bond, taken
 * from `u` at *cursor (recorded stream) or from the production Philox stream (domain 3).
 * Returns the cluster size, or -1 when the recorded stream runs out.
 * ---------------------------------------------------------------------------------- */
static int wolff_move(int n, const float *J, int64_t ld, const int32_t *rowptr, const int32_t *colidx,
                      const float *val, int8_t *s, int start, double T, const float *u, int64_t *cursor,
                      int64_t u_cap, uint64_t seed, uint32_t replica, uint32_t sweep, uint32_t t,
                      int32_t *queue, uint8_t *in_cluster) {
    memset(in_cluster, 0, (size_t)n);
    int head = 0, tail = 1;
    int64_t drawn = 0;
    queue[0] = start;
    in_cluster[start] = 1;
    while (head < tail) { /* :222  while queue: current_site = queue.pop(0) */
        int cur = queue[head++];
        int8_t sc = s[cur];
        int32_t beg = J ? 0 : rowptr[cur], end = J ? n : rowptr[cur + 1];
        for (int32_t k = beg; k < end; ++k) { /* :227  for neighbor in range(n_spins) */
            int j = J ? k : colidx[k];
            float c = J ? J[(int64_t)cur * ld + j] : val[k];
            if (j == cur || in_cluster[j]) continue;                    /* :228 */
            if (!(c < 0.0f && sc == s[j])) continue;                    /* :235 */
            /* :237  prob_add = 1.0 - torch.exp(torch.tensor(2.0 * coupling / T))  (fp32 tensor) */
            float p_add = 1.0f - sgo_expf((float)(2.0 * (double)c / T));
            float uu;
            if (u) {
                if (*cursor >= u_cap) return -1;
                uu = u[(*cursor)++];
            } else {
                uint32_t o[4];
                stream_block(seed, (uint32_t)(drawn >> 2), sweep, replica, 3u | (t << 2), o);
                uu = word_to_u(o[drawn & 3]);
            }
            ++drawn;
            if (uu < p_add) { /* :239 */
                in_cluster[j] = 1;
                queue[tail++] = j;
            }
        }
    }
    for (int i = 0; i < tail; ++i) s[queue[i]] = (int8_t)-s[queue[i]]; /* :244-245 */
    return tail;
}

int sgo_metropolis_update(int n, const float *J, int64_t ld, const int32_t *rowptr,
                          const int32_t *colidx, const float *val, const float *h, int8_t *s,
                          int site, double T, float u, int arith, int rule, double *dE_out) {
    upd_t o = metropolis_core(n, J, ld, rowptr, colidx, val, h, s, site, T, u, arith, rule);
    /* spin_dynamics.py:142,149,152,167,188 (heat bath returns -delta_energy) */
    if (dE_out) *dE_out = o.accepted ? (rule == SGO_RULE_HEAT_BATH ? -o.dE : o.dE) : 0.0;
    return o.accepted;
}

/* ------------------------------------------------------------------------------------
 * sweeps driver
 * ---------------------------------------------------------------------------------- */
int sgo_sweeps(int n, const float *J, int64_t ld, const int32_t *rowptr, const int32_t *colidx,
               const float *val, const float *h, int R, int8_t *spins, double *energy,
               const double *temps, int64_t t_sweep_stride, int64_t t_replica_stride,
               int n_sweeps, int site_mode, int arith, int rule, uint64_t seed, uint32_t sweep0,
               uint32_t replica0, const int32_t *replay_site, const float *replay_u,
               int u_compact, int64_t u_capacity, double *energy_trace, int64_t *n_accepted,
               double *best_energy, int8_t *best_spins, uint8_t *accept_trace,
               double *dE_trace, int recompute_energy, int n_threads) {
    if (n <= 0 || R <= 0 || n_sweeps < 0 || !spins || !energy || !temps || !h) return -1;
    if (!J && !(rowptr && colidx && val)) return -1;
    if (site_mode == SGO_SITE_REPLAY && !replay_site) return -1;
    if ((site_mode != SGO_SITE_RANDOM) && !replay_u && rule != SGO_RULE_WOLFF) return -1;
    volatile int err = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int r = 0; r < R; ++r) {
        int8_t *s = spins + (int64_t)r * n;
        double E = energy[r];
        int64_t acc = 0;
        int64_t per_rep = (int64_t)n_sweeps * n;
        int64_t ucur = 0;
        int32_t *wq = 0;
        uint8_t *wc = 0;
        if (rule == SGO_RULE_WOLFF) {
            wq = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
            wc = (uint8_t *)malloc((size_t)n);
            if (!wq || !wc) { err = 3; free(wq); free(wc); continue; }
        }
        const float *ru = replay_u ? replay_u + (u_compact ? r * u_capacity : r * per_rep) : 0;
        for (int k = 0; k < n_sweeps; ++k) {
            double T = temps[k * t_sweep_stride + r * t_replica_stride];
            uint32_t blk[4] = {0, 0, 0, 0};
            for (int t = 0; t < n; ++t) { /* SpinDynamics.sweep, spin_dynamics.py:82-85 */
                int64_t idx = (int64_t)k * n + t;
                int site;
                float u = 0.0f;
                if (site_mode == SGO_SITE_RANDOM) {
                    if ((t & 1) == 0)
                        stream_block(seed, (uint32_t)t >> 1, sweep0 + (uint32_t)k,
                                     replica0 + (uint32_t)r, 0, blk);
                    site = (int)word_to_site(blk[2 * (t & 1)], (uint32_t)n);
                    u = word_to_u(blk[2 * (t & 1) + 1]);
                } else {
                    site = (site_mode == SGO_SITE_SEQUENTIAL) ? t
                                                              : replay_site[r * per_rep + idx];
                    if (!u_compact) u = ru[idx];
                }
                if (site < 0 || site >= n) { err = 1; site = 0; }
                if (rule == SGO_RULE_WOLFF) {
                    /* always accepted; delta = compute_energy() after - before (:248-249);
                     * the uniforms: a flat recorded stream per replica (u_compact layout) | Philox */
                    double e_before = (dE_trace || !recompute_energy)
                                          ? sgo_energy(n, J, ld, rowptr, colidx, val, h, s) : 0.0;
                    int size = wolff_move(n, J, ld, rowptr, colidx, val, s, site, T,
                                          (replay_u && u_compact) ? ru : 0, &ucur, u_capacity, seed,
                                          replica0 + (uint32_t)r, sweep0 + (uint32_t)k, (uint32_t)t, wq, wc);
                    if (size < 0) { err = 2; size = 0; }
                    acc += size;
                    if (dE_trace || !recompute_energy) {
                        double d = sgo_energy(n, J, ld, rowptr, colidx, val, h, s) - e_before;
                        E += d;
                        if (dE_trace) dE_trace[r * per_rep + idx] = d;
                    }
                    if (accept_trace) accept_trace[r * per_rep + idx] = 1;
                    continue;
                }
                upd_t o;
                if (site_mode != SGO_SITE_RANDOM && u_compact) {
                    /* consume the recorded uniform only where the reference draws one */
                    float cand = (ucur < u_capacity) ? ru[ucur] : 2.0f;
                    o = metropolis_core(n, J, ld, rowptr, colidx, val, h, s, site, T, cand, arith, rule);
                    if (o.used_u) {
                        if (ucur >= u_capacity) err = 2;
                        ++ucur;
                    }
                } else {
                    o = metropolis_core(n, J, ld, rowptr, colidx, val, h, s, site, T, u, arith, rule);
                }
                if (o.accepted) { E += o.dE; ++acc; }
                if (accept_trace) accept_trace[r * per_rep + idx] = (uint8_t)o.accepted;
                if (dE_trace)
                    dE_trace[r * per_rep + idx] =
                        o.accepted ? (rule == SGO_RULE_HEAT_BATH ? -o.dE : o.dE) : 0.0;
            }
            if (recompute_energy) /* spin_dynamics.py:87 */
                E = sgo_energy(n, J, ld, rowptr, colidx, val, h, s);
            if (energy_trace) energy_trace[(int64_t)k * R + r] = E;
            if (best_energy && E < best_energy[r]) { /* gpu_annealer.py:151-153 */
                best_energy[r] = E;
                if (best_spins) memcpy(best_spins + (int64_t)r * n, s, (size_t)n);
            }
        }
        energy[r] = E;
        if (n_accepted) n_accepted[r] += acc;
        free(wq);
        free(wc);
    }
    return err ? -2 : 0;
}

/* ------------------------------------------------------------------------------------
 * TSP-structured couplings written out as CSR (problems/routing.py:250-328 in the convention of
 * the build's encoders.tsp_csr): spin (c, p) = c * n + p has the 4 (n - 1) neighbours
 *   (c, p')      p' != p            -A/2     one position per city
 *   (c', p)      c' != c            -B/2     one city per position
 *   (c', p - 1)  c' != c      -d[c'][c]/4    c' precedes c in the tour
 *   (c', p + 1)  c' != c      -d[c][c']/4    c' follows c
 * columns ascending within a row.  The checker for the engine's implicit form (sga_set_tsp): the
 * CSR functions above run on what this writes.
 * ---------------------------------------------------------------------------------- */
int sgo_tsp_to_csr(int n, const float *d, float A, float B, int64_t *rowptr, int32_t *colidx, float *val) {
    if (n < 3 || !d || !rowptr || !colidx || !val) return -1;
    const int deg = 4 * (n - 1);
    for (int c = 0; c < n; ++c)
        for (int p = 0; p < n; ++p) {
            const int64_t row = (int64_t)c * n + p;
            int64_t at = row * deg;
            rowptr[row] = at;
            int trip[3] = {(p + n - 1) % n, p, (p + 1) % n}, kind[3] = {0, 1, 2};
            for (int i = 0; i < 3; ++i) /* sort the three positions, carrying what each one is */
                for (int j = i + 1; j < 3; ++j)
                    if (trip[j] < trip[i]) {
                        int t = trip[i]; trip[i] = trip[j]; trip[j] = t;
                        t = kind[i]; kind[i] = kind[j]; kind[j] = t;
                    }
            for (int c2 = 0; c2 < n; ++c2) {
                if (c2 == c) {
                    for (int p2 = 0; p2 < n; ++p2)
                        if (p2 != p) {
                            colidx[at] = c * n + p2;
                            val[at++] = -(A / 2.0f);
                        }
                } else {
                    for (int i = 0; i < 3; ++i) {
                        colidx[at] = c2 * n + trip[i];
                        val[at++] = kind[i] == 1 ? -(B / 2.0f)
                                  : kind[i] == 0 ? -(d[(int64_t)c2 * n + c] / 4.0f)
                                                 : -(d[(int64_t)c * n + c2] / 4.0f);
                    }
                }
            }
        }
    rowptr[(int64_t)n * n] = (int64_t)n * n * deg;
    return 0;
}

/* ------------------------------------------------------------------------------------
 * replica exchange
 * ---------------------------------------------------------------------------------- */
int sgo_pt_exchange_round(int R, const double *slot_temps, const double *rep_energy,
                          int32_t *slot_to_rep, int start, const double *u, uint64_t seed,
                          uint32_t round, uint32_t ladder, int64_t *attempts, int64_t *accepts) {
    if (start < 0) { /* np.random.randint(0, 2), parallel_tempering.py:217 */
        uint32_t o[4];
        stream_block(seed, 0xFFFFFFFFu, round, ladder, 1, o);
        start = (int)(o[0] & 1u);
    }
    int n_acc = 0, k = 0;
    for (int i = start; i < R - 1; i += 2, ++k) { /* :219-220 */
        int j = i + 1;
        /* _attempt_single_exchange, :234-258 */
        double beta_i = 1.0 / slot_temps[i], beta_j = 1.0 / slot_temps[j];
        double Ei = rep_energy[slot_to_rep[i]], Ej = rep_energy[slot_to_rep[j]];
        double x = (beta_j - beta_i) * (Ej - Ei);
        double prob = (x >= 0.0) ? 1.0 : sgo_exp(x); /* min(1.0, np.exp(x)), :246 */
        double uu;
        if (u) {
            uu = u[k];
        } else {
            uint32_t o[4];
            stream_block(seed, (uint32_t)i, round, ladder, 1, o);
            uu = ((double)(o[0] >> 5) * 67108864.0 + (double)(o[1] >> 6)) * 0x1.0p-53;
        }
        if (attempts) attempts[i] += 1;
        if (uu < prob) { /* :252 */
            int32_t tmp = slot_to_rep[i];
            slot_to_rep[i] = slot_to_rep[j];
            slot_to_rep[j] = tmp;
            if (accepts) accepts[i] += 1;
            ++n_acc;
        }
    }
    return n_acc;
}

int sgo_pt_exchange_pairs(int R, const double *slot_temps, const double *rep_energy,
                          int32_t *slot_to_rep, const int32_t *pairs, const double *u, int count,
                          uint64_t seed, uint32_t round, int64_t *attempts, int64_t *accepts) {
    /* ParallelTempering._all_pairs_exchange, CPU branch (parallel_tempering.py:222-232): the
     * caller has applied the `rand() < 0.1` gate; each listed pair runs
     * _attempt_single_exchange (:234-258) and sees the swaps before it. */
    int n_acc = 0;
    for (int k = 0; k < count; ++k) {
        int i = pairs[2 * k], j = pairs[2 * k + 1];
        if (i < 0 || j < 0 || i >= R || j >= R) return -1;
        double beta_i = 1.0 / slot_temps[i], beta_j = 1.0 / slot_temps[j];
        double Ei = rep_energy[slot_to_rep[i]], Ej = rep_energy[slot_to_rep[j]];
        double x = (beta_j - beta_i) * (Ej - Ei);
        double prob = (x >= 0.0) ? 1.0 : sgo_exp(x);
        double uu;
        if (u) {
            uu = u[k];
        } else {
            uint32_t o[4];
            stream_block(seed, 0x40000000u | (uint32_t)k, round, 0, 1, o);
            uu = ((double)(o[0] >> 5) * 67108864.0 + (double)(o[1] >> 6)) * 0x1.0p-53;
        }
        int lo = i < j ? i : j; /* pair_idx = min(i, j), :249 */
        if (attempts) attempts[lo] += 1;
        if (uu < prob) {
            int32_t tmp = slot_to_rep[i];
            slot_to_rep[i] = slot_to_rep[j];
            slot_to_rep[j] = tmp;
            if (accepts) accepts[lo] += 1;
            ++n_acc;
        }
    }
    return n_acc;
}

int sgo_pt_exchange_operator(int R, int n, int8_t *spins, float *energies, const float *temps,
                             const float *u) {
    /* CUDAKernelManager._parallel_tempering_fallback, cuda_kernels.py:415-443 */
    int n_acc = 0;
    int8_t *tmp = (int8_t *)malloc((size_t)n);
    if (!tmp) return -1;
    for (int i = 0; i < R - 1; ++i) {
        float beta1 = 1.0f / temps[i], beta2 = 1.0f / temps[i + 1];
        float db = beta2 - beta1;
        float de = energies[i] - energies[i + 1];
        float prob = sgo_expf(db * de);
        if (u[i] < prob) {
            memcpy(tmp, spins + (int64_t)i * n, (size_t)n);
            memcpy(spins + (int64_t)i * n, spins + (int64_t)(i + 1) * n, (size_t)n);
            memcpy(spins + (int64_t)(i + 1) * n, tmp, (size_t)n);
            float e = energies[i];
            energies[i] = energies[i + 1];
            energies[i + 1] = e;
            ++n_acc;
        }
    }
    free(tmp);
    return n_acc;
}
