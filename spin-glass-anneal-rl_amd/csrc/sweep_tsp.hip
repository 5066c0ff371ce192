// sweep_tsp.hip -- single-spin sweeps on TSP-structured couplings that are never stored
// (BASELINE configs[4]: examples/tsp_example.py at 1000 cities = 10^6 spins; as CSR the couplings
// are 32 GB and every update streams a 32 KB row from HBM).
//
// The Ising problem is the one problems/routing.py:250-328 compiles (spin (c, p) = city c at tour
// position p, index c * n + p), in the convention of encoders.tsp_csr:
//     J[(c,p),(c,p')] = -A/2  (p' != p)          one position per city
//     J[(c,p),(c',p)] = -B/2  (c' != c)          one city per position
//     J[(c,p),(c',p-1)] = -d[c'][c]/4,  J[(c,p),(c',p+1)] = -d[c][c']/4   (c' != c, positions mod n)
// so row (c,p) . s = -A/2 (S_city[c] - s) - B/2 (S_pos[p] - s)
//                    + sum_c' ( -d[c'][c]/4 s(c',p-1) - d[c][c']/4 s(c',p+1) ),
// with S_city / S_pos the spin sums of a city's row / a position's column.  Per replica the kernel
// keeps the spins as bits in LDS, POSITION-major (a position's column of n cities is contiguous:
// 32 words at n = 1000), and the 2n sums as integers; an update reads two 4-KB rows of the scaled
// distance table (4 MB at n = 1000: cache resident) instead of a 32-KB coupling row from HBM.
//
// Arithmetic: every product is exact in fp32 (a distance times +-1, a penalty times an integer
// below 2^11).  The engine checks at set time that the fp64 sum of a row is exact as well (all
// values within 53 binary places of each other, carries included -- true for distances of any
// realistic range); then row . s rounded to fp32 is THE correctly rounded value, which is also what
// the CSR kernels and the oracle produce: the chains are bit-identical to the stored-coupling forms.
// (Integer instances accumulate in fp32.)  One workgroup per replica, W waves; lane l of wave w
// holds cities 4 (l + 64 (w + W k)) ... + 3 of pass k.
#include "sweep_common.h"

namespace sga {

constexpr int TSP_MAX_WAVES = 8;
#ifndef TSP_ROWS_AHEAD
#define TSP_ROWS_AHEAD 2  // distance rows requested this many updates before their reduction
#endif

// spins of the replica: int8 city-major in HBM -> bits position-major in LDS, plus the 2n sums
__device__ inline void tsp_load_spins(const int8_t *src, unsigned int *bits, int *sums, int n, int cw,
                                      int first, int step) {
    for (int i = first; i < n * cw; i += step) bits[i] = 0u;
    for (int i = first; i < 2 * n; i += step) sums[i] = 0;
    __syncthreads();
    const int N = n * n;
    for (int idx = first; idx < N; idx += step) {
        const int c = idx / n, p = idx - c * n;
        const int s = src[idx];
        if (s < 0) atomicOr(&bits[p * cw + (c >> 5)], 1u << (c & 31));
        atomicAdd(&sums[c], s);       // S_city[c]
        atomicAdd(&sums[n + p], s);   // S_pos[p]
    }
    __syncthreads();
}

__device__ inline void tsp_store_spins(const unsigned int *bits, int8_t *dst, int n, int cw, int sstride,
                                       int first, int step) {
    const int N = n * n;
    for (int idx = first; idx < sstride; idx += step) {
        int8_t v = 0;
        if (idx < N) {
            const int c = idx / n, p = idx - c * n;
            v = ((bits[p * cw + (c >> 5)] >> (c & 31)) & 1u) ? (int8_t)-1 : (int8_t)1;
        }
        dst[idx] = v;
    }
}

// this lane's four cities of pass k against the columns `pm` (previous position) and `pn` (next);
// FIRST: the accumulator starts from its first term (0 + x is an instruction of its own).  (One chain of
// adds: two accumulators measured 3-4 % slower, profiles/r02_experiments.md 14.)
template <bool FIRST = false, typename acc_t>
__device__ __forceinline__ void tsp_accumulate(acc_t &acc, const float4 &xprev, const float4 &xnext,
                                               const unsigned int *bits, int cw, int pm, int pn, int city0) {
    const unsigned int wp = bits[pm * cw + (city0 >> 5)] >> (city0 & 31);
    const unsigned int wn = bits[pn * cw + (city0 >> 5)] >> (city0 & 31);
    auto signed_val = [](float v, unsigned int word, int q) -> float {
        return __int_as_float(__float_as_int(v) ^ (int)(((word >> q) & 1u) << 31));
    };
    if constexpr (FIRST) acc = (acc_t)signed_val(xprev.x, wp, 0);
    else acc += (acc_t)signed_val(xprev.x, wp, 0);
    acc += (acc_t)signed_val(xprev.y, wp, 1);
    acc += (acc_t)signed_val(xprev.z, wp, 2);
    acc += (acc_t)signed_val(xprev.w, wp, 3);
    acc += (acc_t)signed_val(xnext.x, wn, 0);
    acc += (acc_t)signed_val(xnext.y, wn, 1);
    acc += (acc_t)signed_val(xnext.z, wn, 2);
    acc += (acc_t)signed_val(xnext.w, wn, 3);
}

// NP = passes per wave over a distance row (256 cities per wave and pass); F64 = fp64 accumulation
template <int NP, bool F64, bool LEAN>
__global__ void __launch_bounds__(64 * TSP_MAX_WAVES) sweep_tsp_kernel(const SweepArgs a, const TspArgs t) {
    using acc_t = typename std::conditional<F64, double, float>::type;
    const int rule = a.rule;
    const int arith = LEAN ? SGA_ARITH_F64 : a.arith;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = t.n_cities, cw = t.npad >> 5;
    unsigned int *bits = reinterpret_cast<unsigned int *>(smem);      // [n positions][cw words]
    int *sums = reinterpret_cast<int *>(smem + 4ll * n * cw);         // S_city[n], S_pos[n]
    double *part = reinterpret_cast<double *>(smem + ((4ll * n * cw + 8ll * n + 7) & ~7ll));  // [2][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = (int)(blockDim.x >> 6);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    tsp_load_spins(a.spins + (long long)r * a.sstride, bits, sums, n, cw, tid, (int)blockDim.x);

    double E = a.energy[r], bestE = a.best_energy[r], T = 1.0;
    unsigned long long nacc = 0;
    int pp = 0;
    const unsigned int lane16 = (unsigned int)(lane + 64 * w) * 16u;  // byte offset of this lane's float4 in pass 0
    const unsigned int pass_bytes = (unsigned int)(64 * W) * 16u;

    struct Slot {
        float4 prev[NP], next[NP];
        int site, c, p;
        float u, h;
    };
    auto request = [&](Slot &sl) {
        // c = site / n by the multiply-shift the host verified for every site
        sl.c = (int)(((unsigned long long)(unsigned int)sl.site * t.div_magic) >> 32);
        sl.p = sl.site - sl.c * n;
        const unsigned char *rp = reinterpret_cast<const unsigned char *>(t.nd4t) + (unsigned long long)sl.c * t.row_bytes;
        const unsigned char *rn = reinterpret_cast<const unsigned char *>(t.nd4) + (unsigned long long)sl.c * t.row_bytes;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            unsigned int off = lane16 + (unsigned int)k * pass_bytes;
            asm volatile("" : "+v"(off));
            sl.prev[k] = *reinterpret_cast<const float4 *>(rp + off);
            sl.next[k] = *reinterpret_cast<const float4 *>(rn + off);
        }
        sl.h = a.h[sl.site];
    };

    PairSource<LEAN> rng;
    UpdatePair pairP{0, 0, 2.0f, 2.0f};
    int kP = 0, tP = 0;  // producer cursor
    const int N = a.n;
    auto produce = [&](Slot &sl) {
        const bool second = tP & 1;
        if (!second) pairP = rng.get(a, r, kP, tP >> 1, kP < a.n_sweeps, lane);  // past the end: site 0
        const int sA = pairP.sA, sB = pairP.sB;
        const float uA = pairP.uA, uB = pairP.uB;
        sl.site = second ? sB : sA;
        sl.u = second ? uB : uA;
        request(sl);
        if (++tP == N) {
            tP = 0;
            ++kP;
        }
    };

    auto step = [&](const Slot &sl, long long upd) {
        const int c = sl.c, p = sl.p;
        const int pm = p == 0 ? n - 1 : p - 1, pn = p == n - 1 ? 0 : p + 1;
        acc_t acc = 0;
        tsp_accumulate<true>(acc, sl.prev[0], sl.next[0], bits, cw, pm, pn, 4 * (lane + 64 * w));
#pragma unroll
        for (int k = 1; k < NP; ++k)
            tsp_accumulate(acc, sl.prev[k], sl.next[k], bits, cw, pm, pn, 4 * (lane + 64 * (w + W * k)));
        acc_t dist = wave_sum(acc);
        // what the flip needs, read before any wave can have applied THIS update's flip
        const int si = ((bits[p * cw + (c >> 5)] >> (c & 31)) & 1u) ? -1 : 1;
        const int sc = sums[c], sp = sums[n + p];
        if (W > 1) {
            acc_t *slot = reinterpret_cast<acc_t *>(part + pp * TSP_MAX_WAVES);
            if (lane == 0) slot[w] = dist;
            __syncthreads();
            acc_t s = slot[0];
            for (int i = 1; i < W; ++i) s += slot[i];
            dist = s;
            pp ^= 1;
        }
        // exact products, exact sum (set-time check), one rounding to fp32 (core/ising_model.py:183)
        const double row = (double)t.a2 * (double)(sc - si) + (double)t.b2 * (double)(sp - si) + (double)dist;
        const float dot = (float)row;
        double dE;
        const bool flip = metropolis_accept(rule, arith, dot, si, sl.h, 0.0f, T, sl.u, dE);
        if (flip) {
            E += dE;
            ++nacc;
            if (lane == 0) {  // the same absolute values from every wave: idempotent
                if (si > 0) atomicOr(&bits[p * cw + (c >> 5)], 1u << (c & 31));
                else atomicAnd(&bits[p * cw + (c >> 5)], ~(1u << (c & 31)));
                sums[c] = sc - 2 * si;
                sums[n + p] = sp - 2 * si;
            }
        }
        if constexpr (!LEAN) {
            if (tid == 0) {
                if (a.accept_trace) a.accept_trace[(long long)r * a.replay_stride + upd] = flip ? 1 : 0;
                if (a.dE_trace)
                    a.dE_trace[(long long)r * a.replay_stride + upd] =
                        flip ? (rule == SGA_RULE_HEAT_BATH ? -dE : dE) : 0.0;
            }
        }
    };

    // The rows of the next TSP_ROWS_AHEAD updates are in flight while update g is reduced (a ring of
    // slots, every index a constant after unrolling); a sweep is whole groups of ring-size updates
    // plus a remainder that shifts the ring back into phase -- the per-sweep work stays outside the
    // unrolled body.
    constexpr int NB = TSP_ROWS_AHEAD + 1;
    Slot ring[NB];
#pragma unroll
    for (int j = 0; j + 1 < NB; ++j) produce(ring[j]);
    for (int k = 0; k < a.n_sweeps; ++k) {
        T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        const long long g0 = (long long)k * N;
        int tt = 0;
        for (; tt + NB <= N; tt += NB) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                produce(ring[(j + NB - 1) % NB]);
                step(ring[j], g0 + tt + j);
            }
        }
        for (; tt < N; ++tt) {
            produce(ring[NB - 1]);
            step(ring[0], g0 + tt);
#pragma unroll
            for (int j = 0; j + 1 < NB; ++j) ring[j] = ring[j + 1];
        }
        if (tid == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE && !a.no_best) {  // annealing/gpu_annealer.py:151-153
            bestE = E;
            __syncthreads();  // every wave has applied the last flip
            tsp_store_spins(bits, a.best_spins + (long long)r * a.sstride, n, cw, a.sstride, tid, (int)blockDim.x);
            __syncthreads();
        }
    }
    __syncthreads();
    tsp_store_spins(bits, a.spins + (long long)r * a.sstride, n, cw, a.sstride, tid, (int)blockDim.x);
    if (tid == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

// ---------------------------------------------------------------------------------------
// SEVERAL UPDATES PER STEP (production arguments): wave w of the workgroup works on update W m + w of the
// sweep -- the whole row, NPF = npad / 256 passes -- against the state as it stands before the first of the W.
// An update at (c, p) reads the columns p - 1 and p + 1 and the sums of city c and position p, so a flip at
// (c, p) matters to a later update (c', p') of the step only if c' == c or p' is p - 1, p or p + 1 (mod n):
// four of n^2 sites in n -- at 1000 cities eight updates are independent in 97 % of the steps.  The waves
// publish (flips?, c, p, dE), meet at ONE barrier, and everyone checks the accepted updates against the later
// ones by index arithmetic; no hit: every wave applies its own flip (distinct cities and positions, so the
// sums do not collide), a second barrier, next step.  A hit: the step is replayed one update at a time (wave q
// decides again from the rows it still holds, a barrier per update).  The energy is added in chain order by
// every wave from the published dE.  Same chain bit for bit as the one-update form above (row sums are exact
// in any order: set-time check), which remains the form for traces / replayed streams.
// One update per wave instead of one per workgroup: at 1000 cities (one 136 KB workgroup per CU) the chip ran
// two waves per CU on one dependent chain; now eight chains per CU.
// ---------------------------------------------------------------------------------------
constexpr int TSP_PAR_SLOT_INTS = 8;  // flip, c, p, pad, dE (2 ints), pad, pad
template <int NPF, bool F64>
__global__ void __launch_bounds__(64 * TSP_MAX_WAVES) sweep_tsp_par_kernel(const SweepArgs a, const TspArgs t) {
    using acc_t = typename std::conditional<F64, double, float>::type;
    const int rule = a.rule;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = t.n_cities, cw = t.npad >> 5;
    unsigned int *bits = reinterpret_cast<unsigned int *>(smem);      // [n positions][cw words]
    int *sums = reinterpret_cast<int *>(smem + 4ll * n * cw);         // S_city[n], S_pos[n]
    int *slots = reinterpret_cast<int *>(smem + ((4ll * n * cw + 8ll * n + 15) & ~15ll) + 2 * TSP_MAX_WAVES * sizeof(double));
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = (int)(blockDim.x >> 6);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    tsp_load_spins(a.spins + (long long)r * a.sstride, bits, sums, n, cw, tid, (int)blockDim.x);

    double E = a.energy[r], bestE = a.best_energy[r], T = 1.0;
    unsigned long long nacc = 0;
    const unsigned int lane16 = (unsigned int)lane * 16u;  // byte offset of this lane's float4 in pass 0
    const int N = a.n;
    const int steps = (N + W - 1) / W;  // steps per sweep

    struct Slot {
        float4 prev[NPF], next[NPF];
        int site, c, p, live;
        float u, h;
    };
    // this wave's Philox window: blocks 64 q .. 64 q + 63 of sweep k (lane l: block 64 q + l)
    uint32_t vsa = 0, vua = 0, vsb = 0, vub = 0;
    int win_k = -1, win_q = -1;
    auto produce = [&](Slot &sl, int k, int m) {  // update W m + w of sweep k (past the end: dead, site 0)
        const int tu = W * m + w;
        sl.live = (k < a.n_sweeps && tu < N) ? 1 : 0;
        sl.site = 0;
        sl.u = 2.0f;
        if (sl.live) {  // wave-uniform
            const int b = tu >> 1;
            if (win_k != k || win_q != (b >> 6)) {
                win_k = k, win_q = b >> 6;
                const u32x4 x = philox4x32_10((uint32_t)(64 * win_q + lane), a.sweep0 + (uint32_t)k, a.replica0 + (uint32_t)r,
                                              DOMAIN_SWEEP, a.seed_lo, a.seed_hi);
                vsa = word_to_site(x.x, (uint32_t)N), vua = x.y, vsb = word_to_site(x.z, (uint32_t)N), vub = x.w;
            }
            const int l = b & 63;
            sl.site = __builtin_amdgcn_readlane((int)((tu & 1) ? vsb : vsa), l);
            sl.u = word_to_u((uint32_t)__builtin_amdgcn_readlane((int)((tu & 1) ? vub : vua), l));
        }
        sl.c = (int)(((unsigned long long)(unsigned int)sl.site * t.div_magic) >> 32);
        sl.p = sl.site - sl.c * n;
        const unsigned char *rp = reinterpret_cast<const unsigned char *>(t.nd4t) + (unsigned long long)sl.c * t.row_bytes;
        const unsigned char *rn = reinterpret_cast<const unsigned char *>(t.nd4) + (unsigned long long)sl.c * t.row_bytes;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            unsigned int off = lane16 + (unsigned int)q * 1024u;
            asm volatile("" : "+v"(off));
            sl.prev[q] = *reinterpret_cast<const float4 *>(rp + off);
            sl.next[q] = *reinterpret_cast<const float4 *>(rn + off);
        }
        sl.h = a.h[sl.site];
    };
    // this wave's decision against the state as it stands
    auto decide = [&](const Slot &sl, int &si, double &dE) -> bool {
        const int c = sl.c, p = sl.p;
        const int pm = p == 0 ? n - 1 : p - 1, pn = p == n - 1 ? 0 : p + 1;
        acc_t acc = 0;
        tsp_accumulate<true>(acc, sl.prev[0], sl.next[0], bits, cw, pm, pn, 4 * lane);
#pragma unroll
        for (int q = 1; q < NPF; ++q) tsp_accumulate(acc, sl.prev[q], sl.next[q], bits, cw, pm, pn, 4 * (lane + 64 * q));
        const acc_t dist = wave_sum(acc);
        si = ((bits[p * cw + (c >> 5)] >> (c & 31)) & 1u) ? -1 : 1;
        const int sc = sums[c], sp = sums[n + p];
        // exact products, exact sum (set-time check), one rounding to fp32 (core/ising_model.py:183)
        const double row = (double)t.a2 * (double)(sc - si) + (double)t.b2 * (double)(sp - si) + (double)dist;
        const bool flip = metropolis_accept(rule, SGA_ARITH_F64, (float)row, si, sl.h, 0.0f, T, sl.u, dE);
        return sl.live != 0 && flip;
    };
    auto apply = [&](const Slot &sl, int si) {  // this wave's own flip
        if (lane == 0) {
            atomicXor(&bits[sl.p * cw + (sl.c >> 5)], 1u << (sl.c & 31));
            sums[sl.c] -= 2 * si;
            sums[n + sl.p] -= 2 * si;
        }
    };
    int turn = 0;
    auto step = [&](const Slot &sl) {
        int si;
        double dE;
        const bool flip = decide(sl, si, dE);
        int *mine = slots + (turn * TSP_MAX_WAVES + w) * TSP_PAR_SLOT_INTS;
        if (lane == 0) {
            const long long db = __double_as_longlong(dE);
            *reinterpret_cast<int4 *>(mine) = make_int4(flip ? 1 : 0, sl.c, sl.p, sl.live);
            *reinterpret_cast<int2 *>(mine + 4) = make_int2((int)(unsigned int)db, (int)(db >> 32));
        }
        __syncthreads();  // (A) every wave has decided against the old state and published
        int4 q0 = make_int4(0, -1, -1, 0);
        int2 q1 = make_int2(0, 0);
        if (lane < W) {
            const int *theirs = slots + (turn * TSP_MAX_WAVES + lane) * TSP_PAR_SLOT_INTS;
            q0 = *reinterpret_cast<const int4 *>(theirs);
            q1 = *reinterpret_cast<const int2 *>(theirs + 4);
        }
        turn ^= 1;
        const unsigned long long acc = ballot64(q0.x != 0);
        // does an accepted update matter to a later one of the step?
        unsigned long long hit = 0;
        unsigned long long rest = acc;
        while (rest) {  // (few: wave-uniform loop over the accepted updates)
            const int q = (int)__builtin_ctzll(rest);
            rest &= rest - 1;
            const int cq = __builtin_amdgcn_readlane(q0.y, q), pq = __builtin_amdgcn_readlane(q0.z, q);
            int dp = q0.z - pq;
            dp = dp < 0 ? -dp : dp;
            const bool near = dp <= 1 || dp == n - 1;  // p' in {p - 1, p, p + 1} (mod n)
            hit |= ballot64(lane > q && lane < W && q0.w != 0 && (q0.y == cq || near));
        }
        if (hit == 0ull) {
            if (flip) apply(sl, si);
            unsigned long long order = acc;
            while (order) {  // chain order
                const int q = (int)__builtin_ctzll(order);
                order &= order - 1;
                const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane(q1.x, q), hi = (unsigned int)__builtin_amdgcn_readlane(q1.y, q);
                E += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                ++nacc;
            }
            __syncthreads();  // (B) the new state is visible
            return;
        }
        // one update at a time: wave q decides again from the rows it holds, against what the earlier ones left
        for (int q = 0; q < W; ++q) {
            int *slot = slots + (turn * TSP_MAX_WAVES) * TSP_PAR_SLOT_INTS;  // one shared record per pass
            if (w == q) {
                asm volatile("" ::: "memory");  // (the state is re-read: other waves' flips behind barriers)
                int si2;
                double dE2;
                const bool flip2 = decide(sl, si2, dE2);
                if (flip2) apply(sl, si2);
                if (lane == 0) {
                    const long long db = __double_as_longlong(dE2);
                    slot[0] = flip2 ? 1 : 0;
                    *reinterpret_cast<int2 *>(slot + 4) = make_int2((int)(unsigned int)db, (int)(db >> 32));
                }
            }
            __syncthreads();
            if (slot[0]) {
                const int2 d = *reinterpret_cast<const int2 *>(slot + 4);
                E += __longlong_as_double((long long)(((unsigned long long)(unsigned int)d.y << 32) | (unsigned int)d.x));
                ++nacc;
            }
            __syncthreads();  // (the record is free again)
        }
    };
    auto later = [&](int k, int m, int ahead, int &ko, int &mo) {
        mo = m + ahead;
        ko = k;
        while (mo >= steps) {
            mo -= steps;
            ++ko;
        }
    };
    // the rows of the next step are in flight while a step is reduced (eight waves: 128 VGPRs each -- two
    // steps of rows at four passes would not fit)
#ifndef TSP_PAR_RING
#define TSP_PAR_RING 2
#endif
    constexpr int NB = TSP_PAR_RING;
    Slot ring[NB];
#pragma unroll
    for (int j = 0; j + 1 < NB; ++j) {
        int kj, mj;
        later(0, 0, j, kj, mj);
        produce(ring[j], kj, mj);
    }
    for (int k = 0; k < a.n_sweeps; ++k) {
        T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        int m = 0;
        for (; m + NB <= steps; m += NB) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                int k2, m2;
                later(k, m + j, NB - 1, k2, m2);
                produce(ring[(j + NB - 1) % NB], k2, m2);
                step(ring[j]);
            }
        }
        for (; m < steps; ++m) {
            int k2, m2;
            later(k, m, NB - 1, k2, m2);
            produce(ring[NB - 1], k2, m2);
            step(ring[0]);
#pragma unroll
            for (int j = 0; j + 1 < NB; ++j) ring[j] = ring[j + 1];
        }
        if (tid == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE && !a.no_best) {  // annealing/gpu_annealer.py:151-153
            bestE = E;
            tsp_store_spins(bits, a.best_spins + (long long)r * a.sstride, n, cw, a.sstride, tid, (int)blockDim.x);
            __syncthreads();
        }
    }
    __syncthreads();
    tsp_store_spins(bits, a.spins + (long long)r * a.sstride, n, cw, a.sstride, tid, (int)blockDim.x);
    if (tid == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

size_t tsp_lds_bytes(int n_cities, int npad) {
    // bits | sums | [2][8] partial sums (one-update form) | [2][8] decision records (several-updates form)
    return (size_t)(((4ll * n_cities * (npad >> 5) + 8ll * n_cities + 15) & ~15ll) + 2 * TSP_MAX_WAVES * sizeof(double) +
                    2 * TSP_MAX_WAVES * TSP_PAR_SLOT_INTS * sizeof(int));
}

template <int NP>
static hipError_t launch_tsp_np(const SweepArgs &a, const TspArgs &t, int waves, hipStream_t st) {
    const bool lean = sweep_args_are_lean(a);
    void (*kern)(const SweepArgs, const TspArgs) =
        t.f64 ? (lean ? sweep_tsp_kernel<NP, true, true> : sweep_tsp_kernel<NP, true, false>)
              : (lean ? sweep_tsp_kernel<NP, false, true> : sweep_tsp_kernel<NP, false, false>);
    const size_t lds = tsp_lds_bytes(t.n_cities, t.npad);
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.R), dim3(64 * waves), lds, st, a, t);
    return hipGetLastError();
}

// updates per step of the several-updates form for n cities (0: the one-update form): a pair of updates is
// independent with probability 1 - 4/n
int tsp_parallel_updates(int n_cities, int npad, int forced /* engine option "tsp_updates_per_step", -1: auto */) {
    int want = n_cities >= 256 ? 8 : n_cities >= 64 ? 4 : n_cities >= 24 ? 2 : 0;
    if (forced >= 0) want = forced;  // A/B switch, parity tests
    if (want < 2 || npad > 1024) return 0;  // (builds for rows of up to 4 passes per wave)
    return std::min(want, TSP_MAX_WAVES);
}

template <int NPF>
static hipError_t launch_tsp_par(const SweepArgs &a, const TspArgs &t, int waves, hipStream_t st) {
    void (*kern)(const SweepArgs, const TspArgs) = t.f64 ? sweep_tsp_par_kernel<NPF, true> : sweep_tsp_par_kernel<NPF, false>;
    const size_t lds = tsp_lds_bytes(t.n_cities, t.npad);
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.R), dim3(64 * waves), lds, st, a, t);
    note_sweep_kernel("sweep_tsp_par_kernel<%d passes, %s> x %d updates per step", NPF, t.f64 ? "f64" : "f32", waves);
    return hipGetLastError();
}

hipError_t launch_sweep_tsp(const SweepArgs &a, const TspArgs &t, int waves, int passes, hipStream_t st) {
    if (waves < 1 || waves > TSP_MAX_WAVES || 256 * waves * passes != t.npad) return hipErrorInvalidValue;
    const int par = sweep_args_are_lean(a) ? tsp_parallel_updates(t.n_cities, t.npad, a.tsp_parallel) : 0;
    if (par >= 2) {
        switch (t.npad / 256) {
            case 1: return launch_tsp_par<1>(a, t, par, st);
            case 2: return launch_tsp_par<2>(a, t, par, st);
            case 3: return launch_tsp_par<3>(a, t, par, st);
            case 4: return launch_tsp_par<4>(a, t, par, st);
            default: break;
        }
    }
    note_sweep_kernel("sweep_tsp_kernel<%d passes> x %d wave(s)", passes, waves);
    switch (passes) {
        case 1: return launch_tsp_np<1>(a, t, waves, st);
        case 2: return launch_tsp_np<2>(a, t, waves, st);
        case 4: return launch_tsp_np<4>(a, t, waves, st);
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------
// Full energy on the same structure: -1/2 fp32(sum_i mv_i s_i) - fp32(h . s) with mv_i the fp32
// row sum (core/ising_model.py:149-174).  Grid (replica, slice of cities); 4 waves, one row each.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) energy_tsp_kernel(const EnergyArgs a, const TspArgs t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = t.n_cities, cw = t.npad >> 5;
    unsigned int *bits = reinterpret_cast<unsigned int *>(smem);
    int *sums = reinterpret_cast<int *>(smem + 4ll * n * cw);
    double *red = reinterpret_cast<double *>(smem + ((4ll * n * cw + 8ll * n + 7) & ~7ll));
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    tsp_load_spins(a.spins + (long long)r * a.sstride, bits, sums, n, cw, tid, (int)blockDim.x);
    const int per = (n + a.slices - 1) / a.slices;
    const int c0 = blockIdx.y * per, c1 = min(n, c0 + per);
    double e_acc = 0.0, h_acc = 0.0;
    for (int c = c0; c < c1; ++c) {
        const float *rp = t.nd4t + (long long)c * (t.row_bytes >> 2);
        const float *rn = t.nd4 + (long long)c * (t.row_bytes >> 2);
        for (int p = w; p < n; p += 4) {
            const int pm = p == 0 ? n - 1 : p - 1, pn = p == n - 1 ? 0 : p + 1;
            double acc = 0.0;
            for (int city0 = 4 * lane; city0 < t.npad; city0 += 256)
                tsp_accumulate(acc, *reinterpret_cast<const float4 *>(rp + city0),
                               *reinterpret_cast<const float4 *>(rn + city0), bits, cw, pm, pn, city0);
            const double dist = wave_sum(acc);
            const int si = ((bits[p * cw + (c >> 5)] >> (c & 31)) & 1u) ? -1 : 1;
            const double row = (double)t.a2 * (double)(sums[c] - si) + (double)t.b2 * (double)(sums[n + p] - si) + dist;
            const float mv = (float)row;
            e_acc += (double)mv * (double)si;
            h_acc += (double)a.h[c * n + p] * (double)si;
        }
    }
    if (lane == 0) {
        red[w] = e_acc;
        red[4 + w] = h_acc;
    }
    __syncthreads();
    if (tid == 0) {
        const double e = (red[0] + red[1]) + (red[2] + red[3]);
        const double hs = (red[4] + red[5]) + (red[6] + red[7]);
        if (a.slices <= 1) {
            a.energy[r] = -0.5 * (double)(float)e + (-(double)(float)hs);
        } else {
            double *o = a.partial + ((long long)r * a.slices + blockIdx.y) * 2;
            o[0] = e;
            o[1] = hs;
        }
    }
}

hipError_t launch_energy_tsp(const EnergyArgs &a, const TspArgs &t, hipStream_t st) {
    const size_t lds = tsp_lds_bytes(t.n_cities, t.npad);
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(energy_tsp_kernel), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(energy_tsp_kernel, dim3(a.R, a.slices), dim3(256), lds, st, a, t);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Local fields of single sites (IsingModel.get_local_field, core/ising_model.py:176-185) on the
// structure, spins read from HBM: one wave per requested site.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) fields_tsp_kernel(const TspArgs t, const int8_t *spins, const float *h,
                                                        const int32_t *sites, double *out) {
    const int n = t.n_cities, lane = threadIdx.x;
    const int site = sites[blockIdx.x];
    const int c = site / n, p = site - c * n;
    const int pm = p == 0 ? n - 1 : p - 1, pn = p == n - 1 ? 0 : p + 1;
    const float *rp = t.nd4t + (long long)c * (t.row_bytes >> 2);
    const float *rn = t.nd4 + (long long)c * (t.row_bytes >> 2);
    double acc = 0.0;
    int sc = 0, sp = 0;
    for (int q = lane; q < n; q += 64) {
        acc += (double)(rp[q] * (float)spins[q * n + pm]);
        acc += (double)(rn[q] * (float)spins[q * n + pn]);
        sc += spins[c * n + q];
        sp += spins[q * n + p];
    }
    const double dist = wave_sum(acc);
    sc = wave_sum(sc);
    sp = wave_sum(sp);
    const int si = spins[site];
    const double row = (double)t.a2 * (double)(sc - si) + (double)t.b2 * (double)(sp - si) + dist;
    if (lane == 0) out[blockIdx.x] = (double)(float)row + (double)h[site];
}

hipError_t launch_fields_tsp(const TspArgs &t, const int8_t *spins, const float *h, const int32_t *sites,
                             int count, double *out, hipStream_t st) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(fields_tsp_kernel, dim3(count), dim3(64), 0, st, t, spins, h, sites, out);
    return hipGetLastError();
}

// scaled distance tables: nd4[c][c'] = -d[c][c']/4 and its transpose, rows zero padded to npad,
// diagonal forced to zero (a city is no neighbour of itself)
__global__ void tsp_tables_kernel(const float *d, long long ldd, int n, int npad, float *nd4, float *nd4t) {
    const long long total = (long long)n * npad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i / npad), q = (int)(i - (long long)c * npad);
        float a = 0.0f, b = 0.0f;
        if (q < n && q != c) {
            a = -d[(long long)c * ldd + q] / 4.0f;   // exact scaling
            b = -d[(long long)q * ldd + c] / 4.0f;
        }
        nd4[i] = a;
        nd4t[i] = b;
    }
}
hipError_t launch_tsp_tables(const float *d, long long ldd, int n, int npad, float *nd4, float *nd4t,
                             hipStream_t st) {
    hipLaunchKernelGGL(tsp_tables_kernel, dim3(1024), dim3(256), 0, st, d, ldd, n, npad, nd4, nd4t);
    return hipGetLastError();
}

}  // namespace sga
