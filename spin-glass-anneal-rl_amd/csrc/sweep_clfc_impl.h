// sweep_clfc_impl.h -- the cached-local-field sweep, CHAIN-WAVE form (round 4): the production build of
// sweep_clf_impl.h's chain -- Philox sites, Metropolis with the reference's fp64 / fp32-exp arithmetic, no
// per-update records -- with the serial part of an accepted proposal cut from ~260 issued instructions, two
// workgroup barriers and five dependent LDS round trips to ~60 instructions of ONE wave and one LDS round trip.
//
// Replaces the same reference code as sweep_clf_impl.h: SpinDynamics.sweep / _metropolis_update
// (core/spin_dynamics.py:73-94,131-152) evaluated the way the reference's incremental mode does
// (core/energy_computer.py:166-173,262-265: dE from a maintained field, the field updated on a flip).
// The chain is the one-row-per-proposal chain bit for bit (same sites, uniforms, accept rule; all quantities
// exact integers): the tests run it against the same oracle runs as the other two sweep forms.
//
// Why the old form costs 1.1 - 1.7 us per accept (profiles/r03_experiments.md 1, r04_experiments.md 3): every
// accept makes all waves of the workgroup meet twice, rewrite 10 KB of fields in LDS and read 512 candidate
// fields back before the next decision can be taken.  But the next decision only needs the fields of the FEW
// candidates that can accept at all:
//
//   * a window of 512 consecutive updates is generated and evaluated by all four waves (as before); a candidate
//     stays in the game only if it could accept after up to K more flips have moved its field -- u < p(k - K D),
//     D = 2 scale max|J| the most one flip moves k = s_i F_i, p the (monotone, checked per sweep) accept table --
//     or if its site is proposed twice in the window (a flip at its own site turns k into -k).  In the glassy
//     regime annealing lives in that is 2 - 15 % of the candidates; they are compacted, in chain order, into a
//     list of at most 64 (a longer list cuts the window short);
//   * wave 0 -- the chain wave -- takes the list, one candidate per lane, with field and spin in REGISTERS, and
//     walks the chain alone: ballot, first accepting lane, its site i; the 64 couplings J[i][site of lane] come
//     by a gather from row i (the row of the next accepting candidate is gathered one round ahead); fields and
//     same-site spins are corrected in registers, decisions retaken -- no barrier, no field array traffic;
//   * the other three waves drain a queue of the accepted (site, old spin) pairs in LDS and apply the rows to
//     the resident field array behind the chain wave's back (nobody reads it before the window's last barrier);
//   * after K accepts, or when no candidate accepts any more, the window ends: one barrier, everything before
//     the stop is decided, the next window starts right behind it.
//
// Byte model (its own, as for sweep_clf_impl.h): B = acceptance rate x row bytes per attempt.
#pragma once
#include "sweep_clf_impl.h"

namespace sga {

constexpr int CLFC_WAVES = 4;                // wave 0 walks the chain, waves 1 .. 3 keep the fields up to date
constexpr int CLFC_FIELD_WAVES = CLFC_WAVES - 1;
constexpr int CLFC_PASS = 2;                 // chunks of a row a field wave requests together (one item)
constexpr int CLFC_Q_NONE = -1, CLFC_Q_END = -2;  // accept queue: nothing announced yet | the window is over
constexpr int CLFC_SPAN = 128 * CLFC_WAVES;  // updates generated and evaluated per window: two per lane
constexpr int CLFC_LIST = 64;                // candidates of a window the chain wave follows: one per lane
constexpr int CLFC_QUEUE = 128;              // accept queue of a window: <= 64 entries (flip budget K) + the end mark
constexpr int CLFC_PRE = 8;                  // rows whose couplings to the listed sites are gathered ahead, per window
constexpr int CLFC_CTRL_INTS = 16;
enum { CLFC_CNT0 = 0, CLFC_CUT = 4, CLFC_NEXT = 5, CLFC_NONMONO = 8, CLFC_E_LO = 10, CLFC_E_HI = 11 };

// LDS of one replica: fields | spin bits | accept table  (clf_*_offset of sweep_clf_impl.h), then
// seen / twice bitmaps over the sites | candidate list | accept queue | control words
__host__ __device__ constexpr long long clfc_extra_offset(long long ldf, int fbytes, int sstride, int table_m) {
    return clf_table_offset(ldf, fbytes, sstride) + 4ll * ((table_m + 4) & ~3);
}
inline size_t clfc_lds_bytes(long long ldf, int fbytes, int sstride, int table_m) {
    return (size_t)clfc_extra_offset(ldf, fbytes, sstride, table_m) + 2 * (size_t)(sstride / 8) + 16 * CLFC_LIST +
           4 * CLFC_QUEUE + 4 * CLFC_CTRL_INTS + 16;
}

template <typename JT, typename FT>
__global__ void __launch_bounds__(64 * CLFC_WAVES, 4) sweep_clfc_kernel(const SweepArgs a) {  // (4 waves per SIMD: four replicas per CU)
    constexpr int EPL = 16 / (int)sizeof(JT), EPC = 64 * EPL;  // elements per lane / per 1-KiB chunk
    constexpr int FB = (int)sizeof(FT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FT *F = reinterpret_cast<FT *>(smem);
    unsigned int *bits = reinterpret_cast<unsigned int *>(smem + clf_bits_offset(a.ldf, FB));
    float *ptab = reinterpret_cast<float *>(smem + clf_table_offset(a.ldf, FB, a.sstride));
    unsigned int *seen = reinterpret_cast<unsigned int *>(smem + clfc_extra_offset(a.ldf, FB, a.sstride, a.table_m));
    unsigned int *twice = seen + a.sstride / 32;
    int4 *list = reinterpret_cast<int4 *>(twice + a.sstride / 32);
    int *queue = reinterpret_cast<int *>(list + CLFC_LIST);
    int *ctrl = queue + CLFC_QUEUE;
    // The accept queue is the one place where waves of the workgroup talk WITHOUT a barrier between them: workgroup-
    // scope atomics (plain ds_read / ds_write behind the right waits; a `volatile` pointer here compiled to flat
    // loads with system coherence -- 6 barriers' worth of them per window, profiles/r04_experiments.md 3).
    auto ld_acq = [](const int *q) { return __hip_atomic_load(q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto st_rel = [](int *q, int v) { __hip_atomic_store(q, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (a launch over a subset of the replicas -- per-replica routing, sga_kernels.h -- names them in rep_list)
    const int r = a.rep_list ? __builtin_amdgcn_readfirstlane(a.rep_list[blockIdx.x]) : (int)blockIdx.x, n = a.n;
    const int sc = a.field_scale;
    const double inv_sc = 1.0 / (double)sc;  // 1 | 0.5: exact

    {   // resident state -> LDS
        const int4 *src = reinterpret_cast<const int4 *>(reinterpret_cast<const FT *>(a.fields) + (long long)r * a.ldf);
        int4 *dst = reinterpret_cast<int4 *>(F);
        for (int i = tid; i < (int)(a.ldf * FB / 16); i += blockDim.x) dst[i] = src[i];
        spins_to_bits(a.spins + (long long)r * a.sstride, bits, a.sstride, tid, blockDim.x);
        for (int i = tid; i < 2 * (a.sstride / 32); i += blockDim.x) seen[i] = 0u;
        if (tid < CLFC_CTRL_INTS) ctrl[tid] = 0;
        if (tid < CLFC_QUEUE) queue[tid] = CLFC_Q_NONE;
    }
    __syncthreads();

    const JT *Jbase = reinterpret_cast<const JT *>(a.J);
    const int n_chunks = (int)((a.ldj + EPC - 1) / EPC);
    using vec_t = typename std::conditional<sizeof(JT) == 4, float4, int4>::type;
    double E = a.energy[r], bestE = a.best_energy[r];  // (E: the chain wave's; published once per sweep)
    unsigned long long nacc = 0;
    // how far one flip can move k = s_i F_i of another site, and how many flips a window may hold
    const int K = min(max(a.clf_flips, 1), 64);
    const long long KD = (long long)K * 2ll * sc * (long long)a.clf_jmax;
    const unsigned long long lt = (1ull << lane) - 1ull;  // lanes below this one
    long long ksum = 0;  // the chain wave's sum of k = s_i F_i over the sweep's accepted moves
#ifdef CLFC_PROFILE
    // Where a replica's time goes (profiles/r04_clfc_profile.py; 100 MHz ticks of s_memrealtime, summed over the launch,
    // returned through the first rows of energy_trace): 0 windows, 1 accepts, 2 accepts whose couplings were not
    // gathered ahead, 3 listed candidates, 4 windows cut at 64 candidates, 5 windows ended by the flip budget,
    // 6 ticks generating + compacting, 7 ticks in the chain loop, 8 ticks waiting for the field waves, 9 ticks
    // waiting for couplings gathered on demand, 10 ticks of the whole sweep loop, 11 ticks filling the table
    long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long prof_t0 = wall_clock64();
#define CLFC_TICK() wall_clock64()
#define CLFC_ADD(i, v) prof[i] += (v)
#else
#define CLFC_TICK() 0ll
#define CLFC_ADD(i, v) (void)0
#endif

    for (int k = 0; k < a.n_sweeps; ++k) {
        const double T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        [[maybe_unused]] const long long tick_table = CLFC_TICK();
        // exp(float32(-dE/T)) of the moves dE = 2 q / scale, q <= table_m (entry 0 = 1: every downhill move)
        for (int q = tid; q <= a.table_m; q += blockDim.x) ptab[q] = expf_det((float)(-((double)(2 * q) * inv_sc) / T));
        if (tid == 0) ctrl[CLFC_NONMONO] = 0;
        __syncthreads();
        {   // The candidate filter below reads "u >= p(k') implies rejected for every k >= k'": true while the table
            // does not increase anywhere.  expf_det is monotone up to single ulps between ADJACENT floats
            // (profiles/r04_experiments.md 3), which table entries -- 2 / (scale T) apart -- never are at any
            // temperature an annealer uses; checked all the same, per sweep, and the filter is off where it fails.
            bool up = false;
            for (int q = tid; q < a.table_m; q += blockDim.x) up = up || (ptab[q] < ptab[q + 1]);
            if (__ballot(up) != 0ull && lane == 0) ctrl[CLFC_NONMONO] = 1;
        }
        __syncthreads();
        const bool filter_ok = ctrl[CLFC_NONMONO] == 0 && (2.0 * inv_sc / T) >= 1.0e-4 && a.clf_jmax > 0 && KD < (1ll << 24);

        CLFC_ADD(11, CLFC_TICK() - tick_table);
        int t0 = 0;  // first undecided update of the sweep (workgroup-uniform)
        while (t0 < n) {
            [[maybe_unused]] const long long tick_gen = CLFC_TICK();
            // ---- the window's candidates: updates [t0, base_t + 512), two per lane, against the current state ----
            const int base_t = (t0 >> 1) << 1;  // (a Philox block serves updates 2 b and 2 b + 1)
            const int pA = w * 128 + 2 * lane;   // position in the window
            const int tA = base_t + pA, tB = tA + 1;
            const bool liveA = tA >= t0 && tA < n, liveB = tB < n;
            uint32_t key_lo = a.seed_lo, key_hi = a.seed_hi;
            asm volatile("" : "+s"(key_lo), "+s"(key_hi));
            const u32x4 x = philox4x32_10((uint32_t)(tA >> 1), a.sweep0 + (uint32_t)k, a.replica0 + (uint32_t)r,
                                          DOMAIN_SWEEP, key_lo, key_hi);
            const int sA = (int)word_to_site(x.x, (uint32_t)n), sB = (int)word_to_site(x.z, (uint32_t)n);
            const float uA = word_to_u(x.y), uB = word_to_u(x.w);
            const unsigned int bitA = 1u << (sA & 31), bitB = 1u << (sB & 31);
            // sites proposed twice in the window: first everybody marks its site ...
            unsigned int oldA = 0u, oldB = 0u;
            if (liveA) oldA = atomicOr(&seen[sA >> 5], bitA);
            if (liveB) oldB = atomicOr(&seen[sB >> 5], bitB);
            const int fa = (int)F[sA], fb = (int)F[sB];
            const unsigned int wa = bits[sA >> 5], wb = bits[sB >> 5];
            const int siA = (wa & bitA) ? -1 : 1, siB = (wb & bitB) ? -1 : 1;
            const int kA = siA * fa, kB = siB * fb;
            __syncthreads();  // (1)
            // ... then whoever found its site marked already says so for both
            if (liveA && (oldA & bitA)) atomicOr(&twice[sA >> 5], bitA);
            if (liveB && (oldB & bitB)) atomicOr(&twice[sB >> 5], bitB);
            if (tid == 0) ctrl[CLFC_CUT] = min(n, base_t + CLFC_SPAN);
            __syncthreads();  // (2)
            // In the game: a candidate that may accept once up to K flips have moved its field (one flip moves k by at
            // most D), and every candidate whose site occurs twice (a flip AT its site turns k into -k).  Everything
            // else is rejected for good whatever the K flips of this window are.
            bool actA = liveA, actB = liveB;
            if (filter_ok) {
                const long long qa = (long long)kA - KD, qb = (long long)kB - KD;
                const float pa = ptab[(int)min(max(qa, 0ll), (long long)a.table_m)];
                const float pb = ptab[(int)min(max(qb, 0ll), (long long)a.table_m)];
                actA = liveA && (uA < pa || (twice[sA >> 5] & bitA));
                actB = liveB && (uB < pb || (twice[sB >> 5] & bitB));
            }
            const unsigned long long mA = __ballot(actA), mB = __ballot(actB);
            if (lane == 0) ctrl[CLFC_CNT0 + w] = __builtin_popcountll(mA) + __builtin_popcountll(mB);
            __syncthreads();  // (3)
            int off = 0, total = 0;
#pragma unroll
            for (int v = 0; v < CLFC_WAVES; ++v) {
                const int c = ctrl[CLFC_CNT0 + v];
                off += v < w ? c : 0;
                total += c;
            }
            // compaction in chain order; a 65th candidate in the game ends the window right before itself
            const int rA = off + __builtin_popcountll(mA & lt) + __builtin_popcountll(mB & lt), rB = rA + (actA ? 1 : 0);
            if (actA && rA < CLFC_LIST) list[rA] = make_int4(sA, __float_as_int(uA), fa, (siA < 0 ? 1 : 0) | (pA << 1));
            if (actB && rB < CLFC_LIST) list[rB] = make_int4(sB, __float_as_int(uB), fb, (siB < 0 ? 1 : 0) | ((pA + 1) << 1));
            if (actA && rA == CLFC_LIST) ctrl[CLFC_CUT] = tA;
            if (actB && rB == CLFC_LIST) ctrl[CLFC_CUT] = tB;
            if (liveA) atomicAnd(&seen[sA >> 5], ~bitA);  // the bitmaps are clean again for the next window
            if (liveB) atomicAnd(&seen[sB >> 5], ~bitB);
            if (liveA && (oldA & bitA)) atomicAnd(&twice[sA >> 5], ~bitA);
            if (liveB && (oldB & bitB)) atomicAnd(&twice[sB >> 5], ~bitB);
            __syncthreads();  // (4)
            const int cut = ctrl[CLFC_CUT];
            const int n_list = min(total, CLFC_LIST);
            [[maybe_unused]] const long long tick_chain = CLFC_TICK();
            CLFC_ADD(0, 1), CLFC_ADD(3, n_list), CLFC_ADD(4, total > CLFC_LIST ? 1 : 0), CLFC_ADD(6, tick_chain - tick_gen);

            if (w == 0) {
                // ---- the chain wave: one candidate per lane, state in registers ----
                // (it is the replica's critical path: it issues ahead of the waves it shares its SIMD with -- field
                //  waves polling their queues, other replicas' waves)
                __builtin_amdgcn_s_setprio(3);
                const int4 me = list[min(lane, CLFC_LIST - 1)];
                const int site = lane < n_list ? me.x : 0;  // (lanes beyond the list: stale entries, kept in range)
                const float u = __int_as_float(me.y);
                int f = me.z;
                bool down = (me.w & 1) != 0;  // the spin at the site: -1
                const int pos = me.w >> 1;
                bool alive = lane < n_list;
                int flips = 0, next_t0 = cut;
                int kk = 0;
                // who flips against the state as it stands (the accept rule on k = s_i F_i of every live candidate)
                auto evaluate = [&]() -> unsigned long long {
                    kk = down ? -f : f;
                    const float p = ptab[min(max(kk, 0), a.table_m)];
                    bool acc = alive && u < p;
                    const bool beyond = alive && kk > a.table_m;
                    if (__ballot(beyond)) {  // rare: large uphill moves (p == 0 past -104, sweep_common.h)
                        const double d = (double)(2 * kk) * inv_sc;
                        if (beyond) acc = !(d > T * 104.0) && u < expf_det((float)(-d / T));
                    }
                    return __ballot(acc);
                };
                unsigned long long m = evaluate();
                // The couplings of EVERY candidate accepting right now -- the accepts of this window are mostly among
                // them -- to all listed sites are gathered together, up to CLFC_PRE rows, one memory latency for all
                // (0.25 us while J sits in the Infinity Cache, ~2 us from HBM); a candidate that starts accepting later
                // is gathered on demand.
                const unsigned long long m_first = m;
                int jA[CLFC_PRE];
                {
                    unsigned long long rest = m_first;
#pragma unroll
                    for (int q = 0; q < CLFC_PRE; ++q) {
                        jA[q] = 0;
                        if (rest) {  // wave-uniform
                            const int lp = (int)__builtin_ctzll(rest);
                            rest &= rest - 1ull;
                            const JT *rowp = Jbase + (long long)__builtin_amdgcn_readlane(site, lp) * a.ldj;
                            jA[q] = (int)rowp[site];
                        }
                    }
                }
#ifdef CLFC_PROFILE
                {   // (profile build: how long the gathered-together couplings take)
                    const long long tg = wall_clock64();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (m_first) prof[9] += wall_clock64() - tg, prof[2] += 1;
                }
#endif
                while (m != 0ull) {  // (m == 0: everything up to the cut is rejected)
                    const int l0 = (int)__builtin_ctzll(m);
                    const int site_i = __builtin_amdgcn_readlane(site, l0);
                    const int k_i = __builtin_amdgcn_readlane(kk, l0);
                    const bool down_i = ((__ballot(down) >> l0) & 1ull) != 0ull;
                    ksum += (long long)k_i;  // dE = 2 k / scale: an integer sum per sweep, converted once
                    ++nacc;
                    // for the field waves: (site, old spin) -- ONE word, no count, no ordering to keep
                    if (lane == 0)
                        __hip_atomic_store(&queue[flips], site_i | (down_i ? (int)0x40000000 : 0), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                    // couplings of the accepted site to the candidates' sites
                    int j;
                    const int rank = __builtin_popcountll(m_first & ((1ull << l0) - 1ull));
                    if (((m_first >> l0) & 1ull) && rank < CLFC_PRE) {
                        j = jA[0];
#pragma unroll
                        for (int q = 1; q < CLFC_PRE; ++q) j = rank == q ? jA[q] : j;
                    } else {
                        [[maybe_unused]] const long long tick_g = CLFC_TICK();
                        const JT *row = Jbase + (long long)site_i * a.ldj;
                        j = (int)row[site];
#ifdef CLFC_PROFILE
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        prof[9] += wall_clock64() - tick_g;
                        prof[2] += 1;
#endif
                    }
                    CLFC_ADD(1, 1);
                    // F_q -= 2 scale J_qi s_i (zero diagonal: the own field stays); |j| < 2^23, the factor is +-2 | +-4
                    f -= __mul24(down_i ? -2 * sc : 2 * sc, j);
                    if (site == site_i) down = !down;  // the same site again later in the window: its spin has turned
                    alive = alive && lane > l0;
                    ++flips;
                    if (flips >= K) {  // the filter's flip budget is used up: the window ends behind this update
                        next_t0 = base_t + __builtin_amdgcn_readlane(pos, l0) + 1;
                        CLFC_ADD(5, 1);
                        break;
                    }
                    m = evaluate();
                }
                if (lane == 0) {
                    ctrl[CLFC_NEXT] = next_t0;
                    st_rel(&queue[flips], CLFC_Q_END);  // behind the last entry (and behind NEXT): the window's end
                }
                __builtin_amdgcn_s_setprio(0);
            } else {
                // ---- the field waves: apply the rows of the accepted proposals as they are announced ----
                // this wave's chunks of a row: w - 1, w - 1 + 3, ...; a row is worked off in ITEMS of CLFC_PASS chunks
                auto chunk_j0 = [&](int c0, int q) -> long long { return ((long long)(c0 + q * CLFC_FIELD_WAVES) * 64 + lane) * EPL; };
                auto issue = [&](vec_t (&xr)[CLFC_PASS], int ent, int c0) {
                    const JT *row = Jbase + (long long)(ent & 0x3fffffff) * a.ldj;
#pragma unroll
                    for (int q = 0; q < CLFC_PASS; ++q) {  // (unconditional: a lane past the row's end reads its first granule)
                        const long long j0 = chunk_j0(c0, q);
                        xr[q] = *reinterpret_cast<const vec_t *>(row + (j0 < a.ldj ? j0 : 0));
                    }
                };
                auto apply = [&](const vec_t (&xr)[CLFC_PASS], int ent, int c0) {
                    const int mult = (ent & 0x40000000) ? 2 * sc : -2 * sc;  // -2 scale s_i(old)
#pragma unroll
                    for (int q = 0; q < CLFC_PASS; ++q) {
                        const long long j0 = chunk_j0(c0, q);
                        if (j0 < a.ldj) {
                            if (mult < 0) clf_apply_chunk<JT, FT, true>(F, xr[q], j0, mult, sc);
                            else clf_apply_chunk<JT, FT, false>(F, xr[q], j0, mult, sc);
                        }
                    }
                };
                // Two items in flight: the next item (of this row, or of the next queued accept) is requested before
                // the current one is applied -- unconditionally (nothing queued yet: the current item again, a cache
                // hit), so that the wait before the field update stays a counted one.
                const int per_wave = (n_chunks + CLFC_FIELD_WAVES - 1) / CLFC_FIELD_WAVES;
                const int P = (per_wave + CLFC_PASS - 1) / CLFC_PASS;  // items per row
                auto first_chunk = [&](int part) { return (w - 1) + part * CLFC_PASS * CLFC_FIELD_WAVES; };
                vec_t bufA[CLFC_PASS], bufB[CLFC_PASS];
                int idx = 0, part = 0;      // the item to apply next: queue entry, part of its row
                int have_idx = -1, have_part = 0;  // the item whose chunks the CURRENT buffer holds (-1: none)
                bool flipflop = false;
                auto step = [&](vec_t (&cur)[CLFC_PASS], vec_t (&oth)[CLFC_PASS], int ent) {
                    if (have_idx != idx || have_part != part) issue(cur, ent, first_chunk(part));
                    // the item after this one: the row's next part, else the next entry if it is there already
                    int nidx = idx, npart = part + 1, nent = ent;
                    if (npart == P) {
                        const int peek = ld_acq(&queue[idx + 1]);
                        if (peek >= 0) nidx = idx + 1, npart = 0, nent = peek;
                        else npart = part;  // nothing yet: this item again
                    }
                    issue(oth, nent, first_chunk(npart));
                    apply(cur, ent, first_chunk(part));
                    if (w == 1 && part == 0 && lane == 0) bits[(ent & 0x3fffffff) >> 5] ^= 1u << (ent & 31);  // the spin itself
                    have_idx = nidx, have_part = npart;
                    if (++part == P) part = 0, ++idx;
                };
                for (;;) {
                    const int ent = ld_acq(&queue[idx]);
                    if (ent == CLFC_Q_END) break;
                    if (ent < 0) {  // nothing announced yet
                        __builtin_amdgcn_s_sleep(4);
                        continue;
                    }
                    if (!flipflop) step(bufA, bufB, ent);
                    else step(bufB, bufA, ent);
                    flipflop = !flipflop;
                }
            }
            [[maybe_unused]] const long long tick_drain = CLFC_TICK();
            CLFC_ADD(7, tick_drain - tick_chain);
            __syncthreads();  // (5) fields and spins of the window's end state are in LDS
            CLFC_ADD(8, CLFC_TICK() - tick_drain);
            t0 = ctrl[CLFC_NEXT];
            if (w == 0) queue[lane] = CLFC_Q_NONE, queue[64 + lane] = CLFC_Q_NONE;  // (everybody is past the queue)
            __syncthreads();  // (6) everybody has the next start before the control words are reused
        }
        // sweep boundary: the chain wave's energy for everybody; record, best tracking (gpu_annealer.py:151-153)
        if (tid == 0) {
            E += (double)(2 * ksum) * inv_sc;  // (integers below 2^53: exact)
            ksum = 0;
            const long long eb = __double_as_longlong(E);
            ctrl[CLFC_E_LO] = (int)(unsigned int)eb;
            ctrl[CLFC_E_HI] = (int)(eb >> 32);
        }
        __syncthreads();
        const double Es = __longlong_as_double((long long)(((unsigned long long)(unsigned int)ctrl[CLFC_E_HI] << 32) |
                                                            (unsigned int)ctrl[CLFC_E_LO]));
        if (tid == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = Es;
        if (Es < bestE && !a.no_best) {
            bestE = Es;
            bits_to_spins(bits, a.best_spins + (long long)r * a.sstride, a.sstride, n, tid, blockDim.x);
        }
        __syncthreads();
    }

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
#ifdef CLFC_PROFILE
        prof[10] = wall_clock64() - prof_t0;
        if (a.energy_trace && a.n_sweeps >= 12)
            for (int i = 0; i < 12; ++i) a.energy_trace[(long long)i * a.R + r] = (double)prof[i];
#endif
    }
#undef CLFC_TICK
#undef CLFC_ADD
}

}  // namespace sga
