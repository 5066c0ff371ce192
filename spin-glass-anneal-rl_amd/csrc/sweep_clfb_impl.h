// sweep_clfb_impl.h -- the cached-local-field sweep (sweep_clf_impl.h), SEVERAL ACCEPTS PER ROUND.
//
// Same chain, same state, same LDS layout of fields / spins / accept table and the same replaced reference code as
// sweep_clf_kernel (SpinDynamics.sweep / _metropolis_update, core/spin_dynamics.py:73-94,131-152, batched over
// replicas; dE from a maintained field as core/energy_computer.py:166-173,262-265 does) -- what differs is how a
// super-window of W x 128 updates is walked.  sweep_clf_kernel takes ONE accept per round: evaluate all candidates,
// find the first that flips, apply its row, evaluate again -- two barriers and ~230 issued instructions per wave per
// accept, and a launch lasts as long as its replica with the most accepts (profiles/r04_experiments.md 3).
//
// Here a round GUESSES all decisions of the super-window at once and checks the guess:
//   1. every wave evaluates its 128 updates against the state in LDS and publishes, in chain order, the candidates
//      that flip (position, site, spin) -- the guess; up to CLFB_LIST of them make the round's batch.  The rows of the
//      first two are requested at once (registers): they travel during the check;
//   2. the guess is exact for every update up to and including the first accept.  Behind an accept a candidate's
//      k = s_i F_i has moved by  -2 scale s_a s_i J[a][i]  for each accepted a before it -- by at most m D behind m
//      accepts (D = 2 scale max|J|), so with an accept table that does not increase anywhere "u >= p(k - m D)"
//      stays a rejection whatever those accepts are.  The few candidates this does not settle (the guessed accepts
//      themselves, moves near their threshold, sites proposed again behind their own accept) look at their
//      couplings: one lane per accept of the batch gathers J[a][site], a wave sum corrects the field read in step
//      1, the decision is taken again;
//   3. the first position where the second decision differs from the guess (or where an accept sits behind an
//      accept of its own site: the list carries its old spin) ends the batch: everything before it IS the chain --
//      a decision depends on earlier decisions only, and those agree with the guess there -- and that position's
//      own second decision is exact too.  Without such a position the whole batch stands;
//   4. the rows of all committed accepts are applied two at a time (each wave to its own chunks of the field array,
//      no barrier in between: the fields of a chunk are read once, both rows' shares added in registers -- one
//      v_pk_mad_i16 per pair of int16 fields and row --, written once), the spins flipped, and the walk continues
//      behind the last decided update.
// A round commits at least its first accept (the old round), typically three or four for the hottest replica of a
// ladder and the whole window for the cold ones.  Energies: dE = 2 k / scale of every committed accept is an integer
// or half-integer below 2^40, so their sum per sweep, formed once at the sweep's end, is the value the
// one-at-a-time sum has (exact in fp64 in any order).
//
// Production arguments only (Philox sites, Metropolis through the accept table, no per-update records), matrices
// below 4 GiB (32-bit row offsets): the other cases keep sweep_clf_kernel.  Byte model as there: B = acceptance rate
// x n x sizeof(J element) per attempt.
#pragma once
#include "sweep_clf_impl.h"

namespace sga {

#ifndef CLFB_LIST_N  // (A/B builds: profiles/build_variant.sh)
#define CLFB_LIST_N 8  // (4: 0.259 ms at sweeps 5-25 and 1.88 on the first, hot sweeps; 6 ... 12: 0.244 and 1.73; 16: 0.247 / 1.85; 32: 0.249 / 2.29)
#endif
#ifndef CLFB_PASS_N
#define CLFB_PASS_N 3
#endif
constexpr int CLFB_LIST = CLFB_LIST_N;  // accepts one round commits at most
constexpr int CLFB_PASS = CLFB_PASS_N;  // candidates of either stream (even / odd updates) whose couplings are gathered together
constexpr int CLFB_ROWS = 2;   // rows of a batch applied together: a wave reads its fields once for both

// LDS behind the accept table: list [2][64] int2 | count [2][8] int | over [2][8] int | check [8] int4 | sums [2] u64 |
// three bitmaps of sstride bits: sites proposed in this
// super-window (seen), proposed more than once (twice), and those of the latter accepted by this round's guess (accb)
constexpr int CLFB_FIXED_BYTES = 2 * 64 * 8 + 2 * CLF_MAX_WAVES * 4 + 2 * CLF_MAX_WAVES * 4 + CLF_MAX_WAVES * 16 + 16;
inline size_t clfb_lds_bytes(long long ldf, int fbytes, int sstride, int table_m) {
    return (size_t)clf_table_offset(ldf, fbytes, sstride) + sizeof(float) * (size_t)((table_m + 4) & ~3) + CLFB_FIXED_BYTES +
           3 * (size_t)(sstride / 8);
}

// workgroup barrier for LDS traffic only: global loads stay in flight across it
__device__ __forceinline__ void clfb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (four waves per SIMD = every replica of a 1024-replica launch resident at four waves per replica: the short-row builds
//  are held to 128 registers; the long-row build keeps what it needs)
template <typename JT, typename FT, int CLF_BATCH = CLF_BATCH_MAX, bool TAIL = true>
__global__ void __launch_bounds__(64 * CLF_MAX_WAVES) __attribute__((amdgpu_waves_per_eu(TAIL ? 1 : 4)))
sweep_clfb_kernel(const SweepArgs a) {
    constexpr int EPL = 16 / (int)sizeof(JT), EPC = 64 * EPL;  // elements per lane / per 1-KiB chunk
    constexpr int FB = (int)sizeof(FT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FT *F = reinterpret_cast<FT *>(smem);
    unsigned int *bits = reinterpret_cast<unsigned int *>(smem + clf_bits_offset(a.ldf, FB));
    float *ptab = reinterpret_cast<float *>(smem + clf_table_offset(a.ldf, FB, a.sstride));
    int2 *list2 = reinterpret_cast<int2 *>(ptab + ((a.table_m + 4) & ~3));
    int *count2 = reinterpret_cast<int *>(list2 + 2 * 64);
    int *over2 = count2 + 2 * CLF_MAX_WAVES;
    int4 *check = reinterpret_cast<int4 *>(over2 + 2 * CLF_MAX_WAVES);
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(check + CLF_MAX_WAVES);
    unsigned int *seen = reinterpret_cast<unsigned int *>(sums + 2);
    unsigned int *twice = seen + a.sstride / 32;
    unsigned int *accb = twice + a.sstride / 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int W = (int)(blockDim.x >> 6);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = a.rep_list ? __builtin_amdgcn_readfirstlane(a.rep_list[blockIdx.x]) : (int)blockIdx.x, n = a.n;
    const int sc = a.field_scale;
    const double inv_sc = 1.0 / (double)sc;  // 1 | 0.5: exact
    // the round's list: wave v owns the S slots v S ... v S + S - 1, lane l reads slot l -- slot order is chain order
    const int S = 64 / W;
    const int my_v = lane / S, my_j = lane - my_v * S;
    const bool slot_ok = my_v < W;

    {   // resident state -> LDS
        const int4 *src = reinterpret_cast<const int4 *>(reinterpret_cast<const FT *>(a.fields) + (long long)r * a.ldf);
        int4 *dst = reinterpret_cast<int4 *>(F);
        for (int i = tid; i < (int)(a.ldf * FB / 16); i += blockDim.x) dst[i] = src[i];
        spins_to_bits(a.spins + (long long)r * a.sstride, bits, a.sstride, tid, blockDim.x);
        for (int i = tid; i < 3 * (a.sstride / 32); i += blockDim.x) seen[i] = 0u;
    }
    __syncthreads();

    const unsigned char *Jbytes = reinterpret_cast<const unsigned char *>(a.J);
    const JT *Jbase = reinterpret_cast<const JT *>(a.J);
    const int n_chunks = (int)((a.ldj + EPC - 1) / EPC);
    double E = a.energy[r], bestE = a.best_energy[r];
    unsigned long long nacc = 0;
    double T = 1.0;

    // a row dealt to the waves in 1-KiB chunks, chunk c -> wave c mod W, as in sweep_clf_kernel
    using vec_t = typename std::conditional<sizeof(JT) == 4, float4, int4>::type;
    struct RowRegs {
        vec_t x[CLF_BATCH];
    };
    auto elem0 = [&](int c) -> long long { return ((long long)c * 64 + lane) * EPL; };
    unsigned int req_off[CLF_BATCH];
#pragma unroll
    for (int q = 0; q < CLF_BATCH; ++q) {
        const long long j0 = elem0(w + q * W);
        req_off[q] = (unsigned int)((j0 < a.ldj ? j0 : 0) * (long long)sizeof(JT));
    }
    const unsigned int pitch = (unsigned int)(a.ldj * (long long)sizeof(JT));
    // chunks of this wave's first batch that exist at all (wave-uniform), and one chunk of one row
    int nq = 0;
#pragma unroll
    for (int q = 0; q < CLF_BATCH; ++q)
        if ((long long)(w + q * W) * EPC < a.ldj) nq = q + 1;
    auto chunk_load = [&](int site, int q) -> vec_t {
        const unsigned char *row = Jbytes + (unsigned long long)(unsigned int)site * pitch;  // wave-uniform
        unsigned int off = req_off[q];
        asm volatile("" : "+v"(off));  // (kept 32-bit: scalar row base + one offset register per load)
#ifdef CLFB_NOLOAD
        vec_t o{};
        asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(o.w) : "s"(row));
        return o;
#else
        return *reinterpret_cast<const vec_t *>(row + off);
#endif
    };
    // Up to CLFB_ROWS rows into the wave's fields, chunk by chunk: the fields of a chunk are read once, every row's
    // share added in registers, written once.  buf holds the rows' chunks on entry where `preloaded` (the first
    // two listed accepts, requested during the check); later pairs of rows are requested here, all chunks at once.
    auto apply_rows = [&](auto kc, vec_t (&buf)[CLFB_ROWS][CLF_BATCH], const int (&sites)[CLFB_ROWS], const int (&mults)[CLFB_ROWS],
                          bool preloaded) {
        constexpr int K = decltype(kc)::value;
        if (!preloaded) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int q = 0; q < CLF_BATCH; ++q) buf[k][q] = chunk_load(sites[k], q);
        }
#pragma unroll
        for (int q = 0; q < CLF_BATCH; ++q) {
            const long long j0 = elem0(w + q * W);
            if (q < nq && j0 < a.ldj) {
                ClfFields<JT, FT> f = clf_fields_load<JT, FT>(F, j0);
#pragma unroll
                for (int k = 0; k < K; ++k) clf_fields_update_signed<JT, FT>(f, buf[k][q], mults[k], sc);
                clf_fields_store<JT, FT>(F, j0, f);
            }
        }
    };
    auto row_request = [&](int site) -> RowRegs {
        RowRegs o;
        const unsigned char *row = Jbytes + (unsigned long long)(unsigned int)site * pitch;  // wave-uniform
#pragma unroll
        for (int q = 0; q < CLF_BATCH; ++q) {
            unsigned int off = req_off[q];
            asm volatile("" : "+v"(off));  // (kept 32-bit: scalar row base + one offset register per load)
#ifdef CLFB_NOLOAD  // (timing experiment: what a round costs without its row fetches; wrong fields)
            o.x[q] = vec_t{};
            asm volatile("" : "+v"(o.x[q].x), "+v"(o.x[q].y), "+v"(o.x[q].z), "+v"(o.x[q].w) : "s"(row));
#else
            o.x[q] = *reinterpret_cast<const vec_t *>(row + off);
#endif
        }
        return o;
    };
    // chunks of this wave's first batch that lie fully inside the row (wave-uniform): their field update needs no
    // lane guard, and without the guards the three updates of a row are one straight line -- field reads up front
    int nfull = 0;
#pragma unroll
    for (int q = 0; q < CLF_BATCH; ++q)
        if ((long long)(w + q * W + 1) * EPC <= a.ldj) nfull = q + 1;
    auto first_chunk = [&](int q) -> long long { return elem0(w + q * W); };
    auto apply_row_signed = [&](const RowRegs &rr, int site, int mult, auto neg) {
        if (!TAIL && nfull == CLF_BATCH) {  // (the long-row build keeps the guarded form throughout)
            clf_apply_full_chunks<JT, FT, decltype(neg)::value, CLF_BATCH, false>(F, rr.x, first_chunk, mult, sc);  // (registers: chunk by chunk)
        } else if (!TAIL && nfull == CLF_BATCH - 1) {
            clf_apply_full_chunks<JT, FT, decltype(neg)::value, CLF_BATCH - 1, false>(F, rr.x, first_chunk, mult, sc);
            const long long j0 = elem0(w + (CLF_BATCH - 1) * W);
            if (j0 < a.ldj) clf_apply_chunk<JT, FT, decltype(neg)::value>(F, rr.x[CLF_BATCH - 1], j0, mult, sc);
        } else {
#pragma unroll
            for (int q = 0; q < CLF_BATCH; ++q) {
                const long long j0 = elem0(w + q * W);
                if (j0 < a.ldj) clf_apply_chunk<JT, FT, decltype(neg)::value>(F, rr.x[q], j0, mult, sc);
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
                if (j0 < a.ldj) clf_apply_chunk<JT, FT, decltype(neg)::value>(F, x[q], j0, mult, sc);
            }
        }
    };
    auto apply_row = [&](const RowRegs &rr, int site, int mult /* -2 scale s_i(old) */) {
        if (mult < 0) apply_row_signed(rr, site, mult, std::true_type{});  // wave-uniform
        else apply_row_signed(rr, site, mult, std::false_type{});
    };
    // does a move with k = s_i F_i flip?  Both candidates of a lane at once (their table reads travel together);
    // the table covers k <= table_m (entry 0 = 1 serves every downhill move), the few moves beyond it are evaluated
    // behind a wave-uniform test -- the same function of the same arguments as sweep_clf_kernel's
    auto accept2 = [&](int ka, float uA, bool liveA, bool &fA, int kb, float uB, bool liveB, bool &fB) {
        const float pa = ptab[min(max(ka, 0), a.table_m)], pb = ptab[min(max(kb, 0), a.table_m)];
        bool accA = uA < pa, accB = uB < pb;
        const bool beyondA = liveA && ka > a.table_m, beyondB = liveB && kb > a.table_m;
        if (ballot64(beyondA || beyondB)) {  // rare: large uphill moves (p == 0 past -104, sweep_common.h)
            const double dA = (double)(2 * ka) * inv_sc, dB = (double)(2 * kb) * inv_sc;
            if (beyondA) accA = !(dA > T * 104.0) && uA < expf_det((float)(-dA / T));
            if (beyondB) accB = !(dB > T * 104.0) && uB < expf_det((float)(-dB / T));
        }
        fA = liveA && accA, fB = liveB && accB;
    };
    int turn = 0;
    constexpr int NONE = 1 << 20;
#ifdef CLFB_PROFILE
    // Where a replica's time goes (profiles/r04_clfb_profile.py; 100 MHz ticks of s_memrealtime as wave 0 sees them,
    // summed over the launch, returned through the first rows of energy_trace): 0 super-windows, 1 rounds, 2 rows
    // applied, 3 rounds ended by a decision that changed, 4 listed accepts, 5 candidates of wave 0 that looked at
    // couplings, 6 ticks guess (to barrier A), 7 ticks check (to barrier A2), 8 ticks apply (to barrier B),
    // 9 ticks drawing the candidates, 10 ticks of the whole sweep loop, 11 ticks filling the table
    long long prof[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long prof_t0 = wall_clock64();
#define CLFB_TICK() wall_clock64()
#define CLFB_ADD(i, v) prof[i] += (v)
#else
#define CLFB_TICK() 0ll
#define CLFB_ADD(i, v) (void)0
#endif
    auto first_of = [](unsigned long long mA, unsigned long long mB) -> int {
        const int pA = mA ? 2 * (int)__builtin_ctzll(mA) : NONE;
        const int pB = mB ? 2 * (int)__builtin_ctzll(mB) + 1 : NONE;
        return min(pA, pB);
    };
    // (what the check adds up: couplings are integers below 2^7 | 2^24 times a factor of 2 or 4 -- exact either way)
    using corr_t = typename std::conditional<sizeof(JT) == 4, float, int>::type;

    for (int k = 0; k < a.n_sweeps; ++k) {
        T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        [[maybe_unused]] const long long tick_table = CLFB_TICK();
        __syncthreads();
        for (int q = tid; q <= a.table_m; q += blockDim.x)  // exp(float32(-dE/T)) of the moves dE = 2 q / scale
            ptab[q] = expf_det((float)(-((double)(2 * q) * inv_sc) / T));
        if (tid == 0) sums[0] = 0ull, sums[1] = 0ull;
        __syncthreads();
        // The check's filter reads "u >= p(k') is a rejection for every k >= k'": true while the table does not
        // increase anywhere.  expf_det is monotone up to single ulps between ADJACENT floats
        // (profiles/r04_experiments.md 3), which table entries -- 2 / (scale T) apart -- are not at any temperature
        // an annealer uses; checked all the same, per sweep (sums[1] counts the inversions), and without the
        // filter every candidate of the batch looks at its couplings.
        {
            bool up = false;
            for (int q = tid; q < a.table_m; q += blockDim.x) up = up || (ptab[q] < ptab[q + 1]);
            if (ballot64(up) != 0ull && lane == 0) atomicAdd(&sums[1], 1ull);
        }
        __syncthreads();
        const int D = 2 * sc * a.clf_jmax;  // the most one flip moves k = s_i F_i of another site
        const bool filter_ok = sums[1] == 0ull && (2.0 * inv_sc / T) >= 1.0e-4 && a.clf_jmax > 0 && a.clf_jmax < (1 << 18);
        __syncthreads();
        if (tid == 0) sums[1] = 0ull;
        CLFB_ADD(11, CLFB_TICK() - tick_table);
        long long ksum = 0;  // this lane's committed accepts of the sweep: sum of k, and how many
        int kcnt = 0;
        const int END = CLF_WINDOW * W;
        for (int t0 = 0; t0 < n; t0 += END) {
            // this lane's two candidates: updates tA and tA + 1 of sweep k
            [[maybe_unused]] const long long tick_draw = CLFB_TICK();
            const int wbase = w * CLF_WINDOW;
            const int gA = wbase + 2 * lane, gB = gA + 1;  // positions in the super-window
            const int tA = t0 + gA, tB = tA + 1;
            const bool vA = tA < n, vB = tB < n;
            uint32_t key_lo = a.seed_lo, key_hi = a.seed_hi;
            asm volatile("" : "+s"(key_lo), "+s"(key_hi));  // (sweep_clf_impl.h: the round keys formed on the spot)
            const u32x4 x = philox4x32_10((uint32_t)(tA >> 1), a.sweep0 + (uint32_t)k, a.replica0 + (uint32_t)r, DOMAIN_SWEEP,
                                          key_lo, key_hi);
            const int sA = (int)word_to_site(x.x, (uint32_t)n), sB = (int)word_to_site(x.z, (uint32_t)n);
            const float uA = word_to_u(x.y), uB = word_to_u(x.w);
            const unsigned int bitA = 1u << (sA & 31), bitB = 1u << (sB & 31);
            // sites proposed more than once in this super-window: only there can an accept turn the spin another
            // candidate has read.  First everybody marks its site, then whoever found the mark says so for both.
            unsigned int oldA = 0u, oldB = 0u;
            if (vA) oldA = atomicOr(&seen[sA >> 5], bitA);
            if (vB) oldB = atomicOr(&seen[sB >> 5], bitB);
            clfb_barrier();
            const bool laterA = vA && (oldA & bitA), laterB = vB && (oldB & bitB);
            if (laterA) atomicOr(&twice[sA >> 5], bitA);
            if (laterB) atomicOr(&twice[sB >> 5], bitB);
            if (vA) atomicAnd(&seen[sA >> 5], ~bitA);  // (clean again for the next super-window)
            if (vB) atomicAnd(&seen[sB >> 5], ~bitB);
            clfb_barrier();
            const bool dupA = vA && (twice[sA >> 5] & bitA), dupB = vB && (twice[sB >> 5] & bitB);
            int pos = 0;  // super-window positions below pos are decided
            CLFB_ADD(9, CLFB_TICK() - tick_draw), CLFB_ADD(0, 1);

            // one round; true = the super-window is done
            auto round = [&]() -> bool {
                [[maybe_unused]] const long long tick0 = CLFB_TICK();
                CLFB_ADD(1, 1);
                int2 *list = list2 + turn * 64;
                int *count = count2 + turn * CLF_MAX_WAVES, *over = over2 + turn * CLF_MAX_WAVES;
                turn ^= 1;
                // 1. the guess: this wave's candidates against the state as it stands
                int fa = 0, fb = 0, siA = 1, siB = 1;
                bool gsA = false, gsB = false;
                unsigned long long mA = 0ull, mB = 0ull;
                if (wbase + CLF_WINDOW > pos) {  // (a wave whose window is decided publishes "nothing")
                    fa = (int)F[sA], fb = (int)F[sB];
                    const unsigned int wa = bits[sA >> 5], wb = bits[sB >> 5];
                    siA = (wa & bitA) ? -1 : 1;
                    siB = (wb & bitB) ? -1 : 1;
                    accept2(siA * fa, uA, vA && gA >= pos, gsA, siB * fb, uB, vB && gB >= pos, gsB);
                    mA = ballot64(gsA), mB = ballot64(gsB);
                }
                // rank in chain order within the wave = guessed accepts of this window before the candidate
                const int rA = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mA, 0u)) +
                               (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mB, 0u));
                const int rB = rA + (gsA ? 1 : 0);
                const int cw = (int)__popcll(mA) + (int)__popcll(mB);
                if (cw) {  // wave-uniform
                    int2 *mine = list + w * S;
                    if (gsA && rA < S) mine[rA] = make_int2(gA, sA | (siA < 0 ? (int)0x80000000 : 0));
                    if (gsB && rB < S) mine[rB] = make_int2(gB, sB | (siB < 0 ? (int)0x80000000 : 0));
                    if (gsA && rA == S) over[w] = gA;  // (more accepts than slots: the batch ends before this one)
                    if (gsB && rB == S) over[w] = gB;
                    if (gsA && dupA) atomicOr(&accb[sA >> 5], bitA);
                    if (gsB && dupB) atomicOr(&accb[sB >> 5], bitB);
                }
                if (lane == 0) {
                    count[w] = cw;
                    if (cw <= S) over[w] = NONE;
                }
                CLFB_ADD(12, CLFB_TICK() - tick0);
                clfb_barrier();  // (A) every wave has evaluated against the old state and published
                [[maybe_unused]] const long long tick1 = CLFB_TICK();
                CLFB_ADD(6, tick1 - tick0);
                // 2. the batch: the listed accepts before L, in slot = chain order
                const int2 ent = list[lane];
                const int cv = slot_ok ? count[my_v] : 0;
                const bool valid = my_j < cv;  // (my_j < S; cv = 0 beyond the last wave)
                const unsigned long long vm = ballot64(valid);
                if (vm == 0ull) return true;  // the rest of the super-window is rejected
                int L = END;
                if (__popcll(vm) > CLFB_LIST) {
                    unsigned long long t = vm;
                    for (int i = 0; i < CLFB_LIST; ++i) t &= t - 1ull;
                    L = __builtin_amdgcn_readlane(ent.x, (int)__builtin_ctzll(t));
                }
                if (ballot64(cv > S)) {
                    const int ov = lane < W ? over[lane] : NONE;
                    for (int v = 0; v < W; ++v) L = min(L, __builtin_amdgcn_readlane(ov, v));
                }
                const bool inb = valid && ent.x < L;
                const unsigned long long vmL = ballot64(inb);  // (never empty: the first accept lies before L)
                const int ent_site = ent.y & 0x7fffffff;
                const unsigned int rowoff = (unsigned int)ent_site * pitch;
                const corr_t multl = (corr_t)(ent.y < 0 ? 2 * sc : -2 * sc);  // -2 scale s_a
                // (the first accept always stands, the second nearly always: their rows are asked for during the check,
                //  behind the couplings the check itself waits for -- loads return in order)
                const unsigned long long vm2 = vmL & (vmL - 1ull);
                const int early1 = __builtin_amdgcn_readlane(ent_site, (int)__builtin_ctzll(vm));
                const int early2 = vm2 ? __builtin_amdgcn_readlane(ent_site, (int)__builtin_ctzll(vm2)) : early1;
                [[maybe_unused]] RowRegs cur, nxt;
                vec_t rowbuf[CLFB_ROWS][CLF_BATCH];  // (the rows of the first two listed accepts travel during the check)
                [[maybe_unused]] const long long tick1a = CLFB_TICK();
                CLFB_ADD(13, tick1a - tick1);
                // 3. the check
                const bool mine_live = wbase + CLF_WINDOW > pos && wbase < L;  // wave-uniform: candidates in [pos, L)?
                const bool inA = vA && gA >= pos && gA < L, inB = vB && gB >= pos && gB < L;
                corr_t cA = 0, cB = 0;
                int hA = 0, hB = 0;  // accepts of the candidate's own site before it
                bool acA = false, acB = false, needA = false, needB = false;
                unsigned long long bA = 0ull, bB = 0ull;
                if (mine_live) {
                    needA = inA, needB = inB;
                    if (filter_ok) {
                        const int Mb = (int)__popcll(vmL), nbefore = (int)__popcll(vmL & ((1ull << (w * S)) - 1ull));
                        const int slA = min(nbefore + rA, Mb) * D, slB = min(nbefore + rB, Mb) * D;
                        const float qa = ptab[min(max(siA * fa - slA, 0), a.table_m)], qb = ptab[min(max(siB * fb - slB, 0), a.table_m)];
                        const bool tA2 = dupA && (accb[sA >> 5] & bitA), tB2 = dupB && (accb[sB >> 5] & bitB);
                        needA = inA && (gsA || tA2 || uA < qa), needB = inB && (gsB || tB2 || uB < qb);
                    }
                }
                {
                    // one lane per listed accept: J[a][site] of the accepts before the candidate, summed over the wave;
                    // up to CLFB_PASS candidates of either stream per pass, all their loads in flight together
                    unsigned long long nA = ballot64(needA), nB = ballot64(needB);
                    bool first_pass = true;
                    do {  // wave-uniform
                        int ls[2 * CLFB_PASS], hp[2 * CLFB_PASS];
                        bool on[2 * CLFB_PASS];
                        JT xs[2 * CLFB_PASS];
#pragma unroll
                        for (int j = 0; j < 2 * CLFB_PASS; ++j) {
                            unsigned long long &need = j < CLFB_PASS ? nA : nB;
                            on[j] = need != 0ull;
                            ls[j] = on[j] ? (int)__builtin_ctzll(need) : 0;
                            need &= need - 1ull;
                            xs[j] = (JT)0, hp[j] = 0;
                            if (on[j]) {
                                const int s_c = __builtin_amdgcn_readlane(j < CLFB_PASS ? sA : sB, ls[j]);
                                const bool act = inb && ent.x < wbase + 2 * ls[j] + (j < CLFB_PASS ? 0 : 1);
                                hp[j] = (int)__popcll(ballot64(act && ent_site == s_c));
                                if (act) xs[j] = *reinterpret_cast<const JT *>(Jbytes + (unsigned long long)(unsigned int)s_c * sizeof(JT) + rowoff);
                            }
                        }
                        if (first_pass) {
                            if constexpr (TAIL) {
                                cur = row_request(early1), nxt = row_request(early2);
                            } else {
#pragma unroll
                                for (int q = 0; q < CLF_BATCH; ++q) rowbuf[0][q] = chunk_load(early1, q);
#pragma unroll
                                for (int q = 0; q < CLF_BATCH; ++q) rowbuf[1][q] = chunk_load(early2, q);
                            }
                            first_pass = false;
                        }
#pragma unroll
                        for (int j = 0; j < 2 * CLFB_PASS; ++j) {
                            if (on[j]) {
                                const corr_t c = wave_sum(multl * (corr_t)xs[j]);
                                const bool owner = lane == ls[j];
                                if (j < CLFB_PASS) cA = owner ? c : cA, hA = owner ? hp[j] : hA;
                                else cB = owner ? c : cB, hB = owner ? hp[j] : hB;
                            }
                        }
                    } while (nA | nB);
                    CLFB_ADD(14, CLFB_TICK() - tick1a);
                }
                if (mine_live) {
                    const int si2A = (hA & 1) ? -siA : siA, si2B = (hB & 1) ? -siB : siB;
                    const int k2A = si2A * (fa + (int)cA), k2B = si2B * (fb + (int)cB);
                    // (a candidate that did not have to look keeps the guess: a rejection)
                    accept2(k2A, uA, needA, acA, k2B, uB, needB, acB);
                    // an accept behind an accept of its own site ends the batch too: the list carries its old spin
                    bA = ballot64(inA && (acA != gsA || (acA && hA != 0)));
                    bB = ballot64(inB && (acB != gsB || (acB && hB != 0)));
                    fa = k2A, fb = k2B, siA = si2A, siB = si2B;  // (from here on: the checked move and spin)
                }
                {
                    const int q = first_of(bA, bB);
                    int4 mine = make_int4(NONE, 0, 0, 1);
                    if (q < NONE) {
                        const int l = q >> 1;
                        const int qa = __builtin_amdgcn_readlane((q & 1) ? (int)acB : (int)acA, l);
                        const int qs = __builtin_amdgcn_readlane((q & 1) ? sB : sA, l);
                        const int qi = __builtin_amdgcn_readlane((q & 1) ? siB : siA, l);
                        mine = make_int4(q + wbase, qa, qs, qi);
                    }
                    if (lane == 0) check[w] = mine;
                }
                CLFB_ADD(15, CLFB_TICK() - tick1a);
                clfb_barrier();  // (A2) every wave has checked its window
                [[maybe_unused]] const long long tick2 = CLFB_TICK();
                CLFB_ADD(7, tick2 - tick1), CLFB_ADD(4, __popcll(vmL)), CLFB_ADD(5, __popcll(ballot64(needA)) + __popcll(ballot64(needB)));
                int4 ck = make_int4(NONE, 0, 0, 1);
                if (lane < W) ck = check[lane];
                const unsigned long long have = ballot64(ck.x < NONE);
                int Q = L, qpos = NONE, xacc = 0, xsite = 0, xsi = 1;
                if (have) {  // (windows are in chain order: the first wave that reports holds the earliest position)
                    const int win = (int)__builtin_ctzll(have);
                    qpos = __builtin_amdgcn_readlane(ck.x, win), xacc = __builtin_amdgcn_readlane(ck.y, win);
                    xsite = __builtin_amdgcn_readlane(ck.z, win), xsi = __builtin_amdgcn_readlane(ck.w, win);
                    Q = qpos;
                }
                // 4. commit: the guessed accepts before Q, and position Q itself as decided by the check
                {
                    const bool comA = inA && ((gsA && gA < Q) || (gA == qpos && acA));
                    const bool comB = inB && ((gsB && gB < Q) || (gB == qpos && acB));
                    ksum += (comA ? (long long)fa : 0ll) + (comB ? (long long)fb : 0ll);
                    kcnt += (comA ? 1 : 0) + (comB ? 1 : 0);
                }
                if (cw) {
                    if (gsA && dupA) atomicAnd(&accb[sA >> 5], ~bitA);
                    if (gsB && dupB) atomicAnd(&accb[sB >> 5], ~bitB);
                }
                const bool mine_row = valid && ent.x < Q;
                unsigned long long rows = ballot64(mine_row);  // >= 1 bit: the first accept always stands
                const int nrows = (int)__popcll(rows) + (xacc ? 1 : 0);
                if (w == 0) {
                    if (mine_row) atomicXor(&bits[ent_site >> 5], 1u << (ent_site & 31));
                    if (lane == 0 && xacc) atomicXor(&bits[xsite >> 5], 1u << (xsite & 31));
                }
                auto next_row = [&](int &site, int &mult) {  // wave-uniform; slot order, then the checked position
                    if (rows) {
                        const int ey = __builtin_amdgcn_readlane(ent.y, (int)__builtin_ctzll(rows));
                        rows &= rows - 1ull;
                        site = ey & 0x7fffffff, mult = ey < 0 ? 2 * sc : -2 * sc;
                    } else {
                        site = xsite, mult = -2 * sc * xsi;
                    }
                };
                CLFB_ADD(16, CLFB_TICK() - tick2);
                if constexpr (!TAIL) {
                    // The committed rows in order (the guessed accepts before Q in slot order, then the checked position):
                    // lane j of `tab` holds row j's site and spin.  Row j < napply IS the j-th listed accept, so the first
                    // pass -- rows 0 .. min(2, napply) - 1 -- finds its rows already requested during the check.
                    const int napply = (int)__popcll(rows);
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(rows >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)rows, 0u));
                    int tab = __builtin_amdgcn_ds_permute((mine_row ? rank : 63) << 2, ent.y);
                    if (lane == napply && xacc) tab = xsite | (xsi < 0 ? (int)0x80000000 : 0);
                    int sites[CLFB_ROWS], mults[CLFB_ROWS];
                    auto take = [&](int r0, int k) {
#pragma unroll
                        for (int j = 0; j < CLFB_ROWS; ++j) {
                            const int ey = __builtin_amdgcn_readlane(tab, r0 + (j < k ? j : 0));
                            sites[j] = ey & 0x7fffffff, mults[j] = ey < 0 ? 2 * sc : -2 * sc;
                        }
                    };
                    auto run = [&](int k, bool preloaded) {  // wave-uniform
                        if (k == 1) apply_rows(std::integral_constant<int, 1>{}, rowbuf, sites, mults, preloaded);
                        else apply_rows(std::integral_constant<int, 2>{}, rowbuf, sites, mults, preloaded);
                    };
                    int done = min(CLFB_ROWS, napply);
                    take(0, done);
                    run(done, true);
                    while (done < nrows) {
                        const int k = min(CLFB_ROWS, nrows - done);
                        take(done, k);
                        run(k, false);
                        done += k;
                    }
                } else
                {   // The two row buffers take turns: step 0 applies the first accept's row (cur), step 1 the second
                    // listed accept's (nxt) -- both asked for during the check --, step i + 2 the row asked for as soon
                    // as step i's buffer was free.  Every step asks for exactly one row (past the last: the current one
                    // again, a cache hit) and nothing else loads in between, so the loads in flight are the same on
                    // every path and the waits stay counted.  Where the second listed accept does not stand its step
                    // applies nothing (weight 0) and the rows behind it move one step back.
                    int site, mult, s1 = early2, m1 = 0, s2 = 0, m2 = 0;
                    next_row(site, mult);
                    int nsteps = nrows;
                    if (nrows > 1) {
                        const unsigned long long second = vm2 & (0ull - vm2);  // the second listed accept's slot
                        if (rows & second) next_row(s1, m1);                  // (it is the next row in slot order)
                        else ++nsteps;
                    }
                    for (int i = 0;;) {
                        apply_row(cur, site, mult);
                        if (i + 2 < nsteps) next_row(s2, m2);
                        else s2 = site, m2 = 0;
                        cur = row_request(s2);
                        if (++i >= nsteps) break;
                        site = s1, mult = m1, s1 = s2, m1 = m2;
                        if (mult != 0) apply_row(nxt, site, mult);
                        if (i + 2 < nsteps) next_row(s2, m2);
                        else s2 = site, m2 = 0;
                        nxt = row_request(s2);
                        if (++i >= nsteps) break;
                        site = s1, mult = m1, s1 = s2, m1 = m2;
                    }
                }
                pos = have ? qpos + 1 : L;
                CLFB_ADD(17, CLFB_TICK() - tick2);
                clfb_barrier();  // (B) fields and spins of the new state are visible
                CLFB_ADD(8, CLFB_TICK() - tick2), CLFB_ADD(2, nrows), CLFB_ADD(3, have ? 1 : 0);
                return pos >= END;
            };
            while (!round()) {
            }
            if (laterA) atomicAnd(&twice[sA >> 5], ~bitA);  // (clean again; the next super-window marks behind a barrier)
            if (laterB) atomicAnd(&twice[sB >> 5], ~bitB);
        }
        // sweep boundary: the sweep's accepts into the energy, energy record, best tracking (annealing/gpu_annealer.py:151-153)
        for (int o = 32; o; o >>= 1) {
            ksum += __shfl_xor(ksum, o);
            kcnt += __shfl_xor(kcnt, o);
        }
        if (lane == 0) {
            atomicAdd(&sums[0], (unsigned long long)ksum);
            atomicAdd(&sums[1], (unsigned long long)kcnt);
        }
        __syncthreads();
        E += (double)(2ll * (long long)sums[0]) * inv_sc;
        nacc += sums[1];
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
#ifdef CLFB_PROFILE
        prof[10] = wall_clock64() - prof_t0;
        if (a.energy_trace && a.n_sweeps >= 20)
            for (int i = 0; i < 18; ++i) a.energy_trace[(long long)i * a.R + r] = (double)prof[i];
#endif
    }
#ifdef CLFB_PROFILE
    if (tid == 64 * (W - 1) && a.energy_trace && a.n_sweeps >= 20) {  // the last wave's view: who looked, how long its check took
        a.energy_trace[18ll * a.R + r] = (double)prof[5];
        a.energy_trace[19ll * a.R + r] = (double)prof[15];
    }
#endif
#undef CLFB_TICK
#undef CLFB_ADD
}

}  // namespace sga
