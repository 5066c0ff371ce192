// sweep_csr_rows4.hip -- narrow CSR sweep, FOUR UPDATES PER STEP: BASELINE configs[2] (10 000 spins,
// degree ~32, 4096 replicas) in the production configuration (integer couplings and fields, Philox sites,
// Metropolis with the accept table, int8 spins in LDS).
//
// Replaces the same reference functions as sweep_csr_kernel (core/spin_dynamics.py:73-94,131-152,
// core/ising_model.py:176-185) and walks the same chain bit for bit.
//
// The one-update-at-a-time form (sweep_csr_impl.h) spends ~65 instructions of one wave on every update and
// uses 32 of its 64 lanes at degree 32; it is bound by instruction issue and its dependent chain, not by
// memory (the 2.6 MB structure is L2 resident).  Here a wave works on the four consecutive updates
// t = 4m .. 4m + 3 at once, one per ROW of 16 lanes (lane j of a row holds entries 4j .. 4j + 3 of the
// row's coupling row: rows of up to 64 entries), every row sum against the spins as they stand before
// the first of the four:
//   * sparse couplings make that exact almost always: flipping site A changes the local field of B only
//     if J[B][A] != 0, and B's own spin only if B == A.  The four decisions are formed together, then
//     checked in chain order: for every accepted update the later rows look for its site among their
//     columns (and their own site) -- one compare per entry and a ballot.  No hit (99 % of the steps at
//     degree 32 of 10 000): all four decisions are the chain's.  A hit: the step is replayed one update
//     at a time (same data, already in registers).
//   * everything is an integer below 2^24 (the table form's precondition), so row sums, dE and the
//     energy are exact in any order: E += the sum of the accepted dE of the step.
// Per step: 4 x 16 row entries in two loads per lane, 4 LDS spin gathers per lane, a 4-step DPP row sum
// (all four rows in the same instructions), one table look-up; sites, row extents and row entries are
// requested two / one steps ahead (the site sequence is known from the counter RNG).
#include "sweep_csr_impl.h"

namespace sga {

__global__ void __launch_bounds__(64 * CSR_WAVES_PER_BLOCK) sweep_csr_rows4_kernel(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int nw = blockDim.x >> 6;
    const int r = (int)blockIdx.x * nw + w;
    if (r >= a.R) return;  // wave-uniform; no barriers in this kernel
    const int n = a.n;
    // LDS as in the narrow form of sweep_csr_kernel: [nw] spin slices, then [nw] accept tables
    int8_t *s = reinterpret_cast<int8_t *>(smem) + (long long)w * a.sstride;
    unsigned int *itab = reinterpret_cast<unsigned int *>(smem + (long long)nw * a.sstride) + (long long)w * (a.table_m + 1);
    {
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (long long)r * a.sstride);
        int4 *dst = reinterpret_cast<int4 *>(s);
        for (int i = lane; i < a.sstride / 16; i += 64) dst[i] = src[i];
    }
    const int g = lane >> 4, j = lane & 15;  // row of the wave = update of the step, lane in the row
    double E = a.energy[r], bestE = a.best_energy[r];
    unsigned long long nacc = 0;
    double T = 1.0;
    const int nb = (n + 1) >> 1;     // Philox pairs per sweep
    const int steps = (n + 3) >> 2;  // steps per sweep (the last one may hold fewer than four updates)

    struct Step {
        int site;       // this row's site
        uint32_t ru;    // its uniform's 24 raw bits
        int live;       // the update exists (t < n, sweep < n_sweeps)
        int beg, end;   // row extent
        float h;
        int2 e[4];      // entries beg + 4 j + q
    };
    PairSource<true> rng;
    // the sites of step m of sweep k (valid: the sweep exists; past the end every row is dead, site 0)
    auto stage_sites = [&](Step &st, int k, int m) {
        const bool valid = k < a.n_sweeps;
        const int b0 = 2 * m, b1 = 2 * m + 1;
        const UpdatePair p0 = rng.get(a, r, k, b0, valid && b0 < nb, lane);
        const UpdatePair p1 = rng.get(a, r, k, b1, valid && b1 < nb, lane);
        const int sa = (g & 2) ? p1.sA : p0.sA, sb = (g & 2) ? p1.sB : p0.sB;
        const uint32_t ra = (g & 2) ? p1.rA : p0.rA, rb = (g & 2) ? p1.rB : p0.rB;
        st.live = (valid && 4 * m + g < n) ? 1 : 0;
        st.site = st.live ? ((g & 1) ? sb : sa) : 0;
        st.ru = (g & 1) ? rb : ra;
    };
    // scalar base + 32-bit lane offset: the scalar-base form of global_load (as in sweep_csr_impl.h)
    auto stage_extents = [&](Step &st) {
        unsigned int off = (unsigned int)st.site * 4u;
        asm volatile("" : "+v"(off));
        const unsigned char *rp = reinterpret_cast<const unsigned char *>(a.rowptr);
        st.beg = *reinterpret_cast<const int *>(rp + off);
        st.end = *reinterpret_cast<const int *>(rp + off + 4);
        st.h = *reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.h) + off);
    };
    auto stage_heads = [&](Step &st) {
        // (64 zeroed entries follow the array: lanes past the row's end read what lies behind it)
        unsigned int off = (unsigned int)(st.beg + 4 * j) * 8u;
        asm volatile("" : "+v"(off));
        const unsigned char *cv = reinterpret_cast<const unsigned char *>(a.cv);
#pragma unroll
        for (int q = 0; q < 4; ++q) st.e[q] = *reinterpret_cast<const int2 *>(cv + off + 8 * q);
    };
    // sum over the 16 lanes of a row, in every lane of the row (exact: integers below 2^24)
    auto row_sum = [&](float v) -> float {
        v += dpp_move<DPP_QUAD_XOR1>(v);
        v += dpp_move<DPP_QUAD_XOR2>(v);
        v += dpp_move<DPP_ROW_HALF_MIRROR>(v);
        v += dpp_move<DPP_ROW_MIRROR>(v);
        return v;
    };
    // this row's decision against the spins as they stand: flips?, fk = s_i (row sum + h)
    auto decide = [&](const Step &st, int &si, float &fk) -> bool {
        const int left = st.end - st.beg - 4 * j;  // entries of the row from this lane's first on
        si = s[st.site];
        float dot = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = q < left ? __int_as_float(st.e[q].y) : 0.0f;
            dot += v * (float)s[st.e[q].x];
        }
        dot = row_sum(dot);
        // core/spin_dynamics.py:131-152 with every quantity an integer: dE = 2 fk exactly (half-integer
        // fields: table_scale = 2, the table is indexed by 2 fk = dE); sweep_csr_impl.h, TABLE branch
        fk = (float)si * (dot + st.h);
        const float fq = fk * (float)a.table_scale;
        const int idx = min(max((int)fq, 0), a.table_m);
        bool flip = fk <= 0.0f || st.ru < itab[idx];  // u < p on the uniform's raw bits
        const bool beyond = fq > (float)a.table_m;
        if (__ballot(beyond)) {  // beyond the table (p == 0 past -104)
            const double dE = (double)(2.0f * fk);
            if (beyond) flip = (dE > T * 104.0) ? false : ((float)st.ru * 0x1.0p-24f < expf_det((float)(-dE / T)));
        }
        return st.live != 0 && flip;
    };
    auto step = [&](const Step &st) {
        int si;
        float fk;
        const bool flip = decide(st, si, fk);
        const unsigned long long heads = 0x0001000100010001ull;  // lane 0 of every row
        const unsigned long long acc = __ballot(flip) & heads;
        // does an accepted update touch a LATER one of the step?  (its site among their columns or their sites)
        unsigned long long hit = 0;
        if (acc & 0x0000000100010001ull) {  // (an accept in the last row touches nobody)
            const int left = st.end - st.beg - 4 * j;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if ((acc >> (16 * q)) & 1ull) {  // wave-uniform
                    const int sq = __builtin_amdgcn_readlane(st.site, 16 * q);
                    bool mine = st.site == sq;
#pragma unroll
                    for (int x = 0; x < 4; ++x) mine = mine || (x < left && st.e[x].x == sq);
                    hit |= __ballot(mine && st.live != 0 && g > q);
                }
            }
        }
        if (hit == 0ull) {
            if (acc) {
                if (flip && j == 0) s[st.site] = (int8_t)(-si);
                asm volatile("" ::: "memory");  // (the next step's gathers are reloads as well)
                float tot = flip ? 2.0f * fk : 0.0f;  // the same in every lane of a row
                tot += dpp_move<DPP_ROW_BCAST15, 0xa>(tot);
                tot += dpp_move<DPP_ROW_BCAST31, 0xc>(tot);
                E += (double)read_lane(tot, 63);
                nacc += (unsigned long long)__builtin_popcountll(acc);
            }
            return;
        }
        // one update at a time: the row's decision against the spins as the earlier rows left them
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            // The spins are re-read from LDS in every pass: another LANE may have flipped one in the pass before.
            // (To the compiler a lane is a thread of its own and this a plain reload of what it just read --
            // it forwarded the old values, and two accepted updates at ONE site of a step went wrong.  LDS
            // operations of a wave execute in order, so no hardware fence is needed, only the reload.)
            asm volatile("" ::: "memory");
            int si2;
            float fk2;
            const bool flip2 = decide(st, si2, fk2);
            if ((__ballot(flip2) >> (16 * q)) & 1ull) {
                if (lane == 16 * q) s[st.site] = (int8_t)(-si2);
                E += (double)(2.0f * read_lane(fk2, 16 * q));
                ++nacc;
            }
        }
    };
    // position of the step `ahead` steps after (k, m)
    auto later = [&](int k, int m, int ahead, int &ko, int &mo) {
        mo = m + ahead;
        ko = k;
        while (mo >= steps) {
            mo -= steps;
            ++ko;
        }
    };
    Step S0, S1, S2;
    stage_sites(S0, 0, 0);
    stage_extents(S0);
    {
        int k1, m1;
        later(0, 0, 1, k1, m1);
        stage_sites(S1, k1, m1);
        stage_extents(S1);
    }
    stage_heads(S0);
    auto step3 = [&](Step &c, Step &n1, Step &n2, int k, int m) {
        int k2, m2;
        later(k, m, 2, k2, m2);
        stage_sites(n2, k2, m2);  // two steps ahead: sites and row extents
        stage_extents(n2);
        stage_heads(n1);          // one step ahead: its extents were requested a step ago
        step(c);
    };
    for (int k = 0; k < a.n_sweeps; ++k) {
        T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        // exp(float32(-dE / T)) for dE = 2 q / table_scale as integer thresholds on the uniform's raw bits
        for (int q = lane; q <= a.table_m; q += 64)
            itab[q] = (unsigned int)__builtin_ceilf(
                expf_det((float)(-((double)(2 * q) / (double)a.table_scale) / T)) * 16777216.0f);
        int m = 0;
        for (; m + 3 <= steps; m += 3) {
            step3(S0, S1, S2, k, m);
            step3(S1, S2, S0, k, m + 1);
            step3(S2, S0, S1, k, m + 2);
        }
        for (; m < steps; ++m) {  // up to two steps left: the stages rotated back into phase
            step3(S0, S1, S2, k, m);
            const Step t = S0;
            S0 = S1;
            S1 = S2;
            S2 = t;
        }
        if (lane == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE && !a.no_best) {  // annealing/gpu_annealer.py:151-153
            bestE = E;
            int4 *dst = reinterpret_cast<int4 *>(a.best_spins + (long long)r * a.sstride);
            const int4 *src = reinterpret_cast<const int4 *>(s);
            for (int i = lane; i < a.sstride / 16; i += 64) dst[i] = src[i];
        }
    }
    {
        int4 *dst = reinterpret_cast<int4 *>(a.spins + (long long)r * a.sstride);
        const int4 *src = reinterpret_cast<const int4 *>(s);
        for (int i = lane; i < a.sstride / 16; i += 64) dst[i] = src[i];
    }
    if (lane == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

// the form applies to: production arguments with the accept table, int8 spins, 32-bit row extents whose
// byte offsets fit 32 bits (the engine checks the row lengths: every row <= 64 entries)
bool sweep_csr_rows4_applies(const SweepArgs &a) {
    const bool lean = csr_args_are_lean(a);
    return a.csr_pair_ahead == 4 && !a.big && a.rowptr && lean && csr_effective_acc(a, lean) == CSR_ACC_F32_TABLE;
}

hipError_t launch_sweep_csr_rows4(const SweepArgs &a, int waves_per_block, hipStream_t st) {
    const hipError_t e = launch_csr_kernel(sweep_csr_rows4_kernel, a, false, false, waves_per_block, st);
    note_sweep_kernel("sweep_csr_rows4_kernel x %d replica(s) per workgroup", waves_per_block);
    return e;
}

}  // namespace sga
