// sweep_csr_rows.hip -- narrow CSR sweep, SEVERAL UPDATES PER STEP: sparse problems whose longest row has <= 64
// entries (BASELINE configs[2]: 10 000 spins, degree ~32, 4096 replicas; lattices; low-degree graphs) under the
// production arguments (Philox sites, Metropolis, no per-update traces), spins in LDS as int8 or as bits.
//
// Replaces the same reference functions as sweep_csr_kernel (core/spin_dynamics.py:73-94,131-152,
// core/ising_model.py:176-185) and walks the same chain bit for bit.
//
// The one-update-at-a-time form (sweep_csr_impl.h) spends ~65 instructions of one wave on every update and
// uses 32 of its 64 lanes at degree 32; it is bound by instruction issue and its dependent chain, not by
// memory (the 2.6 MB structure is L2 resident).  Here a wave works on the G consecutive updates
// t = G m .. G m + G - 1 at once (G = 4 | 8), one per ROW of 64 / G lanes (lane j of a row holds entries
// EPL j .. EPL j + EPL - 1 of the row's coupling row, EPL = 1 .. 8 picked by the problem's longest row), every
// row sum against the spins as they stand before the first of the G:
//   * sparse couplings make that exact almost always: flipping site A changes the local field of B only
//     if J[B][A] != 0, and B's own spin only if B == A.  The G decisions are formed together, then
//     checked in chain order: for every accepted update the later rows look for its site among their
//     columns (and their own site) -- one compare per entry and a ballot.  No hit (98-99 % of the steps at
//     degree 32 of 10 000): all G decisions are the chain's.  A hit: the step is replayed one update
//     at a time (same data, already in registers).
//   * integer problems (the accept-table builds): everything is an integer below 2^24, so row sums, dE and
//     the energy are exact in any order -- a row's lane 0 keeps its share of the sweep's dE as an integer.
//     Real-valued problems (REAL builds): fp64 row sums in the canonical order of sweep_csr_impl.h, the energy
//     added in chain order.
// Per step: the row entries in EPL / 2 16-byte loads per lane, EPL LDS spin gathers per lane, a DPP row sum
// (all rows in the same instructions), one table look-up.  Sites and uniforms of 128 updates come from one
// vectorised Philox pass (lane l: block l), re-laid so that lane i holds update i (ds_bpermute); a step
// picks its G updates with one more permute each.  Sites, row extents and row entries are requested
// two / one steps ahead (the site sequence is known from the counter RNG).
#include <type_traits>

#include "sweep_csr_impl.h"

namespace sga {

// (One replica per workgroup -- the spin slice at LDS address 0, a gather taking the column as its address --
// measured SLOWER: C3 2.10 vs 1.53 ms per sweep at four updates per step.)
// EPL = entries per lane: a row of the wave covers coupling rows of up to (64 / G) * EPL entries; the engine
// picks the smallest build that covers the problem's longest row (a slot that can never hold an entry costs as
// much as one that does): lattices and other low-degree graphs run with one entry per lane.
// BIG: the replica's spins as one bit each in LDS (1 = down), for problems whose int8 spins would not fit or
// would leave few replicas resident -- the narrow bit-spin form's layout (a.big == 2).
// REAL: couplings or fields that are not (half-)integers -- no accept table, row sums in fp64 in the CANONICAL
// order of sweep_csr_impl.h (entry e in lane e of a 64-lane adjacent-pairs tree; lanes past the row's end add
// +-0): a lane's EPL consecutive entries are the leaves of one subtree, folded locally by adjacent pairs, and the
// DPP steps over the row's lanes continue the same tree (IEEE addition commutes, so which side a partner comes
// from does not matter) -- the same bits as the one-update form for every launch geometry.  The energy is
// then added in chain order.
// MODE: 0 = accept table, moves beyond the table computed | 1 = REAL | 2 = accept table that holds EVERY move the problem can
// propose (a.table_covers: C3, C4 -- the beyond-the-table test and its exp path are not compiled in)
template <int G, int EPL, bool BIG, int MODE>
__global__ void __launch_bounds__(64 * CSR_WAVES_PER_BLOCK) sweep_csr_rows_kernel(const SweepArgs a) {
    constexpr bool REAL = MODE == 1, COVERS = MODE == 2;
    constexpr int LPR = 64 / G;   // lanes per row
    // (rows of 65 ... 256 entries: G = 4 with 8 | 16 entries per lane, accept-table builds only -- the canonical
    //  order of real-valued sums is defined on 64-entry virtual waves, which a lane's 16 consecutive entries straddle)
    static_assert((G == 4 || G == 8) && LPR * EPL <= 256 && (!REAL || LPR * EPL <= 64),
                  "rows of 16 or 8 lanes; coupling rows of up to 256 entries (64 for real-valued sums)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int nw = blockDim.x >> 6;
    const int r = (int)blockIdx.x * nw + w;
    if (r >= a.R) return;  // wave-uniform; no barriers in this kernel
    const int n = a.n;
    // LDS as in the narrow form of sweep_csr_kernel: [nw] spin slices, then [nw] accept tables
    const long long sbytes = BIG ? a.sstride / 8 : a.sstride;  // LDS bytes of one replica's spins
    int8_t *s = reinterpret_cast<int8_t *>(smem) + (long long)w * sbytes;
    unsigned int *sbits = reinterpret_cast<unsigned int *>(smem + (long long)w * sbytes);
    unsigned int *itab = reinterpret_cast<unsigned int *>(smem + (long long)nw * sbytes) + (long long)w * (a.table_m + 1);
    auto load_spins = [&]() {
        if constexpr (BIG) {
            spins_to_bits(a.spins + (long long)r * a.sstride, sbits, a.sstride, lane, 64);
        } else {
            const int4 *src = reinterpret_cast<const int4 *>(a.spins + (long long)r * a.sstride);
            int4 *dst = reinterpret_cast<int4 *>(s);
            for (int i = lane; i < a.sstride / 16; i += 64) dst[i] = src[i];
        }
    };
    auto store_spins = [&](int8_t *dst_row) {
        if constexpr (BIG) {
            bits_to_spins(sbits, dst_row, a.sstride, n, lane, 64);
        } else {
            int4 *dst = reinterpret_cast<int4 *>(dst_row);
            const int4 *src = reinterpret_cast<const int4 *>(s);
            for (int i = lane; i < a.sstride / 16; i += 64) dst[i] = src[i];
        }
    };
    load_spins();
    // spin at a site; value * spin of a column (bit spins: the spin bit XORed into the value's sign bit)
    auto spin_at = [&](int c) -> int {
        if constexpr (BIG) return ((sbits[c >> 5] >> (c & 31)) & 1u) ? -1 : 1;
        else return s[c];
    };
    auto flip_at = [&](int c, int si_old) {  // one lane per site (distinct sites; words may be shared)
        if constexpr (BIG) atomicXor(&sbits[c >> 5], 1u << (c & 31));
        else s[c] = (int8_t)(-si_old);
    };
    const int g = lane / LPR, j = lane % LPR;  // row of the wave = update of the step, lane in the row
#if defined(ROWS_SENS_VALU2) || defined(ROWS_SENS_VALU4) || defined(ROWS_SENS_SALU) || defined(ROWS_SENS_LDS)
    unsigned int sens_v = (unsigned int)lane, sens_s = 0u;
    unsigned int sens_a = (unsigned int)((long long)w * sbytes + ((lane * 37) % 997));  // (random-ish bytes of the wave's own slice)
#endif
    double E = a.energy[r], bestE = a.best_energy[r];
    unsigned long long nacc = 0;
    double T = 1.0;
    double dE_lane = 0.0;  // this lane's share of the sweep's energy change (an integer: exact; see step())
    const int steps = (n + G - 1) / G;  // steps per sweep (the last one may hold fewer than G updates)

    struct Step {
        int site;       // this row's site
        uint32_t ru;    // its uniform's 24 raw bits
        int live;       // the update exists (t < n, sweep < n_sweeps)
        int beg, end;   // row extent
        float h;
        int2 e[EPL];    // entries beg + EPL j + q
    };
    // 128 consecutive updates of a sweep: lane i holds update 128 W + i (lo) and 128 W + 64 + i (hi)
    int ws_lo = 0, ws_hi = 0, wu_lo = 0, wu_hi = 0;
    auto permute = [&](int from_lane, int v) -> int { return __builtin_amdgcn_ds_bpermute(from_lane << 2, v); };
    auto refill = [&](int k, int window) {
        // the stream of sweep_common.h: block b serves updates 2 b (words x, y) and 2 b + 1 (words z, w)
        const u32x4 x = philox4x32_10((uint32_t)(64 * window + lane), a.sweep0 + (uint32_t)k, a.replica0 + (uint32_t)r,
                                      DOMAIN_SWEEP, a.seed_lo, a.seed_hi);
        const int sA = (int)word_to_site(x.x, (uint32_t)n), sB = (int)word_to_site(x.z, (uint32_t)n);
        const int uA = (int)(x.y >> 8), uB = (int)(x.w >> 8);
        const bool odd = lane & 1;
        const int lo = lane >> 1, hi = 32 + (lane >> 1);
        const int a0 = permute(lo, sA), b0 = permute(lo, sB), a1 = permute(hi, sA), b1 = permute(hi, sB);
        const int c0 = permute(lo, uA), d0 = permute(lo, uB), c1 = permute(hi, uA), d1 = permute(hi, uB);
        ws_lo = odd ? b0 : a0;
        ws_hi = odd ? b1 : a1;
        wu_lo = odd ? d0 : c0;
        wu_hi = odd ? d1 : c1;
    };
    // the sites of step m of sweep k (past the last sweep every row is dead, site 0)
    auto stage_sites = [&](Step &st, int k, int m) {
        const bool valid = k < a.n_sweeps;  // wave-uniform
        const int t0 = G * m, p = t0 & 127;
        if (valid && p == 0) refill(k, t0 >> 7);
        const int from = (p & 63) + g;
        const int sv = permute(from, p < 64 ? ws_lo : ws_hi), uv = permute(from, p < 64 ? wu_lo : wu_hi);
        st.live = (valid && t0 + g < n) ? 1 : 0;
        st.site = st.live ? sv : 0;
        // (a dead row never flips: table builds give it a uniform above every table entry, so that the accept compare
        //  alone is the row's decision -- 24-bit uniforms, entries <= 2^24)
        st.ru = (REAL || st.live) ? (uint32_t)uv : 0xFFFFFFFFu;
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
        // (256 zeroed entries follow the array -- CSR_TAIL_PAD, sga_engine.cpp: a row of the wave reaches up to
        //  LPR * EPL <= 256 entries past its first one; lanes past the row's end read what lies behind it)
        unsigned int off = (unsigned int)(st.beg + EPL * j) * 8u;
        asm volatile("" : "+v"(off));
        const unsigned char *cv = reinterpret_cast<const unsigned char *>(a.cv);
#pragma unroll
        for (int q = 0; q < EPL; ++q) st.e[q] = *reinterpret_cast<const int2 *>(cv + off + 8 * q);
    };
    // sum over the lanes of a row, in every lane of the row (table form: exact, integers below 2^24; REAL: the
    // upper levels of the canonical tree)
    auto row_sum = [&](auto v) {
        v += dpp_move<DPP_QUAD_XOR1>(v);
        v += dpp_move<DPP_QUAD_XOR2>(v);
        v += dpp_move<DPP_ROW_HALF_MIRROR>(v);
        if constexpr (LPR == 16) v += dpp_move<DPP_ROW_MIRROR>(v);
        return v;
    };
    // this row's decision against the spins as they stand: flips?, dE of the move (table form: 2 fk, fk = s_i
    // (row sum + h), an integer)
    // (the mask of a wave-wide predicate, straight from the compare: __ballot() materialises the bool in a VGPR first)
    auto ballot = [](bool p) -> unsigned long long { return __builtin_amdgcn_ballot_w64(p); };
    // dE of a move: REAL builds a double; table builds 2 fk as a float (an integer below 2^25: exact), widened only
    // where a move is accepted
    using DE = typename std::conditional<REAL, double, float>::type;
    auto decide = [&](const Step &st, int &si, DE &dE) -> bool {
        const int left = st.end - st.beg - EPL * j;  // entries of the row from this lane's first on
        si = spin_at(st.site);
        if constexpr (REAL) {
            double tr[EPL];
#pragma unroll
            for (int q = 0; q < EPL; ++q) {
                const float v = q < left ? __int_as_float(st.e[q].y) : 0.0f;
                tr[q] = (double)(v * (float)spin_at(st.e[q].x));  // (exact product)
            }
#pragma unroll
            for (int stride = 1; stride < EPL; stride *= 2)
#pragma unroll
                for (int q = 0; q + stride < EPL; q += 2 * stride) tr[q] += tr[q + stride];
            const float dotr = (float)row_sum(tr[0]);  // rounded to fp32 once (core/ising_model.py:183)
            const float u = (float)st.ru * 0x1.0p-24f;
            double dEr;
            const bool flipr = metropolis_accept(SGA_RULE_METROPOLIS, SGA_ARITH_F64, dotr, si, st.h, 0.0f, T, u, dEr);
            dE = (DE)dEr;
            return st.live != 0 && flipr;
        }
        float fk;
        float dot;
        if constexpr (BIG) {
            auto term = [&](int q) -> float {
                const int c = st.e[q].x;
                const unsigned int sign = (sbits[c >> 5] >> (c & 31)) << 31;
                return __int_as_float((q < left ? st.e[q].y : 0) ^ (int)sign);
            };
            dot = term(0);
#pragma unroll
            for (int q = 1; q < EPL; ++q) dot += term(q);
        } else {
            dot = (0 < left ? __int_as_float(st.e[0].y) : 0.0f) * (float)s[st.e[0].x];
#pragma unroll
            for (int q = 1; q < EPL; ++q) {
                const float v = q < left ? __int_as_float(st.e[q].y) : 0.0f;
                dot = __builtin_fmaf(v, (float)s[st.e[q].x], dot);  // (exact either way: small integers)
            }
        }
        dot = row_sum(dot);
        // core/spin_dynamics.py:131-152 with every quantity an integer: dE = 2 fk exactly (half-integer
        // fields: table_scale = 2, the table is indexed by 2 fk = dE); sweep_csr_impl.h, TABLE branch
        fk = (float)si * (dot + st.h);
        const float fq = fk * (float)a.table_scale;
        const int idx = COVERS ? max((int)fq, 0) : min(max((int)fq, 0), a.table_m);  // (COVERS: no fq beyond the table)
        // u < p on the uniform's raw bits.  Entry 0 (fk <= 0: downhill or flat, p = 1) holds 2^24, above every 24-bit
        // uniform: those moves are accepted by the same compare, no branch around the look-up
        bool flip = st.ru < itab[idx];
        dE = (DE)(2.0f * fk);
        if constexpr (!COVERS) {
            const bool beyond = fq > (float)a.table_m;
            if (ballot(beyond)) {  // beyond the table (p == 0 past -104)
                const double dEd = (double)(2.0f * fk);
                if (beyond) flip = (dEd > T * 104.0) ? false : ((float)st.ru * 0x1.0p-24f < expf_det((float)(-dEd / T)));
            }
        }
        if constexpr (COVERS) return flip;  // (st.live is in st.ru: the compare above is the whole decision)
        return st.live != 0 && flip;
    };
    constexpr unsigned long long HEADS = G == 4 ? 0x0001000100010001ull : 0x0101010101010101ull;  // lane 0 of every row
    auto step = [&](const Step &st) {
        int si;
        DE dE;
#ifdef ROWS_SENS_VALU2   // sensitivity builds (profiles/r05_experiments.md 2): N more instructions of one class per step
#pragma unroll
        for (int z = 0; z < ROWS_SENS_VALU2; ++z) asm volatile("v_add_u32 %0, %0, %0" : "+v"(sens_v));
#endif
#ifdef ROWS_SENS_VALU4
#pragma unroll
        for (int z = 0; z < ROWS_SENS_VALU4; ++z) asm volatile("v_bfe_u32 %0, %0, 1, 7" : "+v"(sens_v));
#endif
#ifdef ROWS_SENS_SALU
#pragma unroll
        for (int z = 0; z < ROWS_SENS_SALU; ++z) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sens_s) : : "scc");
#endif
#ifdef ROWS_SENS_LDS
#pragma unroll
        for (int z = 0; z < ROWS_SENS_LDS; ++z) { int t; asm volatile("ds_read_i8 %0, %1" : "=v"(t) : "v"(sens_a)); asm volatile("" : : "v"(t)); }
#endif
        const bool flip = decide(st, si, dE);
        const unsigned long long acc = ballot(flip) & HEADS;
        // does an accepted update touch a LATER one of the step?  (its site among their columns or their
        // sites; entries past a row's end take part -- a stray hit costs a replay, nothing else)
        unsigned long long hit = 0;
        if (acc & (HEADS >> LPR)) {  // (an accept in the last row touches nobody)
#pragma unroll
            for (int q = 0; q + 1 < G; ++q) {
                if ((acc >> (LPR * q)) & 1ull) {  // wave-uniform
                    const int sq = __builtin_amdgcn_readlane(st.site, LPR * q);
                    bool mine = st.site == sq;
#pragma unroll
                    for (int x = 0; x < EPL; ++x) mine = mine || st.e[x].x == sq;
                    hit |= ballot(mine && st.live != 0 && g > q);
                }
            }
        }
        if (hit == 0ull) {
            if (acc) {
                if (flip && j == 0) flip_at(st.site, si);
                asm volatile("" ::: "memory");  // (the next step's gathers are reloads as well)
                if constexpr (REAL) {
                    unsigned long long order = acc;
                    while (order) {  // chain order
                        const int q = (int)__builtin_ctzll(order);
                        order &= order - 1;
                        E += read_lane((double)dE, q);
                    }
                } else {
                    // dE = 2 fk is an integer: lane 0 of a row keeps the sum of its accepted moves, added to E
                    // at the end of the sweep (exact in any order, far below 2^53; a wave sum per step cost ~14
                    // instructions)
                    dE_lane += (flip && j == 0) ? (double)dE : 0.0;
                }
                nacc += (unsigned long long)__builtin_popcountll(acc);
            }
            return;
        }
        // one update at a time: the row's decision against the spins as the earlier rows left them
#pragma unroll 1
        for (int q = 0; q < G; ++q) {
            // The spins are re-read from LDS in every pass: another LANE may have flipped one in the pass before.
            // (To the compiler a lane is a thread of its own and this a plain reload of what it just read --
            // it forwarded the old values, and two accepted updates at ONE site of a step went wrong.  LDS
            // operations of a wave execute in order, so no hardware fence is needed, only the reload.)
            asm volatile("" ::: "memory");
            int si2;
            DE dE2;
            const bool flip2 = decide(st, si2, dE2);
            if ((ballot(flip2) >> (LPR * q)) & 1ull) {
                if (lane == LPR * q) flip_at(st.site, si2);
                E += read_lane((double)dE2, LPR * q);
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
        if constexpr (!REAL)
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
        E += wave_sum(dE_lane);
        dE_lane = 0.0;
        if (lane == 0 && a.energy_trace) a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE && !a.no_best) {  // annealing/gpu_annealer.py:151-153
            bestE = E;
            store_spins(a.best_spins + (long long)r * a.sstride);
        }
    }
    store_spins(a.spins + (long long)r * a.sstride);
    if (lane == 0) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}

// the form applies to: production arguments (Philox sites, Metropolis, no per-update traces), int8 spins or the
// narrow bit-spin layout, 32-bit row extents whose byte offsets fit 32 bits; the engine checks the row lengths
// (every row <= 64 entries: a.csr_row_cap) and sets a.csr_pair_ahead = 4 | 8
static bool rows_table_form(const SweepArgs &a) {  // integer problems (moves beyond the table's 2048 entries: computed)
    return csr_effective_acc(a, true) == CSR_ACC_F32_TABLE;
}
bool sweep_csr_rows_applies(const SweepArgs &a) {
    if (!((a.csr_pair_ahead == 4 || a.csr_pair_ahead == 8) && a.csr_row_cap >= 1 && a.csr_row_cap <= 256 &&
          (a.big == 0 || a.big == 2) && a.rowptr && csr_args_are_lean(a)))
        return false;
    return a.csr_row_cap <= 64 || (a.csr_pair_ahead == 4 && rows_table_form(a));  // longer rows: integer problems, four per step
}

template <bool BIG, int MODE>
static hipError_t launch_rows(const SweepArgs &a, int waves_per_block, hipStream_t st) {
    constexpr bool REAL = MODE == 1;
    const int cap = a.csr_row_cap;  // entries of the problem's longest row (<= 64)
    void (*kern)(const SweepArgs) = nullptr;
    int g = 4, epl = 4;
    if (a.csr_pair_ahead == 8) {  // rows of 8 lanes
        g = 8;
        epl = cap <= 8 ? 1 : cap <= 16 ? 2 : cap <= 32 ? 4 : 8;
        kern = epl == 1 ? sweep_csr_rows_kernel<8, 1, BIG, MODE> : epl == 2 ? sweep_csr_rows_kernel<8, 2, BIG, MODE>
             : epl == 4 ? sweep_csr_rows_kernel<8, 4, BIG, MODE> : sweep_csr_rows_kernel<8, 8, BIG, MODE>;
    } else {                      // rows of 16 lanes
        epl = cap <= 16 ? 1 : cap <= 32 ? 2 : cap <= 64 ? 4 : cap <= 128 ? 8 : 16;
        kern = epl == 1 ? sweep_csr_rows_kernel<4, 1, BIG, MODE> : epl == 2 ? sweep_csr_rows_kernel<4, 2, BIG, MODE>
             : epl == 4 ? sweep_csr_rows_kernel<4, 4, BIG, MODE> : nullptr;
        if constexpr (!REAL) {
            if (epl == 8) kern = sweep_csr_rows_kernel<4, 8, BIG, MODE>;
            if (epl == 16) kern = sweep_csr_rows_kernel<4, 16, BIG, MODE>;
        }
        if (!kern) return hipErrorInvalidValue;
    }
    const hipError_t e = launch_csr_kernel(kern, a, false, BIG, waves_per_block, st);
    note_sweep_kernel("sweep_csr_rows_kernel<%d rows, %d entries per lane, %s spins, %s> x %d replica(s) per workgroup", g, epl,
                      BIG ? "bit" : "int8", REAL ? "fp64 canonical sums" : "accept table", waves_per_block);
    return e;
}

hipError_t launch_sweep_csr_rows(const SweepArgs &a, int waves_per_block, hipStream_t st) {
    // Everything but the accept-table class -- real-valued couplings, integer problems with larger sums -- runs the
    // fp64 builds (exact for integers too).  Table builds: with or without the beyond-the-table path (a.table_covers).
    const bool table = rows_table_form(a);
    if (table && a.table_covers) return a.big ? launch_rows<true, 2>(a, waves_per_block, st) : launch_rows<false, 2>(a, waves_per_block, st);
    if (a.big) return table ? launch_rows<true, 0>(a, waves_per_block, st) : launch_rows<true, 1>(a, waves_per_block, st);
    return table ? launch_rows<false, 0>(a, waves_per_block, st) : launch_rows<false, 1>(a, waves_per_block, st);
}

}  // namespace sga
