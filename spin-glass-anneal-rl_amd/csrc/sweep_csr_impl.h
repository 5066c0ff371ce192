// sweep_csr_impl.h -- single-spin sweeps over CSR couplings: BASELINE configs[2] (10 000 spins, degree
// ~32, 4096 replicas), configs[3] (50 000 spins, degree 598) and configs[4] (TSP, up to 10^6 spins).
//
// Replaces the same reference functions as the dense kernel (core/spin_dynamics.py:73-94,
// core/ising_model.py:176-185) for IsingModelConfig(use_sparse=True) models; the reference's
// own sparse branch (ising_model.py:133-135) raises under the container's torch, so the math
// is the dense path's with the row restricted to its stored entries.
//
// NARROW forms: one replica per wavefront, up to CSR_WAVES_PER_BLOCK independent replicas per
// workgroup (as many as fit LDS), no workgroup barrier anywhere: each wave owns a private LDS slice
// holding its replica's spins (int8, or bits when that keeps more replicas resident).  A row of ~32
// entries is one (column, value) wave-load; the spin gather goes through LDS; the dot is a DPP wave
// sum.  The structure (2.6 MB at C3) is L2-resident, so these forms are bound by instruction issue
// and the dependent-load chain, not by HBM; hence:
//   * the site sequence is known ahead of time (counter RNG): row extents are loaded one PAIR of
//     updates ahead and row entries one update ahead, so no update waits on a rowptr -> entry load;
//   * integer-valued problems accumulate in fp32 (exact) and look the Metropolis probability
//     exp(float32(-dE/T)) of the M possible uphill moves dE = 2k up in a per-sweep LDS table -- the
//     same function of the same arguments, so decisions are bit-identical to the general path.
#pragma once
#include "sweep_common.h"

namespace sga {

constexpr int TAIL_UNROLL = 8;  // wave-loads of a long row kept in flight together
#ifndef CSR_WIDE_ROWS_AHEAD
#define CSR_WIDE_ROWS_AHEAD 1    // wide forms: rows requested this many updates before their reduction
#endif
// Row records of the wide forms through the scalar cache (1) or as a vector load (0).  An LDS wait
// with scalar loads in flight has to be lgkmcnt(0) (they return out of order), so the record
// requested at the top of an update is waited for by that update's first spin gather: same-box A/B
// (profiles/ab3.sh) C4 37.1 vs 34.8 ms per sweep, C5 at 1000 cities 1263-1302 vs 1286-1295 ms ->
// vector loads (in-order vmcnt).  The narrow forms, 4 waves per SIMD, gain 1-2 % from the scalar loads.
// Wide forms with (column, fp32 value) entries: lanes behind the last entry of a row's last slot read the
// zero slot (1) or the padding (0).
#ifndef CSR_TRIM_LAST_SLOT
#define CSR_TRIM_LAST_SLOT 1
#endif
#ifndef CSR_NT_LOADS
#define CSR_NT_LOADS 1  // non-temporal entry loads in the wide builds for very long rows (0: A/B switch)
#endif
#ifndef CSR_WIDE_SCALAR_EXTENTS
#define CSR_WIDE_SCALAR_EXTENTS 0
#endif
constexpr int CSR_MAX_WIDE = 8;   // most waves one replica's row is dealt to (16 measured slower)

// WIDE = one replica per workgroup, its row dealt to NW = 1, 2, 4 or 8 waves in 64-entry SLOTS of
// the padded row layout (slot w + NW q is wave-uniform: scalar addressing, no per-lane bounds
// tests), the per-wave sums meet in LDS with one barrier per update (double-buffered slots, as in
// the dense kernel).  Every wave applies an accepted flip to the shared spin itself before its
// next gather (same value from all waves), so no second barrier is needed.
//
// BIG = the replica's spins sit in LDS as one bit each (1 = spin down; 125 KB at n = 10^6): problems
// beyond the int8 LDS capacity, or whose int8 spins would not keep the whole launch resident; flips
// are idempotent LDS atomics (or / and-not) so that every wave can still apply them itself.
//
// The CANONICAL ORDER of the real-valued row sums whose fp64 sum is not provably exact, which no
// launch geometry changes: entry e of the row (storage order) belongs to lane e % 64 of virtual wave (e / 64) % 8;
// a virtual lane adds its entries in storage order (fp64), each virtual wave folds its 64 lanes by
// the adjacent-pairs tree (wave_sum), and the 8 wave sums are added in order.  A replica dealt to
// NW = 1, 2, 4 or 8 real waves reproduces that exactly with 8 / NW accumulators per lane (NW is a
// template parameter of the wide real-valued builds, so every accumulator index is a constant);
// the CPU checker forms the same sum.
// ACC = how a row sum is formed (CSR_ACC_*, chosen at set time): fp32 where that is exact
// (integer J), with the accept table if h is integer too and the moves are few; fp64 in any order
// where THAT is exact (all J within 53 binary places of each other, row length included -- e.g. the
// TSP distances); the canonical fp64 order otherwise.
// HD = head slots per wave of the wide forms (entries requested ahead and reduced branch-free): the
// engine picks the smallest build that covers the longest row, ceil(max row slots / NW) -- a slot
// that can never hold an entry costs as much as one that does (C4: 5 of 8 -> 50.1 vs 45.1 ms).
// PK = packed entries (a.cvp: one dword per entry = 24-bit column | 8-bit value << 24; integer-valued
// problems with |J| <= 127 and n < 2^24 in the bit-spin wide forms): half the bytes per entry, the row
// sum accumulated as an integer -- the same value, so the same chain (storage variant with its own byte
// model, B = deg * 4 + 8).
template <int ACC, bool LEAN, bool WIDE, bool BIG, int NW = 0, int HD = 8, bool PK = false>
__global__ void __launch_bounds__(64 * (WIDE ? CSR_MAX_WIDE : CSR_WAVES_PER_BLOCK))
    sweep_csr_kernel(const SweepArgs a) {
    constexpr bool FAST = ACC == CSR_ACC_F32_TABLE || ACC == CSR_ACC_F32;  // fp32 accumulation
    constexpr bool TABLE = ACC == CSR_ACC_F32_TABLE;
    constexpr bool CANON = ACC == CSR_ACC_F64_CANON;
    static_assert(!PK || (WIDE && BIG && LEAN && FAST), "packed entries: production bit-spin wide forms of integer problems");
    static_assert(!CANON || !WIDE || NW == 1 || NW == 2 || NW == 4 || NW == 8,
                  "canonical-order wide builds are made per wave count");
    // (the production wide builds are made per wave count as well: with the slot arithmetic and
    // the cross-wave sum on constants the 1000-city instance runs 12 % faster)
    // Row extents: the narrow forms (one wave per replica) index entries (a.rowptr, 32 bit); the
    // wide forms address a row by its 64-entry SLOTS (a.rowinfo: rows are padded to whole slots,
    // pad entries carry the value 0), so a slot number is wave-uniform: no lane tests a bound.
    // 32-bit slot numbers cover nnz >= 2^31 (config 5 at 1000 cities: 63 M slots).  A wave asks
    // for a fixed number of slots per row; the ones past the row's end are redirected to an
    // all-zero slot (rowinfo.z slots from the row's first), so nothing is masked when the row is
    // summed: per slot one scalar select, one VALU op for the lane offset and the load.
    using rp_t = int;
    const rp_t *rowptr = a.rowptr;
    const int rule = LEAN ? SGA_RULE_METROPOLIS : a.rule;  // (a run-time rule costs C3 12 %: issue bound)
    const int arith = LEAN ? SGA_ARITH_F64 : a.arith;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int nw = blockDim.x >> 6;                     // waves in the workgroup
    const int r = WIDE ? (int)blockIdx.x : (int)blockIdx.x * nw + w;
    if (!WIDE && r >= a.R) return;  // wave-uniform; the narrow form has no barriers
    const int n = a.n;
    const int slots = WIDE ? 1 : nw;                    // replicas sharing this workgroup's LDS
    const int me = WIDE ? 0 : w;
    const int stride_lanes = WIDE ? 64 * nw : 64;       // entries between a lane's row elements
    const int first_lane = WIDE ? w * 64 + lane : lane; // this lane's first entry of a row
    const long long sbytes = BIG ? a.sstride / 8 : a.sstride;  // LDS bytes of one replica's spins
    int8_t *s = reinterpret_cast<int8_t *>(smem) + (long long)me * sbytes;
    unsigned int *sbits = reinterpret_cast<unsigned int *>(smem + (long long)me * sbytes);
    float *ptab = reinterpret_cast<float *>(smem + (long long)slots * sbytes) +
                  (long long)me * (a.table_m + 1);
    unsigned int *itab = reinterpret_cast<unsigned int *>(ptab);  // the accept table holds integer thresholds
    double *part = reinterpret_cast<double *>(smem + (long long)slots * sbytes +
                                              sizeof(float) * ((a.table_m + 2) & ~1) * slots);
    int pp = 0;
    const int cstep = WIDE ? (int)blockDim.x : 64, cfirst = WIDE ? tid : lane;
    if constexpr (BIG) {
        spins_to_bits(a.spins + (long long)r * a.sstride, sbits, a.sstride, cfirst, cstep);
    } else {
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (long long)r * a.sstride);
        int4 *dst = reinterpret_cast<int4 *>(s);
        for (int i = cfirst; i < a.sstride / 16; i += cstep) dst[i] = src[i];
    }
    if constexpr (WIDE) __syncthreads();
    // value * spin of column c: with bit spins the sign bit is XORed in (3 VALU ops instead of a
    // compare, a select and a multiply -- the wide forms spend most of their time here)
    auto term = [&](float v, int c) -> float {
        if constexpr (BIG) {
            const unsigned int sign = (sbits[c >> 5] >> (c & 31)) << 31;
            return __int_as_float(__float_as_int(v) ^ (int)sign);
        } else {
            return v * (float)s[c];
        }
    };
    auto spin_i = [&](int c) -> int {
        if constexpr (BIG) return ((sbits[c >> 5] >> (c & 31)) & 1u) ? -1 : 1;
        else return s[c];
    };
    auto store_spins = [&](int8_t *dst_row) {
        if constexpr (BIG) {
            bits_to_spins(sbits, dst_row, a.sstride, n, cfirst, cstep);
        } else {
            int4 *dst = reinterpret_cast<int4 *>(dst_row);
            const int4 *src = reinterpret_cast<const int4 *>(s);
            for (int i = cfirst; i < a.sstride / 16; i += cstep) dst[i] = src[i];
        }
    };
    const bool arith32 = arith == SGA_ARITH_F32;
    double E = a.energy[r], bestE = a.best_energy[r];
    unsigned long long nacc = 0;

    struct Extent {  // what is indexed by the site alone
        rp_t beg, end;  // entries [beg, end) | wide forms: first slot, slot count
        int zrel;       // wide forms: slots from the first slot to an all-zero slot
        int rem;        // wide forms: entries in the row's last slot (1..64)
        float h, d;
    };
    // The entries of a row requested one update ahead: its first 64 (one per lane) in the narrow
    // form, where that is the whole row at degree ~32; HEAD per lane in the wide forms, so that a
    // row of up to HEAD * 64 * waves entries is in flight while the previous update is reduced.
    constexpr int HEAD = WIDE ? HD : 1;
    static_assert(!CANON || !WIDE || HD == TAIL_UNROLL, "canonical order: head = one pass of the 8 virtual waves");
    struct Head {
        int col[HEAD];
        float val[HEAD];
        int len, zrel;            // wide forms: the row's slot count, its zero slot (wave-uniform)
        int rem;                  // wide forms: entries in the row's last slot
        const int2 *row;          // wide forms: the row's first entry (wave-uniform pointer; packed: of a.cvp)
    };
    // wave-uniform value -> SGPR
    auto uniform = [&](int v) -> int { return __builtin_amdgcn_readfirstlane(v); };
    // Narrow forms: what is indexed by the (wave-uniform) site comes through the scalar data cache:
    // s_load into SGPRs -- no address VALU, no vector-memory slot, no readfirstlane when the value
    // is used as a scalar.  (The arrays are written at set time only; the cast to the constant
    // address space is what makes the compiler pick the scalar load.)
    auto sload_i = [&](const int *q) -> int { return *(const __attribute__((address_space(4))) int *)q; };
    auto sload_f = [&](const float *q) -> float { return *(const __attribute__((address_space(4))) float *)q; };
    // this lane's entry of slot `slot` (wave-uniform, < 2^23) of the row starting at `row`
    // (wave-uniform): scalar base + 32-bit lane offset (slot * 512 + lane * 8, one VALU op) = the
    // scalar-base load form
    const unsigned int lane8 = (unsigned int)lane * 8u;
    auto slot_entry = [&](const int2 *row, int slot) -> int2 {
        // the offset stays a 32-bit value opaque to the optimiser (a 64-bit zero extension of
        // it folded into the pointer arithmetic loses the base + zext(VGPR) address form)
        unsigned int off = ((unsigned int)slot << 9) + lane8;
        asm volatile("" : "+v"(off));
        // Rows of 32 slots and more (head slots x waves >= 32: 2000+ entries) belong to structures far
        // beyond the 256 MB Infinity Cache that are streamed once per update and never re-used:
        // their entries are loaded non-temporal (C5 at 1000 cities, same box: 1283-1334 -> 1266 ms per
        // sweep).  Shorter rows keep the default policy -- C4's 239 MB are partly cache served, and
        // for the dense 400 MB matrix non-temporal loads measured 3-5 % slower.
        constexpr bool NT = CSR_NT_LOADS && WIDE && NW * HD >= 32;
        if constexpr (NT) {
            const long long raw = __builtin_nontemporal_load(
                reinterpret_cast<const long long *>(reinterpret_cast<const unsigned char *>(row) + off));
            return make_int2((int)raw, (int)(raw >> 32));
        } else {
            return *reinterpret_cast<const int2 *>(reinterpret_cast<const unsigned char *>(row) + off);
        }
    };
    // (per-lane slot number)
    auto slot_entry_lanes = [&](const int2 *row, int slot_of_lane) -> int2 {
        const unsigned int off = ((unsigned int)slot_of_lane << 9) + lane8;  // (varies per load: nothing to hoist)
        return *reinterpret_cast<const int2 *>(reinterpret_cast<const unsigned char *>(row) + off);
    };
    // the same for packed entries: 256-byte slots, one dword per lane
    auto slot_entry_pk = [&](const int2 *row, int slot) -> int {
        unsigned int off = ((unsigned int)slot << 8) + (lane8 >> 1);
        asm volatile("" : "+v"(off));
        return *reinterpret_cast<const int *>(reinterpret_cast<const unsigned char *>(row) + off);
    };
    // packed entry e times the spin at its column: v * (1 - 2 bit) -- v_bfe_u32, v_lshl_add, [ds_read], v_bfe_i32,
    // v_or, v_mul_i32_i24 (byte 3, sign extended), half a v_add3
    auto term_pk = [&](int e) -> int {
        unsigned int widx = ((unsigned int)e >> 5) & 0x7FFFFu;  // column >> 5: bits 5..23 of the entry (v_bfe_u32)
        asm("" : "+v"(widx));  // (kept apart from the scaling: the LDS base then folds into v_lshl_add)
        const unsigned int word = sbits[widx];
        const int m = __builtin_amdgcn_sbfe(word, (unsigned int)e, 1u);         // -bit (v_bfe_i32 takes offset[4:0])
        return (e >> 24) * ((m << 1) | 1);
    };
    const int nwc = (WIDE && NW > 0) ? NW : nw;  // waves of this replica (a constant in the real-valued wide builds)
    auto load_extent = [&](int site) {
        Extent o;
        const int us = uniform(site);
        if constexpr (WIDE) {
#if CSR_WIDE_SCALAR_EXTENTS
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i ri = *(const __attribute__((address_space(4))) v4i *)(a.rowinfo + us);  // one 16-byte load
#else
            const int4 ri = a.rowinfo[us];  // one 16-byte load
#endif
            o.beg = ri.x;
            o.end = ri.y & 0xFFFFFF;
            o.rem = (int)((unsigned int)ri.y >> 24);
            o.zrel = ri.z;
            o.h = __int_as_float(ri.w);
        } else {
            o.beg = sload_i(rowptr + us);
            o.end = sload_i(rowptr + us + 1);
            o.zrel = 0;
            o.rem = 0;
            o.h = sload_f(a.h + us);
        }
        o.d = arith32 ? sload_f(a.diag + us) : 0.0f;
        return o;
    };
    auto load_head = [&](const Extent &x) {
        Head o;
        o.len = 0;
        o.zrel = 0;
        o.rem = 0;
        o.row = nullptr;
        if constexpr (WIDE) {
            // HEAD slots per wave: slot w + nw q of the row, q = 0..HEAD-1.  The extent arrived
            // updates ago: pin it to SGPRs.  A slot past the row's end reads the zero slot.
            o.len = uniform(x.end);
            o.zrel = uniform(x.zrel);
            if constexpr (PK) {
                o.row = reinterpret_cast<const int2 *>(a.cvp + ((long long)uniform(x.beg) << 6));
#pragma unroll
                for (int q = 0; q < HEAD; ++q) {
                    const int sq = w + nwc * q;
                    o.col[q] = slot_entry_pk(o.row, sq < o.len ? sq : o.zrel);
                    o.val[q] = 0.0f;
                }
                return o;
            }
            o.row = a.cv + ((long long)uniform(x.beg) << 6);
            o.rem = uniform(x.rem);
#pragma unroll
            for (int q = 0; q < HEAD; ++q) {
                const int sq = w + nwc * q;
                // The lanes behind the row's last entry read the zero slot as well (cache resident)
                // instead of the padding: its cache lines are never fetched from HBM.  Not in the
                // builds for rows of 32 slots and more (HD x NW >= 32: under 2 % padding, where the
                // four extra instructions per slot cost more than the lines -- C5 at 1000 cities -0.7 %).
                constexpr bool TRIM = CSR_TRIM_LAST_SLOT && (NW == 0 || NW * HD < 32);
                int2 ent;
                if constexpr (TRIM) {
                    const int eff = sq < o.len ? sq : o.zrel;
                    const int remq = sq == o.len - 1 ? o.rem : 64;
                    ent = slot_entry_lanes(o.row, lane < remq ? eff : o.zrel);
                } else {
                    ent = slot_entry(o.row, sq < o.len ? sq : o.zrel);
                }
                o.col[q] = ent.x;
                o.val[q] = __int_as_float(ent.y);
            }
            return o;
        }
        // one wave: the row's first 64 entries, scalar row base + lane offset, no bounds test (the
        // lanes past the row's end hold entries of the rows behind it -- 64 zeroed entries follow
        // the array -- and are masked when the row is summed)
        static_assert(WIDE || HEAD == 1, "narrow forms: one wave-load ahead");
        const int beg = uniform(x.beg);
        o.len = uniform(x.end) - beg;
        o.row = a.cv + beg;
        const int2 ent = slot_entry(o.row, 0);
        o.col[0] = ent.x;
        o.val[0] = __int_as_float(ent.y);
        return o;
    };

    double T = 1.0;
    auto update = [&](int site, float u, uint32_t ru, const Extent &x, const Head &hd, long long upd) {
        // read s_i before any wave can have applied THIS update's flip (WIDE: before the barrier)
        const int si = spin_i(site);
        // J[site,:].s over the stored entries; products val * (+-1) are exact
        float dot;
        {
            // one code path for the four row-sum forms: fp32 | fp64 accumulators, one per virtual wave
            // of the canonical order (a single one in the exact forms, where the order is free)
            using acc_t = typename std::conditional<PK, int, typename std::conditional<FAST, float, double>::type>::type;
            constexpr int NVA = !CANON ? 1 : (WIDE ? 8 / (NW > 0 ? NW : 8) : 8);
            // (accumulators start from their first term: 0 + x is an instruction the compiler has
            // to keep for x = -0)
            acc_t acc[NVA];
#pragma unroll
            for (int j = 0; j < NVA; ++j) acc[j] = 0;
            if constexpr (PK) {
                acc[0] = term_pk(hd.col[0]);
#pragma unroll
                for (int q = 1; q < HEAD; ++q) acc[0] += term_pk(hd.col[q]);
                for (int s0 = nwc * HEAD; s0 < hd.len; s0 += nwc * TAIL_UNROLL) {
                    int ev[TAIL_UNROLL];
#pragma unroll
                    for (int q = 0; q < TAIL_UNROLL; ++q) {
                        const int sq = s0 + w + nwc * q;
                        ev[q] = slot_entry_pk(hd.row, sq < hd.len ? sq : hd.zrel);
                    }
#pragma unroll
                    for (int q = 0; q < TAIL_UNROLL; ++q) acc[0] += term_pk(ev[q]);
                }
            } else if constexpr (WIDE) {
                // Every head slot is computed; a slot past the row's end holds the zero slot's entries
                // (value 0).  Measured against one wave-uniform branch per slot and against a
                // straight-line block per valid-slot count (same box, profiles/r02_experiments.md): the
                // branch-free form wins on both graded instances -- C4 49.3 vs 57.8 / 55.4 ms per sweep,
                // C5 at 1000 cities 1478 vs 1699 / 1597 ms -- because the LDS gathers of all eight
                // slots stay in flight together.  Virtual wave of slot q: (w + nw q) % 8 -> q % NVA.
#pragma unroll
                for (int q = 0; q < HEAD; ++q) {
                    if (q < NVA) acc[q] = (acc_t)term(hd.val[q], hd.col[q]);
                    else acc[q % NVA] += (acc_t)term(hd.val[q], hd.col[q]);
                }
                // rows beyond HEAD slots per wave (degree > 4096 at 8 waves): eight more slots per pass
                for (int s0 = nwc * HEAD; s0 < hd.len; s0 += nwc * TAIL_UNROLL) {
                    int c[TAIL_UNROLL];
                    float v[TAIL_UNROLL];
#pragma unroll
                    for (int q = 0; q < TAIL_UNROLL; ++q) {
                        const int sq = s0 + w + nwc * q;
                        const int2 ent = slot_entry(hd.row, sq < hd.len ? sq : hd.zrel);
                        c[q] = ent.x;
                        v[q] = __int_as_float(ent.y);
                    }
#pragma unroll
                    for (int q = 0; q < TAIL_UNROLL; ++q) acc[q % NVA] += (acc_t)term(v[q], c[q]);
                }
            } else {
                // one wave: 64-entry chunk q' = 0 of the row is the head, the tail batch t holds the
                // chunks q' = 1 + 8 t + q, i.e. virtual wave (1 + q) % 8 of the canonical order; chunk
                // numbers and the entries left are scalars, a lane past the row's end adds 0
                static_assert(WIDE || (TAIL_UNROLL == 8 && HEAD == 1), "virtual wave of a tail entry = (1 + q) % 8");
                const int len = hd.len;
                acc[0] = (acc_t)term(lane < len ? hd.val[0] : 0.0f, hd.col[0]);
                for (int c0 = 1; 64 * c0 < len; c0 += TAIL_UNROLL) {
                    int c[TAIL_UNROLL];
                    float v[TAIL_UNROLL];
#pragma unroll
                    for (int q = 0; q < TAIL_UNROLL; ++q) {
                        const int left = len - 64 * (c0 + q);  // entries of the row from this chunk on
                        const int2 ent = slot_entry(hd.row, left > 0 ? c0 + q : 0);
                        c[q] = ent.x;
                        v[q] = lane < left ? __int_as_float(ent.y) : 0.0f;
                    }
#pragma unroll
                    for (int q = 0; q < TAIL_UNROLL; ++q) acc[(1 + q) % NVA] += (acc_t)term(v[q], c[q]);
                }
            }
            if constexpr (!CANON) {
                // exact sums (set-time classification): any order, one wave reduction
                acc_t tot = wave_sum(acc[0]);
                if constexpr (WIDE) {
                    acc_t *slot = reinterpret_cast<acc_t *>(part + pp * CSR_MAX_WIDE);
                    if (lane == 0) slot[w] = tot;
                    __syncthreads();
                    acc_t t = slot[0];
                    if constexpr (NW > 0) {
#pragma unroll
                        for (int i = 1; i < NW; ++i) t += slot[i];
                    } else {
                        for (int i = 1; i < nw; ++i) t += slot[i];
                    }
                    tot = t;
                    pp ^= 1;
                }
                dot = (float)tot;  // fp64: rounded to fp32 once (core/ising_model.py:183)
            } else if constexpr (WIDE) {
                // canonical order: one tree per virtual wave, all 8 slots written, summed in order
                double *slot = part + pp * CSR_MAX_WIDE;
#pragma unroll
                for (int j = 0; j < NVA; ++j) {
                    const double sv = wave_sum(acc[j]);  // (+0 when the row has no slot of that wave)
                    if (lane == 0) slot[w + (NW > 0 ? NW : 8) * j] = sv;
                }
                __syncthreads();
                double t = slot[0];
#pragma unroll
                for (int v = 1; v < CSR_MAX_WIDE; ++v) t += slot[v];
                pp ^= 1;
                dot = (float)t;
            } else {
                const int len = hd.len;
                double t = wave_sum(acc[0]);
#pragma unroll
                for (int j = 1; j < NVA; ++j)
                    if (len > 64 * j) t += wave_sum(acc[j]);  // wave-uniform test
                dot = (float)t;
            }
        }
        double dE;
        bool flip;
        if (TABLE && rule == SGA_RULE_METROPOLIS && arith == SGA_ARITH_F64) {
            // core/spin_dynamics.py:131-152 with every quantity an integer: dE = 2k exactly
            // (half-integer fields: table_scale = 2, the table is indexed by 2 fk = dE)
            const float fk = (float)si * (dot + x.h);
            dE = (double)(2.0f * fk);
            const float fq = fk * (float)a.table_scale;
            if (fk <= 0.0f) flip = true;
            else if (fq <= (float)a.table_m) flip = ru < itab[(int)fq];  // u < p, on the uniform's raw bits
            else  // beyond the table (p == 0 past -104); u from its raw bits here: the table builds never form it otherwise
                flip = (dE > T * 104.0) ? false : ((float)ru * 0x1.0p-24f < expf_det((float)(-dE / T)));
        } else {
            flip = metropolis_accept(rule, arith, dot, si, x.h, x.d, T, u, dE);
        }
        if (flip) {
            E += dE;
            ++nacc;
            if (lane == 0) {
                if constexpr (BIG) {  // the same idempotent operation from every wave
                    if (si > 0) atomicOr(&sbits[site >> 5], 1u << (site & 31));
                    else atomicAnd(&sbits[site >> 5], ~(1u << (site & 31)));
                } else {
                    s[site] = (int8_t)(-si);
                }
            }
        }
        if constexpr (!LEAN) {
            if (lane == 0) {
                if (a.accept_trace)
                    a.accept_trace[(long long)r * a.replay_stride + upd] = flip ? 1 : 0;
                if (a.dE_trace)
                    a.dE_trace[(long long)r * a.replay_stride + upd] =
                        flip ? (rule == SGA_RULE_HEAT_BATH ? -dE : dE) : 0.0;
            }
        }
    };

    auto sweep_start = [&](int k) {
        T = a.sched ? a.sched[k * a.sched_ss + r * a.sched_rs] : a.rep_temp[r];
        if constexpr (TABLE) {  // exp(float32(-dE / T)) for dE = 2k, k = 0..M
            if constexpr (WIDE) __syncthreads();  // nobody still reads last sweep's table
            // as integer thresholds on the uniform's 24 raw bits: u = r 2^-24 < p <=> r < ceil(p 2^24)
            // (p 2^24 is exact, so is its ceiling) -- no int -> float conversion of u per update
            for (int q = first_lane; q <= a.table_m; q += stride_lanes)
                itab[q] = (unsigned int)__builtin_ceilf(
                    expf_det((float)(-((double)(2 * q) / (double)a.table_scale) / T)) * 16777216.0f);
            if constexpr (WIDE) __syncthreads();
        }
    };
    auto sweep_end = [&](int k) {
        if (lane == 0 && (!WIDE || w == 0) && a.energy_trace)
            a.energy_trace[(long long)k * a.R + r] = E;
        if (E < bestE && !a.no_best) {  // annealing/gpu_annealer.py:151-153
            bestE = E;
            if constexpr (WIDE) __syncthreads();  // every wave has applied the last flip
            store_spins(a.best_spins + (long long)r * a.sstride);
            if constexpr (WIDE) __syncthreads();
        }
    };

    PairSource<LEAN> rng;
    if constexpr (BIG && WIDE && LEAN) {
        // Rows are requested CSR_WIDE_ROWS_AHEAD updates before they are reduced and their extents
        // NB - 1 updates before (two rings, NB a multiple of NH so that every index is a
        // compile-time constant after unrolling).  The extent of a row has to be back before its
        // entries can be requested; vmcnt retires in order, so the gap NB - NH keeps that wait from
        // draining the rows requested in between.  Same-box A/B of 1 / 2 / 3 rows ahead after the
        // zero-slot change (profiles/ab_wide_r02.sh): C4 34.7 / 35.6 / 35.6 ms per sweep, C5 at
        // 1000 cities (32 GB, HBM resident) 1315 / 1315 / 1318 ms: one row ahead is enough -- an
        // update lasts ~0.7-5 us, several loaded HBM round trips.
        // (A two-update look-ahead -- both rows reduced together, one barrier, the second
        // row sum corrected by -2 J[B][A] s_A when the first flips -- was built, verified against the
        // oracle and measured in round 2: 25 % SLOWER at C4, 16 % at C5-1000, with 128 or 256 VGPRs;
        // see profiles/r02_experiments.md.  The chain, not the barrier, is what an update costs.)
        constexpr int NH = CSR_WIDE_ROWS_AHEAD + 1, NB = 2 * NH;
        struct Pending {
            int site;
            float u;
            uint32_t ru;
            Extent x;
        };
        Pending er[NB];
        Head hr[NH];
        UpdatePair pairP{0, 0, 2.0f, 2.0f};
        int kP = 0, tP = 0;  // producer cursor: next update whose extent is requested
        auto request_extent = [&](Pending &sl) {
            const bool second = tP & 1;
            if (!second) pairP = rng.get(a, r, kP, tP >> 1, kP < a.n_sweeps, lane);  // past the end: site 0
            const int sA = pairP.sA, sB = pairP.sB;  // values first, then select (no scratch)
            const float uA = pairP.uA, uB = pairP.uB;
            const uint32_t rA = pairP.rA, rB = pairP.rB;
            sl.site = second ? sB : sA;
            sl.u = second ? uB : uA;
            sl.ru = second ? rB : rA;
            sl.x = load_extent(sl.site);
            if (++tP == n) {
                tP = 0;
                ++kP;
            }
        };
#pragma unroll
        for (int j = 0; j + 1 < NB; ++j) request_extent(er[j]);
#pragma unroll
        for (int j = 0; j + 1 < NH; ++j) hr[j] = load_head(er[j].x);
        // A sweep = whole groups of NB updates (every ring index a constant), then the < NB left
        // over one at a time with the rings shifted back into phase (register moves, a few per
        // sweep), so that the per-sweep work stays outside the unrolled body.
        for (int k = 0; k < a.n_sweeps; ++k) {
            sweep_start(k);
            int t = 0;
            for (; t + NB <= n; t += NB) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    request_extent(er[(j + NB - 1) % NB]);
                    hr[(j + NH - 1) % NH] = load_head(er[(j + NH - 1) % NB].x);
                    update(er[j].site, er[j].u, er[j].ru, er[j].x, hr[j % NH], (long long)k * n + t + j);
                }
            }
            for (; t < n; ++t) {
                request_extent(er[NB - 1]);
                hr[NH - 1] = load_head(er[NH - 1].x);
                update(er[0].site, er[0].u, er[0].ru, er[0].x, hr[0], (long long)k * n + t);
#pragma unroll
                for (int j = 0; j + 1 < NB; ++j) er[j] = er[j + 1];
#pragma unroll
                for (int j = 0; j + 1 < NH; ++j) hr[j] = hr[j + 1];
            }
            sweep_end(k);
        }
    } else if (!WIDE && LEAN && TABLE && a.csr_pair_ahead) {
        // PAIR LOOK-AHEAD (narrow production form, integer problems, rows of <= 64 entries): the two
        // updates of a Philox pair are reduced TOGETHER against the spins as they stand before the first
        // -- two independent gathers, two interleaved wave sums -- and the chain is replayed on scalars:
        // if A flips, B's row sum is corrected by -2 J[B][A] s_A (the entries of row B at column sA, found
        // by a ballot over the lanes that hold them) and B's own spin is negated when both sit on one
        // site.  Every quantity is an integer below 2^24, so decisions, energies and spins equal the
        // one-at-a-time chain's.  A wave issues dependent instructions several cycles apart and the
        // degree-32 form is bound by exactly that chain: two chains interleaved fill the
        // gaps.  Pipeline: row extents two pairs ahead (scalar loads), row entries one pair ahead.
        struct Stage {
            UpdatePair p;
            Extent xA, xB;
            Head hA, hB;
        };
        const int nb = (n + 1) >> 1;
        auto table_rule = [&](float fk, uint32_t ru, double &dE) -> bool {
            dE = (double)(2.0f * fk);
            const float fq = fk * (float)a.table_scale;
            if (fk <= 0.0f) return true;
            if (fq <= (float)a.table_m) return ru < itab[(int)fq];
            return (dE > T * 104.0) ? false : ((float)ru * 0x1.0p-24f < expf_det((float)(-dE / T)));
        };
        auto stage_extents = [&](Stage &st, int k, int b) {
            st.p = rng.get(a, r, k, b, k < a.n_sweeps, lane);  // past the end: site 0
            st.xA = load_extent(st.p.sA);
            st.xB = load_extent(st.p.sB);
        };
        auto stage_heads = [&](Stage &st) {
            st.hA = load_head(st.xA);
            st.hB = load_head(st.xB);
        };
        auto reduce_pair = [&](const Stage &c, bool hasB) {
            const int sA = c.p.sA, sB = c.p.sB;
            const int siA = spin_i(sA);
            int siB = spin_i(sB);
            const float tA = term(lane < c.hA.len ? c.hA.val[0] : 0.0f, c.hA.col[0]);
            const bool inB = hasB && lane < c.hB.len;
            const float tB = term(inB ? c.hB.val[0] : 0.0f, c.hB.col[0]);
            float dotA = tA, dotB = tB;
            if (a.csr_pair_ahead == 2) {
                wave_sum2(dotA, dotB);  // the two trees step by step in turn (A/B: SGA_CSR_PAIR_AHEAD=2)
            } else {
                dotA = wave_sum(tA);
                dotB = wave_sum(tB);
            }
            double dEA, dEB = 0.0;
            const bool flipA = table_rule((float)siA * (dotA + c.xA.h), c.p.rA, dEA);
            if (flipA) {
                E += dEA;
                ++nacc;
                unsigned long long mm = ballot64(inB && c.hB.col[0] == sA);
                while (mm) {  // (duplicate entries add up; usually no entry at all)
                    dotB -= 2.0f * read_lane(c.hB.val[0], (int)__builtin_ctzll(mm)) * (float)siA;
                    mm &= mm - 1;
                }
                if (sB == sA) siB = -siB;
            }
            bool flipB = false;
            if (hasB) {
                flipB = table_rule((float)siB * (dotB + c.xB.h), c.p.rB, dEB);
                if (flipB) {
                    E += dEB;
                    ++nacc;
                }
            }
            if (lane == 0) {  // in chain order
                if constexpr (BIG) {
                    if (flipA) {
                        if (siA > 0) atomicOr(&sbits[sA >> 5], 1u << (sA & 31));
                        else atomicAnd(&sbits[sA >> 5], ~(1u << (sA & 31)));
                    }
                    if (flipB) {
                        if (siB > 0) atomicOr(&sbits[sB >> 5], 1u << (sB & 31));
                        else atomicAnd(&sbits[sB >> 5], ~(1u << (sB & 31)));
                    }
                } else {
                    if (flipA) s[sA] = (int8_t)(-siA);
                    if (flipB) s[sB] = (int8_t)(-siB);
                }
            }
        };
        // position of the pair `ahead` pairs after (k, b)
        auto later = [&](int k, int b, int ahead, int &ko, int &bo) {
            bo = b + ahead;
            ko = k;
            while (bo >= nb) {
                bo -= nb;
                ++ko;
            }
        };
        Stage S0, S1, S2;
        stage_extents(S0, 0, 0);
        {
            int k1, b1;
            later(0, 0, 1, k1, b1);
            stage_extents(S1, k1, b1);
        }
        stage_heads(S0);
        auto pair3 = [&](Stage &c, Stage &n1, Stage &n2, int k, int b) {
            int k2, b2;
            later(k, b, 2, k2, b2);
            stage_extents(n2, k2, b2);  // two pairs ahead
            stage_heads(n1);            // one pair ahead: its extents were requested a pair ago
            reduce_pair(c, (2 * b + 1) < n);
        };
        for (int k = 0; k < a.n_sweeps; ++k) {
            sweep_start(k);
            int b = 0;
            for (; b + 3 <= nb; b += 3) {
                pair3(S0, S1, S2, k, b);
                pair3(S1, S2, S0, k, b + 1);
                pair3(S2, S0, S1, k, b + 2);
            }
            for (; b < nb; ++b) {  // up to two pairs left: one at a time, the stages rotated back into phase
                pair3(S0, S1, S2, k, b);
                const Stage t = S0;
                S0 = S1;
                S1 = S2;
                S2 = t;
            }
            sweep_end(k);
        }
    } else {
        // Pairs of updates (one Philox block each); the extents of the next pair and the head of its
        // first row are requested while this pair is reduced.  The loop is unrolled over TWO pairs
        // that swap roles (current <-> next): rotating one set of variables cost ~5 of the ~46 VALU
        // instructions per update of the issue-bound degree-32 form in register moves (C3, same box:
        // 4.47 -> 4.14 ms per sweep, 9.17e9 -> 9.9e9 attempts/s).
        struct PairState {
            UpdatePair p;
            Extent xA, xB;
            Head hA;
        };
        const int nb = (n + 1) >> 1;
        PairState S0, S1;
        S0.p = rng.get(a, r, 0, 0, a.n_sweeps > 0, lane);
        S0.xA = load_extent(S0.p.sA);
        S0.xB = load_extent(S0.p.sB);
        S0.hA = load_head(S0.xA);
        auto pair = [&](PairState &c, PairState &nx, int k, int b) {
            const bool last = (b + 1 == nb);
            const int kn = last ? k + 1 : k, bn = last ? 0 : b + 1;
            nx.p = rng.get(a, r, kn, bn, kn < a.n_sweeps, lane);
            const bool hasB = (2 * b + 1) < n;
            nx.xA = load_extent(nx.p.sA);  // a pair ahead
            nx.xB = load_extent(nx.p.sB);
            Head hB{};
            if (hasB) hB = load_head(c.xB);  // in flight while A is reduced
            update(c.p.sA, c.p.uA, c.p.rA, c.xA, c.hA, (long long)k * n + 2 * b);
            nx.hA = load_head(nx.xA);  // in flight while B is reduced
            if (hasB) update(c.p.sB, c.p.uB, c.p.rB, c.xB, hB, (long long)k * n + 2 * b + 1);
        };
        for (int k = 0; k < a.n_sweeps; ++k) {
            sweep_start(k);
            int b = 0;
            for (; b + 1 < nb; b += 2) {
                pair(S0, S1, k, b);
                pair(S1, S0, k, b + 1);
            }
            if (b < nb) {  // an odd number of pairs: one more, then the roles are set straight (once a sweep)
                pair(S0, S1, k, b);
                S0 = S1;
            }
            sweep_end(k);
        }
    }
    if constexpr (WIDE) __syncthreads();
    store_spins(a.spins + (long long)r * a.sstride);
    if (lane == 0 && (!WIDE || w == 0)) {
        a.energy[r] = E;
        a.best_energy[r] = bestE;
        a.n_accepted[r] += nacc;
    }
}


// LDS of one replica (spins as bytes | bits + accept table) and of the cross-wave exchange slots
inline size_t csr_lds_per_replica(int sstride, int table_m, bool big = false) {
    return (size_t)(big ? sstride / 8 : sstride) + sizeof(float) * (size_t)((table_m + 2) & ~1);
}
template <typename K>
inline hipError_t launch_csr_kernel(K kern, const SweepArgs &a, bool wide, bool big, int waves, hipStream_t st) {
    const int slots = wide ? 1 : waves;
    const size_t lds = csr_lds_per_replica(a.sstride, a.table_m, big) * slots + 2 * CSR_MAX_WIDE * sizeof(double);
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int blocks = wide ? a.R : (a.R + waves - 1) / waves;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), lds, st, a);
    note_sweep_kernel("sweep_csr_kernel<acc=%d, %s, %s spins> x %d %s", a.csr_acc, wide ? "one replica per workgroup" : "narrow",
                      big ? "bit" : "int8", waves, wide ? "wave(s)" : "replica(s) per workgroup");
    return hipGetLastError();
}
// the accept table is a specialisation of the production (LEAN) builds
inline int csr_effective_acc(const SweepArgs &a, bool lean) {
    return (a.csr_acc == CSR_ACC_F32_TABLE && (!lean || a.table_m <= 0)) ? CSR_ACC_F32 : a.csr_acc;
}
inline bool csr_args_are_lean(const SweepArgs &a) { return sweep_args_are_lean(a) && a.rule == SGA_RULE_METROPOLIS; }

// wide forms, defined in their own translation units (parallel compilation)
hipError_t launch_csr_wide_bytes(const SweepArgs &a, int waves, hipStream_t st);           // int8 spins
hipError_t launch_csr_wide_bits(const SweepArgs &a, int waves, int head, hipStream_t st);  // bit spins

}  // namespace sga
