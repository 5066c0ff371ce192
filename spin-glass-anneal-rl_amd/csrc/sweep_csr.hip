// sweep_csr.hip -- launchers of the CSR sweep kernels (sweep_csr_impl.h): the narrow forms (one wave
// per replica, several replicas per workgroup) are built here, the wide forms in sweep_csr_wide*.hip.
#include "sweep_csr_impl.h"

namespace sga {

size_t csr_lds_bytes(int sstride, int table_m, bool bits) { return csr_lds_per_replica(sstride, table_m, bits); }

constexpr size_t CSR_LDS_BUDGET = 160 * 1024 - 256;

int csr_waves_per_block(int sstride, int table_m) {
    const int wpb = (int)(CSR_LDS_BUDGET / csr_lds_per_replica(sstride, table_m));
    return wpb > CSR_WAVES_PER_BLOCK ? CSR_WAVES_PER_BLOCK : wpb;  // 0: does not fit
}

bool csr_big_fits(int sstride, int table_m) {
    // 128 spins = 16 B of bits: keeps the table and the partial-sum slots behind them aligned
    return sstride % 128 == 0 && csr_lds_per_replica(sstride, table_m, true) <= CSR_LDS_BUDGET;
}

int csr_bits_waves_per_block(int sstride, int table_m) {  // narrow bit-spin form
    if (sstride % 128 != 0) return 0;
    const int wpb = (int)(CSR_LDS_BUDGET / csr_lds_per_replica(sstride, table_m, true));
    return wpb > CSR_WAVES_PER_BLOCK ? CSR_WAVES_PER_BLOCK : wpb;
}

template <bool BIG>
static hipError_t launch_csr_narrow(const SweepArgs &a, int waves, hipStream_t st) {
    const bool lean = csr_args_are_lean(a);
    void (*kern)(const SweepArgs) = nullptr;
    switch (csr_effective_acc(a, lean)) {
        case CSR_ACC_F32_TABLE: kern = sweep_csr_kernel<CSR_ACC_F32_TABLE, true, false, BIG>; break;
        case CSR_ACC_F32:
            kern = lean ? sweep_csr_kernel<CSR_ACC_F32, true, false, BIG> : sweep_csr_kernel<CSR_ACC_F32, false, false, BIG>;
            break;
        case CSR_ACC_F64:
            kern = lean ? sweep_csr_kernel<CSR_ACC_F64, true, false, BIG> : sweep_csr_kernel<CSR_ACC_F64, false, false, BIG>;
            break;
        default:
            kern = lean ? sweep_csr_kernel<CSR_ACC_F64_CANON, true, false, BIG>
                        : sweep_csr_kernel<CSR_ACC_F64_CANON, false, false, BIG>;
    }
    return launch_csr_kernel(kern, a, false, BIG, waves, st);
}

// waves_per_replica == 1: several replicas per workgroup (as many as fit LDS), no barriers;
// > 1: one replica per workgroup, its rows dealt to that many waves (1, 2, 4 or 8: the engine rounds
// up).  a.big: the bit-spin forms -- 2 = narrow, 1 = one replica per workgroup (1..8 waves).
hipError_t launch_sweep_csr(const SweepArgs &a0, int waves_per_replica, hipStream_t st) {
    // four | eight updates per step (sweep_csr_rows.hip): its own kernel; every other form reads 4 | 8 as "off"
    if (waves_per_replica == 1 && sweep_csr_rows_applies(a0)) {
        const int wpb = a0.big ? csr_bits_waves_per_block(a0.sstride, a0.table_m) : csr_waves_per_block(a0.sstride, a0.table_m);
        if (wpb < 1) return hipErrorInvalidValue;
        return launch_sweep_csr_rows(a0, wpb, st);
    }
    SweepArgs a = a0;
    if (a.csr_pair_ahead >= 4) a.csr_pair_ahead = 0;
    if (a.big) {
        if (waves_per_replica < 1 || waves_per_replica > CSR_MAX_WIDE || !csr_big_fits(a.sstride, a.table_m))
            return hipErrorInvalidValue;
        if (waves_per_replica == 1 && a.rowptr && a.big == 2) {  // several replicas per workgroup
            const int wpb = csr_bits_waves_per_block(a.sstride, a.table_m);
            if (wpb < 1) return hipErrorInvalidValue;
            return launch_csr_narrow<true>(a, wpb, st);
        }
        if (!a.rowinfo) return hipErrorInvalidValue;  // wide forms read the slotted layout
        return launch_csr_wide_bits(a, waves_per_replica, a.csr_head, st);
    }
    if (waves_per_replica > 1) {
        if (waves_per_replica > CSR_MAX_WIDE || csr_waves_per_block(a.sstride, a.table_m) < 1 || !a.rowinfo)
            return hipErrorInvalidValue;
        return launch_csr_wide_bytes(a, waves_per_replica, st);
    }
    const int wpb = csr_waves_per_block(a.sstride, a.table_m);
    if (wpb < 1 || !a.rowptr) return hipErrorInvalidValue;
    return launch_csr_narrow<false>(a, wpb, st);
}

}  // namespace sga
