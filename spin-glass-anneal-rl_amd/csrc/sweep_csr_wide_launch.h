// sweep_csr_wide_launch.h -- which instantiation of sweep_csr_kernel (sweep_csr_impl.h) a launch of the
// wide CSR forms takes: one replica per workgroup, its row dealt to 1, 2, 4 or 8 waves.  Shared by the
// translation units that build them -- sweep_csr_wide.hip (int8 spins) and one sweep_csr_wide_bits<N>.hip
// per head-slot count N (bit spins; one unit each so that they compile in parallel) -- so that a change
// of the launch table is ONE edit.
#pragma once
#include "sweep_csr_impl.h"

namespace sga {

#define SGA_WIDE_PICK(ACC_, NW_, HD_)                                                                 \
    (lean ? sweep_csr_kernel<ACC_, true, true, BIG, NW_, HD_>                                         \
          : sweep_csr_kernel<ACC_, false, true, BIG, (ACC_ == CSR_ACC_F64_CANON ? NW_ : 0), 8>)
#define SGA_WIDE_NW(ACC_, HD_)                                                                        \
    (waves == 1 ? SGA_WIDE_PICK(ACC_, 1, HD_) : waves == 2 ? SGA_WIDE_PICK(ACC_, 2, HD_)              \
     : waves == 4 ? SGA_WIDE_PICK(ACC_, 4, HD_) : SGA_WIDE_PICK(ACC_, 8, HD_))
#define SGA_WIDE_TABLE(HD_)                                                                           \
    (waves == 1 ? sweep_csr_kernel<CSR_ACC_F32_TABLE, true, true, BIG, 1, HD_>                        \
     : waves == 2 ? sweep_csr_kernel<CSR_ACC_F32_TABLE, true, true, BIG, 2, HD_>                      \
     : waves == 4 ? sweep_csr_kernel<CSR_ACC_F32_TABLE, true, true, BIG, 4, HD_>                      \
                  : sweep_csr_kernel<CSR_ACC_F32_TABLE, true, true, BIG, 8, HD_>)
#define SGA_WIDE_PACKED(ACC_, HD_)                                                                     \
    (waves == 1 ? sweep_csr_kernel<ACC_, true, true, true, 1, HD_, true>                              \
     : waves == 2 ? sweep_csr_kernel<ACC_, true, true, true, 2, HD_, true>                            \
     : waves == 4 ? sweep_csr_kernel<ACC_, true, true, true, 4, HD_, true>                            \
                  : sweep_csr_kernel<ACC_, true, true, true, 8, HD_, true>)
// the production builds and the canonical-order builds are made per wave count (1, 2, 4, 8); the other
// traced builds take it at run time
template <bool BIG, int HD>
static hipError_t launch_wide(const SweepArgs &a, int waves, hipStream_t st) {
    if (waves != 1 && waves != 2 && waves != 4 && waves != 8) return hipErrorInvalidValue;
    const bool lean = csr_args_are_lean(a);
    void (*kern)(const SweepArgs) = nullptr;
    if constexpr (BIG) {  // packed entries (a.cvp): integer problems, production builds
        if (lean && a.cvp) {
            const int acc = csr_effective_acc(a, lean);
            if (acc == CSR_ACC_F32_TABLE) return launch_csr_kernel(SGA_WIDE_PACKED(CSR_ACC_F32_TABLE, HD), a, true, BIG, waves, st);
            if (acc == CSR_ACC_F32) return launch_csr_kernel(SGA_WIDE_PACKED(CSR_ACC_F32, HD), a, true, BIG, waves, st);
        }
    }
    switch (csr_effective_acc(a, lean)) {
        case CSR_ACC_F32_TABLE: kern = SGA_WIDE_TABLE(HD); break;
        case CSR_ACC_F32: kern = SGA_WIDE_NW(CSR_ACC_F32, HD); break;
        case CSR_ACC_F64: kern = SGA_WIDE_NW(CSR_ACC_F64, HD); break;
        default: kern = SGA_WIDE_NW(CSR_ACC_F64_CANON, 8); break;  // (always eight head slots)
    }
    return launch_csr_kernel(kern, a, true, BIG, waves, st);
}

}  // namespace sga
