// sweep_csr_wide_bits.hip -- picks the head-slot build of the bit-spin wide forms.
#include "sweep_csr_impl.h"

namespace sga {
hipError_t launch_csr_wide_bits_h2(const SweepArgs &, int, hipStream_t);
hipError_t launch_csr_wide_bits_h3(const SweepArgs &, int, hipStream_t);
hipError_t launch_csr_wide_bits_h4(const SweepArgs &, int, hipStream_t);
hipError_t launch_csr_wide_bits_h5(const SweepArgs &, int, hipStream_t);
hipError_t launch_csr_wide_bits_h6(const SweepArgs &, int, hipStream_t);
hipError_t launch_csr_wide_bits_h7(const SweepArgs &, int, hipStream_t);
hipError_t launch_csr_wide_bits_h8(const SweepArgs &, int, hipStream_t);
hipError_t launch_csr_wide_bits_h10(const SweepArgs &, int, hipStream_t);

// head = slots per wave that the longest row needs (builds for 2 ... 8 and 10; beyond that the tail loop
// takes the rest);
// the canonical-order and the traced builds always keep eight
hipError_t launch_csr_wide_bits(const SweepArgs &a, int waves, int head, hipStream_t st) {
    const bool fixed8 = !csr_args_are_lean(a) || a.csr_acc == CSR_ACC_F64_CANON;
    if (!fixed8 && (head == 9 || head == 10)) return launch_csr_wide_bits_h10(a, waves, st);
    if (fixed8 || head <= 0 || head > 7) return launch_csr_wide_bits_h8(a, waves, st);
    switch (head) {
        case 1:
        case 2: return launch_csr_wide_bits_h2(a, waves, st);
        case 3: return launch_csr_wide_bits_h3(a, waves, st);
        case 4: return launch_csr_wide_bits_h4(a, waves, st);
        case 5: return launch_csr_wide_bits_h5(a, waves, st);
        case 6: return launch_csr_wide_bits_h6(a, waves, st);
        default: return launch_csr_wide_bits_h7(a, waves, st);
    }
}
}  // namespace sga
