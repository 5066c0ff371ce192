// sweep_csr_wide.hip -- wide CSR forms with int8 spins in LDS (one replica per workgroup, its row dealt
// to 2, 4 or 8 waves); kernel in sweep_csr_impl.h, launch table in sweep_csr_wide_launch.h.
#include "sweep_csr_wide_launch.h"

namespace sga {
hipError_t launch_csr_wide_bytes(const SweepArgs &a, int waves, hipStream_t st) {
    return launch_wide<false, 8>(a, waves, st);
}
}  // namespace sga
