// Dense sweep, ternary couplings (J in {-1, 0, +1}) held as two bit-planes: 2 bits per coupling,
// row dot = nnz_i - 2 * popcount(nz & (sign ^ spin_bits)).  Production (LEAN) configuration.
#include "sweep_dense_impl.h"
namespace sga {
hipError_t launch_sweep_dense_t2(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    if (waves < 1 || waves > MAX_WAVES || cpw < 0 || cpw > T2_MAX_CPW) return hipErrorInvalidValue;
    switch (cpw) {
        case 0: return launch_one<Tern2, false, 0>(a, waves, st);  // streaming form
        case 1: return launch_one<Tern2, false, 1>(a, waves, st);
        case 2: return launch_one<Tern2, false, 2>(a, waves, st);
        case 3: return launch_one<Tern2, false, 3>(a, waves, st);
        default: return launch_one<Tern2, false, 4>(a, waves, st);
    }
}
}  // namespace sga
