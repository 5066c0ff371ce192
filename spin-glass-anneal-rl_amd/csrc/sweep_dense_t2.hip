// Dense sweep, ternary couplings (J in {-1, 0, +1}) held as two bit-planes: 2 bits per coupling,
// row dot = nnz_i - 2 * popcount(nz & (sign ^ spin_bits)).  Production (LEAN) configuration.
#include "sweep_dense_impl.h"
namespace sga {
hipError_t launch_sweep_dense_t2(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    if (waves < 1 || waves > MAX_WAVES || cpw < 0 || cpw > MAX_CPW) return hipErrorInvalidValue;
    return launch_variant<Tern2, false>(a, waves, cpw, st);
}
}  // namespace sga
