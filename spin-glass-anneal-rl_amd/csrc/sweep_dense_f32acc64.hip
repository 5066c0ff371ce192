// Dense sweep, fp32 couplings, fp64 accumulation, real-valued J whose fp64 row sums are EXACT (all
// set bits of all J within 53 binary places, carries included: checked at set time): any order.
#include "sweep_dense_impl.h"
namespace sga {
hipError_t launch_sweep_dense_f32acc64(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    return launch_variant<float, true, false>(a, waves, cpw, st);
}
}  // namespace sga
