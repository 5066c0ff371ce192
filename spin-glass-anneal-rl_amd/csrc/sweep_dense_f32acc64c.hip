// Dense sweep, fp32 couplings, fp64 accumulation in the CANONICAL order (real-valued J of any
// dynamic range: the row sum, rounded to fp32 once, does not depend on the launch geometry).
#include "sweep_dense_impl.h"
namespace sga {
hipError_t launch_sweep_dense_f32acc64c(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    return launch_variant<float, true, true>(a, waves, cpw, st);
}
}  // namespace sga
