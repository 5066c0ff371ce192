// Dense sweep, fp32 couplings, fp64 accumulation (general real-valued J: the row sum is
// rounded to fp32 once, independent of the launch geometry).
#include "sweep_dense_impl.h"
namespace sga {
hipError_t launch_sweep_dense_f32acc64(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    return launch_variant<float, true>(a, waves, cpw, st);
}
}  // namespace sga
