// Dense sweep, fp32 couplings, fp32 accumulation (exact when J is integer valued).
#include "sweep_dense_impl.h"
namespace sga {
hipError_t launch_sweep_dense_f32(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    return launch_variant<float, false>(a, waves, cpw, st);
}
}  // namespace sga
