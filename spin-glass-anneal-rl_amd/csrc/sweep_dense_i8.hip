// Dense sweep, int8 couplings (integer J in [-127,127]), v_dot4_i32_i8 accumulation.
#include "sweep_dense_impl.h"
namespace sga {
hipError_t launch_sweep_dense_i8(const SweepArgs &a, int waves, int cpw, hipStream_t st) {
    return launch_variant<int8_t, false>(a, waves, cpw, st);
}
hipError_t launch_sweep_dense_f32(const SweepArgs &, int, int, hipStream_t);
hipError_t launch_sweep_dense_f32acc64(const SweepArgs &, int, int, hipStream_t);
hipError_t launch_sweep_dense_f32acc64c(const SweepArgs &, int, int, hipStream_t);

// acc64: 0 = fp32 accumulation (exact: integer J), 1 = fp64 in any order (exact: set-time scan),
// 2 = fp64 in the canonical order
hipError_t launch_sweep_dense(const SweepArgs &a, bool j_is_i8, int acc64, int waves, int cpw,
                              hipStream_t st) {
    if (waves < 1 || waves > MAX_WAVES || cpw < 0 || cpw > MAX_CPW) return hipErrorInvalidValue;
    if (j_is_i8) return launch_sweep_dense_i8(a, waves, cpw, st);
    return acc64 == 2 ? launch_sweep_dense_f32acc64c(a, waves, cpw, st)
         : acc64 == 1 ? launch_sweep_dense_f32acc64(a, waves, cpw, st)
                      : launch_sweep_dense_f32(a, waves, cpw, st);
}
int dense_look_ahead(bool t2, bool j_is_i8, bool acc64, int cpw, int waves, int R) {
    if (acc64 || cpw < 1) return 1;
    const int top = t2 ? 2 : 4;  // = has_look_ahead<JT, ACC64, CPW>() and launch_one's test
    if (cpw > (t2 ? 4 : 6)) return 1;
    if (cpw > top) return 2;
    if (cpw == top && (waves > 4 || (long long)R * waves > 3 * 1024)) return 1;
    return LOOK;
}
size_t sweep_dense_lds_bytes(long long ld, int table_m, bool acc64) {
    if (acc64)  // fp32 couplings, real valued: + the per-chunk sums of the canonical order
        return (size_t)(dense_canon_offset(ld, table_m) + dense_canon_bytes(ld, 1024));
    return (size_t)ld + DENSE_LDS_EXTRA + sizeof(float) * (size_t)(table_m + 1);
}
}  // namespace sga
