// Cached-local-field sweep, chain-wave form (sweep_clfc_impl.h): instantiations and launcher.
#include "sweep_clfc_impl.h"

namespace sga {

size_t sweep_clfc_lds_bytes(long long ldf, int field_bits, int sstride, int table_m) {
    return clfc_lds_bytes(ldf, field_bits / 8, sstride, table_m);
}

// production arguments only (Philox sites, Metropolis in the reference's fp64 / fp32-exp arithmetic, no per-update
// records): every other mode keeps sweep_clf_kernel
bool sweep_clfc_applies(const SweepArgs &a) {
    return a.clf_chain != 0 && sweep_args_are_lean(a) && a.rule == SGA_RULE_METROPOLIS && a.clf_jmax > 0 &&
           clfc_lds_bytes(a.ldf, a.field_bits / 8, a.sstride, a.table_m) <= 160 * 1024;
}

template <typename JT, typename FT>
static hipError_t launch_clfc(const SweepArgs &a, hipStream_t st) {
    const size_t lds = clfc_lds_bytes(a.ldf, (int)sizeof(FT), a.sstride, a.table_m);
    void (*kern)(const SweepArgs) = sweep_clfc_kernel<JT, FT>;
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.rep_list ? a.rep_count : a.R), dim3(64 * CLFC_WAVES), lds, st, a);
    note_sweep_kernel("sweep_clfc_kernel<%s, %s> (chain wave + %d field waves, <= %d accepts per window)",
                      sizeof(JT) == 4 ? "float" : "int8_t", sizeof(FT) == 2 ? "int16_t" : "int32_t", CLFC_FIELD_WAVES,
                      a.clf_flips);
    return hipGetLastError();
}

hipError_t launch_sweep_clfc(const SweepArgs &a, bool j_is_i8, hipStream_t st) {
    if (!a.fields || (a.field_bits != 16 && a.field_bits != 32) || (a.ldf * (a.field_bits / 8)) % 16 != 0 ||
        a.sstride % 32 != 0 || !sweep_clfc_applies(a))
        return hipErrorInvalidValue;
    if (j_is_i8) return a.field_bits == 16 ? launch_clfc<int8_t, int16_t>(a, st) : launch_clfc<int8_t, int32_t>(a, st);
    return a.field_bits == 16 ? launch_clfc<float, int16_t>(a, st) : launch_clfc<float, int32_t>(a, st);
}

}  // namespace sga
