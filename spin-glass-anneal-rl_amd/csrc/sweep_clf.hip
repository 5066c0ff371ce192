// Cached-local-field sweep (sweep_clf_impl.h): instantiations and launcher.
#include "sweep_clf_impl.h"

namespace sga {

size_t sweep_clf_lds_bytes(long long ldf, int field_bits, int sstride, int table_m) {
    return clf_lds_bytes(ldf, field_bits / 8, sstride, table_m);
}

// waves per replica: enough that a wave asks for its share of a row in one batch of loads
int sweep_clf_waves(long long ldj, bool j_is_i8) {
    static const int forced = std::getenv("SGA_CLF_WAVES") ? std::atoi(std::getenv("SGA_CLF_WAVES")) : 0;  // A/B switch
    if (forced >= 1 && forced <= MAX_WAVES) return forced;
    const int epc = j_is_i8 ? 1024 : 256;
    const int chunks = (int)((ldj + epc - 1) / epc);
    return std::max(1, std::min(MAX_WAVES, (chunks + CLF_BATCH - 1) / CLF_BATCH));
}

template <typename JT, typename FT>
static hipError_t launch_clf(const SweepArgs &a, int waves, hipStream_t st) {
    const size_t lds = clf_lds_bytes(a.ldf, (int)sizeof(FT), a.sstride, a.table_m);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const bool lean = sweep_args_are_lean(a) && a.rule == SGA_RULE_METROPOLIS;
    void (*kern)(const SweepArgs) = lean ? sweep_clf_kernel<JT, FT, true> : sweep_clf_kernel<JT, FT, false>;
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.R), dim3(64 * waves), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_sweep_clf(const SweepArgs &a, bool j_is_i8, int waves, hipStream_t st) {
    if (waves < 1 || waves > MAX_WAVES || !a.fields || (a.field_bits != 16 && a.field_bits != 32) ||
        (a.ldf * (a.field_bits / 8)) % 16 != 0 || a.sstride % 32 != 0)
        return hipErrorInvalidValue;
    if (j_is_i8)
        return a.field_bits == 16 ? launch_clf<int8_t, int16_t>(a, waves, st) : launch_clf<int8_t, int32_t>(a, waves, st);
    return a.field_bits == 16 ? launch_clf<float, int16_t>(a, waves, st) : launch_clf<float, int32_t>(a, waves, st);
}

}  // namespace sga
