// Cached-local-field sweep, several accepts per round (sweep_clfb_impl.h): instantiations and launcher.
#include "sweep_clfb_impl.h"

namespace sga {

size_t sweep_clfb_lds_bytes(long long ldf, int field_bits, int sstride, int table_m) {
    return clfb_lds_bytes(ldf, field_bits / 8, sstride, table_m);
}

// production arguments only (Philox sites, Metropolis through the accept table, no per-update records), matrices below
// 4 GiB (32-bit row offsets in the check): everything else keeps sweep_clf_kernel
bool sweep_clfb_applies(const SweepArgs &a, bool j_is_i8) {
    return a.clf_batched != 0 && sweep_args_are_lean(a) && a.rule == SGA_RULE_METROPOLIS &&
           (unsigned long long)a.n * (unsigned long long)a.ldj * (j_is_i8 ? 1ull : 4ull) < (1ull << 32) &&
           clfb_lds_bytes(a.ldf, a.field_bits / 8, a.sstride, a.table_m) <= 160 * 1024;
}

template <typename JT, typename FT>
static hipError_t launch_clfb(const SweepArgs &a, int waves, hipStream_t st) {
    const size_t lds = clfb_lds_bytes(a.ldf, (int)sizeof(FT), a.sstride, a.table_m);
    const int batch = sweep_clf_batch(a.ldj, sizeof(JT) == 1, waves);
    const int epc = sizeof(JT) == 1 ? 1024 : 256;
    const bool tail = (int)((a.ldj + epc - 1) / epc) > batch * waves;  // (never with 3 chunks per wave)
    void (*kern)(const SweepArgs) = batch == 3 ? sweep_clfb_kernel<JT, FT, 3, false>
                                    : tail     ? sweep_clfb_kernel<JT, FT, CLF_BATCH_MAX, true>
                                               : sweep_clfb_kernel<JT, FT, CLF_BATCH_MAX, false>;
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.rep_list ? a.rep_count : a.R), dim3(64 * waves), lds, st, a);
    note_sweep_kernel("sweep_clfb_kernel<%s, %s, BATCH=%d, TAIL=%d> x %d wave(s), <= %d accepts per round",
                      sizeof(JT) == 4 ? "float" : "int8_t", sizeof(FT) == 2 ? "int16_t" : "int32_t", batch, (int)tail, waves,
                      CLFB_LIST);
    return hipGetLastError();
}

hipError_t launch_sweep_clfb(const SweepArgs &a, bool j_is_i8, int waves, hipStream_t st) {
    if (waves < 1 || waves > CLF_MAX_WAVES || !a.fields || (a.field_bits != 16 && a.field_bits != 32) ||
        (a.ldf * (a.field_bits / 8)) % 16 != 0 || a.sstride % 32 != 0 || !sweep_clfb_applies(a, j_is_i8))
        return hipErrorInvalidValue;
    if (j_is_i8)
        return a.field_bits == 16 ? launch_clfb<int8_t, int16_t>(a, waves, st) : launch_clfb<int8_t, int32_t>(a, waves, st);
    return a.field_bits == 16 ? launch_clfb<float, int16_t>(a, waves, st) : launch_clfb<float, int32_t>(a, waves, st);
}

}  // namespace sga
