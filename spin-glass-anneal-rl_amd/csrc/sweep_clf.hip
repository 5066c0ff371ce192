// Cached-local-field sweep (sweep_clf_impl.h): instantiations and launcher.
#include "sweep_clf_impl.h"

namespace sga {

size_t sweep_clf_lds_bytes(long long ldf, int field_bits, int sstride, int table_m) {
    return clf_lds_bytes(ldf, field_bits / 8, sstride, table_m);
}

// waves per replica: enough that a wave asks for its share of a row in one batch of loads, as long as
// every replica stays resident (32 waves per CU)
int sweep_clf_waves(long long ldj, bool j_is_i8, int R, int cus, int forced /* engine option "clf_waves" */) {
    if (forced >= 1) return std::min(forced, CLF_MAX_WAVES);
    const int epc = j_is_i8 ? 1024 : 256;
    const int chunks = (int)((ldj + epc - 1) / epc);
    const int per_cu = std::max(1, (R + std::max(cus, 1) - 1) / std::max(cus, 1));
    // (the kernel holds two row buffers: ~120 VGPRs = 4 waves per SIMD = 16 resident waves per CU -- 6 or 8
    //  waves x 4 workgroups ran as two rounds of workgroups: 13.3 / 10.3 ms against 6.4 for the hot sweep)
    //  fp32 rows (4 x the chunks) measured the other way: 8 waves in two rounds 16.7 / 0.195 ms (hot / cold
    //  sweep) against 19.4 / 0.253 for 4 resident waves that stream their second batch of chunks)
    const int cap = std::max(1, std::min(CLF_MAX_WAVES, (j_is_i8 ? 16 : 32) / per_cu));
    // about three chunks per wave, from {1, 2, 3, 4, 8} (measured at 10 chunks, 1024 replicas: 1 / 2 / 3 / 4 /
    // 6 / 8 waves -> 0.28 / 0.19 / 0.175 / 0.167 / 0.173 / 0.165 ms per cold sweep, 9.3 / 6.6 / 7.2 / 7.1 /
    // 15.0 / 11.4 ms for the first, hot one: profiles/r03_experiments.md)
    const int want = (chunks + 2) / 3;  // (CLF_BATCH_MAX chunks per wave when the cap binds)
    const int pick = want <= 4 ? std::max(want, 1) : 8;
    return std::max(1, std::min(cap, pick));
}

// chunks a wave requests per row in one batch (the kernel is built for 3 and for 5)
int sweep_clf_batch(long long ldj, bool j_is_i8, int waves) {
    const int epc = j_is_i8 ? 1024 : 256;
    const int chunks = (int)((ldj + epc - 1) / epc);
    return (chunks + waves - 1) / waves <= 3 ? 3 : CLF_BATCH_MAX;
}

template <typename JT, typename FT>
static hipError_t launch_clf(const SweepArgs &a, int waves, hipStream_t st) {
    const size_t lds = clf_lds_bytes(a.ldf, (int)sizeof(FT), a.sstride, a.table_m);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const bool lean = sweep_args_are_lean(a) && a.rule == SGA_RULE_METROPOLIS;
    const int batch = sweep_clf_batch(a.ldj, sizeof(JT) == 1, waves);
    const int epc = sizeof(JT) == 1 ? 1024 : 256;
    const bool tail = (int)((a.ldj + epc - 1) / epc) > batch * waves;  // (never with 3 chunks per wave)
    void (*kern)(const SweepArgs) =
        batch == 3 ? (lean ? sweep_clf_kernel<JT, FT, true, 3, false> : sweep_clf_kernel<JT, FT, false, 3, false>)
        : tail     ? (lean ? sweep_clf_kernel<JT, FT, true, CLF_BATCH_MAX, true> : sweep_clf_kernel<JT, FT, false, CLF_BATCH_MAX, true>)
                   : (lean ? sweep_clf_kernel<JT, FT, true, CLF_BATCH_MAX, false>
                           : sweep_clf_kernel<JT, FT, false, CLF_BATCH_MAX, false>);
    hipError_t e = ensure_lds_limit(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.rep_list ? a.rep_count : a.R), dim3(64 * waves), lds, st, a);
    note_sweep_kernel("sweep_clf_kernel<%s, %s, %s, BATCH=%d, TAIL=%d> x %d wave(s)", sizeof(JT) == 4 ? "float" : "int8_t",
                      sizeof(FT) == 2 ? "int16_t" : "int32_t", lean ? "LEAN" : "general", batch, (int)tail, waves);
    return hipGetLastError();
}

hipError_t launch_sweep_clf(const SweepArgs &a, bool j_is_i8, int waves, hipStream_t st) {
    if (waves < 1 || waves > CLF_MAX_WAVES || !a.fields || (a.field_bits != 16 && a.field_bits != 32) ||
        (a.ldf * (a.field_bits / 8)) % 16 != 0 || a.sstride % 32 != 0)
        return hipErrorInvalidValue;
    if (j_is_i8)
        return a.field_bits == 16 ? launch_clf<int8_t, int16_t>(a, waves, st) : launch_clf<int8_t, int32_t>(a, waves, st);
    return a.field_bits == 16 ? launch_clf<float, int16_t>(a, waves, st) : launch_clf<float, int32_t>(a, waves, st);
}

}  // namespace sga
