/* Plain-C user of the boundary (include/sga.h): what a reference-side FFI would bind.  Built by
 * tests/test_c_abi.py with gcc -std=c99; runs one small problem through create / set_dense /
 * init_replicas / set_ladder / sweep / exchange / get_best and prints the result. */
#include <stdio.h>
#include <stdlib.h>

#include "sga.h"

#define N 96
#define R 8

int main(void) {
    static float J[N * N], h[N];
    static int8_t best[N];
    double ladder[R], energies[R], energies_cached[R], e_best = 0.0, e_best_cached = 0.0;
    char what[512];
    uint64_t checksum = 0, checksum2 = 0;
    unsigned int x = 12345u;
    int i, j, rc, swaps = 0, r_best = -1;
    sga_engine *eng = NULL;
    for (i = 0; i < N; ++i) {
        h[i] = 0.0f;
        for (j = i + 1; j < N; ++j) {
            x = x * 1664525u + 1013904223u;
            J[i * N + j] = J[j * N + i] = (x >> 16) & 1u ? 1.0f : -1.0f;
        }
        J[i * N + i] = 0.0f;
    }
    for (i = 0; i < R; ++i) ladder[i] = 10.0 / (1.0 + 3.0 * i);
    rc = sga_create(0, &eng);
    if (rc != SGA_OK) {
        printf("NO_DEVICE %d %s\n", rc, sga_last_error());
        return rc == SGA_ERR_DEVICE ? 3 : 1;
    }
    if (sga_set_dense(eng, J, N, h, N, SGA_J_AUTO) != SGA_OK || sga_init_replicas(eng, R, R, 0, 42u, NULL) != SGA_OK ||
        sga_set_ladder(eng, ladder, 1) != SGA_OK) {
        printf("SETUP_FAILED %s\n", sga_last_error());
        return 1;
    }
    for (i = 0; i < 20; ++i) {
        if (sga_sweep(eng, 10, SGA_SITE_RANDOM, SGA_ARITH_F64, NULL, 0, 0, NULL, NULL, NULL, NULL, NULL) != SGA_OK ||
            sga_exchange(eng, NULL, NULL, NULL, &swaps) != SGA_OK) {
            printf("RUN_FAILED %s\n", sga_last_error());
            return 1;
        }
    }
    if (sga_get_energies(eng, energies) != SGA_OK || sga_get_best(eng, -1, &e_best, best, &r_best) != SGA_OK) return 1;
    {
        double check = 0.0; /* H = -1/2 s J s - h s, recomputed here in C */
        for (i = 0; i < N; ++i)
            for (j = 0; j < N; ++j) check -= 0.5 * J[i * N + j] * best[i] * best[j];
        printf("OK version=%d best=%.1f recomputed=%.1f replica=%d\n", sga_version(), e_best, check, r_best);
        if (check != e_best) return 2;
    }
    /* the same run through the cached-local-field sweep (sga_set_field_cache): the identical chain */
    if (sga_problem_checksum(eng, &checksum) != SGA_OK) return 1;
    if (sga_set_field_cache(eng, SGA_FIELD_CACHE_ON) != SGA_OK || sga_init_replicas(eng, R, R, 0, 42u, NULL) != SGA_OK ||
        sga_set_ladder(eng, ladder, 1) != SGA_OK) {
        printf("SETUP_FAILED %s\n", sga_last_error());
        return 1;
    }
    for (i = 0; i < 20; ++i) {
        if (sga_sweep(eng, 10, SGA_SITE_RANDOM, SGA_ARITH_F64, NULL, 0, 0, NULL, NULL, NULL, NULL, NULL) != SGA_OK ||
            sga_exchange(eng, NULL, NULL, NULL, &swaps) != SGA_OK) {
            printf("RUN_FAILED %s\n", sga_last_error());
            return 1;
        }
    }
    if (sga_get_energies(eng, energies_cached) != SGA_OK || sga_get_best(eng, -1, &e_best_cached, NULL, NULL) != SGA_OK ||
        sga_last_kernel(what, (int)sizeof(what)) != SGA_OK || sga_problem_checksum(eng, &checksum2) != SGA_OK)
        return 1;
    for (i = 0; i < R; ++i)
        if (energies_cached[i] != energies[i]) {
            printf("CACHED_FIELDS_DIFFER replica %d: %.1f vs %.1f\n", i, energies_cached[i], energies[i]);
            return 2;
        }
    printf("OK cached-fields best=%.1f kernel=%s checksum=%016llx\n", e_best_cached, what, (unsigned long long)checksum);
    if (e_best_cached != e_best || checksum != checksum2) return 2;
    sga_destroy(eng);
    return 0;
}
