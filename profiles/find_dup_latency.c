/* Host-to-host latency of tvz_find_duplicates through the plain C ABI (no Python in the loop):
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude profiles/find_dup_latency.c \
 *       -Ltvidz_amd -ltvz -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/tvidz_amd -Wl,-rpath,/opt/rocm/lib
 * configs[2] shape: ONE query of ~200 cuts against C videos of ~200 cuts on 24/25/30 fps grids. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tvz.h"

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp_d(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }
static unsigned long long rng = 88172645463325252ULL;
static unsigned next_u32(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 16); }

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 5000, L = 200, REPS = argc > 2 ? atoi(argv[2]) : 3000;
    const int fps[3] = {24, 25, 30};
    int32_t *ids = malloc(sizeof(int32_t) * C);
    int64_t *offs = malloc(sizeof(int64_t) * (C + 1));
    double *keys = malloc(sizeof(double) * (size_t)C * L);
    for (int c = 0; c < C; c++) {
        ids[c] = c + 1;
        offs[c] = (int64_t)c * L;
        const int f = fps[next_u32() % 3], nfr = f * (600 + next_u32() % 6600);
        for (int i = 0; i < L; i++) keys[(size_t)c * L + i] = (double)(1 + next_u32() % (nfr - 1)) / f;
    }
    offs[C] = (int64_t)C * L;
    tvz_corpus *corp = NULL;
    if (tvz_corpus_create(&corp, 0) || tvz_corpus_upload(corp, ids, offs, keys, C, (int64_t)C * L)) {
        fprintf(stderr, "setup failed: %s\n", tvz_last_error());
        return 1;
    }
    double q[200];
    const int f = 30, nfr = 30 * 3600;
    for (int i = 0; i < L; i++) q[i] = (double)(1 + next_u32() % (nfr - 1)) / f;
    int32_t *oid = malloc(4 * C), *ocnt = malloc(4 * C), *okth = malloc(4 * C);
    int64_t n = 0;
    double *t = malloc(sizeof(double) * REPS);
    for (int r = 0; r < REPS + 50; r++) {
        const double t0 = now_us();
        if (tvz_find_duplicates(corp, q, L, 2, -1, C, oid, ocnt, okth, &n)) { fprintf(stderr, "%s\n", tvz_last_error()); return 1; }
        if (r >= 50) t[r - 50] = now_us() - t0;
    }
    qsort(t, REPS, sizeof(double), cmp_d);
    printf("{\"C\": %d, \"query_len\": %d, \"hits\": %lld, \"find_duplicates_us\": {\"p10\": %.1f, \"median\": %.1f, \"p90\": %.1f, \"min\": %.1f}",
           C, L, (long long)n, t[REPS / 10], t[REPS / 2], t[REPS * 9 / 10], t[0]);
    /* the streaming driver's call shape (app.py:234-237): the upload's own row was just upserted - it
     * sits in the delta table, so the call is index lookup + delta sweep - and is excluded by id */
    if (tvz_corpus_upsert(corp, C + 1, q, 50)) { fprintf(stderr, "%s\n", tvz_last_error()); return 1; }
    for (int r = 0; r < REPS + 50; r++) {
        const double t0 = now_us();
        if (tvz_find_duplicates(corp, q, L, 2, C + 1, C, oid, ocnt, okth, &n)) { fprintf(stderr, "%s\n", tvz_last_error()); return 1; }
        if (r >= 50) t[r - 50] = now_us() - t0;
    }
    qsort(t, REPS, sizeof(double), cmp_d);
    printf(", \"with_own_row_in_delta_us\": {\"p10\": %.1f, \"median\": %.1f, \"p90\": %.1f}, \"hits_excluding_self\": %lld}\n",
           t[REPS / 10], t[REPS / 2], t[REPS * 9 / 10], (long long)n);
    tvz_corpus_destroy(corp);
    return 0;
}
