/* Lookups while the index is rebuilt (no Python, no GIL in the picture): one thread calls
 * tvz_find_duplicates in a loop and times every call, the main thread upserts enough new videos to
 * trigger several background rebuilds of the inverted index.  The reference's readers never wait
 * for writers (inspector/db.py:83 under Postgres MVCC); here a rebuild fills a shadow generation
 * and swaps it in, so no lookup may stall for a rebuild either.
 * Prints one JSON line; exit code 1 on a wrong result.  Built and run by tests/test_c_abi_gpu.py:
 *   gcc -O1 -pthread -Iinclude tests/rebuild_latency.c -Ltvidz_amd -ltvz -lm */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tvz.h"

#define CHECK(x) do { int _rc = (x); if (_rc != 0) { fprintf(stderr, "FAIL %s -> %d: %s\n", #x, _rc, tvz_last_error()); exit(1); } } while (0)

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static uint64_t rng_state = 88172645463325252ULL;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

enum { ROWS = 100000, LEN = 40, NEW_ROWS = 40000, PROBE = 30, MAXLAT = 1 << 20 };

static tvz_corpus *corpus;
static double probe[PROBE];
static volatile int stop_flag;
static volatile int64_t builds_now;
static float *lat;              /* us per call */
static int64_t *lat_builds;     /* builds finished when the call returned */
static int n_lat, wrong;
static double t_start, *lat_at;   /* when each call returned (us since start) */

static void *finder(void *arg) {
    (void)arg;
    int32_t ids[64], cnt[64], kth[64];
    while (!stop_flag && n_lat < MAXLAT) {
        int64_t n = -1;
        const double t0 = now_us();
        CHECK(tvz_find_duplicates(corpus, probe, PROBE, 5, -1, 64, ids, cnt, kth, &n));
        const double t1 = now_us();
        if (!(n == 3 && ids[0] == 17 && ids[1] == 1234 && ids[2] == 49999 && cnt[0] >= 12 && cnt[2] >= 12)) wrong++;
        lat[n_lat] = (float)(t1 - t0);
        lat_builds[n_lat] = builds_now;
        lat_at[n_lat] = t1 - t_start;
        n_lat++;
    }
    return NULL;
}

static int cmp_f(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return x < y ? -1 : x > y; }

int main(void) {
    CHECK(tvz_corpus_create(&corpus, 0));
    /* ROWS videos of LEN cuts on a 0.1 s grid below 2000 s; three of them share 12 cuts with the probe */
    int32_t *ids = malloc(ROWS * sizeof *ids);
    int64_t *offs = malloc((ROWS + 1) * sizeof *offs);
    double *keys = malloc((size_t)ROWS * LEN * sizeof *keys);
    for (int i = 0; i < PROBE; ++i) probe[i] = (double)(1 + (rnd() % 19999)) / 10.0 + 0.05;   /* off the rows' grid */
    for (int r = 0; r < ROWS; ++r) {
        ids[r] = r + 1;
        offs[r] = (int64_t)r * LEN;
        for (int j = 0; j < LEN; ++j) keys[(size_t)r * LEN + j] = (double)(1 + (rnd() % 19999)) / 10.0;
        if (r + 1 == 17 || r + 1 == 1234 || r + 1 == 49999)
            for (int j = 0; j < 12; ++j) keys[(size_t)r * LEN + j] = probe[j];
    }
    offs[ROWS] = (int64_t)ROWS * LEN;
    CHECK(tvz_corpus_reserve(corpus, ROWS + NEW_ROWS + 1024, (int64_t)(ROWS + NEW_ROWS + 1024) * LEN));
    CHECK(tvz_corpus_upload(corpus, ids, offs, keys, ROWS, (int64_t)ROWS * LEN));
    lat = malloc(MAXLAT * sizeof *lat);
    lat_builds = malloc(MAXLAT * sizeof *lat_builds);
    lat_at = malloc(MAXLAT * sizeof *lat_at);
    t_start = now_us();
    int64_t b0 = 0;
    CHECK(tvz_corpus_index_stats(corpus, NULL, NULL, NULL, NULL, &b0));
    builds_now = b0;
    pthread_t th;
    pthread_create(&th, NULL, finder, NULL);
    /* quiet phase: what a lookup costs with nothing else going on */
    struct timespec nap = {0, 200 * 1000 * 1000};
    nanosleep(&nap, NULL);
    const int n_quiet = n_lat;
    /* upserts far from the probe's keys (> 2500 s): none of them can become a hit */
    double row[LEN], max_upsert_us = 0;
    enum { MAXWIN = 512 };
    static double win0[MAXWIN], win1[MAXWIN];   /* the upsert calls that rebuilt the index: [start, end] since t_start */
    int n_win = 0;
    for (int v = 0; v < NEW_ROWS; ++v) {
        for (int j = 0; j < LEN; ++j) row[j] = 2500.0 + (double)(rnd() % 100000) / 7.0;
        const double t0 = now_us();
        CHECK(tvz_corpus_upsert(corpus, 1000000 + v, row, LEN));
        const double dt = now_us() - t0;
        if (dt > max_upsert_us) max_upsert_us = dt;
        if (dt > 500.0) fprintf(stderr, "slow upsert (rebuild?) #%d: %.0f us, ended at %.0f us\n", v, dt, now_us() - t_start);
        /* a service's rate, not a tight loop: upserts are ordered on the mutation stream and a lookup
         * waits for those that returned before it (read-your-writes), so a backlog of thousands of
         * queued upserts would show up as lookup latency that has nothing to do with rebuilds */
        struct timespec gap = {0, 20 * 1000};
        nanosleep(&gap, NULL);
        int64_t b = 0;
        CHECK(tvz_corpus_index_stats(corpus, NULL, NULL, NULL, NULL, &b));
        if (b != builds_now && n_win < MAXWIN) { win0[n_win] = t0 - t_start; win1[n_win] = t0 - t_start + dt; n_win++; }
        builds_now = b;
    }
    stop_flag = 1;
    pthread_join(th, NULL);
    int64_t n_ix = 0, n_delta = 0, b1 = 0;
    CHECK(tvz_corpus_index_stats(corpus, &n_ix, &n_delta, NULL, NULL, &b1));
    const int n_busy = n_lat - n_quiet;
    /* lookups that STARTED AND FINISHED while a rebuild was running: none if readers waited for rebuilds,
     * dozens if they do not (robust against the occasional multi-millisecond hiccup of a shared host,
     * which hits lookups and upserts alike whether or not a rebuild is running) */
    int inside = 0, min_inside = 1 << 30;
    double rebuild_us = 0;
    for (int w = 0; w < n_win; ++w) {
        int k = 0;
        for (int i = n_quiet; i < n_lat; ++i)
            if (lat_at[i] - lat[i] >= win0[w] && lat_at[i] <= win1[w]) k++;
        inside += k;
        if (k < min_inside) min_inside = k;
        rebuild_us += win1[w] - win0[w];
    }
    /* every lookup is either OVERLAPPING a rebuild window (it ran, at least partly, while the upsert call that
     * rebuilt the index was in progress) or OUTSIDE all of them: the two maxima are reported apart, so that a
     * slow lookup can be attributed - a reader stalled by a rebuild shows up in the first, a hiccup of the
     * shared host in either */
    float max_in = 0.f, max_out = 0.f;
    int n_overlap = 0;
    for (int i = n_quiet; i < n_lat; ++i) {
        const double a = lat_at[i] - lat[i], b = lat_at[i];
        int in = 0;
        for (int w = 0; w < n_win && !in; ++w) in = a < win1[w] && b > win0[w];
        if (in) { n_overlap++; if (lat[i] > max_in) max_in = lat[i]; }
        else if (lat[i] > max_out) max_out = lat[i];
        if (lat[i] > 500.f) fprintf(stderr, "slow lookup: %.0f us, returned at %.0f us (builds then: %lld, %s a rebuild window)\n",
                                    lat[i], lat_at[i], (long long)lat_builds[i], in ? "overlapping" : "outside");
    }
    float *q = malloc(n_quiet * sizeof *q), *w = malloc((n_busy > 0 ? n_busy : 1) * sizeof *w);
    memcpy(q, lat, n_quiet * sizeof *q);
    memcpy(w, lat + n_quiet, n_busy * sizeof *w);
    qsort(q, n_quiet, sizeof *q, cmp_f);
    qsort(w, n_busy, sizeof *w, cmp_f);
    printf("{\"rows\": %d, \"upserts\": %d, \"rebuilds\": %lld, \"indexed_rows\": %lld, \"delta_rows\": %lld, "
           "\"lookups_quiet\": %d, \"quiet_median_us\": %.1f, \"quiet_max_us\": %.1f, "
           "\"lookups_during_upserts\": %d, \"median_us\": %.1f, \"p99_us\": %.1f, \"p999_us\": %.1f, \"max_us\": %.1f, "
           "\"max_upsert_us\": %.1f, \"rebuild_calls_us_mean\": %.1f, \"lookups_completed_inside_rebuilds\": %d, "
           "\"min_lookups_inside_one_rebuild\": %d, \"rebuild_windows\": %d, \"lookups_overlapping_rebuilds\": %d, "
           "\"max_us_inside_rebuild\": %.1f, \"max_us_outside\": %.1f, \"wrong_results\": %d}\n",
           ROWS, NEW_ROWS, (long long)(b1 - b0), (long long)n_ix, (long long)n_delta, n_quiet, q[n_quiet / 2],
           q[n_quiet - 1], n_busy, n_busy ? w[n_busy / 2] : 0.f, n_busy ? w[(int)(n_busy * 0.99)] : 0.f,
           n_busy ? w[(int)(n_busy * 0.999)] : 0.f, n_busy ? w[n_busy - 1] : 0.f, max_upsert_us,
           n_win ? rebuild_us / n_win : 0.0, inside, n_win ? min_inside : 0, n_win, n_overlap, max_in, max_out, wrong);
    CHECK(tvz_corpus_destroy(corpus));
    return wrong ? 1 : 0;
}
