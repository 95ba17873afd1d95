/* The sharded pipeline of profiles/shard_trace.py driven from plain C (no Python, no torch): `depth` batches in
 * flight, each on its own stream, one tvz_match_sharded call per batch (lookup with the top-k -> ncclAllGather over a
 * one-rank communicator -> merge), the host waiting for the oldest batch's event.  Prints the pace and the host's
 * share.  The corpus and the queries come from a file written by profiles/shard_pipeline.sh (bench.py's synthetic
 * corpus, rank 0's 1/N shard):
 *   int64 n_rows, n_keys, Q, n_qkeys, max_len | int32 ids[n_rows] | int64 offs[n_rows+1] | double keys[n_keys]
 *   | int64 qoffs[Q+1] | double qkeys[n_qkeys]
 *   ./shard_pipeline <file> [depth=3] [steps=200] */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "tvz.h"

#define CHECK(x) do { int _rc = (x); if (_rc != 0) { fprintf(stderr, "FAIL %s -> %d: %s\n", #x, _rc, tvz_last_error()); return 1; } } while (0)
#define HIPCHECK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "FAIL %s: %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)
#define MAXD 8

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <file> [depth] [steps]\n", argv[0]); return 2; }
    const int depth = argc > 2 ? atoi(argv[2]) : 3, steps = argc > 3 ? atoi(argv[3]) : 200;
    const int shape = argc > 4 ? (int)strtol(argv[4], NULL, 0) : 0;     /* TVZ_ALGO_WAVE 0x400 / TVZ_ALGO_NO_WAVE 0x800 / ... */
    if (depth < 1 || depth > MAXD) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int64_t h[5];
    if (fread(h, 8, 5, f) != 5) return 1;
    const int64_t n_rows = h[0], n_keys = h[1], Q = h[2], n_qk = h[3], max_len = h[4];
    int32_t *ids = malloc(4 * n_rows);
    int64_t *offs = malloc(8 * (n_rows + 1)), *qoffs = malloc(8 * (Q + 1));
    double *keys = malloc(8 * n_keys), *qkeys = malloc(8 * n_qk);
    if (fread(ids, 4, n_rows, f) != (size_t)n_rows || fread(offs, 8, n_rows + 1, f) != (size_t)n_rows + 1 ||
        fread(keys, 8, n_keys, f) != (size_t)n_keys || fread(qoffs, 8, Q + 1, f) != (size_t)Q + 1 ||
        fread(qkeys, 8, n_qk, f) != (size_t)n_qk) { fprintf(stderr, "short file\n"); return 1; }
    fclose(f);
    tvz_corpus *c = NULL;
    CHECK(tvz_corpus_create(&c, 0));
    CHECK(tvz_corpus_upload(c, ids, offs, keys, n_rows, n_keys));
    unsigned char uid[TVZ_UNIQUE_ID_BYTES];
    tvz_comm *comm = NULL;
    CHECK(tvz_comm_unique_id(uid));
    CHECK(tvz_comm_init(&comm, uid, 1, 0, 0));
    enum { K = 16, CAP = 16384 };
    double *d_q; int64_t *d_off;
    HIPCHECK(hipMalloc((void **)&d_q, 8 * n_qk)); HIPCHECK(hipMalloc((void **)&d_off, 8 * (Q + 1)));
    HIPCHECK(hipMemcpy(d_q, qkeys, 8 * n_qk, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_off, qoffs, 8 * (Q + 1), hipMemcpyHostToDevice));
    const size_t ws = tvz_match_workspace_bytes((int32_t)Q, (int32_t)max_len, CAP, K, 1);
    hipStream_t st[MAXD]; hipEvent_t ev[MAXD]; void *d_ws[MAXD]; int32_t *d_top[MAXD], *d_tot[MAXD];
    for (int i = 0; i < depth; i++) {
        HIPCHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        HIPCHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        HIPCHECK(hipMalloc(&d_ws[i], ws));
        HIPCHECK(hipMalloc((void **)&d_top[i], (size_t)Q * K * 3 * 4)); HIPCHECK(hipMalloc((void **)&d_tot[i], (size_t)Q * 4));
    }
#define SUBMIT(i) do { CHECK(tvz_match_sharded(c, comm, d_q, d_off, (int32_t)Q, (int32_t)max_len, 2, NULL, CAP, K, d_top[i], d_tot[i], \
                                               d_ws[i], ws, TVZ_ALGO_AUTO | shape, st[i])); HIPCHECK(hipEventRecord(ev[i], st[i])); } while (0)
    for (int w = 0; w < 2 * depth; w++) { SUBMIT(w % depth); }
    HIPCHECK(hipDeviceSynchronize());
    double t_submit = 0, t_wait = 0;
    const double t0 = now_us();
    int inflight = 0;
    for (int n = 0; n < steps; n++) {
        const int i = n % depth;
        const double a = now_us();
        if (inflight >= depth) { HIPCHECK(hipEventSynchronize(ev[i])); inflight--; }      /* slot i's previous batch */
        const double b = now_us();
        SUBMIT(i);
        inflight++;
        t_wait += b - a;
        t_submit += now_us() - b;
    }
    HIPCHECK(hipDeviceSynchronize());
    const double dt = now_us() - t0;
    int32_t tot0 = 0;
    HIPCHECK(hipMemcpy(&tot0, d_tot[0], 4, hipMemcpyDeviceToHost));
    printf("{\"host\": \"C\", \"rows\": %lld, \"Q\": %lld, \"batches_in_flight\": %d, \"us_per_batch\": %.1f, \"host_us_in_submit\": %.1f, "
           "\"host_us_waiting\": %.1f, \"hits_of_query_0\": %d}\n", (long long)n_rows, (long long)Q, depth, dt / steps, t_submit / steps,
           t_wait / steps, tot0);
    CHECK(tvz_comm_destroy(comm));
    CHECK(tvz_corpus_destroy(c));
    return 0;
}
