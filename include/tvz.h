/*
 * tvz.h — C ABI of libtvz.so, the MI355X (gfx950) duplicate-detection core.
 *
 * The reference (infraheads/tvidz) has NO FFI / plugin boundary on this path.
 * Its two seams are
 *   (1) a child process:  ffmpeg -vf select=gt(scene\,0.3),showinfo   whose
 *       stderr is text-parsed             inspector/app.py:202-232
 *   (2) a Python function: db.find_duplicates(new_timestamps, min_match=5)
 *       -> [(video_id, match_count)]       inspector/db.py:76-94,
 *       driven per new cut by               inspector/app.py:234-255
 * Each entry point below cites the seam it replaces.  Everything is plain C:
 * pointers + sizes, no torch / C++ types.  `d_` = device (HBM) pointer,
 * `h_` = host pointer.  `hip_stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  All functions are re-entrant and the library
 * keeps NO process-global mutable state: kernel-shape / algorithm choices are
 * per-call arguments, carried stream state lives in caller-owned device
 * memory, and the only shared mutable objects are the tvz_corpus / tvz_comm
 * handles (internally locked).  Hot calls allocate nothing: scratch is a
 * caller-provided workspace sized by the *_workspace_bytes functions.
 *
 * Error convention: 0 on success, negative tvz_status otherwise; the message
 * is in tvz_last_error() (thread-local).  The Python shim raises
 * RuntimeError(msg) so failures land in the same `except Exception` as the
 * reference's (inspector/app.py:303).
 */
#ifndef TVZ_H
#define TVZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVZ_VERSION 400 /* 0.4.0: the index answers every min_match >= 1; tvz_match_topk keeps the top-k inside the lookup; tvz_match_topk_shards */

typedef enum tvz_status {
    TVZ_OK = 0,
    TVZ_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, ...) */
    TVZ_ERR_HIP = -2,         /* a HIP runtime call failed */
    TVZ_ERR_NOMEM = -3,       /* device or host allocation failed */
    TVZ_ERR_UNSUPPORTED = -4, /* shape outside what the kernels handle */
    TVZ_ERR_WORKSPACE = -5,   /* caller-provided workspace too small */
    TVZ_ERR_COMM = -6         /* RCCL missing or a collective failed */
} tvz_status;

#define TVZ_KTH_NEVER 0x7fffffff /* kth_hit_idx of a candidate that never reaches min_match */

int tvz_version(void);
const char *tvz_last_error(void);

/* ------------------------------------------------------------------------
 * Scene-cut scoring   — replaces the arithmetic reached through
 * inspector/app.py:202-209 (ffmpeg `select=gt(scene,0.3)`; upstream
 * libavfilter/f_select.c get_scene_score + scene_sad.c).
 * ------------------------------------------------------------------------ */

/* Bytes of scratch the scene entry points need for a batch of up to T frames. */
size_t tvz_scene_workspace_bytes(int64_t T, int32_t H, int32_t W);

/* Stream state: what get_scene_score carries from one frame to the next (the
 * previous frame and prev_mafd), kept in DEVICE memory so that a stream scored
 * in micro-batches needs no host round trip between batches and a chain of
 * batches can be captured in one HIP graph.  The caller owns a buffer of
 * tvz_scene_state_bytes() bytes (256-byte aligned), resets it once per video
 * and passes it to every tvz_scene_scores_* call of that video; each call READS
 * the predecessor (frame + mafd) from it and WRITES its own last frame + mafd
 * back.  bytes_per_sample: 1 (8-bit) or 2 (9..16-bit luma in uint16). */
size_t tvz_scene_state_bytes(int32_t H, int32_t W, int32_t bytes_per_sample);
int tvz_scene_state_reset(void *d_state, void *hip_stream);

/* Kernel-shape override for A/B runs, PER CALL (there is no global knob):
 * 0 = automatic; otherwise U | (tc << 8) | TVZ_SHAPE_NO_NT where U in {1,2,4,8} is the strip
 * width (16-byte loads per lane and frame) and tc the frames per time chunk (8,16,32 or a
 * multiple of 64).  Results never depend on it. */
#define TVZ_SHAPE_AUTO 0u
#define TVZ_SHAPE_NO_NT (1u << 30)
#define TVZ_SHAPE(U, tc) ((uint32_t)(U) | ((uint32_t)(tc) << 8))

/* Luma sum of absolute differences between consecutive frames of a batch.
 *   d_luma : uint8 luma planes, frame t at d_luma + t*frame_stride_bytes,
 *            row y at + y*row_stride_bytes, W bytes per row.
 *   d_sad_out[T] : sad[0] = 0, sad[t] = sum |luma[t] - luma[t-1]|  (exact).
 * Every luma byte is read from HBM once (+1 halo frame per time chunk).
 * Fast path: row_stride == W, base and frame stride 16-byte aligned; anything
 * else takes a slower generic kernel with identical results. */
int tvz_luma_sad_u8(const uint8_t *d_luma, int64_t T, int32_t H, int32_t W,
                    int64_t frame_stride_bytes, int64_t row_stride_bytes,
                    uint64_t *d_sad_out, void *d_workspace, size_t workspace_bytes,
                    void *hip_stream);

/* get_scene_score epilogue over a SAD vector:
 *   mafd  = sad / (W*H) / 2^(bitdepth-8);  diff = |mafd - prev_mafd|
 *   score = clipf((float)(min(mafd, diff) / 100), 0, 1);  selected = score > threshold
 * d_prev_mafd = NULL: element 0 is the first frame of the stream (score 0, and
 * element 1 sees ffmpeg's zero-initialised prev_mafd).
 * d_prev_mafd != NULL (device pointer to one double): the batch continues a
 * stream; sad[0] is a real SAD against the previous batch's last frame and
 * *d_prev_mafd is that batch's last mafd.  d_score / d_mafd may be NULL. */
int tvz_scene_select(const uint64_t *d_sad, int64_t T, int32_t H, int32_t W,
                     int32_t bitdepth, double threshold, const double *d_prev_mafd,
                     uint8_t *d_selected, double *d_score, double *d_mafd, void *hip_stream);

/* Fused batch call: luma -> sad, mafd, score, selected (+ the compacted cut list).
 *   d_state (nullable): stream state (above).  NULL = a self-contained batch whose frame 0 is
 *       the first frame of a stream (score 0).  Non-NULL: frame 0 is scored against the state's
 *       frame unless the state was just reset, and the state is advanced to this batch's end.
 *   d_cuts (nullable): int32[1 + cuts_cap]; [0] = number of selected frames of the batch, then
 *       their indices in ascending order (the first cuts_cap of them) - ONE small device-to-host
 *       copy per micro-batch gives the host everything showinfo would have printed.
 *   Any of d_sad_out / d_score / d_mafd may be NULL; d_selected is required with d_cuts.
 * Enqueues kernels only (no allocation, no synchronisation): graph-capturable. */
int tvz_scene_scores_u8(const uint8_t *d_luma, int64_t T, int32_t H, int32_t W,
                        int64_t frame_stride_bytes, int64_t row_stride_bytes, void *d_state,
                        int32_t bitdepth, double threshold, uint64_t *d_sad_out, double *d_mafd,
                        double *d_score, uint8_t *d_selected, int32_t *d_cuts, int32_t cuts_cap,
                        void *d_workspace, size_t workspace_bytes, uint32_t shape,
                        void *hip_stream);

/* The same for 9..16-bit luma stored as uint16 (yuv420p10 etc.): ffmpeg's ff_scene_sad16_c sums
 * |a-b| over uint16 samples and get_scene_score divides mafd by 2^(bitdepth-8).  Strides are in
 * BYTES. */
int tvz_scene_scores_u16(const uint16_t *d_luma, int64_t T, int32_t H, int32_t W,
                         int64_t frame_stride_bytes, int64_t row_stride_bytes, void *d_state,
                         int32_t bitdepth, double threshold, uint64_t *d_sad_out, double *d_mafd,
                         double *d_score, uint8_t *d_selected, int32_t *d_cuts, int32_t cuts_cap,
                         void *d_workspace, size_t workspace_bytes, uint32_t shape,
                         void *hip_stream);

/* ------------------------------------------------------------------------
 * Timestamp corpus   — device image of the `video_timestamps` table
 * (inspector/db.py:21-27): one row = (video_id, float8[] timestamps).
 * ------------------------------------------------------------------------ */
typedef struct tvz_corpus tvz_corpus;

int tvz_corpus_create(tvz_corpus **out, int device);
/* Must not run concurrently with any other call on the same handle. */
int tvz_corpus_destroy(tvz_corpus *c);

/* Pre-size the device arena / row table / single-query staging for at least this many rows and
 * keys, so later upserts and queries allocate nothing (upload reserves 2x its input itself). */
int tvz_corpus_reserve(tvz_corpus *c, int64_t n_rows, int64_t n_keys);

/* Replace the whole corpus (the `session.query(VideoTimestamps).all()` of
 * db.py:83 done once instead of per call).  Row r owns
 * h_keys[h_offsets[r] .. h_offsets[r+1]).  Rows need not be sorted or unique;
 * NaN keys are dropped (NaN == x is false), -0.0 is folded into +0.0. */
int tvz_corpus_upload(tvz_corpus *c, const int32_t *h_video_ids, const int64_t *h_offsets,
                      const double *h_keys, int64_t n_rows, int64_t n_keys);

/* add_timestamps (db.py:43-64): replace the first row of `video_id` with the
 * new list, or append a row if the video has none.  Does NOT wait for matches
 * in flight: the new keys go to fresh arena space and the 16-byte row entry is
 * swapped by a stream-ordered device write, so a concurrent match sees the old
 * or the new row, never a mix; every match ENQUEUED after this call returns
 * sees the new row (read-your-writes, as a committed db.py:58-62 would give). */
int tvz_corpus_upsert(tvz_corpus *c, int32_t video_id, const double *h_keys, int64_t n);

/* `/admin/clear-db` (app.py:325-333).  Matches already enqueued keep sweeping the rows they were
 * launched with, unchanged: the arena is reused by later upserts only after those matches (the
 * mutation stream waits for them on the device; the host does not). */
int tvz_corpus_clear(tvz_corpus *c);

int tvz_corpus_stats(tvz_corpus *c, int64_t *n_rows, int64_t *n_keys, int64_t *arena_keys);

/* Inverted index (no reference counterpart: db.py:83 has no index and reads the whole table per
 * call).  The handle keeps, next to the row table, key -> rows posting lists over the rows as
 * they were at the last build - ONE directory over the distinct keys of all rows (one probe per
 * query timestamp), the postings of a key contiguous and ordered by sub-index of 16384 rows - and a
 * DELTA table with the current entry of every row added or replaced since.  A match with
 * min_match 1..5 is then a lookup (cost ~ postings of the query's keys, independent of the corpus
 * size) + a sweep of the delta table; results are identical to a full sweep (every row is in
 * exactly one of the two).
 * Built by tvz_corpus_upload and by this call (both wait for matches in flight).  Rebuilt IN THE
 * BACKGROUND - by the upserting thread that crosses the threshold, into a shadow generation that is
 * swapped in under the handle's lock; matches and other upserts go on meanwhile, no reader waits -
 * when a corpus grown by upserts reaches 4096 rows and when the delta table holds
 * max(512, indexed rows / 256) rows.  Both generations are sized with the corpus reservation
 * (tvz_corpus_reserve / upload), so rebuilds allocate only once the corpus has outgrown it.
 * A corpus with >= 2^32 keys (per GPU) gets no index and is swept. */
int tvz_corpus_build_index(tvz_corpus *c);
/* indexed rows / rows in the delta table / postings / distinct keys (directory entries in use) /
 * builds so far (any may be NULL); all 0 while there is no index. */
int tvz_corpus_index_stats(tvz_corpus *c, int64_t *n_indexed_rows, int64_t *n_delta_rows,
                           int64_t *n_postings, int64_t *n_distinct_keys, int64_t *n_builds);
/* The directory of a handle of ONE sub-index (up to 16,384 indexed rows) is a table of 128-byte buckets - a key's
 * entry and its postings in one cache line.  out[0] = buckets (0: the handle has no index, or more than one
 * sub-index: the open-addressing directory), out[1] = keys that do not live in their home bucket, out[2] = how far
 * the farthest of them walked (buckets), out[3] = keys whose posting list lives in the external area,
 * out[4] = external postings (uint16 units, whole lines), out[5] = sub-indexes. */
int tvz_corpus_bucket_stats(tvz_corpus *c, int64_t out[6]);

/* ------------------------------------------------------------------------
 * Corpus match   — replaces db.find_duplicates (inspector/db.py:76-94) and
 * the per-prefix loop around it (inspector/app.py:231-255).
 *
 * For every (query q, corpus row r):
 *   count = #{ i : query_q[i] in row_r }                (db.py:86-89; query
 *           multiplicity counts, the row is a set, exact float64 ==)
 *   kth   = index i of the min_match-th such hit        (TVZ_KTH_NEVER if
 *           count < min_match, -1 if min_match <= 0)
 * A pair is a HIT iff count >= min_match (db.py:90) and video_id !=
 * exclude_id[q] (app.py:237).  Hits are appended per query as int32 triples
 * (video_id, count, kth), in unspecified order (as db.py:83 has no ORDER BY).
 * `kth` makes the streaming loop a single call: prefix k+1 is the first prefix
 * on which find_duplicates returns row r  <=>  kth == k.
 * ------------------------------------------------------------------------ */

/* How the corpus is matched, PER CALL (there is no global knob); results never depend on it.
 *   AUTO : INDEX when the handle has one and min_match >= 1 (1..5: kth from the smallest positions kept per
 *          candidate; more: count + a kth fix-up walk).  Otherwise (and for the delta
 *          table): one query (or <= 4 against a small corpus) -> Q1; >= 64 queries x >= 2,100 + 340,000 / Q rows
 *          with min_match 1..2 -> JOIN; else TILE
 *   INDEX: posting-list lookup - one block per query walks its sub-indexes (a small batch: one block per
 *          query and sub-index) - + a sweep of the delta table (error if the handle has no index or
 *          min_match < 1)
 *   Q1   : one corpus sweep per query, the query's keys in a small LDS table, per-lane counters
 *   TILE : one LDS hash table per tile of <= 16 queries
 *   JOIN : device-memory hash join per tile of <= 1024 queries (min_match 1..2; other values take TILE) */
#define TVZ_ALGO_AUTO 0
#define TVZ_ALGO_Q1 1
#define TVZ_ALGO_TILE 2
#define TVZ_ALGO_JOIN 3
#define TVZ_ALGO_INDEX 4
/* Shape of the lookup that keeps the top-k (tvz_match_topk, tvz_match_topk_shards, tvz_match_sharded), OR-ed into
 * `algo`.  By default a block takes TWO queries - their directory probes share one probe phase - when the batch
 * still fills the chip with half as many blocks (Q >= 2048) and the LDS of both fits four blocks per CU (queries of
 * up to ~380 timestamps on a handle of one or two sub-indexes, ~150 at seven); _PAIR asks for it at any batch size
 * (where the LDS allows), _NO_PAIR never.  Same results either way. */
#define TVZ_ALGO_PAIR 0x100
#define TVZ_ALGO_NO_PAIR 0x200
/* On a handle of ONE sub-index (up to 16,384 indexed rows: a rank's share of an 8-way sharded 100k-video table) and
 * queries of up to 512 timestamps the lookup that keeps the top-k gives every query to one WAVE instead of a block
 * (ts_match_wq_topk_kernel: no barriers, every probe of the query in flight at once, the postings in registers
 * between the passes).  _NO_WAVE keeps the block kernel, _WAVE asks for the wave kernel (an error where the handle or
 * the batch does not fit it).  Same results either way. */
#define TVZ_ALGO_WAVE 0x400
#define TVZ_ALGO_NO_WAVE 0x800
/* Which of the two is faster depends on what else the GPU is doing: ONE batch alone is answered sooner by the block
 * kernel (59 against 72 us for 4,096 queries on a 12.5k-row shard: all waves of the wave kernel start their probe
 * bursts together), a STREAM of batches - two or three in flight on their own streams, what tvz_match_sharded's
 * callers run - by the wave kernel, which executes a third fewer instructions (40 against 42 us per batch).  The
 * default is the block kernel; _PREFER_WAVE takes the wave kernel wherever it fits and says nothing where it
 * does not (sharded.RcclShardedMatcher sets it). */
#define TVZ_ALGO_PREFER_WAVE 0x1000

/* Scratch for the batched calls below.  k = 0 for tvz_match (tables of the hash join only);
 * k > 0 adds the hit lists + per-shard top-k block of tvz_match_topk / tvz_match_sharded
 * (and n_ranks gathered blocks for the latter; pass n_ranks = 1 otherwise). */
size_t tvz_match_workspace_bytes(int32_t Q, int32_t max_query_len, int32_t cap, int32_t k,
                                 int32_t n_ranks);
/* ... for a batch that holds queries of MORE than 4,095 timestamps (inspector/db.py:87 has no limit).  Such a query
 * is swept on its own: its sorted distinct keys and their multiplicities are made ON THE DEVICE, in a tail of this
 * workspace - the call allocates nothing and synchronises once (the host has to read the query offsets to learn
 * which queries are long).  tvz_match_workspace_bytes has room for ONE query of max_query_len keys;
 * total_query_keys = the key count of the whole batch (the length of d_queries) makes room for any split of it
 * into long queries.  A call whose workspace is too small for its long queries returns TVZ_ERR_WORKSPACE and says
 * how many bytes they need. */
size_t tvz_match_workspace_bytes_long(int32_t Q, int32_t max_query_len, int32_t cap, int32_t k,
                                      int32_t n_ranks, int64_t total_query_keys);

/* Batched, device-resident form.
 *   d_queries   : float64 keys of all queries back to back
 *   d_q_offsets : int64[Q+1]
 *   d_exclude_ids : int32[Q] or NULL
 *   d_hits      : int32[Q][cap][3] out;  d_hits_n : int32[Q] out = number of
 *                 hits found (may exceed cap; only the first cap are stored)
 *   max_query_len : upper bound on any query's length (sizes the query tables).  If it is NOT an
 *                 upper bound the affected queries get d_hits_n = INT32_MIN.
 *                 Any length is accepted (db.py:87 has no limit).  Up to 4095 the call only enqueues
 *                 kernels.  Above, the batch may hold queries of more than 4095 timestamps: the
 *                 library then reads the offsets back, runs the short queries as usual and sweeps
 *                 every long one on its own (sorted-query search + a kth fix-up pass) - the one
 *                 batched path that synchronises the stream and allocates scratch.  Rare by nature
 *                 (a video with thousands of cuts). */
int tvz_match(tvz_corpus *c, const double *d_queries, const int64_t *d_q_offsets,
              int32_t Q, int32_t max_query_len, int32_t min_match,
              const int32_t *d_exclude_ids, int32_t cap,
              int32_t *d_hits, int32_t *d_hits_n, void *d_workspace, size_t workspace_bytes,
              int32_t algo, void *hip_stream);

/* Match + per-shard top-k behind ONE call (hit lists stay in the workspace):
 *   d_out : int32[Q][k+1][3] = the k best hits by (kth, video_id, count), padded with
 *           (-1, 0, TVZ_KTH_NEVER), + a row (-1, n_hits, TVZ_KTH_NEVER); n_hits is NEGATED when
 *           the hit list overflowed `cap` (the top-k may then be inexact: re-run with more).
 * On an indexed handle (min_match 1..5, k <= 64, a batch large enough that a block owns its query) the
 * lookup kernel keeps the k best ITSELF: no hit list is written, no top-k kernel runs; rows added or
 * replaced since the index was built are swept and merged in.  The contract is unchanged (n_hits is
 * still negated when it exceeds `cap`), the rows are then the exact k best nevertheless. */
int tvz_match_topk(tvz_corpus *c, const double *d_queries, const int64_t *d_q_offsets,
                   int32_t Q, int32_t max_query_len, int32_t min_match,
                   const int32_t *d_exclude_ids, int32_t cap, int32_t k, int32_t *d_out,
                   void *d_workspace, size_t workspace_bytes, int32_t algo, void *hip_stream);

/* Host-in / host-out single query: the drop-in for db.find_duplicates.
 * Results sorted by (video_id, count).  *n_out = number of hits (if > cap only
 * cap are returned).  h_out_kth may be NULL.  Any query length is accepted.
 * One kernel launch + one stream synchronisation for queries of <= 4095 timestamps - also when the
 * handle has an index AND rows in the delta table (the streaming driver's call: its own row was
 * upserted a moment ago): lookup and delta sweep are one fused launch - the kernel writes its hits
 * straight into pinned host memory owned by the handle. */
int tvz_find_duplicates(tvz_corpus *c, const double *h_query, int64_t n, int32_t min_match,
                        int32_t exclude_id, int64_t cap, int32_t *h_out_ids,
                        int32_t *h_out_counts, int32_t *h_out_kth, int64_t *n_out);

/* Per-query top-k of hit lists ordered by (kth asc, video_id asc, count asc).
 *   d_lists : int32[n_lists][Q][cap][3]  (n_lists = 1 for a local tvz_match
 *             result; = world size for an all-gathered set of per-shard top-k)
 *   d_lists_n : int32[n_lists][Q] valid counts, or NULL = all cap entries
 *             (entries with video_id < 0 are padding and sort last)
 *   d_topk  : int32[Q][k][3] out, padded with (-1, 0, TVZ_KTH_NEVER). */
int tvz_topk(const int32_t *d_lists, const int32_t *d_lists_n, int32_t n_lists, int32_t Q,
             int32_t cap, int32_t k, int32_t *d_topk, void *hip_stream);

/* Sharded form (one process per GPU): each rank reduces its local hit lists to
 *   d_out : int32[Q][k+1][3] = its k best hits + a row (-1, n_local_hits, TVZ_KTH_NEVER),
 * the ranks all-gather those blocks (RCCL), and every rank merges
 *   d_gathered : int32[n_ranks][Q][k+1][3]  ->  d_topk int32[Q][k][3], d_totals int32[Q]
 * (d_totals = hits over all shards; > k means the merged list is truncated to the k best;
 * NEGATIVE means some shard's hit list overflowed `cap`, so its top-k may be inexact: re-run with
 * a larger cap).  The gathered blocks are what tvz_match_topk / tvz_topk_shard wrote: k rows in
 * ascending (kth, video_id, count) order, padding last - up to 16 ranks and k <= 64 are merged as
 * SORTED lists (a k-way merge, not a selection). */
int tvz_topk_shard(const int32_t *d_hits, const int32_t *d_hits_n, int32_t Q, int32_t cap,
                   int32_t k, int32_t *d_out, void *hip_stream);
/* The shards of ONE process (several handles on one device: service.ShardedCorpus, the one-GPU form of
 * configs[4]): tvz_match_topk on every handle in turn, the blocks written where the merge reads them
 *   d_blocks : int32[n_shards][Q][k+1][3]  (as an all-gather would deliver them)
 * and the merge -> d_topk int32[Q][k][3], d_totals int32[Q] - ONE call for the whole tick (a Python
 * host otherwise re-takes its interpreter lock after every launch; with 16 upload threads busy in the
 * ORM that was ~4 ms per tick against ~0.2 ms of GPU work).  One workspace sized for ONE handle
 * (tvz_match_workspace_bytes(Q, max_query_len, cap, k, 1)) serves all of them: same stream, in order. */
int tvz_match_topk_shards(tvz_corpus *const *shards, int32_t n_shards, const double *d_queries,
                          const int64_t *d_q_offsets, int32_t Q, int32_t max_query_len, int32_t min_match,
                          const int32_t *d_exclude_ids, int32_t cap, int32_t k, int32_t *d_blocks,
                          int32_t *d_topk, int32_t *d_totals, void *d_workspace, size_t workspace_bytes,
                          int32_t algo, void *hip_stream);
int tvz_topk_merge(const int32_t *d_gathered, int32_t n_ranks, int32_t Q, int32_t k,
                   int32_t *d_topk, int32_t *d_totals, void *hip_stream);

/* ------------------------------------------------------------------------
 * Multi-GPU corpus match: one process per GPU, the corpus rows sharded over
 * the ranks, ONE RCCL all-gather of the per-shard top-k blocks per query batch.  RCCL is bound at
 * run time (dlopen of librccl.so, the copy already mapped by the process if there is one), so
 * libtvz.so loads on hosts without it; tvz_comm_* then return TVZ_ERR_COMM.
 *   rank 0: tvz_comm_unique_id(id)  -> ship the 128 bytes to the other ranks by any means
 *   every rank, BEFORE or after its first GPU call: tvz_comm_init(&comm, id, nranks, rank, dev)
 * ------------------------------------------------------------------------ */
#define TVZ_UNIQUE_ID_BYTES 128
typedef struct tvz_comm tvz_comm;

int tvz_comm_unique_id(void *out_id /* TVZ_UNIQUE_ID_BYTES */);
int tvz_comm_init(tvz_comm **out, const void *unique_id, int32_t n_ranks, int32_t rank,
                  int32_t device);
int tvz_comm_info(tvz_comm *comm, int32_t *n_ranks, int32_t *rank);
int tvz_comm_destroy(tvz_comm *comm);

/* local tvz_match_topk -> ncclAllGather of int32[Q][k+1][3] on `hip_stream` -> tvz_topk_merge.
 * Every rank gets the same d_topk int32[Q][k][3] and d_totals int32[Q] (see tvz_topk_merge).
 * Workspace: tvz_match_workspace_bytes(Q, max_query_len, cap, k, n_ranks).  Two calls on two
 * streams with two workspaces overlap one batch's collective with the next batch's match. */
int tvz_match_sharded(tvz_corpus *c, tvz_comm *comm, const double *d_queries,
                      const int64_t *d_q_offsets, int32_t Q, int32_t max_query_len,
                      int32_t min_match, const int32_t *d_exclude_ids, int32_t cap, int32_t k,
                      int32_t *d_topk, int32_t *d_totals, void *d_workspace,
                      size_t workspace_bytes, int32_t algo, void *hip_stream);

/* ------------------------------------------------------------------------
 * Opt-in alignment score.  NOT the reference's verdict: db.py:79 matches
 * exactly; README.md:291 ("0.1 s tolerance") and north_star's "alignment/Jaccard" describe a
 * shift/tolerance-aware comparison, which this reports ALONGSIDE the exact one.  For one query
 * against every corpus row, every (query_i, row_j) difference votes into bins of width eps over
 * [-max_offset, +max_offset] (bin = floor(diff/eps + 0.5)):
 *   d_out : int32[n_rows][5] = (video_id, row_len, best_bin, votes_in_best_bin, votes_in_bin_0)
 * best_bin ties: smaller |bin| first, then the negative one.  A cut-shifted copy shows as
 * votes ~ min(n, row_len) at best_bin = shift/eps; tolerant Jaccard = v / (n + row_len - v).
 * Needs 2*round(max_offset/eps)+1 <= 4096 bins. */
int tvz_align(tvz_corpus *c, const double *d_query, int32_t n, double eps, double max_offset,
              int32_t *d_out, void *hip_stream);

/* ------------------------------------------------------------------------
 * Frame feeder I/O - the host side of what replaces the stderr pipe of
 * inspector/app.py:209-216: a micro-batch of luma planes from a file or a decoder pipe into the
 * caller's (pinned) buffer in one call, no per-frame host-language loop.
 * ------------------------------------------------------------------------ */

/* n_records fixed-size records from a seekable file (positioned reads: the descriptor's offset is not
 * used): record i starts at file_offset + i*record_bytes = [header_bytes header][payload_bytes payload]
 * [rest skipped].  YUV4MPEG2: header "FRAME\n", payload = the Y plane, rest = chroma.  `magic`
 * (nullable): header_bytes bytes every header must equal.  *n_done = whole records read (fewer at end
 * of file). */
int tvz_read_records(int fd, int64_t file_offset, int64_t n_records, int64_t record_bytes,
                     const void *magic, int64_t header_bytes, int64_t payload_bytes, void *h_dst,
                     int64_t *n_done);

/* The same from a pipe (`ffmpeg -f rawvideo -`): payload_bytes are kept, the following skip_bytes
 * (chroma planes) are read and dropped. */
int tvz_read_stream(int fd, int64_t n_records, int64_t payload_bytes, int64_t skip_bytes, void *h_dst,
                    int64_t *n_done);

#ifdef __cplusplus
}
#endif
#endif /* TVZ_H */
