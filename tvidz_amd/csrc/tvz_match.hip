// tvz_match.hip — timestamp-corpus matcher for MI355X (gfx950, wave64).
//
// Replaces /root/reference inspector/db.py:76-94 (find_duplicates) and the per-prefix
// loop around it, inspector/app.py:231-255.  Semantics (db.py:85-91): for every corpus
// row, count the query elements that are `in` the row (exact float64 ==; the query keeps
// its multiplicity, the row acts as a set); a row is a hit iff count >= min_match.
//
// Device image of `video_timestamps` (db.py:21-27):
//   rows[r] = {key offset, key count, video_id}           16 B each
//   keys    = arena of canonical float64 bit patterns (int64): per row sorted, unique,
//             NaN dropped, -0.0 folded to +0.0; every row starts 16-byte aligned.
//
// Kernel ts_match_tile_kernel<TOP5>: grid = (row chunks, query tiles).  A 1024-thread block builds
//   ONE hash table in LDS for a tile of up to 16 queries (16-bit tag + chain head per slot, two
//   slots per aligned 8-byte probe; every query element is a chained (key, position, query) entry,
//   so query multiplicity is exact), then sweeps its chunk of rows: a 16-lane group owns one row
//   at a time, streams its keys with 16-byte loads and probes the table ONCE per key for all the
//   queries of the tile - the tile plays the role a GEMM tile plays: every corpus byte loaded is
//   reused 16 times on chip.  The per-key fast path is branch-free; lanes that found their tag
//   (or a full home pair) push the key into a per-wave LDS ring by ballot/mbcnt compaction and the
//   wave drains the ring 64 entries at a time with every lane busy on the exact probe, full-key
//   verification and accounting (LDS atomics: hit count + the smallest matching positions).
//   After a row, lane q of the group emits query q's hit (video_id, count, kth); kth = the
//   min_match-th smallest matching position, read from the tracked minima for min_match <= 5
//   (the reference's default 5 and the driver's 2), by ts_kth_fixup_kernel beyond.
//   Integer / LDS / issue-bound: no MFMA.
// Kernel ts_match_longq_kernel: single queries longer than a tile (> 4095 timestamps).
// Kernel ts_topk_kernel: per query bitonic selection of the k best hits ordered by
//   (kth, video_id, count) over one or several (all-gathered) hit lists.
// Kernel ts_align_kernel: opt-in shift/tolerance score (never the verdict).
#include <algorithm>
#include <climits>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <vector>

#include "tvz_common.h"

#ifndef TVZ_MATCH_STEP
#define TVZ_MATCH_STEP 2   // 16-byte key loads per lane and sweep step (4 keys); 3/4/6 measured no faster
#endif

namespace {

constexpr int kBlock = 256;
constexpr int kGroup = 16;                  // lanes per corpus row
constexpr int kGroupsPerBlock = kBlock / kGroup;
constexpr int64_t kEmpty = 0x7ff8dead00000000LL;  // a NaN pattern: never a canonical key
constexpr int kMaxQueryLen = 4095;          // positions 0..4094 fit 12 bits with 0xfff as "none"

struct Row {
    int64_t off;
    int32_t len;
    int32_t vid;
};
static_assert(sizeof(Row) == 16, "Row must be 16 bytes");

// ---- canonical key: integer-only so subnormals / signed zero never meet FP modes ----
__host__ __device__ inline bool canon_key(double x, int64_t &k) {
    int64_t b;
    memcpy(&b, &x, 8);
    const uint64_t mag = (uint64_t)b & 0x7fffffffffffffffULL;
    if (mag > 0x7ff0000000000000ULL) return false;  // NaN: == is always false
    k = (mag == 0) ? 0 : b;                         // -0.0 == +0.0
    return true;
}

// ---- query tile: up to 16 queries share ONE hash table in LDS -----------------------------
// One probe of a corpus key serves every query of the tile, and each corpus row is read once
// per tile instead of once per query.
//   slots   : 16384 x u32 = (16-bit tag << 16) | (head entry index); 0xffffffff = empty.
//             Probed two at a time (one aligned ds_read_b64); load factor <= 0.25, so a probe
//             almost never needs a second read - what matters on a 64-lane wave is the LONGEST
//             probe of the wave, not the average.
//   entries : one per query element of the tile: full canonical key (verification) and
//             (position, query-in-tile, next entry with the same slot).  Query multiplicity is
//             therefore exact: every occurrence is its own entry.
constexpr int kTileBlock = 1024;                      // 16 waves, 64 row groups
constexpr int kTileGroups = kTileBlock / kGroup;
constexpr int kTileQ = kGroup;                        // lane <-> query mapping at emission
constexpr int kTileSlots = 16384;
constexpr int kTilePairs = kTileSlots / 2;
constexpr int kTileMaxEntries = 4096;                 // load factor <= 0.25
constexpr uint32_t kEnd = 0xffffu;
constexpr uint32_t kFree = 0xffffffffu;
static_assert(kTileMaxEntries >= kMaxQueryLen, "a single maximal query must fit one tile");
constexpr int kRing = 128;                            // per-wave slow-path ring (entries)
constexpr size_t kTileLds = (size_t)kTileSlots * 4 + (size_t)kTileMaxEntries * 8 +
                            (size_t)kTileMaxEntries * 4 + (size_t)kTileGroups * kTileQ * 3 * 4 +
                            (size_t)(kTileBlock / 64) * kRing * 12;

// Per (row group, query of the tile) state in LDS: a hit counter and the FIVE smallest matching
// query positions, packed as 5 x 12 bits (ascending from bit 0, 0xfff = none) in one 64-bit word
// updated with a CAS loop.  kth for min_match <= 5 (the reference's default and the driver's 2)
// is read straight from it.
constexpr int kTop = 5;
constexpr unsigned long long kTopNone = 0x0fffffffffffffffULL;   // 5 fields of 0xfff

__device__ __forceinline__ unsigned long long top5_insert(unsigned long long p, uint32_t x) {
    uint32_t a[kTop];
#pragma unroll
    for (int i = 0; i < kTop; ++i) a[i] = (uint32_t)(p >> (12 * i)) & 0xfffu;
#pragma unroll
    for (int i = 0; i < kTop; ++i) {      // insertion network: keep the smaller, carry the larger
        const uint32_t lo = a[i] < x ? a[i] : x;
        x = a[i] < x ? x : a[i];
        a[i] = lo;
    }
    unsigned long long r = 0;
#pragma unroll
    for (int i = 0; i < kTop; ++i) r |= (unsigned long long)a[i] << (12 * i);
    return r;
}

// pair index (13 bits) and tag (16 bits) from one mix of the key: 9 full-rate VALU ops (one
// v_mul_u32_u24, no quarter-rate v_mul_lo_u32).  Quality only affects speed: every tag match is
// verified against the full key.  A tag of 0xffff may "match" a free slot's upper half; the slow
// path then finds an empty chain (head 0xffff = kEnd), which is the right answer.
__device__ __forceinline__ void hash_pair_tag(int64_t k, uint32_t &pair, uint32_t &tag) {
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)((uint64_t)k >> 32);
    uint32_t x = lo ^ (hi + (hi << 3)) ^ (hi >> 9);
    x ^= x >> 20;                                   // fold the bits v_mul_u32_u24 ignores
    const uint32_t y = __umul24(x, 0x9E3779u);
    pair = y >> (32 - 13);
    tag = (y >> 3) & 0xffffu;
}
static_assert((1 << 13) == kTilePairs, "pair bits must match kTilePairs");

// one matching (query, position) entry: count it and keep the two smallest positions
// TOP5 = false (min_match <= 2, the streaming driver's case): the 8-byte word holds the smallest
// and second smallest position as two u32 updated with two LDS atomicMin (7 % faster).
template <bool TOP5>
__device__ __forceinline__ void account(uint32_t *cnt, unsigned long long *top, uint32_t ent) {
    const uint32_t q = (ent >> 12) & 15u;
    const uint32_t pos = ent & 0xfffu;
    atomicAdd(&cnt[q], 1u);
    if constexpr (!TOP5) {
        uint32_t *m = reinterpret_cast<uint32_t *>(&top[q]);
        const uint32_t old = atomicMin(&m[0], pos);
        atomicMin(&m[1], old > pos ? old : pos);   // the larger of two distinct hits: >= 2nd smallest
        return;
    }
    unsigned long long seen = top[q];
    while (true) {
        if (((uint32_t)(seen >> (12 * (kTop - 1))) & 0xfffu) <= pos) break;   // not among the 5 smallest
        const unsigned long long old = atomicCAS(&top[q], seen, top5_insert(seen, pos));
        if (old == seen) break;
        seen = old;
    }
}

template <bool TOP5>
__global__ __launch_bounds__(kTileBlock) void ts_match_tile_kernel(
    const Row *__restrict__ rows, int64_t n_rows, const int64_t *__restrict__ keys,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t Q,
    int32_t nq_tile, int32_t min_match, const int32_t *__restrict__ exclude_ids, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *slots = reinterpret_cast<uint32_t *>(smem);
    int64_t *ekey = reinterpret_cast<int64_t *>(smem + (size_t)kTileSlots * 4);
    uint32_t *epack = reinterpret_cast<uint32_t *>(ekey + kTileMaxEntries);
    unsigned long long *top = reinterpret_cast<unsigned long long *>(epack + kTileMaxEntries);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(top + kTileGroups * kTileQ);
    __shared__ int64_t s_qoff[kTileQ + 1];

    const int q0 = blockIdx.y * nq_tile;
    const int nq = (Q - q0 < nq_tile) ? Q - q0 : nq_tile;
    if (threadIdx.x <= nq) s_qoff[threadIdx.x] = q_offsets[q0 + threadIdx.x];
    for (int i = threadIdx.x; i < kTileSlots; i += kTileBlock) slots[i] = kFree;
    for (int i = threadIdx.x; i < kTileGroups * kTileQ; i += kTileBlock) {
        top[i] = TOP5 ? kTopNone : ~0ULL;
        cnt[i] = 0;
    }
    __syncthreads();
    const int64_t qbase = s_qoff[0];
    if (s_qoff[nq] - qbase > kTileMaxEntries) {
        // the caller's max_query_len was not an upper bound: poison the affected counters instead
        // of returning silently truncated matches (every row chunk of this tile takes this exit)
        if (threadIdx.x < nq) hits_n[q0 + threadIdx.x] = INT32_MIN;
        return;
    }
    const int total = (int)(s_qoff[nq] - qbase);
    for (int e = threadIdx.x; e < total; e += kTileBlock) {
        int ql = 0;
        while (ql + 1 < nq && s_qoff[ql + 1] - qbase <= e) ++ql;
        const uint32_t pos = (uint32_t)(e - (int)(s_qoff[ql] - qbase));
        int64_t k;
        if (!canon_key(queries[qbase + e], k)) continue;   // NaN never matches
        ekey[e] = k;
        uint32_t pair, tag;
        hash_pair_tag(k, pair, tag);
        // first slot of the probe sequence that is free or already carries this tag
        uint32_t s = pair * 2, prev = kEnd;
        while (true) {
            uint32_t w = slots[s];
            if (w == kFree) {
                w = atomicCAS(&slots[s], kFree, (tag << 16) | (uint32_t)e);
                if (w == kFree) break;                      // claimed an empty slot
            }
            if ((w >> 16) == tag) {                         // push on this tag's chain
                uint32_t seen = w;
                while (true) {
                    const uint32_t old = atomicCAS(&slots[s], seen, (seen & 0xffff0000u) | (uint32_t)e);
                    if (old == seen) break;
                    seen = old;
                }
                prev = seen & 0xffffu;
                break;
            }
            s = (s + 1) & (kTileSlots - 1);
        }
        epack[e] = pos | ((uint32_t)ql << 12) | (prev << 16);
    }
    __syncthreads();

    const int gl = threadIdx.x & (kGroup - 1);
    const int g = threadIdx.x / kGroup;
    uint32_t *gcnt = cnt + g * kTileQ;
    unsigned long long *gtop = top + g * kTileQ;
    const bool my_q = gl < nq;
    const int32_t excl = (exclude_ids && my_q) ? exclude_ids[q0 + gl] : -1;
    const bool use_excl = exclude_ids != nullptr;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n_rows) r1 = n_rows;
    const uint2 *pairs = reinterpret_cast<const uint2 *>(slots);

    // ---- sweep -------------------------------------------------------------------------------
    // SIMT rule that shapes this loop: an event that is rare per LANE (a corpus key that is in
    // the tile, ~6 % on the synthetic corpora; a displaced key) still happens in almost every
    // 64-lane wave-instruction, so handling it inline costs every probe the full slow path.
    // Instead the per-key fast path is branch-free (hash, one aligned 8-byte LDS read of the home
    // slot pair, tag compares) and lanes that need more push (key, pair|tag|group) into a per-wave
    // LDS ring with a ballot/mbcnt compaction; whenever 64 entries are pending the whole wave
    // drains them with every lane busy on the exact probe + chain verification + accounting.
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t gw = (threadIdx.x >> 4) & 3u;                 // group within the wave
    const uint32_t gwbits = gw << 29;
    int64_t *qbase_k = reinterpret_cast<int64_t *>(cnt + kTileGroups * kTileQ);
    int64_t *qk = qbase_k + wave * kRing;
    uint32_t *qm = reinterpret_cast<uint32_t *>(qbase_k + (kTileBlock / 64) * kRing) + wave * kRing;
    uint32_t *wcnt = cnt + wave * 4 * kTileQ;                     // the wave's 4 groups
    unsigned long long *wtop = top + wave * 4 * kTileQ;
    uint32_t qhead = 0, qtail = 0;                                // wave-uniform

    auto drain = [&](uint32_t n) {                                // n <= 64 pending entries
        if ((uint32_t)lane < n) {
            const uint32_t idx = (qhead + lane) & (kRing - 1);
            const int64_t k = qk[idx];
            const uint32_t m = qm[idx];
            const uint32_t tag = m & 0xffffu;
            uint32_t pair = (m >> 16) & (uint32_t)(kTilePairs - 1);
            uint32_t *scnt = wcnt + (m >> 29) * kTileQ;
            unsigned long long *stop = wtop + (m >> 29) * kTileQ;
            uint32_t e = kEnd;
            while (true) {      // first slot of the probe sequence that is free or carries the tag
                const uint2 w = pairs[pair];
                if ((w.x >> 16) == tag) { e = w.x & 0xffffu; break; }
                if (w.x == kFree) break;
                if ((w.y >> 16) == tag) { e = w.y & 0xffffu; break; }
                if (w.y == kFree) break;
                pair = (pair + 1) & (uint32_t)(kTilePairs - 1);
            }
            while (e != kEnd) {  // every (query, position) entry of that slot; verify the full key
                const uint32_t ent = epack[e];
                if (ekey[e] == k) account<TOP5>(scnt, stop, ent);
                e = ent >> 16;
            }
        }
        qhead += n;
    };

    const int64_t rw0 = r0 + (int64_t)wave * 4;                   // first row of the wave's groups
    for (int64_t rr = rw0; rr < r1; rr += kTileGroups) {          // wave-uniform trip count
        const int64_t r = rr + gw;
        const bool live = r < r1;
        Row row = Row{0, 0, -1};
        if (live) row = rows[r];
        const int64_t *rk = keys + row.off + gl * 2;
        const int nmine = row.len - gl * 2;                       // keys at or after this lane's first
        // kStep 16-byte loads (2 keys each) per lane and step, the next step's loads in flight
        constexpr int kStep = TVZ_MATCH_STEP;
        constexpr int kStride = kGroup * 2;                       // keys between a lane's loads
        longlong2 v[kStep];
#pragma unroll
        for (int j = 0; j < kStep; ++j)
            v[j] = (nmine > j * kStride) ? *reinterpret_cast<const longlong2 *>(rk + j * kStride)
                                         : make_longlong2(0, 0);
        for (int i = 0; __ballot(i < nmine) != 0ull; i += kStep * kStride) {
            int64_t kk[2 * kStep];
#pragma unroll
            for (int j = 0; j < kStep; ++j) {
                kk[2 * j] = v[j].x;
                kk[2 * j + 1] = v[j].y;
            }
            const int in = i + kStep * kStride;
#pragma unroll
            for (int j = 0; j < kStep; ++j)
                if (in + j * kStride < nmine) v[j] = *reinterpret_cast<const longlong2 *>(rk + in + j * kStride);
            uint32_t pr[2 * kStep], tg[2 * kStep];
            uint2 w[2 * kStep];
#pragma unroll
            for (int j = 0; j < 2 * kStep; ++j) {
                hash_pair_tag(kk[j], pr[j], tg[j]);
                w[j] = pairs[pr[j]];
            }
#pragma unroll
            for (int j = 0; j < 2 * kStep; ++j) {
                const bool valid = i + (j / 2) * kStride + (j & 1) < nmine;
                // needs the slow path: tag present in the home pair, or the pair is full
                const bool slow = valid & (((w[j].x >> 16) == tg[j]) | (w[j].y != kFree));
                const unsigned long long bal = __ballot(slow);
                if (bal) {                                        // wave-uniform
                    const uint32_t ofs = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                         __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (slow) {
                        const uint32_t idx = (qtail + ofs) & (kRing - 1);
                        qk[idx] = kk[j];
                        qm[idx] = tg[j] | (pr[j] << 16) | gwbits;
                    }
                    qtail += (uint32_t)__popcll(bal);
                    if (qtail - qhead >= 64u) drain(64u);
                }
            }
        }
        if (qtail != qhead) drain(qtail - qhead);                 // row boundary: settle the counts
        // lane q of the group owns query q of the tile (LDS ops of a wave complete in order)
        if (my_q & live) {
            const uint32_t c = gcnt[gl];
            const unsigned long long t5 = gtop[gl];
            if (c) {
                gcnt[gl] = 0;
                gtop[gl] = TOP5 ? kTopNone : ~0ULL;
            }
            if ((int64_t)c >= (int64_t)min_match && !(use_excl && row.vid == excl)) {
                int32_t kth;
                if (min_match <= 0) kth = -1;
                else if (!TOP5) kth = (int32_t)(uint32_t)(min_match == 1 ? t5 : t5 >> 32);
                else if (min_match <= kTop) kth = (int32_t)((t5 >> (12 * (min_match - 1))) & 0xfffu);
                else kth = -2 - (int32_t)r;          // resolved by ts_kth_fixup_kernel
                const int slot = atomicAdd(&hits_n[q0 + gl], 1);
                if (slot < cap) {
                    int32_t *h = hits + ((int64_t)(q0 + gl) * cap + slot) * 3;
                    h[0] = row.vid;
                    h[1] = (int32_t)c;
                    h[2] = kth;
                }
            }
        }
    }
}

// ---- hash join for large query batches (Q >= 32, min_match <= 2) ----------------------------
// The LDS tile kernel probes every corpus key once per 16 queries.  For big batches against big
// corpora a database hash JOIN does less work: build one multimap per tile of 128 queries in device
// memory (sized to stay in one XCD's 4 MiB L2), then sweep the corpus once per tile - one probe of a
// corpus key serves 128 queries, 8x fewer probes than the LDS tile.  Blocks of one tile are mapped
// to one XCD (blockIdx % 8) so its table is served from that XCD's L2.  What it buys is bounded by
// the L2: a random 16-byte probe moves a whole cache line, and the measured probe rate (~180 G/s)
// is the L2's random-line rate - LDS has no line granularity, which is why the LDS tile stays
// competitive with 8x the probes (1.18 vs 2.30 ms at C=100k, Q=1024).  An LDS presence bitmap of
// the tile's keys (64 KiB) keeps ~70 % of the corpus keys from touching the L2 at all.  Per (row group, query) state lives in LDS: a u16 hit
// counter and the two smallest matching positions (u16 + u16 in one CAS word); after a row each
// lane scans 8 of the tile's 128 queries and emits the hits.
constexpr int kJoinQ = 128;
constexpr int kJoinBlock = 1024;
constexpr int kJoinGroups = kJoinBlock / kGroup;
constexpr int kJoinBloomBits = 1 << 19;                           // 64 KiB presence bitmap per tile
constexpr size_t kJoinLds = (size_t)kJoinGroups * kJoinQ * 10 + kJoinBloomBits / 8;   // 80 + 64 KiB
constexpr int64_t kJEmpty = -1;                                    // 0xff..ff: a NaN pattern

__device__ __forceinline__ uint32_t hash32(int64_t k) {
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)((uint64_t)k >> 32);
    uint32_t x = lo ^ (hi + (hi << 3)) ^ (hi >> 9);
    x ^= x >> 20;
    const uint32_t y = __umul24(x, 0x9E3779u);
    return y ^ (y >> 15);
}

// Table layout: keys int64[S] and packs u32[S] apart, so ONE 16-byte load fetches the two keys of a
// slot pair; the (position | query-in-tile << 12) pack is only loaded on a match.  Random global
// accesses cost the CU's address pipeline ~1 lane-address per cycle, so loads per probe are what
// bounds this kernel (an array-of-structs slot needed two loads per probe: 1.6x slower).

// The table is a MULTIMAP: every query element takes its own slot (the first free one of its key's
// probe sequence), so a lookup needs no dependent chain loads - it walks the probe sequence up to
// the first free slot and accounts every slot that carries the key.  (A chained layout was tried:
// on a 64-lane wave some lane almost always has a chain to follow, and each hop is a dependent
// ~1 us L2 access.)
__global__ __launch_bounds__(kBlock) void ts_join_build_kernel(
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t Q,
    int32_t max_len, int32_t s_log2, unsigned long long *__restrict__ tkeys,
    uint32_t *__restrict__ tpack, uint32_t *__restrict__ tbloom, int32_t *__restrict__ hits_n) {
    const int q = blockIdx.y;
    const int64_t o = q_offsets[q];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int64_t len = q_offsets[q + 1] - o;
    if (len > max_len) {
        // max_query_len was not an upper bound (the table is sized from it): nothing of this
        // query is inserted and its counter is poisoned instead (stays negative)
        if (i == 0) hits_n[q] = INT32_MIN;
        return;
    }
    if (i >= (int)len) return;
    int64_t k;
    if (!canon_key(queries[o + i], k)) return;                     // NaN never matches
    const uint32_t smask = (1u << s_log2) - 1u;
    const size_t tb = (size_t)(q / kJoinQ) << s_log2;
    const uint32_t hv = hash32(k);
    const uint32_t bit = hv & (uint32_t)(kJoinBloomBits - 1);      // presence bit (low hash bits)
    atomicOr(&tbloom[(size_t)(q / kJoinQ) * (kJoinBloomBits / 32) + (bit >> 5)], 1u << (bit & 31));
    uint32_t h = (hv >> (32 - s_log2)) & ~1u;                      // home pair (high hash bits)
    while (true) {
        const unsigned long long old = atomicCAS(&tkeys[tb + h], (unsigned long long)kJEmpty,
                                                 (unsigned long long)k);
        if (old == (unsigned long long)kJEmpty) break;             // claimed a free slot
        h = (h + 1) & smask;
    }
    tpack[tb + h] = (uint32_t)i | ((uint32_t)(q % kJoinQ) << 12);
}

__global__ __launch_bounds__(kJoinBlock) void ts_match_join_kernel(
    const Row *__restrict__ rows, int64_t n_rows, const int64_t *__restrict__ keys,
    const int64_t *__restrict__ tkeys, const uint32_t *__restrict__ tpack,
    const uint32_t *__restrict__ tbloom, int32_t s_log2, int32_t Q, int32_t n_tiles, int32_t n_chunks,
    int32_t min_match, const int32_t *__restrict__ exclude_ids, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *m1_all = reinterpret_cast<uint32_t *>(smem);                     // [groups][128] smallest pos
    uint32_t *m2_all = m1_all + kJoinGroups * kJoinQ;                          // [groups][128] 2nd smallest
    uint32_t *cnt_all = m2_all + kJoinGroups * kJoinQ;                         // [groups][64] 2 x u16
    uint32_t *bloom = cnt_all + kJoinGroups * (kJoinQ / 2);                    // [2^19 bits]
    // block -> (tile, chunk): blocks of one tile share blockIdx % 8, i.e. (observed) one XCD and its L2
    const int b = blockIdx.x;
    int tile, chunk;
    if (8 % n_tiles == 0) {
        const int g = 8 / n_tiles;                                             // XCDs per tile
        tile = (b % 8) / g;
        chunk = (b / 8) * g + (b % 8) % g;
    } else if (n_tiles % 8 == 0) {
        tile = (b % 8) + 8 * ((b / 8) % (n_tiles / 8));
        chunk = (b / 8) / (n_tiles / 8);
    } else {
        tile = b / n_chunks;
        chunk = b % n_chunks;
    }
    if (tile >= n_tiles || chunk >= n_chunks) return;
    const int gl = threadIdx.x & (kGroup - 1);
    const int g = threadIdx.x / kGroup;
    uint32_t *m1 = m1_all + g * kJoinQ;
    uint32_t *m2 = m2_all + g * kJoinQ;
    uint32_t *cntw = cnt_all + g * (kJoinQ / 2);
    for (int i = gl; i < kJoinQ; i += kGroup) { m1[i] = 0xffffffffu; m2[i] = 0xffffffffu; }
    {   // the tile's presence bitmap: 64 KiB copied from device memory into LDS once per block
        const uint4 *src = reinterpret_cast<const uint4 *>(tbloom + (size_t)tile * (kJoinBloomBits / 32));
        uint4 *dst = reinterpret_cast<uint4 *>(bloom);
        for (int i = threadIdx.x; i < kJoinBloomBits / 128; i += kJoinBlock) dst[i] = src[i];
    }
    __syncthreads();
    for (int i = gl; i < kJoinQ / 2; i += kGroup) cntw[i] = 0;
    const int q0 = tile * kJoinQ;
    const uint32_t smask = (1u << s_log2) - 1u;
    const int64_t *tk = tkeys + ((size_t)tile << s_log2);
    const uint32_t *tp = tpack + ((size_t)tile << s_log2);
    const int64_t r0 = (int64_t)chunk * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n_rows) r1 = n_rows;

    auto account = [&](uint32_t pk) {                    // one matching (query, position) entry
        const uint32_t ql = pk >> 12, pos = pk & 0xfffu;
        atomicAdd(&cntw[ql >> 1], 1u << (16 * (ql & 1)));
        const uint32_t old = atomicMin(&m1[ql], pos);    // two plain LDS atomics, no CAS loop
        atomicMin(&m2[ql], old > pos ? old : pos);       // larger of two distinct hits >= 2nd smallest
    };

    constexpr int kK = 4;                                // keys per lane and step: 8 table loads in flight
    for (int64_t r = r0 + g; r < r1; r += kJoinGroups) {
        const Row row = rows[r];
        const int64_t *rk = keys + row.off;
        for (int i0 = gl * 2; i0 < row.len; i0 += kGroup * kK) {
            int64_t kk[kK];
            bool valid[kK];
#pragma unroll
            for (int j = 0; j < kK / 2; ++j) {
                const int i = i0 + j * kGroup * 2;
                longlong2 v = make_longlong2(0, 0);
                if (i < row.len) v = *reinterpret_cast<const longlong2 *>(rk + i);
                kk[2 * j] = v.x;
                kk[2 * j + 1] = v.y;
                valid[2 * j] = i < row.len;
                valid[2 * j + 1] = i + 1 < row.len;
            }
            uint32_t h[kK];
            longlong2 sk[kK];
#pragma unroll
            for (int j = 0; j < kK; ++j) {               // independent L2 reads, all in flight
                // LDS presence filter first: a random probe of the table costs a whole L2 line,
                // and ~70 % of the corpus keys are in no query of the tile
                const uint32_t hv = hash32(kk[j]);
                const uint32_t bit = hv & (uint32_t)(kJoinBloomBits - 1);
                h[j] = (hv >> (32 - s_log2)) & ~1u;
                sk[j] = make_longlong2(kJEmpty, kJEmpty);
                if (valid[j] && ((bloom[bit >> 5] >> (bit & 31)) & 1u))
                    sk[j] = *reinterpret_cast<const longlong2 *>(tk + h[j]);
            }
#pragma unroll
            for (int j = 0; j < kK; ++j) {
                if (!valid[j]) continue;
                if (sk[j].x == kk[j]) account(tp[h[j]]);
                if (sk[j].x == kJEmpty) continue;
                if (sk[j].y == kk[j]) account(tp[h[j] + 1]);
                if (sk[j].y == kJEmpty) continue;
                uint32_t hh = h[j];                      // home pair full: keep walking (rare)
                while (true) {
                    hh = (hh + 2) & smask;
                    const longlong2 a = *reinterpret_cast<const longlong2 *>(tk + hh);
                    if (a.x == kk[j]) account(tp[hh]);
                    if (a.x == kJEmpty) break;
                    if (a.y == kk[j]) account(tp[hh + 1]);
                    if (a.y == kJEmpty) break;
                }
            }
        }
        // lane gl owns queries gl*8 .. gl*8+7 of the tile (4 counter words of 2 x u16)
        const uint4 cw = *reinterpret_cast<const uint4 *>(cntw + gl * 4);
        if ((cw.x | cw.y | cw.z | cw.w) == 0 && min_match > 0) continue;      // nothing matched
        const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
        uint32_t todo = 0;                               // bit j: query gl*8+j reaches min_match
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = (int)((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
            todo |= (uint32_t)(c >= min_match && q0 + gl * 8 + j < Q) << j;
        }
        while (todo) {                                   // usually 0 or 1 iterations
            const int j = __ffs(todo) - 1;
            todo &= todo - 1;
            const int ql = gl * 8 + j;
            const int q = q0 + ql;
            const uint32_t c = (w[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            if (!(exclude_ids && exclude_ids[q] == row.vid)) {
                const int32_t kth = min_match <= 0 ? -1 : (int32_t)(min_match == 1 ? m1[ql] : m2[ql]);
                const int slot = atomicAdd(&hits_n[q], 1);
                if (slot < cap) {
                    int32_t *hp = hits + ((int64_t)q * cap + slot) * 3;
                    hp[0] = row.vid;
                    hp[1] = (int32_t)c;
                    hp[2] = kth;
                }
            }
        }
        // unconditional reset of this lane's 8 queries: 5 wide LDS stores, no per-query branches
        const uint4 ones = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        *reinterpret_cast<uint4 *>(cntw + gl * 4) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(m1 + gl * 8) = ones;
        *reinterpret_cast<uint4 *>(m1 + gl * 8 + 4) = ones;
        *reinterpret_cast<uint4 *>(m2 + gl * 8) = ones;
        *reinterpret_cast<uint4 *>(m2 + gl * 8 + 4) = ones;
    }
}

// ---- queries longer than a tile (> 4095 timestamps): counts by searching the SORTED query ----
// Rare (a video with thousands of cuts), so simple beats fast: a 16-lane group owns a row, every
// row key is binary-searched in the query's sorted distinct keys (sq, with multiplicities) and the
// hit (video_id, count) is emitted with kth = -2 - row, which ts_kth_fixup_kernel resolves.
template <int CTRL>
__device__ __forceinline__ int dpp_row16(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

__global__ __launch_bounds__(kBlock) void ts_match_longq_kernel(
    const Row *__restrict__ rows, int64_t n_rows, const int64_t *__restrict__ keys,
    const int64_t *__restrict__ sq, const int32_t *__restrict__ smult, int32_t m, int32_t min_match,
    int32_t cap, int32_t *__restrict__ hits, int32_t *__restrict__ hits_n) {
    const int gl = threadIdx.x & (kGroup - 1);
    const int64_t r = (int64_t)blockIdx.x * kGroupsPerBlock + threadIdx.x / kGroup;
    if (r >= n_rows) return;                       // whole 16-lane groups leave together
    const Row row = rows[r];
    const int64_t *rk = keys + row.off;
    int cnt = 0;
    for (int i = gl; i < row.len; i += kGroup) {
        const int64_t k = rk[i];
        int lo = 0, hi = m;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sq[mid] < k) lo = mid + 1; else hi = mid;
        }
        if (lo < m && sq[lo] == k) cnt += smult[lo];
    }
    cnt += dpp_row16<0xB1>(cnt);    // quad_perm [1,0,3,2]
    cnt += dpp_row16<0x4E>(cnt);    // quad_perm [2,3,0,1]
    cnt += dpp_row16<0x141>(cnt);   // row_half_mirror
    cnt += dpp_row16<0x140>(cnt);   // row_mirror
    if (gl == 0 && cnt >= min_match) {
        const int slot = atomicAdd(&hits_n[0], 1);
        if (slot < cap) {
            hits[slot * 3 + 0] = row.vid;
            hits[slot * 3 + 1] = cnt;
            hits[slot * 3 + 2] = (min_match <= 0) ? -1 : -2 - (int32_t)r;
        }
    }
}

// kth for min_match > 5: per stored hit, walk the query in order and binary-search the row.
__global__ __launch_bounds__(kBlock) void ts_kth_fixup_kernel(
    const Row *__restrict__ rows, const int64_t *__restrict__ keys,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t min_match,
    int32_t cap, int32_t *__restrict__ hits, const int32_t *__restrict__ hits_n) {
    const int q = blockIdx.x;
    const int gl = threadIdx.x & (kGroup - 1);
    const int g = threadIdx.x / kGroup;
    const int gshift = (threadIdx.x & 63) & ~(kGroup - 1);
    int n = hits_n[q];
    if (n > cap) n = cap;
    const int64_t qo = q_offsets[q];
    const int32_t qlen = (int32_t)(q_offsets[q + 1] - qo);
    const double *qv = queries + qo;
    for (int j = g; j < n; j += kGroupsPerBlock) {
        int32_t *h = hits + ((int64_t)q * cap + j) * 3;
        const int32_t code = h[2];
        if (code > -2) continue;
        const Row row = rows[-2 - code];
        const int64_t *rk = keys + row.off;
        int kth = TVZ_KTH_NEVER;
        int running = 0;
        for (int base = 0; base < qlen && kth == TVZ_KTH_NEVER; base += kGroup) {
            const int i = base + gl;
            bool hit = false;
            int64_t k;
            if (i < qlen && canon_key(qv[i], k)) {
                int lo = 0, hi = row.len;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (rk[mid] < k) lo = mid + 1; else hi = mid;
                }
                hit = lo < row.len && rk[lo] == k;
            }
            const uint32_t m16 = (uint32_t)(__ballot(hit) >> gshift) & 0xffffu;
            const int c = __popc(m16);
            if (running + c >= min_match) {
                uint32_t m = m16;
                for (int need = min_match - running; need > 1; --need) m &= m - 1;
                kth = base + (__ffs(m) - 1);
            }
            running += c;
        }
        if (gl == 0) h[2] = kth;
    }
}

// ---------------------------------------------------------------- top-k
constexpr int kSortCap = 2048;

__device__ __forceinline__ uint64_t sort_key(int32_t vid, int32_t kth) {
    return ((uint64_t)(uint32_t)(kth + 1) << 32) | (uint32_t)vid;
}

__device__ void bitonic_sort(uint64_t *key, int32_t *cnt, int n /* power of two */) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n / 2; i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = key[lo], b = key[hi];
                const int32_t ca = cnt[lo], cb = cnt[hi];
                const bool gt = (a > b) || (a == b && ca > cb);
                if (gt == up) {
                    key[lo] = b; key[hi] = a;
                    cnt[lo] = cb; cnt[hi] = ca;
                }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void ts_topk_kernel(const int32_t *__restrict__ lists,
                                                         const int32_t *__restrict__ lists_n,
                                                         int32_t n_lists, int32_t Q, int32_t cap,
                                                         int32_t k, int32_t *__restrict__ topk,
                                                         int32_t mode, int32_t *__restrict__ totals) {
    // mode 0: plain.  mode 1 (shard side): the output has k+1 rows per query, row k carries the
    // true number of hits as (-1, n, NEVER) so one all-gather moves lists and totals together; n is
    // NEGATED when the shard's hit list overflowed its capacity (its top-k may then be inexact).
    // mode 2 (merge side): every input list ends with such a row; |n| is summed into totals[q], and
    // the sum is negated if any shard overflowed, so the caller knows to re-run with a larger cap.
    __shared__ uint64_t key[kSortCap];
    __shared__ int32_t cnt[kSortCap];
    const int q = blockIdx.x;
    int pos = 0;  // block-uniform fill level
    long long total = 0;
    bool overflow = false;
    auto sort_and_keep = [&]() {
        int P = 2;
        while (P < pos) P <<= 1;
        for (int i = pos + threadIdx.x; i < P; i += kBlock) { key[i] = ~0ULL; cnt[i] = 0; }
        bitonic_sort(key, cnt, P);
        if (pos > k) pos = k;
    };
    for (int l = 0; l < n_lists; ++l) {
        int n = lists_n ? lists_n[(int64_t)l * Q + q] : cap;
        const int32_t *src = lists + ((int64_t)l * Q + q) * (int64_t)cap * 3;
        if (mode == 1) {
            total += n;
            if (n > cap) overflow = true;          // this shard's list was truncated
        }
        if (n > cap) n = cap;
        if (mode == 2) {
            n = cap - 1;
            const int32_t t = src[(cap - 1) * 3 + 1];   // negative: that shard overflowed
            total += t < 0 ? -(long long)t : t;
            if (t < 0) overflow = true;
        }
        int j = 0;
        while (j < n) {
            int m = n - j;
            if (m > kSortCap - pos) m = kSortCap - pos;
            for (int i = threadIdx.x; i < m; i += kBlock) {
                const int32_t vid = src[(j + i) * 3 + 0];
                key[pos + i] = vid < 0 ? ~0ULL : sort_key(vid, src[(j + i) * 3 + 2]);
                cnt[pos + i] = src[(j + i) * 3 + 1];
            }
            pos += m;
            j += m;
            __syncthreads();
            if (pos == kSortCap) sort_and_keep();
        }
    }
    __syncthreads();
    sort_and_keep();
    const int orows = (mode == 1) ? k + 1 : k;
    if (threadIdx.x == 0) {
        int32_t t = total > 0x7fffffffLL ? 0x7fffffff : (int32_t)total;
        if (overflow) t = (t == 0) ? INT32_MIN : -t;   // negative total = some hit list was truncated
        if (mode == 1) {
            int32_t *o = topk + ((int64_t)q * orows + k) * 3;
            o[0] = -1; o[1] = t; o[2] = TVZ_KTH_NEVER;
        } else if (mode == 2 && totals) {
            totals[q] = t;
        }
    }
    for (int i = threadIdx.x; i < k; i += kBlock) {
        int32_t *o = topk + ((int64_t)q * orows + i) * 3;
        const uint64_t kk = (i < pos) ? key[i] : ~0ULL;
        if (kk == ~0ULL) {
            o[0] = -1; o[1] = 0; o[2] = TVZ_KTH_NEVER;
        } else {
            o[0] = (int32_t)(uint32_t)kk;
            o[1] = cnt[i];
            o[2] = (int32_t)(uint32_t)(kk >> 32) - 1;
        }
    }
}

// ---------------------------------------------------------------- opt-in alignment score
// NOT the reference's verdict (db.py:79 is exact-only); north_star's "alignment/Jaccard" and the
// stale README.md:291 ("0.1 s tolerance") ask for a shift/tolerance-aware score, reported alongside.
// One wave per row: every (query_i, row_j) difference votes into an LDS histogram of bins of
// width eps over [-max_offset, +max_offset]; output = best bin (ties: smaller |bin|, then the
// negative one), its votes, and the votes of bin 0 (tolerant count without shift).
constexpr int kAlignMaxBins = 4096;   // 16 KB of u32 per wave, 4 waves per block

__global__ __launch_bounds__(kBlock) void ts_align_kernel(
    const Row *__restrict__ rows, int64_t n_rows, const int64_t *__restrict__ keys,
    const double *__restrict__ query, int32_t n, double eps, int32_t B,
    int32_t *__restrict__ out) {
    __shared__ uint32_t hist_all[kBlock / 64][kAlignMaxBins];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t *hist = hist_all[wave];
    const int nbins = 2 * B + 1;
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + wave; r < n_rows;
         r += (int64_t)gridDim.x * (kBlock / 64)) {
        const Row row = rows[r];
        for (int b = lane; b < nbins; b += 64) hist[b] = 0;
        const int64_t *rk = keys + row.off;
        for (int j = lane; j < row.len; j += 64) {
            const double c = __longlong_as_double(rk[j]);
            for (int i = 0; i < n; ++i) {
                const double q = query[i];
                if (q != q) continue;                              // NaN never aligns
                const double d = floor((c - q) / eps + 0.5);
                if (d >= -(double)B && d <= (double)B) atomicAdd(&hist[(int)d + B], 1u);
            }
        }
        // LDS ops of one wave complete in order: the votes above are visible to the scan below
        unsigned long long best = 0;
        for (int b = lane; b < nbins; b += 64) {
            const int bin = b - B;
            const uint32_t order = 2u * (uint32_t)(bin < 0 ? -bin : bin) + (bin > 0 ? 1u : 0u);
            const unsigned long long key = ((unsigned long long)hist[b] << 14) | (16383u - order);
            best = key > best ? key : best;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off);
            best = o > best ? o : best;
        }
        if (lane == 0) {
            const uint32_t order = 16383u - (uint32_t)(best & 16383u);
            const int mag = (int)(order >> 1);
            int32_t *o = out + r * 5;
            o[0] = row.vid;
            o[1] = row.len;
            o[2] = (order & 1u) ? mag : -mag;
            o[3] = (int32_t)(best >> 14);
            o[4] = (int32_t)hist[B];
        }
    }
}

// ---------------------------------------------------------------- host side
struct Staging {
    hipStream_t stream = nullptr;
    double *d_query = nullptr;   int64_t query_cap = 0;
    int64_t *d_qoff = nullptr;
    int32_t *d_hits = nullptr;   int64_t hits_cap = 0;
    int32_t *d_hits_n = nullptr;
    int32_t *h_hits = nullptr;   // pinned, hits_cap entries
    int64_t *d_sq = nullptr;     int32_t *d_smult = nullptr;  int64_t sq_cap = 0;  // long queries
    int64_t *h_small = nullptr;                          // pinned: qoff[2] + hits_n
};

// device tables of one hash join; reusable once `done` has completed
struct JoinWs {
    unsigned char *base = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    bool busy = false;
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    int64_t cap = 0;
};

}  // namespace

struct tvz_corpus {
    int device = 0;
    std::shared_mutex mu;        // exclusive: mutation; shared: enqueueing a match
    std::mutex ev_mu;
    std::mutex stage_mu;
    DevBuf<int64_t> keys;
    DevBuf<Row> rows;
    std::vector<int64_t> h_keys;  // host mirror of the arena (for compaction)
    std::vector<Row> h_rows;
    std::unordered_map<int32_t, int64_t> first_row;  // video_id -> first row index
    int64_t live_keys = 0;
    static constexpr int kEvents = 32;
    hipEvent_t events[kEvents] = {};
    bool ev_pending[kEvents] = {};
    int ev_next = 0;
    std::vector<Staging *> free_staging;
    std::mutex join_mu;
    std::vector<JoinWs *> join_ws;   // tables of in-flight / reusable hash joins
};

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// canonical, sorted, unique keys of one row appended to `out` (padded to an even count)
int64_t canon_row(const double *src, int64_t n, std::vector<int64_t> &out) {
    const size_t start = out.size();
    for (int64_t i = 0; i < n; ++i) {
        int64_t k;
        if (canon_key(src[i], k)) out.push_back(k);
    }
    std::sort(out.begin() + start, out.end());
    out.erase(std::unique(out.begin() + start, out.end()), out.end());
    const int64_t len = (int64_t)(out.size() - start);
    if (out.size() & 1) out.push_back(kEmpty);  // keep every row 16-byte aligned
    return len;
}

template <typename T>
int ensure(DevBuf<T> &b, int64_t need, int64_t keep) {
    if (need <= b.cap) return TVZ_OK;
    int64_t cap = std::max<int64_t>(need, b.cap * 2);
    cap = std::max<int64_t>(cap, 1024);
    T *np = nullptr;
    if (hipMalloc(&np, (size_t)cap * sizeof(T)) != hipSuccess)
        return tvz::fail(TVZ_ERR_NOMEM, "hipMalloc of %lld bytes failed",
                         (long long)(cap * (int64_t)sizeof(T)));
    if (b.p && keep > 0) TVZ_HIP(hipMemcpy(np, b.p, (size_t)keep * sizeof(T), hipMemcpyDeviceToDevice));
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.cap = cap;
    return TVZ_OK;
}

// wait for every match kernel enqueued so far (caller holds mu exclusively)
int drain(tvz_corpus *c) {
    std::lock_guard<std::mutex> lk(c->ev_mu);
    for (int i = 0; i < tvz_corpus::kEvents; ++i)
        if (c->ev_pending[i]) {
            TVZ_HIP(hipEventSynchronize(c->events[i]));
            c->ev_pending[i] = false;
        }
    return TVZ_OK;
}

int record(tvz_corpus *c, hipStream_t st) {
    std::lock_guard<std::mutex> lk(c->ev_mu);
    const int i = c->ev_next;
    c->ev_next = (i + 1) % tvz_corpus::kEvents;
    if (c->ev_pending[i]) TVZ_HIP(hipEventSynchronize(c->events[i]));
    TVZ_HIP(hipEventRecord(c->events[i], st));
    c->ev_pending[i] = true;
    return TVZ_OK;
}

int upload_all(tvz_corpus *c) {
    if (int rc = ensure(c->keys, (int64_t)c->h_keys.size() + 2, 0)) return rc;
    if (int rc = ensure(c->rows, (int64_t)c->h_rows.size() + 1, 0)) return rc;
    if (!c->h_keys.empty())
        TVZ_HIP(hipMemcpy(c->keys.p, c->h_keys.data(), c->h_keys.size() * 8, hipMemcpyHostToDevice));
    if (!c->h_rows.empty())
        TVZ_HIP(hipMemcpy(c->rows.p, c->h_rows.data(), c->h_rows.size() * sizeof(Row),
                          hipMemcpyHostToDevice));
    return TVZ_OK;
}

int compact(tvz_corpus *c) {
    std::vector<int64_t> nk;
    nk.reserve((size_t)c->live_keys + c->h_rows.size());
    for (Row &r : c->h_rows) {
        const int64_t off = (int64_t)nk.size();
        nk.insert(nk.end(), c->h_keys.begin() + r.off, c->h_keys.begin() + r.off + r.len);
        if (nk.size() & 1) nk.push_back(kEmpty);
        r.off = off;
    }
    c->h_keys.swap(nk);
    return upload_all(c);
}

// Dispatch (measured A/B grid, profiles/r1_match_join_ab.txt): the join wins once its fixed cost
// (table memset + build, ~30 us) is amortised, i.e. from about 5 M (query, row) pairs per batch:
// C=100k x Q=1024: 1.18 ms vs 2.30 ms for the LDS tile kernel, C=5k x Q=1024: 0.138 vs 0.165,
// C=20k x Q=256: 0.177 vs 0.198; below that (C=5k x Q=256: 0.109 vs 0.076) the tile kernel wins.
constexpr int kJoinMinQ = 64;
constexpr int64_t kJoinMinPairs = 5000000;
int g_use_join = 1;             // 0 = never, 1 = by the rule above, 2 = whenever legal (A/B knob)

int join_ws_get(tvz_corpus *c, size_t bytes, JoinWs **out) {
    std::lock_guard<std::mutex> lk(c->join_mu);
    for (JoinWs *w : c->join_ws)
        if (!w->busy || hipEventQuery(w->done) == hipSuccess) {
            w->busy = true;
            if (w->bytes < bytes) {
                if (w->base) (void)hipFree(w->base);
                w->base = nullptr;
                TVZ_HIP(hipMalloc(&w->base, bytes));
                w->bytes = bytes;
            }
            *out = w;
            return TVZ_OK;
        }
    JoinWs *w = new JoinWs();
    TVZ_HIP(hipEventCreateWithFlags(&w->done, hipEventDisableTiming));
    TVZ_HIP(hipMalloc(&w->base, bytes));
    w->bytes = bytes;
    w->busy = true;
    c->join_ws.push_back(w);
    *out = w;
    return TVZ_OK;
}

int launch_join(tvz_corpus *c, const double *d_queries, const int64_t *d_q_offsets, int32_t Q,
                int32_t max_query_len, int32_t min_match, const int32_t *d_exclude_ids,
                int32_t cap, int32_t *d_hits, int32_t *d_hits_n, hipStream_t st) {
    const int64_t n_rows = (int64_t)c->h_rows.size();
    const int n_tiles = (int)tvz::ceil_div(Q, kJoinQ);
    // slots per tile: load factor <= 0.5 (typically ~0.25) AND small enough to live in one XCD's
    // 4 MiB L2 next to the streamed corpus (2 MiB at max_query_len <= 512); a table of 4 MiB was
    // measured at the Infinity-Cache random-line rate (8.5 TB/s) instead of the L2's
    int s_log2 = 10;
    while (((int64_t)1 << s_log2) < (int64_t)2 * kJoinQ * max_query_len) ++s_log2;
    const size_t S = (size_t)1 << s_log2;
    const size_t b_keys = (size_t)n_tiles * S * 8, b_pack = (size_t)n_tiles * S * 4;
    const size_t b_bloom = (size_t)n_tiles * (kJoinBloomBits / 8);
    JoinWs *ws = nullptr;
    if (int rc = join_ws_get(c, b_keys + b_pack + b_bloom, &ws)) return rc;
    unsigned long long *tkeys = reinterpret_cast<unsigned long long *>(ws->base);
    uint32_t *tpack = reinterpret_cast<uint32_t *>(ws->base + b_keys);
    uint32_t *tbloom = reinterpret_cast<uint32_t *>(ws->base + b_keys + b_pack);
    TVZ_HIP(hipMemsetAsync(ws->base, 0xff, b_keys, st));    // every key = kJEmpty
    TVZ_HIP(hipMemsetAsync(tbloom, 0, b_bloom, st));
    hipLaunchKernelGGL(ts_join_build_kernel, dim3((unsigned)tvz::ceil_div(max_query_len, kBlock), (unsigned)Q),
                       dim3(kBlock), 0, st, d_queries, d_q_offsets, Q, max_query_len, s_log2, tkeys, tpack,
                       tbloom, d_hits_n);
    TVZ_HIP(hipGetLastError());
    // two 1024-thread blocks per CU: about two rounds of 512 blocks, >= 8 rows per 16-lane group
    const int64_t g = (8 % n_tiles == 0) ? 8 / n_tiles : 1;
    int64_t chunks = std::max<int64_t>(1, 1024 / n_tiles);
    chunks = std::min(chunks, std::max<int64_t>(1, n_rows / (2 * kJoinGroups)));
    chunks = tvz::round_up(chunks, g);
    const int64_t rpb = tvz::round_up(tvz::ceil_div(n_rows, chunks), kJoinGroups);
    int64_t blocks = (int64_t)n_tiles * chunks;
    if (8 % n_tiles == 0) blocks = tvz::round_up(blocks, 8);
    hipLaunchKernelGGL(ts_match_join_kernel, dim3((unsigned)blocks), dim3(kJoinBlock), kJoinLds, st,
                       c->rows.p, n_rows, c->keys.p, reinterpret_cast<const int64_t *>(tkeys), tpack, tbloom,
                       s_log2, Q, n_tiles, (int32_t)chunks, min_match, d_exclude_ids, cap, d_hits,
                       d_hits_n, (int32_t)rpb);
    TVZ_HIP(hipGetLastError());
    TVZ_HIP(hipEventRecord(ws->done, st));                  // the tables are free again after this
    return TVZ_OK;
}

int launch_match(tvz_corpus *c, const double *d_queries, const int64_t *d_q_offsets, int32_t Q,
                 int32_t max_query_len, int32_t min_match, const int32_t *d_exclude_ids,
                 int32_t cap, int32_t *d_hits, int32_t *d_hits_n, hipStream_t st) {
    if (max_query_len > kMaxQueryLen)
        return tvz::fail(TVZ_ERR_UNSUPPORTED, "query of %d timestamps exceeds the supported %d",
                         max_query_len, kMaxQueryLen);
    TVZ_HIP(hipMemsetAsync(d_hits_n, 0, (size_t)Q * sizeof(int32_t), st));
    const int64_t n_rows = (int64_t)c->h_rows.size();
    if (n_rows == 0 || Q == 0) return TVZ_OK;
    const bool join_legal = min_match <= 2 && max_query_len > 0;
    if (join_legal && (g_use_join == 2 || (g_use_join == 1 && Q >= kJoinMinQ && (int64_t)Q * n_rows >= kJoinMinPairs)))
        return launch_join(c, d_queries, d_q_offsets, Q, max_query_len, min_match, d_exclude_ids, cap,
                           d_hits, d_hits_n, st);
    // queries per tile: as many as keep the shared table at load <= 0.5 (at most 16)
    int nq = max_query_len > 0 ? kTileMaxEntries / max_query_len : kTileQ;
    nq = std::max(1, std::min(nq, kTileQ));
    const int64_t tiles = tvz::ceil_div(Q, nq);
    // one 1024-thread block per CU (LDS): aim at just under two full rounds of 256 blocks (a
    // third, mostly empty round costs a whole block time), but >= 2 rows per 16-lane group so
    // that building the tile's table (per block) stays a small part of the block's life
    int64_t chunks = std::max<int64_t>(1, 512 / tiles);
    chunks = std::min(chunks, std::max<int64_t>(1, n_rows / (2 * kTileGroups)));
    const int64_t rpb = tvz::round_up(tvz::ceil_div(n_rows, chunks), kTileGroups);
    chunks = tvz::ceil_div(n_rows, rpb);
    if (tiles > 65535)
        return tvz::fail(TVZ_ERR_UNSUPPORTED, "too many query tiles (%lld)", (long long)tiles);
    if (min_match <= 2)
        hipLaunchKernelGGL(ts_match_tile_kernel<false>, dim3((unsigned)chunks, (unsigned)tiles),
                           dim3(kTileBlock), kTileLds, st, c->rows.p, n_rows, c->keys.p, d_queries,
                           d_q_offsets, Q, nq, min_match, d_exclude_ids, cap, d_hits, d_hits_n,
                           (int32_t)rpb);
    else
        hipLaunchKernelGGL(ts_match_tile_kernel<true>, dim3((unsigned)chunks, (unsigned)tiles),
                           dim3(kTileBlock), kTileLds, st, c->rows.p, n_rows, c->keys.p, d_queries,
                           d_q_offsets, Q, nq, min_match, d_exclude_ids, cap, d_hits, d_hits_n,
                           (int32_t)rpb);
    TVZ_HIP(hipGetLastError());
    if (min_match > kTop) {
        hipLaunchKernelGGL(ts_kth_fixup_kernel, dim3((unsigned)Q), dim3(kBlock), 0, st, c->rows.p,
                           c->keys.p, d_queries, d_q_offsets, min_match, cap, d_hits, d_hits_n);
        TVZ_HIP(hipGetLastError());
    }
    return TVZ_OK;
}

int staging_get(tvz_corpus *c, Staging **out) {
    {
        std::lock_guard<std::mutex> lk(c->stage_mu);
        if (!c->free_staging.empty()) {
            *out = c->free_staging.back();
            c->free_staging.pop_back();
            return TVZ_OK;
        }
    }
    Staging *s = new Staging();
    TVZ_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    TVZ_HIP(hipMalloc(&s->d_qoff, 2 * sizeof(int64_t)));
    TVZ_HIP(hipMalloc(&s->d_hits_n, sizeof(int32_t)));
    TVZ_HIP(hipHostMalloc(&s->h_small, 4 * sizeof(int64_t)));
    *out = s;
    return TVZ_OK;
}

void staging_put(tvz_corpus *c, Staging *s) {
    std::lock_guard<std::mutex> lk(c->stage_mu);
    c->free_staging.push_back(s);
}

void staging_free(Staging *s) {
    if (s->d_query) (void)hipFree(s->d_query);
    if (s->d_qoff) (void)hipFree(s->d_qoff);
    if (s->d_hits) (void)hipFree(s->d_hits);
    if (s->d_hits_n) (void)hipFree(s->d_hits_n);
    if (s->h_hits) (void)hipHostFree(s->h_hits);
    if (s->h_small) (void)hipHostFree(s->h_small);
    if (s->d_sq) (void)hipFree(s->d_sq);
    if (s->d_smult) (void)hipFree(s->d_smult);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

}  // namespace

static int tvz_corpus_create_impl(tvz_corpus **out, int device) {
    TVZ_REQUIRE(out != nullptr, "out is NULL");
    int n = 0;
    TVZ_HIP(hipGetDeviceCount(&n));
    TVZ_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    DeviceGuard dg(device);
    tvz_corpus *c = new tvz_corpus();
    c->device = device;
    for (int i = 0; i < tvz_corpus::kEvents; ++i)
        TVZ_HIP(hipEventCreateWithFlags(&c->events[i], hipEventDisableTiming));
    TVZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ts_match_tile_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileLds));
    TVZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ts_match_tile_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileLds));
    TVZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ts_match_join_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kJoinLds));
    if (int rc = upload_all(c)) { delete c; return rc; }
    *out = c;
    return TVZ_OK;
}

static int tvz_corpus_destroy_impl(tvz_corpus *c) {
    if (!c) return TVZ_OK;
    DeviceGuard dg(c->device);
    {
        std::unique_lock<std::shared_mutex> lk(c->mu);
        (void)drain(c);
        for (Staging *s : c->free_staging) staging_free(s);
        c->free_staging.clear();
        for (JoinWs *w : c->join_ws) {
            if (w->done) { (void)hipEventSynchronize(w->done); (void)hipEventDestroy(w->done); }
            if (w->base) (void)hipFree(w->base);
            delete w;
        }
        c->join_ws.clear();
        if (c->keys.p) (void)hipFree(c->keys.p);
        if (c->rows.p) (void)hipFree(c->rows.p);
        for (int i = 0; i < tvz_corpus::kEvents; ++i)
            if (c->events[i]) (void)hipEventDestroy(c->events[i]);
    }
    delete c;
    return TVZ_OK;
}

static int tvz_corpus_upload_impl(tvz_corpus *c, const int32_t *h_video_ids,
                                 const int64_t *h_offsets, const double *h_keys, int64_t n_rows,
                                 int64_t n_keys) {
    TVZ_REQUIRE(c != nullptr, "corpus is NULL");
    TVZ_REQUIRE(n_rows >= 0 && n_keys >= 0, "negative size");
    TVZ_REQUIRE(n_rows == 0 || (h_video_ids && h_offsets), "NULL row arrays");
    TVZ_REQUIRE(n_keys == 0 || h_keys, "NULL keys");
    for (int64_t r = 0; r < n_rows; ++r) {
        TVZ_REQUIRE(h_offsets[r] <= h_offsets[r + 1] && h_offsets[r] >= 0 &&
                        h_offsets[r + 1] <= n_keys,
                    "offsets of row %lld are not monotone within [0, n_keys]", (long long)r);
        TVZ_REQUIRE(h_offsets[r + 1] - h_offsets[r] <= INT32_MAX, "row %lld too long", (long long)r);
    }
    DeviceGuard dg(c->device);
    std::unique_lock<std::shared_mutex> lk(c->mu);
    if (int rc = drain(c)) return rc;
    c->h_keys.clear();
    c->h_rows.clear();
    c->first_row.clear();
    c->h_keys.reserve((size_t)n_keys + (size_t)n_rows);
    c->h_rows.reserve((size_t)n_rows);
    c->live_keys = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        Row row;
        row.off = (int64_t)c->h_keys.size();
        row.len = (int32_t)canon_row(h_keys + h_offsets[r], h_offsets[r + 1] - h_offsets[r], c->h_keys);
        row.vid = h_video_ids[r];
        c->first_row.emplace(row.vid, r);
        c->h_rows.push_back(row);
        c->live_keys += row.len;
    }
    return upload_all(c);
}

static int tvz_corpus_upsert_impl(tvz_corpus *c, int32_t video_id, const double *h_keys, int64_t n) {
    TVZ_REQUIRE(c != nullptr, "corpus is NULL");
    TVZ_REQUIRE(n >= 0 && n <= INT32_MAX && (n == 0 || h_keys), "bad key array");
    DeviceGuard dg(c->device);
    std::unique_lock<std::shared_mutex> lk(c->mu);
    if (int rc = drain(c)) return rc;
    const int64_t off = (int64_t)c->h_keys.size();
    const int64_t len = canon_row(h_keys, n, c->h_keys);
    const int64_t added = (int64_t)c->h_keys.size() - off;
    auto it = c->first_row.find(video_id);
    const int64_t rows_before = (int64_t)c->h_rows.size();
    int64_t r;
    if (it == c->first_row.end()) {
        r = rows_before;
        c->h_rows.push_back(Row{off, (int32_t)len, video_id});
        c->first_row.emplace(video_id, r);
    } else {
        r = it->second;
        c->live_keys -= c->h_rows[r].len;
        c->h_rows[r].off = off;
        c->h_rows[r].len = (int32_t)len;
    }
    c->live_keys += len;
    // garbage-collect the arena when more than half of it is dead
    if ((int64_t)c->h_keys.size() > 2 * (c->live_keys + (int64_t)c->h_rows.size()) + 4096)
        return compact(c);
    if (int rc = ensure(c->keys, (int64_t)c->h_keys.size() + 2, off)) return rc;
    if (int rc = ensure(c->rows, (int64_t)c->h_rows.size() + 1, rows_before)) return rc;
    if (added)
        TVZ_HIP(hipMemcpy(c->keys.p + off, c->h_keys.data() + off, (size_t)added * 8,
                          hipMemcpyHostToDevice));
    TVZ_HIP(hipMemcpy(c->rows.p + r, &c->h_rows[r], sizeof(Row), hipMemcpyHostToDevice));
    return TVZ_OK;
}

static int tvz_corpus_clear_impl(tvz_corpus *c) {
    TVZ_REQUIRE(c != nullptr, "corpus is NULL");
    DeviceGuard dg(c->device);
    std::unique_lock<std::shared_mutex> lk(c->mu);
    if (int rc = drain(c)) return rc;
    c->h_keys.clear();
    c->h_rows.clear();
    c->first_row.clear();
    c->live_keys = 0;
    return TVZ_OK;
}

static int tvz_corpus_stats_impl(tvz_corpus *c, int64_t *n_rows, int64_t *n_keys,
                                int64_t *arena_keys) {
    TVZ_REQUIRE(c != nullptr, "corpus is NULL");
    std::shared_lock<std::shared_mutex> lk(c->mu);
    if (n_rows) *n_rows = (int64_t)c->h_rows.size();
    if (n_keys) *n_keys = c->live_keys;
    if (arena_keys) *arena_keys = (int64_t)c->h_keys.size();
    return TVZ_OK;
}

static int tvz_match_impl(tvz_corpus *c, const double *d_queries, const int64_t *d_q_offsets,
                         int32_t Q, int32_t max_query_len, int32_t min_match,
                         const int32_t *d_exclude_ids, int32_t cap, int32_t *d_hits,
                         int32_t *d_hits_n, void *hip_stream) {
    TVZ_REQUIRE(c != nullptr, "corpus is NULL");
    TVZ_REQUIRE(Q >= 0 && Q <= 65535, "Q=%d out of range [0, 65535]", Q);
    TVZ_REQUIRE(max_query_len >= 0 && cap >= 0, "negative size");
    if (Q == 0) return TVZ_OK;
    TVZ_REQUIRE(d_q_offsets && d_hits_n && (cap == 0 || d_hits), "NULL output / offsets");
    TVZ_REQUIRE(d_queries || max_query_len == 0, "d_queries is NULL");
    DeviceGuard dg(c->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    std::shared_lock<std::shared_mutex> lk(c->mu);
    if (int rc = launch_match(c, d_queries, d_q_offsets, Q, max_query_len, min_match,
                              d_exclude_ids, cap, d_hits, d_hits_n, st))
        return rc;
    return record(c, st);
}

static int tvz_find_duplicates_impl(tvz_corpus *c, const double *h_query, int64_t n,
                                   int32_t min_match, int32_t exclude_id, int64_t cap,
                                   int32_t *h_out_ids, int32_t *h_out_counts, int32_t *h_out_kth,
                                   int64_t *n_out) {
    TVZ_REQUIRE(c != nullptr && n_out != nullptr, "NULL argument");
    TVZ_REQUIRE(n >= 0 && cap >= 0 && cap <= INT32_MAX, "bad size");
    TVZ_REQUIRE(n == 0 || h_query, "h_query is NULL");
    TVZ_REQUIRE(cap == 0 || (h_out_ids && h_out_counts), "NULL outputs");
    TVZ_REQUIRE(n <= INT32_MAX, "query too long");
    DeviceGuard dg(c->device);
    Staging *s = nullptr;
    if (int rc = staging_get(c, &s)) return rc;
    struct Put { tvz_corpus *c; Staging *s; ~Put() { staging_put(c, s); } } put{c, s};
    if (n > s->query_cap) {
        if (s->d_query) (void)hipFree(s->d_query);
        s->query_cap = std::max<int64_t>(n, 256);
        TVZ_HIP(hipMalloc(&s->d_query, (size_t)s->query_cap * 8));
    }
    if (cap > s->hits_cap) {
        if (s->d_hits) (void)hipFree(s->d_hits);
        if (s->h_hits) (void)hipHostFree(s->h_hits);
        s->hits_cap = std::max<int64_t>(cap, 1024);
        TVZ_HIP(hipMalloc(&s->d_hits, (size_t)s->hits_cap * 12));
        TVZ_HIP(hipHostMalloc(&s->h_hits, (size_t)s->hits_cap * 12));
    }
    s->h_small[0] = 0;
    s->h_small[1] = n;
    TVZ_HIP(hipMemcpyAsync(s->d_qoff, s->h_small, 16, hipMemcpyHostToDevice, s->stream));
    if (n) TVZ_HIP(hipMemcpyAsync(s->d_query, h_query, (size_t)n * 8, hipMemcpyHostToDevice, s->stream));
    int32_t *d_excl = nullptr;  // exclusion is applied on the host for the single-query form
    if (n <= kMaxQueryLen) {
        std::shared_lock<std::shared_mutex> lk(c->mu);
        if (int rc = launch_match(c, s->d_query, s->d_qoff, 1, (int32_t)n, min_match, d_excl,
                                  (int32_t)cap, s->d_hits, s->d_hits_n, s->stream))
            return rc;
        if (int rc = record(c, s->stream)) return rc;
    } else {
        // longer than a query tile: sorted distinct keys + multiplicities, searched per row key
        std::vector<int64_t> sk;
        sk.reserve((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            int64_t k;
            if (canon_key(h_query[i], k)) sk.push_back(k);
        }
        std::sort(sk.begin(), sk.end());
        std::vector<int64_t> uq;
        std::vector<int32_t> mult;
        for (size_t i = 0; i < sk.size(); ++i) {
            if (!uq.empty() && uq.back() == sk[i]) ++mult.back();
            else { uq.push_back(sk[i]); mult.push_back(1); }
        }
        const int64_t m = (int64_t)uq.size();
        if (m + 1 > s->sq_cap) {
            if (s->d_sq) (void)hipFree(s->d_sq);
            if (s->d_smult) (void)hipFree(s->d_smult);
            s->sq_cap = m + 1;
            TVZ_HIP(hipMalloc(&s->d_sq, (size_t)s->sq_cap * 8));
            TVZ_HIP(hipMalloc(&s->d_smult, (size_t)s->sq_cap * 4));
        }
        if (m) {
            TVZ_HIP(hipMemcpyAsync(s->d_sq, uq.data(), (size_t)m * 8, hipMemcpyHostToDevice, s->stream));
            TVZ_HIP(hipMemcpyAsync(s->d_smult, mult.data(), (size_t)m * 4, hipMemcpyHostToDevice, s->stream));
        }
        TVZ_HIP(hipMemsetAsync(s->d_hits_n, 0, sizeof(int32_t), s->stream));
        std::shared_lock<std::shared_mutex> lk(c->mu);
        const int64_t n_rows = (int64_t)c->h_rows.size();
        if (n_rows) {
            hipLaunchKernelGGL(ts_match_longq_kernel, dim3((unsigned)tvz::ceil_div(n_rows, kGroupsPerBlock)),
                               dim3(kBlock), 0, s->stream, c->rows.p, n_rows, c->keys.p, s->d_sq,
                               s->d_smult, (int32_t)m, min_match, (int32_t)cap, s->d_hits, s->d_hits_n);
            TVZ_HIP(hipGetLastError());
            if (min_match > 0) {
                hipLaunchKernelGGL(ts_kth_fixup_kernel, dim3(1), dim3(kBlock), 0, s->stream, c->rows.p,
                                   c->keys.p, s->d_query, s->d_qoff, min_match, (int32_t)cap, s->d_hits,
                                   s->d_hits_n);
                TVZ_HIP(hipGetLastError());
            }
        }
        // the staging stream reads the vectors above asynchronously: finish before they go away
        TVZ_HIP(hipStreamSynchronize(s->stream));
        if (int rc = record(c, s->stream)) return rc;
    }
    int32_t *h_n = reinterpret_cast<int32_t *>(s->h_small + 2);
    TVZ_HIP(hipMemcpyAsync(h_n, s->d_hits_n, 4, hipMemcpyDeviceToHost, s->stream));
    TVZ_HIP(hipStreamSynchronize(s->stream));
    int64_t found = *h_n;
    int64_t stored = std::min<int64_t>(found, cap);
    if (stored) {
        TVZ_HIP(hipMemcpyAsync(s->h_hits, s->d_hits, (size_t)stored * 12, hipMemcpyDeviceToHost, s->stream));
        TVZ_HIP(hipStreamSynchronize(s->stream));
    }
    struct Hit { int32_t vid, cnt, kth; };
    Hit *hh = reinterpret_cast<Hit *>(s->h_hits);
    std::sort(hh, hh + stored, [](const Hit &a, const Hit &b) {
        if (a.vid != b.vid) return a.vid < b.vid;
        if (a.cnt != b.cnt) return a.cnt < b.cnt;
        return a.kth < b.kth;
    });
    int64_t w = 0;
    for (int64_t i = 0; i < stored; ++i) {
        if (hh[i].vid == exclude_id && exclude_id >= 0) { --found; continue; }
        h_out_ids[w] = hh[i].vid;
        h_out_counts[w] = hh[i].cnt;
        if (h_out_kth) h_out_kth[w] = hh[i].kth;
        ++w;
    }
    // truncated: report the device count so the caller can retry with cap >= *n_out
    *n_out = (*h_n > cap) ? found : w;
    return TVZ_OK;
}

static int tvz_topk_impl(const int32_t *d_lists, const int32_t *d_lists_n, int32_t n_lists,
                        int32_t Q, int32_t cap, int32_t k, int32_t *d_topk, void *hip_stream) {
    TVZ_REQUIRE(n_lists >= 1 && Q >= 0 && cap >= 0, "bad list shape");
    TVZ_REQUIRE(k >= 1 && k <= kSortCap / 2, "k=%d out of range [1, %d]", k, kSortCap / 2);
    if (Q == 0) return TVZ_OK;
    TVZ_REQUIRE((d_lists || cap == 0) && d_topk, "NULL argument");
    hipLaunchKernelGGL(ts_topk_kernel, dim3((unsigned)Q), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(hip_stream), d_lists, d_lists_n, n_lists, Q,
                       cap, k, d_topk, 0, nullptr);
    TVZ_HIP(hipGetLastError());
    return TVZ_OK;
}

static int tvz_topk_shard_impl(const int32_t *d_hits, const int32_t *d_hits_n, int32_t Q,
                              int32_t cap, int32_t k, int32_t *d_out, void *hip_stream) {
    TVZ_REQUIRE(Q >= 0 && cap >= 0, "bad list shape");
    TVZ_REQUIRE(k >= 1 && k <= kSortCap / 2, "k=%d out of range [1, %d]", k, kSortCap / 2);
    if (Q == 0) return TVZ_OK;
    TVZ_REQUIRE((d_hits || cap == 0) && d_hits_n && d_out, "NULL argument");
    hipLaunchKernelGGL(ts_topk_kernel, dim3((unsigned)Q), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(hip_stream), d_hits, d_hits_n, 1, Q, cap, k,
                       d_out, 1, nullptr);
    TVZ_HIP(hipGetLastError());
    return TVZ_OK;
}

static int tvz_topk_merge_impl(const int32_t *d_gathered, int32_t n_ranks, int32_t Q, int32_t k,
                              int32_t *d_topk, int32_t *d_totals, void *hip_stream) {
    TVZ_REQUIRE(n_ranks >= 1 && Q >= 0, "bad list shape");
    TVZ_REQUIRE(k >= 1 && k <= kSortCap / 2, "k=%d out of range [1, %d]", k, kSortCap / 2);
    if (Q == 0) return TVZ_OK;
    TVZ_REQUIRE(d_gathered && d_topk, "NULL argument");
    hipLaunchKernelGGL(ts_topk_kernel, dim3((unsigned)Q), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(hip_stream), d_gathered, nullptr, n_ranks, Q,
                       k + 1, k, d_topk, 2, d_totals);
    TVZ_HIP(hipGetLastError());
    return TVZ_OK;
}

static int tvz_align_impl(tvz_corpus *c, const double *d_query, int32_t n, double eps,
                         double max_offset, int32_t *d_out, void *hip_stream) {
    TVZ_REQUIRE(c != nullptr, "corpus is NULL");
    TVZ_REQUIRE(n >= 0 && (n == 0 || d_query), "bad query");
    TVZ_REQUIRE(eps > 0.0 && max_offset >= 0.0, "eps must be > 0 and max_offset >= 0");
    const double nb = floor(max_offset / eps + 0.5);
    if (2 * nb + 1 > kAlignMaxBins)
        return tvz::fail(TVZ_ERR_UNSUPPORTED, "max_offset/eps = %.0f needs more than %d bins", nb,
                         kAlignMaxBins);
    DeviceGuard dg(c->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    std::shared_lock<std::shared_mutex> lk(c->mu);
    const int64_t n_rows = (int64_t)c->h_rows.size();
    if (n_rows == 0) return TVZ_OK;
    TVZ_REQUIRE(d_out != nullptr, "d_out is NULL");
    const int64_t blocks = std::min<int64_t>(tvz::ceil_div(n_rows, kBlock / 64), 256 * 8);
    hipLaunchKernelGGL(ts_align_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, c->rows.p, n_rows,
                       c->keys.p, d_query, n, eps, (int32_t)nb, d_out);
    TVZ_HIP(hipGetLastError());
    return record(c, st);
}

// Not part of the stable ABI: 0 = LDS tile kernel only, 1 = dispatch rule, 2 = hash join whenever legal.
static int tvz_match_set_tuning_impl(int use_join) {
    g_use_join = use_join < 0 ? 0 : (use_join > 2 ? 2 : use_join);
    return TVZ_OK;
}

TVZ_EXPORT int tvz_corpus_create(tvz_corpus **out, int device) {
    TVZ_GUARDED(tvz_corpus_create_impl(out, device));
}

TVZ_EXPORT int tvz_corpus_destroy(tvz_corpus *c) {
    TVZ_GUARDED(tvz_corpus_destroy_impl(c));
}

TVZ_EXPORT int tvz_corpus_upload(tvz_corpus *c, const int32_t *h_video_ids,
                                 const int64_t *h_offsets, const double *h_keys, int64_t n_rows,
                                 int64_t n_keys) {
    TVZ_GUARDED(tvz_corpus_upload_impl(c, h_video_ids, h_offsets, h_keys, n_rows, n_keys));
}

TVZ_EXPORT int tvz_corpus_upsert(tvz_corpus *c, int32_t video_id, const double *h_keys, int64_t n) {
    TVZ_GUARDED(tvz_corpus_upsert_impl(c, video_id, h_keys, n));
}

TVZ_EXPORT int tvz_corpus_clear(tvz_corpus *c) {
    TVZ_GUARDED(tvz_corpus_clear_impl(c));
}

TVZ_EXPORT int tvz_corpus_stats(tvz_corpus *c, int64_t *n_rows, int64_t *n_keys,
                                int64_t *arena_keys) {
    TVZ_GUARDED(tvz_corpus_stats_impl(c, n_rows, n_keys, arena_keys));
}

TVZ_EXPORT int tvz_match(tvz_corpus *c, const double *d_queries, const int64_t *d_q_offsets,
                         int32_t Q, int32_t max_query_len, int32_t min_match,
                         const int32_t *d_exclude_ids, int32_t cap, int32_t *d_hits,
                         int32_t *d_hits_n, void *hip_stream) {
    TVZ_GUARDED(tvz_match_impl(c, d_queries, d_q_offsets, Q, max_query_len, min_match, d_exclude_ids, cap, d_hits, d_hits_n, hip_stream));
}

TVZ_EXPORT int tvz_find_duplicates(tvz_corpus *c, const double *h_query, int64_t n,
                                   int32_t min_match, int32_t exclude_id, int64_t cap,
                                   int32_t *h_out_ids, int32_t *h_out_counts, int32_t *h_out_kth,
                                   int64_t *n_out) {
    TVZ_GUARDED(tvz_find_duplicates_impl(c, h_query, n, min_match, exclude_id, cap, h_out_ids, h_out_counts, h_out_kth, n_out));
}

TVZ_EXPORT int tvz_topk(const int32_t *d_lists, const int32_t *d_lists_n, int32_t n_lists,
                        int32_t Q, int32_t cap, int32_t k, int32_t *d_topk, void *hip_stream) {
    TVZ_GUARDED(tvz_topk_impl(d_lists, d_lists_n, n_lists, Q, cap, k, d_topk, hip_stream));
}

TVZ_EXPORT int tvz_topk_shard(const int32_t *d_hits, const int32_t *d_hits_n, int32_t Q,
                              int32_t cap, int32_t k, int32_t *d_out, void *hip_stream) {
    TVZ_GUARDED(tvz_topk_shard_impl(d_hits, d_hits_n, Q, cap, k, d_out, hip_stream));
}

TVZ_EXPORT int tvz_topk_merge(const int32_t *d_gathered, int32_t n_ranks, int32_t Q, int32_t k,
                              int32_t *d_topk, int32_t *d_totals, void *hip_stream) {
    TVZ_GUARDED(tvz_topk_merge_impl(d_gathered, n_ranks, Q, k, d_topk, d_totals, hip_stream));
}

TVZ_EXPORT int tvz_align(tvz_corpus *c, const double *d_query, int32_t n, double eps,
                         double max_offset, int32_t *d_out, void *hip_stream) {
    TVZ_GUARDED(tvz_align_impl(c, d_query, n, eps, max_offset, d_out, hip_stream));
}

TVZ_EXPORT int tvz_match_set_tuning(int use_join) {
    TVZ_GUARDED(tvz_match_set_tuning_impl(use_join));
}
