// tvz_match_kernels.h — device kernels of the timestamp-corpus matcher (gfx950, wave64).
// Included by tvz_match.hip only.  See that file for the semantics (db.py:85-91) and the device
// image of `video_timestamps`.
//
// ts_match_q1_kernel     one query per block column: the query's keys in a small LDS multimap,
//                        a 16-lane group per row, per-lane counters, no per-row LDS state.
//                        HBM/MALL-stream-bound: the single-query and small-batch path.
// ts_match_tile_kernel   one LDS hash table per tile of <= 16 queries, ring/drain slow path.
// ts_join_build_kernel + ts_match_join_kernel   device-memory hash join per tile of <= 1024 queries,
//                        a wave per corpus row, per-wave LDS accounting (min_match 1..2).
// ts_match_longq_kernel  single queries longer than a tile (> 4095 timestamps).
// ts_kth_fixup_kernel    kth for min_match > 5.
// ts_topk_select_kernel  per-query k best of a long hit list: kth histogram in LDS picks the
//                        threshold, only the candidates are sorted.
// ts_topk_kernel         bitonic k best over short / gathered lists (merge side).
// ts_prep_kernel         zeroes hit counters and clears hash-join tables in ONE launch.
// ts_row_write_kernel    stream-ordered 16-byte swap of one row entry (upsert).
// ts_align_kernel        opt-in shift/tolerance score (never the verdict).
#pragma once
#include <climits>
#include <cstring>

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

// One row entry = ONE 16-byte load / store (global_load_dwordx4): an upsert swaps it with a single
// store while matches may be reading it, so a reader sees the old or the new entry, never a mix.
struct alignas(16) Row {
    int64_t off;
    int32_t len;
    int32_t vid;
};
static_assert(sizeof(Row) == 16, "Row must be 16 bytes");

__device__ __forceinline__ Row load_row(const Row *p) {
    const int4 v = *reinterpret_cast<const int4 *>(p);
    Row r;
    r.off = (int64_t)(((uint64_t)(uint32_t)v.y << 32) | (uint32_t)v.x);
    r.len = v.z;
    r.vid = v.w;
    return r;
}

// LDS counters updated by other lanes of the SAME wave are read back by plain loads: make the
// compiler keep the order (the hardware completes a wave's LDS operations in order).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- wave64 inclusive prefix sum on the VALU (six DPP adds: row_shr 1/2/4/8 inside the 16-lane rows,
// then row_bcast 15 and 31 across them).  A __shfl_up ladder is six ds_bpermute round trips through
// the LDS crossbar (~100 cycles each, and LDS-pipe time): the index lookup runs five scans per
// sub-index and was paying ~3,000 cycles of pure latency for them.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
    // lanes whose DPP source is invalid or masked take `old` = 0
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v = dpp_add_u32<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add_u32<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add_u32<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add_u32<0x118, 0xf>(v);  // row_shr:8   -> inclusive scan inside each row
    v = dpp_add_u32<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add_u32<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3
    return v;
}
__device__ __forceinline__ uint32_t wave_total(uint32_t incl) {       // of an inclusive scan
    return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}

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
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t ns, int32_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *slots = reinterpret_cast<uint32_t *>(smem);
    int64_t *ekey = reinterpret_cast<int64_t *>(smem + (size_t)kTileSlots * 4);
    uint32_t *epack = reinterpret_cast<uint32_t *>(ekey + kTileMaxEntries);
    unsigned long long *top = reinterpret_cast<unsigned long long *>(epack + kTileMaxEntries);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(top + kTileGroups * kTileQ);
    __shared__ int64_t s_qoff[kTileQ + 1];
    __shared__ int32_t s_cbase[kTileQ + 1];   // first table entry of query ql; -1: longer than a tile can hold

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
    // A query of more than 4095 timestamps is not this kernel's (the batched calls sweep such a query
    // on its own, tvz_match.hip launch_match_with_long): it gets no table entries and its counter is
    // marked; the other queries of the tile are matched as usual.
    if (threadIdx.x == 0) {
        int32_t c = 0;
        for (int ql = 0; ql < nq; ++ql) {
            const int64_t len = s_qoff[ql + 1] - s_qoff[ql];
            s_cbase[ql] = len <= kMaxQueryLen ? c : -1;
            if (len <= kMaxQueryLen) c += (int32_t)len;
        }
        s_cbase[nq] = c;
    }
    __syncthreads();
    if (threadIdx.x < nq && s_cbase[threadIdx.x] < 0) hits_n[(size_t)(q0 + threadIdx.x) * ns] = INT32_MIN;
    if (s_cbase[nq] > kTileMaxEntries) {
        // the caller's max_query_len was not an upper bound: poison the affected counters instead
        // of returning silently truncated matches (every row chunk of this tile takes this exit)
        if (threadIdx.x < nq) hits_n[(size_t)(q0 + threadIdx.x) * ns] = INT32_MIN;
        return;
    }
    // (keys of skipped long queries are walked over: a tile has at most 16 queries, such a query is rare)
    const int64_t total = s_qoff[nq] - qbase;
    for (int64_t ee = threadIdx.x; ee < total; ee += kTileBlock) {
        int ql = 0;
        while (ql + 1 < nq && s_qoff[ql + 1] - qbase <= ee) ++ql;
        if (s_cbase[ql] < 0) continue;
        const uint32_t pos = (uint32_t)(ee - (s_qoff[ql] - qbase));
        const int e = s_cbase[ql] + (int)pos;
        int64_t k;
        if (!canon_key(queries[qbase + ee], k)) continue;   // NaN never matches
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
    const bool my_q = gl < nq && s_cbase[gl < nq ? gl : 0] >= 0;
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
        if (live) row = load_row(rows + r);
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
        wave_lds_fence();
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
                const int slot = atomicAdd(&hits_n[(size_t)(q0 + gl) * ns], 1);
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

// ---- hash join for large query batches (min_match 1..2) -------------------------------------
// The LDS tile kernel probes every corpus key once per 16 queries.  For big batches a database
// hash JOIN does less work: build ONE multimap per tile of up to 1024 queries in device memory,
// then sweep the corpus ONCE per tile - one probe of a corpus key serves 1024 queries.
//
// What bounds such a probe is where the table lives: a random 16-byte probe moves a whole cache
// line, and beyond an XCD's 4 MiB L2 the lines come at the Infinity Cache's random-line rate
// (a 12 MiB table of 8-byte keys + packs: 47 % L2 hits, 3.6 GB fetched per sweep, no faster than
// eight sweeps over 128-query tables that fit - profiles/r2c_match_pmc.txt).  So the table is made
// SMALL instead: one 32-bit slot per query element,
//      [ fingerprint : 10 | query-in-tile : 10 | position : 12 ]       (0xffffffff = free)
// in 16-byte buckets of four; 2^19 slots = 2 MiB hold the 205 k elements of 1024 queries at load
// 0.4 and stay in L2.  A probe loads the key's home bucket (one 16-byte load), compares four
// fingerprints, and only on a fingerprint match (a real match, or 1 probe in ~700 by chance)
// verifies the FULL key against the query element itself (queries[...], 1.6 MB, also cache
// resident), so results stay exact.  A bucket with no free slot sends the probe on to the next.
//
// Sweep: a whole WAVE owns one corpus row at a time (64 lanes x 16 B = 128 keys per load
// instruction).  Matches are accounted in a per-wave LDS table of 256 slots keyed by the query's
// index in the tile - (tag | count) claimed with one CAS, the two smallest matching positions kept
// with two atomicMin - 3 KiB per wave, so two 16-wave blocks fit a CU at full occupancy.  After a
// row each lane scans 4 slots, emits the (query, row) pairs that reached min_match and resets
// them.  A row that touches more than 256 distinct queries of the tile sets an overflow flag and is
// re-done in 4 passes of 256 queries each, which always fit.
// SIMT shapes the inner loop as it did the tile kernel's: ~40 % of the lanes have a fingerprint
// match per key, so handling matches where they are found runs the verify + account path at
// every one of the 16 (key, slot) positions of a step with a fraction of the lanes (first
// version: 1,400 wave-instructions per 64 keys, profiles/r2d_match_pmc.txt).  Instead the
// fingerprint stage only PUSHES (slot, key index) candidates into a per-wave LDS ring by
// ballot/mbcnt compaction, and whenever 64 are pending the whole wave drains them with every lane
// busy on the exact compare and the accounting.
constexpr int kJoinQ = 1024;                          // queries per tile (10 bits of a slot)
constexpr int kJoinBlock = 1024;
constexpr int kJoinWaves = kJoinBlock / 64;
constexpr int kJoinSlots = 256;                       // per-wave accounting slots
#ifndef TVZ_JOIN_MAX_LOG2
#define TVZ_JOIN_MAX_LOG2 19                          // 2^19 x 4 B = 2 MiB per tile
#endif
constexpr int kJoinMaxSlotsLog2 = TVZ_JOIN_MAX_LOG2;
constexpr int kJoinRing = 128;                        // per-wave candidate ring (entries of 8 B)
constexpr size_t kJoinLds = (size_t)kJoinWaves * kJoinSlots * 12 + (size_t)kJoinWaves * kJoinRing * 8 +
                            kJoinWaves * 4 + (kJoinQ + 1) * 8;
constexpr uint32_t kJFree = 0xffffffffu;

// bucket index (high bits) and 10-bit fingerprint (low bits) of a canonical key
__device__ __forceinline__ void join_hash(int64_t k, int s_log2, uint32_t &bucket, uint32_t &fp) {
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)((uint64_t)k >> 32);
    uint32_t x = lo ^ (hi + (hi << 3)) ^ (hi >> 9);
    x ^= x >> 20;
    const uint32_t y = __umul24(x, 0x9E3779u);
    bucket = y >> (34 - s_log2);                      // s_log2 - 2 bits: buckets of 4 slots
    fp = (x ^ (x >> 11)) & 0x3ffu;
}

// One WAVE per query (4 per block), lanes striding over its elements: a block per query with 47 of
// 256 threads busy was 4096 nearly empty blocks and 110 us per 4096 queries - more than a sweep of
// a 1/8 shard.
__global__ __launch_bounds__(kBlock) void ts_join_build_kernel(
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t Q,
    int32_t max_len, int32_t q_per_tile, int32_t s_log2, uint32_t *__restrict__ table,
    int32_t *__restrict__ hits_n, int32_t ns) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int64_t o = q_offsets[q];
    const int64_t len = q_offsets[q + 1] - o;
    if (len > max_len) {
        // max_query_len was not an upper bound (the table is sized from it): nothing of this
        // query is inserted and its counter is poisoned instead (stays negative)
        if (lane == 0) hits_n[(size_t)q * ns] = INT32_MIN;
        return;
    }
    const int tile = q / q_per_tile;
    uint32_t *tb = table + ((size_t)tile << s_log2);
    const uint32_t bmask = (1u << (s_log2 - 2)) - 1u;
    const uint32_t qfield = (uint32_t)(q - tile * q_per_tile) << 12;
    for (int i = lane; i < (int)len; i += 64) {
        int64_t k;
        if (!canon_key(queries[o + i], k)) continue;               // NaN never matches
        uint32_t b, fp;
        join_hash(k, s_log2, b, fp);
        const uint32_t entry = (fp << 22) | qfield | (uint32_t)i;
        // first free slot from the home bucket on: look, then claim it.  The look may be stale (this
        // CU's L1), but slots only ever go from free to taken, so a stale view can only offer a slot
        // that is gone - the failed CAS returns what is there and corrects the view.
        uint4 bk = *reinterpret_cast<const uint4 *>(tb + b * 4);
        while (true) {
            const int j = bk.x == kJFree ? 0 : bk.y == kJFree ? 1 : bk.z == kJFree ? 2 : bk.w == kJFree ? 3 : 4;
            if (j == 4) {                                          // bucket full: next one
                b = (b + 1) & bmask;
                bk = *reinterpret_cast<const uint4 *>(tb + b * 4);
                continue;
            }
            const uint32_t old = atomicCAS(&tb[b * 4 + j], kJFree, entry);
            if (old == kJFree) break;
            if (j == 0) bk.x = old; else if (j == 1) bk.y = old; else if (j == 2) bk.z = old; else bk.w = old;
        }
    }
}

__global__ __launch_bounds__(kJoinBlock, 8) void ts_match_join_kernel(
    const Row *__restrict__ rows, int64_t n_rows, const int64_t *__restrict__ keys,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets,
    const uint32_t *__restrict__ table, int32_t s_log2, int32_t Q, int32_t q_per_tile, int32_t tile,
    int32_t min_match, const int32_t *__restrict__ exclude_ids, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t ns) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the wave index in an SGPR: the row counter of the sweep below is wave-uniform, and as a 64-bit VGPR
    // pair at the 64-register cap it was spilled to scratch and reloaded once per row
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *w0 = reinterpret_cast<uint32_t *>(smem) + wave * kJoinSlots;                     // (q+1) << 20 | count
    uint32_t *m1 = reinterpret_cast<uint32_t *>(smem) + (kJoinWaves + wave) * kJoinSlots;      // smallest position
    uint32_t *m2 = reinterpret_cast<uint32_t *>(smem) + (2 * kJoinWaves + wave) * kJoinSlots;  // second smallest
    uint32_t *rslot = reinterpret_cast<uint32_t *>(smem + (size_t)kJoinWaves * kJoinSlots * 12) + wave * kJoinRing;
    uint32_t *rkidx = reinterpret_cast<uint32_t *>(smem + (size_t)kJoinWaves * kJoinSlots * 12) +
                      (kJoinWaves + wave) * kJoinRing;
    uint32_t *ovf = reinterpret_cast<uint32_t *>(smem + (size_t)kJoinWaves * kJoinSlots * 12 +
                                                 (size_t)kJoinWaves * kJoinRing * 8) + wave;
    int64_t *s_qoff = reinterpret_cast<int64_t *>(smem + (size_t)kJoinWaves * kJoinSlots * 12 +
                                                  (size_t)kJoinWaves * kJoinRing * 8 + kJoinWaves * 4);
    const int q0 = tile * q_per_tile;
    const int nq = (Q - q0 < q_per_tile) ? Q - q0 : q_per_tile;
    for (int i = threadIdx.x; i <= nq; i += kJoinBlock) s_qoff[i] = q_offsets[q0 + i];
    for (int i = lane; i < kJoinSlots; i += 64) { w0[i] = 0; m1[i] = 0xffffffffu; m2[i] = 0xffffffffu; }
    if (lane == 0) *ovf = 0;
    __syncthreads();
    const uint32_t bmask = (1u << (s_log2 - 2)) - 1u;
    const uint4 *tb = reinterpret_cast<const uint4 *>(table + ((size_t)tile << s_log2));
    const int64_t n_waves = (int64_t)gridDim.x * kJoinWaves;
    uint32_t qhead = 0, qtail = 0;                                    // wave-uniform ring cursors

    for (int64_t r = (int64_t)blockIdx.x * kJoinWaves + wave; r < n_rows; r += n_waves) {
        const Row row = load_row(rows + r);
        const int64_t *rk = keys + row.off;
        int part = -1;                                                // -1: every query of the tile
        // one verified (query, position) match; part >= 0 restricts to queries with (q & 3) == part
        auto account = [&](uint32_t ql, uint32_t pos) {
            if (part >= 0 && (int)(ql & 3u) != part) return;
            const uint32_t tag = (ql + 1u) << 20;
            uint32_t s = (ql * 157u) & (uint32_t)(kJoinSlots - 1);
            for (int tries = 0;; ++tries) {
                const uint32_t old = atomicCAS(&w0[s], 0u, tag);
                if (old == 0u || (old >> 20) == ql + 1u) break;       // claimed, or this query's slot
                if (tries == kJoinSlots) { *ovf = 1u; return; }        // > 256 distinct queries: redo the row
                s = (s + 1) & (uint32_t)(kJoinSlots - 1);
            }
            atomicAdd(&w0[s], 1u);
            const uint32_t o = atomicMin(&m1[s], pos);                // two plain LDS atomics, no CAS loop
            atomicMin(&m2[s], o > pos ? o : pos);                     // larger of two distinct hits >= 2nd smallest
        };
        // n <= 64 pending candidates: every lane takes one - exact compare against the query
        // element itself, then the accounting
        auto drain = [&](uint32_t n) {
            wave_lds_fence();
            if ((uint32_t)lane < n) {
                const uint32_t idx = (qhead + lane) & (uint32_t)(kJoinRing - 1);
                const uint32_t slot = rslot[idx];
                const int64_t k = rk[rkidx[idx]];
                const uint32_t ql = (slot >> 12) & 0x3ffu, pos = slot & 0xfffu;
                int64_t qk;
                if (canon_key(queries[s_qoff[ql] + pos], qk) && qk == k) account(ql, pos);
            }
            qhead += n;
        };
        auto push = [&](bool cand, uint32_t slot, uint32_t ki) {
            const unsigned long long bal = __ballot(cand);
            if (bal) {                                                // wave-uniform
                const uint32_t ofs = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (cand) {
                    const uint32_t idx = (qtail + ofs) & (uint32_t)(kJoinRing - 1);
                    rslot[idx] = slot;
                    rkidx[idx] = ki;
                }
                qtail += (uint32_t)__popcll(bal);
                if (qtail - qhead >= 64u) drain(64u);
            }
        };
        // fingerprint stage of one key: candidates of its home bucket (and of the following ones
        // while they are full) go to the ring
        auto probe = [&](bool valid, uint32_t ki, uint32_t b, uint32_t fp, uint4 bk) {
            bool more = valid;
            while (true) {
                push(more && (bk.x >> 22) == fp && bk.x != kJFree, bk.x, ki);
                push(more && (bk.y >> 22) == fp && bk.y != kJFree, bk.y, ki);
                push(more && (bk.z >> 22) == fp && bk.z != kJFree, bk.z, ki);
                push(more && (bk.w >> 22) == fp && bk.w != kJFree, bk.w, ki);
                more = more && bk.x != kJFree && bk.y != kJFree && bk.z != kJFree && bk.w != kJFree;
                if (__ballot(more) == 0ull) break;                     // wave-uniform; a full bucket is rare
                if (more) {
                    b = (b + 1) & bmask;
                    bk = tb[b];
                }
            }
        };
        while (true) {
            for (int base = 0; base < row.len; base += 256) {
                // 2 x 16-byte loads per lane = 4 keys; unconditional (clamped) loads keep them all in flight
                const int i0 = base + lane * 2, i1 = i0 + 128;
                const longlong2 v0 = *reinterpret_cast<const longlong2 *>((i0 < row.len) ? rk + i0 : keys);
                const longlong2 v1 = *reinterpret_cast<const longlong2 *>((i1 < row.len) ? rk + i1 : keys);
                uint32_t b0, b1, b2, b3, f0, f1, f2, f3;
                join_hash(v0.x, s_log2, b0, f0);
                join_hash(v0.y, s_log2, b1, f1);
                join_hash(v1.x, s_log2, b2, f2);
                join_hash(v1.y, s_log2, b3, f3);
                const uint4 k0 = tb[b0], k1 = tb[b1], k2 = tb[b2], k3 = tb[b3];   // independent, all in flight
                probe(i0 < row.len, (uint32_t)i0, b0, f0, k0);
                probe(i0 + 1 < row.len, (uint32_t)i0 + 1u, b1, f1, k1);
                probe(i1 < row.len, (uint32_t)i1, b2, f2, k2);
                probe(i1 + 1 < row.len, (uint32_t)i1 + 1u, b3, f3, k3);
            }
            while (qtail != qhead) drain(qtail - qhead < 64u ? qtail - qhead : 64u);   // row boundary: settle
            wave_lds_fence();
            const bool overflow = *ovf != 0u;
            // every lane owns 4 slots: emit the (query, row) pairs that reached min_match, reset all
            const uint4 cw = *reinterpret_cast<const uint4 *>(w0 + lane * 4);
            if (!overflow && (cw.x | cw.y | cw.z | cw.w)) {
                const uint4 a1 = *reinterpret_cast<const uint4 *>(m1 + lane * 4);
                const uint4 a2 = *reinterpret_cast<const uint4 *>(m2 + lane * 4);
                auto emit = [&](uint32_t wv, uint32_t p1, uint32_t p2) {
                    if (!wv) return;
                    const int c = (int)(wv & 0xfffffu);
                    const int q = q0 + (int)(wv >> 20) - 1;
                    if (c < min_match || q >= Q) return;
                    if (exclude_ids && exclude_ids[q] == row.vid) return;
                    const int32_t kth = (int32_t)(min_match == 1 ? p1 : p2);
                    const int slot = atomicAdd(&hits_n[(size_t)q * ns], 1);
                    if (slot < cap) {
                        int32_t *hp = hits + ((int64_t)q * cap + slot) * 3;
                        hp[0] = row.vid;
                        hp[1] = c;
                        hp[2] = kth;
                    }
                };
                emit(cw.x, a1.x, a2.x);
                emit(cw.y, a1.y, a2.y);
                emit(cw.z, a1.z, a2.z);
                emit(cw.w, a1.w, a2.w);
            }
            const uint4 ones = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            *reinterpret_cast<uint4 *>(w0 + lane * 4) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4 *>(m1 + lane * 4) = ones;
            *reinterpret_cast<uint4 *>(m2 + lane * 4) = ones;
            if (lane == 0) *ovf = 0;
            wave_lds_fence();
            if (overflow && part < 0) { part = 0; continue; }          // redo in 4 passes of <= 256 queries
            if (part >= 0 && part < 3) { ++part; continue; }
            break;
        }
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

// ---- a batch's long queries, prepared ON THE DEVICE (no host copy, no allocation, no synchronisation) ----------
// One block per long query: its canonical keys into a power-of-two scratch (NaN -> the largest int64: sorted to the
// end, never a key), bitonic sort there, then the distinct keys and their multiplicities - what
// ts_match_longq_kernel searches per row key - and their number.  The scratch is a tail of the caller's workspace
// (tvz_match_workspace_bytes_long); a query of more than 4,095 timestamps is rare, this is not a fast path.
constexpr int kLongSortBlock = 1024;
constexpr int64_t kLongNoKey = 0x7fffffffffffffffLL;        // (a NaN pattern: never a canonical key)
struct LongQ {                 // one long query of a batch
    int64_t q_off;             // first key in d_queries
    int32_t n;                 // keys
    int32_t pow2;              // sort scratch entries
    int64_t sort_at;           // int64 index of the sort scratch in the long-query area
    int64_t uq_at;             // int64 index of the distinct keys [n + 1]
    int64_t mult_at;           // int32 index of the multiplicities [n + 1]
    int64_t m_at;              // int32 index of the number of distinct keys
};

constexpr int kLongPerLaunch = 64;
struct LongQTable { LongQ e[kLongPerLaunch]; };            // travels BY VALUE in the kernel arguments: no copy to enqueue

__global__ __launch_bounds__(kLongSortBlock) void ts_longq_sort_kernel(const double *__restrict__ queries,
                                                                        const LongQTable table,
                                                                        int64_t *__restrict__ area) {
    const LongQ lq = table.e[blockIdx.x];
    int64_t *sk = area + lq.sort_at;
    int64_t *uq = area + lq.uq_at;
    int32_t *mult = reinterpret_cast<int32_t *>(area) + lq.mult_at;
    int32_t *m_out = reinterpret_cast<int32_t *>(area) + lq.m_at;
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < lq.pow2; i += kLongSortBlock) {
        int64_t k = kLongNoKey;
        if (i < lq.n && !canon_key(queries[lq.q_off + i], k)) k = kLongNoKey;
        sk[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= lq.pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < lq.pow2 / 2; t += kLongSortBlock) {
                const int lo = 2 * t - (t & (stride - 1));          // index with bit `stride` clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const int64_t a = sk[lo], b = sk[hi];
                if ((a > b) == up) { sk[lo] = b; sk[hi] = a; }
            }
            __syncthreads();
        }
    }
    // distinct keys: thread t owns a contiguous range; heads counted, scanned over the block, then written
    __shared__ int s_cnt[kLongSortBlock];
    __shared__ int s_total;
    const int per = (lq.n + kLongSortBlock - 1) / kLongSortBlock;
    const int i0 = tid * per, i1 = i0 + per < lq.n ? i0 + per : lq.n;
    int heads = 0;
    for (int i = i0; i < i1; ++i) heads += (sk[i] != kLongNoKey && (i == 0 || sk[i] != sk[i - 1])) ? 1 : 0;
    s_cnt[tid] = heads;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < kLongSortBlock; ++t) { const int c = s_cnt[t]; s_cnt[t] = run; run += c; }
        s_total = run;
    }
    __syncthreads();
    int at = s_cnt[tid];
    for (int i = i0; i < i1; ++i) {
        if (sk[i] != kLongNoKey && (i == 0 || sk[i] != sk[i - 1])) {
            int j = i + 1;
            while (j < lq.n && sk[j] == sk[i]) ++j;                // (a run may go on into the next thread's range)
            uq[at] = sk[i];
            mult[at] = j - i;
            ++at;
        }
    }
    if (tid == 0) *m_out = s_total;
}


__global__ __launch_bounds__(kBlock) void ts_match_longq_kernel(
    const Row *__restrict__ rows, int64_t n_rows, const int64_t *__restrict__ keys,
    const int64_t *__restrict__ sq, const int32_t *__restrict__ smult, int32_t m, int32_t min_match,
    int32_t exclude_one, int32_t cap, int32_t *__restrict__ hits, int32_t *__restrict__ hits_n,
    const int32_t *__restrict__ m_dev, const int32_t *__restrict__ exclude_dev) {
    // (a batch's long queries are prepared on the device - ts_longq_sort_kernel: the number of distinct keys and the
    // query's own video id are then read here, the host never learns them)
    if (m_dev) m = *m_dev;
    if (exclude_dev) exclude_one = *exclude_dev;
    const int gl = threadIdx.x & (kGroup - 1);
    const int64_t r = (int64_t)blockIdx.x * kGroupsPerBlock + threadIdx.x / kGroup;
    if (r >= n_rows) return;                       // whole 16-lane groups leave together
    const Row row = load_row(rows + r);
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
    if (gl == 0 && cnt >= min_match && row.vid != exclude_one) {
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
    int32_t cap, int32_t *__restrict__ hits, const int32_t *__restrict__ hits_n, int32_t ns) {
    const int q = blockIdx.x;
    const int gl = threadIdx.x & (kGroup - 1);
    const int g = threadIdx.x / kGroup;
    const int gshift = (threadIdx.x & 63) & ~(kGroup - 1);
    int n = hits_n[(size_t)q * ns];
    if (n > cap) n = cap;
    const int64_t qo = q_offsets[q];
    const int32_t qlen = (int32_t)(q_offsets[q + 1] - qo);
    const double *qv = queries + qo;
    for (int j = g; j < n; j += kGroupsPerBlock) {
        int32_t *h = hits + ((int64_t)q * cap + j) * 3;
        const int32_t code = h[2];
        if (code > -2) continue;
        const Row row = load_row(rows + (-2 - code));
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
    return ((uint64_t)((uint32_t)kth + 1u) << 32) | (uint32_t)vid;   // NEVER + 1 wraps in unsigned
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
                                                         int32_t mode, int32_t *__restrict__ totals,
                                                         const int32_t *__restrict__ flags) {
    // mode 0: plain.  mode 1 (shard side): the output has k+1 rows per query, row k carries the
    // true number of hits as (-1, n, NEVER) so one all-gather moves lists and totals together; n is
    // NEGATED when the shard's hit list overflowed its capacity (its top-k may then be inexact).
    // mode 2 (merge side): every input list ends with such a row; |n| is summed into totals[q], and
    // the sum is negated if any shard overflowed, so the caller knows to re-run with a larger cap.
    // flags != NULL: ts_topk_wave_kernel went first; this kernel is launched with a small grid and
    // takes only the queries it flagged (block-uniform loop).
    __shared__ uint64_t key[kSortCap];
    __shared__ int32_t cnt[kSortCap];
    for (int q = blockIdx.x; q < Q; q += gridDim.x) {
    if (flags && flags[q] == 0) continue;
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
    __syncthreads();
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
        const Row row = load_row(rows + r);
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
        wave_lds_fence();
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


// ---- single-query sweep: the streaming driver's per-micro-batch call and small batches -------
// One query per block column (blockIdx.y).  Every block rebuilds the query in LDS as
//   (a) a blocked Bloom filter: one 64-bit word per key, one bit in each half, chosen by hash bits
//       (words >= 4 x keys: ~0.02 % false positives per probe, never a false negative), and
//   (b) an exact multimap (8-byte keys, load <= 0.5; every query element takes its own slot, so
//       multiplicity is exact; a lookup walks slot PAIRS up to the first free slot).
// A 16-lane group owns one corpus row at a time and streams its keys with 16-byte loads.
// FAST PASS, branch-free: per key one hash, one ds_read_b64 of its Bloom word, two bit tests, one
// add - about 16 wave-instructions per 64 keys - which only COUNTS the row's keys that may be in
// the query.  A row can reach min_match only if that count does (query with repeated keys: if it
// is >= 1), so the EXACT PASS - full-key probe of the multimap, hit count and the smallest
// matching query positions in per-lane registers, DPP reduction over the 16 lanes - runs for the
// ~2 % of rows that are real candidates instead of for every wave-instruction that has one
// matching lane out of 64 (which is all of them: the first version did that and was bound by
// instruction issue at 78 instructions per 64 keys, profiles/r2a_match_pmc.txt).
// No per-row LDS state, no atomics on the hot path; rows are spread over up to 2048 blocks, so the
// sweep is bounded by how fast the corpus streams out of HBM / Infinity Cache: algorithmic bytes =
// 16 B row entry + 8 B per key, each read once per query.
constexpr int kQ1Block = 256;
constexpr int kQ1Groups = kQ1Block / kGroup;          // rows in flight per block
constexpr int kQ1MinLog2 = 8, kQ1MaxLog2 = 13;        // 256 .. 8192 slots (2 KiB .. 64 KiB of keys)
constexpr int kQ1BloomMaxLog2 = 12;                   // <= 4096 words (32 KiB)
constexpr int kQ1ModeM2 = 0;                          // min_match 1..2: two smallest positions
constexpr int kQ1ModeTop5 = 1;                        // min_match 3..5: five smallest positions
constexpr int kQ1ModeCount = 2;                       // min_match <= 0 (kth = -1) or > 5 (fix-up)

struct HostOut {           // tvz_find_duplicates: hits go straight to pinned host memory
    int32_t *hits;         // [blocks][region][3]
    int32_t *counts;       // [blocks]
    int32_t region;        // hit slots per block = rows one block can sweep
};

// A short single query travels BY VALUE in the kernel-argument segment (q_offsets == nullptr):
// tvz_find_duplicates then needs no host-to-device copy at all - one launch, one synchronisation.
constexpr int kQ1ByValKeys = 440;                     // 3520 B of the 4 KiB kernarg segment
struct QByVal {
    int32_t n;
    int32_t pad;
    double k[kQ1ByValKeys];
};

inline int q1_slots_log2(int64_t n) {
    int s = kQ1MinLog2;
    while (s < kQ1MaxLog2 && ((int64_t)1 << s) < 4 * n) ++s;
    return s;
}
inline int q1_bloom_log2(int s_log2) { return s_log2 < kQ1BloomMaxLog2 ? s_log2 : kQ1BloomMaxLog2; }
// exact table: 8 B key + 2 B position per slot; Bloom: 8 B per word
inline size_t q1_lds_bytes(int s_log2) {
    return (((size_t)8 + 2) << s_log2) + ((size_t)8 << q1_bloom_log2(s_log2));
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp16(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp16_64(unsigned long long v) {
    return ((unsigned long long)dpp16<CTRL>((uint32_t)(v >> 32)) << 32) | dpp16<CTRL>((uint32_t)v);
}

// butterfly over a 16-lane DPP row: after the four steps every lane holds the reduction of all
// 16; each step combines two DISJOINT sets of lanes (needed for the second-smallest merge)
#define TVZ_ROW16_BUTTERFLY(STEP) \
    STEP(0xB1)  /* quad_perm [1,0,3,2] */ \
    STEP(0x4E)  /* quad_perm [2,3,0,1] */ \
    STEP(0x141) /* row_half_mirror     */ \
    STEP(0x140) /* row_mirror          */

// the 32-bit mix every Q1 hash is cut from (bits 0-4 / 5-9: the two Bloom bits; x * C: the slots)
__device__ __forceinline__ uint32_t q1_mix(int64_t k) {
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)((uint64_t)k >> 32);
    uint32_t x = lo ^ (hi + (hi << 3)) ^ (hi >> 9);
    x ^= x >> 20;
    return x;
}

// The body is a device function over (block bx of nbx along the rows, query q) with BS threads, so
// that tvz_find_duplicates can run it NEXT TO the index lookup in one launch (ts_find_fused_kernel:
// the lookup's blocks answer the indexed rows, these sweep the delta table).
// Hits that are not written to the host (HOSTOUT = false) are staged per block in LDS and the
// block reserves its range of the query's list with ONE global atomic at the end: at min_match 2 a
// query has thousands of accidental hits, and a returning atomic per hit on one counter serialised
// them (1.96 TB/s against 3.66 TB/s for the same sweep at min_match 5).
constexpr int kQ1Stage = 256;                         // staged hits per block (12 B each); more go out directly
template <int MODE, bool HOSTOUT, int BS>
__device__ __forceinline__ void q1_body(
    const Row *__restrict__ rows, int64_t n_rows, const int64_t *__restrict__ keys,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t min_match,
    const int32_t *__restrict__ exclude_ids, int32_t exclude_one, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t ns, int32_t s_log2, HostOut ho,
    const QByVal &qv, const int bx, const int nbx, const int q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int32_t s_stage[HOSTOUT ? 1 : kQ1Stage * 3];
    __shared__ int32_t s_stage_base;
    const int S = 1 << s_log2;
    const int b_log2 = s_log2 < kQ1BloomMaxLog2 ? s_log2 : kQ1BloomMaxLog2;
    int64_t *skey = reinterpret_cast<int64_t *>(smem);
    uint2 *bloom = reinterpret_cast<uint2 *>(skey + S);
    uint16_t *spos = reinterpret_cast<uint16_t *>(bloom + ((size_t)1 << b_log2));
    __shared__ int32_t s_nhits, s_dups;
    const bool byval = q_offsets == nullptr;          // the query is in the kernel arguments
    const int64_t qo = byval ? 0 : q_offsets[q];
    const int64_t n = byval ? qv.n : q_offsets[q + 1] - qo;
    if (threadIdx.x == 0) { s_nhits = 0; s_dups = 0; }
    if (2 * n > S) {
        // the caller's max_query_len was not an upper bound (the tables are sized from it)
        if (!HOSTOUT && threadIdx.x == 0) hits_n[(size_t)q * ns] = INT32_MIN;
        if (HOSTOUT && threadIdx.x == 0) ho.counts[bx] = INT32_MIN;
        return;
    }
    for (int i = threadIdx.x * 2; i < S; i += BS * 2)
        *reinterpret_cast<longlong2 *>(skey + i) = make_longlong2(kEmpty, kEmpty);
    for (int i = threadIdx.x; i < (1 << b_log2); i += BS) bloom[i] = make_uint2(0u, 0u);
    __syncthreads();
    const int pair_shift = 33 - s_log2;               // home PAIR from the top hash bits
    const int word_shift = 32 - b_log2;
    const uint32_t smask = (uint32_t)S - 1u;
    for (int e = threadIdx.x; e < (int)n; e += BS) {
        int64_t k;
        const double qk = byval ? qv.k[e] : queries[qo + e];
        if (!canon_key(qk, k)) continue;                         // NaN never matches
        const uint32_t x = q1_mix(k);
        const uint32_t y = __umul24(x, 0x9E3779u);              // __umul24 returns int: keep it unsigned
        uint32_t *bw = reinterpret_cast<uint32_t *>(bloom + (y >> word_shift));
        atomicOr(bw, 1u << (x & 31u));
        atomicOr(bw + 1, 1u << ((x >> 5) & 31u));
        uint32_t h = (y >> pair_shift) << 1;
        while (true) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&skey[h]),
                                                     (unsigned long long)kEmpty, (unsigned long long)k);
            if (old == (unsigned long long)kEmpty) break;
            if (old == (unsigned long long)k) s_dups = 1;       // the query repeats a key
            h = (h + 1) & smask;
        }
        spos[h] = (uint16_t)e;
    }
    __syncthreads();
    // a row reaches min_match only if >= thr of its keys pass the filter
    const int32_t thr = min_match <= 0 ? 0 : (s_dups ? 1 : min_match);

    const int gl = threadIdx.x & (kGroup - 1);
    const int g = threadIdx.x / kGroup;
    const int32_t excl = exclude_ids ? exclude_ids[q] : exclude_one;
    constexpr int kGroups = BS / kGroup;
    const int64_t stride = (int64_t)nbx * kGroups;
    // A step = EIGHT 16-byte loads per lane = 256 keys per group: a typical row (~200 cuts) is one
    // step, i.e. 8 KiB per wave in flight while it waits - with 16-32 waves per CU that is the
    // 100+ KiB per CU an HBM stream needs (Little's law; the first versions kept 2-4 KiB per wave
    // in flight and levelled off at 2.5 TB/s whatever their instruction count).  The group's
    // next row entry is fetched one row ahead, so a row costs ONE dependent round trip.
    // UNCONDITIONAL loads (lanes past the row's end read the arena's first 16 bytes instead and are
    // masked at compare time): a branch around a load makes the compiler wait with vmcnt(0).
    constexpr int kLd = 8;                                  // 16-byte loads per lane and step
    constexpr int kStepKeys = kLd * 2 * kGroup;             // 256
    auto maybe = [&](int64_t k) -> uint32_t {               // 1 if k may be in the query
        const uint32_t x = q1_mix(k);
        const uint32_t y = __umul24(x, 0x9E3779u);
        const uint2 w = bloom[y >> word_shift];
        return (w.x >> (x & 31u)) & (w.y >> ((x >> 5) & 31u)) & 1u;
    };
    int64_t r = (int64_t)bx * kGroups + g;
    const int64_t last_row = n_rows - 1;
    Row row = load_row(rows + (r < n_rows ? r : last_row));       // past the end: a valid row, never used
    while (r < n_rows) {
        const int64_t rn = r + stride;
        // lands while this row is tested; past the end it is the last row again (valid memory, and
        // the loop ends before anything of it is used)
        const Row nrow = load_row(rows + (rn < n_rows ? rn : last_row));
        uint32_t may = 0;
        for (int base = 0; base < row.len; base += kStepKeys) {
            longlong2 cur[kLd];
#pragma unroll
            for (int j = 0; j < kLd; ++j) {
                const int i = base + gl * 2 + j * 2 * kGroup;
                const int64_t *p = (i < row.len) ? keys + row.off + i : keys;
                cur[j] = *reinterpret_cast<const longlong2 *>(p);
            }
#pragma unroll
            for (int j = 0; j < kLd; ++j) {
                const int i = base + gl * 2 + j * 2 * kGroup;
                // unconditional probes, masked afterwards: no branch per key
                may += (maybe(cur[j].x) & (uint32_t)(i < row.len)) + (maybe(cur[j].y) & (uint32_t)(i + 1 < row.len));
            }
        }
#define TVZ_SUM_STEP(C) may += dpp16<C>(may);
        TVZ_ROW16_BUTTERFLY(TVZ_SUM_STEP)
#undef TVZ_SUM_STEP
        const bool cand = (int32_t)may >= thr;
        if (__ballot(cand) != 0ull) {
            // ---- exact pass over the candidate rows of this wave (their keys are L2-hot) ----
            uint32_t cnt = 0, m1 = 0xffffffffu, m2 = 0xffffffffu;
            unsigned long long top = kTopNone;
            auto acc = [&](uint32_t pos) {
                ++cnt;
                if constexpr (MODE == kQ1ModeM2) {
                    const uint32_t lo = m1 < pos ? m1 : pos, hi = m1 < pos ? pos : m1;
                    m1 = lo;
                    m2 = m2 < hi ? m2 : hi;
                } else if constexpr (MODE == kQ1ModeTop5) {
                    top = top5_insert(top, pos);
                }
            };
            if (cand) {
                // four 16-byte loads per lane in flight (one load per trip was a chain of ~13 L2 round
                // trips per candidate row; a block lives as long as its slowest wave, and at min_match 2
                // nearly every block has a candidate row)
                const int64_t *rk = keys + row.off;
                constexpr int kEx = 4;
                for (int base = 0; base < row.len; base += kEx * 2 * kGroup) {
                    longlong2 a[kEx];
#pragma unroll
                    for (int j = 0; j < kEx; ++j) {
                        const int i0 = base + gl * 2 + j * 2 * kGroup;
                        a[j] = *reinterpret_cast<const longlong2 *>(i0 < row.len ? rk + i0 : keys);
                    }
#pragma unroll
                    for (int j = 0; j < kEx; ++j) {
                        const int i0 = base + gl * 2 + j * 2 * kGroup;
                        const int64_t kk[2] = {a[j].x, a[j].y};
                        const bool valid[2] = {i0 < row.len, i0 + 1 < row.len};
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            if (!valid[e]) continue;
                            const uint32_t y = __umul24(q1_mix(kk[e]), 0x9E3779u);
                            uint32_t h = (y >> pair_shift) << 1;
                            while (true) {
                                const longlong2 ww = *reinterpret_cast<const longlong2 *>(skey + h);
                                if (ww.x == kEmpty) break;
                                if (ww.x == kk[e]) acc(spos[h]);
                                if (ww.y == kEmpty) break;
                                if (ww.y == kk[e]) acc(spos[h + 1]);
                                h = (h + 2) & smask;
                            }
                        }
                    }
                }
            }
            // the group's totals (every lane ends up with them; non-candidate groups carry zeros)
#define TVZ_SUM_STEP(C) cnt += dpp16<C>(cnt);
            TVZ_ROW16_BUTTERFLY(TVZ_SUM_STEP)
#undef TVZ_SUM_STEP
            const bool hit = cand && (int64_t)cnt >= (int64_t)min_match && row.vid != excl;
            if (__ballot(hit) != 0ull) {
                if constexpr (MODE == kQ1ModeM2) {
#define TVZ_M2_STEP(C) { const uint32_t p1 = dpp16<C>(m1), p2 = dpp16<C>(m2); \
                    const uint32_t lo = m1 < p1 ? m1 : p1, hi = m1 < p1 ? p1 : m1, r2 = m2 < p2 ? m2 : p2; \
                    m1 = lo; m2 = hi < r2 ? hi : r2; }
                    TVZ_ROW16_BUTTERFLY(TVZ_M2_STEP)
#undef TVZ_M2_STEP
                } else if constexpr (MODE == kQ1ModeTop5) {
#define TVZ_T5_STEP(C) { const unsigned long long p = dpp16_64<C>(top); \
                    _Pragma("unroll") for (int i = 0; i < kTop; ++i) top = top5_insert(top, (uint32_t)(p >> (12 * i)) & 0xfffu); }
                    TVZ_ROW16_BUTTERFLY(TVZ_T5_STEP)
#undef TVZ_T5_STEP
                }
            }
            if (hit && gl == 0) {
                int32_t kth;
                if (min_match <= 0) kth = -1;
                else if constexpr (MODE == kQ1ModeM2) kth = (int32_t)(min_match == 1 ? m1 : m2);
                else if constexpr (MODE == kQ1ModeTop5) kth = (int32_t)((top >> (12 * (min_match - 1))) & 0xfffu);
                else kth = -2 - (int32_t)r;                 // resolved by ts_kth_fixup_kernel
                const int slot = atomicAdd(&s_nhits, 1);            // LDS
                if constexpr (HOSTOUT) {
                    int32_t *h = ho.hits + ((int64_t)bx * ho.region + slot) * 3;
                    h[0] = row.vid;
                    h[1] = (int32_t)cnt;
                    h[2] = kth;
                } else if (slot < kQ1Stage) {
                    s_stage[slot * 3 + 0] = row.vid;
                    s_stage[slot * 3 + 1] = (int32_t)cnt;
                    s_stage[slot * 3 + 2] = kth;
                } else {                                            // a block with > 256 hits: the rest one by one
                    const int gs = atomicAdd(&hits_n[(size_t)q * ns], 1);
                    if (gs < cap) {
                        int32_t *h = hits + ((int64_t)q * cap + gs) * 3;
                        h[0] = row.vid;
                        h[1] = (int32_t)cnt;
                        h[2] = kth;
                    }
                }
            }
        }
        row = nrow;
        r = rn;
    }
    __syncthreads();
    if constexpr (HOSTOUT) {
        if (threadIdx.x == 0) ho.counts[bx] = s_nhits;
    } else {
        const int staged = s_nhits < kQ1Stage ? s_nhits : kQ1Stage;
        if (staged == 0) return;                                    // block-uniform
        if (threadIdx.x == 0) s_stage_base = atomicAdd(&hits_n[(size_t)q * ns], staged);
        __syncthreads();
        const int base = s_stage_base;
        int32_t *dst = hits + ((int64_t)q * cap + base) * 3;
        const int room = cap - base < staged ? (cap - base > 0 ? cap - base : 0) : staged;
        for (int i = threadIdx.x; i < room * 3; i += BS) dst[i] = s_stage[i];   // consecutive dwords: coalesced
    }
}

template <int MODE, bool HOSTOUT>
__global__ __launch_bounds__(kQ1Block) void ts_match_q1_kernel(
    const Row *__restrict__ rows, int64_t n_rows, const int64_t *__restrict__ keys,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t min_match,
    const int32_t *__restrict__ exclude_ids, int32_t exclude_one, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t ns, int32_t s_log2, HostOut ho,
    const QByVal qv) {
    q1_body<MODE, HOSTOUT, kQ1Block>(rows, n_rows, keys, queries, q_offsets, min_match, exclude_ids, exclude_one, cap,
                                     hits, hits_n, ns, s_log2, ho, qv, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}



// ---- per-query k best of a (long) hit list -------------------------------------------------
// Order: (kth, video_id, count) ascending.  A full bitonic sort of ~2,000 hits per query to keep
// 16 was most of the fixed cost of a sharded batch; instead a histogram of kth (LDS, 4098 bins)
// gives the smallest bin B whose prefix holds k hits, and only the hits in bins <= B (k plus the
// ties of one bin) are sorted.  Lists that are short anyway skip the histogram.
//   mode 0: plain.  mode 1 (shard side): the output has k+1 rows per query, row k carries the
//   true number of hits as (-1, n, NEVER) so one all-gather moves lists and totals together; n is
//   NEGATED when the shard's hit list overflowed its capacity (its top-k may then be inexact).
constexpr int kSelBins = 4098;                // kth -1 .. 4095 exactly, everything above shares the last
constexpr int kSelMin = 64;                   // lists up to this long are sorted directly (a 512-entry bitonic
                                              // sort per query was 24 us per 1024 queries on a 1/8 shard)
constexpr int kSelSmallK = 256;               // k up to this: 1024 candidates held at once (12 KiB; with the
                                              // histogram 29 KiB per block - 5 blocks per CU instead of 3)

__device__ __forceinline__ int sel_bin(int32_t kth) {
    const uint32_t b = (uint32_t)kth + 1u;    // -1 -> 0, NEVER -> 0x80000000
    return b < (uint32_t)(kSelBins - 1) ? (int)b : kSelBins - 1;
}

template <int kSelCap>
__global__ __launch_bounds__(kBlock) void ts_topk_select_kernel(
    const int32_t *__restrict__ lists, const int32_t *__restrict__ lists_n, int32_t ns, int32_t Q,
    int32_t cap, int32_t k, int32_t *__restrict__ topk, int32_t mode, const int32_t *__restrict__ flags) {
    constexpr int kSelChunk = kSelCap / 2;
    static_assert(kSelChunk % kBlock == 0, "whole passes of the block");
    __shared__ uint64_t key[kSelCap];
    __shared__ int32_t cnt[kSelCap];
    __shared__ uint32_t hist[kSelBins];
    __shared__ uint32_t part[kBlock];
    __shared__ int32_t s_pos, s_bin;
    // flags != NULL: ts_topk_wave_kernel went first and flagged the queries it left to this kernel
    for (int q = blockIdx.x; q < Q; q += gridDim.x) {
    if (flags && flags[q] == 0) continue;
    const int32_t total = lists_n ? lists_n[(size_t)q * ns] : cap;
    const bool overflow = total > cap;
    const int n = total > cap ? cap : (total < 0 ? 0 : total);
    const int32_t *src = lists + (int64_t)q * cap * 3;
    int limit = kSelBins - 1;                 // keep hits whose bin is <= limit
    if (threadIdx.x == 0) s_pos = 0;
    if (n > kSelMin) {
        for (int i = threadIdx.x; i < kSelBins; i += kBlock) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kBlock)
            if (src[i * 3] >= 0) atomicAdd(&hist[sel_bin(src[i * 3 + 2])], 1u);
        __syncthreads();
        constexpr int kPer = (kSelBins + kBlock - 1) / kBlock;      // bins per thread
        uint32_t s = 0;
        for (int b = threadIdx.x * kPer; b < (threadIdx.x + 1) * kPer && b < kSelBins; ++b) s += hist[b];
        part[threadIdx.x] = s;
        __syncthreads();
        // threshold bin = first bin whose inclusive prefix reaches k.  One wave: lane l owns the
        // partial sums of threads 4l..4l+3 (a serial walk by one thread was ~10 us of dependent LDS
        // reads per block - most of this kernel)
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            const uint32_t p0 = part[4 * l], p1 = part[4 * l + 1], p2 = part[4 * l + 2], p3 = part[4 * l + 3];
            uint32_t incl = p0 + p1 + p2 + p3;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (l >= d) incl += o;
            }
            const unsigned long long reach = __ballot(incl >= (uint32_t)k);
            const int owner = reach ? __ffsll((long long)reach) - 1 : 63;
            if (l == owner) {
                uint32_t cum = incl - (p0 + p1 + p2 + p3);
                int t = 4 * l;
                if (cum + p0 < (uint32_t)k) { cum += p0; ++t;
                    if (cum + p1 < (uint32_t)k) { cum += p1; ++t;
                        if (cum + p2 < (uint32_t)k) { cum += p2; ++t; } } }
                int b = t * kPer;
                while (b < kSelBins - 1 && cum + hist[b] < (uint32_t)k) cum += hist[b++];
                s_bin = b;
            }
        }
        __syncthreads();
        limit = s_bin;
    }
    int pos = 0;                              // block-uniform fill level
    auto sort_and_keep = [&]() {
        int P = 2;
        while (P < pos) P <<= 1;
        for (int i = pos + threadIdx.x; i < P; i += kBlock) { key[i] = ~0ULL; cnt[i] = 0; }
        bitonic_sort(key, cnt, P);
        if (pos > k) pos = k;
    };
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += kSelChunk) {
        const int i = j0 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < kSelChunk / kBlock; ++u) {
            const int ii = i + u * kBlock;
            if (ii < n && ii < j0 + kSelChunk) {
                const int32_t vid = src[ii * 3], kth = src[ii * 3 + 2];
                if (vid >= 0 && sel_bin(kth) <= limit) {
                    const int p = atomicAdd(&s_pos, 1);
                    key[p] = sort_key(vid, kth);
                    cnt[p] = src[ii * 3 + 1];
                }
            }
        }
        __syncthreads();
        pos = s_pos;
        if (pos > kSelCap - kSelChunk) {     // no room for another chunk: reduce to the k best
            sort_and_keep();
            if (threadIdx.x == 0) s_pos = pos;
            __syncthreads();
        }
    }
    __syncthreads();
    pos = s_pos;
    sort_and_keep();
    const int orows = (mode == 1) ? k + 1 : k;
    if (mode == 1 && threadIdx.x == 0) {
        int32_t t = total;
        if (overflow) t = -t;                 // negative total = the hit list was truncated
        int32_t *o = topk + ((int64_t)q * orows + k) * 3;
        o[0] = -1; o[1] = t; o[2] = TVZ_KTH_NEVER;
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
    __syncthreads();
    }
}

// ---- top-k of SHORT inputs: one wave per query, no block barrier ---------------------------
// A 1/8 shard's hit list (~260 hits per query) and the merge of the gathered per-rank lists
// (n_ranks x k entries) are a few hundred entries; a 256-thread block each, with ~30 block barriers,
// spent 30 us + 15 us per 4096-query batch on them - a third of a sharded batch - and most of that
// is the launch rate of 4096 blocks and barrier latency, not work.  Here a wave takes a query of up
// to 1024 entries (4, 8 or 16 per lane, in registers): a histogram of kth in the wave's own LDS (two 16-bit
// bins per word) gives the threshold bin; the entries up to that bin - k plus the ties of one bin,
// normally a handful more than k - are compacted one per lane and sorted by a bitonic network over
// the lanes (more than 64 of them: k rounds of wave-minimum instead).  Same order and output rows
// as ts_topk_select_kernel / ts_topk_kernel.  A list of more than 1024 entries is FLAGGED and left
// to the block kernel, which follows with a small grid and takes only the flagged queries.
constexpr int kWsE = 16;
constexpr int kWsMax = 64 * kWsE;
constexpr int kWsK = 64;                                      // lane i writes output row i
constexpr int kWsPerLane = ((kSelBins + 1) / 2 + 63) / 64;   // 33 words (66 bins) per lane
constexpr int kWsWords = kWsPerLane * 64;
static_assert(kWsWords * 2 >= kSelBins && (kWsPerLane & 1) == 1, "bins covered; odd stride = no bank conflicts");

// the per-wave work for E entries per lane; a wave picks the smallest E that holds its list
template <int E>
__device__ __forceinline__ void wave_topk_body(
    uint32_t *h, const int lane, const int q, const int n, const int32_t total_row,
    const int32_t *__restrict__ lists, int32_t n_lists, int32_t Q, int32_t cap, int32_t k,
    int32_t *__restrict__ topk, int32_t mode, int32_t *__restrict__ totals, int32_t *__restrict__ flags,
    int32_t hit_cap) {
    uint64_t key[E];
    int32_t cnt[E];
    int bin[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane + 64 * e;
        key[e] = ~0ULL;
        cnt[e] = 0;
        bin[e] = -1;                           // -1: no entry
        if (i < n) {
            const int32_t *src;
            if (mode >= 2) {
                const int l = i / (cap - 1), j = i - l * (cap - 1);
                src = lists + (((int64_t)l * Q + q) * cap + j) * 3;
            } else {
                src = lists + ((int64_t)q * cap + i) * 3;
            }
            const int32_t vid = src[0];
            if (vid >= 0) {
                key[e] = sort_key(vid, src[2]);
                cnt[e] = src[1];
                bin[e] = sel_bin(src[2]);
            }
        }
    }
#pragma unroll
    for (int w = 0; w < kWsPerLane; ++w) h[lane + 64 * w] = 0;
    wave_lds_fence();
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (bin[e] >= 0) atomicAdd(&h[bin[e] >> 1], 1u << ((bin[e] & 1) * 16));
    wave_lds_fence();
    // lane l owns words 33 l .. 33 l + 32 (bins 66 l .. 66 l + 65)
    uint32_t mine = 0;
#pragma unroll
    for (int w = 0; w < kWsPerLane; ++w) {
        const uint32_t v = h[lane * kWsPerLane + w];
        mine += (v & 0xffffu) + (v >> 16);
    }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    const uint32_t tot = __shfl(incl, 63);
    const uint32_t need = tot < (uint32_t)k ? tot : (uint32_t)k;
    int n_cand = 0;
    uint64_t mk = ~0ULL;
    int32_t mc = 0;
    bool fits = true;
    if (tot) {
        // threshold bin B: the first whose inclusive prefix reaches `need`; the owner lane walks its bins
        const unsigned long long reach = __ballot(incl >= need);
        const int owner = __ffsll((long long)reach) - 1;
        // the owner's 33 words are re-read one per lane (lane j: word j of the owner), a scan over the
        // lanes finds the bin (a serial walk by the owner alone was 500 of this kernel's 1,700 instructions)
        const uint32_t before = __shfl(incl - mine, owner);          // entries in the bins of lower lanes
        const uint32_t wv = lane < kWsPerLane ? h[owner * kWsPerLane + lane] : 0u;
        const uint32_t c0 = wv & 0xffffu, c1 = wv >> 16;
        uint32_t wincl = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(wincl, d);
            if (lane >= d) wincl += o;
        }
        const unsigned long long wreach = __ballot(before + wincl >= need);
        const int wl = __ffsll((long long)wreach) - 1;                // the word that holds bin B
        const uint32_t below = before + wincl - (c0 + c1);            // entries before that word (lane wl's view)
        const bool first = below + c0 >= need;                        // B is the word's low bin?
        const int B = __shfl(2 * (owner * kWsPerLane + lane) + (first ? 0 : 1), wl);
        const uint32_t upto = __shfl(below + c0 + (first ? 0u : c1), wl);
        fits = upto <= 64u;
        if (fits) {
            wave_lds_fence();                  // the histogram is dead: its first words become the candidate list
            uint64_t *ck = reinterpret_cast<uint64_t *>(h);          // [64] keys
            int32_t *cc = reinterpret_cast<int32_t *>(h + 128);      // [64] counts
            uint32_t base = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool is = bin[e] >= 0 && bin[e] <= B;
                const unsigned long long bal = __ballot(is);
                const uint32_t ofs = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (is) { ck[base + ofs] = key[e]; cc[base + ofs] = cnt[e]; }
                base += (uint32_t)__popcll(bal);
            }
            wave_lds_fence();
            n_cand = (int)upto;
            if (lane < n_cand) { mk = ck[lane]; mc = cc[lane]; }
            // bitonic network over the 64 lanes, ascending by (key, count)
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const uint64_t ok = __shfl_xor(mk, stride);
                    const int32_t oc = __shfl_xor(mc, stride);
                    const bool keep_min = ((lane & stride) == 0) == ((lane & size) == 0);
                    const bool other_less = ok < mk || (ok == mk && oc < mc);
                    const bool other_more = ok > mk || (ok == mk && oc > mc);
                    if (keep_min ? other_less : other_more) { mk = ok; mc = oc; }
                }
            }
        }
    }
    if (flags && lane == 0) flags[q] = 0;
    if (!fits) {
        // more than 64 entries up to the threshold bin (a big tie: e.g. hundreds of true duplicates
        // with the same kth).  Rare, so simple: k rounds, each takes the minimum of what is left -
        // per-lane minimum, DPP butterfly inside the 16-lane rows, the four rows through scalar
        // registers - and removes exactly one copy of it.  Lane r keeps output row r.
        mk = ~0ULL;
        mc = 0;
        for (int r = 0; r < k; ++r) {
            uint64_t bk = key[0];
            int32_t bc = cnt[0];
#pragma unroll
            for (int e = 1; e < E; ++e)
                if (key[e] < bk || (key[e] == bk && cnt[e] < bc)) { bk = key[e]; bc = cnt[e]; }
            uint64_t wk = bk;
            int32_t wc = bc;
#define TVZ_MIN_STEP(CTRL)                                                               \
            {                                                                              \
                const uint64_t ok = dpp16_64<CTRL>(wk);                                    \
                const int32_t oc = (int32_t)dpp16<CTRL>((uint32_t)wc);                     \
                if (ok < wk || (ok == wk && oc < wc)) { wk = ok; wc = oc; }                \
            }
            TVZ_ROW16_BUTTERFLY(TVZ_MIN_STEP)
#undef TVZ_MIN_STEP
            uint64_t rk = ~0ULL;
            int32_t rc = 0;
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wk, row * 16);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wk >> 32), row * 16);
                const int32_t oc = __builtin_amdgcn_readlane(wc, row * 16);
                const uint64_t ok = ((uint64_t)hi << 32) | lo;
                if (row == 0 || ok < rk || (ok == rk && oc < rc)) { rk = ok; rc = oc; }
            }
            if (rk == ~0ULL) break;            // nothing left: the remaining rows are padding
            if (lane == r) { mk = rk; mc = rc; }
            const unsigned long long holders = __ballot(bk == rk && bc == rc);
            if (lane == __ffsll((long long)holders) - 1) {
                bool gone = false;
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (!gone && key[e] == rk && cnt[e] == rc) { key[e] = ~0ULL; gone = true; }
            }
        }
    }
    const int orows = (mode == 1 || mode == 3) ? k + 1 : k;
    if (lane < k) {
        int32_t *o = topk + ((int64_t)q * orows + lane) * 3;
        if (mk == ~0ULL) {
            o[0] = -1; o[1] = 0; o[2] = TVZ_KTH_NEVER;
        } else {
            o[0] = (int32_t)(uint32_t)mk;
            o[1] = mc;
            o[2] = (int32_t)(uint32_t)(mk >> 32) - 1;
        }
    }
    if (mode == 1 && lane == 0) {
        int32_t *o = topk + ((int64_t)q * orows + k) * 3;
        o[0] = -1; o[1] = total_row; o[2] = TVZ_KTH_NEVER;
    }
    if ((mode == 2 && totals) || mode == 3) {
        // every gathered list ends with (-1, n, NEVER): |n| summed, negated if any shard overflowed
        // (mode 3: or if the sum exceeds hit_cap - the one list an unfused match would have filled)
        long long sum = 0;
        bool over = false;
        for (int l = lane; l < n_lists; l += 64) {
            const int32_t t = lists[(((int64_t)l * Q + q) * cap + (cap - 1)) * 3 + 1];
            sum += t < 0 ? -(long long)t : t;
            over = over || t < 0;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
        over = __ballot(over) != 0ULL || (mode == 3 && sum > (long long)hit_cap);
        if (lane == 0) {
            int32_t t = sum > 0x7fffffffLL ? 0x7fffffff : (int32_t)sum;
            if (over) t = (t == 0) ? INT32_MIN : -t;
            if (mode == 3) {
                int32_t *o = topk + ((int64_t)q * orows + k) * 3;
                o[0] = -1; o[1] = t; o[2] = TVZ_KTH_NEVER;
            } else {
                totals[q] = t;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void ts_topk_wave_kernel(
    const int32_t *__restrict__ lists, const int32_t *__restrict__ lists_n, int32_t ns, int32_t n_lists,
    int32_t Q, int32_t cap, int32_t k, int32_t *__restrict__ topk, int32_t mode,
    int32_t *__restrict__ totals, int32_t *__restrict__ flags) {
    __shared__ uint32_t s_hist[kBlock / 64][kWsWords];
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (q >= Q) return;                        // no block barrier below: waves are on their own
    uint32_t *h = s_hist[threadIdx.x >> 6];
    int n;                                     // entries to look at
    int32_t total_row = 0;                     // mode 1: the shard's hit count (negated on overflow)
    if (mode >= 2) {                           // 2: merge of gathered blocks -> topk[Q][k] + totals[Q]; 3: -> one block [Q][k+1]
        n = n_lists * (cap - 1);               // cap = k + 1 rows per gathered list, the last one = totals
    } else {
        const int32_t total = lists_n ? lists_n[(size_t)q * ns] : cap;
        n = total > cap ? cap : (total < 0 ? 0 : total);
        total_row = total > cap ? -total : total;
    }
    if (n > kWsMax) {                          // the block kernel's (the host passes flags whenever this can happen)
        if (lane == 0) flags[q] = 1;
        return;
    }
    if (n <= 64 * 4)
        wave_topk_body<4>(h, lane, q, n, total_row, lists, n_lists, Q, cap, k, topk, mode, totals, flags, ns);
    else if (n <= 64 * 8)
        wave_topk_body<8>(h, lane, q, n, total_row, lists, n_lists, Q, cap, k, topk, mode, totals, flags, ns);
    else
        wave_topk_body<kWsE>(h, lane, q, n, total_row, lists, n_lists, Q, cap, k, topk, mode, totals, flags, ns);
}

// ---- merge of the gathered per-rank blocks when they are SORTED (they are: tvz_match_topk / tvz_topk_shard write
// their k rows in ascending (kth, video_id, count) order) and there are at most 16 of them ----------------------------
// The one-wave kernel above treats the R x k gathered entries as an unordered set (histogram, compaction, 64-lane
// bitonic network: ~720 VALU instructions per query) - 15 % of a sharded batch's instructions on a 1/8 shard, on the
// stream that shares the GPU with the next batch's lookup.  Sorted inputs need a k-way merge only: a group of G =
// 2^ceil(log2 R) lanes takes one query, lane r walks list r (staged in LDS, one private run per lane: no
// synchronisation), and each of the k steps min-reduces the G heads by a DPP butterfly inside the group; the winner
// writes output row t and moves to its next entry.  64 / G queries per wave: ~50 instructions per query at R = 8,
// a plain copy at R = 1.  Totals: |n| summed over the ranks, negated if any rank's list overflowed (as above).
template <int G>
__global__ __launch_bounds__(64) void ts_topk_merge_sorted_kernel(const int32_t *__restrict__ lists, int32_t n_lists,
                                                                  int32_t Q, int32_t k, int32_t *__restrict__ topk,
                                                                  int32_t *__restrict__ totals) {
    extern __shared__ int32_t s_lists[];                   // [64][3 k + 1] (the odd stride spreads the lanes over the banks)
    const int lane = threadIdx.x;
    const int q = (int)blockIdx.x * (64 / G) + lane / G;
    const int r = lane % G;
    const bool live = q < Q && r < n_lists;
    int32_t *mine = s_lists + (size_t)lane * (3 * k + 1);
    int32_t t_r = 0;
    if (live) {
        const int32_t *src = lists + (((int64_t)r * Q + q) * (k + 1)) * 3;
        t_r = src[3 * k + 1];
        // twelve loads in flight at a time (k is a run-time value: the plain loop was 3 k dependent round trips)
        int i = 0;
        for (; i + 12 <= 3 * k; i += 12) {
            int32_t v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = src[i + j];
#pragma unroll
            for (int j = 0; j < 12; ++j) mine[i + j] = v[j];
        }
        for (; i < 3 * k; ++i) mine[i] = src[i];
    }
    // hit totals over the ranks (each butterfly step adds two disjoint sets of lanes)
    unsigned long long sum = t_r < 0 ? (unsigned long long)(-(long long)t_r) : (unsigned long long)t_r;
    uint32_t over = t_r < 0 ? 1u : 0u;
#define TVZ_MS_SUM(C) { sum += dpp16_64<C>(sum); over |= dpp16<C>(over); }
    if (G >= 2) TVZ_MS_SUM(0xB1)
    if (G >= 4) TVZ_MS_SUM(0x4E)
    if (G >= 8) TVZ_MS_SUM(0x141)
    if (G >= 16) TVZ_MS_SUM(0x140)
#undef TVZ_MS_SUM
    if (q < Q && r == 0 && totals) {
        int32_t t = sum > 0x7fffffffULL ? 0x7fffffff : (int32_t)sum;
        if (over) t = (t == 0) ? INT32_MIN : -t;
        totals[q] = t;
    }
    // the k-way merge: heads compared as (kth + 1, video_id, count, rank) - the order of the top-k kernels, made
    // unique inside a group by the rank.  Two 64-bit words per head and bitwise logic on the comparisons: the
    // short-circuit form compiled to a ladder of branches per butterfly step (and a lambda that captured the head
    // by reference put it in scratch memory: two scratch loads per output row).
    int p = 0;
    unsigned long long ha, hb;                                 // (kth + 1) << 32 | video_id ; count << 32 | rank
#define TVZ_MS_HEAD() do { \
        const int pp = p < k ? p : k - 1; \
        const int32_t v0 = mine[3 * pp], v1 = mine[3 * pp + 1], v2 = mine[3 * pp + 2]; \
        const bool ok = live & (p < k) & (v0 >= 0);            /* exhausted, or padding: the rest of a sorted list is padding too */ \
        ha = ok ? ((unsigned long long)((uint32_t)v2 + 1u) << 32) | (uint32_t)v0 : ~0ULL; \
        hb = ok ? ((unsigned long long)(uint32_t)v1 << 32) | (uint32_t)r : ~0ULL; \
    } while (0)
    TVZ_MS_HEAD();
    for (int t = 0; t < k; ++t) {
        unsigned long long ma = ha, mb = hb;
#define TVZ_MS_MIN(C) { const unsigned long long oa = dpp16_64<C>(ma), ob = dpp16_64<C>(mb); \
                        const bool lt = (oa < ma) | ((oa == ma) & (ob < mb)); \
                        ma = lt ? oa : ma; mb = lt ? ob : mb; }
        if (G >= 2) TVZ_MS_MIN(0xB1)
        if (G >= 4) TVZ_MS_MIN(0x4E)
        if (G >= 8) TVZ_MS_MIN(0x141)
        if (G >= 16) TVZ_MS_MIN(0x140)
#undef TVZ_MS_MIN
        if (q < Q) {
            int32_t *o = topk + ((int64_t)q * k + t) * 3;
            if (ma == ~0ULL) {                                 // every list is exhausted: padding from here on
                if (r == 0) { o[0] = -1; o[1] = 0; o[2] = TVZ_KTH_NEVER; }
            } else if ((uint32_t)mb == (uint32_t)r) {          // this lane's head is the smallest
                o[0] = (int32_t)(uint32_t)ma; o[1] = (int32_t)(mb >> 32); o[2] = (int32_t)((uint32_t)(ma >> 32) - 1u);
                ++p;
                TVZ_MS_HEAD();
            }
        }
    }
#undef TVZ_MS_HEAD
}

// ---- small helpers launched around the sweeps ------------------------------------------------
// hit counters = 0, hash-join key tables = 0xff.. (kJEmpty), presence bitmaps = 0: one launch
__global__ __launch_bounds__(kBlock) void ts_prep_kernel(int32_t *__restrict__ hits_n, int32_t ns, int32_t Q,
                                                         uint4 *__restrict__ ones16, size_t n_ones16,
                                                         uint4 *__restrict__ zero16, size_t n_zero16) {
    const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x, step = (size_t)gridDim.x * kBlock;
    for (size_t i = i0; i < (size_t)Q; i += step) hits_n[i * ns] = 0;
    for (size_t i = i0; i < n_ones16; i += step) ones16[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (size_t i = i0; i < n_zero16; i += step) zero16[i] = make_uint4(0, 0, 0, 0);
}

// Hit counters live one per 128-byte line while the sweeps append (kCountStride int32 apart): a
// returning atomic is serialised per cache LINE at the memory side, and 1024 adjacent int32
// counters are 32 lines - with 2 M hits per batch that alone was ~1 ms, whatever the sweep did
// (profiles/r2c_match_pmc.txt).  This copies them to the caller's dense array afterwards.
constexpr int kCountStride = 32;
__global__ __launch_bounds__(kBlock) void ts_counts_gather_kernel(const int32_t *__restrict__ padded, int32_t ns,
                                                                  int32_t *__restrict__ dense, int32_t Q) {
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q < Q) dense[q] = padded[(size_t)q * ns];
}

// upsert: the row entry is swapped by ONE 16-byte store, ordered on the mutation stream behind
// the copy of the row's new keys
__global__ void ts_row_write_kernel(Row *dst, Row v) {
    int4 w;
    w.x = (int32_t)(uint32_t)(uint64_t)v.off;
    w.y = (int32_t)(uint32_t)((uint64_t)v.off >> 32);
    w.z = v.len;
    w.w = v.vid;
    *reinterpret_cast<int4 *>(dst) = w;
}

}  // namespace
