// tvz_index_kernels.h — inverted index over the device corpus (gfx950, wave64).
//
// find_duplicates (/root/reference inspector/db.py:76-94) asks, for every corpus row, how many
// query elements are `in` the row.  The sweep kernels answer by reading every row; the index
// answers from the other side: for every query element, which rows contain it.
//
// The indexed rows are cut into SUB-INDEXES of 2^14 rows.  Per sub-index:
//   dir  : open-addressing directory of the distinct canonical keys of its rows, 16 B per entry
//          {key, first posting, number of postings}, load <= 0.5 (all directories the same size)
//   post : posting lists, one uint16 LOCAL row number per (row, key) pair, contiguous per key
// and over all of them
//   ivid : video_id per indexed row; -1 once the row was replaced by an upsert (its postings are
//          then stale and ignored; the row's current content lives in the delta table, which the
//          sweep kernels read).
//
// One block per (query, sub-index).  A query of n elements whose keys have p postings in the
// sub-index costs n directory probes + p two-byte posting reads, whatever the corpus size
// (config 4: ~16,500 postings = 33 KB per query over 7 sub-indexes, against 160 MB for a sweep):
//   pass A  every posting sets its row's bit in `seen1`, or in `seen2` if seen1 was set already:
//           only rows in seen2 (seen1 for min_match 1) can reach min_match; the bitmaps cover the
//           sub-index exactly (16,384 bits), so the candidates are known row by row;
//   rank    prefix popcount of the candidate bitmap: candidate -> dense slot, no hashing;
//   pass B  the postings are walked again - from the (row, position) pairs pass A left in LDS (the
//           first 4,096; the lists themselves beyond that); a candidate's count and smallest query
//           positions (two atomicMin words for min_match <= 2, five 12-bit positions in a CAS word
//           for 3..5) accumulate in its slot - 1,024 slots at a time if there are more candidates;
//   emit    candidates with count >= min_match, (video_id, count, kth) exactly as the sweeps emit
//           them: one reservation per block in the query's hit list (or the block's own region of
//           pinned host memory for tvz_find_duplicates).
// The walks are laid out so that every wave owns a contiguous range of the flattened postings:
// consecutive lanes read consecutive postings and a lane's list pointer only moves forward.
//
// Build (on the corpus' mutation stream, readers drained): count postings per key with
// find-or-insert, hand out posting ranges with one atomic per wave, fill.  No sort.
#pragma once
#include "tvz_match_kernels.h"

namespace {

struct alignas(16) DirEnt {
    int64_t key;       // kEmpty = free
    uint32_t off;      // first posting
    uint32_t len;      // postings
};
static_assert(sizeof(DirEnt) == 16, "DirEnt must be 16 bytes");

struct IxBuildInfo {       // device-side build status, read back by the host
    uint32_t cursor;       // postings handed out
    uint32_t max_distinct; // most directory entries in use in one sub-index
    uint32_t failed;       // a probe sequence ran too long: directories too small, rebuild larger
    uint32_t pad;
};

constexpr int kIxMaxProbe = 4096;
#ifndef TVZ_IX_SUB_LOG2
#define TVZ_IX_SUB_LOG2 14
#endif
constexpr int kSubLog2 = TVZ_IX_SUB_LOG2;            // rows per sub-index (local row numbers are uint16)
constexpr int kSubRows = 1 << kSubLog2;

__device__ __forceinline__ uint32_t ix_slot(int64_t k, int dir_log2) {
    return (q1_mix(k) * 0x9E3779B1u) >> (32 - dir_log2);
}

// directories = every key free; per-sub-index distinct counters = 0
__global__ __launch_bounds__(kBlock) void ix_clear_kernel(DirEnt *__restrict__ dir, size_t n,
                                                          uint32_t *__restrict__ sub_distinct, int n_sub,
                                                          IxBuildInfo *info) {
    const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x, step = (size_t)gridDim.x * kBlock;
    int4 e;
    e.x = (int32_t)(uint32_t)(uint64_t)kEmpty;
    e.y = (int32_t)(uint32_t)((uint64_t)kEmpty >> 32);
    e.z = 0;
    e.w = 0;
    for (size_t i = i0; i < n; i += step) reinterpret_cast<int4 *>(dir)[i] = e;
    for (size_t i = i0; i < (size_t)n_sub; i += step) sub_distinct[i] = 0;
    if (i0 == 0) { info->cursor = 0; info->max_distinct = 0; info->failed = 0; info->pad = 0; }
}

// find (or, with INSERT, claim) the directory entry of key k; returns the slot or -1
template <bool INSERT>
__device__ __forceinline__ int64_t ix_find(DirEnt *dir, int dir_log2, int64_t k, bool &is_new) {
    const uint32_t mask = (1u << dir_log2) - 1u;
    uint32_t s = ix_slot(k, dir_log2);
    is_new = false;
    for (int probes = 0; probes < kIxMaxProbe; ++probes) {
        // look first: keys only ever go from free to taken, a stale view is corrected by the CAS
        const int64_t cur = INSERT ? *reinterpret_cast<volatile int64_t *>(&dir[s].key) : dir[s].key;
        if (cur == k) return s;
        if (cur == kEmpty) {
            if (!INSERT) return -1;
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&dir[s].key),
                                                     (unsigned long long)kEmpty, (unsigned long long)k);
            if (old == (unsigned long long)kEmpty) { is_new = true; return s; }
            if ((int64_t)old == k) return s;
        }
        s = (s + 1) & mask;
    }
    return -1;
}

// one wave per row: count the postings of every key, remember the row's video id
__global__ __launch_bounds__(kBlock) void ix_count_kernel(const Row *__restrict__ rows, int64_t n_rows,
                                                          const int64_t *__restrict__ keys, DirEnt *dir,
                                                          int dir_log2, int32_t *__restrict__ ivid,
                                                          uint32_t *__restrict__ sub_distinct,
                                                          IxBuildInfo *info) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < n_rows;
         r += (int64_t)gridDim.x * (kBlock / 64)) {
        const Row row = load_row(rows + r);
        if (lane == 0) ivid[r] = row.vid;
        DirEnt *d = dir + ((size_t)(r >> kSubLog2) << dir_log2);
        uint32_t mine = 0;
        for (int i = lane; i < row.len; i += 64) {
            bool is_new;
            const int64_t s = ix_find<true>(d, dir_log2, keys[row.off + i], is_new);
            if (s < 0) { info->failed = 1; continue; }
            atomicAdd(&d[s].len, 1u);
            mine += is_new ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if (lane == 0 && mine) {
            const uint32_t now = atomicAdd(&sub_distinct[r >> kSubLog2], mine) + mine;
            atomicMax(&info->max_distinct, now);
        }
    }
}

// hand out posting ranges: off = END of the range (the fill pass counts it down to the start)
__global__ __launch_bounds__(kBlock) void ix_offsets_kernel(DirEnt *dir, size_t n, IxBuildInfo *info) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t len = i < n ? dir[i].len : 0u;
    uint32_t incl = len;                               // inclusive prefix over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    const uint32_t total = __shfl(incl, 63);
    uint32_t base = 0;
    if (lane == 0 && total) base = atomicAdd(&info->cursor, total);
    base = __shfl(base, 0);
    if (i < n && len) dir[i].off = base + incl;
}

__global__ __launch_bounds__(kBlock) void ix_fill_kernel(const Row *__restrict__ rows, int64_t n_rows,
                                                         const int64_t *__restrict__ keys, DirEnt *dir,
                                                         int dir_log2, uint16_t *__restrict__ post) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < n_rows;
         r += (int64_t)gridDim.x * (kBlock / 64)) {
        const Row row = load_row(rows + r);
        DirEnt *d = dir + ((size_t)(r >> kSubLog2) << dir_log2);
        for (int i = lane; i < row.len; i += 64) {
            bool is_new;
            const int64_t s = ix_find<false>(d, dir_log2, keys[row.off + i], is_new);
            if (s < 0) continue;                       // cannot happen after a successful count pass
            const uint32_t p = atomicSub(&d[s].off, 1u) - 1u;
            post[p] = (uint16_t)(r & (kSubRows - 1));
        }
    }
}

// upsert with an index: current row table entry, delta table entry, and (first time an indexed row
// changes) the dead mark of its stale postings - one launch on the mutation stream
__global__ void ts_row_write3_kernel(Row *dst, Row *delta_dst, int32_t *ivid_dead, Row v) {
    int4 w;
    w.x = (int32_t)(uint32_t)(uint64_t)v.off;
    w.y = (int32_t)(uint32_t)((uint64_t)v.off >> 32);
    w.z = v.len;
    w.w = v.vid;
    *reinterpret_cast<int4 *>(dst) = w;
    if (delta_dst) *reinterpret_cast<int4 *>(delta_dst) = w;
    if (ivid_dead) *ivid_dead = -1;
}

// ---- lookup: one block per (query, sub-index) -------------------------------------------------
#ifndef TVZ_IX_SLOT_BITS
#define TVZ_IX_SLOT_BITS 10
#endif
#ifndef TVZ_IX_CACHE
#define TVZ_IX_CACHE 4096
#endif
#ifndef TVZ_IX_BLOCK
#define TVZ_IX_BLOCK 512
#endif
constexpr int kIxWords = kSubRows / 32;              // words per bitmap
constexpr int kIxBlock = TVZ_IX_BLOCK;
constexpr int kIxWpt = kIxWords / kIxBlock;          // bitmap words per thread (rank, emit)
constexpr int kIxWaves = kIxBlock / 64;
static_assert(kIxWpt >= 1 && kIxWpt * kIxBlock == kIxWords, "whole bitmap words per thread");
constexpr int kIxSlotBits = TVZ_IX_SLOT_BITS;
constexpr int kIxSlots = 1 << kIxSlotBits;           // candidate slots per part (12 B each)
constexpr int kIxCache = TVZ_IX_CACHE;               // (row, position) of postings kept in LDS between the passes
static_assert(kSubLog2 <= 16 && kSubLog2 + 12 <= 32 && kSubLog2 + kIxSlotBits <= 32, "packed LDS entries");

// dynamic LDS: [bm1][bm2][rank][tcnt][ttop][elist][cache][s_off[L]][s_pre[L + 1]][s_pos[L]] - 76 KiB + 10 L:
// two 16-wave blocks (32 waves) per CU for queries of up to ~400 timestamps
inline size_t ix_lds_bytes(int max_len) {
    return (size_t)3 * kIxWords * 4 + (size_t)kIxSlots * 16 + (size_t)kIxCache * 4 +
           (size_t)(2 * max_len + 2) * 4 + (size_t)(max_len + 2) * 2;
}

// TOP5 = false (min_match 1..2): a slot keeps the two smallest positions in two atomicMin words -
// two plain LDS atomics per candidate posting instead of the 5 x 12-bit CAS loop (min_match 3..5).
template <bool HOSTOUT, bool TOP5>
__global__ __launch_bounds__(kIxBlock) void ts_match_index_kernel(
    const DirEnt *__restrict__ dir_all, int dir_log2, const uint16_t *__restrict__ post,
    const int32_t *__restrict__ ivid, int64_t n_indexed, const double *__restrict__ queries,
    const int64_t *__restrict__ q_offsets, int32_t max_len, int32_t min_match,
    const int32_t *__restrict__ exclude_ids, int32_t exclude_one, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t ns, const QByVal qv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *bm1 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *bm2 = bm1 + kIxWords;
    uint32_t *rank = bm2 + kIxWords;                                    // candidates before word j
    uint32_t *tcnt = rank + kIxWords;
    unsigned long long *ttop = reinterpret_cast<unsigned long long *>(tcnt + kIxSlots);
    uint32_t *m1 = reinterpret_cast<uint32_t *>(ttop), *m2 = m1 + kIxSlots;   // !TOP5: the same 8 B per slot
    uint32_t *elist = reinterpret_cast<uint32_t *>(ttop + kIxSlots);    // emit list of one part
    uint32_t *pcache = elist + kIxSlots;                                // first kIxCache postings: row | position << kSubLog2
    uint32_t *s_off = reinterpret_cast<uint32_t *>(pcache + kIxCache);  // first posting of list j
    uint32_t *s_pre = s_off + max_len;                                  // [m + 1] postings before list j
    uint16_t *s_pos = reinterpret_cast<uint16_t *>(s_pre + max_len + 1);  // query position of list j
    __shared__ uint32_t s_wsum[kIxWaves], s_lsum[kIxWaves];
    __shared__ uint32_t s_base, s_m, s_total, s_nlist;
    __shared__ int32_t s_emitted;

    const int q = blockIdx.x;
    const int sub = blockIdx.y;
    const bool byval = q_offsets == nullptr;
    const int64_t qo = byval ? 0 : q_offsets[q];
    const int64_t n64 = byval ? qv.n : q_offsets[q + 1] - qo;
    // HOSTOUT: the block's own hit region [sub][kSubRows][3] and count in pinned host memory
    int32_t *out_n = HOSTOUT ? hits_n + sub : hits_n + (size_t)q * ns;
    int32_t *out_hits = HOSTOUT ? hits + (int64_t)sub * kSubRows * 3 : hits + (int64_t)q * cap * 3;
    if (n64 > max_len) {       // max_query_len was not an upper bound (the LDS arrays are sized from it)
        if (threadIdx.x == 0) *out_n = INT32_MIN;
        return;
    }
    const int n = (int)n64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_m = 0; s_total = 0; s_emitted = 0; s_nlist = 0; }
    for (int i = threadIdx.x; i < kIxWords; i += kIxBlock) { bm1[i] = 0; bm2[i] = 0; }
    for (int i = threadIdx.x; i < kIxSlots; i += kIxBlock) {
        tcnt[i] = 0;
        if (TOP5) ttop[i] = kTopNone; else { m1[i] = 0xffffffffu; m2[i] = 0xffffffffu; }
    }
    __syncthreads();

    // ---- directory: the NON-EMPTY posting lists of the query's positions, compacted ----
    const DirEnt *dir = dir_all + ((size_t)sub << dir_log2);
    const uint32_t dmask = (1u << dir_log2) - 1u;
    for (int i0 = 0; i0 < n; i0 += kIxBlock) {
        const int i = i0 + threadIdx.x;
        uint32_t off = 0, len = 0;
        int64_t k;
        if (i < n && canon_key(byval ? qv.k[i] : queries[qo + i], k)) {   // NaN never matches
            uint32_t s = ix_slot(k, dir_log2);
            for (int probes = 0; probes < kIxMaxProbe; ++probes) {
                const int4 e = *reinterpret_cast<const int4 *>(dir + s);
                const int64_t ek = (int64_t)(((uint64_t)(uint32_t)e.y << 32) | (uint32_t)e.x);
                if (ek == k) { off = (uint32_t)e.z; len = (uint32_t)e.w; break; }
                if (ek == kEmpty) break;
                s = (s + 1) & dmask;
            }
        }
        // block-wide exclusive prefixes of (len, len != 0), continued from the previous chunk
        uint32_t incl = len, lincl = len ? 1u : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d), lo = __shfl_up(lincl, d);
            if (lane >= d) { incl += o; lincl += lo; }
        }
        if (lane == 63) { s_wsum[wave] = incl; s_lsum[wave] = lincl; }
        __syncthreads();
        uint32_t pbase = s_total, lbase = s_m;
        for (int w = 0; w < wave; ++w) { pbase += s_wsum[w]; lbase += s_lsum[w]; }
        if (len) {
            const uint32_t j = lbase + lincl - 1u;
            s_off[j] = off;
            s_pre[j] = pbase + incl - len;
            s_pos[j] = (uint16_t)i;
        }
        __syncthreads();
        if (threadIdx.x == kIxBlock - 1) { s_total = pbase + incl; s_m = lbase + lincl; }
        __syncthreads();
    }
    const int m = (int)s_m;
    const uint32_t total = s_total;                    // <= 4095 lists x 32768 rows: fits 32 bits
    if (threadIdx.x == 0) s_pre[m] = total;
    __syncthreads();

    // every wave owns a contiguous range of the flattened postings; a lane's list pointer only
    // moves forward (by a bounded binary search when it has to skip short lists)
    const uint32_t w_lo = (uint32_t)(((unsigned long long)total * wave) / kIxWaves);
    const uint32_t w_hi = (uint32_t)(((unsigned long long)total * (wave + 1)) / kIxWaves);
    auto seek = [&](int j, uint32_t t) -> int {        // largest j' >= j with s_pre[j'] <= t
        if (s_pre[j + 1] > t) return j;
        int lo = j + 1, hi = m;                        // s_pre[lo] <= t < s_pre[hi] = total
        if (hi - lo > 64 && s_pre[lo + 64] > t) hi = lo + 64;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid] <= t) lo = mid; else hi = mid;
        }
        return lo;
    };
    // ---- pass A: which rows are touched (twice); eight posting loads in flight per lane ----
    {
        int j = 0;
        for (uint32_t t0 = w_lo; t0 < w_hi; t0 += 512u) {   // 8 x 64 postings
            uint32_t r[8];
            uint16_t pj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // unconditional (clamped) loads: a branch around a load makes the compiler wait for
                // every load before it
                const uint32_t t = t0 + (uint32_t)(u * 64 + lane);
                const uint32_t tc = t < w_hi ? t : w_hi - 1u;
                j = seek(j, tc);
                const uint32_t v = post[s_off[j] + (tc - s_pre[j])];
                r[u] = t < w_hi ? v : 0xffffffffu;
                pj[u] = s_pos[j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (r[u] == 0xffffffffu) continue;
                const uint32_t t = t0 + (uint32_t)(u * 64 + lane);
                if (t < (uint32_t)kIxCache) pcache[t] = r[u] | ((uint32_t)pj[u] << kSubLog2);
                const uint32_t bit = 1u << (r[u] & 31u);
                const uint32_t old = atomicOr(&bm1[r[u] >> 5], bit);
                if (min_match >= 2 && (old & bit)) atomicOr(&bm2[r[u] >> 5], bit);
            }
        }
    }
    __syncthreads();
    const uint32_t *cand = min_match >= 2 ? bm2 : bm1;
    // ---- rank: candidates before every bitmap word (thread t owns words t*kIxWpt .. +kIxWpt-1) ----
    {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < kIxWpt; ++w) c += __popc(cand[threadIdx.x * kIxWpt + w]);
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += s_wsum[w];
        uint32_t run = base + incl - c;
#pragma unroll
        for (int w = 0; w < kIxWpt; ++w) {
            rank[threadIdx.x * kIxWpt + w] = run;
            run += __popc(cand[threadIdx.x * kIxWpt + w]);
        }
        if (threadIdx.x == kIxBlock - 1) s_base = base + incl;         // candidates in all
        __syncthreads();
    }
    static_assert(kIxSlots % kIxBlock == 0, "whole rounds of the block over the emit list");
    const uint32_t n_cand = s_base;
    const int32_t excl = exclude_ids ? exclude_ids[q] : exclude_one;
    const int64_t row0 = (int64_t)sub << kSubLog2;

    // ---- pass B + emit, kIxSlots candidates at a time ----
    for (uint32_t lo = 0; lo < n_cand; lo += kIxSlots) {
        {
            int j = 0;
            for (uint32_t t = w_lo + lane; t < w_hi; t += 64u) {
                uint32_t r, pos;
                if (t < (uint32_t)kIxCache) {
                    const uint32_t e = pcache[t];
                    r = e & (uint32_t)(kSubRows - 1);
                    pos = e >> kSubLog2;
                } else {                                                  // beyond the LDS copy: read it again
                    j = seek(j, t);
                    r = post[s_off[j] + (t - s_pre[j])];
                    pos = s_pos[j];
                }
                const uint32_t w = cand[r >> 5], bit = r & 31u;
                if (!((w >> bit) & 1u)) continue;
                const uint32_t idx = rank[r >> 5] + __popc(w & ((1u << bit) - 1u)) - lo;
                if (idx >= (uint32_t)kIxSlots) continue;                 // another part's (wraps below lo)
                atomicAdd(&tcnt[idx], 1u);
                if constexpr (TOP5) {
                    unsigned long long seen = ttop[idx];
                    while (true) {
                        if (((uint32_t)(seen >> (12 * (kTop - 1))) & 0xfffu) <= pos) break;   // not among the 5 smallest
                        const unsigned long long old = atomicCAS(&ttop[idx], seen, top5_insert(seen, pos));
                        if (old == seen) break;
                        seen = old;
                    }
                } else {
                    const uint32_t o = atomicMin(&m1[idx], pos);         // positions of one row are distinct
                    atomicMin(&m2[idx], o > pos ? o : pos);              // larger of two hits >= 2nd smallest
                }
            }
        }
        __syncthreads();
        // emit, stage 1: thread <-> bitmap word; candidates of this part that reached min_match go to
        // an LDS list (row in the sub-index << kIxSlotBits | slot).  Looking their video ids up right here,
        // bit after bit, was one dependent global load per bit and 60 % of the kernel.
#pragma unroll
        for (int ww = 0; ww < kIxWpt; ++ww) {
            const uint32_t wi = threadIdx.x * kIxWpt + ww;
            const uint32_t w = cand[wi];
            const uint32_t rk = rank[wi];
            for (uint32_t rest = w, i = 0; rest; rest &= rest - 1, ++i) {
                const uint32_t idx = rk + i - lo;
                if (idx >= (uint32_t)kIxSlots || (int32_t)tcnt[idx] < min_match) continue;
                const uint32_t bit = (uint32_t)__ffs(rest) - 1u;
                elist[atomicAdd(&s_nlist, 1u)] = ((wi * 32u + bit) << kIxSlotBits) | idx;
            }
        }
        __syncthreads();
        // stage 2: one list entry per thread and round (<= 2 rounds): video id, filters, one
        // reservation per block, write
        const uint32_t n_list = s_nlist;
        uint32_t e[kIxSlots / kIxBlock];
        int32_t vid[kIxSlots / kIxBlock];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < kIxSlots / kIxBlock; ++u) {
            const uint32_t k = (uint32_t)u * kIxBlock + threadIdx.x;
            e[u] = k < n_list ? elist[k] : 0u;
            const int64_t row = row0 + (e[u] >> kIxSlotBits);
            vid[u] = ivid[k < n_list && row < n_indexed ? row : row0];  // unconditional load
            // replaced since the build (-1) / the query's own video: not a hit
            if (k >= n_list || row >= n_indexed || vid[u] < 0 || vid[u] == excl) vid[u] = -1;
            mine += vid[u] >= 0 ? 1u : 0u;
        }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0, all = 0;
        for (int x = 0; x < kIxWaves; ++x) { if (x < wave) base += s_wsum[x]; all += s_wsum[x]; }
        if (threadIdx.x == 0) {
            s_base = (HOSTOUT || all == 0) ? (uint32_t)s_emitted : (uint32_t)atomicAdd(out_n, (int32_t)all);
            s_emitted += (int32_t)all;
            s_nlist = 0;
        }
        __syncthreads();
        uint32_t o = s_base + base + incl - mine;
        const int64_t room = HOSTOUT ? (int64_t)kSubRows : (int64_t)cap;
#pragma unroll
        for (int u = 0; u < kIxSlots / kIxBlock; ++u) {
            if (vid[u] < 0) continue;
            const uint32_t idx = e[u] & (uint32_t)(kIxSlots - 1);
            if ((int64_t)o < room) {
                int32_t *hp = out_hits + (int64_t)o * 3;
                hp[0] = vid[u];
                hp[1] = (int32_t)tcnt[idx];
                hp[2] = TOP5 ? (int32_t)((uint32_t)(ttop[idx] >> (12 * (min_match - 1))) & 0xfffu)
                             : (int32_t)(min_match == 1 ? m1[idx] : m2[idx]);
            }
            ++o;
        }
        __syncthreads();
        if (lo + kIxSlots < n_cand)
            for (int i = threadIdx.x; i < kIxSlots; i += kIxBlock) {
                tcnt[i] = 0;
                if (TOP5) ttop[i] = kTopNone; else { m1[i] = 0xffffffffu; m2[i] = 0xffffffffu; }
            }
        __syncthreads();
    }
    if (HOSTOUT && threadIdx.x == 0) *out_n = s_emitted;
}

}  // namespace
