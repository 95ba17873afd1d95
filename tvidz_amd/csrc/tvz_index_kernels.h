// tvz_index_kernels.h — inverted index over the device corpus (gfx950, wave64).
//
// find_duplicates (/root/reference inspector/db.py:76-94) asks, for every corpus row, how many
// query elements are `in` the row.  The sweep kernels answer by reading every row; the index
// answers from the other side: for every query element, which rows contain it.
//
//   dir  : open-addressing directory of the DISTINCT canonical keys of the indexed rows,
//          16 B per entry {key, first posting, number of postings}, load <= 0.5
//   post : posting lists, one uint32 row index per (row, key) pair, contiguous per key
//   ivid : video_id per indexed row; -1 once the row was replaced by an upsert (its postings are
//          then stale and ignored; the row's current content lives in the delta table, which the
//          sweep kernels read)
//
// A query of n elements with p postings in total costs n directory probes + 2 p posting reads,
// whatever the corpus size: ~9,000 postings (36 KB) per query on the config-4 corpus against
// 160 MB for a sweep.  Counting per (query, row) pair happens in LDS, one block per query:
//   pass A  every posting sets its row's bit in `seen1`, or in `seen2` if seen1 was set already:
//           only rows in seen2 (seen1 for min_match 1) can reach min_match;
//   pass B  the postings are walked again and the candidates' (count, five smallest query
//           positions) are accumulated in an LDS hash table keyed by row index - in P parts (by a
//           hash of the row) when there are more candidates than the table holds;
//   emit    rows with count >= min_match: (video_id, count, kth) exactly as the sweeps emit them.
// Rows beyond the bitmap size share bits (row mod bits): a shared bit only adds candidates, counts
// come from the table, which is keyed by the row itself.
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
    uint32_t distinct;     // directory entries in use
    uint32_t failed;       // a probe sequence ran too long: directory too small, rebuild larger
    uint32_t pad;
};

constexpr int kIxMaxProbe = 4096;

__device__ __forceinline__ uint32_t ix_slot(int64_t k, int dir_log2) {
    return (q1_mix(k) * 0x9E3779B1u) >> (32 - dir_log2);
}

__global__ __launch_bounds__(kBlock) void ix_clear_kernel(DirEnt *__restrict__ dir, size_t n, IxBuildInfo *info) {
    const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x, step = (size_t)gridDim.x * kBlock;
    int4 e;
    e.x = (int32_t)(uint32_t)(uint64_t)kEmpty;
    e.y = (int32_t)(uint32_t)((uint64_t)kEmpty >> 32);
    e.z = 0;
    e.w = 0;
    for (size_t i = i0; i < n; i += step) reinterpret_cast<int4 *>(dir)[i] = e;
    if (i0 == 0) { info->cursor = 0; info->distinct = 0; info->failed = 0; info->pad = 0; }
}

// find (or, with INSERT, claim) the directory entry of key k; returns the slot or -1
template <bool INSERT>
__device__ __forceinline__ int64_t ix_find(DirEnt *dir, int dir_log2, int64_t k, bool &is_new) {
    const uint32_t mask = (1u << dir_log2) - 1u;
    uint32_t s = ix_slot(k, dir_log2);
    is_new = false;
    for (int probes = 0; probes < kIxMaxProbe; ++probes) {
        // look first: keys only ever go from free to taken, a stale view is corrected by the CAS
        int64_t cur = *reinterpret_cast<volatile int64_t *>(&dir[s].key);
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
                                                          IxBuildInfo *info) {
    __shared__ uint32_t s_new;
    if (threadIdx.x == 0) s_new = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t mine = 0;
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < n_rows;
         r += (int64_t)gridDim.x * (kBlock / 64)) {
        const Row row = load_row(rows + r);
        if (lane == 0) ivid[r] = row.vid;
        for (int i = lane; i < row.len; i += 64) {
            bool is_new;
            const int64_t s = ix_find<true>(dir, dir_log2, keys[row.off + i], is_new);
            if (s < 0) { info->failed = 1; continue; }
            atomicAdd(&dir[s].len, 1u);
            mine += is_new ? 1u : 0u;
        }
    }
    if (mine) atomicAdd(&s_new, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_new) atomicAdd(&info->distinct, s_new);
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
                                                         int dir_log2, uint32_t *__restrict__ post) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < n_rows;
         r += (int64_t)gridDim.x * (kBlock / 64)) {
        const Row row = load_row(rows + r);
        for (int i = lane; i < row.len; i += 64) {
            bool is_new;
            const int64_t s = ix_find<false>(dir, dir_log2, keys[row.off + i], is_new);
            if (s < 0) continue;                       // cannot happen after a successful count pass
            const uint32_t p = atomicSub(&dir[s].off, 1u) - 1u;
            post[p] = (uint32_t)r;
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

// ---- lookup: one block per query ---------------------------------------------------------------
constexpr int kIxBlock = 1024;
constexpr int kIxBitsLog2 = 17;                      // 2 x 16 KiB of bitmaps
constexpr int kIxTableLog2 = 11;                     // 2048 candidates x 16 B = 32 KiB
constexpr int kIxTable = 1 << kIxTableLog2;
constexpr int kIxTableFill = kIxTable * 3 / 4;
constexpr int kIxMaxTries = 128;                    // longer probe runs mean a crowded table

// dynamic LDS: [bm1][bm2][tkey][tcnt][ttop][s_off[L]][s_pre[L + 1]]
inline size_t ix_lds_bytes(int max_len) {
    return (size_t)2 * ((size_t)1 << kIxBitsLog2) / 8 + (size_t)kIxTable * 16 + (size_t)(2 * max_len + 2) * 4;
}

__global__ __launch_bounds__(kIxBlock) void ts_match_index_kernel(
    const DirEnt *__restrict__ dir, int dir_log2, const uint32_t *__restrict__ post,
    const int32_t *__restrict__ ivid, int64_t n_indexed, const double *__restrict__ queries,
    const int64_t *__restrict__ q_offsets, int32_t max_len, int32_t min_match,
    const int32_t *__restrict__ exclude_ids, int32_t exclude_one, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t ns, const QByVal qv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kWords = (1 << kIxBitsLog2) / 32;
    uint32_t *bm1 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *bm2 = bm1 + kWords;
    uint32_t *tkey = bm2 + kWords;                                      // row + 1, 0 = free
    uint32_t *tcnt = tkey + kIxTable;
    unsigned long long *ttop = reinterpret_cast<unsigned long long *>(tcnt + kIxTable);
    uint32_t *s_off = reinterpret_cast<uint32_t *>(ttop + kIxTable);
    uint32_t *s_pre = s_off + max_len;                                  // [n + 1] exclusive prefix of list lengths
    __shared__ uint32_t s_cand, s_ovf, s_nout, s_big;
    __shared__ uint32_t s_wsum[kIxBlock / 64];

    const int q = blockIdx.x;
    const bool byval = q_offsets == nullptr;
    const int64_t qo = byval ? 0 : q_offsets[q];
    const int64_t n64 = byval ? qv.n : q_offsets[q + 1] - qo;
    int32_t *out_n = hits_n + (size_t)q * ns;
    if (n64 > max_len) {       // max_query_len was not an upper bound (the LDS arrays are sized from it)
        if (threadIdx.x == 0) *out_n = INT32_MIN;
        return;
    }
    const int n = (int)n64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_cand = 0; s_ovf = 0; s_nout = 0; s_big = 0; }
    for (int i = threadIdx.x; i < kWords; i += kIxBlock) { bm1[i] = 0; bm2[i] = 0; }
    for (int i = threadIdx.x; i < kIxTable; i += kIxBlock) { tkey[i] = 0; tcnt[i] = 0; ttop[i] = kTopNone; }

    // ---- directory: posting list of every query position (NaN / absent key: empty list) ----
    const uint32_t dmask = (1u << dir_log2) - 1u;
    for (int i0 = 0; i0 < n; i0 += kIxBlock) {
        const int i = i0 + threadIdx.x;
        uint32_t off = 0, len = 0;
        int64_t k;
        if (i < n && canon_key(byval ? qv.k[i] : queries[qo + i], k)) {
            uint32_t s = ix_slot(k, dir_log2);
            for (int probes = 0; probes < kIxMaxProbe; ++probes) {
                const int4 e = *reinterpret_cast<const int4 *>(dir + s);
                const int64_t ek = (int64_t)(((uint64_t)(uint32_t)e.y << 32) | (uint32_t)e.x);
                if (ek == k) { off = (uint32_t)e.z; len = (uint32_t)e.w; break; }
                if (ek == kEmpty) break;
                s = (s + 1) & dmask;
            }
        }
        // exclusive prefix of len over the block (wave scan + wave sums), carried across chunks
        uint32_t incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        __syncthreads();                               // s_wsum / s_pre[i0] of the previous chunk are consumed
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t wbase = (i0 == 0) ? 0u : s_pre[i0];
        unsigned long long wide = wbase;
        for (int w = 0; w < wave; ++w) { wbase += s_wsum[w]; wide += s_wsum[w]; }
        if (wide + incl > 0xffffffffULL) s_big = 1;    // > 4 G postings for one query: not this kernel's job
        if (i < n) { s_off[i] = off; s_pre[i] = wbase + incl - len; }
        const int last = (i0 + kIxBlock < n ? i0 + kIxBlock : n) - 1;
        __syncthreads();
        if (i == last) s_pre[i + 1] = wbase + incl;
        __syncthreads();
    }
    if (n == 0 && threadIdx.x == 0) s_pre[0] = 0;
    __syncthreads();
    if (s_big) {
        if (threadIdx.x == 0) *out_n = INT32_MIN;
        return;
    }
    const uint32_t total = s_pre[n];
    constexpr uint32_t bmask = (1u << kIxBitsLog2) - 1u;
    // posting t of the flattened lists -> (query position, row)
    auto locate = [&](uint32_t t, int &pos) -> uint32_t {
        int lo = 0, hi = n;                            // largest pos with s_pre[pos] <= t
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid] <= t) lo = mid; else hi = mid;
        }
        pos = lo;
        return post[s_off[lo] + (t - s_pre[lo])];
    };

    // ---- pass A: which rows are touched twice ----
    const uint32_t *cand_bm = bm1;
    if (min_match >= 2) {
        cand_bm = bm2;
        for (uint32_t t0 = threadIdx.x; t0 < total; t0 += 4u * kIxBlock) {
            uint32_t r[4];
            int pos;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t t = t0 + (uint32_t)u * kIxBlock;
                r[u] = t < total ? locate(t, pos) : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r[u] == 0xffffffffu) continue;
                const uint32_t b = r[u] & bmask, bit = 1u << (b & 31u);
                const uint32_t old = atomicOr(&bm1[b >> 5], bit);
                if (old & bit) atomicOr(&bm2[b >> 5], bit);
            }
        }
    } else {
        for (uint32_t t = threadIdx.x; t < total; t += kIxBlock) {
            int pos;
            const uint32_t r = locate(t, pos);
            const uint32_t b = r & bmask;
            atomicOr(&bm1[b >> 5], 1u << (b & 31u));
        }
    }
    __syncthreads();
    {
        uint32_t c = 0;
        for (int i = threadIdx.x; i < kWords; i += kIxBlock) c += __popc(cand_bm[i]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
        if (lane == 0 && c) atomicAdd(&s_cand, c);
    }
    __syncthreads();
    // parts: enough that a part's candidates fit the table (exact when rows <= bitmap bits; rows
    // sharing a bit can exceed the estimate - the overflow path below doubles P and starts over)
    int p_log2 = 0;
    while (((uint32_t)kIxTableFill << p_log2) < s_cand && p_log2 < 20) ++p_log2;
    const int32_t excl = exclude_ids ? exclude_ids[q] : exclude_one;

    // ---- pass B + emit, part by part ----
    while (true) {
        bool restart = false;
        for (uint32_t part = 0; part < (1u << p_log2); ++part) {
            for (uint32_t t0 = threadIdx.x; t0 < total; t0 += 4u * kIxBlock) {
                uint32_t r[4];
                int pos[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t t = t0 + (uint32_t)u * kIxBlock;
                    pos[u] = 0;
                    r[u] = t < total ? locate(t, pos[u]) : 0xffffffffu;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (r[u] == 0xffffffffu) continue;
                    const uint32_t b = r[u] & bmask;
                    if (!((cand_bm[b >> 5] >> (b & 31u)) & 1u)) continue;
                    if (p_log2 && ((r[u] * 0x85EBCA6Bu) >> (32 - p_log2)) != part) continue;
                    uint32_t s = (r[u] * 0x9E3779B1u) >> (32 - kIxTableLog2);
                    int tries = 0;
                    for (; tries < kIxMaxTries; ++tries) {
                        const uint32_t old = atomicCAS(&tkey[s], 0u, r[u] + 1u);
                        if (old == 0u || old == r[u] + 1u) break;
                        s = (s + 1) & (uint32_t)(kIxTable - 1);
                    }
                    if (tries == kIxMaxTries) { s_ovf = 1; continue; }   // crowded part: split further
                    atomicAdd(&tcnt[s], 1u);
                    unsigned long long seen = ttop[s];
                    while (true) {
                        if (((uint32_t)(seen >> (12 * (kTop - 1))) & 0xfffu) <= (uint32_t)pos[u]) break;
                        const unsigned long long old = atomicCAS(&ttop[s], seen, top5_insert(seen, (uint32_t)pos[u]));
                        if (old == seen) break;
                        seen = old;
                    }
                }
            }
            __syncthreads();
            if (s_ovf) { restart = true; }
            if (!restart) {
                for (int i = threadIdx.x; i < kIxTable; i += kIxBlock) {
                    const uint32_t rk = tkey[i];
                    if (rk == 0u) continue;
                    const int32_t c = (int32_t)tcnt[i];
                    if (c >= min_match && (int64_t)(rk - 1u) < n_indexed) {
                        const int32_t vid = ivid[rk - 1u];
                        if (vid >= 0 && vid != excl) {
                            const int32_t kth = (int32_t)((uint32_t)(ttop[i] >> (12 * (min_match - 1))) & 0xfffu);
                            const uint32_t o = atomicAdd(&s_nout, 1u);
                            if ((int64_t)o < (int64_t)cap) {
                                int32_t *hp = hits + ((int64_t)q * cap + o) * 3;
                                hp[0] = vid;
                                hp[1] = c;
                                hp[2] = kth;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < kIxTable; i += kIxBlock) { tkey[i] = 0; tcnt[i] = 0; ttop[i] = kTopNone; }
            if (threadIdx.x == 0) s_ovf = 0;
            __syncthreads();
            if (restart) break;
        }
        if (!restart) break;
        // more candidates in one part than the table holds: twice the parts, from the start (what was
        // emitted so far is overwritten)
        if (threadIdx.x == 0) s_nout = 0;
        ++p_log2;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_n = (int32_t)s_nout;
}

}  // namespace
