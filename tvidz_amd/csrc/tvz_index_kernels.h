// tvz_index_kernels.h — inverted index over the device corpus (gfx950, wave64).
//
// find_duplicates (/root/reference inspector/db.py:76-94) asks, for every corpus row, how many
// query elements are `in` the row.  The sweep kernels answer by reading every row; the index
// answers from the other side: for every query element, which rows contain it.
//
// Layout (round 3: ONE directory probe per query element, whatever the corpus size):
//   dir  : ONE open-addressing directory over the distinct canonical keys of ALL indexed rows.
//          An entry is a 16-byte head {key, first posting, postings in all} followed by `ks`
//          uint16 counts - the key's postings in each SUB-INDEX of 2^14 rows (ks = sub-indexes
//          rounded up to 8; ks = 0 for a corpus of one sub-index: the head's total is the count).
//          One probe (head + counts share a cache line) tells a query element everything.
//          Probing is linear INSIDE a slice of 2^slice_log2 entries (kernel argument dir_bits).
//   post : posting lists, one uint16 LOCAL row number per (row, key) pair; the postings of a key
//          are contiguous, ordered by sub-index - the pieces a block reads one sub-index after the
//          other lie next to each other in memory - and placed so that a key crosses no 128-byte
//          line it need not cross (size classes, see ix_slice_count_kernel): a key of up to 64
//          postings is ONE line, which the block's seven passes then find in the L2
//   ivid : video_id per indexed row; -1 once the row was replaced by an upsert (its postings are
//          then stale and ignored; the row's current content lives in the delta table, which the
//          sweep kernels read).
//
// Lookup: one block per QUERY walks the query's sub-indexes one after the other (a lone query gets
// one block per sub-index instead: latency).  The query's keys are probed once; per sub-index
//   pass A  every posting sets its row's bit in `seen1`, or in `seen2` if seen1 was set already:
//           only rows in seen2 (seen1 for min_match 1) can reach min_match; the bitmaps cover the
//           sub-index exactly (16,384 bits), so the candidates are known row by row;
//   rank    prefix popcount of the candidate bitmap: candidate -> dense slot, no hashing;
//   pass B  the postings are walked again - from the (row, position) pairs pass A left in LDS (the
//           first kIxCache; the lists themselves beyond that); a candidate's count and smallest query
//           positions (two atomicMin words for min_match <= 2, five 12-bit positions in a CAS word
//           for 3..5) accumulate in its slot - 1,024 slots at a time if there are more candidates;
//   emit    candidates with count >= min_match, (video_id, count, kth) exactly as the sweeps emit
//           them.  A block that owns its query alone appends to the hit list with NO global atomic
//           and stores the count once at the end.
// The walks are laid out so that every wave owns a contiguous range of the flattened postings:
// consecutive lanes read consecutive postings and a lane's list pointer only moves forward.
//
// Build (tvz_match.hip: into a SHADOW buffer set, readers keep using the old index meanwhile): the
// directory is cut into SLICES of ~32 KB; an entry lives in the slice of its home slot (linear
// probing wraps inside the slice).  The (key, row) pairs of all rows are partitioned by slice
// (histogram, scan, LDS-staged scatter: whole lines leave the chip), then ONE block per slice makes
// the slice's entries, counts and posting places in LDS and writes them out once.  No sort, no
// global scatter: 1.0 ms for 100k rows / 19.9 M keys (count + fill over the whole directory, kept
// for directories of more than 4,096 slices: 2.7 ms).
#pragma once
#include "tvz_match_kernels.h"

namespace {

struct alignas(16) DirHead {
    int64_t key;       // kEmpty = free
    uint32_t base;     // first posting of the key
    uint32_t total;    // postings of the key over all sub-indexes
};
static_assert(sizeof(DirHead) == 16, "DirHead must be 16 bytes");

struct IxBuildInfo {       // device-side build status, read back by the host
    uint32_t cursor;       // postings handed out
    uint32_t n_distinct;   // directory entries in use
    uint32_t failed;       // a probe sequence ran too long: directory too small, rebuild larger
    uint32_t pad;
    // bucket directory (tvz_bucket_dir.h)
    uint32_t ext_cursor;   // postings handed out in the external area (uint16 units, whole lines)
    uint32_t n_spilled;    // keys that do not live in their home bucket
    uint32_t n_ext;        // keys whose postings are external
    uint32_t max_spill;
};

constexpr int kIxMaxProbe = 4096;
// entries behind the last posting that the lookups may READ (and discard): the lanes of a wave's last step of 64
// postings (ix_lookup_body), the dead steps of a live group of four (ts_match_wq_topk_kernel)
constexpr int kIxPostPad = 64 + 512;   // (dead steps of a live group of up to eight)
#ifndef TVZ_IX_SUB_LOG2
#define TVZ_IX_SUB_LOG2 14
#endif
constexpr int kSubLog2 = TVZ_IX_SUB_LOG2;            // rows per sub-index (local row numbers are uint16)
constexpr int kSubRows = 1 << kSubLog2;

}  // namespace
#include "tvz_bucket_dir.h"
namespace {

inline int ix_ks(int n_sub) { return n_sub <= 1 ? 0 : (n_sub + 7) & ~7; }   // uint16 counts per entry
inline int ix_entry_bytes(int ks) { return 16 + 2 * ks; }

// kernel argument `dir_bits` = log2 of the directory entries | log2 of the entries per SLICE << 8: an
// entry lives in the slice of its home slot (linear probing wraps inside the slice), so that a build
// can make a slice in LDS from the keys that hash into it and nothing else
inline int ix_dir_bits(int dir_log2, int slice_log2) { return dir_log2 | (slice_log2 << 8); }

__device__ __forceinline__ uint32_t ix_slot(int64_t k, int dir_log2) {
    return (q1_mix(k) * 0x9E3779B1u) >> (32 - dir_log2);
}

// every entry = {free, 0, 0, counts 0}; the fill cursors = 0
__global__ __launch_bounds__(kBlock) void ix_clear_kernel(uint4 *__restrict__ dir16, size_t n16, int es16,
                                                          uint4 *__restrict__ zero16, size_t nz16,
                                                          IxBuildInfo *info) {
    const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x, step = (size_t)gridDim.x * kBlock;
    uint4 head;
    head.x = (uint32_t)(uint64_t)kEmpty;
    head.y = (uint32_t)((uint64_t)kEmpty >> 32);
    head.z = 0;
    head.w = 0;
    for (size_t i = i0; i < n16; i += step) dir16[i] = (i % (size_t)es16 == 0) ? head : make_uint4(0, 0, 0, 0);
    for (size_t i = i0; i < nz16; i += step) zero16[i] = make_uint4(0, 0, 0, 0);
    if (i0 == 0) { info->cursor = 0; info->n_distinct = 0; info->failed = 0; info->pad = 0;
                   info->ext_cursor = 0; info->n_spilled = 0; info->n_ext = 0; info->max_spill = 0; }
}

// the directory slice of a key: the partition kernels' bin.  dir_bits < 0: the BUCKET directory of a one-sub-index
// handle (tvz_bucket_dir.h) with -dir_bits buckets, slices of kBkSlice buckets
__device__ __forceinline__ uint32_t ix_part_of(int64_t k, int dir_bits) {
    if (dir_bits < 0) return bk_bucket(k, (uint32_t)(-dir_bits)) >> kBkSliceLog2;
    return ix_slot(k, dir_bits & 0xff) >> (dir_bits >> 8);
}

// find (or, with INSERT, claim) the directory entry of key k; returns the slot or -1
template <bool INSERT>
__device__ __forceinline__ int64_t ix_find(unsigned char *dir, int es, int dir_bits, int64_t k, bool &is_new) {
    const int dir_log2 = dir_bits & 0xff;
    const uint32_t smask = (1u << (dir_bits >> 8)) - 1u;      // probes wrap inside the key's directory slice
    uint32_t s = ix_slot(k, dir_log2);
    is_new = false;
    for (int probes = 0; probes < kIxMaxProbe; ++probes) {
        int64_t *kp = reinterpret_cast<int64_t *>(dir + (size_t)s * es);
        // look first: keys only ever go from free to taken, a stale view is corrected by the CAS
        const int64_t cur = INSERT ? *reinterpret_cast<volatile int64_t *>(kp) : *kp;
        if (cur == k) return s;
        if (cur == kEmpty) {
            if (!INSERT) return -1;
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(kp),
                                                     (unsigned long long)kEmpty, (unsigned long long)k);
            if (old == (unsigned long long)kEmpty) { is_new = true; return s; }
            if ((int64_t)old == k) return s;
        }
        s = (s & ~smask) | ((s + 1) & smask);
    }
    return -1;
}

// one wave per row: count the postings of every (key, sub-index), remember the row's video id.
// The uint16 counts are bumped through their 32-bit word (a count never exceeds the 2^14 rows of a
// sub-index, so the low half cannot carry into the high half).
__global__ __launch_bounds__(kBlock) void ix_count_kernel(const Row *__restrict__ rows, int64_t n_rows,
                                                          const int64_t *__restrict__ keys, unsigned char *dir,
                                                          int es, int ks, int dir_bits, int32_t *__restrict__ ivid,
                                                          IxBuildInfo *info) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < n_rows;
         r += (int64_t)gridDim.x * (kBlock / 64)) {
        const Row row = load_row(rows + r);
        if (lane == 0) ivid[r] = row.vid;
        const uint32_t sub = (uint32_t)(r >> kSubLog2);
        uint32_t mine = 0;
        for (int i = lane; i < row.len; i += 64) {
            bool is_new;
            const int64_t s = ix_find<true>(dir, es, dir_bits, keys[row.off + i], is_new);
            if (s < 0) { info->failed = 1; continue; }
            unsigned char *e = dir + (size_t)s * es;
            if (ks) atomicAdd(reinterpret_cast<uint32_t *>(e + 16) + (sub >> 1), 1u << ((sub & 1u) * 16u));
            else atomicAdd(&reinterpret_cast<DirHead *>(e)->total, 1u);
            mine += is_new ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if (lane == 0 && mine) atomicAdd(&info->n_distinct, mine);
    }
}

// hand out posting ranges: a thread per directory entry, block-wide exclusive scan of the entries'
// totals, ONE atomic per block on the cursor (a wave per atomic was 229,376 waves queueing on one
// address: 2.6 ms of a 5.2 ms build in round 2)
__global__ __launch_bounds__(kBlock) void ix_offsets_kernel(unsigned char *dir, size_t n, int es, int ks,
                                                            IxBuildInfo *info) {
    __shared__ uint32_t s_w[kBlock / 64];
    __shared__ uint32_t s_base;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t total = 0;
    if (i < n) {
        DirHead *h = reinterpret_cast<DirHead *>(dir + i * es);
        if (ks) {
            const uint4 *c = reinterpret_cast<const uint4 *>(dir + i * es + 16);
            for (int j = 0; j < ks / 8; ++j) {
                const uint4 v = c[j];
                total += (v.x & 0xffffu) + (v.x >> 16) + (v.y & 0xffffu) + (v.y >> 16) + (v.z & 0xffffu) +
                         (v.z >> 16) + (v.w & 0xffffu) + (v.w >> 16);
            }
            h->total = total;
        } else {
            total = h->total;
        }
    }
    uint32_t incl = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { if (w < wave) before += s_w[w]; all += s_w[w]; }
    if (threadIdx.x == 0) s_base = all ? atomicAdd(&info->cursor, all) : 0u;
    __syncthreads();
    if (i < n && total) reinterpret_cast<DirHead *>(dir + i * es)->base = s_base + before + incl - total;
}

// one wave per row: every key takes the next free posting of its (key, sub-index) range.
// fillc: one 32-bit word per two sub-indexes of an entry (ks = 0: one word per entry), zeroed.
__global__ __launch_bounds__(kBlock) void ix_fill_kernel(const Row *__restrict__ rows, int64_t n_rows,
                                                         const int64_t *__restrict__ keys, unsigned char *dir, int es,
                                                         int ks, int dir_bits, uint32_t *__restrict__ fillc,
                                                         uint16_t *__restrict__ post) {
    const int lane = threadIdx.x & 63;
    const int fw = ks ? ks / 2 : 1;                    // fill-cursor words per entry
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < n_rows;
         r += (int64_t)gridDim.x * (kBlock / 64)) {
        const Row row = load_row(rows + r);
        const uint32_t sub = (uint32_t)(r >> kSubLog2);
        for (int i = lane; i < row.len; i += 64) {
            bool is_new;
            const int64_t s = ix_find<false>(dir, es, dir_bits, keys[row.off + i], is_new);
            if (s < 0) continue;                       // cannot happen after a successful count pass
            const unsigned char *e = dir + (size_t)s * es;
            uint32_t p = reinterpret_cast<const DirHead *>(e)->base;
            uint32_t k;
            if (ks) {
                const uint16_t *cn = reinterpret_cast<const uint16_t *>(e + 16);
                for (uint32_t t = 0; t < sub; ++t) p += cn[t];
                k = (atomicAdd(&fillc[(size_t)s * fw + (sub >> 1)], 1u << ((sub & 1u) * 16u)) >> ((sub & 1u) * 16u)) & 0xffffu;
            } else {
                k = atomicAdd(&fillc[s], 1u);
            }
            post[p + k] = (uint16_t)(r & (kSubRows - 1));
        }
    }
}

// ---- partitioned build (round 3): no global scatter --------------------------------------------
// count + fill above touch the whole directory and the whole posting array at random from every
// block: 20 M probes, 20 M memory-side atomics and 20 M two-byte writes = 9.6 GB of line traffic for
// a 160 MB corpus.  Here the (key, row) pairs are first PARTITIONED by directory slice (two
// streaming passes: histogram, then scatter with one reservation per block and slice), and each
// slice - its entries, counts and posting ranges - is then built by ONE block in LDS and written out
// once: the directory and the postings leave the chip as whole lines.
constexpr int kIxMaxParts = 4096;          // slices (LDS histogram + cursors of the partition kernels: 8 B each)
constexpr int kIxSliceBlock = 512;

__global__ void ix_part_clear_kernel(uint32_t *__restrict__ hist, int n, IxBuildInfo *info) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) hist[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { info->cursor = 0; info->n_distinct = 0; info->failed = 0; info->pad = 0;
                   info->ext_cursor = 0; info->n_spilled = 0; info->n_ext = 0; info->max_spill = 0; }
}

// the slices' pair counts and the rows' video ids: a block takes `rpb` consecutive rows (a wave per row,
// in turns), counts in an LDS histogram and adds its non-zero counts to the global ones
__global__ __launch_bounds__(kBlock) void ix_partition_kernel(
    const Row *__restrict__ rows, int64_t n_rows, int32_t rpb, const int64_t *__restrict__ keys, int dir_bits,
    int n_parts, uint32_t *__restrict__ gcnt, int32_t *__restrict__ ivid) {
    extern __shared__ uint32_t ix_part_sh[];
    uint32_t *hist = ix_part_sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < n_parts; i += kBlock) hist[i] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = r0 + rpb < n_rows ? r0 + rpb : n_rows;
    for (int64_t r = r0 + wave; r < r1; r += kBlock / 64) {
        const Row row = load_row(rows + r);
        if (lane == 0) ivid[r] = row.vid;
        for (int i = lane; i < row.len; i += 64)
            atomicAdd(&hist[ix_part_of(keys[row.off + i], dir_bits)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_parts; i += kBlock)
        if (hist[i]) atomicAdd(&gcnt[i], hist[i]);
}

// The scatter, staged: a block of 1024 threads takes rows worth ~kIxStagePairs pairs, counts them per
// slice, gives every slice its LOCAL range in an LDS staging area (block scan over the slices) and
// its GLOBAL range (one atomic per non-empty slice), drops the pairs into the staging area in slice
// order, and then writes the staging area out with consecutive threads on consecutive pairs - the
// pairs of one slice go out as one run of whole lines.  (Writing every pair to its slice directly
// left ~2,000 partly written lines per block in flight - 67 MB over the chip against 32 MB of L2 -
// and took 1.15 ms of a 1.6 ms build.)  Pairs beyond the staging area
// (rows longer than expected) are written directly.
constexpr int kIxScatterBlock = 1024;
constexpr int kIxStagePairs = 8192;        // 96 KB of LDS: 8 B key + 4 B row

__global__ __launch_bounds__(kIxScatterBlock) void ix_scatter_kernel(
    const Row *__restrict__ rows, int64_t n_rows, int32_t rpb, const int64_t *__restrict__ keys, int dir_bits,
    int n_parts, uint32_t *__restrict__ gcur, int64_t *__restrict__ pkeys, uint32_t *__restrict__ prows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ix_scat_sh[];
    int64_t *st_k = reinterpret_cast<int64_t *>(ix_scat_sh);                       // [kIxStagePairs]
    uint32_t *st_r = reinterpret_cast<uint32_t *>(st_k + kIxStagePairs);           // [kIxStagePairs]
    uint32_t *hist = st_r + kIxStagePairs;                                         // [n_parts] counts, then fill cursors
    uint32_t *lstart = hist + n_parts;                                             // [n_parts + 1] first staged pair of a slice
    uint32_t *base = lstart + n_parts + 1;                                         // [n_parts] first global pair of the block's run
    __shared__ uint32_t s_w[kIxScatterBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < n_parts; i += kIxScatterBlock) hist[i] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = r0 + rpb < n_rows ? r0 + rpb : n_rows;
    for (int64_t r = r0 + wave; r < r1; r += kIxScatterBlock / 64) {
        const Row row = load_row(rows + r);
        for (int i = lane; i < row.len; i += 64)
            atomicAdd(&hist[ix_part_of(keys[row.off + i], dir_bits)], 1u);
    }
    __syncthreads();
    // local and global ranges: thread t owns the slices [t * per, (t + 1) * per)
    const int per = (n_parts + kIxScatterBlock - 1) / kIxScatterBlock;
    const int p0 = threadIdx.x * per;
    uint32_t mine = 0;
    for (int i = p0; i < p0 + per && i < n_parts; ++i) mine += hist[i];
    const uint32_t incl = wave_scan_incl(mine);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kIxScatterBlock / 64; ++w) { if (w < wave) before += s_w[w]; all += s_w[w]; }
    {
        uint32_t run = before + incl - mine;
        for (int i = p0; i < p0 + per && i < n_parts; ++i) {
            const uint32_t c = hist[i];
            lstart[i] = run;
            base[i] = c ? atomicAdd(&gcur[i], c) : 0u;
            hist[i] = 0;
            run += c;
        }
        if (threadIdx.x == 0) lstart[n_parts] = all;
    }
    __syncthreads();
    for (int64_t r = r0 + wave; r < r1; r += kIxScatterBlock / 64) {
        const Row row = load_row(rows + r);
        for (int i = lane; i < row.len; i += 64) {
            const int64_t k = keys[row.off + i];
            const uint32_t p = ix_part_of(k, dir_bits);
            const uint32_t j = atomicAdd(&hist[p], 1u);
            const uint32_t t = lstart[p] + j;
            if (t < (uint32_t)kIxStagePairs) {
                st_k[t] = k;
                st_r[t] = (uint32_t)r;
            } else {
                pkeys[base[p] + j] = k;
                prows[base[p] + j] = (uint32_t)r;
            }
        }
    }
    __syncthreads();
    const uint32_t staged = all < (uint32_t)kIxStagePairs ? all : (uint32_t)kIxStagePairs;
    for (uint32_t t = threadIdx.x; t < staged; t += kIxScatterBlock) {
        int lo = 0, hi = n_parts;                       // the slice of staged pair t: lstart[lo] <= t < lstart[lo + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (lstart[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t g = base[lo] + (t - lstart[lo]);
        pkeys[g] = st_k[t];
        prows[g] = st_r[t];
    }
}

// exclusive scan of the slices' pair counts (one block; n <= kIxMaxParts): start[i], start[n] = all
// pairs; cur[i] = start[i] (the scatter's cursors); info->cursor = all pairs
__global__ __launch_bounds__(1024) void ix_part_scan_kernel(const uint32_t *__restrict__ cnt, int n,
                                                            uint32_t *__restrict__ start, uint32_t *__restrict__ cur,
                                                            IxBuildInfo *info) {
    __shared__ uint32_t s_w[16];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per;
    uint32_t mine = 0;
    for (int i = lo; i < lo + per && i < n; ++i) mine += cnt[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl(mine);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wave) before += s_w[w]; all += s_w[w]; }
    uint32_t run = before + incl - mine;
    for (int i = lo; i < lo + per && i < n; ++i) {
        const uint32_t c = cnt[i];
        start[i] = run;
        cur[i] = run;
        run += c;
    }
    if (threadIdx.x == 0) { start[n] = all; if (info) info->cursor = all; }
}

// ---- one block per slice -------------------------------------------------------------------------
// ix_slice_count_kernel: the slice's directory entries are made in LDS from the slice's pairs (find-
// or-insert by 64-bit LDS CAS, the uint16 count of the (key, sub-index) bumped through its 32-bit
// word) and every key's postings get a place in the slice's posting range such that NO KEY CROSSES A
// 128-BYTE LINE IT NEED NOT CROSS: keys are placed by size class - 1, 2, 4 .. 32 postings (a
// power-of-two slot each, the class regions start on line boundaries) and "more than 32" (a whole
// number of lines each).  A lookup touches a key's postings once per sub-index; with the keys packed
// back to back a key of 45 postings (90 B) lay in 1.7 lines on average, and the lines a block comes
// back to did not fit the XCD's L2.  The padding is bounded by 2x.  The slice goes to the directory
// with bases RELATIVE to the slice's posting range; its padded size goes to ptot[slice].
// ix_slice_fill_kernel (after a scan of ptot): the slice comes back into LDS, the bases become
// absolute, and every pair takes a place in its (key, sub-index) piece by counting the LDS count
// DOWN (the copy in the directory keeps the counts; a piece starts at base + the counts of the lower
// sub-indexes, read from that copy).
constexpr int kIxClasses = 7;              // posting-count classes: 2^0 .. 2^5, and whole lines
constexpr uint32_t kIxLine = 64;           // postings per 128-byte line

__device__ __forceinline__ int ix_find_lds(unsigned char *sl, int es, uint32_t smask, uint32_t home, int64_t k,
                                           bool insert) {
    uint32_t s = home & smask;
    for (uint32_t probes = 0; probes <= smask; ++probes) {
        unsigned long long *kp = reinterpret_cast<unsigned long long *>(sl + (size_t)s * es);
        const int64_t cur = (int64_t)*reinterpret_cast<volatile unsigned long long *>(kp);
        if (cur == k) return (int)s;
        if (cur == kEmpty) {
            if (!insert) return -1;
            const unsigned long long old = atomicCAS(kp, (unsigned long long)kEmpty, (unsigned long long)k);
            if (old == (unsigned long long)kEmpty || (int64_t)old == k) return (int)s;
        }
        s = (s + 1) & smask;
    }
    return -1;                             // the slice is full
}

__device__ __forceinline__ void ix_class_of(uint32_t total, int &cls, uint32_t &size) {
    if (total > 32u) { cls = kIxClasses - 1; size = (total + kIxLine - 1u) & ~(kIxLine - 1u); }
    else { cls = total <= 1u ? 0 : 32 - __clz((int)(total - 1u)); size = 1u << cls; }
}

__global__ __launch_bounds__(kIxSliceBlock) void ix_slice_count_kernel(
    const int64_t *__restrict__ pkeys, const uint32_t *__restrict__ prows, const uint32_t *__restrict__ start,
    unsigned char *dir, int es, int ks, int dir_bits, uint32_t *__restrict__ ptot, IxBuildInfo *info) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ix_slice_sh[];
    __shared__ uint32_t s_w[kIxClasses][kIxSliceBlock / 64], s_d[kIxSliceBlock / 64];
    unsigned char *sl = ix_slice_sh;
    const int dir_log2 = dir_bits & 0xff, slice_log2 = dir_bits >> 8;
    const uint32_t se = 1u << slice_log2, smask = se - 1u;
    const uint32_t part = blockIdx.x;
    const uint32_t lo = start[part], hi = start[part + 1];
    const int e16 = es / 16;
    {
        uint4 head;
        head.x = (uint32_t)(uint64_t)kEmpty;
        head.y = (uint32_t)((uint64_t)kEmpty >> 32);
        head.z = 0;
        head.w = 0;
        uint4 *s16 = reinterpret_cast<uint4 *>(sl);
        for (uint32_t i = threadIdx.x; i < se * (uint32_t)e16; i += kIxSliceBlock)
            s16[i] = (i % (uint32_t)e16 == 0) ? head : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    for (uint32_t j = lo + threadIdx.x; j < hi; j += kIxSliceBlock) {
        const int64_t k = pkeys[j];
        const int s = ix_find_lds(sl, es, smask, ix_slot(k, dir_log2), k, true);
        if (s < 0) { info->failed = 1; continue; }
        const uint32_t sub = prows[j] >> kSubLog2;
        unsigned char *e = sl + (size_t)s * es;
        if (ks) atomicAdd(reinterpret_cast<uint32_t *>(e + 16) + (sub >> 1), 1u << ((sub & 1u) * 16u));
        else atomicAdd(&reinterpret_cast<DirHead *>(e)->total, 1u);
    }
    __syncthreads();
    // totals, then places by size class: thread t owns the entries [t * per, (t + 1) * per)
    const uint32_t per = se >= (uint32_t)kIxSliceBlock ? se / kIxSliceBlock : 1u;
    const uint32_t e0 = threadIdx.x * per;
    uint32_t mine[kIxClasses], used = 0;
#pragma unroll
    for (int c = 0; c < kIxClasses; ++c) mine[c] = 0;
    if (e0 < se) {
        for (uint32_t i = e0; i < e0 + per; ++i) {
            DirHead *h = reinterpret_cast<DirHead *>(sl + (size_t)i * es);
            uint32_t total = 0;
            if (ks) {
                const uint32_t *c = reinterpret_cast<const uint32_t *>(sl + (size_t)i * es + 16);
                for (int w = 0; w < ks / 2; ++w) total += (c[w] & 0xffffu) + (c[w] >> 16);
                h->total = total;
            } else {
                total = h->total;
            }
            if (total) {
                int cls;
                uint32_t size;
                ix_class_of(total, cls, size);
#pragma unroll
                for (int c = 0; c < kIxClasses; ++c) mine[c] += c == cls ? size : 0u;
            }
            used += h->key != kEmpty ? 1u : 0u;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t excl[kIxClasses];
#pragma unroll
    for (int c = 0; c < kIxClasses; ++c) {
        const uint32_t incl = wave_scan_incl(mine[c]);
        if (lane == 63) s_w[c][wave] = incl;
        excl[c] = incl - mine[c];
    }
    {
        const uint32_t dsum = wave_total(wave_scan_incl(used));
        if (lane == 63) s_d[wave] = dsum;
    }
    __syncthreads();
    uint32_t region = 0;                   // first posting (relative to the slice) of the class regions, in turn
#pragma unroll
    for (int c = 0; c < kIxClasses; ++c) {
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kIxSliceBlock / 64; ++w) { const uint32_t a = s_w[c][w]; if (w < wave) before += a; all += a; }
        excl[c] += before + region;
        region += (all + kIxLine - 1u) & ~(kIxLine - 1u);
    }
    if (threadIdx.x == 0) {
        uint32_t dall = 0;
#pragma unroll
        for (int w = 0; w < kIxSliceBlock / 64; ++w) dall += s_d[w];
        if (dall) atomicAdd(&info->n_distinct, dall);
        ptot[part] = region;
    }
    if (e0 < se) {
        for (uint32_t i = e0; i < e0 + per; ++i) {
            DirHead *h = reinterpret_cast<DirHead *>(sl + (size_t)i * es);
            if (h->total) {
                int cls;
                uint32_t size;
                ix_class_of(h->total, cls, size);
#pragma unroll
                for (int c = 0; c < kIxClasses; ++c)
                    if (c == cls) { h->base = excl[c]; excl[c] += size; }
            }
        }
    }
    __syncthreads();
    const uint4 *s16 = reinterpret_cast<const uint4 *>(sl);
    uint4 *g16 = reinterpret_cast<uint4 *>(dir + (size_t)part * se * es);
    for (uint32_t i = threadIdx.x; i < se * (uint32_t)e16; i += kIxSliceBlock) g16[i] = s16[i];
}

__global__ __launch_bounds__(kIxSliceBlock) void ix_slice_fill_kernel(
    const int64_t *__restrict__ pkeys, const uint32_t *__restrict__ prows, const uint32_t *__restrict__ start,
    const uint32_t *__restrict__ pstart, unsigned char *dir, int es, int ks, int dir_bits,
    uint16_t *__restrict__ post) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ix_slice_sh[];
    unsigned char *sl = ix_slice_sh;
    const int dir_log2 = dir_bits & 0xff, slice_log2 = dir_bits >> 8;
    const uint32_t se = 1u << slice_log2, smask = se - 1u;
    const uint32_t part = blockIdx.x;
    const uint32_t lo = start[part], hi = start[part + 1];
    const uint32_t first = pstart[part];                         // the slice's first posting
    const int e16 = es / 16;
    unsigned char *gsl = dir + (size_t)part * se * es;           // this slice in the directory
    {
        uint4 *s16 = reinterpret_cast<uint4 *>(sl);
        uint4 *g16 = reinterpret_cast<uint4 *>(gsl);
        for (uint32_t i = threadIdx.x; i < se * (uint32_t)e16; i += kIxSliceBlock) {
            uint4 v = g16[i];
            if (i % (uint32_t)e16 == 0 && v.w) {                 // a head with postings: base becomes absolute
                v.z += first;
                g16[i] = v;
            }
            s16[i] = v;
        }
    }
    __syncthreads();
    for (uint32_t j = lo + threadIdx.x; j < hi; j += kIxSliceBlock) {
        const int64_t key = pkeys[j];
        const int s = ix_find_lds(sl, es, smask, ix_slot(key, dir_log2), key, false);
        if (s < 0) continue;                                     // only after a failed insert in the count kernel
        const uint32_t row = prows[j], sub = row >> kSubLog2;
        unsigned char *e = sl + (size_t)s * es;
        uint32_t p = reinterpret_cast<const DirHead *>(e)->base, k;
        if (ks) {
            const uint32_t sh = (sub & 1u) * 16u;
            k = ((atomicSub(reinterpret_cast<uint32_t *>(e + 16) + (sub >> 1), 1u << sh) >> sh) & 0xffffu) - 1u;
            const uint16_t *cn = reinterpret_cast<const uint16_t *>(gsl + (size_t)s * es + 16);
            for (uint32_t c = 0; c * 8 < sub; ++c) {             // 16-byte pieces of the counts
                const uint4 v = *reinterpret_cast<const uint4 *>(cn + c * 8);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int h = 0; h < 8; ++h)
                    if (c * 8 + h < sub) p += (w[h >> 1] >> ((h & 1) * 16)) & 0xffffu;
            }
        } else {
            k = atomicSub(&reinterpret_cast<DirHead *>(e)->total, 1u) - 1u;   // (the directory copy keeps the total)
        }
        post[p + k] = (uint16_t)(row & (uint32_t)(kSubRows - 1));
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

// after an index swap: the rows that were upserted while the new index was being built are dead in
// it (their current entries were copied to the new delta table by the host)
__global__ __launch_bounds__(kBlock) void ix_mark_dead_kernel(int32_t *__restrict__ ivid,
                                                              const int32_t *__restrict__ rows_dead, int32_t n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) ivid[rows_dead[i]] = -1;
}

// ---- lookup: one block per query (and group of sub-indexes) ------------------------------------
#ifndef TVZ_IX_SLOT_BITS
#define TVZ_IX_SLOT_BITS 9
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
constexpr int kIxSlots = 1 << kIxSlotBits;           // candidate slots per part (16 B each)
constexpr int kIxCache = TVZ_IX_CACHE;               // (row, position) of postings kept in LDS between the passes
constexpr int kIxPW = kIxCache / kIxWaves;           // ... per wave
constexpr int kIxPWSteps = kIxPW / 64;               // steps (64 postings) of a wave's cached range
constexpr int kIxLW = kIxPW / 32;                    // words of a wave's list-start bitmap
static_assert(kIxPW % 64 == 0 && kIxPWSteps >= 1 && kIxPWSteps <= 8, "a wave's cached postings = one trip of <= 8 steps");
static_assert(kSubLog2 <= 16 && kSubLog2 + 12 <= 32, "packed LDS entries");
template <int N> struct IxN { static constexpr int value = N; };   // compile-time trip lengths
#ifdef TVZ_IX_STAMP
// diagnostic build only (profiles/ix_stamps.py): cycles wave 0 of every block spends per phase
__device__ unsigned long long g_ix_stamps[16];
#define TVZ_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
                          st_acc[i] += _t - st_last; st_last = _t; } } while (0)
#else
#define TVZ_STAMP(i) do { } while (0)
#endif

// fused per-shard top-k (TOPK): the block keeps the k best hits of its query by (kth, video_id, count)
constexpr int kIxTkCap = 128;              // candidate entries (8 B each) held between compactions
constexpr int kIxTkMaxK = 64;              // k up to this (kIxTkCap >= 2 k)
constexpr int kIxTkBins = 64;              // kth histogram: bins 0..62 exact, bin 63 = "63 or later"
constexpr size_t kIxTkBytes = (size_t)kIxTkCap * 8 + (size_t)kIxTkBins * 4 + 8;

// dynamic LDS: [bm1][bm2][tcnt][ttop][cache][lst u64 x waves x 64][lbits][top-k: entries, histogram,
// fill][e_cur x L][elist u16][rank u16][e_len u16 x L x nsb] (nsb = sub-indexes this block walks,
// rounded up to even): 32.5 KiB (+ 1.3 KiB with the fused top-k) + 20 B per query position at 7..8
// sub-indexes - FOUR blocks (32 waves) per CU for queries of up to ~300 timestamps.
inline int ix_nsb_padded(int spb) { return (spb + 1) & ~1; }
inline size_t ix_lds_bytes(int max_len, int spb, bool topk = false) {
    const size_t L = (size_t)(max_len > 0 ? max_len : 1);
    return (size_t)2 * kIxWords * 4 + (size_t)kIxSlots * 14 + (size_t)kIxCache * 4 + (size_t)kIxWaves * 64 * 8 +
           (size_t)kIxWaves * kIxLW * 4 + (topk ? kIxTkBytes : 0) + (L + 1) * 4 + (size_t)kIxWords * 2 +
           L * 2 * (size_t)ix_nsb_padded(spb) + 16;
}

// MODE (how a candidate's slot accounts for its postings):
//   kIxM2    (min_match 1..2): the two smallest positions in two atomicMin words - two plain LDS atomics
//            per candidate posting instead of the 5 x 12-bit CAS loop;
//   kIxTop5  (min_match 3..5): the five smallest positions in one 64-bit CAS word;
//   kIxCount (min_match > 5): the count alone; a hit leaves with kth = -2 - row and ts_kth_fixup_kernel
//            resolves it (the walk the sweeps use for min_match > 5), so that an indexed corpus never
//            falls back to sweeping every row.
// grid = (Q, groups): block (q, g) walks sub-indexes [g * spb, min(n_sub, (g + 1) * spb)).
//   HOSTOUT (tvz_find_duplicates): hits = pinned host memory [n_sub][kSubRows][3], hits_n[sub] = the
//            sub-index's hit count (every sub-index has its own region: no atomics, any grouping).
//   else   : hits = [Q][cap][3].  groups == 1: the block owns the query's list - it appends without
//            atomics and STORES hits_n[q] at the end (the caller need not zero it).  groups > 1: the
//            blocks of a query share the list through atomicAdd on hits_n[q] (zeroed by the caller).
//   TOPK   (tvz_match_topk / tvz_match_sharded, groups == 1, k <= kIxTkMaxK): NO hit list at all.  The
//            block keeps the k best hits of its query - ascending (kth, video_id, count), the order of
//            the top-k kernels - and writes hits = [Q][k + 1][3]: k rows + the totals row
//            (-1, n, NEVER), n negated when n > cap (the contract of the unfused path, whose list
//            would have been truncated).  A hit is one 64-bit word kth << 44 | video_id << 12 | count.
//            Every hit bumps its kth's bin in a per-query histogram (LDS); after each part's barrier
//            every wave scans the 64 bins (one per lane) for b* = the first bin whose prefix reaches
//            k: hits of later bins can never make the top-k and are dropped, the others (k + the
//            ties of one bin + what b* let through while it was still large: ~50 per query at
//            config 4) are appended to a 128-entry list by an LDS atomic.  If a part's keepers might
//            not fit (known without a barrier: the fill before the part + min(the part's hits, the
//            histogram's prefix at b*)), the list is first reduced to its k best by rank counting,
//            which also yields an exact 64-bit cut-off.  The final k rows are written by rank.
//            Removes 12 B per hit of HBM writes (102 MB per 4096 x 100k batch), their re-read, and
//            the two top-k launches behind every batch.
//
// Who does what.  Query position i belongs to wave (i mod waves), lane (i / waves): a wave owns the
// posting lists of ITS positions and walks them on its own - compaction, list-start masks, pass A
// and pass B need no block barrier and no cross-wave prefix; only the row bitmaps and the candidate
// slots are shared.  A sub-index costs five block barriers (pass A done, rank, slot rows, pass B
// done, emit scan); every LDS array is reset inside the phases by the threads that used it last.
// (Profiled with s_memtime stamps, profiles/ix_stamps.py: with block-wide compaction 55 % of a
// block's cycles were barrier waits.)
constexpr int kIxM2 = 0, kIxTop5 = 1, kIxCount = 2;

// The loops over a long query's later chunks (more than 512 timestamps: rare) address LDS from the lane
// number; hoisted out of the sub-index loop those addresses were live across the whole kernel - at the
// 64-VGPR cap that was a scratch spill per value.  An opaque copy keeps the arithmetic inside the loop.
__device__ __forceinline__ int ix_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ unsigned long long ix_tk_pack(int32_t kth, int32_t vid, uint32_t cnt) {
    return ((unsigned long long)(uint32_t)kth << 44) | ((unsigned long long)(uint32_t)vid << 12) | cnt;
}

// NQ (TOPK only): queries per block.  With NQ = 2 block b takes queries 2 b and 2 b + 1: BOTH are probed in the one
// probe phase at the start (a query of ~200 timestamps occupies 200 of the block's 512 threads there, and the phase is
// two dependent round trips - keys, directory entries - during which the block does nothing else: a third of a
// block's time on a 1/8 shard), then walked one after the other.  `q` = the block's first query, `Q` = queries in the batch.
template <bool HOSTOUT, int MODE, bool TOPK, int NQ = 1>
__device__ __forceinline__ void ix_lookup_body(
    const unsigned char *__restrict__ dir, int dir_bits, int ks, const uint16_t *__restrict__ post,
    const int32_t *__restrict__ ivid, int64_t n_indexed, int32_t n_sub, int32_t spb,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t max_len,
    int32_t min_match, const int32_t *__restrict__ exclude_ids, int32_t exclude_one, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t ns, const QByVal *qv, const int q_first,
    const int group, const int n_groups, const int32_t tk_k = 0, const int32_t Q = 0) {
    static_assert(!(TOPK && (HOSTOUT || MODE == kIxCount)), "the fused top-k needs kth in the block and a device list");
    static_assert(NQ == 1 || (NQ == 2 && TOPK), "two queries per block: the top-k form only");
    constexpr bool TOP5 = MODE == kIxTop5;
    const int dir_log2 = dir_bits & 0xff;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *bm1 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *bm2 = bm1 + kIxWords;
    uint32_t *tcnt = bm2 + kIxWords;
    unsigned long long *ttop = reinterpret_cast<unsigned long long *>(tcnt + kIxSlots);
    uint32_t *m1 = reinterpret_cast<uint32_t *>(ttop), *m2 = m1 + kIxSlots;   // kIxM2: the same 8 B per slot
    uint32_t *pcache_all = reinterpret_cast<uint32_t *>(ttop + kIxSlots);  // cached postings: row | position << kSubLog2
    const int L = NQ * (max_len > 0 ? max_len : 1);        // positions of all of the block's queries
    uint2 *lst_all = reinterpret_cast<uint2 *>(pcache_all + kIxCache);  // per wave: non-empty list j = {first posting - local start, position}
    uint32_t *lbits_all = reinterpret_cast<uint32_t *>(lst_all + kIxWaves * 64);   // per wave: bit t = a list starts at local posting t
    unsigned long long *tkb = reinterpret_cast<unsigned long long *>(lbits_all + kIxWaves * kIxLW);   // TOPK: kept hits
    uint32_t *kh = reinterpret_cast<uint32_t *>(tkb + kIxTkCap);        // TOPK: hits per kth bin, all sub-indexes so far
    uint32_t *tk_n = kh + kIxTkBins;                                    // TOPK: entries in tkb (+ one pad word)
    uint32_t *e_cur_all = TOPK ? tk_n + 2 : reinterpret_cast<uint32_t *>(tkb);   // [L] first posting of position i in the CURRENT sub-index
    uint16_t *elist = reinterpret_cast<uint16_t *>(e_cur_all + L + 1);  // row (in the sub-index) of slot k
    uint16_t *rank = elist + kIxSlots;                                  // candidates before bitmap word j
    uint16_t *e_len_all = rank + kIxWords;                              // [L][nsb] postings of position i per sub-index
    __shared__ uint32_t s_wb[kIxWaves], s_wc[kIxWaves];
    __shared__ uint32_t s_bcast;
    __shared__ uint32_t s_tk[TOPK ? kIxWaves : 1];
    __shared__ unsigned long long s_tkT;

    const int sub_lo = group * spb;
    const int sub_hi = sub_lo + spb < n_sub ? sub_lo + spb : n_sub;
    const int nsb = (spb + 1) & ~1;
    const bool alone = n_groups == 1;                  // this block owns the query's hit list
    const bool byval = !TOPK && q_offsets == nullptr;      // (the query travels in the kernel arguments)
    int64_t qo_[NQ];
    int n_[NQ];
    bool skip_[NQ];                                        // refused (too long), or past the end of the batch
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q = q_first + j;
        skip_[j] = NQ > 1 && q >= Q;
        qo_[j] = 0;
        n_[j] = 0;
        if (skip_[j]) continue;
        qo_[j] = byval ? 0 : q_offsets[q];
        const int64_t n64 = byval ? qv->n : q_offsets[q + 1] - qo_[j];
        if (n64 > max_len) {   // max_query_len was not an upper bound (the LDS arrays are sized from it)
            if (TOPK) {        // what the top-k kernels make of a refused query: padding + the poisoned total
                int32_t *o = hits + (int64_t)q * (tk_k + 1) * 3;
                for (int i = threadIdx.x; i <= tk_k; i += kIxBlock) {
                    o[i * 3 + 0] = -1;
                    o[i * 3 + 1] = i == tk_k ? INT32_MIN : 0;
                    o[i * 3 + 2] = TVZ_KTH_NEVER;
                }
            } else if (HOSTOUT) { for (int s = sub_lo + threadIdx.x; s < sub_hi; s += kIxBlock) hits_n[s] = INT32_MIN; }
            else if (threadIdx.x == 0) hits_n[(size_t)q * ns] = INT32_MIN;
            skip_[j] = true;
            if (NQ == 1) return;
            continue;
        }
        n_[j] = (int)n64;
    }
    const int n_all = NQ == 1 ? n_[0] : n_[0] + n_[NQ - 1];
#ifdef TVZ_IX_STAMP
    unsigned long long st_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // in an SGPR: scalar loop bounds
    uint32_t *pcache = pcache_all + wave * kIxPW;
    uint2 *lst = lst_all + wave * 64;
    uint32_t *lbits = lbits_all + wave * kIxLW;
    auto reset_slot = [&](uint32_t k) {
        tcnt[k] = 0;
        if (TOP5) ttop[k] = kTopNone; else if (MODE == kIxM2) { m1[k] = 0xffffffffu; m2[k] = 0xffffffffu; }
    };
    for (int i = threadIdx.x; i < kIxWords; i += kIxBlock) { bm1[i] = 0; bm2[i] = 0; }
    for (int i = threadIdx.x; i < kIxWaves * kIxLW; i += kIxBlock) lbits_all[i] = 0;
    for (int i = threadIdx.x; i < kIxSlots; i += kIxBlock) reset_slot((uint32_t)i);
    if (TOPK) {
        for (int i = threadIdx.x; i < kIxTkBins + 2; i += kIxBlock) kh[i] = 0;      // histogram, fill, pad
    }
    unsigned long long tk_cut = ~0ull;                     // TOPK: hits >= this cannot make the top-k (block-uniform)
    uint32_t tk_bmax = 0xffffffffu;                        // TOPK: hits with kth beyond this neither (block-uniform)

    // ---- directory: ONE probe per query position; the counts of this block's sub-indexes to LDS ----
    const uint32_t smask = dir_bits < 0 ? 0u : (1u << (dir_bits >> 8)) - 1u;   // probes wrap inside the directory slice
    const int es = 16 + 2 * ks;
    for (int i = threadIdx.x; i < n_all; i += kIxBlock) {
        uint32_t base = 0, total = 0;
        const unsigned char *ent = nullptr;
        int64_t k;
        // (position i of the block = position i of its first query, or i - n_[0] of its second)
        const int64_t kq = NQ == 1 || i < n_[0] ? qo_[0] + i : qo_[NQ - 1] + (i - n_[0]);
        if (canon_key(byval ? qv->k[i] : queries[kq], k)) {               // NaN never matches
            if (dir_bits < 0) {                                              // the bucket directory of a one-sub-index handle
                const BkHit hb = bk_find(dir, (uint32_t)(-dir_bits), k);
                base = hb.base;
                total = hb.n;
            } else {
            uint32_t s = ix_slot(k, dir_log2);
            for (int probes = 0; probes < kIxMaxProbe; ++probes) {
                const unsigned char *e = dir + (size_t)s * es;
                const int4 h = *reinterpret_cast<const int4 *>(e);
                const int64_t ek = (int64_t)(((uint64_t)(uint32_t)h.y << 32) | (uint32_t)h.x);
                if (ek == k) { base = (uint32_t)h.z; total = (uint32_t)h.w; ent = e; break; }
                if (ek == kEmpty) break;
                s = (s & ~smask) | ((s + 1) & smask);
            }
            }
        }
        uint16_t *el = e_len_all + (size_t)i * nsb;
        if (ks == 0) {                                                   // one sub-index: the total is its count
            el[0] = (uint16_t)total;
            el[1] = 0;
        } else {
            for (int t = 0; t < nsb; ++t) el[t] = 0;
            if (ent) {
                const uint16_t *cn = reinterpret_cast<const uint16_t *>(ent + 16);
                for (int c = 0; c * 8 < sub_hi; ++c) {                   // 16-byte pieces of the counts
                    const uint4 v = *reinterpret_cast<const uint4 *>(cn + c * 8);
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        const int t = c * 8 + h;
                        const uint32_t len = (w[h >> 1] >> ((h & 1) * 16)) & 0xffffu;
                        if (t < sub_lo) base += len;
                        else if (t < sub_hi) el[t - sub_lo] = (uint16_t)len;
                    }
                }
            }
        }
        e_cur_all[i] = base;
    }
    __syncthreads();
    TVZ_STAMP(0);
#if defined(TVZ_IX_STOP) && TVZ_IX_STOP == 1
    return;
#endif

#pragma unroll 1
    for (int qj = 0; qj < NQ; ++qj) {                      // the block's queries, one after the other
    if (skip_[qj]) continue;                               // (block-uniform)
    const int q = q_first + qj;
    const int n = n_[qj];
    uint32_t *e_cur = e_cur_all + (qj ? n_[0] : 0);
    uint16_t *e_len = e_len_all + (size_t)(qj ? n_[0] : 0) * nsb;
    if (NQ > 1 && qj) {
        // the first query's epilogue has read the kept hits: start the second one's list and histogram
        __syncthreads();
        for (int i = threadIdx.x; i < kIxTkBins + 2; i += kIxBlock) kh[i] = 0;
        tk_cut = ~0ull;
        tk_bmax = 0xffffffffu;
        __syncthreads();
    }
    const int32_t excl = exclude_ids ? exclude_ids[q] : exclude_one;
    const int n_chunks = (n + kIxBlock - 1) / kIxBlock;    // query positions come in chunks of one per thread
    uint32_t emitted = 0;                                  // hits so far (identical in every thread)
    for (int sub = sub_lo; sub < sub_hi; ++sub) {
        if (HOSTOUT) emitted = 0;                          // every sub-index has its own region and count
        // HOSTOUT: the sub-index's own hit region [sub][kSubRows][3] and count in pinned host memory
        int32_t *out_n = HOSTOUT ? hits_n + sub : hits_n + (size_t)q * ns;
        int32_t *out_hits = HOSTOUT ? hits + (int64_t)sub * kSubRows * 3 : hits + (int64_t)q * cap * 3;
        auto touch = [&](uint32_t r) {                     // pass A's bookkeeping for one posting of row r
            const uint32_t bit = 1u << (r & 31u);
            const uint32_t old = atomicOr(&bm1[r >> 5], bit);
            if (min_match >= 2 && (old & bit)) atomicOr(&bm2[r >> 5], bit);
        };
        // ---- this wave's posting lists (position i = chunk * block + lane * waves + wave) ----
        // chunk 0 - every position of a query of up to 512 timestamps - is laid out in the wave's LOCAL
        // flat posting space: non-empty lists compacted by a ballot, their starts marked in a bitmap
        uint32_t off0 = 0, len0 = 0, p0 = 0;               // this lane's chunk-0 list: first posting, length, local start
        uint32_t tw;                                       // postings of the wave's chunk-0 lists
        {
            const int i = lane * kIxWaves + wave;
            if (i < n) {
                len0 = e_len[(size_t)i * nsb + (sub - sub_lo)];
                off0 = e_cur[i];
                e_cur[i] = off0 + len0;                                  // the key's next piece follows
            }
            const uint32_t incl = wave_scan_incl(len0);
            tw = wave_total(incl);
            p0 = incl - len0;
            const unsigned long long some = __ballot(len0 != 0u);
            if (len0) {
                const uint32_t j = __builtin_amdgcn_mbcnt_hi((uint32_t)(some >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)some, 0u));
                lst[j] = make_uint2(off0 - p0, (uint32_t)i);             // local posting t of the wave = post[.x + t]
                if (p0 < (uint32_t)kIxPW) atomicOr(&lbits[p0 >> 5], 1u << (p0 & 31u));
            }
            wave_lds_fence();
        }
        const uint32_t c_hi = tw < (uint32_t)kIxPW ? tw : (uint32_t)kIxPW;   // cached part of the local space
        TVZ_STAMP(1);
        // ---- pass A: which rows are touched (twice) ----
        if (c_hi) {
            // the list of local posting t = (list starts at or before t) - 1.  A step is 64 consecutive
            // postings = two words of the start bitmap, read by the whole wave (one broadcast read); the
            // starts before the step are carried in a scalar: no search, no per-posting table lookup.
            // A trip = N steps with all N posting loads in flight, N = the exact number of steps
            // (eight-step trips with masked-off steps were 40 % of this loop's instructions).
            auto trip = [&](auto nc) {
                constexpr int N = decltype(nc)::value;
                // stage by stage, so that the N steps' LDS round trips overlap: start masks -> list
                // entries -> posting loads.  Lanes past the end of the postings (last step only) read up
                // to 63 postings beyond the wave's last list: the posting buffer is padded for that, and
                // the value is discarded.
                unsigned long long M[N];
#pragma unroll
                for (int u = 0; u < N; ++u) M[u] = *reinterpret_cast<const unsigned long long *>(lbits + 2 * u);
                uint2 e[N];
                uint32_t before = 0;
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(M[u] >> 32),
                                           __builtin_amdgcn_mbcnt_lo((uint32_t)M[u], 0u));   // starts before this lane
                    const uint32_t here = (uint32_t)((M[u] >> lane) & 1ull);
                    e[u] = lst[before + below + here - 1u];
                    before += (uint32_t)__popcll(M[u]);
                }
                uint32_t r[N];
#pragma unroll
#ifdef TVZ_IX_FAKEPOST      // diagnostic build only (WRONG results): no posting line is fetched, the row is made up from the address
                for (int u = 0; u < N; ++u) r[u] = ((e[u].x + (uint32_t)(u * 64 + lane)) * 2654435761u) >> (32 - kSubLog2);
#else
                for (int u = 0; u < N; ++u) r[u] = post[e[u].x + (uint32_t)(u * 64 + lane)];
#endif
#pragma unroll
                for (int u = 0; u < N; ++u) lbits[2 * u + (lane & 1)] = 0;      // done with: ready for the next sub-index
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    const uint32_t t = (uint32_t)(u * 64 + lane);
                    if (t >= c_hi) continue;
                    pcache[t] = r[u] | (e[u].y << kSubLog2);
                    touch(r[u]);
                }
            };
            switch ((c_hi + 63u) >> 6) {                   // wave-uniform (scalar)
                case 1: trip(IxN<1>{}); break;
                case 2: trip(IxN<2>{}); break;
                case 3: trip(IxN<3>{}); break;
                case 4: trip(IxN<4>{}); break;
                case 5: trip(IxN<5>{}); break;
                case 6: trip(IxN<6>{}); break;
                case 7: trip(IxN<7>{}); break;
                default: trip(IxN<8>{}); break;
            }
        }
        // postings outside the cached range - the tail of a wave with more than kIxPW postings in this
        // sub-index (5 % of the waves on the config-4 corpus, 1 % of the postings), and every later chunk
        // of a query of more than 512 timestamps.  The WAVE walks those lists one after the other, 64
        // postings at a time (a lane walking its own list alone was a chain of dependent loads).
        // `each_uncached(f)` calls f(row, position) for all of them; pass B uses it again.
        auto walk_lists = [&](unsigned long long todo, uint32_t off, uint32_t len, uint32_t first, uint32_t posn, auto f) {
            while (todo) {                                 // wave-uniform
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                todo &= todo - 1;
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, src);
                const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, src);
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)first, src);
                const uint32_t ps = (uint32_t)__builtin_amdgcn_readlane((int)posn, src);
                for (uint32_t k = k0 + (uint32_t)lane; k < l; k += 64u) f((uint32_t)post[o + k], ps);
            }
        };
        auto each_uncached = [&](auto f) {
            if (tw > (uint32_t)kIxPW) {
                const uint32_t first = p0 < (uint32_t)kIxPW ? (uint32_t)kIxPW - p0 : 0u;
                walk_lists(__ballot(first < len0), off0, len0, first, (uint32_t)(lane * kIxWaves + wave), f);
            }
            for (int c = 1; c < n_chunks; ++c) {
                const int i = c * kIxBlock + ix_opaque(lane) * kIxWaves + wave;
                const uint32_t len = i < n ? e_len[(size_t)i * nsb + (sub - sub_lo)] : 0u;
                const uint32_t off = i < n ? e_cur[i] - len : 0u;         // (e_cur was advanced below)
                walk_lists(__ballot(len != 0u), off, len, 0u, (uint32_t)i, f);
            }
        };
        for (int c = 1; c < n_chunks; ++c) {               // advance the later chunks' cursors (once per sub-index)
            const int i = c * kIxBlock + ix_opaque(lane) * kIxWaves + wave;
            if (i < n) e_cur[i] += e_len[(size_t)i * nsb + (sub - sub_lo)];
        }
        if (tw > (uint32_t)kIxPW || n_chunks > 1) each_uncached([&](uint32_t r, uint32_t) { touch(r); });
        TVZ_STAMP(2);
        __syncthreads();
        TVZ_STAMP(3);
#if defined(TVZ_IX_STOP) && TVZ_IX_STOP == 3
        for (int i = threadIdx.x; i < kIxWords; i += kIxBlock) { bm1[i] = 0; bm2[i] = 0; }
        __syncthreads();
        continue;
#endif
        const uint32_t *cand = min_match >= 2 ? bm2 : bm1;
        // ---- rank: candidates before every bitmap word (thread t owns words t*kIxWpt .. +kIxWpt-1) ----
        uint32_t cw[kIxWpt], c = 0;
#pragma unroll
        for (int w = 0; w < kIxWpt; ++w) { cw[w] = cand[threadIdx.x * kIxWpt + w]; c += __popc(cw[w]); }
        uint32_t rk0;                                      // candidates before this thread's first word
        {
            const uint32_t incl = wave_scan_incl(c);
            if (lane == 63) s_wb[wave] = incl;
            __syncthreads();
            rk0 = incl - c;
        }
        uint32_t n_cand = 0;
#pragma unroll
        for (int w = 0; w < kIxWaves; ++w) {
            const uint32_t a = s_wb[w];
            if (w < wave) rk0 += a;
            n_cand += a;
        }
        n_cand = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_cand);
        {
            uint32_t run = rk0;
#pragma unroll
            for (int w = 0; w < kIxWpt; ++w) { rank[threadIdx.x * kIxWpt + w] = (uint16_t)run; run += __popc(cw[w]); }
        }
        static_assert(kIxSlots % kIxBlock == 0, "whole rounds of the block over the slots");
        const int64_t row0 = (int64_t)sub << kSubLog2;

        // ---- pass B + emit, kIxSlots candidates at a time ----
        for (uint32_t lo = 0; lo < n_cand; lo += kIxSlots) {
            // the rows of this part's slots (slot = rank of the candidate - lo): written by the owners of
            // the bitmap words, no compaction, no atomics
            {
                uint32_t run = rk0;
#pragma unroll
                for (int ww = 0; ww < kIxWpt; ++ww) {
                    const uint32_t wi = threadIdx.x * kIxWpt + ww;
                    for (uint32_t rest = cw[ww], i = 0; rest; rest &= rest - 1, ++i) {
                        const uint32_t idx = run + i - lo;
                        if (idx < (uint32_t)kIxSlots) elist[idx] = wi * 32u + ((uint32_t)__ffs(rest) - 1u);
                    }
                    run += __popc(cw[ww]);
                }
            }
            TVZ_STAMP(4);
            __syncthreads();                               // (first round: also publishes rank)
            TVZ_STAMP(5);
            const uint32_t n_list = n_cand - lo < (uint32_t)kIxSlots ? n_cand - lo : (uint32_t)kIxSlots;
            // the video ids of the slots' rows: loads issued now, used after pass B (which touches LDS only)
            int32_t vid[kIxSlots / kIxBlock];
#pragma unroll
            for (int u = 0; u < kIxSlots / kIxBlock; ++u) {
                const uint32_t k = (uint32_t)u * kIxBlock + threadIdx.x;
                const int64_t row = row0 + (k < n_list ? elist[k] : 0u);
#ifdef TVZ_IX_NOIVID
                vid[u] = (int32_t)row;
#else
                vid[u] = ivid[k < n_list && row < n_indexed ? row : row0];  // unconditional load
#endif
                // replaced since the build (-1) / the query's own video: not a hit
                if (k >= n_list || row >= n_indexed || vid[u] == excl) vid[u] = -1;
            }
            auto account = [&](uint32_t r, uint32_t pos, uint32_t w, uint32_t rkw) {
                const uint32_t bit = r & 31u;
                if (!((w >> bit) & 1u)) return;
                const uint32_t idx = rkw + __popc(w & ((1u << bit) - 1u)) - lo;
                if (idx >= (uint32_t)kIxSlots) return;                       // another part's (wraps below lo)
                atomicAdd(&tcnt[idx], 1u);
                if constexpr (TOP5) {
                    unsigned long long seen = ttop[idx];
                    while (true) {
                        if (((uint32_t)(seen >> (12 * (kTop - 1))) & 0xfffu) <= pos) break;   // not among the 5 smallest
                        const unsigned long long old = atomicCAS(&ttop[idx], seen, top5_insert(seen, pos));
                        if (old == seen) break;
                        seen = old;
                    }
                } else if constexpr (MODE == kIxM2) {
                    const uint32_t o = atomicMin(&m1[idx], pos);             // positions of one row are distinct
                    atomicMin(&m2[idx], o > pos ? o : pos);                  // larger of two hits >= 2nd smallest
                }
            };
            // the wave's cached postings, up to four per lane at a time: all reads of a stage before the
            // next stage (one posting per trip was a chain of four dependent LDS round trips per posting)
            auto tripb = [&](auto nc, const uint32_t t0) {
                constexpr int N = decltype(nc)::value;
                uint32_t e[N], w[N], rkw[N];
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    const uint32_t t = t0 + (uint32_t)(u * 64 + lane);
                    e[u] = pcache[t < c_hi ? t : c_hi - 1u];
                }
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    const uint32_t r = e[u] & (uint32_t)(kSubRows - 1);
                    w[u] = cand[r >> 5];
                    rkw[u] = rank[r >> 5];
                }
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    if (t0 + (uint32_t)(u * 64 + lane) >= c_hi) continue;
                    account(e[u] & (uint32_t)(kSubRows - 1), e[u] >> kSubLog2, w[u], rkw[u]);
                }
            };
            for (uint32_t t0 = 0; t0 < c_hi; t0 += 256u) {
                const uint32_t steps = (c_hi - t0 + 63u) >> 6;                // wave-uniform (scalar)
                switch (steps >= 4u ? 4u : steps) {
                    case 1: tripb(IxN<1>{}, t0); break;
                    case 2: tripb(IxN<2>{}, t0); break;
                    case 3: tripb(IxN<3>{}, t0); break;
                    default: tripb(IxN<4>{}, t0); break;
                }
            }
            if (tw > (uint32_t)kIxPW || n_chunks > 1)
                each_uncached([&](uint32_t r, uint32_t pos) { account(r, pos, cand[r >> 5], rank[r >> 5]); });
            TVZ_STAMP(6);
            __syncthreads();
            TVZ_STAMP(7);
#if defined(TVZ_IX_STOP) && TVZ_IX_STOP == 4
            for (int i = threadIdx.x; i < kIxSlots; i += kIxBlock) reset_slot((uint32_t)i);
            break;
#endif
            // emit: one slot per thread and round; the slots that reached min_match and are live hits get
            // a place by a block-wide scan - one reservation per block, none when the block owns the list
            auto kth_of = [&](uint32_t k) -> int32_t {
                if constexpr (MODE == kIxCount) return -2 - (int32_t)(row0 + elist[k]);   // ts_kth_fixup_kernel resolves it
                else if constexpr (TOP5) return (int32_t)((uint32_t)(ttop[k] >> (12 * (min_match - 1))) & 0xfffu);
                else return (int32_t)(min_match == 1 ? m1[k] : m2[k]);
            };
            uint32_t mine = 0;
            unsigned long long ek[kIxSlots / kIxBlock];    // TOPK: this thread's hits as sortable words
            uint32_t tk_before = 0;                        // TOPK: kept hits before this part (its appends are
            if constexpr (TOPK) tk_before = *tk_n;         // behind >= 1 barrier: the same value in every thread)
#pragma unroll
            for (int u = 0; u < kIxSlots / kIxBlock; ++u) {
                const uint32_t k = (uint32_t)u * kIxBlock + threadIdx.x;
                if (vid[u] >= 0 && (int32_t)tcnt[k] < min_match) vid[u] = -1;
                mine += vid[u] >= 0 ? 1u : 0u;
                if constexpr (TOPK) {
                    // Only hits that can still make the top-k are looked at any further: kth <= tk_bmax, the
                    // threshold bin of the parts so far (block-uniform; no bound before k hits below position 63
                    // exist).  After the first sub-index that is a few per cent of the hits.  The bin "63 or
                    // later" is never counted: no threshold is ever read from it, and with ~70 % of a part's
                    // hits in it the one LDS address was a serialised atomic per hit.
                    ek[u] = ~0ull;
                    if (vid[u] >= 0) {
                        const uint32_t kth = (uint32_t)kth_of(k);
                        if (kth <= tk_bmax) {
                            ek[u] = ix_tk_pack((int32_t)kth, vid[u], tcnt[k]);
                            if (kth < (uint32_t)kIxTkBins - 1u) atomicAdd(&kh[kth], 1u);
                            mine += 1u << 16;              // (high half: candidates for the list; <= 1024 per part)
                        }
                    }
                }
            }
            const uint32_t incl = wave_scan_incl(mine);
            if (lane == 63) s_wc[wave] = incl;
            // This thread's bitmap words, reset HERE - behind pass B of the last part (every reader is
            // behind the barrier above) and AHEAD of the barrier below: a wave that runs ahead into the
            // next sub-index's pass A sets bits in these words, which a reset after the barrier could
            // wipe (a lost candidate).
            if (lo + kIxSlots >= n_cand) {
#pragma unroll
                for (int w = 0; w < kIxWpt; ++w) { bm1[threadIdx.x * kIxWpt + w] = 0; bm2[threadIdx.x * kIxWpt + w] = 0; }
            }
            __syncthreads();
            TVZ_STAMP(8);
            uint32_t base = 0, all = 0;
#pragma unroll
            for (int x = 0; x < kIxWaves; ++x) {
                const uint32_t a = s_wc[x];
                if (x < wave) base += a;
                all += a;
            }
            if constexpr (TOPK) {
                emitted += all & 0xffffu;
                const uint32_t n_cand_tk = all >> 16;      // this part's hits at or below the threshold bin
                if (n_cand_tk) {                           // block-uniform (nothing to keep otherwise: most later parts)
                    // b* = the first kth bin whose prefix reaches k; every wave scans the bins itself (the
                    // histogram does not change before the next part's emit: same result in all of them)
                    const uint32_t hincl = wave_scan_incl(lane < kIxTkBins - 1 ? kh[lane] : 0u);
                    const unsigned long long reach = __ballot(hincl >= (uint32_t)tk_k);
                    const int bfirst = __builtin_amdgcn_readfirstlane(reach ? __ffsll((long long)reach) - 1 : kIxTkBins);
                    uint32_t bound = n_cand_tk;            // this part's keepers, at most
                    unsigned long long cut = tk_cut;
                    if (bfirst < kIxTkBins - 1) {
                        const uint32_t cum = (uint32_t)__builtin_amdgcn_readlane((int)hincl, bfirst);
                        bound = cum < n_cand_tk ? cum : n_cand_tk;
                        const unsigned long long bc = (unsigned long long)(bfirst + 1) << 44;
                        cut = bc < cut ? bc : cut;
                        tk_bmax = (uint32_t)bfirst < tk_bmax ? (uint32_t)bfirst : tk_bmax;
                    }
                    if (tk_before + bound <= (uint32_t)kIxTkCap) {
#pragma unroll
                        for (int u = 0; u < kIxSlots / kIxBlock; ++u)
                            if (ek[u] < cut) tkb[atomicAdd(tk_n, 1u)] = ek[u];
                    } else {
                        // rare: hundreds of hits in the threshold bin (true duplicates share their kth) or
                        // none of the first k hits below position 63.  Reduce the list to its k best - which
                        // gives the exact cut-off - and feed the part's keepers in rounds of what fits.
                        auto reduce = [&](uint32_t N) -> uint32_t {
                            unsigned long long e = ~0ull;
                            uint32_t r = 0;
                            if (threadIdx.x < N) {
                                e = tkb[threadIdx.x];
                                for (uint32_t i = 0; i < N; ++i) {
                                    const unsigned long long o = tkb[i];
                                    r += (o < e || (o == e && i < threadIdx.x)) ? 1u : 0u;
                                }
                            }
                            __syncthreads();
                            if (threadIdx.x < N && r < (uint32_t)tk_k) tkb[r] = e;
                            if (threadIdx.x < N && r == (uint32_t)tk_k - 1u) s_tkT = e;
                            const uint32_t left = N < (uint32_t)tk_k ? N : (uint32_t)tk_k;
                            if (threadIdx.x == 0) *tk_n = left;
                            __syncthreads();
                            if (N >= (uint32_t)tk_k) {
                                const unsigned long long t = s_tkT;
                                tk_cut = t < tk_cut ? t : tk_cut;
                                const uint32_t tb = (uint32_t)(tk_cut >> 44);          // nothing beyond the k-th best's kth matters
                                tk_bmax = tb < tk_bmax ? tb : tk_bmax;
                            }
                            return left;
                        };
                        uint32_t nb = reduce(tk_before);
                        while (true) {                     // block-uniform
                            cut = tk_cut < cut ? tk_cut : cut;
                            uint32_t c = 0;
#pragma unroll
                            for (int u = 0; u < kIxSlots / kIxBlock; ++u) c += ek[u] < cut ? 1u : 0u;
                            const uint32_t ci = wave_scan_incl(c);
                            if (lane == 63) s_tk[wave] = ci;
                            __syncthreads();
                            uint32_t before = 0, tot = 0;
#pragma unroll
                            for (int x = 0; x < kIxWaves; ++x) {
                                const uint32_t a = s_tk[x];
                                if (x < wave) before += a;
                                tot += a;
                            }
                            if (tot == 0) break;
                            const uint32_t room = (uint32_t)kIxTkCap - nb;
                            uint32_t off = before + ci - c;
#pragma unroll
                            for (int u = 0; u < kIxSlots / kIxBlock; ++u) {
                                if (ek[u] < cut) {
                                    if (off < room) { tkb[nb + off] = ek[u]; ek[u] = ~0ull; }
                                    ++off;
                                }
                            }
                            const uint32_t placed = tot < room ? tot : room;
                            if (threadIdx.x == 0) *tk_n = nb + placed;
                            __syncthreads();
                            nb += placed;
                            if (tot <= room) break;
                            nb = reduce(nb);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kIxSlots / kIxBlock; ++u) {
                    const uint32_t k = (uint32_t)u * kIxBlock + threadIdx.x;
                    if (k < n_list) reset_slot(k);         // this thread was the slot's last reader
                }
            } else {
            uint32_t start = emitted;
            if (!HOSTOUT && !alone && all) {               // block-uniform: the blocks of a query share its list
                if (threadIdx.x == 0) s_bcast = (uint32_t)atomicAdd(out_n, (int32_t)all);
                __syncthreads();
                start = s_bcast;
            }
            emitted += all;
            uint32_t o = start + base + incl - mine;
            const int64_t room = HOSTOUT ? (int64_t)kSubRows : (int64_t)cap;
#pragma unroll
            for (int u = 0; u < kIxSlots / kIxBlock; ++u) {
                const uint32_t k = (uint32_t)u * kIxBlock + threadIdx.x;
                if (vid[u] >= 0) {
#ifdef TVZ_IX_NOSTORE
                    if ((int64_t)o < room && tcnt[k] == 0x7fffffffu) {
#else
                    if ((int64_t)o < room) {
#endif
                        int32_t *hp = out_hits + (int64_t)o * 3;
                        const int32_t kth = kth_of(k);
                        // streaming stores: 100 MB of hits per batch would otherwise push the posting
                        // lines a block comes back to in its next sub-index out of the XCD's 4 MB L2
                        // (HBM reads per launch 559 -> 507 MB, profiles/r3_match_pmc.txt)
                        __builtin_nontemporal_store(vid[u], &hp[0]);
                        __builtin_nontemporal_store((int32_t)tcnt[k], &hp[1]);
                        __builtin_nontemporal_store(kth, &hp[2]);
                    }
                    ++o;
                }
                if (k < n_list) reset_slot(k);             // this thread was the slot's last reader
            }
            }
            if (lo + kIxSlots < n_cand) __syncthreads();   // the next part rewrites elist and refills the slots
        }
        if (n_cand == 0) {                                 // no part ran (pass A touched rows, none twice): reset here,
#pragma unroll                                             // with a barrier before anybody's next pass A
            for (int w = 0; w < kIxWpt; ++w) { bm1[threadIdx.x * kIxWpt + w] = 0; bm2[threadIdx.x * kIxWpt + w] = 0; }
            __syncthreads();
        }
        if (HOSTOUT && threadIdx.x == 0) *out_n = (int32_t)emitted;
        TVZ_STAMP(9);
    }
    if constexpr (TOPK) {
        // the k best of the kept hits, each written to the row of its rank; padding; the totals row
        __syncthreads();                                   // the last part's appends
        const uint32_t N = *tk_n;
        int32_t *o = hits + (int64_t)q * (tk_k + 1) * 3;
        if (threadIdx.x < N) {
            const unsigned long long e = tkb[threadIdx.x];
            uint32_t r = 0;
            for (uint32_t i = 0; i < N; ++i) {
                const unsigned long long x = tkb[i];
                r += (x < e || (x == e && i < threadIdx.x)) ? 1u : 0u;
            }
            if (r < (uint32_t)tk_k) {
                o[r * 3 + 0] = (int32_t)(uint32_t)(e >> 12);
                o[r * 3 + 1] = (int32_t)((uint32_t)e & 0xfffu);
                o[r * 3 + 2] = (int32_t)(e >> 44);
            }
        }
        const uint32_t have = N < (uint32_t)tk_k ? N : (uint32_t)tk_k;
        for (uint32_t i = have + threadIdx.x; i <= (uint32_t)tk_k; i += kIxBlock) {
            o[i * 3 + 0] = -1;
            o[i * 3 + 1] = i == (uint32_t)tk_k ? ((int64_t)emitted > (int64_t)cap ? -(int32_t)emitted : (int32_t)emitted) : 0;
            o[i * 3 + 2] = TVZ_KTH_NEVER;
        }
        TVZ_STAMP(10);
    } else if (!HOSTOUT && alone && threadIdx.x == 0) hits_n[(size_t)q * ns] = (int32_t)emitted;
    }                                                      // (the block's next query)
#ifdef TVZ_IX_STAMP
    if (threadIdx.x == 0) {
        for (int i = 0; i < 11; ++i) atomicAdd(&g_ix_stamps[i], st_acc[i]);
        atomicAdd(&g_ix_stamps[15], 1ull);
    }
#endif
}

template <bool HOSTOUT, int MODE>
__global__ __launch_bounds__(kIxBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void ts_match_index_kernel(
    const unsigned char *__restrict__ dir, int dir_bits, int ks, const uint16_t *__restrict__ post,
    const int32_t *__restrict__ ivid, int64_t n_indexed, int32_t n_sub, int32_t spb,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t max_len,
    int32_t min_match, const int32_t *__restrict__ exclude_ids, int32_t exclude_one, int32_t cap,
    int32_t *__restrict__ hits, int32_t *__restrict__ hits_n, int32_t ns, const QByVal qv) {
    ix_lookup_body<HOSTOUT, MODE, false>(dir, dir_bits, ks, post, ivid, n_indexed, n_sub, spb, queries, q_offsets,
                                         max_len, min_match, exclude_ids, exclude_one, cap, hits, hits_n, ns, &qv,
                                         (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// the lookup with the per-shard top-k in its epilogue: grid = (Q), one block per query over ALL
// sub-indexes; topk = [Q][k + 1][3] (k best + totals row), no hit list, no counters
// NQ = 2: grid = (ceil(Q / 2)), a block takes two queries (probed together, walked in turn)
template <int MODE, int NQ>
__global__ __launch_bounds__(kIxBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void ts_match_index_topk_kernel(
    const unsigned char *__restrict__ dir, int dir_bits, int ks, const uint16_t *__restrict__ post,
    const int32_t *__restrict__ ivid, int64_t n_indexed, int32_t n_sub,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t Q, int32_t max_len,
    int32_t min_match, const int32_t *__restrict__ exclude_ids, int32_t cap, int32_t k,
    int32_t *__restrict__ topk) {
    ix_lookup_body<false, MODE, true, NQ>(dir, dir_bits, ks, post, ivid, n_indexed, n_sub, n_sub, queries, q_offsets,
                                          max_len, min_match, exclude_ids, -1, cap, topk, nullptr, 1, nullptr,
                                          (int)blockIdx.x * NQ, 0, 1, k, Q);
}

// tvz_find_duplicates on an indexed corpus with rows in the delta table - the streaming driver's call:
// the upload's own row was upserted a moment ago - in ONE launch: blocks [0, n_groups) look the query
// up in the index (one sub-index group each), the others sweep the delta table with the single-query
// sweep's body.  Both write to pinned host memory (every block its own region and count), so they
// share nothing; two launches on one stream cost ~8 us more.
template <bool TOP5>
__global__ __launch_bounds__(kIxBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void ts_find_fused_kernel(
    const unsigned char *__restrict__ dir, int dir_bits, int ks, const uint16_t *__restrict__ post,
    const int32_t *__restrict__ ivid, int64_t n_indexed, int32_t n_sub, int32_t spb, int32_t n_groups,
    const double *__restrict__ queries, const int64_t *__restrict__ q_offsets, int32_t max_len,
    int32_t min_match, int32_t exclude_one, int32_t *__restrict__ ix_hits, int32_t *__restrict__ ix_hits_n,
    const Row *__restrict__ delta_rows, int64_t n_delta, const int64_t *__restrict__ keys, int32_t s_log2,
    HostOut ho, const QByVal qv) {
    if ((int)blockIdx.x < n_groups)
        ix_lookup_body<true, TOP5 ? kIxTop5 : kIxM2, false>(dir, dir_bits, ks, post, ivid, n_indexed, n_sub, spb, queries,
                                                            q_offsets, max_len, min_match, nullptr, exclude_one, 0, ix_hits,
                                                            ix_hits_n, 1, &qv, 0, (int)blockIdx.x, n_groups);
    else
        q1_body<TOP5 ? kQ1ModeTop5 : kQ1ModeM2, true, kIxBlock>(delta_rows, n_delta, keys, queries, q_offsets, min_match,
                                                               nullptr, exclude_one, 0, nullptr, nullptr, 1, s_log2, ho,
                                                               qv, (int)blockIdx.x - n_groups,
                                                               (int)gridDim.x - n_groups, 0);
}

}  // namespace
