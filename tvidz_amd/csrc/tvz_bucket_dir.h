// tvz_bucket_dir.h — the directory of a ONE-sub-index handle (up to 2^14 indexed rows): 128-byte BUCKETS that
// hold a key's directory entry AND its postings in the same cache line (gfx950, wave64).
//
// Why: on a rank's share of an 8-way sharded table (BASELINE.json configs[3]) a lookup is one pass over ~2,000
// postings per query; the open-addressing directory of tvz_index_kernels.h (16-byte entries at load <= 0.25:
// 32 MB for 384 k keys) costs every query position TWO random 128-byte lines - the entry's and the posting
// list's (a list is ~6 postings = 12 bytes) - 180 MB of line traffic per batch of 4,096 queries for 37 MB of
// algorithmic bytes (profiles/r4_match_pmc.txt), and the lookup runs at the rate the fabric delivers random
// lines.  Here a probe is ONE line: the bucket of the key's hash holds the key and its postings.
//
// Bucket (128 bytes):
//   byte 0        nk     records in this bucket (0..11)
//   byte 1        spill  keys whose HOME is this bucket lie at most this many buckets further on (inside the
//                        slice of 256 buckets, wrapping): a lookup that does not find its key at home walks on
//                        that far and no further
//   bytes 2..13   off[0..nk]  first posting of record j, in uint16 units from the start of the bucket (bits 0..6);
//                        bit 7: the record's postings are EXTERNAL - its 8 bytes at off[j] hold {first posting as
//                        a uint16 index from the start of the directory, postings} (external records come first in
//                        a bucket: their descriptors stay 8-byte aligned); off[nk] = end of the last record
//   bytes 16 + 8 j      key of record j (canonical float64 bits)
//   from 16 + 8 nk      the records' postings (uint16 row numbers) back to back
// A list of up to kBkInline postings lives in its bucket; longer ones in the external area behind the buckets
// (whole 128-byte lines each).  The whole directory is one allocation that the lookups address as uint16[].
//
// Build: the (key, row) pairs are partitioned by slice exactly as for the classic directory
// (ix_partition_kernel / ix_scatter_kernel), then ONE block per slice makes the slice's 256 buckets in LDS -
// distinct keys + counts in a scratch hash table, greedy packing by home bucket, the few keys that do not fit
// walk on to the next bucket with room - fills the postings and writes 32 KB of whole lines.
#pragma once
// (included by tvz_index_kernels.h behind its constants: kSubRows, IxBuildInfo)

namespace {

constexpr int kBkBytes = 128;                    // a bucket = one cache line
constexpr int kBkU16 = kBkBytes / 2;
constexpr int kBkSliceLog2 = 8;                  // buckets per slice (probing wraps inside a slice)
constexpr int kBkSlice = 1 << kBkSliceLog2;
constexpr int kBkMaxRec = 11;                    // 16 + 10 nk <= 128
constexpr int kBkInline = 48;                    // postings of a record that live in the bucket (8 + 96 bytes)
constexpr int kBkPayload = kBkBytes - 16;        // bytes behind the header: keys + postings
constexpr int kBkTable = 2048;                   // distinct keys of a slice the build's scratch table holds (load <= 0.9)
constexpr int kBkBuildBlock = 512;
constexpr uint32_t kBkExtLine = 64;              // external lists are whole lines of 64 postings
constexpr uint32_t kBkPlaced = 0x40000000u;      // (build) the key's final place is known

// the 32-bit hash a key's bucket is cut from, and the (independent) one of the build's scratch table
__device__ __forceinline__ uint32_t bk_hash(int64_t k) { return q1_mix(k) * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t bk_bucket(int64_t k, uint32_t nb) { return __umulhi(bk_hash(k), nb); }
__device__ __forceinline__ uint32_t bk_thash(int64_t k) {
    uint32_t x = (uint32_t)k * 0x85EBCA6Bu ^ (uint32_t)((uint64_t)k >> 32) * 0xC2B2AE35u;
    x ^= x >> 15;
    return x * 0x27D4EB2Fu;
}

struct BkHit {
    uint32_t base;      // first posting, uint16 index from the start of the directory
    uint32_t n;         // postings (0: the key is in no indexed row)
};

// bytes p and p + 1 (p = 2 .. 13) of a 16-byte header held in four dwords, as one 16-bit value.  (64-bit shifts:
// picking the dword by a ?: chain made the compiler park the header in SCRATCH and index it there.)
__device__ __forceinline__ uint32_t bk_hdr_pair(const uint4 &h, uint32_t p) {
    const unsigned long long lo = (unsigned long long)h.x | ((unsigned long long)h.y << 32);
    const unsigned long long hi = (unsigned long long)h.z | ((unsigned long long)h.w << 32);
    const uint32_t sh = 8u * p;
    const unsigned long long v = sh < 64u ? (lo >> sh) | (hi << (64u - sh)) : hi >> (sh - 64u);   // (p >= 2: 64 - sh < 64)
    return (uint32_t)v & 0xffffu;
}

// record j of the bucket at uint16 index `b16` (header `h`): where its postings are
__device__ __forceinline__ BkHit bk_record(const uint16_t *__restrict__ dir16, uint32_t b16, const uint4 &h, uint32_t j) {
    const uint32_t oo = bk_hdr_pair(h, 2u + j);
    const uint32_t o0 = oo & 0xffu, o1 = oo >> 8;
    BkHit r;
    r.base = b16 + (o0 & 0x7fu);
    r.n = (o1 & 0x7fu) - (o0 & 0x7fu);
    if (o0 & 0x80u) {                                   // external: {first posting, postings}
        const uint2 d = *reinterpret_cast<const uint2 *>(dir16 + r.base);
        r.base = d.x;
        r.n = d.y;
    }
    return r;
}

// index of key (klo, khi) among six keys held in three 16-byte vectors (the smallest; 0xff: none)
__device__ __forceinline__ uint32_t bk_match6(const uint4 &a, const uint4 &b, const uint4 &c, uint32_t klo, uint32_t khi) {
    uint32_t j = 0xffu;
    if (c.z == klo && c.w == khi) j = 5u;
    if (c.x == klo && c.y == khi) j = 4u;
    if (b.z == klo && b.w == khi) j = 3u;
    if (b.x == klo && b.y == khi) j = 2u;
    if (a.z == klo && a.w == khi) j = 1u;
    if (a.x == klo && a.y == khi) j = 0u;
    return j;
}

// Walk on from bucket b0 (the key's home, already searched) as far as its `spill` says.  Rare: a twelfth key of a
// bucket, or a bucket of many keys with lists too short to move out.
__device__ __forceinline__ BkHit bk_walk(const unsigned char *__restrict__ dir, uint32_t b0, uint32_t spill, int64_t k) {
    const uint16_t *dir16 = reinterpret_cast<const uint16_t *>(dir);
    for (uint32_t d = 1; d <= spill; ++d) {
        const uint32_t b = (b0 & ~(uint32_t)(kBkSlice - 1)) | ((b0 + d) & (uint32_t)(kBkSlice - 1));
        const unsigned char *line = dir + (size_t)b * kBkBytes;
        const uint4 h = *reinterpret_cast<const uint4 *>(line);
        const uint32_t nk = h.x & 0xffu;
        const int64_t *keys = reinterpret_cast<const int64_t *>(line + 16);
        for (uint32_t j = 0; j < nk; ++j)
            if (keys[j] == k) return bk_record(dir16, b * (uint32_t)kBkU16, h, j);
    }
    return BkHit{0u, 0u};
}

// Settle a probe whose home bucket's first 64 bytes - header `hd` and keys 0..5 in `ka`, `kb`, `kc` - are loaded.
// Everything else it may need lies in the SAME line (keys 6..10, an external list's descriptor), except after a
// walk (bk_walk).
__device__ __forceinline__ BkHit bk_settle(const unsigned char *__restrict__ dir, uint32_t b, int64_t k, const uint4 &hd,
                                           const uint4 &ka, const uint4 &kb, const uint4 &kc) {
    const uint16_t *dir16 = reinterpret_cast<const uint16_t *>(dir);
    const uint32_t nk = hd.x & 0xffu, spill = (hd.x >> 8) & 0xffu;
    const uint32_t klo = (uint32_t)k, khi = (uint32_t)((uint64_t)k >> 32);
    uint32_t j = bk_match6(ka, kb, kc, klo, khi);
    if (j >= nk) {                                      // (bytes behind the last key are postings: never a match)
        j = 0xffu;
        if (nk > 6u) {
            const uint4 *line = reinterpret_cast<const uint4 *>(dir + (size_t)b * kBkBytes);
            const uint4 kd = line[4], ke = line[5], kf = line[6];
            const uint32_t j2 = bk_match6(kd, ke, kf, klo, khi);
            if (j2 + 6u < nk) j = j2 + 6u;
        }
    }
    if (j < nk) return bk_record(dir16, b * (uint32_t)kBkU16, hd, j);
    if (spill) return bk_walk(dir, b, spill, k);
    return BkHit{0u, 0u};
}

// The whole probe for one key.
__device__ __forceinline__ BkHit bk_find(const unsigned char *__restrict__ dir, uint32_t nb, int64_t k) {
    const uint32_t b = bk_bucket(k, nb);
    const uint4 *line = reinterpret_cast<const uint4 *>(dir + (size_t)b * kBkBytes);
    const uint4 hd = line[0], ka = line[1], kb = line[2], kc = line[3];
    return bk_settle(dir, b, k, hd, ka, kb, kc);
}

// ---- build: one block per slice of 256 buckets ----------------------------------------------------------------
constexpr size_t kBkBuildLds = (size_t)kBkTable * 16      /* keys, counts, places */
                               + (size_t)kBkSlice * kBkBytes /* the slice's image */
                               + (size_t)kBkTable * 2 * 3   /* order, spill list, guest links */
                               + (size_t)kBkSlice * 2 * 6   /* per bucket: home keys, first, used bytes, records, guests, spill */
                               + 64;

__global__ __launch_bounds__(kBkBuildBlock) void bk_slice_build_kernel(
    const int64_t *__restrict__ pkeys, const uint32_t *__restrict__ prows, const uint32_t *__restrict__ start,
    unsigned char *__restrict__ dir, uint32_t nb, uint32_t ext_cap16, IxBuildInfo *info) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bk_sh[];
    int64_t *tkey = reinterpret_cast<int64_t *>(bk_sh);                            // [kBkTable] kEmpty = free
    uint32_t *tcnt = reinterpret_cast<uint32_t *>(tkey + kBkTable);                // postings of the key (counted down by the fill)
    uint32_t *tplace = tcnt + kBkTable;                                            // first the key's bucket; then kBkPlaced | uint16 index of its first posting | external << 31
    uint4 *img = reinterpret_cast<uint4 *>(tplace + kBkTable);                     // [kBkSlice][8] the buckets
    uint16_t *order = reinterpret_cast<uint16_t *>(img + kBkSlice * 8);            // table slots sorted by home bucket
    uint16_t *spl = order + kBkTable;                                              // table slots that did not fit at home
    uint16_t *gnext = spl + kBkTable;                                              // guest list links (by table slot)
    uint16_t *bhome = gnext + kBkTable;                                            // [kBkSlice] keys whose home is the bucket
    uint16_t *bfirst = bhome + kBkSlice;                                           // first of them in `order`
    uint16_t *bused = bfirst + kBkSlice;                                           // payload bytes in use
    uint16_t *bnk = bused + kBkSlice;                                              // records
    uint16_t *bguest = bnk + kBkSlice;                                             // head of the guest list (0xffff: none)
    uint16_t *bspill = bguest + kBkSlice;                                          // how far a key of this home walked
    __shared__ uint32_t s_nkeys, s_nspl, s_ext, s_extbase, s_fail, s_w[kBkBuildBlock / 64];
    const uint32_t part = blockIdx.x;
    const uint32_t lo = start[part], hi = start[part + 1];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kBkTable; i += kBkBuildBlock) { tkey[i] = kEmpty; tcnt[i] = 0; tplace[i] = 0; }
    for (int i = tid; i < kBkSlice * 8; i += kBkBuildBlock) img[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < kBkSlice; i += kBkBuildBlock) { bhome[i] = 0; bused[i] = 0; bnk[i] = 0; bguest[i] = 0xffffu; bspill[i] = 0; }
    if (tid == 0) { s_nkeys = 0; s_nspl = 0; s_ext = 0; s_fail = 0; }
    __syncthreads();
    // ---- distinct keys + posting counts ----
    auto tfind = [&](int64_t k, bool insert) -> int {
        uint32_t s = bk_thash(k) >> (32 - 11);
        static_assert(kBkTable == 1 << 11, "scratch table of 2^11 slots");
        for (int probes = 0; probes < kBkTable; ++probes) {
            unsigned long long *kp = reinterpret_cast<unsigned long long *>(tkey + s);
            const int64_t cur = (int64_t)*reinterpret_cast<volatile unsigned long long *>(kp);
            if (cur == k) return (int)s;
            if (cur == kEmpty) {
                if (!insert) return -1;
                const unsigned long long old = atomicCAS(kp, (unsigned long long)kEmpty, (unsigned long long)k);
                if (old == (unsigned long long)kEmpty) { atomicAdd(&s_nkeys, 1u); return (int)s; }
                if ((int64_t)old == k) return (int)s;
            }
            s = (s + 1) & (uint32_t)(kBkTable - 1);
        }
        return -1;
    };
    for (uint32_t j = lo + (uint32_t)tid; j < hi; j += kBkBuildBlock) {
        // (a table filled to the brim makes every insert a long walk: give up on this directory size early)
        if (s_nkeys > (uint32_t)(kBkTable * 9 / 10)) { s_fail = 1; break; }
        const int s = tfind(pkeys[j], true);
        if (s < 0) { s_fail = 1; break; }
        atomicAdd(&tcnt[s], 1u);
    }
    __syncthreads();
    if (s_fail) { if (tid == 0) info->failed = 1; return; }                        // (block-uniform)
    // ---- keys by home bucket (counting sort) ----
    for (int s = tid; s < kBkTable; s += kBkBuildBlock)
        if (tkey[s] != kEmpty) atomicAdd(reinterpret_cast<uint32_t *>(bhome) + ((bk_bucket(tkey[s], nb) & (kBkSlice - 1)) >> 1),
                                         1u << (16 * (bk_bucket(tkey[s], nb) & 1u)));
    __syncthreads();
    if (tid < kBkSlice) {                                                          // exclusive scan of bhome over 256 buckets (4 waves)
        const uint32_t c = bhome[tid];
        const uint32_t incl = wave_scan_incl(c);
        if (lane == 63) s_w[wave] = incl;
        bfirst[tid] = (uint16_t)(incl - c);
    }
    __syncthreads();
    if (tid < kBkSlice) {
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        bfirst[tid] = (uint16_t)(bfirst[tid] + before);
        bhome[tid] = 0;                                                            // (re-used as the fill cursor of the sort)
    }
    __syncthreads();
    for (int s = tid; s < kBkTable; s += kBkBuildBlock)
        if (tkey[s] != kEmpty) {
            const uint32_t hb = bk_bucket(tkey[s], nb) & (kBkSlice - 1);
            const uint32_t sh = 16 * (hb & 1u);
            const uint32_t at = (atomicAdd(reinterpret_cast<uint32_t *>(bhome) + (hb >> 1), 1u << sh) >> sh) & 0xffffu;
            order[bfirst[hb] + at] = (uint16_t)s;
        }
    __syncthreads();
    // ---- membership.  The KEYS always stay at home (up to 11 per bucket: a twelfth walks on - at ~3 keys per
    // bucket that is one in 10^5); what does not fit is POSTINGS: a record whose list does not fit the bucket keeps
    // its key and an 8-byte descriptor there and its list in the external area.  A lookup then never needs a second
    // bucket, and only the lists that were moved out cost their lookups a second line.  Home keys are sorted by
    // list length, longest first: a query timestamp is drawn from the rows, so a key is asked for in proportion to
    // its postings - the first six keys of a bucket (what a lookup loads at once) are the ones asked for most.
    if (tid < kBkSlice) {
        const uint32_t n_all = bhome[tid], f = bfirst[tid];
        for (uint32_t i = 1; i < n_all; ++i) {                                     // insertion sort, descending by postings
            const uint16_t s = order[f + i];
            const uint32_t n = tcnt[s];
            uint32_t j = i;
            for (; j > 0 && tcnt[order[f + j - 1]] < n; --j) order[f + j] = order[f + j - 1];
            order[f + j] = s;
        }
        // lists over kBkInline postings are external anyway; of the others, while the bucket is over its payload the
        // SHORTEST list that frees enough goes out (else the longest of them): the cost of moving a list out is the
        // share of lookups that ask for it, i.e. its length.  Many keys with lists too short to be worth a descriptor
        // (eleven keys of two postings: 132 bytes): the last key walks on instead.
        uint32_t nk = n_all < (uint32_t)kBkMaxRec ? n_all : (uint32_t)kBkMaxRec;
        uint32_t used = 0, ext_mask = 0;
        while (true) {
            used = 8u * nk;
            ext_mask = 0;
            for (uint32_t i = 0; i < nk; ++i) {
                const uint32_t n = tcnt[order[f + i]];
                if (n > (uint32_t)kBkInline) { ext_mask |= 1u << i; used += 8u; } else used += 2u * n;
            }
            while (used > (uint32_t)kBkPayload) {
                const uint32_t need = used - (uint32_t)kBkPayload;
                int pick = -1;
                for (int i = (int)nk - 1; i >= 0; --i) {                           // from the shortest list up
                    if (ext_mask & (1u << i)) continue;
                    const uint32_t n = tcnt[order[f + i]];
                    if (2u * n > 8u && 2u * n - 8u >= need) { pick = i; break; }
                }
                if (pick < 0)
                    for (int i = 0; i < (int)nk; ++i)                              // none frees enough alone: the longest one
                        if (!(ext_mask & (1u << i)) && 2u * tcnt[order[f + i]] > 8u) { pick = i; break; }
                if (pick < 0) break;
                ext_mask |= 1u << pick;
                used -= 2u * tcnt[order[f + pick]] - 8u;
            }
            if (used <= (uint32_t)kBkPayload) break;
            --nk;
        }
        for (uint32_t i = nk; i < n_all; ++i) { tplace[order[f + i]] = 0xffffffffu; spl[atomicAdd(&s_nspl, 1u)] = order[f + i]; }
        for (uint32_t i = 0; i < nk; ++i) tplace[order[f + i]] = (uint32_t)tid | ((ext_mask >> i) & 1u) << 16;   // bucket | external << 16
        bused[tid] = (uint16_t)used;
        bnk[tid] = (uint16_t)nk;
    }
    __syncthreads();
    if (tid == 0 && s_nspl) {                                                      // a twelfth key: on to the next bucket with 16 bytes to spare
        uint32_t worst = 0;
        for (uint32_t i = 0; i < s_nspl; ++i) {
            const uint32_t s = spl[i];
            const uint32_t hb = bk_bucket(tkey[s], nb) & (kBkSlice - 1);
            uint32_t d = 1;
            for (; d < (uint32_t)kBkSlice; ++d) {
                const uint32_t b = (hb + d) & (kBkSlice - 1);
                if (bused[b] + 16u <= (uint32_t)kBkPayload && bnk[b] < (uint32_t)kBkMaxRec) {
                    bused[b] = (uint16_t)(bused[b] + 16u);
                    bnk[b] = (uint16_t)(bnk[b] + 1);
                    tplace[s] = b | 1u << 16;                                      // a guest's list is always external
                    gnext[s] = bguest[b];
                    bguest[b] = (uint16_t)s;
                    if (d > bspill[hb]) bspill[hb] = (uint16_t)d;
                    break;
                }
            }
            if (d == (uint32_t)kBkSlice) s_fail = 1;                               // the slice is full
            worst = d > worst ? d : worst;
        }
        atomicAdd(&info->n_spilled, s_nspl);
        atomicMax(&info->max_spill, worst);
    }
    __syncthreads();
    if (s_fail) { if (tid == 0) info->failed = 1; return; }
    // ---- records: header, keys, the external records' descriptors, then the inline lists back to back ----
    if (tid < kBkSlice) {
        unsigned char *bk = reinterpret_cast<unsigned char *>(img + tid * 8);
        const uint32_t nk = bnk[tid];
        uint32_t off = 8u + 4u * nk;                                               // uint16 units: behind the header and the keys
        uint32_t j = 0;
        auto put = [&](uint32_t s, bool want_ext) {
            const uint32_t n = tcnt[s];
            const bool ext = (tplace[s] >> 16) & 1u;
            if (ext != want_ext) return;
            reinterpret_cast<int64_t *>(bk + 16)[j] = tkey[s];
            bk[2 + j] = (unsigned char)(off | (ext ? 0x80u : 0u));
            if (ext) {
                const uint32_t e = atomicAdd(&s_ext, (n + kBkExtLine - 1u) & ~(kBkExtLine - 1u));
                tplace[s] = kBkPlaced | 0x80000000u | e;                           // (relative to the block's range: made absolute below)
                reinterpret_cast<uint32_t *>(bk)[off / 2] = e;
                reinterpret_cast<uint32_t *>(bk)[off / 2 + 1] = n;
                off += 4u;
            } else {
                tplace[s] = kBkPlaced | ((uint32_t)tid * kBkU16 + off);            // uint16 index inside the slice
                off += n;
            }
            ++j;
        };
        const uint32_t n_home = bhome[tid], f = bfirst[tid];
        for (int pass = 0; pass < 2; ++pass) {                                     // external records first: aligned descriptors
            for (uint32_t i = 0; i < n_home; ++i) {
                const uint32_t s = order[f + i];
                if (!(tplace[s] & kBkPlaced) && (tplace[s] & 0xffffu) == (uint32_t)tid) put(s, pass == 0);   // (not one that walked on)
            }
            for (uint32_t s = bguest[tid]; s != 0xffffu; s = gnext[s])
                if (!(tplace[s] & kBkPlaced)) put(s, pass == 0);
        }
        bk[2 + j] = (unsigned char)off;                                            // end of the last record
        bk[0] = (unsigned char)nk;
        bk[1] = (unsigned char)bspill[tid];
    }
    __syncthreads();
    if (tid == 0) {
        s_extbase = s_ext ? atomicAdd(&info->ext_cursor, s_ext) : 0u;
        if (s_ext && (s_extbase + s_ext > ext_cap16 || s_extbase + s_ext < s_extbase)) { s_fail = 1; info->failed = 2; }
        atomicAdd(&info->n_distinct, s_nkeys);
    }
    __syncthreads();
    if (s_fail) return;
    const uint32_t ext0 = nb * (uint32_t)kBkU16 + s_extbase;                       // uint16 index of the block's external range
    if (s_ext) {
        if (tid < kBkSlice) {                                                      // external descriptors: absolute first posting
            unsigned char *bk = reinterpret_cast<unsigned char *>(img + tid * 8);
            const uint32_t nk = bk[0];
            uint32_t n_here = 0;
            for (uint32_t j = 0; j < nk; ++j)
                if (bk[2 + j] & 0x80u) { reinterpret_cast<uint32_t *>(bk)[(bk[2 + j] & 0x7fu) / 2] += ext0; ++n_here; }
            if (n_here) atomicAdd(&info->n_ext, n_here);
        }
        __syncthreads();
    }
    // ---- postings ----
    uint16_t *img16 = reinterpret_cast<uint16_t *>(img);
    uint16_t *dir16 = reinterpret_cast<uint16_t *>(dir);
    for (uint32_t j = lo + (uint32_t)tid; j < hi; j += kBkBuildBlock) {
        const int s = tfind(pkeys[j], false);
        const uint32_t k = atomicSub(&tcnt[s], 1u) - 1u;
        const uint32_t pl = tplace[s];
        const uint16_t row = (uint16_t)(prows[j] & (uint32_t)(kSubRows - 1));
        if (pl & 0x80000000u) dir16[(size_t)ext0 + (pl & 0x3fffffffu) + k] = row;
        else img16[(pl & 0x3fffffffu) + k] = row;
    }
    __syncthreads();
    uint4 *g = reinterpret_cast<uint4 *>(dir + (size_t)part * kBkSlice * kBkBytes);
    for (int i = tid; i < kBkSlice * 8; i += kBkBuildBlock) g[i] = img[i];
}

}  // namespace
