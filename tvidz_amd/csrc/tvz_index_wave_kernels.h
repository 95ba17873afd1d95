// tvz_index_wave_kernels.h — the index lookup with the per-shard top-k, ONE WAVE PER QUERY (gfx950, wave64).
//
// For a handle of ONE sub-index (up to 2^14 rows: what every GPU of an 8-way shard of BASELINE.json configs[3]
// holds) the block-per-query lookup of tvz_index_kernels.h pays its per-query part once per block of eight
// waves: three dependent global round trips (offsets -> keys -> directory line) in which 200 of 512 threads
// take part, five block barriers, rank / slot / emit phases for 512 threads - for ONE pass over ~2,000
// postings.  The counters of round 4 (profiles/r4_match_pmc.txt) show the result: 59 % of the wave-cycles
// waiting, 1,187 VALU instructions per wave x 8 waves per query.
//
// Here a wave owns a query from its keys to its k + 1 output rows (find_duplicates, /root/reference
// inspector/db.py:85-91, + the first-hit order of inspector/app.py:238-245):
//   * no barrier anywhere - a workgroup is one wave; LDS traffic of a wave completes in order;
//   * the probes of ALL query positions (chunks of 64) are in flight together, two directory slots per probe
//     (a probe chain of up to two steps is one round trip), the rare longer chains of all chunks advance
//     together;
//   * the postings - the wave's flat posting space, exactly the layout of ix_lookup_body's pass A: list starts
//     in a bitmap, `starts before me` by mbcnt - stay in REGISTERS between the passes (64 steps of 64 postings:
//     the (row, position) cache of the block kernel costs 16 KiB of LDS per query);
//   * pass A (row bitmaps seen1 / seen2), rank, slots, pass B (count + smallest positions per candidate), the
//     kth histogram and the kept-hit list are those of the block kernel, by one wave.
// ~14 KiB of LDS per wave and <= 168 VGPRs: eleven waves per CU, each with 200+ probes and up to 64 posting loads
// in flight.
// Results are identical to ts_match_index_topk_kernel (tests/test_index_topk_gpu.py runs both on the same handle).
#pragma once
#include "tvz_index_kernels.h"

namespace {

constexpr int kWqChunks = 8;                         // chunks of 64 query positions
constexpr int kWqMaxLen = kWqChunks * 64;            // longest query this kernel takes
constexpr int kWqSteps = 64;                         // register-resident steps of 64 postings
constexpr int kWqRegPost = kWqSteps * 64;            // postings kept in registers between the passes
#ifndef TVZ_WQ_GS
#define TVZ_WQ_GS 4
#endif
constexpr int kWqGs = TVZ_WQ_GS;                       // steps per group: their LDS (and global) round trips overlap
constexpr int kWqSlots = 512;                        // candidate slots per part (384: 12 waves per CU instead of 11, but every sixth query of the 1/8 shard took two parts - no faster)
// This kernel's own row range: handles of up to 2^14 indexed rows, whatever a sub-index of the block kernel holds
// (TVZ_IX_SUB_LOG2) - eight bitmap words per lane, rank and slot rows in one bitmap, row | position << 14 in a register.
constexpr int kWqRowsLog2 = 14;
constexpr int kWqRows = 1 << kWqRowsLog2;
constexpr int kWqWords = kWqRows / 32;               // words per bitmap
constexpr bool kWqUsable = kWqRowsLog2 <= kSubLog2;  // (a handle this kernel takes is ONE sub-index of the build)
constexpr int kWqPosShift = kWqRowsLog2;             // register entry = row | position << 14
static_assert(kWqRowsLog2 + 9 <= 32 && kWqMaxLen <= 512, "packed register entries");
constexpr size_t kWqTkBytes = (size_t)kIxTkCap * 8 + (size_t)kIxTkBins * 4 + 16;   // (16: the list table behind it stays 16-byte aligned)

// dynamic LDS of one wave (bytes), by the longest query of the batch
inline size_t wq_lds_bytes(int max_len) {
    const size_t L = (size_t)((max_len > 0 ? max_len : 1) + 1) & ~(size_t)1;
    return (size_t)2 * kWqWords * 4        /* seen1, seen2; behind pass A the one that is not the candidates' holds rank + slot rows */
           + (size_t)kWqSlots * 12         /* count + smallest positions */
           + (size_t)kWqRegPost / 8        /* list-start bitmap */
           + kWqTkBytes                    /* kept hits, kth histogram, fill */
           + L * 8;                        /* non-empty lists */
}

#ifdef TVZ_IX_STAMP
// diagnostic build only (profiles/wq_stamps.py): cycles every wave spends per phase, and the XCC / CU it ran on
#define TVZ_WQ_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
                             if (lane == 0) st_acc[i] += _t - st_last; st_last = _t; } while (0)
#else
#define TVZ_WQ_STAMP(i) do { } while (0)
#endif

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void ts_match_wq_topk_kernel(
    const unsigned char *__restrict__ dir, int dir_bits, const uint16_t *__restrict__ post,
    const int32_t *__restrict__ ivid, int64_t n_indexed, const double *__restrict__ queries,
    const int64_t *__restrict__ q_offsets, int32_t Q, int32_t max_len, int32_t min_match,
    const int32_t *__restrict__ exclude_ids, int32_t cap, int32_t tk_k, int32_t *__restrict__ topk) {
    static_assert(MODE == kIxM2 || MODE == kIxTop5, "kth is known inside the wave for min_match 1..5");
    constexpr bool TOP5 = MODE == kIxTop5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *bm1 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *bm2 = bm1 + kWqWords;
    uint32_t *tcnt = bm2 + kWqWords;
    unsigned long long *ttop = reinterpret_cast<unsigned long long *>(tcnt + kWqSlots);
    uint32_t *m12 = reinterpret_cast<uint32_t *>(ttop);   // kIxM2: the same 8 B per slot = {smallest, second smallest} position
    uint32_t *lbits = reinterpret_cast<uint32_t *>(ttop + kWqSlots);                 // bit t: a list starts at local posting t
    unsigned long long *tkb = reinterpret_cast<unsigned long long *>(lbits + kWqRegPost / 32);   // kept hits
    uint32_t *kh = reinterpret_cast<uint32_t *>(tkb + kIxTkCap);                     // hits per kth bin
    uint32_t *tk_n = kh + kIxTkBins;                                                 // entries in tkb (+ one pad word)
    uint2 *lst = reinterpret_cast<uint2 *>(tk_n + 4);                                // non-empty list j = {first posting - local start, position}
    // rank (candidates before bitmap word j) and the rows of the part's slots live in the bitmap that is NOT the
    // candidates' - seen1 for min_match >= 2, seen2 for min_match 1 - which is dead once pass A is over
    uint16_t *rank = reinterpret_cast<uint16_t *>(min_match >= 2 ? bm1 : bm2);
    uint16_t *elist = rank + kWqWords;                                               // row of slot k
    static_assert((kWqWords + kWqSlots) * 2 <= kWqWords * 4, "rank + slot rows fit one bitmap");

    const int q = (int)blockIdx.x;
    const int lane = (int)threadIdx.x;
    int32_t *o = topk + (int64_t)q * (tk_k + 1) * 3;
    const int64_t qo = q_offsets[q];
    const int64_t n64 = q_offsets[q + 1] - qo;
    if (n64 > max_len || n64 > kWqMaxLen) {   // max_query_len was not an upper bound: padding + the poisoned total
        for (int i = lane; i <= tk_k; i += 64) {
            o[i * 3 + 0] = -1;
            o[i * 3 + 1] = i == tk_k ? INT32_MIN : 0;
            o[i * 3 + 2] = TVZ_KTH_NEVER;
        }
        return;
    }
    const int n = (int)n64;
    const int nch = __builtin_amdgcn_readfirstlane((n + 63) >> 6);
#ifdef TVZ_IX_STAMP
    unsigned long long st_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif

    // ---- LDS set-up (one wave: stores complete in order, no barrier) ----
    {
        uint4 *z = reinterpret_cast<uint4 *>(bm1);                       // seen1 + seen2: 4 KiB = 256 x 16 B
#pragma unroll
        for (int u = 0; u < 4; ++u) z[u * 64 + lane] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < kWqSlots / 64; ++u) {
            const int k = u * 64 + lane;
            tcnt[k] = 0;
            ttop[k] = TOP5 ? kTopNone : ~0ull;
        }
        lbits[lane] = 0;
        lbits[lane + 64] = 0;
        static_assert(kWqRegPost / 32 == 128, "two start-bitmap words per lane");
        kh[lane] = 0;
        if (lane < 2) tk_n[lane] = 0;
    }

    TVZ_WQ_STAMP(0);
    // ---- directory: ONE probe per query position, all chunks in flight together ----
    // ---- then the wave's flat posting space: list i starts at local posting p0; non-empty lists compacted ----
    // list entry j = {first posting - local start, position | length << 9}: all the tail walk needs is in LDS.
    // One straight-line instance per chunk count (a branch per chunk made every chunk wait for its own loads: a
    // chain of up to 16 global round trips in front of the first posting).
    // The bucket directory (tvz_bucket_dir.h): the home bucket's header and first six keys - 64 bytes of ONE line,
    // which also holds the postings - for every position at once; what that leaves (a bucket of more than six
    // records, a key that walked on, an external list's descriptor) is rare and resolved per lane.
    const uint32_t nb = (uint32_t)(-dir_bits);
    const uint4 *dirv = reinterpret_cast<const uint4 *>(dir);
    uint32_t tw = 0, n_lists = 0;
    auto probe_and_layout = [&](auto nc, const int c0) {   // chunks c0 .. c0 + N - 1
        constexpr int N = decltype(nc)::value;
        uint32_t off_[N], len_[N];
        int64_t key_[N];
        bool ok_[N];
        double x_[N];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const int i = (c0 + c) * 64 + lane;
            x_[c] = queries[qo + (i < n ? i : n - 1)];                   // (unconditional: the loads go out together)
        }
#pragma unroll
        for (int c = 0; c < N; ++c) {
            key_[c] = 0;
            ok_[c] = (c0 + c) * 64 + lane < n && canon_key(x_[c], key_[c]);     // NaN never matches
        }
        uint32_t b_[N];
        uint4 hd_[N], ka_[N], kb_[N], kc_[N];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            b_[c] = ok_[c] ? bk_bucket(key_[c], nb) : 0u;
            const uint4 *line = dirv + (size_t)b_[c] * (kBkBytes / 16);
            hd_[c] = line[0];
            ka_[c] = line[1];
            kb_[c] = line[2];
            kc_[c] = line[3];
        }
        unsigned long long slow = 0;                                     // (wave-uniform) lanes the first step did not settle
        bool slow_[N];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            off_[c] = 0;
            len_[c] = 0;
            slow_[c] = false;
            const uint32_t nk = hd_[c].x & 0xffu, spill = (hd_[c].x >> 8) & 0xffu;
            const uint32_t j = bk_match6(ka_[c], kb_[c], kc_[c], (uint32_t)key_[c], (uint32_t)((uint64_t)key_[c] >> 32));
            if (ok_[c]) {
                if (j < nk) {                                            // (bytes behind the last key are postings: never a match)
                    const uint32_t oo = bk_hdr_pair(hd_[c], 2u + j);
                    const uint32_t o0 = oo & 0xffu, o1 = oo >> 8;
                    if (o0 & 0x80u) slow_[c] = true;                     // external: the descriptor is one more (cached) load
                    else { off_[c] = b_[c] * (uint32_t)kBkU16 + (o0 & 0x7fu); len_[c] = (o1 & 0x7fu) - (o0 & 0x7fu); }
                } else if (nk > 6u || spill) slow_[c] = true;
            }
            slow |= __ballot(slow_[c]);
        }
        if (slow) {
#pragma unroll
            for (int c = 0; c < N; ++c) {
                if (slow_[c]) {
                    const BkHit h = bk_settle(dir, b_[c], key_[c], hd_[c], ka_[c], kb_[c], kc_[c]);
                    off_[c] = h.base;
                    len_[c] = h.n;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const uint32_t incl = wave_scan_incl(len_[c]);
            const uint32_t p0 = tw + incl - len_[c];
            const unsigned long long some = __ballot(len_[c] != 0u);
            if (len_[c]) {
                const uint32_t j = n_lists + __builtin_amdgcn_mbcnt_hi((uint32_t)(some >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)some, 0u));
                lst[j] = make_uint2(off_[c] - p0, (uint32_t)((c0 + c) * 64 + lane) | (len_[c] << 9));    // local posting t of the wave = post[.x + t]
                if (p0 < (uint32_t)kWqRegPost) atomicOr(&lbits[p0 >> 5], 1u << (p0 & 31u));
            }
            tw += wave_total(incl);
            n_lists += (uint32_t)__popcll(some);
        }
    };
    // four chunks (256 positions) at a time: their 16 directory loads are in flight together; a longer query takes
    // a second round (eight at a time needed more registers than three waves per SIMD leave)
#pragma unroll 1
    for (int c0 = 0; c0 < nch; c0 += 4) {                  // wave-uniform
        switch (nch - c0) {
            case 1: probe_and_layout(IxN<1>{}, c0); break;
            case 2: probe_and_layout(IxN<2>{}, c0); break;
            case 3: probe_and_layout(IxN<3>{}, c0); break;
            default: probe_and_layout(IxN<4>{}, c0); break;
        }
    }
    wave_lds_fence();
    TVZ_WQ_STAMP(1);
    const uint32_t c_hi = tw < (uint32_t)kWqRegPost ? tw : (uint32_t)kWqRegPost;     // postings held in registers
    const int n_steps = __builtin_amdgcn_readfirstlane((int)((c_hi + 63u) >> 6));
    const bool tail = tw > (uint32_t)kWqRegPost;           // (wave-uniform) postings beyond the registers: walked from memory

    auto touch = [&](uint32_t r) {                         // pass A's bookkeeping for one posting of row r
        const uint32_t bit = 1u << (r & 31u);
        const uint32_t old = atomicOr(&bm1[r >> 5], bit);
        if (min_match >= 2 && (old & bit)) atomicOr(&bm2[r >> 5], bit);
    };
    // the lists' postings beyond the register range, one list after the other, 64 postings at a time (rare: a query
    // with more than 4,096 postings on this shard).  Everything it needs comes from the list table in LDS - local
    // starts are the prefix sums of the lengths - so that no per-chunk register stays live across the passes.
    auto each_tail = [&](auto f) {
        uint32_t run = 0;                                  // local start of the batch's first list (wave-uniform)
#pragma unroll 1
        for (uint32_t j0 = 0; j0 < n_lists; j0 += 64u) {
            const uint32_t j = j0 + (uint32_t)lane;
            const uint2 e = j < n_lists ? lst[j] : make_uint2(0u, 0u);
            const uint32_t len = j < n_lists ? e.y >> 9 : 0u;
            const uint32_t incl = wave_scan_incl(len);
            const uint32_t p0 = run + incl - len;
            run += wave_total(incl);
            if (run <= (uint32_t)kWqRegPost) continue;     // (wave-uniform) every list so far ends inside the registers
            // the list's first posting outside the registers
            const uint32_t p0c = p0 < (uint32_t)kWqRegPost ? p0 : (uint32_t)kWqRegPost;
            const uint32_t first = (uint32_t)kWqRegPost - p0c < len ? (uint32_t)kWqRegPost - p0c : len;
            const uint32_t ob_l = e.x + p0;                // the list's first posting
            unsigned long long todo = __ballot(first < len);
            while (todo) {                                 // wave-uniform
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                todo &= todo - 1;
                const uint32_t ob = (uint32_t)__builtin_amdgcn_readlane((int)ob_l, src);
                const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, src);
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)first, src);
                const uint32_t ps = (uint32_t)__builtin_amdgcn_readlane((int)(e.y & 0x1ffu), src);
                for (uint32_t k = k0 + (uint32_t)lane; k < l; k += 64u) f((uint32_t)post[ob + k], ps);
            }
        }
    };

    // ---- pass A: which rows are touched (twice); the postings stay in registers ----
    uint32_t R[kWqSteps];
    {
        // Stage 1 - ALL posting loads of the query are issued before the first one is used (up to 64 in flight per
        // lane; a group at a time, each waiting for its own loads, was a chain of 8-16 global round trips per query).
        // The list of local posting t = (list starts at or before t) - 1: a step is 64 consecutive postings = two
        // words of the start bitmap, read by the whole wave; the starts before the step are carried in a scalar.
        // A live group of four steps runs all four loads: steps past the end read up to 255 + 63 entries beyond the
        // last list - the posting buffer is padded for that (kIxPostPad) - and their values are discarded.
        uint32_t P[kWqSteps / 2];                          // positions of the steps' lists, two per register (stage 2 folds them into R)
        uint32_t before = 0;                               // list starts before the current step (wave-uniform)
#pragma unroll
        for (int g = 0; g < kWqSteps / kWqGs; ++g) {
            if (g * kWqGs < n_steps) {                         // (scalar branch)
                uint2 e[kWqGs];
#pragma unroll
                for (int u = 0; u < kWqGs; ++u) {
                    const unsigned long long Mv = *reinterpret_cast<const unsigned long long *>(lbits + 2 * (g * kWqGs + u));
                    const uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)Mv);
                    const uint32_t mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(Mv >> 32));
                    const unsigned long long M = ((unsigned long long)mhi << 32) | mlo;      // in SGPRs
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));   // starts before this lane
                    const uint32_t here = (uint32_t)((M >> lane) & 1ull);
                    const uint32_t j = before + below + here;          // (>= 1 wherever a posting exists)
                    e[u] = lst[j ? j - 1u : 0u];
                    before += (uint32_t)__popcll(M);
                }
#pragma unroll
#ifdef TVZ_WQ_NOPOST      // diagnostic build only (WRONG results): no posting is fetched, the row is made up from the address
                for (int u = 0; u < kWqGs; ++u) R[g * kWqGs + u] = ((e[u].x + (uint32_t)((g * kWqGs + u) * 64 + lane)) * 2654435761u) >> (32 - kWqRowsLog2);
#else
                for (int u = 0; u < kWqGs; ++u) R[g * kWqGs + u] = (uint32_t)post[e[u].x + (uint32_t)((g * kWqGs + u) * 64 + lane)];
#endif
#pragma unroll
                for (int u = 0; u < kWqGs; u += 2)
                    P[(g * kWqGs + u) >> 1] = (e[u].y & 0x1ffu) | ((e[u + 1].y & 0x1ffu) << 16);
            } else {
#pragma unroll
                for (int u = 0; u < kWqGs; ++u) R[g * kWqGs + u] = 0;
#pragma unroll
                for (int u = 0; u < kWqGs; u += 2) P[(g * kWqGs + u) >> 1] = 0;
            }
        }
        TVZ_WQ_STAMP(2);
        // Stage 2 - pass A proper: every posting sets its row's bit in seen1, or in seen2 if it was set already
#pragma unroll
        for (int g = 0; g < kWqSteps / kWqGs; ++g) {
            if (g * kWqGs < n_steps) {
                uint32_t r[kWqGs], old[kWqGs];
#pragma unroll
                for (int u = 0; u < kWqGs; ++u) {
                    const bool live = (uint32_t)((g * kWqGs + u) * 64 + lane) < c_hi;
                    r[u] = live ? R[g * kWqGs + u] : 0u;
                    // (a posting that does not exist ORs nothing into word 0)
                    old[u] = atomicOr(&bm1[r[u] >> 5], live ? 1u << (r[u] & 31u) : 0u);
                    R[g * kWqGs + u] = r[u] | (((P[(g * kWqGs + u) >> 1] >> ((u & 1) * 16)) & 0x1ffu) << kWqPosShift);
                }
                if (min_match >= 2) {
#pragma unroll
                    for (int u = 0; u < kWqGs; ++u) {
                        const uint32_t bit = 1u << (r[u] & 31u);
                        const bool live = (uint32_t)((g * kWqGs + u) * 64 + lane) < c_hi;
                        if (live && (old[u] & bit)) atomicOr(&bm2[r[u] >> 5], bit);
                    }
                }
            }
        }
    }
    if (tail) each_tail([&](uint32_t r, uint32_t) { touch(r); });
    wave_lds_fence();
    TVZ_WQ_STAMP(3);

    // ---- rank: candidates before every bitmap word (lane l owns words 8 l .. 8 l + 7) ----
    const uint32_t *cand = min_match >= 2 ? bm2 : bm1;
    constexpr int kWpl = kWqWords / 64;                    // bitmap words per lane
    uint32_t cw[kWpl], rk0, n_cand;
    {
        const uint4 a = *reinterpret_cast<const uint4 *>(cand + lane * kWpl);
        const uint4 b = *reinterpret_cast<const uint4 *>(cand + lane * kWpl + 4);
        cw[0] = a.x; cw[1] = a.y; cw[2] = a.z; cw[3] = a.w; cw[4] = b.x; cw[5] = b.y; cw[6] = b.z; cw[7] = b.w;
        static_assert(kWpl == 8, "eight bitmap words per lane");
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < kWpl; ++w) c += __popc(cw[w]);
        const uint32_t incl = wave_scan_incl(c);
        rk0 = incl - c;
        n_cand = wave_total(incl);
        uint32_t run = rk0, pk[kWpl / 2];
#pragma unroll
        for (int w = 0; w < kWpl; w += 2) {
            const uint32_t lo16 = run;
            run += __popc(cw[w]);
            pk[w / 2] = lo16 | (run << 16);
            run += __popc(cw[w + 1]);
        }
        *reinterpret_cast<uint4 *>(rank + lane * kWpl) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    wave_lds_fence();

    TVZ_WQ_STAMP(4);
    unsigned long long tk_cut = ~0ull;                     // hits >= this cannot make the top-k (wave-uniform)
    uint32_t tk_bmax = 0xffffffffu;                        // hits with kth beyond this neither (wave-uniform)
    uint32_t emitted = 0;
    const int32_t excl = exclude_ids ? exclude_ids[q] : -1;

    // ---- pass B + emit, kWqSlots candidates at a time ----
    for (uint32_t lo = 0; lo < n_cand; lo += kWqSlots) {
        {   // the rows of this part's slots (slot = rank of the candidate - lo), written by the owners of the bitmap words
            uint32_t run = rk0;
#pragma unroll
            for (int ww = 0; ww < kWpl; ++ww) {
                const uint32_t wi = (uint32_t)(lane * kWpl + ww);
                for (uint32_t rest = cw[ww], i = 0; rest; rest &= rest - 1, ++i) {
                    const uint32_t idx = run + i - lo;
                    if (idx < (uint32_t)kWqSlots) elist[idx] = (uint16_t)(wi * 32u + ((uint32_t)__ffs(rest) - 1u));
                }
                run += __popc(cw[ww]);
            }
        }
        wave_lds_fence();
        const uint32_t n_list = n_cand - lo < (uint32_t)kWqSlots ? n_cand - lo : (uint32_t)kWqSlots;
        // the video ids of the slots' rows: loads issued now, used after pass B (which touches LDS only)
        int32_t vid[kWqSlots / 64];
#pragma unroll
        for (int u = 0; u < kWqSlots / 64; ++u) {
            const uint32_t k = (uint32_t)(u * 64 + lane);
            const int64_t row = k < n_list ? (int64_t)elist[k] : 0;
            vid[u] = ivid[k < n_list && row < n_indexed ? row : 0];         // unconditional load
            // replaced since the build (-1) / the query's own video: not a hit
            if (k >= n_list || row >= n_indexed || vid[u] == excl) vid[u] = -1;
        }
        // a candidate's slot accounts for one of its postings: the count, and the smallest query positions
        TVZ_WQ_STAMP(5);
        auto account_slot = [&](uint32_t idx, uint32_t pos) {
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
                const uint32_t o1 = atomicMin(&m12[2 * idx], pos);       // positions of one row are distinct
                atomicMin(&m12[2 * idx + 1], o1 > pos ? o1 : pos);       // larger of two hits >= 2nd smallest
            }
        };
        auto slot_of = [&](uint32_t r, uint32_t w, uint32_t rkw) -> uint32_t {   // kWqSlots or more: not a candidate of this part
            const uint32_t bit = r & 31u;
            const uint32_t idx = rkw + __popc(w & ((1u << bit) - 1u)) - lo;      // (another part's wraps below lo)
            return ((w >> bit) & 1u) ? idx : 0xffffffffu;
        };
        auto account = [&](uint32_t r, uint32_t pos) {
            const uint32_t idx = slot_of(r, cand[r >> 5], rank[r >> 5]);
            if (idx < (uint32_t)kWqSlots) account_slot(idx, pos);
        };
#pragma unroll
        for (int g = 0; g < kWqSteps / kWqGs; ++g) {
            if (g * kWqGs < n_steps) {
                // all reads of a stage before the next stage (dead steps hold row 0 / position 0: masked below)
                // (an opaque copy of the register entry: everything derived from it - word index, bit mask, position -
                // is otherwise hoisted out of the loop over the parts for all 64 steps at once, and spilled)
                uint32_t w[kWqGs], rkw[kWqGs], rr[kWqGs];
#pragma unroll
                for (int u = 0; u < kWqGs; ++u) {
                    rr[u] = (uint32_t)ix_opaque((int)R[g * kWqGs + u]);
                    const uint32_t r = rr[u] & (uint32_t)(kWqRows - 1);
                    w[u] = cand[r >> 5];
                    rkw[u] = rank[r >> 5];
                }
#pragma unroll
                for (int u = 0; u < kWqGs; ++u) {
                    const uint32_t r = rr[u] & (uint32_t)(kWqRows - 1);
                    const uint32_t idx = slot_of(r, w[u], rkw[u]);
                    if ((uint32_t)((g * kWqGs + u) * 64 + lane) < c_hi && idx < (uint32_t)kWqSlots)
                        account_slot(idx, rr[u] >> kWqPosShift);
                }
            }
        }
        if (tail) each_tail([&](uint32_t r, uint32_t pos) { account(r, pos); });
        wave_lds_fence();
        TVZ_WQ_STAMP(6);

        // emit: the slots that reached min_match and are live hits; only those that can still make the top-k are kept
        auto kth_of = [&](uint32_t k) -> int32_t {
            if constexpr (TOP5) return (int32_t)((uint32_t)(ttop[k] >> (12 * (min_match - 1))) & 0xfffu);
            else return (int32_t)m12[2 * k + (min_match == 1 ? 0 : 1)];
        };
        // A slot that is a hit turns into its sortable word IN PLACE (the 8 bytes that held its smallest positions):
        // keeping eight 64-bit words per lane in registers across the keep phase spilled.
        uint32_t mine = 0;
        const uint32_t tk_before = *tk_n;
#pragma unroll
        for (int u = 0; u < kWqSlots / 64; ++u) {
            const uint32_t k = (uint32_t)(u * 64 + lane);
            const uint32_t cnt = tcnt[k];
            if (vid[u] >= 0 && (int32_t)cnt < min_match) vid[u] = -1;
            mine += vid[u] >= 0 ? 1u : 0u;
            unsigned long long ek = ~0ull;
            if (vid[u] >= 0) {
                const uint32_t kth = (uint32_t)kth_of(k);
                if (kth <= tk_bmax) {
                    ek = ix_tk_pack((int32_t)kth, vid[u], cnt);
                    if (kth < (uint32_t)kIxTkBins - 1u) atomicAdd(&kh[kth], 1u);
                    mine += 1u << 16;                      // (high half: candidates for the list; <= 512 per part)
                }
            }
            ttop[k] = ek;
        }
        const uint32_t all = wave_total(wave_scan_incl(mine));
        emitted += all & 0xffffu;
        const uint32_t n_cand_tk = all >> 16;              // this part's hits at or below the threshold bin
        wave_lds_fence();
        if (n_cand_tk) {                                   // wave-uniform
            // b* = the first kth bin whose prefix reaches k
            const uint32_t hincl = wave_scan_incl(lane < kIxTkBins - 1 ? kh[lane] : 0u);
            const unsigned long long reach = __ballot(hincl >= (uint32_t)tk_k);
            const int bfirst = __builtin_amdgcn_readfirstlane(reach ? __ffsll((long long)reach) - 1 : kIxTkBins);
            uint32_t bound = n_cand_tk;                    // this part's keepers, at most
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
                for (int u = 0; u < kWqSlots / 64; ++u) {
                    const unsigned long long ek = ttop[u * 64 + lane];
                    if (ek < cut) tkb[atomicAdd(tk_n, 1u)] = ek;
                }
            } else {
                // rare: hundreds of hits in the threshold bin (true duplicates share their kth) or none of the
                // first k hits below position 63.  Reduce the list to its k best - which gives the exact cut-off -
                // and feed the part's keepers in rounds of what fits.
                auto reduce = [&](uint32_t N) -> uint32_t {            // N <= kIxTkCap = 128: two entries per lane
                    unsigned long long e2[2] = {~0ull, ~0ull};
                    uint32_t r2[2] = {0, 0};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t me = (uint32_t)(h * 64 + lane);
                        if (me < N) {
                            e2[h] = tkb[me];
                            for (uint32_t i = 0; i < N; ++i) {
                                const unsigned long long x = tkb[i];
                                r2[h] += (x < e2[h] || (x == e2[h] && i < me)) ? 1u : 0u;
                            }
                        }
                    }
                    wave_lds_fence();                      // every read of the old list before the first write of the new
                    unsigned long long kthbest = ~0ull;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t me = (uint32_t)(h * 64 + lane);
                        if (me < N && r2[h] < (uint32_t)tk_k) tkb[r2[h]] = e2[h];
                        if (me < N && r2[h] == (uint32_t)tk_k - 1u) kthbest = e2[h];
                    }
                    const uint32_t left = N < (uint32_t)tk_k ? N : (uint32_t)tk_k;
                    if (lane == 0) *tk_n = left;
                    wave_lds_fence();
                    if (N >= (uint32_t)tk_k) {
                        // the k-th best entry: held by exactly one lane; min over the wave (the others hold ~0)
                        unsigned long long t = kthbest;
#pragma unroll
                        for (int d = 32; d > 0; d >>= 1) {
                            const unsigned long long other = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(t >> 32), d) << 32) |
                                                             (uint32_t)__shfl_xor((int)(uint32_t)t, d);
                            t = other < t ? other : t;
                        }
                        tk_cut = t < tk_cut ? t : tk_cut;
                        const uint32_t tb = (uint32_t)(tk_cut >> 44);  // nothing beyond the k-th best's kth matters
                        tk_bmax = tb < tk_bmax ? tb : tk_bmax;
                    }
                    return left;
                };
                uint32_t nb = reduce(tk_before);
                while (true) {                             // wave-uniform
                    cut = tk_cut < cut ? tk_cut : cut;
                    uint32_t c = 0;
#pragma unroll 1
                    for (int u = 0; u < kWqSlots / 64; ++u) c += ttop[u * 64 + lane] < cut ? 1u : 0u;
                    const uint32_t ci = wave_scan_incl(c);
                    const uint32_t tot = wave_total(ci);
                    if (tot == 0) break;
                    const uint32_t room = (uint32_t)kIxTkCap - nb;
                    uint32_t off = ci - c;
#pragma unroll 1
                    for (int u = 0; u < kWqSlots / 64; ++u) {
                        const unsigned long long ek = ttop[u * 64 + lane];
                        if (ek < cut) {
                            if (off < room) { tkb[nb + off] = ek; ttop[u * 64 + lane] = ~0ull; }
                            ++off;
                        }
                    }
                    const uint32_t placed = tot < room ? tot : room;
                    if (lane == 0) *tk_n = nb + placed;
                    wave_lds_fence();
                    nb += placed;
                    if (tot <= room) break;
                    nb = reduce(nb);
                }
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int u = 0; u < kWqSlots / 64; ++u) {          // the part's slots: ready for the next part
            const uint32_t k = (uint32_t)(u * 64 + lane);
            if (k < n_list) {
                tcnt[k] = 0;
                ttop[k] = TOP5 ? kTopNone : ~0ull;
            }
        }
        wave_lds_fence();
        TVZ_WQ_STAMP(7);
    }

    // ---- the k best of the kept hits, each written to the row of its rank; padding; the totals row ----
    {
        wave_lds_fence();
        const uint32_t N = *tk_n;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t me = (uint32_t)(h * 64 + lane);
            if (me < N) {
                const unsigned long long e = tkb[me];
                uint32_t r = 0;
                for (uint32_t i = 0; i < N; ++i) {
                    const unsigned long long x = tkb[i];
                    r += (x < e || (x == e && i < me)) ? 1u : 0u;
                }
                if (r < (uint32_t)tk_k) {
                    o[r * 3 + 0] = (int32_t)(uint32_t)(e >> 12);
                    o[r * 3 + 1] = (int32_t)((uint32_t)e & 0xfffu);
                    o[r * 3 + 2] = (int32_t)(e >> 44);
                }
            }
        }
        const uint32_t have = N < (uint32_t)tk_k ? N : (uint32_t)tk_k;
        for (uint32_t i = have + (uint32_t)lane; i <= (uint32_t)tk_k; i += 64u) {
            o[i * 3 + 0] = -1;
            o[i * 3 + 1] = i == (uint32_t)tk_k ? ((int64_t)emitted > (int64_t)cap ? -(int32_t)emitted : (int32_t)emitted) : 0;
            o[i * 3 + 2] = TVZ_KTH_NEVER;
        }
    }
    TVZ_WQ_STAMP(8);
#ifdef TVZ_IX_STAMP
    if (lane == 0) {
        for (int i = 0; i < 9; ++i) atomicAdd(&g_ix_stamps[i], st_acc[i]);
        atomicAdd(&g_ix_stamps[14], (unsigned long long)tw);
        atomicAdd(&g_ix_stamps[15], 1ull);
    }
#endif
}

}  // namespace
