// tvz_scene.hip — scene-cut scoring kernels for MI355X (gfx950, wave64).
//
// Replaces the arithmetic reached through /root/reference inspector/app.py:202-209
// (ffmpeg `select=gt(scene\,0.3)`: upstream libavfilter f_select.c get_scene_score +
// scene_sad.c).  Integer part (luma SAD) is exact; the epilogue reproduces the
// double / float32 operation order of get_scene_score.
//
// Kernel 1  luma_sad_flat_kernel<U>   HBM-bound streaming reduction.
//   A wave owns a spatial strip of U KiB (64 lanes x U x 16 B) and WALKS TIME over a
//   chunk of `tc` frames keeping the previous frame's strip in registers, so every
//   luma byte crosses HBM once (+1 halo frame per chunk).  16-byte loads, 1 KiB per
//   wave-instruction, v_sad_u8 (4 bytes / instruction) into a per-lane u32, DPP row
//   reduction + row_bcast to lane 63 per frame, result parked in lane (t & 63) and
//   written as one 256-byte store per 64 frames to partial[strip][t].
//   No LDS, no atomics, no MFMA (a reduction, not a contraction).
// Kernel 2  scene_finalize_kernel     sums partial[.][t] over strips (u64) and applies
//   the get_scene_score epilogue (mafd, diff, float32 clip, threshold).
// Kernel 1g luma_sad_generic_kernel   same structure on 4-byte granules for padded
//   rows / unaligned planes (slower, identical results).
// Kernel 3  scene_tail_kernel         one block: compacts the selected frame indices into the
//   caller's cut list and advances the device-resident stream state.
//
// Stream state (tvz.h tvz_scene_state_*): what get_scene_score carries between frames lives in
// caller-owned DEVICE memory: {prev_mafd, have_prev, parity} + two frame buffers.  The SAD kernel
// reads the predecessor frame from buffer[parity] and the waves that reach the batch's last
// frame store their strip of it into buffer[parity ^ 1] (the bytes are in registers anyway);
// the tail kernel flips parity.  No host value crosses a batch boundary, so a chain of
// micro-batches is one stream-ordered (graph-capturable) sequence of launches.
#include <algorithm>

#include "tvz_common.h"

namespace {

constexpr int kWave = 64;
#ifndef TVZ_SCENE_BLOCK
#define TVZ_SCENE_BLOCK 256
#endif
constexpr int kBlock = TVZ_SCENE_BLOCK;  // waves per block x 64
constexpr int kWavesPerBlock = kBlock / kWave;

// per-call kernel shape (tvz.h TVZ_SHAPE): there is no process-global knob
struct Tuning {
    int U = 0;     // 16-byte loads per lane per frame on the flat path (1,2,4,8); 0 = auto
    int tc = 0;    // frames per time chunk (8, 16, 32 or a multiple of 64); 0 = auto
    int nt = 1;    // non-temporal loads on the flat path (+8..12 % on MI355X, profiles/r1_tune_scene.txt)
};

// device-resident stream state: header + two tightly packed frame buffers
struct StateHdr {
    double prev_mafd;      // mafd of the last frame scored so far (0 before any)
    int32_t have_prev;     // 0 right after a reset: the next frame is the first of the stream
    int32_t parity;        // which frame buffer holds the predecessor frame
    int64_t frames_seen;
};
constexpr size_t kStateHdrBytes = 256;
// first 256 bytes of the (aligned) workspace
struct WsHdr {
    double last_mafd;      // mafd of the batch's last frame (finalize -> tail kernel)
};
constexpr size_t kWsHdrBytes = 256;

struct StateArgs {
    StateHdr *hdr;         // nullptr: self-contained batch
    uint8_t *buf0, *buf1;
};

// predecessor of frame 0, where to keep this batch's last frame, first scored frame
struct Carry {
    const uint8_t *prev0;
    uint8_t *tail;
    int32_t t_first;
};
__device__ __forceinline__ Carry read_carry(const StateArgs &sa) {
    Carry c{nullptr, nullptr, 1};
    if (sa.hdr) {                      // wave-uniform scalar loads
        const int32_t have = sa.hdr->have_prev, par = sa.hdr->parity;
        c.prev0 = par ? sa.buf1 : sa.buf0;
        c.tail = par ? sa.buf0 : sa.buf1;
        c.t_first = have ? 0 : 1;
    }
    return c;
}

// ---- wave64 sum to lane 63 with DPP (row_shr within 16-lane rows, then row_bcast) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    // lanes whose DPP source is invalid or masked take `old` = 0
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row = row total
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = wave total
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// S16 = false: 8-bit samples, v_sad_u8 (4 per dword).  S16 = true: 9..16-bit samples stored as
// uint16 (ffmpeg's ff_scene_sad16_c), v_sad_u16 (2 per dword).
template <bool S16>
__device__ __forceinline__ uint32_t sad_dword(uint32_t a, uint32_t b, uint32_t acc) {
    if constexpr (S16) return __builtin_amdgcn_sad_u16(a, b, acc);
    else return __builtin_amdgcn_sad_u8(a, b, acc);
}

template <bool S16>
__device__ __forceinline__ uint32_t sad16(const uint4 &a, const uint4 &b, uint32_t acc) {
    acc = sad_dword<S16>(a.x, b.x, acc);
    acc = sad_dword<S16>(a.y, b.y, acc);
    acc = sad_dword<S16>(a.z, b.z, acc);
    acc = sad_dword<S16>(a.w, b.w, acc);
    return acc;
}

// Park per-frame totals in lanes and flush 64 of them with one coalesced store.
struct Stash {
    uint32_t v = 0;
    __device__ __forceinline__ void put(int lane, int64_t t, uint32_t total) {
        if (lane == (int)(t & 63)) v = total;
    }
    // call after frame t was put; flushes when the 64-slot group is full or t is the last
    __device__ __forceinline__ void maybe_flush(int lane, int64_t t, int64_t t_lo, int64_t t_last,
                                                uint32_t *__restrict__ row) {
        if ((t & 63) == 63 || t == t_last) {
            const int64_t tb = t & ~(int64_t)63;
            const int64_t lt = tb + lane;
            if (lt >= t_lo && lt <= t) row[lt] = v;
        }
    }
};

template <bool NT>
__device__ __forceinline__ uint4 load16(const uint4 *p) {
    if constexpr (NT) {
        // streamed once: non-temporal hint keeps the frames from displacing L2/MALL lines
        using v4 = __attribute__((ext_vector_type(4))) unsigned int;
        const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}

template <int U, bool NT, bool S16>
__global__ __launch_bounds__(kBlock) void luma_sad_flat_kernel(
    const uint8_t *__restrict__ luma, StateArgs sa, int64_t T, int64_t frame_stride, int64_t n16,
    int32_t n_strips, int32_t tc, uint32_t *__restrict__ partial, int64_t Tpad) {
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (strip >= n_strips) return;  // wave-uniform
    const Carry cy = read_carry(sa);
    int64_t t0 = (int64_t)blockIdx.y * tc;
    const int64_t t1 = (t0 + tc < T) ? t0 + tc : T;
    if (t0 < cy.t_first) t0 = cy.t_first;
    const bool keep_tail = cy.tail != nullptr && t1 == T;      // this chunk ends the batch

    int64_t idx[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = ((int64_t)strip * U + u) * kWave + lane;
        ok[u] = i < n16;
        idx[u] = ok[u] ? i : 0;
    }
    uint4 prev[U], cur[U], nxt[U];
    if (t0 >= t1) {
        // nothing to score (a one-frame batch that starts a stream): only the tail frame moves
        if (keep_tail) {
            const uint4 *pl = reinterpret_cast<const uint4 *>(luma + (T - 1) * frame_stride);
            uint4 *pt = reinterpret_cast<uint4 *>(cy.tail);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) pt[idx[u]] = load16<NT>(pl + idx[u]);
        }
        return;
    }
    const uint4 *pp = (t0 == 0) ? reinterpret_cast<const uint4 *>(cy.prev0)
                                : reinterpret_cast<const uint4 *>(luma + (t0 - 1) * frame_stride);
    const uint4 *pc = reinterpret_cast<const uint4 *>(luma + t0 * frame_stride);
#pragma unroll
    for (int u = 0; u < U; ++u) prev[u] = load16<NT>(pp + idx[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = load16<NT>(pc + idx[u]);

    uint32_t *row = partial + (int64_t)strip * Tpad;
    Stash stash;
    auto score = [&](int64_t t) {
        uint32_t acc = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t a = sad16<S16>(cur[u], prev[u], acc);
            acc = ok[u] ? a : acc;
        }
        const uint32_t total = wave_sum_u32(acc);
        stash.put(lane, t, total);
        stash.maybe_flush(lane, t, t0, t1 - 1, row);
    };
    // steady state: the next frame's strip is always in flight while this one is reduced
    // (no branch around the loads, so the compiler waits with a counted vmcnt, not vmcnt(0))
    for (int64_t t = t0; t < t1 - 1; ++t) {
        const uint4 *pn = reinterpret_cast<const uint4 *>(luma + (t + 1) * frame_stride);
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = load16<NT>(pn + idx[u]);
        score(t);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            prev[u] = cur[u];
            cur[u] = nxt[u];
        }
    }
    score(t1 - 1);
    if (keep_tail) {            // cur[] is the strip of frame T-1: the next batch's predecessor
        uint4 *pt = reinterpret_cast<uint4 *>(cy.tail);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (ok[u]) pt[idx[u]] = cur[u];
    }
}

// 4-byte granules, arbitrary row stride / alignment.  granule g -> row g / gpr, x = 4*(g % gpr)
__device__ __forceinline__ uint32_t load_granule(const uint8_t *p, int nb) {
    uint32_t v = 0;
    if (nb == 4) {
        __builtin_memcpy(&v, p, 4);
    } else {
        for (int i = 0; i < nb; ++i) v |= (uint32_t)p[i] << (8 * i);
    }
    return v;
}

__device__ __forceinline__ void store_granule(uint8_t *p, uint32_t v, int nb) {
    if (nb == 4) {
        __builtin_memcpy(p, &v, 4);
    } else {
        for (int i = 0; i < nb; ++i) p[i] = (uint8_t)(v >> (8 * i));
    }
}

template <int U, bool S16>
__global__ __launch_bounds__(kBlock) void luma_sad_generic_kernel(
    const uint8_t *__restrict__ luma, StateArgs sa, int64_t T, int64_t frame_stride,
    int64_t row_stride, int32_t H, int32_t W /* row BYTES */, int32_t gpr, int64_t n_gran,
    int32_t n_strips, int32_t tc, uint32_t *__restrict__ partial, int64_t Tpad) {
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (strip >= n_strips) return;
    const Carry cy = read_carry(sa);
    int64_t t0 = (int64_t)blockIdx.y * tc;
    const int64_t t1 = (t0 + tc < T) ? t0 + tc : T;
    if (t0 < cy.t_first) t0 = cy.t_first;
    const bool keep_tail = cy.tail != nullptr && t1 == T;

    int64_t off[U], off0[U];
    int nb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t g = ((int64_t)strip * U + u) * kWave + lane;
        if (g < n_gran) {
            const int64_t y = g / gpr;
            const int32_t x = (int32_t)(g - y * gpr) * 4;
            nb[u] = (W - x < 4) ? (W - x) : 4;
            off[u] = y * row_stride + x;
            off0[u] = y * (int64_t)W + x;  // the state's frame buffers are tightly packed
        } else {
            nb[u] = 0;
            off[u] = 0;
            off0[u] = 0;
        }
    }
    uint32_t prev[U], cur[U];
    if (t0 >= t1) {
        if (keep_tail) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                store_granule(cy.tail + off0[u], load_granule(luma + (T - 1) * frame_stride + off[u], nb[u]), nb[u]);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        prev[u] = (t0 == 0) ? load_granule(cy.prev0 + off0[u], nb[u])
                            : load_granule(luma + (t0 - 1) * frame_stride + off[u], nb[u]);
    uint32_t *row = partial + (int64_t)strip * Tpad;
    Stash stash;
    for (int64_t t = t0; t < t1; ++t) {
        const uint8_t *pc = luma + t * frame_stride;
        uint32_t acc = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cur[u] = load_granule(pc + off[u], nb[u]);
            acc = sad_dword<S16>(cur[u], prev[u], acc);
            prev[u] = cur[u];
        }
        const uint32_t total = wave_sum_u32(acc);
        stash.put(lane, t, total);
        stash.maybe_flush(lane, t, t0, t1 - 1, row);
    }
    if (keep_tail) {
#pragma unroll
        for (int u = 0; u < U; ++u) store_granule(cy.tail + off0[u], prev[u], nb[u]);
    }
}

// ---- get_scene_score epilogue (f_select.c): same operation order, true divisions ----
struct SelectParams {
    double count;      // (double)(W*H)
    double depth_div;  // (double)(1ULL << (bitdepth-8))
    double threshold;
};

__device__ __forceinline__ double mafd_of(uint64_t sad, const SelectParams &p) {
    return (double)sad / p.count / p.depth_div;
}

__device__ __forceinline__ void scene_epilogue(uint64_t sad, bool first, double prev_mafd,
                                               const SelectParams &p, double &mafd, double &score,
                                               uint8_t &sel) {
    if (first) {
        mafd = 0.0;
        score = 0.0;
        sel = 0;
        return;
    }
    mafd = mafd_of(sad, p);
    const double diff = fabs(mafd - prev_mafd);
    const double m = (mafd > diff) ? diff : mafd;  // FFMIN
    float f = (float)(m / 100.);
    f = (f < 0.0f) ? 0.0f : ((f > 1.0f) ? 1.0f : f);  // av_clipf
    score = (double)f;
    sel = (score > p.threshold) ? 1 : 0;
}

constexpr int kFinT = 64;   // frames per finalize block
constexpr int kFinG = 16;   // strip groups per finalize block

// sum partial[s][t] over strips, then (optionally) the epilogue.  The first scored frame and the
// predecessor's mafd come from the device-resident state (or: frame 1, 0.0 without one).
__global__ __launch_bounds__(kFinT *kFinG) void scene_finalize_kernel(
    const uint32_t *__restrict__ partial, int32_t n_strips, int64_t Tpad, int64_t T,
    const StateHdr *__restrict__ st, SelectParams sp, uint64_t *__restrict__ sad_out,
    double *__restrict__ mafd_out, double *__restrict__ score_out, uint8_t *__restrict__ sel_out,
    WsHdr *__restrict__ wh) {
    __shared__ uint64_t red[kFinG][kFinT + 1];
    int32_t t_first = 1;
    double prev_mafd_in = 0.0;
    if (st && st->have_prev) {
        t_first = 0;
        prev_mafd_in = st->prev_mafd;
    }
    const int tl = threadIdx.x & (kFinT - 1);
    const int sg = threadIdx.x / kFinT;
    const int64_t bt = (int64_t)blockIdx.x * kFinT;
    const int64_t t = bt + tl;
    uint64_t acc = 0, halo = 0;
    if (t < T && t >= t_first)
        for (int s = sg; s < n_strips; s += kFinG) acc += partial[(int64_t)s * Tpad + t];
    if (tl == 0 && bt - 1 >= t_first)
        for (int s = sg; s < n_strips; s += kFinG) halo += partial[(int64_t)s * Tpad + bt - 1];
    red[sg][tl] = acc;
    if (tl == 0) red[sg][kFinT] = halo;
    __syncthreads();
    if (threadIdx.x <= kFinT) {
        uint64_t s = 0;
#pragma unroll
        for (int g = 0; g < kFinG; ++g) s += red[g][threadIdx.x];
        red[0][threadIdx.x] = s;
    }
    __syncthreads();
    if (sg == 0 && t < T) {
        const uint64_t sad = red[0][tl];
        if (sad_out) sad_out[t] = sad;
        const bool first = t < t_first;
        double prev_mafd = prev_mafd_in;
        if (t - 1 >= t_first) prev_mafd = mafd_of(tl ? red[0][tl - 1] : red[0][kFinT], sp);
        double mafd, score;
        uint8_t sel;
        scene_epilogue(sad, first, prev_mafd, sp, mafd, score, sel);
        if (mafd_out) mafd_out[t] = mafd;
        if (score_out) score_out[t] = score;
        if (sel_out) sel_out[t] = sel;
        // an unscored first frame reports mafd 0 == ffmpeg's zero-initialised prev_mafd
        if (t == T - 1) wh->last_mafd = mafd;
    }
}

__global__ __launch_bounds__(256) void scene_select_kernel(const uint64_t *__restrict__ sad,
                                                           int64_t T,
                                                           const double *__restrict__ d_prev_mafd,
                                                           SelectParams sp,
                                                           uint8_t *__restrict__ sel_out,
                                                           double *__restrict__ score_out,
                                                           double *__restrict__ mafd_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int64_t t_first = d_prev_mafd ? 0 : 1;
    double prev_mafd = d_prev_mafd ? *d_prev_mafd : 0.0;
    if (t - 1 >= t_first) prev_mafd = mafd_of(sad[t - 1], sp);
    double mafd, score;
    uint8_t sel;
    scene_epilogue(sad[t], t < t_first, prev_mafd, sp, mafd, score, sel);
    sel_out[t] = sel;
    if (score_out) score_out[t] = score;
    if (mafd_out) mafd_out[t] = mafd;
}

// One block: ascending indices of the selected frames -> cuts[1..], their number -> cuts[0];
// then the stream state moves to the end of this batch (runs after the SAD + finalize kernels
// of the batch and before those of the next one: stream order is the only synchronisation).
constexpr int kTailBlock = 1024;
__global__ __launch_bounds__(kTailBlock) void scene_tail_kernel(
    const uint8_t *__restrict__ sel, int64_t T, int32_t *__restrict__ cuts, int32_t cuts_cap,
    StateHdr *__restrict__ st, const WsHdr *__restrict__ wh) {
    __shared__ int32_t wsum[kTailBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (cuts) {
        // ONE pass: thread i owns the frames [i*c, (i+1)*c); count, block-wide exclusive scan of the
        // 1024 counts (DPP-free: ballot-less integer scan over LDS per wave, then over the 16 waves),
        // then every thread writes its own frames' indices - ascending by construction
        const int64_t c = (T + kTailBlock - 1) / kTailBlock;
        const int64_t t0 = (int64_t)threadIdx.x * c;
        const int64_t t1 = t0 + c < T ? t0 + c : T;
        int mine = 0;
        for (int64_t t = t0; t < t1; ++t) mine += sel[t] != 0;
        // inclusive scan within the wave
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < kTailBlock / 64; ++w) {
            const int v = wsum[w];
            if (w < wave) base += v;
            total += v;
        }
        int pos = base + incl - mine;
        for (int64_t t = t0; t < t1; ++t)
            if (sel[t] != 0) {
                if (pos < cuts_cap) cuts[1 + pos] = (int32_t)t;
                ++pos;
            }
        if (threadIdx.x == 0) cuts[0] = total;
    }
    if (st && threadIdx.x == 0) {
        st->prev_mafd = wh->last_mafd;
        st->have_prev = 1;
        st->parity ^= 1;
        st->frames_seen += T;
    }
}

__global__ void scene_state_reset_kernel(StateHdr *st) {
    st->prev_mafd = 0.0;
    st->have_prev = 0;
    st->parity = 0;
    st->frames_seen = 0;
}

// ---------------------------------------------------------------- host side
struct Plan {
    bool flat;
    int U;
    int tc;
    int64_t n16;      // flat: 16-byte chunks per frame
    int64_t n_gran;   // generic: 4-byte granules per frame
    int32_t gpr;
    int32_t n_strips;
    int64_t Tpad;
};

constexpr int kGenericU = 8;

bool flat_ok(const void *p, int64_t fs, int64_t rs, int32_t H, int32_t W, int bps) {
    return rs == (int64_t)W * bps && ((int64_t)H * W * bps) % 16 == 0 && fs % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(p) % 16) == 0;
}

// Shape choice for the flat kernel, from the sweeps in profiles/r1_microbatch_*.txt: the widest
// strip (U=8, 8 KiB per wave and frame) wins at every batch size and resolution; the time chunk
// is sized so that the launch has about 1,000 waves (4 per CU) - fewer, longer serial walks beat
// more, shorter ones until the halo re-read (1/tc) stops mattering - and is capped at 256 frames.
void auto_shape(int64_t T, int64_t n16, int &U, int &tc) {
    U = 8;
    const int64_t strips = tvz::ceil_div(n16, (int64_t)kWave * U);
    const int64_t chunks = std::max<int64_t>(1, 1024 / std::max<int64_t>(1, strips));
    const int64_t want = tvz::ceil_div(T, chunks);
    tc = 8;
    while (tc < 256 && tc < want) tc *= 2;
}

int decode_shape(uint32_t shape, Tuning &tn) {
    tn = Tuning{};
    if (shape == TVZ_SHAPE_AUTO) return TVZ_OK;
    tn.nt = (shape & TVZ_SHAPE_NO_NT) ? 0 : 1;
    tn.U = (int)(shape & 0xffu);
    tn.tc = (int)((shape >> 8) & 0xfffffu);
    TVZ_REQUIRE(tn.U == 0 || tn.U == 1 || tn.U == 2 || tn.U == 4 || tn.U == 8,
                "shape: U must be 0 (auto), 1, 2, 4 or 8");
    TVZ_REQUIRE(tn.tc == 0 || tn.tc == 8 || tn.tc == 16 || tn.tc == 32 || tn.tc % 64 == 0,
                "shape: tc must be 0 (auto), 8, 16, 32 or a multiple of 64");
    return TVZ_OK;
}

Plan make_plan(bool flat, int64_t T, int32_t H, int32_t W, int bps, const Tuning &tn) {
    Plan p{};
    p.flat = flat;
    p.tc = tn.tc ? tn.tc : 128;
    p.Tpad = tvz::round_up(T > 0 ? T : 1, 64);
    if (flat) {
        p.n16 = (int64_t)H * W * bps / 16;
        int aU = 4, atc = 128;
        auto_shape(T > 0 ? T : 1, p.n16, aU, atc);
        p.U = tn.U ? tn.U : aU;
        p.tc = tn.tc ? tn.tc : atc;
        p.n_strips = (int32_t)tvz::ceil_div(p.n16, (int64_t)kWave * p.U);
    } else {
        p.U = kGenericU;
        p.gpr = (W * bps + 3) / 4;
        p.n_gran = (int64_t)H * p.gpr;
        p.n_strips = (int32_t)tvz::ceil_div(p.n_gran, (int64_t)kWave * p.U);
    }
    return p;
}

size_t plan_bytes(const Plan &p) { return (size_t)p.n_strips * (size_t)p.Tpad * sizeof(uint32_t); }

int check_dims(int64_t T, int32_t H, int32_t W) {
    TVZ_REQUIRE(T >= 0, "T must be >= 0 (got %lld)", (long long)T);
    TVZ_REQUIRE(H > 0 && W > 0, "H and W must be positive (got %d x %d)", H, W);
    if ((int64_t)H * W > (int64_t)1 << 31)
        return tvz::fail(TVZ_ERR_UNSUPPORTED, "frame of %d x %d exceeds 2^31 pixels", H, W);
    return TVZ_OK;
}

size_t state_frame_bytes(int32_t H, int32_t W, int bps) {
    return (size_t)tvz::round_up((int64_t)H * W * bps, 256);
}

int launch_sad(const uint8_t *d_luma, const StateArgs &sa, int64_t T, int32_t H, int32_t W,
               int64_t fs, int64_t rs, const Plan &p, uint32_t *partial, int bps, int nt,
               hipStream_t st) {
    dim3 grid((unsigned)tvz::ceil_div(p.n_strips, kWavesPerBlock), (unsigned)tvz::ceil_div(T, p.tc));
    if (grid.y > 65535u)
        return tvz::fail(TVZ_ERR_UNSUPPORTED, "batch of %lld frames needs more than 65535 time chunks",
                         (long long)T);
    if (p.flat) {
#define TVZ_FLAT_(UU, NTV, S16V)                                                               \
    hipLaunchKernelGGL((luma_sad_flat_kernel<UU, NTV, S16V>), grid, dim3(kBlock), 0, st, d_luma,   \
                       sa, T, fs, p.n16, p.n_strips, p.tc, partial, p.Tpad)
#define TVZ_FLAT(UU)                                                                           \
    do {                                                                                       \
        if (bps == 2) { if (nt) TVZ_FLAT_(UU, true, true); else TVZ_FLAT_(UU, false, true); }   \
        else          { if (nt) TVZ_FLAT_(UU, true, false); else TVZ_FLAT_(UU, false, false); } \
    } while (0)
        switch (p.U) {
            case 1: TVZ_FLAT(1); break;
            case 2: TVZ_FLAT(2); break;
            case 4: TVZ_FLAT(4); break;
            case 8: TVZ_FLAT(8); break;
            default: return tvz::fail(TVZ_ERR_INVALID, "unsupported U=%d", p.U);
        }
#undef TVZ_FLAT
#undef TVZ_FLAT_
    } else {
        if (bps == 2)
            hipLaunchKernelGGL((luma_sad_generic_kernel<kGenericU, true>), grid, dim3(kBlock), 0, st,
                               d_luma, sa, T, fs, rs, H, W * 2, p.gpr, p.n_gran, p.n_strips,
                               p.tc, partial, p.Tpad);
        else
            hipLaunchKernelGGL((luma_sad_generic_kernel<kGenericU, false>), grid, dim3(kBlock), 0, st,
                               d_luma, sa, T, fs, rs, H, W, p.gpr, p.n_gran, p.n_strips, p.tc,
                               partial, p.Tpad);
    }
    TVZ_HIP(hipGetLastError());
    return TVZ_OK;
}

SelectParams make_sp(int32_t H, int32_t W, int32_t bitdepth, double thr) {
    SelectParams sp;
    sp.count = (double)((uint64_t)W * (uint64_t)H);
    sp.depth_div = (double)(1ULL << (bitdepth - 8));
    sp.threshold = thr;
    return sp;
}

}  // namespace

TVZ_EXPORT size_t tvz_scene_workspace_bytes(int64_t T, int32_t H, int32_t W) {
    if (T < 0 || H <= 0 || W <= 0) return 0;
    // worst case over every shape the launcher may pick: U=1 strips on the flat path
    // (16-bit samples double the plane: sized for them so one workspace serves both depths)
    const size_t a = (size_t)tvz::ceil_div((int64_t)H * W * 2 / 16 + 1, kWave) * (size_t)tvz::round_up(T > 0 ? T : 1, 64) * sizeof(uint32_t);
    const size_t b = plan_bytes(make_plan(false, T, H, W, 2, Tuning{}));
    return (a > b ? a : b) + kWsHdrBytes + 256;
}

TVZ_EXPORT size_t tvz_scene_state_bytes(int32_t H, int32_t W, int32_t bytes_per_sample) {
    if (H <= 0 || W <= 0 || (bytes_per_sample != 1 && bytes_per_sample != 2)) return 0;
    return kStateHdrBytes + 2 * state_frame_bytes(H, W, bytes_per_sample);
}

TVZ_EXPORT int tvz_scene_state_reset(void *d_state, void *hip_stream) {
    TVZ_REQUIRE(d_state != nullptr, "d_state is NULL");
    TVZ_REQUIRE(reinterpret_cast<uintptr_t>(d_state) % 256 == 0, "d_state must be 256-byte aligned");
    hipLaunchKernelGGL(scene_state_reset_kernel, dim3(1), dim3(1), 0,
                       reinterpret_cast<hipStream_t>(hip_stream), reinterpret_cast<StateHdr *>(d_state));
    TVZ_HIP(hipGetLastError());
    return TVZ_OK;
}

namespace {
int scene_scores_impl(const uint8_t *d_luma, int bps, int64_t T, int32_t H, int32_t W,
                      int64_t frame_stride_bytes, int64_t row_stride_bytes, void *d_state,
                      int32_t bitdepth, double threshold, uint64_t *d_sad_out, double *d_mafd,
                      double *d_score, uint8_t *d_selected, int32_t *d_cuts, int32_t cuts_cap,
                      void *d_workspace, size_t workspace_bytes, uint32_t shape, void *hip_stream) {
    if (int rc = check_dims(T, H, W)) return rc;
    if (T == 0) return TVZ_OK;
    TVZ_REQUIRE(d_luma != nullptr, "d_luma is NULL");
    TVZ_REQUIRE(row_stride_bytes >= (int64_t)W * bps, "row stride %lld < %d bytes",
                (long long)row_stride_bytes, W * bps);
    TVZ_REQUIRE(frame_stride_bytes >= (int64_t)(H - 1) * row_stride_bytes + (int64_t)W * bps || T == 1,
                "frame stride %lld smaller than a plane", (long long)frame_stride_bytes);
    if (bps == 1) TVZ_REQUIRE(bitdepth == 8, "8-bit entry point called with bitdepth=%d", bitdepth);
    else {
        TVZ_REQUIRE(bitdepth > 8 && bitdepth <= 16, "16-bit entry point needs 9 <= bitdepth <= 16 (got %d)", bitdepth);
        TVZ_REQUIRE(row_stride_bytes % 2 == 0 && frame_stride_bytes % 2 == 0 &&
                        reinterpret_cast<uintptr_t>(d_luma) % 2 == 0,
                    "16-bit planes must be 2-byte aligned");
    }
    TVZ_REQUIRE(d_sad_out || d_selected || d_score || d_mafd || d_state,
                "nothing to compute: every output is NULL");
    TVZ_REQUIRE(d_cuts == nullptr || (d_selected != nullptr && cuts_cap >= 0),
                "d_cuts needs d_selected and cuts_cap >= 0");
    TVZ_REQUIRE(d_workspace != nullptr, "d_workspace is NULL");
    TVZ_REQUIRE(d_state == nullptr || reinterpret_cast<uintptr_t>(d_state) % 256 == 0,
                "d_state must be 256-byte aligned");
    Tuning tn;
    if (int rc = decode_shape(shape, tn)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const bool flat = flat_ok(d_luma, frame_stride_bytes, row_stride_bytes, H, W, bps);
    const Plan p = make_plan(flat, T, H, W, bps, tn);
    uintptr_t ws = (reinterpret_cast<uintptr_t>(d_workspace) + 255) & ~(uintptr_t)255;
    const size_t lost = ws - reinterpret_cast<uintptr_t>(d_workspace);
    if (workspace_bytes < lost + kWsHdrBytes + plan_bytes(p))
        return tvz::fail(TVZ_ERR_WORKSPACE, "workspace of %zu bytes, need %zu", workspace_bytes,
                         lost + kWsHdrBytes + plan_bytes(p));
    WsHdr *wh = reinterpret_cast<WsHdr *>(ws);
    uint32_t *partial = reinterpret_cast<uint32_t *>(ws + kWsHdrBytes);
    StateArgs sa{nullptr, nullptr, nullptr};
    if (d_state) {
        sa.hdr = reinterpret_cast<StateHdr *>(d_state);
        sa.buf0 = reinterpret_cast<uint8_t *>(d_state) + kStateHdrBytes;
        sa.buf1 = sa.buf0 + state_frame_bytes(H, W, bps);
    }
    if (int rc = launch_sad(d_luma, sa, T, H, W, frame_stride_bytes, row_stride_bytes, p, partial,
                            bps, tn.nt, st))
        return rc;
    const SelectParams sp = make_sp(H, W, bitdepth, threshold);
    hipLaunchKernelGGL(scene_finalize_kernel, dim3((unsigned)tvz::ceil_div(T, kFinT)),
                       dim3(kFinT * kFinG), 0, st, partial, p.n_strips, p.Tpad, T, sa.hdr, sp,
                       d_sad_out, d_mafd, d_score, d_selected, wh);
    TVZ_HIP(hipGetLastError());
    if (d_cuts || d_state) {
        hipLaunchKernelGGL(scene_tail_kernel, dim3(1), dim3(kTailBlock), 0, st, d_selected, T, d_cuts,
                           cuts_cap, sa.hdr, wh);
        TVZ_HIP(hipGetLastError());
    }
    return TVZ_OK;
}
}  // namespace

TVZ_EXPORT int tvz_scene_scores_u8(const uint8_t *d_luma, int64_t T, int32_t H, int32_t W,
                                   int64_t frame_stride_bytes, int64_t row_stride_bytes,
                                   void *d_state, int32_t bitdepth, double threshold,
                                   uint64_t *d_sad_out, double *d_mafd, double *d_score,
                                   uint8_t *d_selected, int32_t *d_cuts, int32_t cuts_cap,
                                   void *d_workspace, size_t workspace_bytes, uint32_t shape,
                                   void *hip_stream) {
    return scene_scores_impl(d_luma, 1, T, H, W, frame_stride_bytes, row_stride_bytes, d_state,
                             bitdepth, threshold, d_sad_out, d_mafd, d_score, d_selected, d_cuts,
                             cuts_cap, d_workspace, workspace_bytes, shape, hip_stream);
}

TVZ_EXPORT int tvz_scene_scores_u16(const uint16_t *d_luma, int64_t T, int32_t H, int32_t W,
                                    int64_t frame_stride_bytes, int64_t row_stride_bytes,
                                    void *d_state, int32_t bitdepth, double threshold,
                                    uint64_t *d_sad_out, double *d_mafd, double *d_score,
                                    uint8_t *d_selected, int32_t *d_cuts, int32_t cuts_cap,
                                    void *d_workspace, size_t workspace_bytes, uint32_t shape,
                                    void *hip_stream) {
    return scene_scores_impl(reinterpret_cast<const uint8_t *>(d_luma), 2, T, H, W,
                             frame_stride_bytes, row_stride_bytes, d_state, bitdepth, threshold,
                             d_sad_out, d_mafd, d_score, d_selected, d_cuts, cuts_cap, d_workspace,
                             workspace_bytes, shape, hip_stream);
}

TVZ_EXPORT int tvz_luma_sad_u8(const uint8_t *d_luma, int64_t T, int32_t H, int32_t W,
                               int64_t frame_stride_bytes, int64_t row_stride_bytes,
                               uint64_t *d_sad_out, void *d_workspace, size_t workspace_bytes,
                               void *hip_stream) {
    TVZ_REQUIRE(d_sad_out != nullptr || T == 0, "d_sad_out is NULL");
    return tvz_scene_scores_u8(d_luma, T, H, W, frame_stride_bytes, row_stride_bytes, nullptr, 8,
                               0.0, d_sad_out, nullptr, nullptr, nullptr, nullptr, 0, d_workspace,
                               workspace_bytes, TVZ_SHAPE_AUTO, hip_stream);
}

TVZ_EXPORT int tvz_scene_select(const uint64_t *d_sad, int64_t T, int32_t H, int32_t W,
                                int32_t bitdepth, double threshold, const double *d_prev_mafd,
                                uint8_t *d_selected, double *d_score, double *d_mafd,
                                void *hip_stream) {
    if (int rc = check_dims(T, H, W)) return rc;
    if (T == 0) return TVZ_OK;
    TVZ_REQUIRE(d_sad && d_selected, "d_sad / d_selected is NULL");
    TVZ_REQUIRE(bitdepth >= 8 && bitdepth <= 16, "bitdepth %d out of range", bitdepth);
    const SelectParams sp = make_sp(H, W, bitdepth, threshold);
    hipLaunchKernelGGL(scene_select_kernel, dim3((unsigned)tvz::ceil_div(T, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(hip_stream), d_sad, T, d_prev_mafd, sp,
                       d_selected, d_score, d_mafd);
    TVZ_HIP(hipGetLastError());
    return TVZ_OK;
}
