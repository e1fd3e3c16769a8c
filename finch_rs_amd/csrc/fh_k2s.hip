// fh_k2s.hip -- the SEGMENT form of the sketch kernel (K = 1..32), hand-written for gfx950 (CDNA4, wave64).
//
// What it replaces is what fh_k2.hip's k2_sketch<K> replaces (mash.rs:67-80, 34-42; hashing.rs:10-12), bit for bit; what it
// adds is the reference's own shape of the work: needletail's canonical_kmers yields len - k + 1 windows per RECORD
// (mash.rs:76), while k2_sketch hashes one window per stream POSITION and throws away, on the admit path, the k of every
// record's len + 1 positions whose window crosses the record's breaker byte -- 21 of 151 at k = 21, 31 of 151 at k = 31.
//
// Here a lane owns one segment of S = SketchArgs::seg_stride consecutive start positions (S = read length + 1 when the host
// knows or finds that the records are of one length), a wave a tile of 64 segments -- or, for records longer than a lane's
// share of the LDS holds (strides 169..672: 2 x 250 and 2 x 300 reads), TWO or FOUR lanes own a record between them
// (SketchArgs::seg_sub, fh_device.h) and a tile is 32 or 16 records:
//   * phase A: the wave loads the tile's 64 S + 96 bytes coalesced (16 bytes a lane a time, all loads in flight together),
//     classifies them (fh_core.h classify_chunk) and leaves THREE tile-wide strings in its own LDS: the complemented 2-bit codes
//     (a window's reverse complement is a bit field of it), the digit-reversed codes (the forward strand's windows) and the
//     good bits;
//   * a lane's 64-base view at segment offset R c (R = 32 or 16 positions per round, as in k2_sketch) is cut out of those
//     strings with funnel shifts by a per-lane constant -- from there on the round is k2_sketch's: windows at compile-time
//     offsets, table lookups, murmur3, the high-word reject, the admit queue;
//   * a round none of whose windows is valid in ANY lane is skipped, and a round ends behind the last position that is valid
//     in any lane (only rounds that reach a segment's last K bases ask: one wave-wide OR).  With S = record stride that is
//     every record's tail; with any other S, or on a stream that is not made of equal records, nothing is skipped that
//     k2_sketch would have admitted: the sketch never depends on S (tests/test_gpu_segments.py).
// One workgroup of sixteen waves per CU (a wave's strings are 6.7 KB); a wave that uses up its insert budget stops at the
// end of a ROUND and hands the rest of its range back as ONE leftover entry (first tile, end of the range, round to resume the
// first tile at), so the table's guard is k2_sketch's with a round in place of a tile: budget + at most 64 x 48 new hashes of the
// round that crosses it + the candidates parked before it (WAVE_OVERSHOOT, fh_device.h) per wave and launch.
//
// Compiled FH_NPARTS times (-DFH_PART=i) like fh_k2.hip.
#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"
#include "fh_k2_common.h"

#include <type_traits>

#ifndef FH_PART
#error "compile with -DFH_PART=<0..FH_NPARTS-1>"
#endif

namespace fh {

#ifndef FH_SEG_WHOLE_TILES
#define FH_SEG_WHOLE_TILES 1 // 0: every chunk through the guarded load (A/B)
#endif
constexpr int K2S_WPB = 16;
// Positions per round.  K >= 23: 16, as k2_sketch (fh_k2.hip, k2_round: the register budget is the same loop's).  K <= 22: as many
// as the lane's 64-base view holds windows, 65 - K (at most 48): a round's set-up -- validity mask, the wave's question, the two
// views, bounds: ~60 instructions, and a window and its lookups that run ahead of the last position -- is paid three times per
// 150-base read at k = 21 (44 + 44 + 42 windows) instead of five times (32 + 32 + 32 + 32 + 2).  FH_SEG_LONG=0: rounds of 32.
// (fh_core.h: seg_round, seg_long, seg_doff and the cut of the views, shared with the host logic test)
constexpr bool k2s_long(int K) { return seg_long(K); }
constexpr int k2s_round(int K) { return seg_round(K); }
static_assert(64 * 48 + QCAP <= WAVE_OVERSHOOT, "a round's positions are what a wave may overshoot its insert budget by (fh_device.h)");
// LDS of the workgroup: lookup tables, admit queues, then a block per wave
constexpr u32 K2S_A1 = 0, K2S_A2 = 4096, K2S_B1 = 8192, K2S_B2 = 10240, K2S_P = 12288, K2S_Q = 20480;
constexpr u32 K2S_TILE = K2S_Q + K2S_WPB * (u32)sizeof(AdmitQueueT<false>);
constexpr u32 K2S_NCH_MAX = 4 * SEG_MAX_STRIDE + 6; // 16-byte chunks of a tile with its 96-byte halo
constexpr u32 K2S_FC_DW = K2S_NCH_MAX + 3, K2S_RV_DW = K2S_NCH_MAX + 2, K2S_G_DW = K2S_NCH_MAX / 2 + 3;
constexpr u32 K2S_WAVE_DW = (K2S_FC_DW + K2S_RV_DW + K2S_G_DW + 3) / 4 * 4;
constexpr u32 K2S_BYTES = K2S_TILE + K2S_WPB * 4 * K2S_WAVE_DW;
static_assert(K2S_BYTES <= 160 * 1024, "one workgroup's LDS");
constexpr int K2S_MAX_LOADS = (K2S_NCH_MAX + 63) / 64;

// OR over the wave, in every lane's... lane 63's register, read out as a scalar (row-wise prefix OR, then the rows joined)
__device__ __forceinline__ u32 wave_or(u32 x) {
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); // row_shr:1
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true); // row_shr:2
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true); // row_shr:4
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true); // row_shr:8
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return (u32)__builtin_amdgcn_readlane((int)x, 63);
}

// max over the wave (same walk)
__device__ __forceinline__ u32 wave_max(u32 x) {
    auto mx = [](u32 a, u32 b) { return a > b ? a : b; };
    x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = mx(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return (u32)__builtin_amdgcn_readlane((int)x, 63);
}

__device__ __forceinline__ u32 lane_now() { // (recomputed where it is needed: a value kept across the position loop is a register the loop lacks)
    u32 l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <int K, bool SEED0>
__global__ __launch_bounds__(64 * K2S_WPB, 4) void k2_sketch_seg(const SketchArgs a) {
    constexpr int WPB = K2S_WPB, NTHR = 64 * WPB, R = k2s_round(K), PRE = pre_shift(K);
    constexpr bool LONG = k2s_long(K);
    constexpr int DOFF = seg_doff(K); // (fh_core.h, Windows: the forward string's view reaches this many bases further)
    using Win = Windows<K, DOFF>;
    using Mask = std::conditional_t<LONG, u64, u32>; // a round's valid windows
    // Rounds of 16 unrolled positions (the K whose registers allow no more: fh_core.h, seg_long) come in PAIRS: one set-up -- the
    // validity mask, the wave's question, the two views -- serves 32 positions, the strings are moved on 32 bits in between as
    // in k2_sketch (Windows::advance<16>).  RO = a round's positions as the tile's bookkeeping counts them.
#ifndef FH_SEG_HALVES
#define FH_SEG_HALVES 1
#endif
    constexpr int HALVES = (FH_SEG_HALVES && !LONG && R == 16) ? 2 : 1, RO = R * HALVES;
    static_assert(RO + K - 1 <= 64, "a round's windows lie inside the lane's 64-base view");
    __shared__ __attribute__((aligned(16))) unsigned char blob[K2S_BYTES];
    Rec4 *const sA1 = (Rec4 *)(blob + K2S_A1), *const sA2 = (Rec4 *)(blob + K2S_A2);
    Rec2 *const sB1 = (Rec2 *)(blob + K2S_B1), *const sB2 = (Rec2 *)(blob + K2S_B2), *const sP = (Rec2 *)(blob + K2S_P);
    static_assert(partial_entries(K) * sizeof(Rec2) <= K2S_Q - K2S_P, "the key's last-word table fits its slot");

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.gate && __hip_atomic_load(&a.ctl->spec_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    {
        if (has_pair_word(K, false))
            for (int q = tid; q < 256; q += NTHR) {
                sA1[q] = lut_rec_A((u32)q, false);
                sB1[q] = lut_rec_B((u32)q, 4, false);
            }
        if (has_pair_word(K, true))
            for (int q = tid; q < 256; q += NTHR) {
                sA2[q] = lut_rec_A((u32)q, true);
                sB2[q] = lut_rec_B((u32)q, 4, true);
            }
        for (int q = tid; q < partial_entries(K); q += NTHR) sP[q] = lut_rec_P<K>((u32)q);
        // the waves' blocks start out zero: the words around the strings are read (never written) as "no good base there"
        u32 *all = (u32 *)(blob + K2S_TILE);
        for (u32 i = (u32)tid; i < (u32)WPB * K2S_WAVE_DW; i += (u32)NTHR) all[i] = 0u;
    }
    const LutTables LT{sA1, sA2, sB1, sB2, sP, 0u};
    __syncthreads();

    auto load_tau = [&]() -> u64 {
        const u64 tau_v = __hip_atomic_load(&a.ctl->tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tau_v >> 32)) << 32) |
               (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)tau_v);
    };
    u64 tau = load_tau();
    u32 tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));

    const u32 gw = blockIdx.x * (u32)WPB + (u32)wave;
    u32 *const Fc = (u32 *)(blob + K2S_TILE) + (u32)wave * K2S_WAVE_DW; // ~codes: chunk i at word 1 + i
    u32 *const Rv = Fc + K2S_FC_DW;                                      // digit-reversed codes: chunk i at word NCH - 1 - i
    u32 *const Gd = Rv + K2S_RV_DW;                                      // good bits: chunk i at half-word i
    // A record of S start positions belongs to 1 << SH lanes: the first ones take H positions each, the last one the rest
    // (LAST >= H: it holds the K positions behind the record's last window).  SH = 0: a lane per record, H = LAST = S.
    const u32 S = (u32)__builtin_amdgcn_readfirstlane((int)a.seg_stride);
    // RAG: records of many lengths -- the lanes' cells become work items, 64 a round (fh_device.h, SEG_RAGGED); not for the K whose
    // rounds are longer than 32 positions
    const bool RAG = !LONG && RO == 32 && (u32)__builtin_amdgcn_readfirstlane((int)a.seg_sub) == SEG_RAGGED;
    const u32 SH = RAG ? 0u : (u32)__builtin_amdgcn_readfirstlane((int)(a.seg_sub >= 4u ? 2u : a.seg_sub >= 2u ? 1u : 0u));
    const u32 SUBM = (1u << SH) - 1u;
    const u32 H = SH ? (S - (u32)K + SUBM) >> SH : S, LAST = S - SUBM * H;
    const u32 tile_pos = (64u >> SH) * S;
    const u32 NCH = (tile_pos >> 4) + 6u, NR = (LAST + (u32)RO - 1u) / (u32)RO;
    // where lane l's segment begins in the tile, and how long it is
    auto seg_start = [&](u32 l) -> u32 { return S * (l >> SH) + (l & SUBM) * H; };
    // RAG: the tile's work items (start position | windows << 16), at most 256, in the room the strings of a 128-position stride
    // leave behind them (two halves: behind the complemented codes and behind the reversed ones; 8 words of zeroes stay between)
    u32 *const ItA = Fc + (NCH + 8u), *const ItB = Rv + (NCH + 8u);
    static_assert(4u * SEG_RAGGED_STRIDE + 6u + 8u + 128u <= K2S_RV_DW, "the item list fits behind the strings of a ragged tile");
    auto item_ptr = [&](u32 i) -> u32 * { return i < 128u ? ItA + i : ItB + (i - 128u); };
    u32 n_items = 0; // (wave-uniform)
    u32 nvalid = 0;

#define FLUSHS(ctl_, q_, qn_, shard_) ([&] { const u32 r_ = (u32)__builtin_amdgcn_readfirstlane((int)flush_queue(ctl_, q_, qn_, shard_)); want_refresh |= r_ >> 31; return r_ & 0x7FFFFFFFu; }())
    u32 want_refresh = 0, wave_inserts = 0, qn = 0;
    AdmitQueueT<false> *queue = (AdmitQueueT<false> *)(blob + K2S_Q) + wave;
    if ((tid & 63) == 0) {
        queue->tau = tau;
        queue->tau_lo = 0ull;
        queue->hash_mask = ~0ull;
        queue->pre = (u32)PRE;
    }
    const u32 shard = gw & (u32)(N_SHARDS - 1);
    u32 last_unit = 0;
    bool first_pull = true;
    for (;;) {
        // work distribution as in k2_sketch: leftover ranges of a stopped launch first (here triples: the range's first tile may
        // resume at a round), the wave's own first units, then guided pulls from the queue
        u32 rt0 = 0xFFFFFFFFu, rt1 = 0u, c0 = 0u; // tiles [rt0, rt1), the first of them from round c0 on
        if ((tid & 63) == 0) {
            u32 li = 0xFFFFFFFFu;
            if (a.n_left_in) li = atomicAdd(&a.ctl->left_in_pos, 1u);
            if (li < a.n_left_in) {
                rt0 = a.left_in[3u * li];
                rt1 = a.left_in[3u * li + 1u];
                c0 = a.left_in[3u * li + 2u];
            } else if (first_pull && a.first_units) {
                const u32 c = gw * a.first_units;
                if (c < a.n_units) {
                    rt0 = c * a.unit_tiles;
                    const u32 e = (c + a.first_units) * a.unit_tiles;
                    rt1 = e < a.tiles_total ? e : a.tiles_total;
                }
                last_unit = gridDim.x * (u32)WPB * a.first_units;
            } else if (a.static_only) {
            } else if (__hip_atomic_load(&a.ctl->stopped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                const u32 left = a.n_units > last_unit ? a.n_units - last_unit : 0u;
                u32 k = left / (4u * a.n_waves);
                k = k < 1u ? 1u : (k > a.max_units ? a.max_units : k);
                const u32 c = atomicAdd(&a.ctl->next_unit, k);
                last_unit = c + k;
                if (c < a.n_units) {
                    rt0 = c * a.unit_tiles;
                    const u32 e = (c + k) * a.unit_tiles;
                    rt1 = e < a.tiles_total ? e : a.tiles_total;
                }
            }
        }
        first_pull = false;
        rt0 = (u32)__builtin_amdgcn_readfirstlane((int)rt0);
        rt1 = (u32)__builtin_amdgcn_readfirstlane((int)rt1);
        if (rt0 == 0xFFFFFFFFu) break;
        u32 c_first = (u32)__builtin_amdgcn_readfirstlane((int)c0);

        bool stop = false;
#pragma unroll 1
        for (u32 t = rt0; t < rt1; ++t) {
            const u64 tile_pos0 = a.p_begin + (u64)t * tile_pos; // wave-uniform; a multiple of 16
            // ---- phase A: the tile's bytes (and 96 behind them) -> the three strings ----
            {
                // (the lane id is worked out anew for every tile: from a loop-invariant one the compiler derives a dozen
                // addresses once per kernel, keeps them across the positions' code in registers that code needs -- 21-30 spilled at
                // K = 29..32 -- and reloads them from scratch in front of every tile's loads)
                const u32 lane = lane_now();
                uint4 buf[K2S_MAX_LOADS];
                // A tile all of whose chunks lie inside the buffer (every tile but a block's last) needs no guard: one scalar base, the
                // lane's 32-bit byte offset, one add per chunk; the guarded form costs a dozen VALU instructions a chunk (64-bit
                // address, two 64-bit compares, the zeroes).  Both forms fill the same registers, chunk by chunk.
                u32 nload = (NCH + 63u) >> 6;
                asm volatile("" : "+s"(nload)); // (asked per tile: eleven conditions hoisted out of the loop are eleven SGPR pairs the kernel lacks)
                const bool whole = FH_SEG_WHOLE_TILES && tile_pos0 + 1024ull * (u64)nload <= a.len_total; // wave-uniform
                const uint8_t *const tb = a.seq + tile_pos0;
                const u32 voff = 16u * lane;
#pragma unroll
                for (int m = 0; m < K2S_MAX_LOADS; ++m) {
                    buf[m] = make_uint4(0u, 0u, 0u, 0u); // (every buffer defined on every path: left undefined past nload, k = 21 spills 109 registers)
                    if ((u32)m >= nload) continue;      // (chunks >= NCH: nobody looks at them)
                    if (__builtin_expect(whole, 1)) buf[m] = *reinterpret_cast<const uint4 *>(tb + (u64)(voff + 1024u * (u32)m));
                    else {
                        const u32 i = lane + 64u * (u32)m;
                        buf[m] = make_uint4(0u, 0u, 0u, 0u);
                        if (i < NCH) buf[m] = load_chunk_guarded(a.seq, tile_pos0 + 16ull * i, a.len_total);
                    }
                }
                // (the strings of the tile before are still being read by nobody: the rounds below are this wave's own)
#pragma unroll
                for (int m = 0; m < K2S_MAX_LOADS; ++m) {
                    const u32 i = lane + 64u * (u32)m;
                    if (i < NCH) {
                        u32 q, g;
                        classify_chunk(buf[m].x, buf[m].y, buf[m].z, buf[m].w, q, g);
                        Fc[1u + i] = ~q;
                        Rv[NCH - 1u - i] = pairrev32(q);
                        reinterpret_cast<unsigned short *>(Gd)[i] = (unsigned short)g;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const u64 tile_stream_pos = a.base_pos + tile_pos0;
            u32 NRt = NR; // rounds of this tile
            // start positions from the tile's first one to the end of the launch's range, as far as 32 bits count (scalar: a lane's
            // share of it is one saturating subtraction per round instead of 64-bit arithmetic per lane)
            const u32 tile_room = a.p_end > tile_pos0 ? (u32)(a.p_end - tile_pos0 < 0x7FFFFFFFull ? a.p_end - tile_pos0 : 0x7FFFFFFFull) : 0u;
            if constexpr (!LONG && RO == 32) {
                if (RAG) {
                    // ---- the tile's work items: every lane's four cells of 32 positions, trimmed to their valid windows ----
                    const u32 lane = lane_now();
                    u32 it[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32 p0 = S * lane + 32u * (u32)q;
                        const u64 g64 = seg_good_bits(Gd, p0);
                        u32 lim = __builtin_elementwise_sub_sat(tile_room, p0);
                        lim = lim < 32u ? lim : 32u;
                        const u32 W = window_valid_mask<K>(g64) & (lim >= 32u ? 0xFFFFFFFFu : ((1u << lim) - 1u));
                        const u32 first = W ? (u32)__builtin_ctz(W) : 0u, last = W ? 31u - (u32)__builtin_clz(W) : 0u;
                        it[q] = W ? ((p0 + first) | ((last - first + 1u) << 16)) : 0u;
                    }
                    // ordered by size class (29..32, 25..28, ... 1..4 windows): a round's lanes then run about equally long
                    n_items = 0;
#ifndef FH_RAG_CLASS_SHIFT
#define FH_RAG_CLASS_SHIFT 2
#endif
#pragma unroll 1
                    for (u32 cls = 0; cls < (32u >> FH_RAG_CLASS_SHIFT); ++cls) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool in = it[q] != 0u && ((32u - (it[q] >> 16)) >> FH_RAG_CLASS_SHIFT) == cls;
                            const u64 m = __builtin_amdgcn_ballot_w64(in);
                            if (in) *item_ptr(n_items + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))) = it[q];
                            n_items += (u32)__popcll(m);
                        }
                    }
                    NRt = (n_items + 63u) >> 6;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
#pragma unroll 1
            for (u32 c = c_first; c < NRt; ++c) {
                const u32 rc0 = RAG ? 0u : (u32)RO * c; // the round's first segment offset (wave-uniform)
                Win win;
                Mask Wc;
                u32 nmax = RAG ? (u32)RO : (LAST - rc0 < (u32)RO ? LAST - rc0 : (u32)RO); // positions of the (longest) segment this round covers
                {
                    const u32 lane = lane_now();
                    u32 p0 = seg_start(lane) + rc0; // the lane's view begins at this tile position
                    u32 item_n = 0;
                    if (RAG) { // the lane's work item of this round (none: no windows)
                        const u32 ii = 64u * c + lane;
                        const u32 itv = ii < n_items ? *item_ptr(ii) : 0u;
                        p0 = itv & 0xFFFFu;
                        item_n = itv >> 16;
                    }
                    // which of its windows carry a k-mer: all K bases good, inside the segment, inside [p_begin, p_end)
                    const u64 g64 = seg_good_bits(Gd, p0);
                    u32 limit = __builtin_elementwise_sub_sat(tile_room, p0);
                    limit = limit < nmax ? limit : nmax; // <= RO
                    if (SH) { // (wave-uniform: the lanes in front of a record's last one own H positions, not LAST)
                        const u32 own = __builtin_elementwise_sub_sat((lane & SUBM) == SUBM ? LAST : H, rc0);
                        limit = limit < own ? limit : own;
                    }
                    if (RAG) limit = limit < item_n ? limit : item_n;
                    if constexpr (LONG) Wc = window_valid_mask64<K>(g64) & ((1ull << limit) - 1ull); // (R <= 48)
                    else Wc = window_valid_mask<K>(g64) & (limit >= 32u ? 0xFFFFFFFFu : ((1u << limit) - 1u));
                    // a round whose windows reach the segment's last K bases may hold nothing, or nothing behind some position,
                    // in every lane at once (the records' breakers): ask the wave
                    // (asked in every round: seven instructions a round, and the compiler does not get to make two copies of
                    // the positions' code, one of them with a spilled register per position)
                    if constexpr (LONG) {
                        const u32 top = wave_max(Wc ? 64u - (u32)__builtin_clzll(Wc) : 0u);
                        if (top == 0u) continue;
                        nmax = top;
                    } else {
                        const u32 any = wave_or(Wc);
                        if (any == 0u) continue;
                        nmax = 32u - (u32)__builtin_clz(any);
                    }
                    nvalid += (u32)__popcll((u64)Wc);
                    // the lane's two strings for this round, cut out of the tile's (fh_core.h, Windows::init_words)
                    u32 nc[5], d[5];
                    seg_cut_views<K>(Fc, Rv, NCH, p0, nc, d);
                    win.init_words(nc, d);
                }

                // ---- the round's positions: k2_sketch's loop (fh_k2.hip), ending behind the last valid one ----
#pragma unroll 1
                for (int h = 0; h < HALVES; ++h) {
                if (HALVES > 1 && h) {
                    if (nmax <= (u32)R) break; // (wave-uniform)
                    nmax -= (u32)R;
                    win.template advance<(HALVES > 1 ? R : 16)>();
                    Wc >>= (HALVES > 1 ? R : 0);
                }
                auto window = [&](int j, u64 &cm, bool &is_rc) {
                    if constexpr (Win::MINF64) cm = win.canonical_word(j), is_rc = false;
                    else cm = win.canonical(j, is_rc);
                };
                u64 cm_cur;
                bool rc_cur;
                KeyWords<K> kw_cur;
                window(0, cm_cur, rc_cur);
                murmur_lookup<K, 1>(cm_cur, LT, kw_cur);
                // The positions are a chain of nested ifs, not a loop: leaving a loop early (`break`) keeps LLVM from unrolling it
                // -- its body holds convergent operations --, and skipping the bodies one by one (`continue`) merges control flow
                // behind every position, with copies of everything the software pipeline carries (four moves per position).
                auto step = [&](auto self, auto jc) __attribute__((always_inline)) -> void {
                    constexpr int j = decltype(jc)::value;
                    if ((u32)j >= nmax) return; // wave-uniform: behind the last position that is valid in any lane
                    u64 cm_nxt = 0;
                    bool rc_nxt = false;
                    KeyWords<K> kw_nxt;
                    if constexpr (j + 1 < R) {
                        window(j + 1, cm_nxt, rc_nxt);
                        murmur_lookup<K, 1>(cm_nxt, LT, kw_nxt);
                    }
                    const u64 cm = cm_cur;
                    const bool rc_loop = rc_cur;
                    const HashParts hp = murmur_finish_parts<K, SEED0>(kw_cur, a.seed);
                    const bool cand = parts_hi_plus1(hp) <= tau_hi1;
                    if (__builtin_expect(__any(cand), 0)) { // wave-uniform branch
                        const bool take = cand && ((Wc >> j) & 1u);
                        const u64 mask = __builtin_amdgcn_ballot_w64(take);
                        const u32 cnt = (u32)__popcll(mask);
                        if (cnt) {
                            if (qn + cnt > (u32)QCAP) {
                                wave_inserts += FLUSHS(a.ctl, queue, qn, shard);
                                qn = 0;
                            }
                            const u32 my = qn + __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
                            if (take) {
                                queue->ka[my] = hp.ka;
                                queue->kb[my] = hp.kb;
                                queue->k[my] = cm;
                                u32 p_here = seg_start(lane_now()) + rc0;
                                if (RAG) p_here = *item_ptr(64u * c + lane_now()) & 0xFFFFu; // (a candidate's lane has an item)
                                const u64 pos = tile_stream_pos + (u64)(p_here + (u32)(h * R + j));
                                bool is_rc = rc_loop;
                                if constexpr (Win::MINF64) is_rc = win.strand_of(j);
                                queue->p[my] = pos | ((u64)(is_rc ? 1u : 0u) << 63);
                            }
                            qn += cnt;
                        }
                    }
                    if constexpr (j + 1 < R) {
                        cm_cur = cm_nxt;
                        rc_cur = rc_nxt;
                        kw_cur = kw_nxt;
                        self(self, std::integral_constant<int, j + 1>{});
                    }
                };
                step(step, std::integral_constant<int, 0>{});
                }
                __builtin_amdgcn_wave_barrier();
                const bool last_round = c + 1u == NRt;
                if (qn >= (u32)(QCAP / 2)) { // drain when half full (and at the end of the pulled range, below)
                    wave_inserts += FLUSHS(a.ctl, queue, qn, shard);
                    qn = 0;
                }
                if (want_refresh) {
                    refresh_tau(a.ctl);
                    want_refresh = 0;
                }
                if (!(last_round && t + 1u == rt1) && wave_inserts >= a.wave_budget) {
                    // the wave's insert budget is spent: the rest of the tile and of the pulled range goes back
                    if (qn) {
                        wave_inserts += FLUSHS(a.ctl, queue, qn, shard);
                        qn = 0;
                    }
                    if ((tid & 63) == 0) {
                        // ONE entry per stopping wave, as in k2_sketch (a relaunch has at least as many waves as the list has
                        // entries, and every wave works its first entry off or hands its rest back: none is left unread)
                        const u32 idx = atomicAdd(&a.ctl->n_left_out, 1u);
                        a.left_out[3u * idx] = last_round ? t + 1u : t;
                        a.left_out[3u * idx + 1u] = rt1;
                        a.left_out[3u * idx + 2u] = last_round ? 0u : c + 1u;
                        atomicExch(&a.ctl->stopped, 1u);
                    }
                    stop = true;
                    break;
                }
            }
            if (stop) break;
            c_first = 0u;
            if (qn && t + 1u == rt1) { // nothing stays parked when the wave asks for more work (or finds none)
                wave_inserts += FLUSHS(a.ctl, queue, qn, shard);
                qn = 0;
                if (want_refresh) {
                    refresh_tau(a.ctl);
                    want_refresh = 0;
                }
            }
            // the threshold may have been lowered by any wave's admit path (fh_k2_common.h, refresh_tau)
            const u64 tau_now = load_tau();
            if (tau_now != tau) { // wave-uniform
                tau = tau_now;
                tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));
                if ((tid & 63) == 0) queue->tau = tau;
            }
        }
        if (stop || wave_inserts >= a.wave_budget) {
            if (!stop && (tid & 63) == 0) atomicExch(&a.ctl->stopped, 1u);
            break;
        }
    }
    // total_kmers (mash.rs:35): one atomic per wave
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_xor(nvalid, off);
    if ((tid & 63) == 0 && nvalid) atomicAdd((unsigned long long *)&a.ctl->kmer_counts[gw & 255u], (unsigned long long)nvalid);
}

template <int K>
static hipError_t launch_k2s_t(const SketchArgs &a, hipStream_t st) {
    const dim3 grid((a.n_waves + K2S_WPB - 1) / K2S_WPB), block(64 * K2S_WPB);
    if (a.seed == 0) hipLaunchKernelGGL((k2_sketch_seg<K, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k2_sketch_seg<K, false>), grid, block, 0, st, a);
    return hipGetLastError();
}

#ifdef FH_ONLY_K
constexpr int PART_LO = FH_ONLY_K, PART_HI = FH_ONLY_K;
#else
constexpr int PART_LO = FH_PART * (32 / FH_NPARTS) + 1;
constexpr int PART_HI = (FH_PART + 1) * (32 / FH_NPARTS);
#endif

template <int K>
static hipError_t launch_k2s_dispatch(int k, const SketchArgs &a, hipStream_t st) {
    if (k == K) return launch_k2s_t<K>(a, st);
    if constexpr (K > PART_LO) return launch_k2s_dispatch<K - 1>(k, a, st);
    return hipErrorInvalidValue;
}

#define FH_CAT2(a, b) a##b
#define FH_CAT(a, b) FH_CAT2(a, b)
hipError_t FH_CAT(launch_k2s_part, FH_PART)(int k, const SketchArgs &a, hipStream_t st) {
    if (k < PART_LO || k > PART_HI) return hipErrorInvalidValue;
    return launch_k2s_dispatch<PART_HI>(k, a, st);
}

} // namespace fh
