// fh_k2b.hip -- MANY sketches per launch: the batch form of the sketch kernel (K = 1..32), hand-written for gfx950.
//
// finch::sketch_files (lib/src/lib.rs:29-49) is the one place the reference is parallel: a sketch per file, files side by
// side (lib.rs:34-36).  Through k2_sketch a file of a batch costs three launches and a host synchronisation -- ~150 us of
// latency-bound kernel time for the ~6 us a 4 Mb genome takes to hash (profiles/r06a_c5_busy.txt).  Here ONE launch covers
// the packed streams of all F files a worker has staged:
//
//   * the files' tiles (2048 k-mer start positions each, as k2_sketch's) form one tile space; wave w takes the contiguous
//     tiles [w q, (w + 1) q) of it -- no queue, no atomics -- and finds the file a tile belongs to by bisection of the files'
//     first tiles; a run of tiles never crosses a file (no k-mer does: the bytes behind a file's end read as breakers);
//   * every file has its own control block, table partition and shard lists (Ctl, fh_device.h), so the admit path
//     (flush_queue / upsert, fh_k2_common.h) is k2_sketch's, called with the file's control block;
//   * every file runs at ONE threshold, taken from its descriptor: the value below which ~4 n of its positions' hashes
//     are expected (fh_batch.hip).  The batch epilogue (fh_kernels.hip, k_batch_epilogue: one workgroup per file) selects
//     the n smallest, sorts them, writes each sketch to the host and leaves the partition reset; a file whose guess came up
//     short (fewer than n distinct hashes below it), or that met a 64-bit collision or any overflow, is reported as "not
//     taken" and the caller sketches it through an fh_sketcher -- the batch path never answers with anything but the exact
//     bottom-n with exact counts (mash.rs:34-63; SURVEY 8e: a function of the multiset of k-mers).
//
// The per-tile code -- phase A into the wave's LDS ring, the windows, the LUT murmur3, the high-word reject, the parked
// candidates -- is k2_sketch's, statement for statement (fh_k2.hip has the commentary); what is gone is everything a
// resident 50 Gbase stream needs and a batch of genomes does not: the work queue, budgets, leftovers, gates, the in-launch
// threshold refresh.  Compiled FH_NPARTS times like fh_k2.hip.
#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"
#include "fh_k2_common.h"
#include "fh_k2_lds.h"

#ifndef FH_PART
#error "compile with -DFH_PART=<0..FH_NPARTS-1>"
#endif

namespace fh {

__device__ __forceinline__ u64 uni64(u64 v) {
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}

// phase A of tile `tile` of a file (classify_tile of fh_k2_common.h with the file's stream in place of the launch's)
__device__ __forceinline__ void classify_tile_b(const uint8_t *seq, u64 len, u64 tile, int lane, u32 *codes_ring, u32 *good_ring) {
    const u64 tile_off = tile * (u64)TILE_POS; // wave-uniform
    uint4 c0, c1;
    if (__builtin_expect(tile_off + (u64)TILE_POS <= len, 1)) {
        const uint8_t *const tb = seq + tile_off;
        const u32 vo = (u32)lane * (u32)LANE_POS;
        c0 = *reinterpret_cast<const uint4 *>(tb + (u64)vo);
        c1 = *reinterpret_cast<const uint4 *>(tb + (u64)(vo + 16u));
    } else {
        const u64 off = tile_off + (u64)lane * LANE_POS;
        c0 = load_chunk_guarded(seq, off, len);
        c1 = load_chunk_guarded(seq, off + 16, len);
    }
    u32 q0, g0, q1, g1;
    classify_chunk(c0.x, c0.y, c0.z, c0.w, q0, g0);
    classify_chunk(c1.x, c1.y, c1.z, c1.w, q1, g1);
    const u32 par = (u32)(tile & 1u);
    *reinterpret_cast<uint2 *>(&codes_ring[par * 128u + 2u * (u32)lane]) = make_uint2(q0, q1);
    good_ring[par * 64u + (u32)lane] = g0 | (g1 << 16);
}

// ... and of a file staged in the two-bit form (fh_pack2.h): the host has done the classification, a tile is the 512 bytes of
// codes and 256 bytes of base bits phase A would have produced, lane by lane; the tile behind a file's last is zeroes
__device__ __forceinline__ void load_tile_two_bit(const uint8_t *region, u64 tile, int lane, u32 *codes_ring, u32 *good_ring) {
    const uint8_t *const tb = region + tile * (u64)TWO_BIT_TILE_BYTES; // wave-uniform
    const uint2 q = *reinterpret_cast<const uint2 *>(tb + 8u * (u32)lane);
    const u32 g = *reinterpret_cast<const u32 *>(tb + TWO_BIT_CODES_BYTES + 4u * (u32)lane);
    const u32 par = (u32)(tile & 1u);
    *reinterpret_cast<uint2 *>(&codes_ring[par * 128u + 2u * (u32)lane]) = q;
    good_ring[par * 64u + (u32)lane] = g;
}

template <int K, bool SEED0>
__global__ __launch_bounds__(64 * k2_wpb_of(K), k2_min_waves(K)) void k2_batch(const BatchArgs a) {
    constexpr int WPB = k2_wpb_of(K), NTHR = 64 * WPB, REP = k2_a1_rep(K);
    K2Lds lds;
    if constexpr (REP == 16) lds = k2_lds_shared<K>();
    else lds = k2_lds_plain<K>();
    Rec4 *const sA1 = lds.A1, *const sA2 = lds.A2;
    Rec2 *const sB1 = lds.B1, *const sB2 = lds.B2, *const sP = lds.P;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        if (has_pair_word(K, false)) {
            for (int i = tid; i < 256 * REP; i += NTHR) sA1[i] = lut_rec_A((u32)(i / REP), false);
            for (int q = tid; q < 256; q += NTHR) sB1[q] = lut_rec_B((u32)q, 4, false);
        }
        if (has_pair_word(K, true)) {
            for (int q = tid; q < 256; q += NTHR) {
                sA2[q] = lut_rec_A((u32)q, true);
                sB2[q] = lut_rec_B((u32)q, 4, true);
            }
        }
        for (int q = tid; q < partial_entries(K); q += NTHR) sP[q] = lut_rec_P<K>((u32)q);
    }
    const LutTables LT{lds.a1_lookup_base, sA2, sB1, sB2, sP, REP == 16 ? (((u32)lane & 15u) << 4) | K2S_A1 : 0u};
    __syncthreads();

    const u32 gw = blockIdx.x * (u32)WPB + (u32)wave;
    u32 *codes_ring = lds.codes + 256 * wave;
    u32 *good_ring = lds.good + 128 * wave;
    AdmitQueueT<false> *queue = (wave < K2_Q_SPLIT ? lds.queue_lo : lds.queue_hi) + wave;
    if (lane == 0) {
        queue->tau_lo = 0ull;
        queue->hash_mask = ~0ull;
        queue->pre = (u32)pre_shift(K);
    }

    // this wave's stretch of the batch's tile space
    u32 t = gw * a.tiles_per_wave;
    const u32 t_stop = (t + a.tiles_per_wave < a.tiles_total) ? t + a.tiles_per_wave : a.tiles_total;
    if (t >= t_stop) return;
    // the file tile t belongs to: the last one whose first tile is <= t (files without tiles share their successor's first
    // tile and are never the last such)
    const BatchFile *const files = uniform_ptr(a.files);
    u32 f = 0;
    {
        u32 lo = 0, hi = a.n_files; // files[lo].tile0 <= t < files[hi].tile0 (hi = n_files: the end of the tile space)
        while (hi - lo > 1u) {
            const u32 mid = (lo + hi) >> 1;
            if (files[mid].tile0 <= t) lo = mid;
            else hi = mid;
        }
        f = lo;
    }
    u32 n_flush = gw * 7u; // every drain goes to another shard list: a file's inserts come from few waves, and one list per wave would overflow

#define FLUSH_B(ctl_) ((void)flush_queue(ctl_, queue, qn, (n_flush++) & (u32)(N_SHARDS - 1)))

    while (t < t_stop) {
        f = (u32)__builtin_amdgcn_readfirstlane((int)f);
        const BatchFile *const fd = files + f;
        const u32 f_tile0 = fd->tile0, f_tiles = fd->n_tiles;
        if (f_tiles == 0u || t >= f_tile0 + f_tiles) { // (an empty file, or the stretch goes on in the next one)
            ++f;
            continue;
        }
        // (everything about the file is wave-uniform: said out loud, it stays in scalar registers across the unrolled loop)
        Ctl *const ctl = (Ctl *)uni64((u64)fd->ctl);
        const u64 f_len = uni64(fd->len);
        const u64 tau = uni64(fd->tau);
        const uint8_t *const f_seq = (const uint8_t *)uni64((u64)fd->seq);
        const u32 tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));
        if (lane == 0) queue->tau = tau;
        const u32 rt0 = t - f_tile0;
        const u32 run_end = (t_stop < f_tile0 + f_tiles ? t_stop : f_tile0 + f_tiles);
        const u32 rt1 = run_end - f_tile0;
        u32 nvalid = 0; // per lane
        u32 qn = 0;     // occupancy of the admit queue (wave-uniform)

        if (a.two_bit) load_tile_two_bit(f_seq, rt0, lane, codes_ring, good_ring);
        else classify_tile_b(f_seq, f_len, rt0, lane, codes_ring, good_ring);
        for (u64 tt = rt0; tt < rt1; ++tt) {
            // (the tile behind: it also provides the halo of lane 63)
            if (a.two_bit) load_tile_two_bit(f_seq, tt + 1, lane, codes_ring, good_ring);
            else classify_tile_b(f_seq, f_len, tt + 1, lane, codes_ring, good_ring);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const u32 par = (u32)(tt & 1u);
            const uint2 own = *reinterpret_cast<const uint2 *>(&codes_ring[par * 128u + 2u * (u32)lane]);
            const uint2 nbr = *reinterpret_cast<const uint2 *>(&codes_ring[(par * 128u + 2u * (u32)lane + 2u) & 255u]);
            const u32 g_own = good_ring[par * 64u + (u32)lane];
            const u32 g_nbr = good_ring[(par * 64u + (u32)lane + 1u) & 127u];
            const u64 clo = (u64)own.x | ((u64)own.y << 32);
            const u64 chi = (u64)nbr.x | ((u64)nbr.y << 32);
            const u64 g64 = (u64)g_own | ((u64)g_nbr << 32);

            const u64 tile_pos0 = tt * (u64)TILE_POS; // wave-uniform; a file's stream coordinates begin at 0
            const u64 lane_pos0 = tile_pos0 + (u64)lane * LANE_POS;
            const u32 limit = (f_len > lane_pos0) ? (u32)((f_len - lane_pos0) < 32 ? (f_len - lane_pos0) : 32) : 0u;
            const u32 W = window_valid_mask<K>(g64) & (limit >= 32u ? 0xFFFFFFFFu : ((1u << limit) - 1u));
            nvalid += (u32)__popc(W);

            Windows<K> win;
            win.init(clo, chi);

            // (k2_sketch runs K = 21, 22 as ONE unrolled pass of 32 at 125 / 120 VGPRs; with a file's descriptor on top that pass spills
            // 2-6 registers here, so these two K take two rounds of 16 like K >= 23: nothing a link-bound path can measure)
            constexpr int R = (K == 21 || K == 22) ? 16 : k2_round(K);
            u32 Wc = W;
#pragma unroll 1
            for (int c = 0; c < LANE_POS / R; ++c) {
                auto window = [&](int j, u64 &cm, bool &is_rc) {
                    if constexpr (Windows<K>::MINF64) cm = win.canonical_word(j), is_rc = false;
                    else cm = win.canonical(j, is_rc);
                };
                u64 cm_cur;
                bool rc_cur;
                KeyWords<K> kw_cur;
                window(0, cm_cur, rc_cur);
                murmur_lookup<K, REP>(cm_cur, LT, kw_cur);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    u64 cm_nxt = 0;
                    bool rc_nxt = false;
                    KeyWords<K> kw_nxt;
                    if (j + 1 < R) {
                        window(j + 1, cm_nxt, rc_nxt);
                        murmur_lookup<K, REP>(cm_nxt, LT, kw_nxt);
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
                                FLUSH_B(ctl);
                                qn = 0;
                            }
                            const u32 my = qn + __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
                            if (take) {
                                queue->ka[my] = hp.ka;
                                queue->kb[my] = hp.kb;
                                queue->k[my] = cm;
                                u32 lane_here;
                                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_here));
                                const u64 pos = tile_pos0 + (u64)(lane_here * (u32)LANE_POS + (u32)(c * R + j));
                                bool is_rc = rc_loop;
                                if constexpr (Windows<K>::MINF64) is_rc = win.strand_of(j);
                                queue->p[my] = pos | ((u64)(is_rc ? 1u : 0u) << 63);
                            }
                            qn += cnt;
                        }
                    }
                    if (j + 1 < R) {
                        cm_cur = cm_nxt;
                        rc_cur = rc_nxt;
                        kw_cur = kw_nxt;
                    }
                }
                if (R < LANE_POS) {
                    win.template advance<(R < LANE_POS ? R : 8)>();
                    Wc >>= (R & 31);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (qn >= (u32)(QCAP / 2) || (qn && tt + 1 == rt1)) { // drain when half full, and before the wave turns to another file
                FLUSH_B(ctl);
                qn = 0;
            }
        }
        // total_kmers (mash.rs:35) of this file: one atomic per wave and run
        for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_xor(nvalid, off);
        if (lane == 0 && nvalid) atomicAdd((unsigned long long *)&ctl->kmer_counts[gw & 255u], (unsigned long long)nvalid);
        t = run_end;
        ++f;
    }
#undef FLUSH_B
}

template <int K>
static hipError_t launch_k2b_t(const BatchArgs &a, uint32_t n_waves, hipStream_t st) {
    constexpr int WPB = k2_wpb_of(K);
    const dim3 grid((n_waves + WPB - 1) / WPB), block(64 * WPB);
    if (a.seed == 0) hipLaunchKernelGGL((k2_batch<K, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k2_batch<K, false>), grid, block, 0, st, a);
    return hipGetLastError();
}

#ifdef FH_ONLY_K // development builds: one K per translation unit
constexpr int PART_LO = FH_ONLY_K, PART_HI = FH_ONLY_K;
#else
constexpr int PART_LO = FH_PART * (32 / FH_NPARTS) + 1;
constexpr int PART_HI = (FH_PART + 1) * (32 / FH_NPARTS);
#endif

template <int K>
static hipError_t launch_k2b_dispatch(int k, const BatchArgs &a, uint32_t n_waves, hipStream_t st) {
    if (k == K) return launch_k2b_t<K>(a, n_waves, st);
    if constexpr (K > PART_LO) return launch_k2b_dispatch<K - 1>(k, a, n_waves, st);
    return hipErrorInvalidValue;
}

#define FH_CAT2(a, b) a##b
#define FH_CAT(a, b) FH_CAT2(a, b)
hipError_t FH_CAT(launch_k2b_part, FH_PART)(int k, const BatchArgs &a, uint32_t n_waves, hipStream_t st) {
    if (k < PART_LO || k > PART_HI) return hipErrorInvalidValue;
    return launch_k2b_dispatch<PART_HI>(k, a, n_waves, st);
}

} // namespace fh
