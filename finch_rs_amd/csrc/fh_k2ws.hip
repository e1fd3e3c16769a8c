// fh_k2ws.hip -- the SEGMENT form of the two-word sketch kernel (K = 33..64), gfx950.
//
// fh_k2w.hip's per-lane arithmetic (fh_core.h WindowsW<K>: a 96-base view, four rounds of eight positions, the canonical
// word as four dwords, run-time seed / test mask / lower threshold) on fh_k2s.hip's shape of the work: a lane owns one segment
// of SketchArgs::seg_stride start positions, a wave a tile of 64 segments whose 2-bit codes and good bits it keeps as
// tile-wide strings in LDS; the lane's view is cut out of them with funnel shifts by a per-lane constant; groups of eight
// positions none of which is valid in any lane are not hashed.  With the stride of reads of one length that is every record's
// last K positions -- 33 of 151 at k = 33, 64 of 151 at k = 64 (canonical_kmers yields len - k + 1 windows per record,
// mash.rs:76).  The sketch never depends on the stride (tests/test_gpu_segments.py).
//
// WindowsW::init digit-reverses the view itself, so ONE code string does (fh_k2s.hip keeps a reversed one as well, its
// rounds being shorter); four waves per workgroup and two workgroups per CU, as fh_k2w.hip (its loop needs the 256 VGPRs).
#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"
#include "fh_k2_common.h"

#ifndef FH_PART
#error "compile with -DFH_PART=<0..FH_NPARTS-1>"
#endif

namespace fh {

constexpr u32 K2WS_NCH_MAX = 4 * SEG_MAX_STRIDE + 8;                  // 16-byte chunks of a tile with its 128-byte halo
constexpr u32 K2WS_C_DW = K2WS_NCH_MAX + 2, K2WS_G_DW = K2WS_NCH_MAX / 2 + 4; // codes: chunk i at word i; good bits: at half-word i
constexpr int K2WS_MAX_LOADS = (K2WS_NCH_MAX + 63) / 64;

__device__ __forceinline__ u32 wave_or_w(u32 x) { // (fh_k2s.hip, wave_or)
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return (u32)__builtin_amdgcn_readlane((int)x, 63);
}

template <int K>
__global__ __launch_bounds__(256, 2) void k2_sketch_ws(const SketchArgs a) {
    __shared__ Rec4 sA1[256];
    __shared__ Rec4 sA2[256];
    __shared__ Rec2 sB1[256];
    __shared__ Rec2 sB2[256];
    __shared__ Rec2 sP[partial_entries(K)];
    __shared__ __attribute__((aligned(16))) u32 sC[WAVES_PER_BLOCK][K2WS_C_DW];
    __shared__ __attribute__((aligned(16))) u32 sG[WAVES_PER_BLOCK][K2WS_G_DW];
    __shared__ __attribute__((aligned(16))) AdmitQueueT<true> sQueue[WAVES_PER_BLOCK];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.gate && __hip_atomic_load(&a.ctl->spec_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    sA1[tid] = lut_rec_A((u32)tid, false);
    sB1[tid] = lut_rec_B((u32)tid, 4, false);
    sA2[tid] = lut_rec_A((u32)tid, true);
    sB2[tid] = lut_rec_B((u32)tid, 4, true);
    for (int q = tid; q < partial_entries(K); q += 256) sP[q] = lut_rec_P<K>((u32)q);
    // (the words around the strings are read, never written: "no good base there")
    for (u32 i = (u32)tid; i < (u32)WAVES_PER_BLOCK * K2WS_C_DW; i += 256u) (&sC[0][0])[i] = 0u;
    for (u32 i = (u32)tid; i < (u32)WAVES_PER_BLOCK * K2WS_G_DW; i += 256u) (&sG[0][0])[i] = 0u;
    const LutTables LT{sA1, sA2, sB1, sB2, sP};
    __syncthreads();

    auto load_tau = [&]() -> u64 {
        const u64 tau_v = __hip_atomic_load(&a.ctl->tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tau_v >> 32)) << 32) |
               (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)tau_v);
    };
    u64 tau = load_tau();
    u32 tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));
    const bool masked = a.hash_mask != ~0ull; // wave-uniform run-time switches
    const bool haslo = a.tau_lo != 0ull;

    const u32 gw = blockIdx.x * WAVES_PER_BLOCK + (u32)wave;
    u32 *const Cd = sC[wave];
    u32 *const Gd = sG[wave];
    const u32 S = (u32)__builtin_amdgcn_readfirstlane((int)a.seg_stride);
    const u32 NCH = 4u * S + 8u, NR = (S + 31u) / 32u;
    const u32 tile_pos = 64u * S;
    u32 nvalid = 0;
#define FLUSHWS(ctl_, q_, qn_, shard_) ([&] { const u32 r_ = (u32)__builtin_amdgcn_readfirstlane((int)flush_queue<true>(ctl_, q_, qn_, shard_)); want_refresh |= r_ >> 31; return r_ & 0x7FFFFFFFu; }())
    u32 wave_inserts = 0, qn = 0, want_refresh = 0;
    AdmitQueueT<true> *queue = &sQueue[wave];
    const u32 shard = gw & (u32)(N_SHARDS - 1);
    u32 last_unit = 0;
    bool first_pull = true;
    for (;;) {
        // (work distribution exactly as in fh_k2s.hip)
        u32 rt0 = 0xFFFFFFFFu, rt1 = 0u, c0 = 0u; // tiles [rt0, rt1), the first of them from round c0 on
        if (lane == 0) {
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
                last_unit = gridDim.x * (u32)WAVES_PER_BLOCK * a.first_units;
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
            const u64 tile_pos0 = a.p_begin + (u64)t * tile_pos;
            { // phase A: the tile's bytes and 128 behind them -> code string and good bits
                uint4 buf[K2WS_MAX_LOADS];
#pragma unroll
                for (int m = 0; m < K2WS_MAX_LOADS; ++m) {
                    const u32 i = (u32)lane + 64u * (u32)m;
                    buf[m] = make_uint4(0u, 0u, 0u, 0u);
                    if (i < NCH) buf[m] = load_chunk_guarded(a.seq, tile_pos0 + 16ull * i, a.len_total);
                }
#pragma unroll
                for (int m = 0; m < K2WS_MAX_LOADS; ++m) {
                    const u32 i = (u32)lane + 64u * (u32)m;
                    if (i < NCH) {
                        u32 q, g;
                        classify_chunk(buf[m].x, buf[m].y, buf[m].z, buf[m].w, q, g);
                        Cd[i] = q;
                        reinterpret_cast<unsigned short *>(Gd)[i] = (unsigned short)g;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const u64 tile_stream_pos = a.base_pos + tile_pos0;
#pragma unroll 1
            for (u32 c = c_first; c < NR; ++c) {
                const u32 rc0 = 32u * c;
                const u32 p0 = S * (u32)lane + rc0; // the lane's 96-base view begins at this tile position
                u32 nmax = S - rc0 < 32u ? S - rc0 : 32u;
                const u32 wi = p0 >> 4, ws = (2u * p0) & 31u; // codes: 16 bases a word
                const u32 gi = p0 >> 5, gs = p0 & 31u;
                u32 cw[7], gw4[4];
#pragma unroll
                for (int w = 0; w < 7; ++w) cw[w] = Cd[wi + (u32)w];
#pragma unroll
                for (int w = 0; w < 4; ++w) gw4[w] = Gd[gi + (u32)w];
                u32 v[6];
#pragma unroll
                for (int w = 0; w < 6; ++w) v[w] = alignbit_b32(cw[w + 1], cw[w], ws);
                const u32 g0 = alignbit_b32(gw4[1], gw4[0], gs), g1 = alignbit_b32(gw4[2], gw4[1], gs), g2 = alignbit_b32(gw4[3], gw4[2], gs);
                const u64 lane_pos0 = tile_pos0 + p0;
                u32 limit = (a.p_end > lane_pos0) ? (u32)((a.p_end - lane_pos0) < 32 ? (a.p_end - lane_pos0) : 32) : 0u;
                limit = limit < nmax ? limit : nmax;
                const u32 W = window_valid_mask_w<K>(g0, g1, g2) & (limit >= 32u ? 0xFFFFFFFFu : ((1u << limit) - 1u));
                {
                    const u32 any = wave_or_w(W);
                    if (any == 0u) continue; // nothing valid in any lane: the records' tails
                    nmax = 32u - (u32)__builtin_clz(any);
                }
                nvalid += (u32)__popc(W);

                WindowsW<K> win;
                win.init((u64)v[0] | ((u64)v[1] << 32), (u64)v[2] | ((u64)v[3] << 32), (u64)v[4] | ((u64)v[5] << 32));
                u32 Wc = W;
#pragma unroll 1
                for (u32 c8 = 0; 8u * c8 < nmax; ++c8) { // groups of eight positions, up to the last one valid in any lane
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        u32 cm[4];
                        bool is_rc;
                        win.canonical(u, cm, is_rc);
                        KeyWords<K> kw;
                        murmur_lookup_w<K>(cm, LT, kw);
                        const HashParts hp = murmur_finish_parts<K, false>(kw, a.seed);
                        const bool cand = masked ? ((parts_hash(hp) & a.hash_mask) <= tau) : (parts_hi_plus1(hp) <= tau_hi1);
                        if (__builtin_expect(__any(cand), 0)) { // wave-uniform branch
                            u64 h = parts_hash(hp);
                            if (masked) h &= a.hash_mask; // test hook only
                            const bool take = (h <= tau) && ((Wc >> u) & 1u) && (!haslo || h > a.tau_lo);
                            const u64 mask = __builtin_amdgcn_ballot_w64(take);
                            const u32 cnt = (u32)__popcll(mask);
                            if (cnt) {
                                if (qn + cnt > (u32)QCAP) {
                                    wave_inserts += FLUSHWS(a.ctl, queue, qn, shard);
                                    qn = 0;
                                }
                                const u32 my = qn + __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
                                if (take) {
                                    const U128 km = kmer_words_w<K>(cm);
                                    queue->h[my] = h;
                                    queue->k[my] = km.lo;
                                    queue->khi[my] = km.hi;
                                    const u64 pos = tile_stream_pos + (u64)((u32)lane * S + rc0 + 8u * c8 + (u32)u);
                                    queue->p[my] = pos | ((u64)(is_rc ? 1u : 0u) << 63);
                                }
                                qn += cnt;
                            }
                        }
                    }
                    win.advance8();
                    Wc >>= 8;
                }
                __builtin_amdgcn_wave_barrier();
                const bool last_round = c + 1u == NR;
                if (qn >= (u32)(QCAP / 2)) {
                    wave_inserts += FLUSHWS(a.ctl, queue, qn, shard);
                    qn = 0;
                }
                if (want_refresh) {
                    refresh_tau(a.ctl);
                    want_refresh = 0;
                }
                if (!(last_round && t + 1u == rt1) && wave_inserts >= a.wave_budget) {
                    if (qn) {
                        wave_inserts += FLUSHWS(a.ctl, queue, qn, shard);
                        qn = 0;
                    }
                    if (lane == 0) {
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
            if (qn && t + 1u == rt1) { // nothing stays parked when the wave asks for more work
                wave_inserts += FLUSHWS(a.ctl, queue, qn, shard);
                qn = 0;
                if (want_refresh) {
                    refresh_tau(a.ctl);
                    want_refresh = 0;
                }
            }
            if (!masked) {
                tau = load_tau();
                tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));
            }
        }
        if (stop || wave_inserts >= a.wave_budget) {
            if (!stop && lane == 0) atomicExch(&a.ctl->stopped, 1u);
            break;
        }
    }
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_xor(nvalid, off);
    if (!haslo && lane == 0 && nvalid)
        atomicAdd((unsigned long long *)&a.ctl->kmer_counts[gw & 255u], (unsigned long long)nvalid);
}

constexpr int PARTW_LO = 33 + FH_PART * (32 / FH_NPARTS);
constexpr int PARTW_HI = 32 + (FH_PART + 1) * (32 / FH_NPARTS);

template <int K>
static hipError_t launch_k2ws_dispatch(int k, const SketchArgs &a, hipStream_t st) {
    if (k == K) {
        hipLaunchKernelGGL((k2_sketch_ws<K>), dim3((a.n_waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if constexpr (K > PARTW_LO) return launch_k2ws_dispatch<K - 1>(k, a, st);
    return hipErrorInvalidValue;
}

#define FH_CAT2(a, b) a##b
#define FH_CAT(a, b) FH_CAT2(a, b)
hipError_t FH_CAT(launch_k2ws_part, FH_PART)(int k, const SketchArgs &a, hipStream_t st) {
    if (k < PARTW_LO || k > PARTW_HI) return hipErrorInvalidValue;
    return launch_k2ws_dispatch<PARTW_HI>(k, a, st);
}

} // namespace fh
