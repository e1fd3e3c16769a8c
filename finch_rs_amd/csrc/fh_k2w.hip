// fh_k2w.hip -- the sketch kernel for K = 33..64 (two-word k-mers), gfx950.
//
// finch's kmer_length is a u8 and the reference hashes k-mers of any length (sketch_schemes/mod.rs:54-71, mash.rs:21); the
// hot kernel (fh_k2.hip) keeps a k-mer in one register pair and stops at 32.  This is the same wave-level design -- persistent
// waves pulling tiles of 2048 start positions, phase A classification into the wave's LDS ring, lookup-table murmur3,
// high-word reject, wave-private admit queue -- with the per-lane arithmetic of fh_core.h's WindowsW<K>: a lane sees 96 bases
// (its own 32 and the next two lanes' -- lane 62/63's come from the next tile, which the ring already holds), the canonical
// word is four dwords, the murmur3 key has up to eight 8-byte words (16 table lookups per position at K = 64).  Seed, hash
// mask and the lower threshold of a re-read are run-time values here (one kernel per K instead of six): this path serves
// unusual k, it is not the one the roofline is quoted on.  Measured: DESIGN.md section 5.
//
// Compiled FH_NPARTS times (-DFH_PART=i), each part instantiating 8 values of K.
#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"
#include "fh_k2_common.h"

#ifndef FH_PART
#error "compile with -DFH_PART=<0..FH_NPARTS-1>"
#endif

namespace fh {

template <int K>
__global__ __launch_bounds__(256, 2) void k2_sketch_w(const SketchArgs a) {
    __shared__ Rec4 sA1[256];
    __shared__ Rec4 sA2[256];
    __shared__ Rec2 sB1[256];
    __shared__ Rec2 sB2[256];
    __shared__ Rec2 sP[partial_entries(K)];
    __shared__ __attribute__((aligned(16))) u32 sCodes[WAVES_PER_BLOCK][256];
    __shared__ __attribute__((aligned(16))) u32 sGood[WAVES_PER_BLOCK][128];
    __shared__ __attribute__((aligned(16))) AdmitQueueT<true> sQueue[WAVES_PER_BLOCK];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.gate && __hip_atomic_load(&a.ctl->spec_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return; // (fh_k2.hip)
    sA1[tid] = lut_rec_A((u32)tid, false);
    sB1[tid] = lut_rec_B((u32)tid, 4, false);
    sA2[tid] = lut_rec_A((u32)tid, true);
    sB2[tid] = lut_rec_B((u32)tid, 4, true);
    for (int q = tid; q < partial_entries(K); q += 256) sP[q] = lut_rec_P<K>((u32)q);
    const LutTables LT{sA1, sA2, sB1, sB2, sP};
    __syncthreads();

    auto load_tau = [&]() -> u64 { // re-read after every tile: refresh_tau (fh_k2_common.h) lowers it while the launch runs
        const u64 tau_v = __hip_atomic_load(&a.ctl->tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tau_v >> 32)) << 32) |
               (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)tau_v);
    };
    u64 tau = load_tau();
    u32 tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));
    const bool masked = a.hash_mask != ~0ull; // wave-uniform run-time switches
    const bool haslo = a.tau_lo != 0ull;

    const u32 gw = blockIdx.x * WAVES_PER_BLOCK + (u32)wave;
    u32 *codes_ring = sCodes[wave];
    u32 *good_ring = sGood[wave];
    u32 nvalid = 0;
#define FLUSHW(ctl_, q_, qn_, shard_) ([&] { const u32 r_ = (u32)__builtin_amdgcn_readfirstlane((int)flush_queue<true>(ctl_, q_, qn_, shard_)); want_refresh |= r_ >> 31; return r_ & 0x7FFFFFFFu; }())
    u32 wave_inserts = 0, qn = 0, want_refresh = 0;
    AdmitQueueT<true> *queue = &sQueue[wave];
    const u32 shard = gw & (u32)(N_SHARDS - 1);
    u32 last_unit = 0;
    bool first_pull = true;
    for (;;) {
        // (work distribution exactly as in k2_sketch: leftover ranges of a stopped launch first, then guided pulls)
        u32 rt0 = 0xFFFFFFFFu, rt1 = 0u;
        if (lane == 0) {
            u32 li = 0xFFFFFFFFu;
            if (a.n_left_in) li = atomicAdd(&a.ctl->left_in_pos, 1u);
            if (li < a.n_left_in) {
                rt0 = a.left_in[2u * li];
                rt1 = a.left_in[2u * li + 1u];
            } else if (first_pull && a.first_units) {
                // this wave's own units, no atomic (the queue begins behind all of them).  They are its to process even if
                // the launch has been stopped meanwhile -- nobody else will; the wave's insert budget still holds
                const u32 c = gw * a.first_units;
                if (c < a.n_units) {
                    rt0 = c * a.unit_tiles;
                    const u32 e = (c + a.first_units) * a.unit_tiles;
                    rt1 = e < a.tiles_total ? e : a.tiles_total;
                }
                last_unit = gridDim.x * (u32)WAVES_PER_BLOCK * a.first_units; // where the queue begins: what is left is behind it
            } else if (a.static_only) {
                // (every unit of the range was somebody's first: nothing to ask the queue for -- 1953 waves finding that out
                // with an atomic each on its one address kept the last of them waiting 20 us)
            } else if (__hip_atomic_load(&a.ctl->stopped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                const u32 left = a.n_units > last_unit ? a.n_units - last_unit : 0u;
                u32 k = left / (4u * a.n_waves);
                k = k < 1u ? 1u : (k > (u32)MAX_UNITS ? (u32)MAX_UNITS : k);
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

        bool stop = false;
        classify_tile(a, rt0, lane, codes_ring, good_ring);
        for (u64 t = rt0; t < rt1; ++t) {
            classify_tile(a, t + 1, lane, codes_ring, good_ring); // the halo of lanes 62 and 63
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const u32 par = (u32)(t & 1u);
            const u32 ci = par * 128u + 2u * (u32)lane;
            const uint2 w0 = *reinterpret_cast<const uint2 *>(&codes_ring[ci]);
            const uint2 w1 = *reinterpret_cast<const uint2 *>(&codes_ring[(ci + 2u) & 255u]);
            const uint2 w2 = *reinterpret_cast<const uint2 *>(&codes_ring[(ci + 4u) & 255u]);
            const u32 gi = par * 64u + (u32)lane;
            const u32 g0 = good_ring[gi], g1 = good_ring[(gi + 1u) & 127u], g2 = good_ring[(gi + 2u) & 127u];

            const u64 tile_pos0 = a.p_begin + t * (u64)TILE_POS;
            const u64 tile_stream_pos = a.base_pos + tile_pos0;
            const u64 lane_pos0 = tile_pos0 + (u64)lane * LANE_POS;
            const u32 limit = (a.p_end > lane_pos0) ? (u32)((a.p_end - lane_pos0) < 32 ? (a.p_end - lane_pos0) : 32) : 0u;
            const u32 W = window_valid_mask_w<K>(g0, g1, g2) & (limit >= 32u ? 0xFFFFFFFFu : ((1u << limit) - 1u));
            nvalid += (u32)__popc(W);

            WindowsW<K> win;
            win.init((u64)w0.x | ((u64)w0.y << 32), (u64)w1.x | ((u64)w1.y << 32), (u64)w2.x | ((u64)w2.y << 32));
            u32 Wc = W; // valid bits of the current round in its low byte
#pragma unroll 1
            for (int c = 0; c < LANE_POS / 8; ++c) {
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
                        const u64 mask = __ballot(take);
                        const u32 cnt = (u32)__popcll(mask);
                        if (cnt) {
                            if (qn + cnt > (u32)QCAP) {
                                wave_inserts += FLUSHW(a.ctl, queue, qn, shard);
                                qn = 0;
                            }
                            const u32 my = qn + __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
                            if (take) {
                                const U128 km = kmer_words_w<K>(cm);
                                queue->h[my] = h;
                                queue->k[my] = km.lo;
                                queue->khi[my] = km.hi;
                                const u64 pos = tile_stream_pos + (u64)((u32)lane * (u32)LANE_POS + (u32)(8 * c + u));
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
            if (qn >= (u32)(QCAP / 2) || (qn && t + 1 == rt1)) {
                wave_inserts += FLUSHW(a.ctl, queue, qn, shard);
                qn = 0;
            }
            if (want_refresh) {
                refresh_tau(a.ctl);
                want_refresh = 0;
            }
            if (!masked) {
                tau = load_tau();
                tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));
            }
            if (t + 1 < rt1 && wave_inserts >= a.wave_budget) {
                if (qn) {
                    wave_inserts += FLUSHW(a.ctl, queue, qn, shard);
                    qn = 0;
                }
                if (lane == 0) {
                    const u32 idx = atomicAdd(&a.ctl->n_left_out, 1u);
                    a.left_out[2u * idx] = (u32)(t + 1);
                    a.left_out[2u * idx + 1u] = rt1;
                    atomicExch(&a.ctl->stopped, 1u);
                }
                stop = true;
                break;
            }
        }
        if (stop || wave_inserts >= a.wave_budget) {
            if (!stop && lane == 0) atomicExch(&a.ctl->stopped, 1u);
            break;
        }
    }
    // total_kmers (mash.rs:35): one atomic per wave (a re-read for the hashes above a speculative threshold counts nothing)
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_xor(nvalid, off);
    if (!haslo && lane == 0 && nvalid)
        atomicAdd((unsigned long long *)&a.ctl->kmer_counts[gw & 255u], (unsigned long long)nvalid);
}

constexpr int PARTW_LO = 33 + FH_PART * (32 / FH_NPARTS);
constexpr int PARTW_HI = 32 + (FH_PART + 1) * (32 / FH_NPARTS);

template <int K>
static hipError_t launch_k2w_dispatch(int k, const SketchArgs &a, int blocks, hipStream_t st) {
    if (k == K) {
        hipLaunchKernelGGL((k2_sketch_w<K>), dim3(blocks), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if constexpr (K > PARTW_LO) return launch_k2w_dispatch<K - 1>(k, a, blocks, st);
    return hipErrorInvalidValue;
}

#define FH_CAT2(a, b) a##b
#define FH_CAT(a, b) FH_CAT2(a, b)
hipError_t FH_CAT(launch_k2w_part, FH_PART)(int k, const SketchArgs &a, int blocks, hipStream_t st) {
    if (k < PARTW_LO || k > PARTW_HI) return hipErrorInvalidValue;
    return launch_k2w_dispatch<PARTW_HI>(k, a, blocks, st);
}

} // namespace fh
