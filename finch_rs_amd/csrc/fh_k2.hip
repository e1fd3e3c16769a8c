// fh_k2.hip -- the hot kernel of the finch sketching path, hand-written for gfx950 (CDNA4, wave64).
//
//   k2_sketch<K>   replaces, per k-mer start position, the reference's
//                  normalize -> reverse_complement -> canonical_kmers -> hash_f -> push admit test
//                  (mash.rs:67-80, 34-42; hashing.rs:10-12).  Integer-ALU bound (the 64-bit multiplies of
//                  murmur3); algorithmic HBM traffic is 1 byte per position.
//
// Wave-level structure (one wavefront = one independent worker, no workgroup barriers in the loop):
//   * a wave owns a contiguous range of 2048-position tiles; lane l owns positions [32l, 32l+32) of a tile;
//   * phase A: every lane loads ITS 32 bytes (2 x global_load_dwordx4, coalesced 2 KiB per wave), classifies
//     them to 2-bit codes + good bits with packed byte arithmetic and stores them in a wave-private LDS
//     ring (2 tiles deep) -- so the (K-1)-base halo of lane l is simply lane l+1's LDS words, and the
//     halo of lane 63 is lane 0 of the next tile, which the ring already holds (classified once, used twice);
//   * phase B: the lane rolls the forward window in two forms (see fh_core.h), picks the canonical word,
//     hashes it with the LUT-based murmur3 (tables in LDS, built once per workgroup), compares with tau and
//     takes the (rare) upsert path with device-scope atomics on the HBM table.
//
// This file is compiled FH_NPARTS times (-DFH_PART=i), each translation unit instantiating the kernel for
// its share of K = 1..32, so that the build parallelises.
#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"
#include "fh_k2_common.h"
#include "fh_k2_lds.h" // waves per workgroup, round length and LDS layout (shared with the batch kernel, fh_k2b.hip)

#ifndef FH_PART
#error "compile with -DFH_PART=<0..FH_NPARTS-1>"
#endif

namespace fh {

// ------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------

template <int K, bool MASKED, bool SEED0, bool HASLO>
__global__ __launch_bounds__(64 * k2_wpb_of(K), k2_min_waves(K)) void k2_sketch(const SketchArgs a) {
    constexpr int WPB = k2_wpb_of(K), NTHR = 64 * WPB, REP = k2_a1_rep(K);
    // murmur3 lookup tables with the second stage folded in (fh_core.h): A / B records of two-group key words,
    // P = the key's last (short) word; sized by what this K uses
    K2Lds lds;
    if constexpr (REP == 16) lds = k2_lds_shared<K>();
    else lds = k2_lds_plain<K>();
    Rec4 *const sA1 = lds.A1, *const sA2 = lds.A2;
    Rec2 *const sB1 = lds.B1, *const sB2 = lds.B2, *const sP = lds.P;

    // the wave index is uniform, and saying so keeps everything derived from it (ring and queue addresses, shard) in
    // scalar registers instead of vector registers the hot loop would have to spill
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a gated launch was queued behind a speculative range before anybody knew how that would end (fh_api.hip): it runs
    // only if the verdict taken on the device says the speculation held
    if (a.gate && __hip_atomic_load(&a.ctl->spec_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    {
        if (has_pair_word(K, false)) {
            // (replicated: entry q's copy r at record q * REP + r, i.e. byte (q << 8) + (r << 4) for REP = 16)
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
    // (REP == 16: the lane's replica offset with the table's 64 KB base in bits 16-23, fh_core.h byte_shl8_or)
    const LutTables LT{lds.a1_lookup_base, sA2, sB1, sB2, sP, REP == 16 ? (((u32)lane & 15u) << 4) | K2S_A1 : 0u};
    __syncthreads();

    // (a loaded value lands in vector registers; the threshold is needed on the admit path only, and a 64-bit vector
    // value that lives across the whole unrolled loop gets spilled and reloaded from scratch there)
    // The threshold is read again after every tile: the admit path of ANY wave may have lowered it meanwhile (refresh_tau,
    // fh_k2_common.h) -- a persistent wave that kept the value of its first microsecond would admit at that rate for the
    // whole launch.
    auto load_tau = [&]() -> u64 {
        const u64 tau_v = __hip_atomic_load(&a.ctl->tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tau_v >> 32)) << 32) |
               (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)tau_v);
    };
    u64 tau = load_tau();
    // readfirstlane keeps the bound an opaque scalar (otherwise the select inside is re-expanded per position)
    u32 tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));
#ifdef FH_EXP_NO_ADMIT // measurement only -- EMPTY sketches: the same code with a bound no hash is under (a zero the compiler cannot
    // see: tau_lo of a launch without one), so the admit branch is there and never taken (what taking it costs the loop).
    // (Leaving the branch OUT is no measurement: without it the compiler interleaves the positions' chains and spills 44-56 registers.)
    tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(a.tau_lo >> 32));
#endif

    const u32 gw = blockIdx.x * (u32)WPB + (u32)wave;
    u32 *codes_ring = lds.codes + 256 * wave;
    u32 *good_ring = lds.good + 128 * wave;
    u32 nvalid = 0; // per lane

    // Persistent wave: pull ranges of tiles until the queue is dry, the live set reaches its soft limit
    // (checked once per pulled chunk -- polling one word per tile from 8192 waves serialises in L2), or the
    // wave has used up its own insert budget.  The budget is what makes overflow impossible whatever the
    // input: a wave inserts at most budget + 2047 new hashes per launch and the host sized the table for
    // (#waves x that) beyond the soft limit.  A stopped launch leaves its unprocessed work in the queue
    // (next_chunk + the leftover list); the host prunes and relaunches.
#ifdef FH_PROFILE_FLUSH
    u64 prof_cycles = 0, prof_calls = 0, prof_entries = 0;
    const u64 prof_t0 = __builtin_readcyclecounter();
#define FLUSH(ctl_, q_, qn_, shard_) ([&] { const u64 t0_ = __builtin_readcyclecounter(); const u32 r_ = flush_queue(ctl_, q_, qn_, shard_); prof_cycles += __builtin_readcyclecounter() - t0_; prof_calls++; prof_entries += qn_; want_refresh |= r_ >> 31; return r_ & 0x7FFFFFFFu; }())
#else
// (the count returned is wave-uniform; saying so keeps wave_inserts, qn and the branches on them scalar.  Bit 31: a refresh
// of the threshold is due -- taken at the end of the tile)
#define FLUSH(ctl_, q_, qn_, shard_) ([&] { const u32 r_ = (u32)__builtin_amdgcn_readfirstlane((int)flush_queue(ctl_, q_, qn_, shard_)); want_refresh |= r_ >> 31; return r_ & 0x7FFFFFFFu; }())
#endif
    u32 want_refresh = 0; // flush_queue asked for a refresh of the threshold (wave-uniform)
    u32 wave_inserts = 0; // new hashes this wave inserted in this launch (wave-uniform)
    u32 qn = 0;           // occupancy of the admit queue (wave-uniform)
    AdmitQueueT<false> *queue = (wave < K2_Q_SPLIT ? lds.queue_lo : lds.queue_hi) + wave;
    if (lane == 0) { // what the drain needs to finish the admit test
        queue->tau = tau;
        queue->tau_lo = HASLO ? a.tau_lo : 0ull;
        queue->hash_mask = MASKED ? a.hash_mask : ~0ull;
        queue->pre = (u32)pre_shift(K);
    }
    const u32 shard = gw & (u32)(N_SHARDS - 1);
    u32 last_unit = 0; // guides the pull size
    bool first_pull = true;
    for (;;) {
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
                last_unit = gridDim.x * (u32)WPB * a.first_units; // where the queue begins: what is left is behind it
            } else if (a.static_only) {
                // (every unit of the range was somebody's first: nothing to ask the queue for -- 1953 waves finding that out
                // with an atomic each on its one address kept the last of them waiting 20 us)
            } else if (__hip_atomic_load(&a.ctl->stopped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                // guided self-scheduling: take 1/(4 x waves) of what seems to be left, 1..MAX_UNITS units
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
#ifdef FH_EXP_NO_PHASE_A // measurement only -- WRONG sketches: the tiles behind a range's second are never loaded (what phase A and the wait for its loads cost)
        if (t == rt0)
#endif
        classify_tile(a, t + 1, lane, codes_ring, good_ring); // also provides the halo of lane 63
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const u32 par = (u32)(t & 1u);
        const uint2 own = *reinterpret_cast<const uint2 *>(&codes_ring[par * 128u + 2u * (u32)lane]);
        const uint2 nbr = *reinterpret_cast<const uint2 *>(&codes_ring[(par * 128u + 2u * (u32)lane + 2u) & 255u]);
        const u32 g_own = good_ring[par * 64u + (u32)lane];
        const u32 g_nbr = good_ring[(par * 64u + (u32)lane + 1u) & 127u];
        const u64 clo = (u64)own.x | ((u64)own.y << 32);
        const u64 chi = (u64)nbr.x | ((u64)nbr.y << 32);
        const u64 g64 = (u64)g_own | ((u64)g_nbr << 32);

        // which of the lane's 32 start positions carry a k-mer: all K bases good and inside [p_begin, p_end)
        const u64 tile_pos0 = a.p_begin + t * (u64)TILE_POS;     // wave-uniform
        const u64 tile_stream_pos = a.base_pos + tile_pos0;      // stream coordinate of the tile's first position
        const u64 lane_pos0 = tile_pos0 + (u64)lane * LANE_POS;
        const u32 limit = (a.p_end > lane_pos0) ? (u32)((a.p_end - lane_pos0) < 32 ? (a.p_end - lane_pos0) : 32) : 0u;
        const u32 W = window_valid_mask<K>(g64) & (limit >= 32u ? 0xFFFFFFFFu : ((1u << limit) - 1u));
        nvalid += (u32)__popc(W);

        Windows<K> win;
        win.init(clo, chi);

        // The lane's 32 positions in rounds of R (k2_round: 32 = one fully unrolled pass for K <= 22; 16 for the register-hungry
        // K >= 23, with the two strings moved on between rounds so that every bit-field offset stays a compile-time constant).
        // software pipeline: the table lookups of position u+1 are issued before the dependent multiply chain of
        // position u runs, so their LDS latency is hidden inside the wave
        constexpr int R = k2_round(K);
        u32 Wc = W; // valid bits of the current round in its low R bits
#pragma unroll 1
        for (int c = 0; c < LANE_POS / R; ++c) {
        // (where the canonical word is a v_min_f64 -- fh_core.h, Windows::MINF64 -- the strand flag is not formed here at all: the
        // admit path works it out for the one candidate)
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
            // reject on the high words alone (fh_core.h, HashParts); the hash_mask test hook needs the full hash.
            // windows that carry no k-mer hash garbage; they are rejected on the (rare) admit path only
            const bool cand = MASKED ? ((parts_hash(hp) & a.hash_mask) <= tau) : (parts_hi_plus1(hp) <= tau_hi1);
            if (__builtin_expect(__any(cand), 0)) { // wave-uniform branch
                // the candidate is parked with its hash unfinished; flush_queue completes and tests it (fh_k2_common.h)
                const bool take = cand && ((Wc >> j) & 1u);
                const u64 mask = __builtin_amdgcn_ballot_w64(take); // (__ballot goes through a v_cndmask and a second compare)
                const u32 cnt = (u32)__popcll(mask);
                if (cnt) {
                    if (qn + cnt > (u32)QCAP) {
                        wave_inserts += FLUSH(a.ctl, queue, qn, shard);
                        qn = 0;
                    }
                    const u32 my = qn + __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
                    if (take) {
                        queue->ka[my] = hp.ka;
                        queue->kb[my] = hp.kb;
                        queue->k[my] = cm; // as the loop carries it: pre-shifted (fh_core.h); the drain shifts
                        // position of this window, from scalars + the lane id recomputed here (two instructions)
                        // rather than a 64-bit per-lane value kept alive -- i.e. spilled -- across the loop
                        u32 lane_here; // (volatile: or the compiler hoists it out of the loop and spills it after all)
                        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_here));
                        const u64 pos = tile_stream_pos + (u64)(lane_here * (u32)LANE_POS + (u32)(c * R + j));
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
        if (qn >= (u32)(QCAP / 2) || (qn && t + 1 == rt1)) { // drain when half full or at the end of the pulled range
            wave_inserts += FLUSH(a.ctl, queue, qn, shard);
            qn = 0;
        }
        if (want_refresh) {
            refresh_tau(a.ctl);
            want_refresh = 0;
        }
        if (!MASKED) { // (the test hook's masked hashes are compared in full inside the loop: that build keeps its first threshold)
            const u64 tau_now = load_tau();
            if (tau_now != tau) { // wave-uniform
                tau = tau_now;
#ifndef FH_EXP_NO_ADMIT
                tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));
#endif
                if (lane == 0) queue->tau = tau; // (the queue is empty or its entries passed a looser test: both fine)
            }
        }
        if (t + 1 < rt1 && wave_inserts >= a.wave_budget) {
            if (qn) { // nothing may stay parked when the wave gives the rest of its range back
                wave_inserts += FLUSH(a.ctl, queue, qn, shard);
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
#ifdef FH_PROFILE_FLUSH
    if (lane == 0) {
        atomicAdd((unsigned long long *)&a.ctl->dbg_flush_cycles, (unsigned long long)prof_cycles);
        atomicAdd((unsigned long long *)&a.ctl->dbg_flush_calls, (unsigned long long)prof_calls);
        atomicAdd((unsigned long long *)&a.ctl->dbg_flush_entries, (unsigned long long)prof_entries);
        atomicAdd((unsigned long long *)&a.ctl->dbg_wave_cycles, (unsigned long long)(__builtin_readcyclecounter() - prof_t0));
    }
#endif
    // total_kmers (mash.rs:35): one atomic per wave (a HASLO launch re-reads positions already counted)
    for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_xor(nvalid, off);
    if (!HASLO && lane == 0 && nvalid)
        atomicAdd((unsigned long long *)&a.ctl->kmer_counts[gw & 255u], (unsigned long long)nvalid);
}

template <int K>
static hipError_t launch_k2_t(const SketchArgs &a, int, hipStream_t st) {
    constexpr int WPB = k2_wpb_of(K);
    const dim3 grid((a.n_waves + WPB - 1) / WPB), block(64 * WPB); // (the waves the host asked for, in workgroups of this K's size)
    const bool lo = a.tau_lo != 0ull;
    if (a.hash_mask != ~0ull) {
        if (lo) hipLaunchKernelGGL((k2_sketch<K, true, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k2_sketch<K, true, false, false>), grid, block, 0, st, a);
    } else if (a.seed == 0) {
        if (lo) hipLaunchKernelGGL((k2_sketch<K, false, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k2_sketch<K, false, true, false>), grid, block, 0, st, a);
    } else {
        if (lo) hipLaunchKernelGGL((k2_sketch<K, false, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k2_sketch<K, false, false, false>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}

#ifdef FH_ONLY_K // development builds: one K per translation unit (tools/k2_regs.py)
constexpr int PART_LO = FH_ONLY_K, PART_HI = FH_ONLY_K;
#else
constexpr int PART_LO = FH_PART * (32 / FH_NPARTS) + 1;
constexpr int PART_HI = (FH_PART + 1) * (32 / FH_NPARTS);
#endif

template <int K>
static hipError_t launch_k2_dispatch(int k, const SketchArgs &a, int blocks, hipStream_t st) {
    if (k == K) return launch_k2_t<K>(a, blocks, st);
    if constexpr (K > PART_LO) return launch_k2_dispatch<K - 1>(k, a, blocks, st);
    return hipErrorInvalidValue;
}

#define FH_CAT2(a, b) a##b
#define FH_CAT(a, b) FH_CAT2(a, b)
hipError_t FH_CAT(launch_k2_part, FH_PART)(int k, const SketchArgs &a, int blocks, hipStream_t st) {
    if (k < PART_LO || k > PART_HI) return hipErrorInvalidValue;
    return launch_k2_dispatch<PART_HI>(k, a, blocks, st);
}

} // namespace fh
