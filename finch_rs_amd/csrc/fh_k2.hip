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

#ifndef FH_PART
#error "compile with -DFH_PART=<0..FH_NPARTS-1>"
#endif

#ifndef FH_UNROLL
#define FH_UNROLL 32
#endif

namespace fh {

constexpr int UNROLL_J = FH_UNROLL;

// ------------------------------------------------------------------------------------------------
// rare path: one k-mer occurrence with hash <= tau
// ------------------------------------------------------------------------------------------------
// (all arguments by value: a by-reference SketchArgs would force every wave to spill the 128-byte
//  argument block to scratch at kernel entry)
struct TableRef {
    Entry *table;
    u32 *live;
    Ctl *ctl;
    CollRec *clog;
    u32 cap, live_cap, clog_cap;
};

__device__ __forceinline__ void log_collision(const TableRef a, u64 h, u64 kmer, u64 pos) {
    u32 i = atomicAdd(&a.ctl->n_coll, 1u);
    if (i < a.clog_cap) {
        a.clog[i].hash = h;
        a.clog[i].kmer = kmer;
        a.clog[i].pos = pos;
    } else {
        atomicExch(&a.ctl->overflow, 2u);
    }
}

// a wave-uniform pointer that arrives in vector registers (arguments of a noinline function do): telling the
// compiler so turns the loads through it into scalar loads and the accesses behind it into global_* with a scalar
// base instead of flat_* instructions
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    const u64 v = (u64)p;
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
    return (T *)(((u64)hi << 32) | lo);
}

// The table and the control block are global memory, but the pointers to them come out of memory / vector registers
// and would be treated as generic: spelled out, the entry accesses are global_* instead of flat_* instructions.
#define FH_GLOBAL __attribute__((address_space(1)))
typedef unsigned long long ull;
typedef FH_GLOBAL ull gull;
__device__ __forceinline__ ull g_load(const ull *p) {
    return __hip_atomic_load((const gull *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ ull g_cas(ull *p, ull expected, ull desired) {
    __hip_atomic_compare_exchange_strong((gull *)p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    return expected;
}
__device__ __forceinline__ void g_add(ull *p, ull v) {
    (void)__hip_atomic_fetch_add((gull *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void g_min(ull *p, ull v) {
    (void)__hip_atomic_fetch_min((gull *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __noinline__ u32 upsert(Ctl *ctl_v, u64 h, u64 kmer, u64 pos, u32 strand, u32 shard_v) {
    Ctl *ctl = uniform_ptr(ctl_v);
    const FH_GLOBAL Ctl *gctl = (const FH_GLOBAL Ctl *)ctl;
    const u32 shard = (u32)__builtin_amdgcn_readfirstlane((int)shard_v);
    const TableRef a{gctl->table, gctl->live, ctl, gctl->clog, gctl->cap, gctl->live_cap, gctl->clog_cap};
    if (h == EMPTY64) { // the one value that cannot be a table key
        atomicAdd((ull *)&a.ctl->sp_count, 1ull);
        if (strand) atomicAdd((ull *)&a.ctl->sp_extra, 1ull);
        atomicMin((ull *)&a.ctl->sp_pos, (ull)pos);
        ull oldk = atomicCAS((ull *)&a.ctl->sp_kmer, (ull)EMPTY64, (ull)kmer);
        if (oldk != EMPTY64 && oldk != kmer) log_collision(a, h, kmer, pos);
        return 0u;
    }
    // admitted hashes are tiny numbers (<= tau): spread them with a multiplicative mix before mapping to a slot
    u32 key32 = slot_key(h);
    u32 slot = (u32)(((u64)key32 * (u64)a.cap) >> 32);
    int probe = 0;
    u32 inserted = 0u;
    // The device sustains ~25 G 64-bit atomics/s whatever the table size, but 50-100 G loads/s
    // (tools/ubench_atomics.hip), and an admitted occurrence is nearly always one of a hash that is already in the
    // table with its k-mer and an earlier first position.  Unless the stream keeps hitting a few hot entries
    // (ctl->read_first, chosen by the host per launch: fh_api.hip, read_first_of) the entry is therefore *read*
    // first (key, k-mer, position in one round trip; agent-scope loads, the atomics of other XCDs are visible to
    // them) and an atomic is only issued where the value read says it could change something.  Stale reads are harmless: keys and k-mers go
    // EMPTY -> value once per launch and positions only decrease, so "already there" / "already smaller" stay true.
    const bool read_first = gctl->read_first != 0u; // wave-uniform
    ull seen_kmer = EMPTY64, seen_pos = EMPTY64;
    for (; probe < MAX_PROBE; ++probe) {
        Entry *e = &a.table[slot];
        ull old = EMPTY64;
        if (read_first) {
            old = g_load((const ull *)&e->hash);
            seen_kmer = g_load((const ull *)&e->kmer);
            seen_pos = g_load((const ull *)&e->pos);
        }
        if (old == EMPTY64) {
            old = g_cas((ull *)&e->hash, (ull)EMPTY64, (ull)h);
            seen_kmer = seen_pos = EMPTY64; // whoever owns the slot now: what was read belongs to nobody
        }
        if (old == EMPTY64) {
            // remember the slot: append to this wave's shard list (flattened into `live` after the launch)
            const u32 idx = atomicAdd(&ctl->shard_cnt[shard * (u32)SHARD_STRIDE], 1u);
            if (idx < ctl->shard_cap) ctl->shard_buf[(size_t)shard * ctl->shard_cap + idx] = slot;
            else atomicExch(&a.ctl->overflow, 1u);
            if (idx + 1u == ctl->shard_soft) atomicExch(&a.ctl->stopped, 1u); // live set full enough: drain & prune
            inserted = 1u;
            break;
        }
        if (old == h) break;
        slot = (slot + 1u == a.cap) ? 0u : slot + 1u;
    }
    if (probe == MAX_PROBE) {
        atomicExch(&a.ctl->overflow, 1u);
        return 0u;
    }
    Entry *e = &a.table[slot];
    g_add((ull *)(strand ? &e->extra : &e->count), 1ull); // one counter per strand: one atomic per occurrence
    if (seen_pos > (ull)pos) g_min((ull *)&e->pos, (ull)pos);
    ull oldk = seen_kmer;
    if (oldk == EMPTY64) oldk = g_cas((ull *)&e->kmer, (ull)EMPTY64, (ull)kmer);
    if (oldk != EMPTY64 && oldk != kmer) log_collision(a, h, kmer, pos);
    return inserted;
}

// The admit path is batched: a lane whose hash passed the threshold parks (hash, k-mer, position|strand) in a
// wave-private LDS queue; the queue is drained with all 64 lanes active, so the round trips of the atomics
// overlap instead of stalling the wave once per event.  Returns the number of NEW hashes inserted.
constexpr int QCAP = 64;
struct AdmitQueue {
    u64 h[QCAP], k[QCAP], p[QCAP];
};

__device__ __noinline__ u32 flush_queue(Ctl *ctl, const AdmitQueue *q_generic, u32 qn_v, u32 shard) {
    const u32 lane = threadIdx.x & 63u;
    u32 ins = 0u;
    // the queue lives in LDS: read it with ds_read, not through the generic (flat) pointer it arrives as
    typedef __attribute__((address_space(3))) const AdmitQueue LdsQueue;
    LdsQueue *q = (LdsQueue *)uniform_ptr(q_generic);
    const u32 qn = (u32)__builtin_amdgcn_readfirstlane((int)qn_v);
    if (lane < qn) {
        const u64 pp = q->p[lane];
        ins = upsert(ctl, q->h[lane], q->k[lane], pp & 0x7FFFFFFFFFFFFFFFull, (u32)(pp >> 63), shard);
    }
    return (u32)__popcll(__ballot(ins != 0u));
}

// ------------------------------------------------------------------------------------------------
// phase A: classify the lane's own 32 bytes of tile `t` into the wave's LDS ring
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 load_chunk_guarded(const uint8_t *seq, u64 off, u64 len) {
    if (off + 16 <= len) return *reinterpret_cast<const uint4 *>(seq + off);
    uint4 r = make_uint4(0u, 0u, 0u, 0u); // bytes past the end read as 0 => k-mer breakers
    if (off < len) {
        u32 w[4] = {0u, 0u, 0u, 0u};
        const u32 n = (u32)(len - off);
        for (u32 i = 0; i < n; ++i) w[i >> 2] |= (u32)seq[off + i] << (8 * (i & 3));
        r = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return r;
}

__device__ __forceinline__ void classify_tile(const SketchArgs &a, u64 tile, int lane, u32 *codes_ring,
                                              u32 *good_ring) {
    const u64 off = a.p_begin + tile * (u64)TILE_POS + (u64)lane * LANE_POS;
    const uint4 c0 = load_chunk_guarded(a.seq, off, a.len_total);
    const uint4 c1 = load_chunk_guarded(a.seq, off + 16, a.len_total);
    u32 q0, g0, q1, g1;
    classify_chunk(c0.x, c0.y, c0.z, c0.w, q0, g0);
    classify_chunk(c1.x, c1.y, c1.z, c1.w, q1, g1);
    const u32 par = (u32)(tile & 1u);
    *reinterpret_cast<uint2 *>(&codes_ring[par * 128u + 2u * (u32)lane]) = make_uint2(q0, q1);
    good_ring[par * 64u + (u32)lane] = g0 | (g1 << 16);
}

// ------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------
// Register budget: four waves per SIMD (<= 128 VGPRs) for every K.  With everything wave-uniform kept scalar
// (threshold, wave index, queue bookkeeping) K <= 24 fits with room to spare (k = 21: 120 VGPRs, no scratch); K >= 25
// would take 150-190 registers, but those kernels do 6-8 table lookups per position and live off the LDS pipe, where a
// fourth wave is worth more than the handful of registers it makes the compiler spill (measured k = 31: 373 Gbases/s
// unbounded at 2 waves, 427 at 3, 443 at 4, 272 at 5).  What must never be spilled is anything the admit path reads:
// a reload there stalls ~40 % of the wave-iterations of a launch that admits 1 %.
#ifndef FH_MINW_BIG
#define FH_MINW_BIG 4
#endif
#ifndef FH_MINW_SMALL
#define FH_MINW_SMALL 4
#endif
constexpr int k2_min_waves(int K) { return K <= 21 ? FH_MINW_SMALL : FH_MINW_BIG; }
template <int K, bool MASKED, bool SEED0, bool HASLO>
__global__ __launch_bounds__(256, k2_min_waves(K)) void k2_sketch(const SketchArgs a) {
    // murmur3 lookup tables with the second stage folded in (fh_core.h): A / B records of two-group key words,
    // P = the key's last (short) word; sized by what this K uses
    __shared__ Rec4 sA1[has_pair_word(K, false) ? 256 : 1];
    __shared__ Rec4 sA2[has_pair_word(K, true) ? 256 : 1];
    __shared__ Rec2 sB1[has_pair_word(K, false) ? 256 : 1];
    __shared__ Rec2 sB2[has_pair_word(K, true) ? 256 : 1];
    __shared__ Rec2 sP[partial_entries(K)];
    __shared__ __attribute__((aligned(16))) u32 sCodes[WAVES_PER_BLOCK][256];
    __shared__ __attribute__((aligned(16))) u32 sGood[WAVES_PER_BLOCK][128];
    __shared__ __attribute__((aligned(16))) AdmitQueue sQueue[WAVES_PER_BLOCK];

    // the wave index is uniform, and saying so keeps everything derived from it (ring and queue addresses, shard) in
    // scalar registers instead of vector registers the hot loop would have to spill
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        if (has_pair_word(K, false)) {
            sA1[tid] = lut_rec_A((u32)tid, false);
            sB1[tid] = lut_rec_B((u32)tid, 4, false);
        }
        if (has_pair_word(K, true)) {
            sA2[tid] = lut_rec_A((u32)tid, true);
            sB2[tid] = lut_rec_B((u32)tid, 4, true);
        }
        for (int q = tid; q < partial_entries(K); q += 256) sP[q] = lut_rec_P<K>((u32)q);
    }
    const LutTables LT{sA1, sA2, sB1, sB2, sP};
    __syncthreads();

    // (a loaded value lands in vector registers; the threshold is needed on the admit path only, and a 64-bit vector
    // value that lives across the whole unrolled loop gets spilled and reloaded from scratch there)
    const u64 tau_v = __hip_atomic_load(&a.ctl->tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 tau = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tau_v >> 32)) << 32) |
                    (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)tau_v);
    // readfirstlane keeps the bound an opaque scalar (otherwise the select inside is re-expanded per position)
    const u32 tau_hi1 = (u32)__builtin_amdgcn_readfirstlane((int)tau_hi_bound(tau));

    const u32 gw = blockIdx.x * WAVES_PER_BLOCK + (u32)wave;
    u32 *codes_ring = sCodes[wave];
    u32 *good_ring = sGood[wave];
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
#define FLUSH(ctl_, q_, qn_, shard_) ([&] { const u64 t0_ = __builtin_readcyclecounter(); const u32 r_ = flush_queue(ctl_, q_, qn_, shard_); prof_cycles += __builtin_readcyclecounter() - t0_; prof_calls++; prof_entries += qn_; return r_; }())
#else
// (the count returned is wave-uniform; saying so keeps wave_inserts, qn and the branches on them scalar)
#define FLUSH(ctl_, q_, qn_, shard_) ((u32)__builtin_amdgcn_readfirstlane((int)flush_queue(ctl_, q_, qn_, shard_)))
#endif
    u32 wave_inserts = 0; // new hashes this wave inserted in this launch (wave-uniform)
    u32 qn = 0;           // occupancy of the admit queue (wave-uniform)
    AdmitQueue *queue = &sQueue[wave];
    const u32 shard = gw & (u32)(N_SHARDS - 1);
    u32 last_unit = 0; // guides the pull size
    for (;;) {
        u32 rt0 = 0xFFFFFFFFu, rt1 = 0u;
        if (lane == 0) {
            u32 li = 0xFFFFFFFFu;
            if (a.n_left_in) li = atomicAdd(&a.ctl->left_in_pos, 1u);
            if (li < a.n_left_in) {
                rt0 = a.left_in[2u * li];
                rt1 = a.left_in[2u * li + 1u];
            } else if (__hip_atomic_load(&a.ctl->stopped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                // guided self-scheduling: take 1/(4 x waves) of what seems to be left, 1..MAX_UNITS units
                const u32 left = a.n_units > last_unit ? a.n_units - last_unit : 0u;
                u32 k = left / (4u * a.n_waves);
                k = k < 1u ? 1u : (k > (u32)MAX_UNITS ? (u32)MAX_UNITS : k);
                const u32 c = atomicAdd(&a.ctl->next_unit, k);
                last_unit = c + k;
                if (c < a.n_units) {
                    rt0 = c * (u32)UNIT_TILES;
                    const u32 e = (c + k) * (u32)UNIT_TILES;
                    rt1 = e < a.tiles_total ? e : a.tiles_total;
                }
            }
        }
        rt0 = (u32)__builtin_amdgcn_readfirstlane((int)rt0);
        rt1 = (u32)__builtin_amdgcn_readfirstlane((int)rt1);
        if (rt0 == 0xFFFFFFFFu) break;

    bool stop = false;
    classify_tile(a, rt0, lane, codes_ring, good_ring);
    for (u64 t = rt0; t < rt1; ++t) {
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

        // software pipeline: the table lookups of position j+1 are issued before the dependent multiply chain of
        // position j runs, so their LDS latency is hidden inside the wave
        auto window = [&](int j, u64 &cm, bool &is_rc) { cm = win.canonical(j, is_rc); };
        u64 cm_cur;
        bool rc_cur;
        KeyWords<K> kw_cur;
        window(0, cm_cur, rc_cur);
        murmur_lookup<K>(cm_cur, LT, kw_cur);
#pragma unroll(UNROLL_J)
        for (int j = 0; j < LANE_POS; ++j) {
            u64 cm_nxt = 0;
            bool rc_nxt = false;
            KeyWords<K> kw_nxt;
            if (j + 1 < LANE_POS) {
                window(j + 1, cm_nxt, rc_nxt);
                murmur_lookup<K>(cm_nxt, LT, kw_nxt);
            }
            const u64 cm = cm_cur;
            const bool is_rc = rc_cur;
            const HashParts hp = murmur_finish_parts<K, SEED0>(kw_cur, a.seed);
            // reject on the high words alone (fh_core.h, HashParts); the hash_mask test hook needs the full hash.
            // windows that carry no k-mer hash garbage; they are rejected on the (rare) admit path only
            const bool cand = MASKED ? ((parts_hash(hp) & a.hash_mask) <= tau) : (parts_hi_plus1(hp) <= tau_hi1);
            if (__builtin_expect(__any(cand), 0)) { // wave-uniform branch
                u64 h = parts_hash(hp);
                if (MASKED) h &= a.hash_mask; // test hook only
                const bool take = (h <= tau) && ((W >> j) & 1u) && (!HASLO || h > a.tau_lo);
                const u64 mask = __ballot(take);
                const u32 cnt = (u32)__popcll(mask);
                if (cnt) {
                    if (qn + cnt > (u32)QCAP) {
                        wave_inserts += FLUSH(a.ctl, queue, qn, shard);
                        qn = 0;
                    }
                    const u32 my = qn + __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
                    if (take) {
                        queue->h[my] = h;
                        queue->k[my] = cm >> pre_shift(K); // the loop carries the canonical word pre-shifted (fh_core.h)
                        // position of this window, from scalars + the lane id recomputed here (two instructions)
                        // rather than a 64-bit per-lane value kept alive -- i.e. spilled -- across the loop
                        u32 lane_here; // (volatile: or the compiler hoists it out of the loop and spills it after all)
                        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_here));
                        const u64 pos = tile_stream_pos + (u64)(lane_here * (u32)LANE_POS + (u32)j);
                        queue->p[my] = pos | ((u64)(is_rc ? 1u : 0u) << 63);
                    }
                    qn += cnt;
                }
            }
            if (j + 1 < LANE_POS) {
                cm_cur = cm_nxt;
                rc_cur = rc_nxt;
                kw_cur = kw_nxt;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (qn >= (u32)(QCAP / 2) || (qn && t + 1 == rt1)) { // drain when half full or at the end of the pulled range
            wave_inserts += FLUSH(a.ctl, queue, qn, shard);
            qn = 0;
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
static hipError_t launch_k2_t(const SketchArgs &a, int blocks, hipStream_t st) {
    const bool lo = a.tau_lo != 0ull;
    if (a.hash_mask != ~0ull) {
        if (lo) hipLaunchKernelGGL((k2_sketch<K, true, false, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k2_sketch<K, true, false, false>), dim3(blocks), dim3(256), 0, st, a);
    } else if (a.seed == 0) {
        if (lo) hipLaunchKernelGGL((k2_sketch<K, false, true, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k2_sketch<K, false, true, false>), dim3(blocks), dim3(256), 0, st, a);
    } else {
        if (lo) hipLaunchKernelGGL((k2_sketch<K, false, false, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k2_sketch<K, false, false, false>), dim3(blocks), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

constexpr int PART_LO = FH_PART * (32 / FH_NPARTS) + 1;
constexpr int PART_HI = (FH_PART + 1) * (32 / FH_NPARTS);

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
