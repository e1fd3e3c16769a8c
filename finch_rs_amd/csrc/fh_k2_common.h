// fh_k2_common.h -- device code shared by the sketch kernels (fh_k2.hip: K = 1..32, fh_k2w.hip: K = 33..64): the admit
// path (upsert into the HBM table, the wave-private admit queue) and phase A (classification of a tile into the wave's LDS
// ring).  Included by both translation units; every function is a device function, so each code object gets its own copy.
#pragma once
#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"

namespace fh {

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

__device__ __forceinline__ void log_collision(const TableRef a, u64 h, u64 kmer, u64 pos, u64 kmer_hi = 0) {
    u32 i = atomicAdd(&a.ctl->n_coll, 1u);
    if (i < a.clog_cap) {
        a.clog[i].hash = h;
        a.clog[i].kmer = kmer;
        a.clog[i].pos = pos;
        a.clog[i].kmer_hi = kmer_hi;
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
__device__ __forceinline__ ull g_fetch_min(ull *p, ull v) {
    return __hip_atomic_fetch_min((gull *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void g_store(ull *p, ull v) {
    __hip_atomic_store((gull *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// K > 32 (fh_k2w.hip): the k-mer of an occurrence is two words, the table entry holds the low one, kmer_hi[slot] the
// high one.  Which k-mer bytes a sketch reports is decided at fh_finish: the table's, unless the collision log holds a
// record with the entry's final (smallest) position -- then that record's.  So the rule here is: the occurrence that
// claims the entry writes both words; every other occurrence whose k-mer differs from what it can see, OR cannot be
// compared yet (the claimer's high word has not landed; the low word is the one value that doubles as "unclaimed",
// which only the 64-mers ...T^32 can have), logs itself if it lowered the entry's first position -- an occurrence that
// did not is not the first one and its bytes cannot matter.
__device__ __forceinline__ void wide_kmer_update(const TableRef a, Entry *e, ull *khi, u32 slot, u64 h, u64 kmer, u64 kmer_hi,
                                                 u64 pos, ull seen_kmer) {
    const bool lowered = g_fetch_min((ull *)&e->pos, (ull)pos) > (ull)pos;
    ull oldk = seen_kmer;
    if (oldk == EMPTY64) oldk = g_cas((ull *)&e->kmer, (ull)EMPTY64, (ull)kmer);
    if (oldk == EMPTY64) {
        g_store(&khi[slot], (ull)kmer_hi);
        if (kmer == EMPTY64 && lowered) log_collision(a, h, kmer, pos, kmer_hi); // (the claim left the low word "unclaimed")
    } else if (oldk != kmer) {
        log_collision(a, h, kmer, pos, kmer_hi);
    } else {
        const ull oh = g_load(&khi[slot]);
        if (oh != (ull)kmer_hi && (oh != EMPTY64 || lowered)) log_collision(a, h, kmer, pos, kmer_hi);
    }
}

template <bool WIDE>
__device__ __forceinline__ u32 upsert_body(Ctl *ctl_v, u64 h, u64 kmer, u64 kmer_hi, u64 pos, u32 strand, u32 shard_v) {
    Ctl *ctl = uniform_ptr(ctl_v);
    const FH_GLOBAL Ctl *gctl = (const FH_GLOBAL Ctl *)ctl;
    const u32 shard = (u32)__builtin_amdgcn_readfirstlane((int)shard_v);
    const TableRef a{gctl->table, gctl->live, ctl, gctl->clog, gctl->cap, gctl->live_cap, gctl->clog_cap};
    ull *const khi = WIDE ? (ull *)gctl->kmer_hi : nullptr; // two-word k-mers (K > 32) only
    if (h == EMPTY64) { // the one value that cannot be a table key
        atomicAdd((ull *)&a.ctl->sp_count, 1ull);
        if (strand) atomicAdd((ull *)&a.ctl->sp_extra, 1ull);
        if (khi) { // two-word k-mers: every occurrence that lowers the first position leaves its bytes in the log
            if (atomicMin((ull *)&a.ctl->sp_pos, (ull)pos) > (ull)pos) log_collision(a, h, kmer, pos, kmer_hi);
            return 0u;
        }
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
    if (khi) {
        wide_kmer_update(a, e, khi, slot, h, kmer, kmer_hi, pos, seen_kmer);
        return inserted;
    }
    if (seen_pos > (ull)pos) g_min((ull *)&e->pos, (ull)pos);
    ull oldk = seen_kmer;
    if (oldk == EMPTY64) oldk = g_cas((ull *)&e->kmer, (ull)EMPTY64, (ull)kmer);
    if (oldk != EMPTY64 && oldk != kmer) log_collision(a, h, kmer, pos);
    return inserted;
}

// (two out-of-line instances, so that the K <= 32 kernels keep the signature and the code they were tuned with)
__device__ __noinline__ u32 upsert(Ctl *ctl_v, u64 h, u64 kmer, u64 pos, u32 strand, u32 shard_v) {
    return upsert_body<false>(ctl_v, h, kmer, 0ull, pos, strand, shard_v);
}
__device__ __noinline__ u32 upsert_w(Ctl *ctl_v, u64 h, u64 kmer, u64 kmer_hi, u64 pos, u32 strand, u32 shard_v) {
    return upsert_body<true>(ctl_v, h, kmer, kmer_hi, pos, strand, shard_v);
}

// The admit path is batched: a lane whose hash passed the threshold parks (hash, k-mer, position|strand) in a
// wave-private LDS queue; the queue is drained with all 64 lanes active, so the round trips of the atomics
// overlap instead of stalling the wave once per event.  Returns the number of NEW hashes inserted.
constexpr int QCAP = 64;
template <bool WIDE>
struct AdmitQueueT {
    u64 h[QCAP], k[QCAP], p[QCAP];
    u64 khi[QCAP]; // K > 32 (fh_k2w.hip): the k-mer's high word
};
// K <= 32: a candidate is parked UNHASHED -- the two fmix64 states the hot loop's high-word test was made on (HashParts), its
// canonical word as the loop carries it and its position -- and the drain finishes the job for up to 64 of them at once:
// last multiply and xor-shift of murmur3, the exact threshold test, the upsert.  In the hot loop that work would run on all
// 64 lanes for the one lane that is a candidate (17 % of the wave-iterations of configs[2] have one: DESIGN.md 5.3).
template <>
struct AdmitQueueT<false> {
    u64 ka[QCAP], kb[QCAP], k[QCAP], p[QCAP];
    u64 tau, tau_lo, hash_mask; // the launch's thresholds and the test hook's mask (all ones otherwise): set once per wave
    u32 pre, pad;               // pre_shift(K): the canonical word is parked as the loop carries it
};

// The threshold from the histogram of new hashes (Ctl::hist, fh_device.h): the upper edge of the first quarter-octave at
// which the running count of distinct hashes seen reaches sel_size.  Called by a whole wave (64 lanes x 4 buckets); the
// counts only grow and tau only drops (atomic min), so concurrent refreshes and inserts need no ordering.
__device__ __noinline__ void refresh_tau(Ctl *ctl_v) {
    Ctl *ctl = uniform_ptr(ctl_v);
    const FH_GLOBAL Ctl *gctl = (const FH_GLOBAL Ctl *)ctl;
    const u64 want = gctl->sel_size;
    if (want == 0ull) return;
    const u32 lane = threadIdx.x & 63u;
    u32 c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __hip_atomic_load(&ctl->hist[4u * lane + (u32)j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 mine = c[0] + c[1] + c[2] + c[3];
    u64 inc = mine; // inclusive scan over the lanes (counts fit 32 bits, their sum need not)
    for (int off = 1; off < 64; off <<= 1) {
        const u64 t = __shfl_up(inc, off);
        if (lane >= (u32)off) inc += t;
    }
    // lane 63 holds the number of hashes inserted since the reset: with what the live list held when the launch began
    // (both fixed while it runs) that is the live set's size NOW, exactly -- the shard lists' own limit (shard_soft) is
    // only the coarse guard
    if (lane == 63u) {
        const u64 n_now = (u64)gctl->n_live + (inc - gctl->inserted_total);
        if (n_now >= (u64)gctl->soft_limit) atomicExch(&ctl->stopped, 1u);
    }
    const u64 reached = __ballot(inc >= want);
    if (reached == 0ull) return; // fewer than sel_size distinct hashes so far: nothing to say yet
    const u32 first = (u32)__builtin_ctzll(reached);
    if (lane == first) {
        u64 run = inc - mine;
        u32 q = 4u * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            run += c[j];
            if (run >= want) break;
            ++q;
        }
        u64 t = qoct_upper_edge(q);
        const u64 floor_ = gctl->tau_floor;
        if (t < floor_) t = floor_;
        if (t < g_load((const ull *)&ctl->tau)) g_min((ull *)&ctl->tau, (ull)t);
    }
}

template <bool WIDE = false>
__device__ __noinline__ u32 flush_queue(Ctl *ctl, const AdmitQueueT<WIDE> *q_generic, u32 qn_v, u32 shard) {
    const u32 lane = threadIdx.x & 63u;
    u32 ins = 0u;
    // the queue lives in LDS: read it with ds_read, not through the generic (flat) pointer it arrives as
    typedef __attribute__((address_space(3))) const AdmitQueueT<WIDE> LdsQueue;
    LdsQueue *q = (LdsQueue *)uniform_ptr(q_generic);
    const u32 qn = (u32)__builtin_amdgcn_readfirstlane((int)qn_v);
    if (lane < qn) {
        const u64 pp = q->p[lane];
        if constexpr (WIDE) {
            ins = upsert_w(ctl, q->h[lane], q->k[lane], q->khi[lane], pp & 0x7FFFFFFFFFFFFFFFull, (u32)(pp >> 63), shard);
        } else {
            const u64 h = parts_hash(HashParts{q->ka[lane], q->kb[lane]}) & q->hash_mask;
            const u64 lo = q->tau_lo; // (non-zero only when a block is re-read for the hashes above a threshold that was too tight)
            if (h <= q->tau && (lo == 0ull || h > lo))
                ins = upsert(ctl, h, q->k[lane] >> q->pre, pp & 0x7FFFFFFFFFFFFFFFull, (u32)(pp >> 63), shard);
        }
    }
    // A NEW hash is counted by quarter-octave of its value (Ctl::hist); every HIST_REFRESH-th of a bucket asks the caller
    // for a refresh of the threshold (bit 31 of the result; the kernels call refresh_tau at the end of the tile, where
    // next to nothing is live across the call).  The hash is formed again from the queue rather than kept across the upsert
    // call: every register this function holds there is one the hot loop cannot use across its call of this function
    // (fh_k2.hip sits at 128 VGPRs), and the few lanes that get here can afford two multiplies.
    u32 want = 0u;
    {
        const FH_GLOBAL Ctl *gctl = (const FH_GLOBAL Ctl *)uniform_ptr(ctl);
        if (gctl->hist_on && ins != 0u) {
            u64 h;
            if constexpr (WIDE) h = q->h[lane];
            else h = parts_hash(HashParts{q->ka[lane], q->kb[lane]}) & q->hash_mask;
            const u32 old = atomicAdd(&uniform_ptr(ctl)->hist[qoct_index(h)], 1u);
            want = ((old + 1u) & (u32)(HIST_REFRESH - 1)) == 0u ? 1u : 0u;
        }
    }
    return (u32)__popcll(__ballot(ins != 0u)) | (__ballot(want != 0u) ? 0x80000000u : 0u);
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
    const u64 tile_off = a.p_begin + tile * (u64)TILE_POS; // wave-uniform
    uint4 c0, c1;
    if (__builtin_expect(tile_off + (u64)TILE_POS <= a.len_total, 1)) {
        // the whole tile lies inside the buffer (every tile but a block's last): one scalar base and the lane's 32-bit offset; the
        // guarded form is a dozen VALU instructions a chunk (64-bit address, two 64-bit compares, the zeroes)
        const uint8_t *const tb = a.seq + tile_off;
        const u32 vo = (u32)lane * (u32)LANE_POS;
        c0 = *reinterpret_cast<const uint4 *>(tb + (u64)vo);
        c1 = *reinterpret_cast<const uint4 *>(tb + (u64)(vo + 16u));
    } else {
        const u64 off = tile_off + (u64)lane * LANE_POS;
        c0 = load_chunk_guarded(a.seq, off, a.len_total);
        c1 = load_chunk_guarded(a.seq, off + 16, a.len_total);
    }
    u32 q0, g0, q1, g1;
    classify_chunk(c0.x, c0.y, c0.z, c0.w, q0, g0);
    classify_chunk(c1.x, c1.y, c1.z, c1.w, q1, g1);
    const u32 par = (u32)(tile & 1u);
    *reinterpret_cast<uint2 *>(&codes_ring[par * 128u + 2u * (u32)lane]) = make_uint2(q0, q1);
    good_ring[par * 64u + (u32)lane] = g0 | (g1 << 16);
}


} // namespace fh
