// fh_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the finch sketching hot path.
//
//   k3_prune_small   bottom-n selection: replaces the heap eviction of push (mash.rs:57-60 /
//                    scaled.rs:54-58) by sorting the live hashes of the device table in LDS, publishing
//                    the new admit threshold tau and leaving the live list sorted (== to_vec order).
//   k4_gather        to_vec (mash.rs:86-102): materialise (hash, count, extra, kmer bytes, first_pos).
//   fill/clear/synth support kernels.
//
// (the hot kernel k2_sketch<K> lives in fh_k2.hip)
#include <hip/hip_runtime.h>

#include <atomic>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"

namespace fh {

hipError_t launch_k2(int k, const SketchArgs &a, int blocks, hipStream_t st) {
    static_assert(FH_NPARTS == 4, "dispatcher below is written for 4 parts");
    if (k < 1 || k > FH_MAX_K) return hipErrorInvalidValue;
    if (k > 32 && a.seg_stride) { // two-word k-mers, segment form (fh_k2ws.hip)
        if (a.seg_stride < SEG_MIN_STRIDE || a.seg_stride > SEG_MAX_STRIDE || a.seg_sub > 1u) return hipErrorInvalidValue;
        switch ((k - 33) / (32 / FH_NPARTS)) {
        case 0: return launch_k2ws_part0(k, a, st);
        case 1: return launch_k2ws_part1(k, a, st);
        case 2: return launch_k2ws_part2(k, a, st);
        default: return launch_k2ws_part3(k, a, st);
        }
    }
    if (k > 32) { // two-word k-mers (fh_k2w.hip)
        switch ((k - 33) / (32 / FH_NPARTS)) {
        case 0: return launch_k2w_part0(k, a, blocks, st);
        case 1: return launch_k2w_part1(k, a, blocks, st);
        case 2: return launch_k2w_part2(k, a, blocks, st);
        default: return launch_k2w_part3(k, a, blocks, st);
        }
    }
    if (a.seg_stride) { // the segment form (fh_k2s.hip)
        const uint32_t sub = a.seg_sub ? a.seg_sub : 1u;
        if (sub == SEG_RAGGED) {
            if (a.seg_stride != SEG_RAGGED_STRIDE || !seg_ragged_k(k) || a.tau_lo || a.hash_mask != ~0ull) return hipErrorInvalidValue;
        } else if (a.seg_stride < SEG_MIN_STRIDE || (sub != 1u && sub != 2u && sub != 4u) || (a.seg_stride + sub - 1u) / sub > SEG_MAX_STRIDE ||
                   a.seg_stride <= (uint32_t)k || a.tau_lo || a.hash_mask != ~0ull) {
            return hipErrorInvalidValue;
        }
        switch ((k - 1) / (32 / FH_NPARTS)) {
        case 0: return launch_k2s_part0(k, a, st);
        case 1: return launch_k2s_part1(k, a, st);
        case 2: return launch_k2s_part2(k, a, st);
        default: return launch_k2s_part3(k, a, st);
        }
    }
    switch ((k - 1) / (32 / FH_NPARTS)) {
    case 0: return launch_k2_part0(k, a, blocks, st);
    case 1: return launch_k2_part1(k, a, blocks, st);
    case 2: return launch_k2_part2(k, a, blocks, st);
    default: return launch_k2_part3(k, a, blocks, st);
    }
}

hipError_t launch_k2b(int k, const BatchArgs &a, uint32_t n_waves, hipStream_t st) {
    if (k < 1 || k > 32) return hipErrorInvalidValue;
    switch ((k - 1) / (32 / FH_NPARTS)) {
    case 0: return launch_k2b_part0(k, a, n_waves, st);
    case 1: return launch_k2b_part1(k, a, n_waves, st);
    case 2: return launch_k2b_part2(k, a, n_waves, st);
    default: return launch_k2b_part3(k, a, n_waves, st);
    }
}

// ------------------------------------------------------------------------------------------------
// K3 (small): single workgroup, bitonic sort of the live hashes in LDS
// ------------------------------------------------------------------------------------------------
// rank-th smallest (1-based) of the M distinct keys in LDS: MSB-first radix select, 8 bits per pass,
// starting at the highest byte any key uses.  Every thread returns the same value.
__device__ u64 lds_select_kth(const u64 *keys, u32 M, u32 rank, u32 *hist /* 256 */, u32 *wsum /* 16 */,
                              u64 *bcast /* 2 */) {
    const u32 tid = threadIdx.x, nthr = blockDim.x;
    // OR of all keys -> first pass
    u64 kor = 0;
    for (u32 i = tid; i < M; i += nthr) kor |= keys[i];
    for (int off = 32; off > 0; off >>= 1) kor |= __shfl_xor(kor, off);
    if (tid == 0) bcast[0] = 0;
    __syncthreads();
    if ((tid & 63u) == 0 && kor) atomicOr((unsigned long long *)&bcast[0], (unsigned long long)kor);
    __syncthreads();
    kor = bcast[0];
    int shift = kor ? (int)((63 - __builtin_clzll(kor)) & ~7) : 0;
    u64 prefix = 0, pmask = 0;
    for (;; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (u32 i = tid; i < M; i += nthr) {
            const u64 key = keys[i];
            if ((key & pmask) == prefix) atomicAdd(&hist[(u32)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // inclusive scan of the 256 bins by the first 256 threads (4 waves)
        u32 c = tid < 256 ? hist[tid] : 0u, inc = c;
        const int lane = tid & 63, wave = tid >> 6;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (tid < 256 && lane == 63) wsum[wave] = inc;
        __syncthreads();
        if (tid < 256) {
            u32 base = 0;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            inc += base;
            const u32 exc = inc - c;
            if (exc < rank && rank <= inc) { // the bucket holding the rank-th key
                bcast[0] = (u64)tid;
                bcast[1] = (u64)exc;
            }
        }
        __syncthreads();
        const u32 b = (u32)bcast[0];
        rank -= (u32)bcast[1];
        prefix |= (u64)b << shift;
        pmask |= 0xFFull << shift;
        __syncthreads();
        if (shift == 0) break;
    }
    return prefix;
}

// The same in two steps for keys that spread evenly over their range -- hashes do: ONE histogram over the top 11 bits any
// key uses finds the bucket that holds the rank-th key, and the handful of keys in that bucket are ranked against each
// other directly.  Two passes over the keys instead of up to eight (the rank-th of 4000 hashes below a speculative
// threshold: 14 us -> 4).  A bucket with more than 1024 keys (keys that do not spread: the test hook's masked hashes) goes
// through the byte-wise loop above.  scratch: 2048 u32 + 1024 u64 of LDS nobody else uses during the call.
__device__ u64 lds_select_kth_fast(const u64 *keys, u32 M, u32 rank, u32 *hist256, u32 *wsum, u64 *bcast, unsigned char *scratch) {
    const u32 tid = threadIdx.x, nthr = blockDim.x; // (1024)
    u32 *hist = reinterpret_cast<u32 *>(scratch);          // [2048]
    u64 *list = reinterpret_cast<u64 *>(scratch + 8192);   // [1024]
    u64 kor = 0;
    for (u32 i = tid; i < M; i += nthr) kor |= keys[i];
    for (int off = 32; off > 0; off >>= 1) kor |= __shfl_xor(kor, off);
    if (tid == 0) bcast[0] = 0;
    for (u32 i = tid; i < 2048u; i += nthr) hist[i] = 0;
    __syncthreads();
    if ((tid & 63u) == 0 && kor) atomicOr((unsigned long long *)&bcast[0], (unsigned long long)kor);
    __syncthreads();
    kor = bcast[0];
    const int msb = kor ? 63 - __builtin_clzll(kor) : 0;
    const int shift = msb > 10 ? msb - 10 : 0;
    for (u32 i = tid; i < M; i += nthr) atomicAdd(&hist[(u32)(keys[i] >> shift) & 2047u], 1u);
    __syncthreads();
    // inclusive scan of the 2048 bins, two per thread
    const u32 c0 = hist[2u * tid], c1 = hist[2u * tid + 1u];
    u32 inc = c0 + c1;
    const u32 lane = tid & 63u, wave = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const u32 t = __shfl_up(inc, off);
        if (lane >= (u32)off) inc += t;
    }
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    u32 base = 0;
    for (u32 w = 0; w < wave; ++w) base += wsum[w];
    inc += base;
    const u32 exc = inc - c0 - c1;
    __syncthreads(); // (wsum[0] becomes the cursor of the list below)
    if (exc < rank && rank <= inc) { // this thread's pair of bins holds the rank-th key
        const bool second = rank > exc + c0;
        bcast[0] = (u64)(2u * tid + (second ? 1u : 0u));                    // the bin
        bcast[1] = (u64)(second ? exc + c0 : exc) | ((u64)(second ? c1 : c0) << 32); // keys below it | keys in it
    }
    if (tid == 0) wsum[0] = 0;
    __syncthreads();
    const u32 bin = (u32)bcast[0], below = (u32)bcast[1], in_bin = (u32)(bcast[1] >> 32);
    __syncthreads();
    if (in_bin > 1024u) return lds_select_kth(keys, M, rank, hist256, wsum, bcast);
    for (u32 i = tid; i < M; i += nthr) {
        const u64 key = keys[i];
        if (((u32)(key >> shift) & 2047u) == bin) list[atomicAdd(&wsum[0], 1u)] = key;
    }
    __syncthreads();
    const u32 want = rank - below - 1u; // the key of the bin with exactly this many smaller ones (keys are distinct)
    if (tid < in_bin) {
        const u64 mine = list[tid];
        u32 smaller = 0;
        for (u32 j = 0; j < in_bin; ++j) smaller += list[j] < mine ? 1u : 0u;
        if (smaller == want) bcast[0] = mine;
    }
    __syncthreads();
    const u64 r = bcast[0];
    __syncthreads();
    return r;
}

// The selection itself, as a device function: the kernels below wrap it.
//   sort_out = 0: radix select of the new threshold + partition of the live list (between launches)
//   sort_out = 1: the same, then the survivors sorted ascending (live list = to_vec order; fh_finish) -- at most
//                 SMALL_SORT_MAX of them, which is every case the host launches this for (kmers_to_sketch <= 3000)
// LDS: keys[SMALL_MAX] (8 B each: what the select reads), then the survivors' (key, slot) pairs for the sort.  The slots of
// the live list are held in registers between the read of the list and its rewrite (12 per thread).
// Returns the number of live entries left (all threads), or 0xFFFFFFFF if it declined (need_big set / nothing to do);
// after a sort skeys / sslots hold the sorted survivors.
struct SmallLds {
    u64 *keys;   // [SMALL_MAX]
    u64 *skeys;  // [SMALL_SORT_MAX]
    u32 *sslots; // [SMALL_SORT_MAX]
    u32 *hist;   // [256]
    u32 *wsum;   // [16]
    u64 *bcast;  // [2]
    u32 *cnt;    // [2]
};
constexpr size_t SMALL_LDS_BYTES = (size_t)SMALL_MAX * 8 + (size_t)SMALL_SORT_MAX * 12;
constexpr u32 PRUNE_DECLINED = 0xFFFFFFFFu;

__device__ u32 prune_small_dev(Entry *table, u32 *live, u32 *dead, u32 dead_cap, Ctl *ctl, u32 M, u32 kind, u64 size, u64 max_hash,
                               u32 sort_out, const SmallLds L) {
    const u32 tid = threadIdx.x, nthr = blockDim.x; // (1024)
    u64 *keys = L.keys;
    constexpr int PER = SMALL_MAX / 1024;
    u32 my_slot[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const u32 i = tid + (u32)j * 1024u;
        my_slot[j] = 0u;
        if (i < M) {
            const u32 sl = live[i];
            my_slot[j] = sl;
            keys[i] = table[sl].hash;
        }
    }
    if (tid < 2) L.cnt[tid] = 0;
    __syncthreads();
    // the new threshold and how many entries stay (mash.rs:37-60 / scaled.rs:41-58 net effect)
    u64 tau;
    u32 keep;
    if (kind == 0u) {
        if ((u64)M >= size && size > 0) {
            tau = lds_select_kth_fast(keys, M, (u32)size, L.hist, L.wsum, L.bcast, reinterpret_cast<unsigned char *>(L.skeys));
            keep = (u32)size;
        } else if (size == 0) {
            tau = 0ull;
            keep = 0;
        } else {
            tau = EMPTY64;
            keep = M;
        }
    } else {
        u32 c = 0;
        for (u32 i = tid; i < M; i += nthr) c += keys[i] <= max_hash ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if ((tid & 63u) == 0 && c) atomicAdd(&L.cnt[1], c);
        __syncthreads();
        const u32 n_le = L.cnt[1];
        __syncthreads();
        if (tid == 0) L.cnt[1] = 0;
        if ((u64)n_le >= size) {
            tau = max_hash;
            keep = n_le;
        } else if ((u64)M >= size) {
            tau = lds_select_kth_fast(keys, M, (u32)size, L.hist, L.wsum, L.bcast, reinterpret_cast<unsigned char *>(L.skeys));
            keep = (u32)size;
        } else {
            tau = (size != 0) ? EMPTY64 : max_hash;
            keep = M;
        }
        __syncthreads();
    }
    if (sort_out && keep > (u32)SMALL_SORT_MAX) { // (never launched that way: the device-wide sort takes such sketches)
        if (tid == 0) ctl->need_big = 1u;
        return PRUNE_DECLINED;
    }
    // partition: keys are distinct, so exactly `keep` of them are <= tau (or keep == M)
    const u32 nd0 = ctl->n_dead;
    const u32 ndrop = M - keep;
    const bool dead_fits = nd0 != 0xFFFFFFFFu && nd0 <= dead_cap && ndrop <= dead_cap - nd0;
    const bool none = size == 0 && kind == 0u;
    if (keep < M || sort_out) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const u32 i = tid + (u32)j * 1024u;
            if (i < M) {
                const u64 key = keys[i];
                const bool k = none ? false : (keep == M || key <= tau);
                if (k) {
                    const u32 pos = atomicAdd(&L.cnt[0], 1u);
                    if (sort_out) {
                        L.skeys[pos] = key;
                        L.sslots[pos] = my_slot[j];
                    } else {
                        live[pos] = my_slot[j];
                    }
                } else if (dead_fits) {
                    dead[nd0 + atomicAdd(&L.cnt[1], 1u)] = my_slot[j];
                }
            }
        }
    }
    __syncthreads();
    if (sort_out && keep <= 1024u) {
        // at most one survivor per thread (every sketch of the default size): a bitonic network on registers -- partners up
        // to 32 lanes away come by wave shuffle, only the ten exchanges across waves go through LDS and a barrier (the
        // all-LDS network below costs 55 barriers for 1024 keys: 16 us against 5)
        u64 key = tid < keep ? L.skeys[tid] : EMPTY64;
        u32 slot = tid < keep ? L.sslots[tid] : 0xFFFFFFFFu;
        for (u32 kk = 2; kk <= 1024u; kk <<= 1) {
            for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                u64 pk;
                u32 ps;
                if (jj >= 64u) {
                    __syncthreads();
                    L.skeys[tid] = key;
                    L.sslots[tid] = slot;
                    __syncthreads();
                    pk = L.skeys[tid ^ jj];
                    ps = L.sslots[tid ^ jj];
                } else {
                    pk = __shfl_xor(key, (int)jj);
                    ps = __shfl_xor(slot, (int)jj);
                }
                const bool keep_min = ((tid & kk) == 0u) == ((tid & jj) == 0u);
                if (keep_min ? (pk < key) : (pk > key)) {
                    key = pk;
                    slot = ps;
                }
            }
        }
        __syncthreads();
        L.skeys[tid] = key;
        L.sslots[tid] = slot;
        __syncthreads();
        if (tid < keep) live[tid] = slot;
    } else if (sort_out) {
        u32 N = 1;
        while (N < keep) N <<= 1;
        for (u32 i = keep + tid; i < N; i += nthr) {
            L.skeys[i] = EMPTY64;
            L.sslots[i] = 0xFFFFFFFFu;
        }
        __syncthreads();
        for (u32 kk = 2; kk <= N; kk <<= 1) {
            for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                for (u32 i = tid; i < N; i += nthr) {
                    const u32 ixj = i ^ jj;
                    if (ixj > i) {
                        const bool up = (i & kk) == 0;
                        const u64 a = L.skeys[i], b = L.skeys[ixj];
                        if ((a > b) == up) {
                            L.skeys[i] = b;
                            L.skeys[ixj] = a;
                            const u32 sa = L.sslots[i];
                            L.sslots[i] = L.sslots[ixj];
                            L.sslots[ixj] = sa;
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (u32 i = tid; i < keep; i += nthr) live[i] = L.sslots[i];
    }
    if (tid == 0) {
        ctl->n_live = keep;
        // The selection only ever LOWERS the threshold.  With fewer than `size` entries it has nothing to say ("no threshold"),
        // and must not undo one the host put there: a speculative range that stopped early with few hashes is relaunched for its
        // remaining tiles, and were the threshold lifted in between, those tiles would admit what the first ones turned away
        // -- a sketch that misses hashes (tools/fuzz_case_debug.py 41414 247: four waves, 27 000 distinct k-mers).
        const u64 cur = ctl->tau;
        ctl->tau = tau < cur ? tau : cur;
        ctl->sorted = sort_out ? 1u : 0u;
        ctl->n_dead = dead_fits ? nd0 + ndrop : 0xFFFFFFFFu; // (overflow marker: fh_reset sweeps the whole table)
    }
    return keep;
}

__device__ __forceinline__ SmallLds small_lds(unsigned char *smem, u32 *hist, u32 *wsum, u64 *bcast, u32 *cnt) {
    SmallLds L;
    L.keys = reinterpret_cast<u64 *>(smem);
    L.skeys = reinterpret_cast<u64 *>(smem + (size_t)SMALL_MAX * 8);
    L.sslots = reinterpret_cast<u32 *>(smem + (size_t)SMALL_MAX * 8 + (size_t)SMALL_SORT_MAX * 8);
    L.hist = hist;
    L.wsum = wsum;
    L.bcast = bcast;
    L.cnt = cnt;
    return L;
}

__global__ __launch_bounds__(1024) void k3_prune_small(Entry *table, u32 *live, u32 *dead, u32 dead_cap, Ctl *ctl,
                                                       u32 kind, u64 size, u64 max_hash, u32 trigger, u32 force,
                                                       u32 sort_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 s_hist[256];
    __shared__ u32 s_wsum[16];
    __shared__ u64 s_bcast[2];
    __shared__ u32 s_cnt[2];
    const u32 M = ctl->n_live;
    if (ctl->need_big) return;
    if (!force && M <= trigger) return;
    if (M > (u32)SMALL_MAX) {
        if (threadIdx.x == 0) ctl->need_big = 1u;
        return;
    }
    (void)prune_small_dev(table, live, dead, dead_cap, ctl, M, kind, size, max_hash, sort_out, small_lds(smem, s_hist, s_wsum, s_bcast, s_cnt));
}

static hipError_t small_lds_attr(const void *fn) {
    // function attributes belong to the current device: once per device and kernel, and harmless if two worker threads
    // of the same device both get here first
    static std::atomic<const void *> done[64][4];
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev >= 0 && dev < 64)
        for (auto &d : done[dev])
            if (d.load(std::memory_order_acquire) == fn) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_BYTES); e != hipSuccess) return e;
    if (dev >= 0 && dev < 64)
        for (auto &d : done[dev]) {
            const void *expect = nullptr;
            if (d.compare_exchange_strong(expect, fn, std::memory_order_acq_rel)) break;
        }
    return hipSuccess;
}

hipError_t launch_prune_small(Entry *table, u32 *live, u32 *dead, u32 dead_cap, Ctl *ctl, u32 kind, u64 size,
                              u64 max_hash, u32 trigger, u32 force, u32 sort_out, hipStream_t st) {
    if (hipError_t e = small_lds_attr(reinterpret_cast<const void *>(k3_prune_small)); e != hipSuccess) return e;
    hipLaunchKernelGGL(k3_prune_small, dim3(1), dim3(1024), SMALL_LDS_BYTES, st, table, live, dead, dead_cap, ctl, kind, size,
                       max_hash, trigger, force, sort_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// the fused epilogue of small sketches (kmers_to_sketch <= 3000, Mash)
// ------------------------------------------------------------------------------------------------
// Everything that follows a sketch launch of a small sketch, in ONE single-workgroup kernel instead of up to seven launches
// and three host round trips: append the shard lists of new inserts to the live list (k_live_flatten + k_live_commit),
// select / sort (k3_prune_small), take the verdict on a speculative range (Ctl::spec_ok), write the finished sketch's
// columns straight into the caller's pinned host buffer (k4_gather + the device-to-host copy) and mirror the control
// block there (the check_ctl copy).  A file of a batch (configs[4]) is reset, sketched and finished with one
// synchronisation; a pass over 50 Gbases queues its speculative prefix, the verdict, the gated main launch and the finish
// back to back.  The host checks the mirrored control block afterwards and falls back to the step-by-step path whenever
// something did not go as queued (speculation failed, launch stopped early, more live entries than the LDS holds).
__device__ __forceinline__ void clear_entry(Entry *e);
__device__ __forceinline__ void init_ctl_dev(Ctl *ctl, u64 tau0, u32 keep_text_bases, u64 sel_size, u64 tau_floor, u32 hist_on);

// (the body of k_small_epilogue and of k_batch_epilogue's workgroups; returns what the gather reported: 0, FIN_OK or FIN_OK_RESET)
__device__ __forceinline__ u32 small_epilogue_body(const EpiArgs &a, unsigned char *smem, u32 *s_hist, u32 *s_wsum, u64 *s_bcast,
                                                   u32 *s_cnt, u32 *s_off, u32 &s_live) {
    Ctl *ctl = a.ctl;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool skip = (a.flags & EPI_GATED) && ctl->spec_ok == 0u; // (the launch this follows did nothing: leave everything as it is)
    if (!skip) {
        u32 M = ctl->n_live;
        if (a.flags & EPI_FLATTEN) {
            // shard lists -> flat live list: an exclusive scan of the 256 cursors, then one wave per shard in turn
            static_assert(N_SHARDS == 256, "the scan below is written for 256 shards");
            u32 c = 0;
            if (tid < (u32)N_SHARDS) {
                c = ctl->shard_cnt[tid * SHARD_STRIDE];
                if (c > ctl->shard_cap) c = ctl->shard_cap; // overflow already flagged by the inserter
            }
            u32 inc = c;
            for (int off = 1; off < 64; off <<= 1) {
                const u32 t = __shfl_up(inc, off);
                if (lane >= (u32)off) inc += t;
            }
            if (tid < (u32)N_SHARDS && lane == 63u) s_wsum[wave] = inc;
            __syncthreads();
            if (tid < (u32)N_SHARDS) {
                u32 base = 0;
                for (u32 w = 0; w < wave; ++w) base += s_wsum[w];
                s_off[tid] = base + inc - c;
                if (tid == (u32)N_SHARDS - 1) s_off[N_SHARDS] = base + inc;
            }
            __syncthreads();
            const u32 total = s_off[N_SHARDS];
            u32 *live = ctl->live;
            const u32 live_cap = ctl->live_cap, shard_cap = ctl->shard_cap;
            const u32 *buf = ctl->shard_buf;
            // (one entry per thread and trip, its shard found by bisection of the offsets: every load of the copy is in
            // flight at once -- a wave per shard in turn made sixteen dependent round trips to HBM, 30 us of a 45 us kernel)
            for (u32 i = tid; i < total; i += 1024u) {
                u32 lo = 0, hi = (u32)N_SHARDS;
                while (hi - lo > 1u) {
                    const u32 mid = (lo + hi) >> 1;
                    if (s_off[mid] <= i) lo = mid;
                    else hi = mid;
                }
                const u32 v = buf[(size_t)lo * shard_cap + (i - s_off[lo])];
                if (M + i < live_cap) live[M + i] = v;
                else atomicExch(&ctl->overflow, 1u);
            }
            if (tid < (u32)N_SHARDS) ctl->shard_cnt[tid * SHARD_STRIDE] = 0;
            if (tid == 0 && total) {
                ctl->inserted_total += total;
                u32 n = M + total;
                if (n > live_cap) n = live_cap;
                ctl->n_live = n;
                ctl->sorted = 0;
                s_live = n;
            } else if (tid == 0) {
                s_live = M;
            }
            __syncthreads(); // (also: the appended slots are visible to the whole workgroup)
            M = s_live;
        }
        u32 kept = PRUNE_DECLINED;
        const u32 force = a.flags & EPI_PRUNE_FORCE, sort_out = (a.flags & EPI_SORT) ? 1u : 0u;
        if ((a.flags & (EPI_PRUNE_FORCE | EPI_PRUNE_TRIGGER)) && !ctl->need_big && (force || M > a.trigger)) {
            if (M > (u32)SMALL_MAX) {
                if (tid == 0) ctl->need_big = 1u;
            } else {
                kept = prune_small_dev(a.table, a.live, a.dead, a.dead_cap, ctl, M, a.kind, a.size, a.max_hash, sort_out,
                                       small_lds(smem, s_hist, s_wsum, s_bcast, s_cnt));
            }
        }
        __syncthreads();
        if (a.flags & EPI_VERDICT) {
            // the speculation held iff the range ran dry (nothing left in the queue, no leftover ranges), nothing overflowed
            // and at least `size` distinct hashes at or below the guess were found
            if (tid == 0) {
                const bool dry = ctl->next_unit >= a.n_units && ctl->n_left_out == 0u;
                const bool ok = dry && kept != PRUNE_DECLINED && (u64)kept >= a.size && !ctl->need_big && !ctl->overflow;
                ctl->spec_ok = ok ? 1u : 0u;
            }
        }
        if ((a.flags & EPI_GATHER) && kept != PRUNE_DECLINED) {
            // to_vec (mash.rs:86-102) from the sorted survivors in LDS, straight into the host's columns
            const SmallLds L = small_lds(smem, s_hist, s_wsum, s_bcast, s_cnt);
            const size_t st = a.out_stride;
            u64 *o_hash = a.out, *o_kmer = o_hash + st, *o_pos = o_kmer + st;
            u64 *o_kmer_hi = a.wide ? o_pos + st : nullptr;
            u32 *o_count = (u32 *)((a.wide ? o_kmer_hi : o_pos) + st), *o_extra = o_count + st;
            for (u32 i = tid; i < kept; i += 1024u) {
                const u32 sl = L.sslots[i];
                const Entry e = a.table[sl];
                o_hash[i] = L.skeys[i];
                const u64 occ = e.count + e.extra; // the table counts the two strands separately (fh_device.h)
                o_count[i] = occ > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)occ;
                o_extra[i] = e.extra > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)e.extra;
                o_kmer[i] = e.kmer;
                if (o_kmer_hi) o_kmer_hi[i] = ctl->kmer_hi[sl];
                o_pos[i] = e.pos;
            }
        }
    }
    // fh_finish's epilogue: did everything that was queued do what it was queued for?  (The host asks the mirrored control
    // block the same questions; with EPI_RESET the answer is taken here, because the handle is to be left reset.)
    u32 fin = 0u;
    if ((a.flags & EPI_GATHER) && !skip) {
        __syncthreads();
        const bool spec_ok = !(a.flags & EPI_NEED_SPEC) || ctl->spec_ok != 0u;
        const bool dry = a.check_units == 0u || (ctl->next_unit >= a.check_units && ctl->n_left_out == 0u);
        const bool ok = spec_ok && dry && !ctl->need_big && !ctl->overflow && ctl->sorted == 1u && (u64)ctl->n_live <= a.size;
        if (ok) fin = ((a.flags & EPI_RESET) && ctl->n_dead != 0xFFFFFFFFu) ? FIN_OK_RESET : FIN_OK;
        __syncthreads();
        if (fin && tid == 0) ctl->sorted = fin;
    }
    if (a.h_ctl) {
        // the control block as the host reads it (what the check_ctl copy used to fetch); every write above is visible to
        // this workgroup after the barrier
        __syncthreads();
        const u32 *src = reinterpret_cast<const u32 *>(ctl);
        u32 *dst = reinterpret_cast<u32 *>(a.h_ctl);
        for (u32 i = tid; i < (u32)(sizeof(Ctl) / 4); i += 1024u) dst[i] = src[i];
        __threadfence_system();
    }
    if (fin == FIN_OK_RESET) {
        // the sketch is with the host: clear the slots this run touched (the survivors -- the sorted list in LDS -- and the
        // dropped ones) and re-initialise the control block, so that the fh_reset in front of the next file launches nothing
        __syncthreads();
        const SmallLds L = small_lds(smem, s_hist, s_wsum, s_bcast, s_cnt);
        const u32 nl = ctl->n_live, nd = ctl->n_dead;
        u64 *khi = ctl->kmer_hi;
        for (u32 i = tid; i < nl; i += 1024u) {
            const u32 sl = L.sslots[i];
            clear_entry(&a.table[sl]);
            if (khi) khi[sl] = EMPTY64;
        }
        for (u32 i = tid; i < nd; i += 1024u) {
            const u32 sl = a.dead[i];
            clear_entry(&a.table[sl]);
            if (khi) khi[sl] = EMPTY64;
        }
        __syncthreads(); // every thread has read the two counts
        init_ctl_dev(ctl, a.tau0, 0u, a.size, 0ull, a.hist_on);
    }
    return fin;
}

__global__ __launch_bounds__(1024) void k_small_epilogue(const EpiArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 s_hist[256];
    __shared__ u32 s_wsum[16];
    __shared__ u64 s_bcast[2];
    __shared__ u32 s_cnt[2];
    __shared__ u32 s_off[N_SHARDS + 1];
    __shared__ u32 s_live;
    (void)small_epilogue_body(a, smem, s_hist, s_wsum, s_bcast, s_cnt, s_off, s_live);
}

// Many sketches per launch (fh_k2b.hip, fh_batch.hip): workgroup f finishes file f of the batch -- the same fused epilogue, F
// of them side by side on F compute units.  A file whose epilogue did not end in FIN_OK_RESET (more hashes than the LDS
// selection holds, a full shard list or table partition: the host is told through the mirrored control block and sketches
// the file the long way) has its whole partition swept here, so that the next batch finds every partition clean.
__global__ __launch_bounds__(1024) void k_batch_epilogue(const EpiArgs *args, u32 read_first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 s_hist[256];
    __shared__ u32 s_wsum[16];
    __shared__ u64 s_bcast[2];
    __shared__ u32 s_cnt[2];
    __shared__ u32 s_off[N_SHARDS + 1];
    __shared__ u32 s_live;
    const EpiArgs a = args[blockIdx.x];
    const u32 fin = small_epilogue_body(a, smem, s_hist, s_wsum, s_bcast, s_cnt, s_off, s_live);
    Ctl *ctl = a.ctl;
    if (fin != FIN_OK_RESET) {
        __syncthreads();
        const u32 cap = ctl->cap;
        for (u32 i = threadIdx.x; i < cap; i += 1024u) clear_entry(&a.table[i]);
        if (threadIdx.x < (u32)N_SHARDS) ctl->shard_cnt[threadIdx.x * SHARD_STRIDE] = 0;
        __syncthreads();
        init_ctl_dev(ctl, a.tau0, 0u, a.size, 0ull, a.hist_on);
    }
    __syncthreads();
    if (threadIdx.x == 0) ctl->read_first = read_first; // (init_ctl_dev leaves it 0; the batch kernel has no queue reset that would set it)
}

hipError_t launch_batch_epilogue(const EpiArgs *args, uint32_t n_files, uint32_t read_first, hipStream_t st) {
    if (n_files == 0) return hipSuccess;
    if (hipError_t e = small_lds_attr(reinterpret_cast<const void *>(k_batch_epilogue)); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_batch_epilogue, dim3(n_files), dim3(1024), SMALL_LDS_BYTES, st, args, read_first);
    return hipGetLastError();
}

__global__ __launch_bounds__(1024) void k_batch_init(const BatchPartition *parts, u64 size, u32 read_first) {
    const BatchPartition p = parts[blockIdx.x];
    for (u32 i = threadIdx.x; i < p.cap; i += 1024u) clear_entry(&p.table[i]);
    if (threadIdx.x < (u32)N_SHARDS) p.shard_cnt[threadIdx.x * SHARD_STRIDE] = 0;
    if (threadIdx.x == 0) {
        Ctl *ctl = p.ctl;
        ctl->kmer_hi = nullptr;
        ctl->table = p.table;
        ctl->live = p.live;
        ctl->clog = p.clog;
        ctl->cap = p.cap;
        ctl->live_cap = p.live_cap;
        ctl->clog_cap = p.clog_cap;
        ctl->shard_cnt = p.shard_cnt;
        ctl->shard_buf = p.shard_buf;
        ctl->shard_cap = p.shard_cap;
        ctl->pad1 = 0;
        ctl->text_bases = 0;
    }
    __syncthreads();
    init_ctl_dev(p.ctl, EMPTY64, 0u, size, 0ull, 0u);
    __syncthreads();
    if (threadIdx.x == 0) p.ctl->read_first = read_first;
}

hipError_t launch_batch_init(const BatchPartition *parts, uint32_t n_files, uint64_t size, uint32_t read_first, hipStream_t st) {
    if (n_files == 0) return hipSuccess;
    hipLaunchKernelGGL(k_batch_init, dim3(n_files), dim3(1024), 0, st, parts, size, read_first);
    return hipGetLastError();
}

hipError_t launch_small_epilogue(const EpiArgs &a, hipStream_t st) {
    if (hipError_t e = small_lds_attr(reinterpret_cast<const void *>(k_small_epilogue)); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_small_epilogue, dim3(1), dim3(1024), SMALL_LDS_BYTES, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K4: to_vec
// ------------------------------------------------------------------------------------------------
__global__ void k4_gather(const Entry *table, const u32 *live, const Ctl *ctl, int k, u64 *o_hash, u32 *o_count,
                          u32 *o_extra, u64 *o_kmer, u64 *o_kmer_hi, u64 *o_pos, u32 cap_out) {
    const u32 n = ctl->n_live < cap_out ? ctl->n_live : cap_out;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Entry e = table[live[i]];
        o_hash[i] = e.hash;
        const u64 occ = e.count + e.extra; // the table counts the two strands separately (fh_device.h)
        o_count[i] = occ > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)occ;
        o_extra[i] = e.extra > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)e.extra;
        o_kmer[i] = e.kmer;
        if (o_kmer_hi) o_kmer_hi[i] = ctl->kmer_hi[live[i]];
        o_pos[i] = e.pos;
    }
    (void)k;
}

hipError_t launch_gather(const Entry *table, const u32 *live, const Ctl *ctl, int k, u64 *o_hash, u32 *o_count,
                         u32 *o_extra, u64 *o_kmer, u64 *o_kmer_hi, u64 *o_pos, u32 cap_out, hipStream_t st) {
    hipLaunchKernelGGL(k4_gather, dim3(64), dim3(256), 0, st, table, live, ctl, k, o_hash, o_count, o_extra, o_kmer,
                       o_kmer_hi, o_pos, cap_out);
    return hipGetLastError();
}

// rows of a finished sketch's hash / k-mer columns (fh_copy_out_rows on a large sketch whose wide columns stayed on the
// device: configs[2] keeps 10 000 of 2 000 000 records): out = n hashes, n k-mer words, [n high k-mer words]
__global__ void k_gather_rows(const u64 *hash, const u64 *kmer, const u64 *kmer_hi, const u32 *rows, u32 n, u64 *out) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u32 r = rows[i];
        out[i] = hash[r];
        out[(size_t)n + i] = kmer[r];
        if (kmer_hi) out[2 * (size_t)n + i] = kmer_hi[r];
    }
}
hipError_t launch_gather_rows(const u64 *hash, const u64 *kmer, const u64 *kmer_hi, const u32 *rows, u32 n, u64 *out, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_rows, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, st, hash, kmer, kmer_hi, rows, n, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sampling pre-pass of large sketches: where will the threshold of a kmers_to_sketch-hash sketch of this block end up?
// ------------------------------------------------------------------------------------------------
// A sketch of millions of hashes (the CLI's 200-fold oversketch, cli.rs:187-192) finds its threshold by filling the
// table: at stream position x everything below n / distinct(x) has to be admitted, ~n ln(D / n) upserts more than a
// pass that knew the final threshold from the start.  So the block is SAMPLED first: the sketch kernel itself runs over
// one run of RUN_TILES tiles out of every `stride` (an occurrence sample, uniform over the block whatever its order) at
// a cap threshold, which leaves the sample's hashes with their sample multiplicities in the table; k_live_count_hist
// turns the live entries into three histograms over hash value (distinct sample hashes, those seen once, those seen
// twice), and the host estimates from them how many DISTINCT k-mers of the whole block lie below each candidate
// threshold: Chao's lower bound S + c1^2 / (2 c2) on the number of species -- an occurrence sample misses most k-mers
// that occur a few times, the singleton / doubleton ratio says how many.  The bound errs low, so the threshold comes out
// loose (more upserts, still far fewer than without); a guess that is too tight after all is caught after the pass
// (fewer than `size` hashes live) and repaired by a second pass for the hashes above it.  The sketch never depends on
// the estimate being right.
__global__ void k_fill_tile_runs(u32 *list, u32 n_runs, u32 stride, u32 run_tiles, u32 tiles_total) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += gridDim.x * blockDim.x) {
        const u32 t0 = i * stride, t1 = t0 + run_tiles;
        list[2u * i] = t0;
        list[2u * i + 1u] = t1 < tiles_total ? t1 : tiles_total;
    }
}

hipError_t launch_fill_tile_runs(u32 *list, u32 n_runs, u32 stride, u32 run_tiles, u32 tiles_total, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_tile_runs, dim3((n_runs + 255u) / 256u), dim3(256), 0, st, list, n_runs, stride, run_tiles, tiles_total);
    return hipGetLastError();
}

// per quarter-octave of hash value (qoct_index): live entries, those with one occurrence, those with two
__global__ __launch_bounds__(256) void k_live_count_hist(const Entry *table, const u32 *live, const Ctl *ctl, u32 *hist /* [3][256] */) {
    __shared__ u32 lh[3][256];
    for (int i = threadIdx.x; i < 768; i += blockDim.x) (&lh[0][0])[i] = 0;
    __syncthreads();
    const u32 n = ctl->n_live;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Entry e = table[live[i]];
        const u32 q = qoct_index(e.hash);
        const u64 c = e.count + e.extra;
        atomicAdd(&lh[0][q], 1u);
        if (c == 1ull) atomicAdd(&lh[1][q], 1u);
        if (c == 2ull) atomicAdd(&lh[2][q], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += blockDim.x)
        if ((&lh[0][0])[i]) atomicAdd(&hist[i], (&lh[0][0])[i]);
}

hipError_t launch_live_count_hist(const Entry *table, const u32 *live, const Ctl *ctl, u32 *hist, hipStream_t st) {
    hipError_t e = hipMemsetAsync(hist, 0, 768 * sizeof(u32), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_live_count_hist, dim3(512), dim3(256), 0, st, table, live, ctl, hist);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// table maintenance
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_table(Entry *table, u64 cap) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride) {
        Entry e;
        e.hash = EMPTY64;
        e.kmer = EMPTY64;
        e.pos = EMPTY64;
        e.count = 0;
        e.extra = 0;
        table[i] = e;
    }
}

hipError_t launch_fill_table(Entry *table, u64 cap, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_table, dim3(2048), dim3(256), 0, st, table, cap);
    return hipGetLastError();
}

__device__ __forceinline__ void clear_entry(Entry *e) {
    e->hash = EMPTY64;
    e->kmer = EMPTY64;
    e->pos = EMPTY64;
    e->count = 0;
    e->extra = 0;
}

// reset support: clear exactly the slots this run touched; a full sweep only if the dropped-slot list overflowed
__global__ void k_clear_slots(Entry *table, u64 cap, const u32 *live, const u32 *dead, const Ctl *ctl) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 t0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nl = ctl->n_live, nd = ctl->n_dead;
    u64 *khi = ctl->kmer_hi; // K > 32: the k-mers' high words, "not written" = EMPTY64
    if (nd == 0xFFFFFFFFu) {
        for (u64 i = t0; i < cap; i += stride) {
            clear_entry(&table[i]);
            if (khi) khi[i] = EMPTY64;
        }
        return;
    }
    for (u64 i = t0; i < nl; i += stride) {
        clear_entry(&table[live[i]]);
        if (khi) khi[live[i]] = EMPTY64;
    }
    for (u64 i = t0; i < nd; i += stride) {
        clear_entry(&table[dead[i]]);
        if (khi) khi[dead[i]] = EMPTY64;
    }
}

hipError_t launch_clear_slots(Entry *table, u64 cap, const u32 *live, const u32 *dead, const Ctl *ctl, hipStream_t st) {
    hipLaunchKernelGGL(k_clear_slots, dim3(1024), dim3(256), 0, st, table, cap, live, dead, ctl);
    return hipGetLastError();
}

__global__ void k_set_table(Ctl *ctl, Entry *table, u32 *live, CollRec *clog, u32 cap, u32 live_cap, u32 clog_cap,
                            u32 *shard_cnt, u32 *shard_buf, u32 shard_cap, u64 *kmer_hi) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctl->kmer_hi = kmer_hi;
        ctl->table = table;
        ctl->live = live;
        ctl->clog = clog;
        ctl->cap = cap;
        ctl->live_cap = live_cap;
        ctl->clog_cap = clog_cap;
        ctl->shard_cnt = shard_cnt;
        ctl->shard_buf = shard_buf;
        ctl->shard_cap = shard_cap;
        ctl->pad1 = 0;
    }
    for (int i = threadIdx.x; i < N_SHARDS; i += blockDim.x) shard_cnt[i * SHARD_STRIDE] = 0;
}

hipError_t launch_set_table(Ctl *ctl, Entry *table, u32 *live, CollRec *clog, u32 cap, u32 live_cap, u32 clog_cap,
                            u32 *shard_cnt, u32 *shard_buf, u32 shard_cap, u64 *kmer_hi, hipStream_t st) {
    hipLaunchKernelGGL(k_set_table, dim3(1), dim3(256), 0, st, ctl, table, live, clog, cap, live_cap, clog_cap, shard_cnt,
                       shard_buf, shard_cap, kmer_hi);
    return hipGetLastError();
}

// Append the per-shard lists of newly inserted slots to the flat live list (block s copies shard s), then
// k_live_commit publishes the new length and rewinds the shard cursors.
__global__ __launch_bounds__(256) void k_live_flatten(Ctl *ctl) {
    __shared__ u32 s_pre, s_cnt;
    const u32 s = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        s_pre = 0;
        s_cnt = 0;
    }
    __syncthreads();
    u32 c = tid < (u32)N_SHARDS ? ctl->shard_cnt[tid * SHARD_STRIDE] : 0u;
    if (c > ctl->shard_cap) c = ctl->shard_cap; // overflow already flagged by the inserter
    if (tid < s) atomicAdd(&s_pre, c);
    if (tid == s) s_cnt = c;
    __syncthreads();
    const u32 base = ctl->n_live + s_pre, n = s_cnt;
    u32 *live = ctl->live;
    const u32 *src = ctl->shard_buf + (size_t)s * ctl->shard_cap;
    const u32 live_cap = ctl->live_cap;
    for (u32 i = tid; i < n; i += blockDim.x) {
        if (base + i < live_cap) live[base + i] = src[i];
        else atomicExch(&ctl->overflow, 1u);
    }
}

__global__ __launch_bounds__(256) void k_live_commit(Ctl *ctl) {
    __shared__ u32 s_tot;
    const u32 tid = threadIdx.x;
    if (tid == 0) s_tot = 0;
    __syncthreads();
    u32 c = tid < (u32)N_SHARDS ? ctl->shard_cnt[tid * SHARD_STRIDE] : 0u;
    if (c > ctl->shard_cap) c = ctl->shard_cap;
    if (c) atomicAdd(&s_tot, c);
    __syncthreads();
    if (tid < (u32)N_SHARDS) ctl->shard_cnt[tid * SHARD_STRIDE] = 0;
    if (tid == 0 && s_tot) {
        ctl->inserted_total += s_tot;
        u32 n = ctl->n_live + s_tot;
        if (n > ctl->live_cap) n = ctl->live_cap;
        ctl->n_live = n;
        ctl->sorted = 0;
    }
}

hipError_t launch_live_flatten(Ctl *ctl, hipStream_t st) {
    hipLaunchKernelGGL(k_live_flatten, dim3(N_SHARDS), dim3(256), 0, st, ctl);
    hipLaunchKernelGGL(k_live_commit, dim3(1), dim3(256), 0, st, ctl);
    return hipGetLastError();
}

__device__ __forceinline__ void init_ctl_dev(Ctl *ctl, u64 tau0, u32 keep_text_bases, u64 sel_size, u64 tau_floor, u32 hist_on) {
    if (threadIdx.x == 0) {
        ctl->tau = tau0;
        ctl->inserted_total = 0;
        ctl->n_live = 0;
        ctl->overflow = 0;
        ctl->n_coll = 0;
        ctl->need_big = 0;
        ctl->sorted = 1;
        ctl->spec_ok = 0;
        ctl->n_dead = 0;
        ctl->hist_on = hist_on;
        ctl->sel_size = sel_size;
        ctl->tau_floor = tau_floor;
        ctl->next_unit = 0;
        ctl->left_in_pos = 0;
        ctl->n_left_out = 0;
        ctl->stopped = 0;
        ctl->soft_limit = 0xFFFFFFFFu;
        ctl->shard_soft = 0xFFFFFFFFu;
        ctl->read_first = 0;
        ctl->dbg_flush_cycles = ctl->dbg_flush_calls = ctl->dbg_flush_entries = ctl->dbg_wave_cycles = 0;
        ctl->sp_count = 0;
        ctl->sp_extra = 0;
        ctl->sp_pos = EMPTY64;
        ctl->sp_kmer = EMPTY64;
        if (!keep_text_bases) ctl->text_bases = 0;
    }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        ctl->kmer_counts[i] = 0;
        ctl->hist[i] = 0;
    }
}

__global__ void k_init_ctl(Ctl *ctl, u64 tau0, u32 keep_text_bases, u64 sel_size, u64 tau_floor, u32 hist_on) {
    if (blockIdx.x == 0) init_ctl_dev(ctl, tau0, keep_text_bases, sel_size, tau_floor, hist_on);
}

// fh_reset of a sketch whose live and dropped-slot lists are short (the host knows: it has the control block of the
// finished sketch): clear exactly those slots and re-initialise the control block in one single-workgroup launch
__global__ __launch_bounds__(1024) void k_reset_small(Entry *table, const u32 *live, const u32 *dead, Ctl *ctl, u64 tau0, u64 sel_size,
                                                      u64 tau_floor, u32 hist_on) {
    const u32 nl = ctl->n_live, nd = ctl->n_dead;
    u64 *khi = ctl->kmer_hi;
    for (u32 i = threadIdx.x; i < nl; i += blockDim.x) {
        clear_entry(&table[live[i]]);
        if (khi) khi[live[i]] = EMPTY64;
    }
    if (nd != 0xFFFFFFFFu) // (the host does not choose this kernel when the dropped-slot list overflowed)
        for (u32 i = threadIdx.x; i < nd; i += blockDim.x) {
            clear_entry(&table[dead[i]]);
            if (khi) khi[dead[i]] = EMPTY64;
        }
    __syncthreads(); // every thread has read the two counts
    init_ctl_dev(ctl, tau0, 0u, sel_size, tau_floor, hist_on);
}

hipError_t launch_reset_small(Entry *table, const u32 *live, const u32 *dead, Ctl *ctl, u64 tau0, u64 sel_size, u64 tau_floor,
                              u32 hist_on, hipStream_t st) {
    hipLaunchKernelGGL(k_reset_small, dim3(1), dim3(1024), 0, st, table, live, dead, ctl, tau0, sel_size, tau_floor, hist_on);
    return hipGetLastError();
}

// new range: empty queue; relaunch of a stopped range: keep next_chunk, swap leftover lists
// (set_tau: the threshold of a speculative range rides along instead of a k_set_tau launch of its own; gate: the range was
//  queued behind a speculation and must leave the queue of that range alone unless it held -- Ctl::spec_ok;
//  first_total: the units the first launch's waves start on without asking the queue -- it begins behind them)
__global__ void k_queue_reset(Ctl *ctl, u32 new_range, u32 soft_limit, u32 read_first, u32 set_tau, u64 tau, u32 gate, u32 first_total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (gate && ctl->spec_ok == 0u) return;
        if (set_tau) ctl->tau = tau;
        if (new_range) ctl->next_unit = first_total;
        ctl->left_in_pos = 0;
        ctl->n_left_out = 0;
        ctl->soft_limit = soft_limit;
        ctl->read_first = read_first;
        // inserts are spread over N_SHARDS lists: stop when one of them has taken its share of the room
        const u32 nl = ctl->n_live;
        const u32 room = soft_limit > nl ? soft_limit - nl : 0u;
        u32 share = room / (u32)N_SHARDS;
        // (with the histogram on, refresh_tau sees the exact total every few inserts and stops the launch on it: the
        // per-list limit, which a lucky list reaches long before the total does, is then only a coarse guard)
        if (ctl->hist_on) share *= 4u;
        ctl->shard_soft = share ? share : 1u;
        // the live set may already sit at/above the limit (nothing pruned since): stop at once
        ctl->stopped = nl >= soft_limit ? 1u : 0u;
    }
}

hipError_t launch_queue_reset(Ctl *ctl, u32 new_range, u32 soft_limit, u32 read_first, hipStream_t st, bool set_tau, u64 tau, bool gate,
                              u32 first_total) {
    hipLaunchKernelGGL(k_queue_reset, dim3(1), dim3(64), 0, st, ctl, new_range, soft_limit, read_first, set_tau ? 1u : 0u, tau,
                       gate ? 1u : 0u, first_total);
    return hipGetLastError();
}

__global__ void k_set_tau(Ctl *ctl, u64 tau) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctl->tau = tau;
}

// test hook (fh_debug_add_counts): every live entry's two counters moved up, so that a test reaches the saturation of the
// reported u32 counts (mash.rs:45-50) without 2^32 occurrences
__global__ void k_debug_add_counts(Entry *table, const u32 *live, const Ctl *ctl, u64 add_count, u64 add_extra) {
    const u32 n = ctl->n_live;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Entry &e = table[live[i]];
        e.count += add_count;
        e.extra += add_extra;
    }
}
hipError_t launch_debug_add_counts(Entry *table, const u32 *live, const Ctl *ctl, u64 add_count, u64 add_extra, hipStream_t st) {
    hipLaunchKernelGGL(k_debug_add_counts, dim3(256), dim3(256), 0, st, table, live, ctl, add_count, add_extra);
    return hipGetLastError();
}

// the threshold as a kernel argument: no host buffer to keep alive, no synchronisation
hipError_t launch_set_tau(Ctl *ctl, u64 tau, hipStream_t st) {
    hipLaunchKernelGGL(k_set_tau, dim3(1), dim3(64), 0, st, ctl, tau);
    return hipGetLastError();
}

hipError_t launch_init_ctl(Ctl *ctl, u64 tau0, hipStream_t st, bool keep_text_bases, u64 sel_size, u64 tau_floor, bool hist_on) {
    hipLaunchKernelGGL(k_init_ctl, dim3(1), dim3(256), 0, st, ctl, tau0, keep_text_bases ? 1u : 0u, sel_size, tau_floor, hist_on ? 1u : 0u);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// streaming-read probe (measured HBM read peak of the box; reporting only)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_read_probe(const uint4 *p, u64 n16, u32 *sink) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) { // four loads in flight per lane
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) {
        const uint4 a = p[i];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9E3779B9u) *sink = acc; // practically never: keeps the loads alive
}

// ------------------------------------------------------------------------------------------------
// records of one length?  (launch_seg_probe, fh_kernels.h)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool probe_is_base(uint8_t c) {
    c &= 0xDFu;
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'U';
}
__global__ __launch_bounds__(64) void k_seg_probe(const uint8_t *seq, u64 len, u32 *out) {
    const u32 lane = threadIdx.x;
    __shared__ u32 cand[4];
    {
        // the first four bytes that are no bases, within the longest record the segment kernel takes: one of them ends record 0
        // (the others are N's inside it).  Three bytes a lane, the lanes' verdicts as ballots.
        u32 n = 0;
        for (u32 base = 0; base < SEG_MAX_RECORD + 63u && n < 4u; base += 64u) { // (wave-uniform loop)
            const u32 i = base + lane;
            const bool bad = i < SEG_MAX_RECORD && (u64)i < len && !probe_is_base(seq[i]);
            u64 m = __builtin_amdgcn_ballot_w64(bad);
            while (m && n < 4u) {
                const u32 b = (u32)__builtin_ctzll(m);
                m &= m - 1ull;
                if (lane == 0) cand[n] = base + b + 1u;
                ++n;
            }
        }
        if (lane == 0)
            for (; n < 4u; ++n) cand[n] = 0u;
    }
    __syncthreads();
    u32 found = 0u;
    for (int ci = 0; ci < 4 && !found; ++ci) {
        const u32 S = cand[ci];
        if (S < SEG_MIN_STRIDE || S > SEG_MAX_RECORD) continue;
        const u64 nrec = len / S;
        if (nrec < 128ull || nrec * S != len) continue; // (a block of whole records)
        // the first 64 records, and 64 spread over the block: the byte where the breaker should be, and the one in front of
        // the next record's breaker being a base more often than not (so that a stream of breakers does not pass)
        const u64 r0 = lane, r1 = (nrec - 1ull) * lane / 63ull;
        const bool ok = !probe_is_base(seq[r0 * S + S - 1u]) && !probe_is_base(seq[r1 * S + S - 1u]);
        const bool inner = probe_is_base(seq[r0 * S]) || probe_is_base(seq[r1 * S + S / 2u]);
        if (__builtin_amdgcn_ballot_w64(ok) == ~0ull && __popcll(__builtin_amdgcn_ballot_w64(inner)) >= 32) found = S;
    }
    // how dense the record ends are: the bytes that are no bases among 64 x 64 looked at all over the block (out[1], of 4096) --
    // records of many lengths but few hundred bases each are what the segment kernel's work-item form is for (SEG_RAGGED)
    u32 nb = 0;
    if (len >= 8192ull) {
        const u64 at = (len - 64ull) * lane / 63ull;
        for (u32 i = 0; i < 64u; ++i) nb += probe_is_base(seq[at + i]) ? 0u : 1u;
    }
    for (int off = 32; off > 0; off >>= 1) nb += __shfl_xor(nb, off);
    if (lane == 0) {
        out[1] = nb;
        out[0] = found;
        __threadfence_system();
    }
}
hipError_t launch_seg_probe(const uint8_t *seq, u64 len, u32 *out, hipStream_t st) {
    hipLaunchKernelGGL(k_seg_probe, dim3(1), dim3(64), 0, st, seq, len, out);
    return hipGetLastError();
}

hipError_t launch_read_probe(const void *p, u64 bytes, u32 *sink, hipStream_t st) {
    hipLaunchKernelGGL(k_read_probe, dim3(256 * 8), dim3(256), 0, st, (const uint4 *)p, bytes / 16, sink);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// synthetic inputs (not part of the timed path)
// ------------------------------------------------------------------------------------------------
__global__ void k_synth_genome(uint8_t *out, u64 len, u64 seed) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) out[i] = synth_genome_base(seed, i);
}

__global__ void k_synth_reads(uint8_t *out, const uint8_t *genome, u64 genome_len, u64 first_read, u64 n_reads,
                              u32 read_len, u64 seed, u32 sub_ppm, u32 n_ppm) {
    const u64 rec = (u64)read_len + 1;
    const u64 total = n_reads * rec;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const u64 r = i / rec;
        const u32 j = (u32)(i - r * rec);
        out[i] = synth_read_byte(genome, genome_len, first_read + r, j, read_len, seed, sub_ppm, n_ppm);
    }
}

hipError_t launch_synth_genome(uint8_t *out, u64 len, u64 seed, hipStream_t st) {
    hipLaunchKernelGGL(k_synth_genome, dim3(1024), dim3(256), 0, st, out, len, seed);
    return hipGetLastError();
}

hipError_t launch_synth_reads(uint8_t *out, const uint8_t *genome, u64 genome_len, u64 first_read, u64 n_reads,
                              u32 read_len, u64 seed, u32 sub_ppm, u32 n_ppm, hipStream_t st) {
    hipLaunchKernelGGL(k_synth_reads, dim3(4096), dim3(256), 0, st, out, genome, genome_len, first_read, n_reads,
                       read_len, seed, sub_ppm, n_ppm);
    return hipGetLastError();
}

} // namespace fh
