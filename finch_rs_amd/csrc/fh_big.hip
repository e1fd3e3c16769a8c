// fh_big.hip -- bottom-n selection for live sets that do not fit the in-LDS sort (kmers_to_sketch in the
// millions: the CLI's oversketch x200, mod cli.rs:187-192; scaled sketches, scaled.rs).
//
// Same rule as k3_prune_small (fh_kernels.hip), but the sort of (hash, slot) pairs is device-wide:
// gather keys -> radix sort (below: eight stable passes of eight bits; this step runs a handful of times per
// stream, never in the per-base hot loop) -> pick tau / keep -> write the (now sorted) live list back and append
// the dropped slots to the dead list.
#include <cstring>

#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"

namespace fh {

__global__ void k_big_gather_keys(const Entry *table, const u32 *live, u32 M, u64 *keys, u32 *slots) {
    const u32 stride = gridDim.x * blockDim.x;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
        const u32 s = live[i];
        keys[i] = table[s].hash;
        slots[i] = s;
    }
}

// one thread: same decision as the tail of k3_prune_small
__global__ void k_big_select(const u64 *keys, u32 M, Ctl *ctl, u32 kind, u64 size, u64 max_hash, u32 *keep_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 keep;
    u64 tau;
    if (kind == 0u) {
        if ((u64)M >= size) {
            keep = (u32)size;
            tau = size ? keys[size - 1] : 0ull;
        } else {
            keep = M;
            tau = EMPTY64;
        }
    } else {
        u32 lo = 0, hi = M;
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (keys[mid] <= max_hash) lo = mid + 1;
            else hi = mid;
        }
        const u32 n_le = lo;
        if ((u64)n_le >= size) {
            keep = n_le;
            tau = max_hash;
        } else if ((u64)M >= size) {
            keep = (u32)size;
            tau = keys[size - 1];
        } else {
            keep = M;
            tau = (size != 0) ? EMPTY64 : max_hash;
        }
    }
    *keep_out = keep;
    ctl->tau = tau;
    ctl->n_live = keep;
    ctl->sorted = 1u;
    ctl->need_big = 0u;
}

__global__ void k_big_writeback(const u32 *slots_sorted, u32 M, const u32 *keep_p, u32 *live, u32 *dead, u32 dead_cap,
                                Ctl *ctl, u32 nd0) {
    const u32 keep = *keep_p;
    const u32 ndrop = M - keep;
    const bool fits = nd0 != 0xFFFFFFFFu && nd0 <= dead_cap && ndrop <= dead_cap - nd0;
    const u32 stride = gridDim.x * blockDim.x;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
        const u32 s = slots_sorted[i];
        if (i < keep) live[i] = s;
        else if (fits) dead[nd0 + (i - keep)] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->n_dead = fits ? nd0 + ndrop : 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------------------------------
// Radix sort of (u64 key, u32 value) pairs: least significant byte first, every pass stable.
//   k_rs_hist      digit counts of each 2048-pair tile                    -> hist[digit][tile]
//   k_rs_scan_rows exclusive scan along every digit's row, row totals     -> hist (in place), tot[digit]
//   k_rs_scan_tot  exclusive scan of the 256 totals
//   k_rs_scatter   the tile again, in the same order: a pair's place is tot[d] + hist[d][tile] + the pairs of digit d before
//                  it in the tile.  That last term keeps the pass stable: the tile is walked 256 pairs at a time (thread
//                  order = pair order), lanes with the same digit find each other with eight ballots, a lane's rank in its
//                  wave is the population count of the lanes below it, waves are ranked through LDS.
// Eight passes return the data to the buffers it came in.
// ---------------------------------------------------------------------------------------------------------------------
constexpr u32 RS_TILE = 2048;

__global__ __launch_bounds__(256) void k_rs_hist(const u64 *keys, u32 M, int shift, u32 *hist, u32 ntile) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const u32 base = blockIdx.x * RS_TILE;
#pragma unroll
    for (u32 j = 0; j < RS_TILE / 256; ++j) {
        const u32 i = base + j * 256 + threadIdx.x;
        if (i < M) atomicAdd(&h[(u32)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * ntile + blockIdx.x] = h[threadIdx.x];
}

__device__ __forceinline__ u32 rs_wave_incl_scan(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// block d: exclusive scan of hist[d * ntile, +ntile) in place, its total to tot[d]
__global__ __launch_bounds__(256) void k_rs_scan_rows(u32 *hist, u32 ntile, u32 *tot) {
    __shared__ u32 sm[4];
    __shared__ u32 carry;
    u32 *row = hist + (size_t)blockIdx.x * ntile;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 base = 0; base < ntile; base += 256) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < ntile ? row[i] : 0u;
        const u32 inc = rs_wave_incl_scan(v);
        if (lane == 63) sm[wave] = inc;
        __syncthreads();
        u32 wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const u32 x = sm[w];
            if (w < wave) wbase += x;
            total += x;
        }
        const u32 c = carry;
        if (i < ntile) row[i] = c + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void k_rs_scan_tot(u32 *tot) {
    __shared__ u32 sm[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 v = tot[threadIdx.x];
    const u32 inc = rs_wave_incl_scan(v);
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    u32 wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += sm[w];
    tot[threadIdx.x] = wbase + inc - v;
}

__global__ __launch_bounds__(256) void k_rs_scatter(const u64 *keys, const u32 *vals, u32 M, int shift, const u32 *hist, const u32 *tot,
                                                    u32 ntile, u64 *keys_out, u32 *vals_out) {
    __shared__ u32 run[256];    // where the next pair of each digit goes
    __shared__ u32 cnt[4][256]; // pairs of each digit in each wave of the current 256
    const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    run[t] = tot[t] + hist[t * ntile + blockIdx.x];
    const u32 base = blockIdx.x * RS_TILE;
    for (u32 j = 0; j < RS_TILE / 256; ++j) {
#pragma unroll
        for (int w = 0; w < 4; ++w) cnt[w][t] = 0;
        __syncthreads();
        const u32 i = base + j * 256 + t;
        const bool valid = i < M;
        const u64 key = valid ? keys[i] : 0ull;
        const u32 val = valid ? vals[i] : 0u;
        const u32 d = (u32)(key >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bit = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bit : ~bit;
        }
        const u32 rank = (u32)__popcll(same & ((1ull << lane) - 1ull));
        if (valid && rank == 0u) cnt[wave][d] = (u32)__popcll(same); // (the lowest lane of each digit's group)
        __syncthreads();
        if (valid) {
            u32 off = 0;
            for (u32 w = 0; w < wave; ++w) off += cnt[w][d];
            const u32 pos = run[d] + off + rank;
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        run[t] += cnt[0][t] + cnt[1][t] + cnt[2][t] + cnt[3][t];
        __syncthreads();
    }
}

static inline u32 rs_ntile(u32 M) { return (M + RS_TILE - 1) / RS_TILE; }

hipError_t big_sort_tmp_bytes(u32 M, size_t *bytes) {
    *bytes = ((size_t)256 * rs_ntile(M ? M : 1u) + 256) * sizeof(u32);
    return hipSuccess;
}

// ascending by key; the sorted pairs end up where they came from (keys, vals); keys_tmp / vals_tmp are scratch
static hipError_t sort_pairs(void *tmp, size_t tmp_bytes, u64 *keys, u64 *keys_tmp, u32 *vals, u32 *vals_tmp, u32 M, hipStream_t st) {
    const u32 ntile = rs_ntile(M);
    if (tmp_bytes < ((size_t)256 * ntile + 256) * sizeof(u32)) return hipErrorInvalidValue;
    u32 *hist = (u32 *)tmp, *tot = hist + (size_t)256 * ntile;
    u64 *ka = keys, *kb = keys_tmp;
    u32 *va = vals, *vb = vals_tmp;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 8 * pass;
        hipLaunchKernelGGL(k_rs_hist, dim3(ntile), dim3(256), 0, st, (const u64 *)ka, M, shift, hist, ntile);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(256), dim3(256), 0, st, hist, ntile, tot);
        hipLaunchKernelGGL(k_rs_scan_tot, dim3(1), dim3(256), 0, st, tot);
        hipLaunchKernelGGL(k_rs_scatter, dim3(ntile), dim3(256), 0, st, (const u64 *)ka, (const u32 *)va, M, shift, (const u32 *)hist,
                           (const u32 *)tot, ntile, kb, vb);
        u64 *tk = ka;
        ka = kb;
        kb = tk;
        u32 *tv = va;
        va = vb;
        vb = tv;
    }
    return hipGetLastError();
}

hipError_t launch_big_prune(Entry *table, u32 *live, u32 *dead, u32 dead_cap, Ctl *ctl, u32 M, u32 n_dead_now, u32 kind,
                            u64 size, u64 max_hash, u64 *keys_a, u64 *keys_b, u32 *slots_a, u32 *slots_b, void *tmp,
                            size_t tmp_bytes, u32 *keep_dev, hipStream_t st) {
    if (M == 0) return hipSuccess;
    const int blocks = (int)((M + 255u) / 256u < 4096u ? (M + 255u) / 256u : 4096u);
    hipLaunchKernelGGL(k_big_gather_keys, dim3(blocks), dim3(256), 0, st, table, live, M, keys_a, slots_a);
    hipError_t e = sort_pairs(tmp, tmp_bytes, keys_a, keys_b, slots_a, slots_b, M, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_big_select, dim3(1), dim3(64), 0, st, (const u64 *)keys_a, M, ctl, kind, size, max_hash, keep_dev);
    hipLaunchKernelGGL(k_big_writeback, dim3(blocks), dim3(256), 0, st, (const u32 *)slots_a, M, keep_dev, live, dead, dead_cap, ctl,
                       n_dead_now);
    return hipGetLastError();
}

// ---- in-stream variant: radix SELECT instead of a full sort ----
// Between launches only the new threshold and the partition of the live list matter, not the order: find the
// size-th smallest key with an MSB-first radix select (six passes of 11 bits over the gathered keys, 2048-bin
// histograms privatised in LDS), then split the live list in one pass.  ~0.2 ms for 6 M live entries where the
// 8-pass pair sort takes ~0.9 ms; fh_finish still sorts (to_vec order).
constexpr int SEL_BITS = 11, SEL_BINS = 1 << SEL_BITS, SEL_PASSES = 6; // 6 x 11 >= 64

struct SelState {
    u64 prefix;   // bits of the answer decided so far
    u64 decided;  // mask of those bits
    u64 k;        // rank still to find inside the current prefix class (1-based)
    u64 tau;      // result
    u32 mode;     // 0 = select running, 1 = tau/keep fixed without a select
    u32 keep;
    u32 lshift;   // keys are histogrammed as key << lshift (lshift = leading zeros of the largest key): admitted
                  // hashes are small numbers, unshifted they would all fall into a handful of top-bit bins
    u32 pad0[21];
    // counters hit by one atomic per workgroup, each on its own 128-byte line (same-line atomics serialise in L2)
    u32 n_le;     // scaled: keys <= max_hash
    u32 pad1[31];
    u64 max_key;
    u32 pad2[30];
    u32 cnt_live;
    u32 pad3[31];
    u32 cnt_dead;
    u32 pad4[31];
};
static_assert(sizeof(SelState) <= 1024, "select state must fit below the histogram");

__global__ __launch_bounds__(256) void k_sel_stats(const u64 *keys, u32 M, u64 max_hash, SelState *st) {
    __shared__ u32 sc[4];
    __shared__ u64 sm[4];
    u32 c = 0;
    u64 mx = 0;
    const u32 stride = gridDim.x * 256u;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < M; i += stride) {
        const u64 key = keys[i];
        c += key <= max_hash ? 1u : 0u;
        mx = key > mx ? key : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        c += __shfl_xor(c, off);
        const u64 o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        sc[threadIdx.x >> 6] = c;
        sm[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 ct = sc[0] + sc[1] + sc[2] + sc[3];
        u64 m = sm[0];
        for (int w = 1; w < 4; ++w) m = sm[w] > m ? sm[w] : m;
        if (ct) atomicAdd(&st->n_le, ct);
        atomicMax((unsigned long long *)&st->max_key, (unsigned long long)m);
    }
}

// one thread: the same decision table as k_big_select, minus the sorted-array lookups
__global__ void k_sel_init(SelState *st, u32 *hist, u32 M, u32 kind, u64 size, u64 max_hash) {
    for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) hist[i] = 0u;
    if (threadIdx.x != 0) return;
    st->prefix = 0;
    st->decided = 0;
    st->k = size;
    st->cnt_live = st->cnt_dead = 0;
    st->lshift = st->max_key ? (u32)__clzll((long long)st->max_key) : 0u;
    if (kind == 0u) {
        if ((u64)M >= size) {
            st->mode = 0u;
            st->keep = (u32)size;
        } else {
            st->mode = 1u;
            st->keep = M;
            st->tau = EMPTY64;
        }
    } else {
        const u32 n_le = st->n_le;
        if ((u64)n_le >= size) {
            st->mode = 1u;
            st->keep = n_le;
            st->tau = max_hash;
        } else if ((u64)M >= size) {
            st->mode = 0u;
            st->keep = (u32)size;
        } else {
            st->mode = 1u;
            st->keep = M;
            st->tau = (size != 0) ? EMPTY64 : max_hash;
        }
    }
}

__global__ __launch_bounds__(256) void k_sel_hist(const u64 *keys, u32 M, const SelState *st, u32 *hist, int pass) {
    if (st->mode != 0u) return;
    __shared__ u32 sh[SEL_BINS];
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) sh[i] = 0u;
    __syncthreads();
    const int hi_bit = 64 - SEL_BITS * pass;                 // exclusive
    const int shift = hi_bit > SEL_BITS ? hi_bit - SEL_BITS : 0;
    const u32 mask = (u32)((1u << (hi_bit - shift)) - 1u);
    const u64 prefix = st->prefix, decided = st->decided;
    const u32 lshift = st->lshift;
    const u32 stride = gridDim.x * 256u;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < M; i += stride) {
        const u64 key = keys[i] << lshift;
        if (((key ^ prefix) & decided) == 0ull) atomicAdd(&sh[(u32)(key >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_BINS; i += 256)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// one workgroup: pick the bin holding rank k, extend the prefix, clear the histogram for the next pass
__global__ __launch_bounds__(256) void k_sel_scan(SelState *st, u32 *hist, int pass) {
    if (st->mode != 0u) return;
    __shared__ u32 part[256];
    const int hi_bit = 64 - SEL_BITS * pass;
    const int shift = hi_bit > SEL_BITS ? hi_bit - SEL_BITS : 0;
    const int nbits = hi_bit - shift;
    const int per = SEL_BINS / 256; // 8 consecutive bins per thread
    u32 loc[SEL_BINS / 256];
    u32 sum = 0;
    for (int j = 0; j < per; ++j) {
        loc[j] = hist[threadIdx.x * per + j];
        sum += loc[j];
        hist[threadIdx.x * per + j] = 0u;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 k = st->k, acc = 0;
        int t = 0;
        for (; t < 256; ++t) {
            if (acc + part[t] >= k) break;
            acc += part[t];
        }
        part[0] = (u32)t;          // owner thread
        st->k = k - acc;           // rank inside the owner's 8 bins (finished below)
    }
    __syncthreads();
    if ((int)threadIdx.x == (int)part[0]) {
        u64 k = st->k, acc = 0;
        int j = 0;
        for (; j < per; ++j) {
            if (acc + loc[j] >= k) break;
            acc += loc[j];
        }
        const u64 bin = (u64)(threadIdx.x * per + j);
        st->k = k - acc;
        st->prefix |= bin << shift;
        st->decided |= ((nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull)) << shift);
    }
}

__global__ void k_sel_commit(SelState *st, Ctl *ctl, u32 *keep_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->mode == 0u) st->tau = st->prefix >> st->lshift;
    *keep_out = st->keep;
    ctl->tau = st->tau;
    ctl->n_live = st->keep;
    ctl->sorted = 0u;
    ctl->need_big = 0u;
}

// keys are distinct, so exactly `keep` of them are <= tau (scaled/fixed modes: by construction of keep).
// Each workgroup owns a contiguous chunk: count, reserve its output ranges with ONE atomic per list, then write.
__global__ __launch_bounds__(256) void k_sel_partition(const u64 *keys, const u32 *slots, u32 M, SelState *st, u32 *live,
                                                       u32 *dead, u32 dead_cap, Ctl *ctl, u32 nd0) {
    __shared__ u32 s_cnt[2], s_base[2], s_run[2];
    const u64 tau = st->tau; // mode 1 with tau == EMPTY64 keeps everything; keys never equal EMPTY64
    const u32 keep = st->keep;
    const u32 ndrop = M - keep;
    const bool fits = nd0 != 0xFFFFFFFFu && nd0 <= dead_cap && ndrop <= dead_cap - nd0;
    const u32 chunk = (((M + gridDim.x - 1) / gridDim.x) + 255u) & ~255u;
    const u32 c0 = blockIdx.x * chunk, c1 = (c0 + chunk < M) ? c0 + chunk : M;
    const u32 lane = threadIdx.x & 63u;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = s_run[threadIdx.x] = 0u;
    __syncthreads();
    u32 nk = 0, nd = 0;
    for (u32 i = c0 + threadIdx.x; i < c1; i += 256u) {
        const bool kp = keys[i] <= tau;
        nk += kp ? 1u : 0u;
        nd += kp ? 0u : 1u;
    }
    for (int off = 32; off > 0; off >>= 1) {
        nk += __shfl_xor(nk, off);
        nd += __shfl_xor(nd, off);
    }
    if (lane == 0) {
        if (nk) atomicAdd(&s_cnt[0], nk);
        if (nd) atomicAdd(&s_cnt[1], nd);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base[0] = s_cnt[0] ? atomicAdd(&st->cnt_live, s_cnt[0]) : 0u;
        s_base[1] = s_cnt[1] ? atomicAdd(&st->cnt_dead, s_cnt[1]) : 0u;
    }
    __syncthreads();
    const u32 bl = s_base[0], bd = nd0 + s_base[1];
    for (u32 i0 = c0; i0 < c1; i0 += 256u) { // uniform trip count inside the block: the ballots see whole waves
        const u32 i = i0 + threadIdx.x;
        const bool in = i < c1;
        const u64 key = in ? keys[i] : 0ull;
        const u32 slot = in ? slots[i] : 0u;
        const bool kp = in && key <= tau;
        const bool dr = in && !kp;
        const unsigned long long mk = __ballot(kp), md = __ballot(dr);
        u32 ok = 0, od = 0;
        if (lane == 0) {
            if (mk) ok = atomicAdd(&s_run[0], (u32)__popcll(mk));
            if (md) od = atomicAdd(&s_run[1], (u32)__popcll(md));
        }
        ok = __shfl(ok, 0);
        od = __shfl(od, 0);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (kp) live[bl + ok + (u32)__popcll(mk & below)] = slot;
        else if (dr && fits) dead[bd + od + (u32)__popcll(md & below)] = slot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->n_dead = fits ? nd0 + ndrop : 0xFFFFFFFFu;
}

hipError_t launch_big_prune_select(Entry *table, u32 *live, u32 *dead, u32 dead_cap, Ctl *ctl, u32 M, u32 n_dead_now,
                                   u32 kind, u64 size, u64 max_hash, u64 *keys, u32 *slots, void *scratch,
                                   u32 *keep_dev, hipStream_t st) {
    if (M == 0) return hipSuccess;
    SelState *state = (SelState *)scratch;
    u32 *hist = (u32 *)((char *)scratch + 1024);
    const int blocks = (int)((M + 255u) / 256u < 2048u ? (M + 255u) / 256u : 2048u);
    hipLaunchKernelGGL(k_big_gather_keys, dim3(blocks), dim3(256), 0, st, table, live, M, keys, slots);
    hipError_t e = hipMemsetAsync(state, 0, sizeof(SelState), st);
    if (e != hipSuccess) return e;
    const int few = blocks < 512 ? blocks : 512; // kernels that end in one atomic per workgroup
    hipLaunchKernelGGL(k_sel_stats, dim3(few), dim3(256), 0, st, keys, M, max_hash, state);
    hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(256), 0, st, state, hist, M, kind, size, max_hash);
    for (int pass = 0; pass < SEL_PASSES; ++pass) {
        hipLaunchKernelGGL(k_sel_hist, dim3(blocks), dim3(256), 0, st, keys, M, state, hist, pass);
        hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(256), 0, st, state, hist, pass);
    }
    hipLaunchKernelGGL(k_sel_commit, dim3(1), dim3(64), 0, st, state, ctl, keep_dev);
    hipLaunchKernelGGL(k_sel_partition, dim3(few), dim3(256), 0, st, keys, slots, M, state, live, dead, dead_cap, ctl,
                       n_dead_now);
    return hipGetLastError();
}

// ---- table growth / garbage compaction: move the live entries into a fresh table ----
__global__ void k_rehash(const Entry *src, const u32 *src_live, u32 M, Entry *dst, u32 dst_cap, u32 *dst_live, Ctl *ctl,
                         const u64 *src_hi, u64 *dst_hi) {
    typedef unsigned long long ull;
    const u32 stride = gridDim.x * blockDim.x;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
        const Entry e = src[src_live[i]];
        const u32 key32 = slot_key(e.hash);
        u32 slot = (u32)(((u64)key32 * (u64)dst_cap) >> 32);
        bool placed = false;
        for (int probe = 0; probe < MAX_PROBE; ++probe) {
            const ull old = atomicCAS((ull *)&dst[slot].hash, (ull)EMPTY64, (ull)e.hash);
            if (old == EMPTY64) {
                placed = true;
                break;
            }
            slot = (slot + 1u == dst_cap) ? 0u : slot + 1u;
        }
        if (!placed) {
            atomicExch(&ctl->overflow, 1u);
            continue;
        }
        dst[slot].kmer = e.kmer;
        if (dst_hi) dst_hi[slot] = src_hi[src_live[i]]; // K > 32: the k-mer's high word moves along
        dst[slot].pos = e.pos;
        dst[slot].count = e.count;
        dst[slot].extra = e.extra;
        dst_live[i] = slot; // keeps the order (and sortedness) of the live list
    }
}

hipError_t launch_rehash(const Entry *src, const u32 *src_live, u32 M, Entry *dst, u32 dst_cap, u32 *dst_live, Ctl *ctl,
                         const u64 *src_hi, u64 *dst_hi, hipStream_t st) {
    if (M == 0) return hipSuccess;
    const int blocks = (int)((M + 255u) / 256u < 4096u ? (M + 255u) / 256u : 4096u);
    hipLaunchKernelGGL(k_rehash, dim3(blocks), dim3(256), 0, st, src, src_live, M, dst, dst_cap, dst_live, ctl, src_hi, dst_hi);
    return hipGetLastError();
}

} // namespace fh
