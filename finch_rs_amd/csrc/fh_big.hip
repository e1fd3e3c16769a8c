// fh_big.hip -- bottom-n selection for live sets that do not fit the in-LDS sort (kmers_to_sketch in the
// millions: the CLI's oversketch x200, mod cli.rs:187-192; scaled sketches, scaled.rs).
//
// Same rule as k3_prune_small (fh_kernels.hip), but the sort of (hash, slot) pairs is device-wide:
// gather keys -> radix sort (rocPRIM's device radix sort is used as a plain library primitive here; this
// step runs a handful of times per stream, never in the per-base hot loop) -> pick tau / keep -> write the
// (now sorted) live list back and append the dropped slots to the dead list.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

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

hipError_t big_sort_tmp_bytes(u32 M, size_t *bytes) {
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, (const u32 *)nullptr,
                                             (u32 *)nullptr, (size_t)M, 0, 64, nullptr);
    *bytes = b;
    return e;
}

hipError_t launch_big_prune(Entry *table, u32 *live, u32 *dead, u32 dead_cap, Ctl *ctl, u32 M, u32 n_dead_now, u32 kind,
                            u64 size, u64 max_hash, u64 *keys_a, u64 *keys_b, u32 *slots_a, u32 *slots_b, void *tmp,
                            size_t tmp_bytes, u32 *keep_dev, hipStream_t st) {
    if (M == 0) return hipSuccess;
    const int blocks = (int)((M + 255u) / 256u < 4096u ? (M + 255u) / 256u : 4096u);
    hipLaunchKernelGGL(k_big_gather_keys, dim3(blocks), dim3(256), 0, st, table, live, M, keys_a, slots_a);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, (const u64 *)keys_a, keys_b, (const u32 *)slots_a, slots_b,
                                             (size_t)M, 0, 64, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_big_select, dim3(1), dim3(64), 0, st, keys_b, M, ctl, kind, size, max_hash, keep_dev);
    hipLaunchKernelGGL(k_big_writeback, dim3(blocks), dim3(256), 0, st, slots_b, M, keep_dev, live, dead, dead_cap, ctl,
                       n_dead_now);
    return hipGetLastError();
}

// ---- table growth / garbage compaction: move the live entries into a fresh table ----
__global__ void k_rehash(const Entry *src, const u32 *src_live, u32 M, Entry *dst, u32 dst_cap, u32 *dst_live, Ctl *ctl) {
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
        dst[slot].pos = e.pos;
        dst[slot].count = e.count;
        dst[slot].extra = e.extra;
        dst_live[i] = slot; // keeps the order (and sortedness) of the live list
    }
}

hipError_t launch_rehash(const Entry *src, const u32 *src_live, u32 M, Entry *dst, u32 dst_cap, u32 *dst_live, Ctl *ctl,
                         hipStream_t st) {
    if (M == 0) return hipSuccess;
    const int blocks = (int)((M + 255u) / 256u < 4096u ? (M + 255u) / 256u : 4096u);
    hipLaunchKernelGGL(k_rehash, dim3(blocks), dim3(256), 0, st, src, src_live, M, dst, dst_cap, dst_live, ctl);
    return hipGetLastError();
}

} // namespace fh
