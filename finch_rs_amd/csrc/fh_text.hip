// fh_text.hip -- K1: FASTQ text -> packed sequence stream on the device.
//
// Replaces, for plain 4-line FASTQ, the host side of needletail's record splitting (lib/src/lib.rs:60-68): the host
// only reads raw file bytes into a pinned buffer and cuts them at a record boundary; which bytes are sequence
// is decided here.  A byte belongs to a sequence line iff the number of newlines before it is 1 mod 4.
// Sequence bytes are copied (CR dropped), the newline that ends a sequence line becomes the '\0' record
// breaker, everything else (headers, '+' lines, qualities) is dropped -- a stream compaction:
//   pass A  newlines per 4 KiB block            -> exclusive scan (one workgroup)
//   pass B  kept bytes per block (needs A)      -> exclusive scan
//   pass C  recompute flags, block-local scan, scatter the kept bytes
// Each pass streams the chunk once with 16-byte loads; at 3 reads + ~0.5 writes per text byte this is HBM bound
// and costs a few percent of the time the sketch kernel spends on the same reads.
#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"

namespace fh {

constexpr int TB = 256;          // threads per block
constexpr int BPT = 16;          // bytes per thread
constexpr int BLK_BYTES = TB * BPT;

__device__ __forceinline__ void load16(const uint8_t *text, u64 len, u64 off, uint8_t b[16]) {
    if (off + 16 <= len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + off);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (off + i < len) ? text[off + i] : (uint8_t)0xFF; // 0xFF: never kept
    }
}

// block-wide exclusive scan of one u32 per thread (256 threads); returns the exclusive prefix, total in *tot
__device__ __forceinline__ u32 block_exscan(u32 v, u32 *smem /* >= 4 */, u32 *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    u32 wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < TB / 64; ++w) {
        const u32 s = smem[w];
        if (w < wave) wbase += s;
        total += s;
    }
    __syncthreads();
    *tot = total;
    return wbase + inc - v;
}

__global__ __launch_bounds__(TB) void k1_count_newlines(const uint8_t *text, u64 len, u32 *blk_nl) {
    __shared__ u32 sm[4];
    const u64 off = ((u64)blockIdx.x * TB + threadIdx.x) * BPT;
    uint8_t b[16];
    load16(text, len, off, b);
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c += (b[i] == '\n');
    u32 tot;
    (void)block_exscan(c, sm, &tot);
    if (threadIdx.x == 0) blk_nl[blockIdx.x] = tot;
}

// single workgroup exclusive scan of n values (n up to a few 100k), in place; total -> *total_out
__global__ __launch_bounds__(1024) void k1_scan(u32 *vals, u32 n, u32 *total_out) {
    __shared__ u32 sm[16];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < n ? vals[i] : 0u;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u32 inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) sm[wave] = inc;
        __syncthreads();
        u32 wbase = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            const u32 s = sm[w];
            if (w < wave) wbase += s;
            total += s;
        }
        const u32 c = carry;
        if (i < n) vals[i] = c + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// per-byte decision for the 16 bytes of one thread; line0 = index of the line the first byte is on
struct Keep16 {
    u32 mask;  // bit i: byte i is emitted
    u32 zmask; // bit i: byte i is emitted as the '\0' breaker (end of a sequence line)
    u32 n_nl;
    u32 bad;   // structure violation seen (header not '@' / separator not '+')
};

__device__ __forceinline__ Keep16 decide16(const uint8_t b[16], u32 line0, bool first_is_line_start) {
    Keep16 k{0u, 0u, 0u, 0u};
    u32 line = line0;
    bool at_start = first_is_line_start;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint8_t c = b[i];
        const u32 ph = line & 3u;
        if (c == 0xFF) { // past the end of the chunk
            at_start = false;
            continue;
        }
        if (at_start) {
            if (ph == 0u && c != '@' && c != '\n' && c != '\r') k.bad = 1u;
            if (ph == 2u && c != '+') k.bad = 1u;
        }
        if (c == '\n') {
            if (ph == 1u) {
                k.mask |= 1u << i;
                k.zmask |= 1u << i;
            }
            line++;
            k.n_nl++;
            at_start = true;
        } else {
            if (ph == 1u && c != '\r') k.mask |= 1u << i;
            at_start = false;
        }
    }
    return k;
}

template <bool WRITE>
__global__ __launch_bounds__(TB) void k1_pack(const uint8_t *text, u64 len, const u32 *blk_nl_ex, u32 *blk_keep,
                                              const u32 *blk_keep_ex, uint8_t *out, Ctl *ctl, u32 *err) {
    __shared__ u32 sm[4];
    const u64 off = ((u64)blockIdx.x * TB + threadIdx.x) * BPT;
    uint8_t b[16];
    load16(text, len, off, b);
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c += (b[i] == '\n');
    u32 tot;
    const u32 nl_before = blk_nl_ex[blockIdx.x] + block_exscan(c, sm, &tot);
    const bool starts_line = (off == 0) || (off < len + 1 && off > 0 && text[off - 1] == '\n');
    const Keep16 k = decide16(b, nl_before, starts_line);
    if (k.bad) atomicExch(err, 1u);
    const u32 nkeep = (u32)__popc(k.mask);
    u32 ktot;
    const u32 kpre = block_exscan(nkeep, sm, &ktot);
    if (!WRITE) {
        if (threadIdx.x == 0) blk_keep[blockIdx.x] = ktot;
        return;
    }
    u64 o = (u64)blk_keep_ex[blockIdx.x] + kpre;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if ((k.mask >> i) & 1u) out[o++] = ((k.zmask >> i) & 1u) ? (uint8_t)0 : b[i];
    }
    // bases = emitted bytes that are not breakers (what total_bases counts for FASTQ, mash.rs:72)
    const u32 nb = nkeep - (u32)__popc(k.zmask);
    u32 btot;
    (void)block_exscan(nb, sm, &btot);
    if (threadIdx.x == 0 && btot) atomicAdd((unsigned long long *)&ctl->text_bases, (unsigned long long)btot);
}

hipError_t launch_fastq_pack(const uint8_t *text, u64 len, uint8_t *out, u32 *blk_a, u32 *blk_b, u32 *totals, Ctl *ctl,
                             u32 *err, hipStream_t st) {
    if (len == 0) return hipSuccess;
    const u32 nblk = (u32)((len + BLK_BYTES - 1) / BLK_BYTES);
    hipLaunchKernelGGL(k1_count_newlines, dim3(nblk), dim3(TB), 0, st, text, len, blk_a);
    hipLaunchKernelGGL(k1_scan, dim3(1), dim3(1024), 0, st, blk_a, nblk, totals);
    hipLaunchKernelGGL((k1_pack<false>), dim3(nblk), dim3(TB), 0, st, text, len, (const u32 *)blk_a, blk_b,
                       (const u32 *)nullptr, (uint8_t *)nullptr, ctl, err);
    hipLaunchKernelGGL(k1_scan, dim3(1), dim3(1024), 0, st, blk_b, nblk, totals + 1);
    hipLaunchKernelGGL((k1_pack<true>), dim3(nblk), dim3(TB), 0, st, text, len, (const u32 *)blk_a, (u32 *)nullptr,
                       (const u32 *)blk_b, out, ctl, err);
    return hipGetLastError();
}

} // namespace fh
