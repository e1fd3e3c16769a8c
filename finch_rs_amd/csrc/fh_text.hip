// fh_text.hip -- K1: FASTQ / FASTA text -> packed sequence stream on the device.
//
// --- FASTQ ---
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

// 16 bytes of one thread; returns how many of them lie inside the text (the rest read as 0)
__device__ __forceinline__ int load16(const uint8_t *text, u64 len, u64 off, uint8_t b[16]) {
    if (off + 16 <= len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + off);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        return 16;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = (off + i < len) ? text[off + i] : (uint8_t)0;
    return off < len ? (int)(len - off) : 0;
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
    const int nv = load16(text, len, off, b);
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c += (i < nv && b[i] == '\n');
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

__device__ __forceinline__ Keep16 decide16(const uint8_t b[16], int nv, u32 line0, bool first_is_line_start) {
    Keep16 k{0u, 0u, 0u, 0u};
    u32 line = line0;
    bool at_start = first_is_line_start;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint8_t c = b[i];
        const u32 ph = line & 3u;
        if (i >= nv) break; // past the end of the chunk
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
    const int nv = load16(text, len, off, b);
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c += (i < nv && b[i] == '\n');
    u32 tot;
    const u32 nl_before = blk_nl_ex[blockIdx.x] + block_exscan(c, sm, &tot);
    const bool starts_line = (off == 0) || (off < len + 1 && off > 0 && text[off - 1] == '\n');
    const Keep16 k = decide16(b, nv, nl_before, starts_line);
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

// ------------------------------------------------------------------------------------------------
// FASTA (multi-line): which bytes are sequence depends on whether the byte's line starts with '>'
// (needletail's FASTA reader: a record starts at a line that begins with '>', everything up to the next such
// line is its sequence region, lib.rs:60-68).  That is a question about the latest "event" at or before a byte:
// a newline (a line ended; the next line is sequence unless it begins with '>') or a '>' at a line start (a
// header line began).  Events are encoded as ((position + 1) << 1 | is_header_start); positions grow with the
// block index, so the latest event before a block is an exclusive prefix MAX over per-block values:
//   pass A  latest event per 4 KiB block          -> exclusive max-scan (one workgroup)
//   pass B  kept bytes per block (needs A)        -> exclusive sum-scan
//   pass C  recompute, block-local scans, scatter
// Kept: bytes of non-header lines except ' ', '\t', '\r', '\n' (normalize(false) drops them, mash.rs:73); every
// header start emits the '\0' record breaker.  start_state tells what the chunk begins in the middle of:
// 0 = a line start, 1 = a sequence line, 2 = a header line (chunks normally end at a line end).
__device__ __forceinline__ u32 block_exscan_max(u32 v, u32 *smem /* >= 4 */, u32 *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 t = __shfl_up(inc, off);
        if (lane >= off) inc = t > inc ? t : inc;
    }
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    u32 wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < TB / 64; ++w) {
        const u32 s = smem[w];
        if (w < wave) wbase = s > wbase ? s : wbase;
        total = s > total ? s : total;
    }
    __syncthreads();
    *tot = total;
    u32 ex = __shfl_up(inc, 1); // exclusive inside the wave
    if (lane == 0) ex = 0;
    return ex > wbase ? ex : wbase;
}

// latest event among the 16 bytes of one thread (0 = none); prev = the byte before the first one ('\n' at a line start)
__device__ __forceinline__ u32 last_event16(const uint8_t b[16], int nv, u64 off, uint8_t prev) {
    u32 ev = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint8_t c = b[i];
        if (i >= nv) break;
        const u32 code = ((u32)(off + i) + 1u) << 1;
        if (c == '\n') ev = code;
        else if (c == '>' && prev == '\n') ev = code | 1u;
        prev = c;
    }
    return ev;
}

__device__ __forceinline__ uint8_t byte_before(const uint8_t *text, u64 off, u32 start_state) {
    if (off == 0) return start_state == 0u ? (uint8_t)'\n' : (uint8_t)'x';
    return text[off - 1];
}

__global__ __launch_bounds__(TB) void k1f_events(const uint8_t *text, u64 len, u32 start_state, u32 *blk_ev) {
    __shared__ u32 sm[4];
    const u64 off = ((u64)blockIdx.x * TB + threadIdx.x) * BPT;
    uint8_t b[16];
    const int nv = load16(text, len, off, b);
    const u32 ev = nv ? last_event16(b, nv, off, byte_before(text, off, start_state)) : 0u;
    u32 tot;
    (void)block_exscan_max(ev, sm, &tot);
    if (threadIdx.x == 0) blk_ev[blockIdx.x] = tot;
}

// single workgroup exclusive max-scan of n values, in place
__global__ __launch_bounds__(1024) void k1_scan_max(u32 *vals, u32 n) {
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
            if (lane >= off) inc = t > inc ? t : inc;
        }
        if (lane == 63) sm[wave] = inc;
        __syncthreads();
        u32 wbase = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            const u32 s = sm[w];
            if (w < wave) wbase = s > wbase ? s : wbase;
            total = s > total ? s : total;
        }
        u32 ex = __shfl_up(inc, 1);
        if (lane == 0) ex = 0;
        ex = ex > wbase ? ex : wbase;
        const u32 c = carry;
        if (i < n) vals[i] = ex > c ? ex : c;
        __syncthreads();
        if (threadIdx.x == 0) carry = total > c ? total : c;
        __syncthreads();
    }
}

template <bool WRITE>
__global__ __launch_bounds__(TB) void k1f_pack(const uint8_t *text, u64 len, u32 start_state, const u32 *blk_ev_ex,
                                               u32 *blk_keep, const u32 *blk_keep_ex, uint8_t *out) {
    __shared__ u32 sm[4];
    const u64 off = ((u64)blockIdx.x * TB + threadIdx.x) * BPT;
    uint8_t b[16];
    const int nv = load16(text, len, off, b);
    uint8_t prev = byte_before(text, nv ? off : 0, start_state);
    const u32 ev = nv ? last_event16(b, nv, off, prev) : 0u;
    u32 tot;
    u32 ev_in = block_exscan_max(ev, sm, &tot);
    const u32 blk_in = blk_ev_ex[blockIdx.x];
    ev_in = ev_in > blk_in ? ev_in : blk_in;
    // the line this thread starts in: after a header start -> header; after a newline -> sequence; no event yet ->
    // whatever the chunk started in
    bool in_header = ev_in ? ((ev_in & 1u) != 0u) : (start_state == 2u);
    u32 mask = 0, zmask = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint8_t c = b[i];
        if (i >= nv) break;
        if (c == '>' && prev == '\n') {
            in_header = true;
            mask |= 1u << i; // the record breaker
            zmask |= 1u << i;
        } else if (c == '\n') {
            in_header = false;
        } else if (!in_header && c != ' ' && c != '\t' && c != '\r') {
            mask |= 1u << i;
        }
        prev = c;
    }
    const u32 nkeep = (u32)__popc(mask);
    u32 ktot;
    const u32 kpre = block_exscan(nkeep, sm, &ktot);
    if (!WRITE) {
        if (threadIdx.x == 0) blk_keep[blockIdx.x] = ktot;
        return;
    }
    u64 o = (u64)blk_keep_ex[blockIdx.x] + kpre;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if ((mask >> i) & 1u) out[o++] = ((zmask >> i) & 1u) ? (uint8_t)0 : b[i];
    }
}

hipError_t launch_fasta_pack(const uint8_t *text, u64 len, u32 start_state, uint8_t *out, u32 *blk_a, u32 *blk_b,
                             u32 *totals, hipStream_t st) {
    if (len == 0) return hipSuccess;
    const u32 nblk = (u32)((len + BLK_BYTES - 1) / BLK_BYTES);
    hipLaunchKernelGGL(k1f_events, dim3(nblk), dim3(TB), 0, st, text, len, start_state, blk_a);
    hipLaunchKernelGGL(k1_scan_max, dim3(1), dim3(1024), 0, st, blk_a, nblk);
    hipLaunchKernelGGL((k1f_pack<false>), dim3(nblk), dim3(TB), 0, st, text, len, start_state, (const u32 *)blk_a, blk_b,
                       (const u32 *)nullptr, (uint8_t *)nullptr);
    hipLaunchKernelGGL(k1_scan, dim3(1), dim3(1024), 0, st, blk_b, nblk, totals + 1);
    hipLaunchKernelGGL((k1f_pack<true>), dim3(nblk), dim3(TB), 0, st, text, len, start_state, (const u32 *)blk_a,
                       (u32 *)nullptr, (const u32 *)blk_b, out);
    return hipGetLastError();
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
