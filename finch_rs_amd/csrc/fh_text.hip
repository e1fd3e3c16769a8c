// fh_text.hip -- K1: FASTQ / FASTA text -> packed sequence stream on the device.
//
// --- FASTQ ---
// Replaces, for plain 4-line FASTQ, the host side of needletail's record splitting (lib/src/lib.rs:60-68): the host
// only reads raw file bytes into a pinned buffer and cuts them at a record boundary; which bytes are sequence
// is decided here.  A byte belongs to a sequence line iff the number of newlines before it is 1 mod 4.
// Sequence bytes are copied (CR dropped), the newline that ends a sequence line becomes the '\0' record
// breaker, everything else (headers, '+' lines, qualities) is dropped -- a stream compaction:
//   pass A  per 4 KiB block: newlines, and the bytes it will emit for each of the four line phases it might start in
//           -> exclusive scan of the newlines (one workgroup), pick each block's class, exclusive scan of those
//   pass C  recompute the flags, block-local scan, scatter the kept bytes
// Two passes over the chunk with 16-byte loads; a thread whose sixteen bytes hold no newline (four of five) decides them
// as one, a thread that keeps all sixteen stores four words.
// What needletail checks per record is checked here too, so that a file the reference refuses (or reads differently) never
// yields a sketch silently: header lines begin with '@', separator lines with '+', a sequence line holds no blank, tab or
// interior CR (normalize(false) would DROP those and let k-mers span them; the packed stream would break k-mers there),
// and every record's sequence and quality lines are equally long (pass C notes where every line ends, k1_check_records
// compares).  Any violation sets the error flag; the host layer then re-reads the file through its own parser, which
// reproduces needletail's behaviour case by case (FINCH_DEVICE_PARSE=1: the error is returned instead).
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
// inclusive prefix sum over the 64 lanes of a wave, in registers (row shifts, then gfx9's two row broadcasts)
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31
    return v;
}

__device__ __forceinline__ u32 block_exscan(u32 v, u32 *smem /* >= 4 */, u32 *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 inc = wave_incl_scan(v);
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

// Pass A of the FASTQ splitter, with pass B folded in: newlines per block, and -- for each of the four line phases the
// block might start in -- how many of its bytes the packer will emit.  A byte is emitted iff its line is a sequence line
// (phase 1) and it is not a CR, the terminating newline included (it becomes the breaker): that depends on the block's
// starting phase only through a rotation, so the count is taken per class of "newlines before the byte within the block,
// mod 4" and the class that applies is picked once the scan of the newline counts is known (k1_pick_keep).
__global__ __launch_bounds__(TB) void k1_count_classes(const uint8_t *text, u64 len, u32 *blk_nl, u32 *blk_keep4) {
    __shared__ u32 sm[4];
    __shared__ u32 cls[4];
    if (threadIdx.x < 4) cls[threadIdx.x] = 0;
    const u64 off = ((u64)blockIdx.x * TB + threadIdx.x) * BPT;
    uint8_t b[16];
    const int nv = load16(text, len, off, b);
    u32 c = 0, k4[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < nv) {
            k4[c & 3u] += b[i] != '\r';
            c += b[i] == '\n';
        }
    }
    u32 tot;
    const u32 base = block_exscan(c, sm, &tot); // (its barriers also order the zeroing of cls before the adds)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (k4[q]) atomicAdd(&cls[(q + base) & 3u], k4[q]);
    __syncthreads();
    if (threadIdx.x == 0) blk_nl[blockIdx.x] = tot;
    if (threadIdx.x < 4) blk_keep4[4u * blockIdx.x + threadIdx.x] = cls[threadIdx.x];
}

// blk_keep[i] = the emitted-byte count of block i for the phase it really starts in (nl_ex: exclusive scan of the newlines)
__global__ __launch_bounds__(256) void k1_pick_keep(const u32 *nl_ex, const u32 *blk_keep4, u32 *blk_keep, u32 nblk) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nblk) blk_keep[i] = blk_keep4[4u * i + ((1u - (nl_ex[i] & 3u)) & 3u)];
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
    u32 bad;   // structure violation seen (header not '@' / separator not '+' / whitespace inside a sequence line)
};

// `next` = the byte after the 16 (0 past the end of the chunk): a CR only ends a line if a newline follows it
__device__ __forceinline__ Keep16 decide16(const uint8_t b[16], int nv, u32 line0, bool first_is_line_start, uint8_t next) {
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
            if (ph == 1u) {
                if (c == '\r') { // a line-ending CR is dropped (and not a base); anywhere else it is not for this path
                    const uint8_t nx = (i + 1 < nv) ? b[(i + 1) & 15] : next;
                    if (nx != '\n') k.bad = 1u;
                } else {
                    if (c == ' ' || c == '\t') k.bad = 1u;
                    k.mask |= 1u << i;
                }
            }
            at_start = false;
        }
    }
    return k;
}

__global__ __launch_bounds__(TB) void k1_pack(const uint8_t *text, u64 len, const u32 *blk_nl_ex, const u32 *blk_keep_ex, uint8_t *out,
                                              u32 *err, u32 *line_end, u32 line_cap) {
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
    const uint8_t next = (nv == 16 && off + 16 < len) ? text[off + 16] : (uint8_t)0;
    Keep16 k;
    if (c == 0u && nv == 16) {
        // no line ends inside these sixteen bytes (four of five threads): one phase for all of them
        const u32 ph = nl_before & 3u;
        k = Keep16{0u, 0u, 0u, 0u};
        if (starts_line && ((ph == 0u && b[0] != '@' && b[0] != '\r') || (ph == 2u && b[0] != '+'))) k.bad = 1u;
        if (ph == 1u) {
            u32 m = 0xFFFFu;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (b[i] == ' ' || b[i] == '\t') k.bad = 1u;
                if (b[i] == '\r') { // only the CR of a CR LF pair may be here, i.e. the last byte with the newline next
                    m &= ~(1u << i);
                    if (i != 15 || next != '\n') k.bad = 1u;
                }
            }
            k.mask = m;
        }
    } else {
        k = decide16(b, nv, nl_before, starts_line, next);
    }
    if (k.bad) atomicExch(err, 1u);
    if (k.n_nl) { // where line j ends: (position << 1) | "a CR precedes the newline"
        u32 line = nl_before;
        uint8_t prev = off ? text[off - 1] : (uint8_t)0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < nv && b[i] == '\n') {
                if (line < line_cap) line_end[line] = ((u32)(off + i) << 1) | (prev == '\r' ? 1u : 0u);
                else atomicExch(err, 1u); // more lines than the index holds (records of a few bytes): host parser
                line++;
            }
            prev = b[i];
        }
    }
    const u32 nkeep = (u32)__popc(k.mask);
    u32 ktot;
    const u32 kpre = block_exscan(nkeep, sm, &ktot);
    u64 o = (u64)blk_keep_ex[blockIdx.x] + kpre;
    if (k.mask == 0xFFFFu) {
        // all sixteen bytes lie on a sequence line (four of ten threads of a 150-base FASTQ; half keep nothing at all):
        // four word stores -- the target is wherever the compaction puts it, rarely aligned -- instead of sixteen byte stores
        u32 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32 v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * q + j;
                v |= (((k.zmask >> i) & 1u) ? 0u : (u32)b[i]) << (8 * j);
            }
            w[q] = v;
        }
        __builtin_memcpy(out + o, w, 16);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if ((k.mask >> i) & 1u) out[o++] = ((k.zmask >> i) & 1u) ? (uint8_t)0 : b[i];
        }
    }
}

// bases = emitted bytes that are not breakers (what total_bases counts for FASTQ, mash.rs:72).  A chunk starts at a record,
// so the newlines that end sequence lines are those with index 1 mod 4: (L + 2) / 4 of L -- no need to count them with
// atomics (one per workgroup on a single address was a good part of the packer's time).  totals: [0] newlines, [1] emitted.
// The same thread judges the shape of the chunk's end.  A chunk is whole records, so its L newlines are 4 per record
// with nothing behind them, or 3 mod 4 when the last record's quality line has no newline (or is not there at all: an
// empty quality line, which k1_check_records then holds against the sequence).  Anything else -- a record cut off after
// its header or its sequence line, a header begun behind the last record, blank lines at the end -- is for the host
// parser to accept or to refuse with needletail's error ("truncated FASTQ record"); silently sketching what is there
// would not be the reference's behaviour.
__global__ void k1_add_bases(Ctl *ctl, const u32 *totals, const uint8_t *text, u64 len, u32 *err) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctl->text_bases += (u64)totals[1] - (u64)((totals[0] + 2u) / 4u);
        const u32 m = totals[0] & 3u;
        const bool open_line = text[len - 1] != '\n'; // bytes behind the last newline
        if (!((m == 0u && !open_line) || m == 3u)) atomicExch(err, 1u);
    }
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

// one thread per record: sequence and quality line equally long (CRs before the newlines not counted); the chunk may end
// without the last quality line's newline
__global__ __launch_bounds__(256) void k1_check_records(const u32 *line_end, const u32 *n_lines_p, u64 len, const uint8_t *text,
                                                        u32 line_cap, u32 *err) {
    const u32 n_lines = *n_lines_p < line_cap ? *n_lines_p : line_cap;
    const u32 n_rec = (n_lines + 1u) / 4u; // 4r + 3 newlines make record r complete if the text goes on to its end
    for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += gridDim.x * blockDim.x) {
        const u32 a = line_end[4u * r], b = line_end[4u * r + 1u], c = line_end[4u * r + 2u];
        u32 d;
        if (4u * r + 3u < n_lines) d = line_end[4u * r + 3u];
        else d = ((u32)len << 1) | ((len && text[len - 1] == '\r') ? 1u : 0u);
        const u32 seq_len = (b >> 1) - (a >> 1) - 1u - (b & 1u), qual_len = (d >> 1) - (c >> 1) - 1u - (d & 1u);
        if (seq_len != qual_len) atomicExch(err, 1u);
    }
}

hipError_t launch_fastq_pack(const uint8_t *text, u64 len, uint8_t *out, u32 *blk_a, u32 *blk_b, u32 *totals, Ctl *ctl,
                             u32 *err, u32 *line_end, u32 line_cap, hipStream_t st) {
    if (len == 0) return hipSuccess;
    if (len >= (1ull << 31)) return hipErrorInvalidValue;
    const u32 nblk = (u32)((len + BLK_BYTES - 1) / BLK_BYTES);
    // (blk_a holds 5 values per block: the newline count, then -- behind all of those -- the four class counts)
    u32 *keep4 = blk_a + nblk;
    hipLaunchKernelGGL(k1_count_classes, dim3(nblk), dim3(TB), 0, st, text, len, blk_a, keep4);
    hipLaunchKernelGGL(k1_scan, dim3(1), dim3(1024), 0, st, blk_a, nblk, totals);
    hipLaunchKernelGGL(k1_pick_keep, dim3((nblk + 255u) / 256u), dim3(256), 0, st, (const u32 *)blk_a, (const u32 *)keep4, blk_b, nblk);
    hipLaunchKernelGGL(k1_scan, dim3(1), dim3(1024), 0, st, blk_b, nblk, totals + 1);
    hipLaunchKernelGGL(k1_add_bases, dim3(1), dim3(64), 0, st, ctl, (const u32 *)totals, text, len, err);
    hipLaunchKernelGGL(k1_pack, dim3(nblk), dim3(TB), 0, st, text, len, (const u32 *)blk_a, (const u32 *)blk_b, out, err, line_end,
                       line_cap);
    hipLaunchKernelGGL(k1_check_records, dim3(256), dim3(256), 0, st, (const u32 *)line_end, (const u32 *)totals, len, text,
                       line_cap, err);
    return hipGetLastError();
}

} // namespace fh
