// fh_bgzf.hip -- BGZF members inflated on the device (gfx950).
//
// needletail hands finch decompressed text whatever the file holds (lib.rs:60), and sequencing reads mostly arrive
// compressed.  A BGZF file (bgzip, every BAM toolchain) is a chain of independent gzip members of at most 64 KiB of text,
// each carrying its compressed size in the header, so the members of a batch can be inflated side by side: on the host
// that is what the read threads of fh_host.cpp do (16 of them reach ~12 GB/s of text); here one wavefront takes one
// member, the compressed bytes cross PCIe instead of the text (4-5x fewer), and the text is born where the FASTQ
// splitter (fh_text.hip) reads it.
//
//   k_bgzf_inflate   one 64-lane workgroup per member.  The symbols of a block are decoded 64 bit offsets at a time: every
//                    lane decodes the symbol that would start at its own offset, the lanes where symbols really start are
//                    found by following the lengths from the first one (a scalar chase, one v_readlane per symbol), output
//                    positions are a DPP prefix sum over those lanes, and their tokens join a queue in LDS.  Block headers
//                    are parsed wave-uniform (bit buffer in scalar registers; lane i holds word i of the current 256 input
//                    bytes), the lanes build the Huffman tables (canonical codes assigned with ballots, replicated entries
//                    filled in parallel), and the queued tokens are written out 64 at a time, every lane copying its own
//                    match; a match whose source lies inside the group waits for the round in which everything before it
//                    has been written.
//   k_bgzf_crc       CRC-32 of every member's text against its trailer: 64 slices per member hashed with slicing-by-4
//                    tables in LDS, joined with the x^n mod P operators of zlib's crc32_combine.
//   k_fastq_cut      where the last whole FASTQ record of the inflated text ends (the rest waits for the next batch).
//
// Damage stays loud: any code zlib would reject, a size or checksum that differs from the member's trailer, or a read
// past the member's bytes sets the status word and the push fails (fh_api.hip), after which the host layer reads the
// file again through its own inflate.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fh_core.h"
#include "fh_kernels.h"

namespace fh {

namespace {

#ifndef BZ_FENCE_SCOPE
#define BZ_FENCE_SCOPE "workgroup"
#endif
constexpr int LIT_BITS = 10, DIST_BITS = 8, CL_BITS = 7;
constexpr u32 KIND_LIT = 0, KIND_BASE = 1, KIND_EOB = 2, KIND_LONG = 3;
// "no such code" in the literal/length table: the kind of the end-of-block symbol with a length of 0, so that the symbol
// loop's common cases (literal, match) need no validity test of their own
constexpr u32 LIT_STOP = KIND_EOB << 8;
// (reasons of failure, see BgzfFail)
constexpr u32 BZ_BAD_BLOCK_ = 1, BZ_BAD_CODE_ = 2, BZ_BAD_MATCH_ = 3, BZ_BAD_SIZE_ = 4, BZ_OVERRUN_ = 5;

// table entry: bits 0-3 code length (0 = no such code), 4-7 extra bits, 8-9 kind, 16-31 literal / base value / symbol
__device__ __forceinline__ u32 make_entry(u32 nbits, u32 extra, u32 kind, u32 value) {
    return nbits | (extra << 4) | (kind << 8) | (value << 16);
}
__device__ __forceinline__ u32 litlen_entry(u32 sym, u32 nbits) {
    if (sym < 256u) return make_entry(nbits, 0, KIND_LIT, sym);
    if (sym == 256u) return make_entry(nbits, 0, KIND_EOB, 0);
    const u32 i = sym - 257u;
    if (i >= 29u) return LIT_STOP; // 286, 287: in the fixed code's space, never valid
    if (i < 8u) return make_entry(nbits, 0, KIND_BASE, 3u + i);
    if (i == 28u) return make_entry(nbits, 0, KIND_BASE, 258u);
    const u32 extra = (i - 4u) >> 2;
    return make_entry(nbits, extra, KIND_BASE, 3u + ((4u + (i & 3u)) << extra));
}
__device__ __forceinline__ u32 dist_entry(u32 sym, u32 nbits) {
    if (sym >= 30u) return 0u;
    if (sym < 4u) return make_entry(nbits, 0, KIND_BASE, 1u + sym);
    const u32 extra = (sym - 2u) >> 1;
    return make_entry(nbits, extra, KIND_BASE, 1u + ((2u + (sym & 1u)) << extra));
}
__device__ __forceinline__ u32 symbol_entry(int which, u32 sym, u32 nbits) {
    return which == 0 ? litlen_entry(sym, nbits) : which == 1 ? dist_entry(sym, nbits) : make_entry(nbits, 0, KIND_LIT, sym);
}

struct CodeSet { // the canonical code itself, for the codes longer than the table's index
    u32 count[16], first[16], offs[16];
    uint16_t sorted[288];
};
template <class QP>
struct LdsT {
    u32 lit[1 << LIT_BITS];
    u32 dist[1 << DIST_BITS]; // (its first 128 entries hold the code-length code while a dynamic header is read)
    CodeSet cs[2];
    uint8_t lens[320];
    uint8_t cl_lens[32];
    u32 qinfo[128];   // (the lane-parallel symbol loop) tokens waiting for a full group of 64 ...
    QP qpos[128];     // ... and where they go
};
using Lds = LdsT<uint16_t>; // BGZF: a member's text is at most 65536 bytes, a token starts below that
using LdsGz = LdsT<u32>;    // plain gzip: a chunk's symbols, the 32768 window slots in front included

__device__ __forceinline__ u32 rfl(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
// v_writelane_b32 (clang has no builtin of that name; the LLVM intrinsic takes care of M0 for the lane select)
extern "C" __device__ int fh_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ u32 writelane(u32 value, u32 lane_idx, u32 vec) { return (u32)fh_llvm_writelane((int)value, (int)lane_idx, (int)vec); }

// Huffman table of the n code lengths at `lens`: root table of 2^R entries (codes longer than R bits: KIND_LONG, decoded
// from `cs`).  false: over-subscribed lengths.
// *unused: code space left over, in units of 2^-15 (0 = complete code, 16384 = one code of length 1, 32768 = no code).
__device__ bool build_table(const uint8_t *lens, u32 n, int R, u32 *table, CodeSet &cs, int which, u32 lane, int *unused = nullptr) {
    if (lane < 16u) cs.count[lane] = 0u;
    for (u32 i = lane; i < (1u << R); i += 64u) table[i] = which == 0 ? LIT_STOP : 0u;
    __syncthreads();
    for (u32 s = lane; s < n; s += 64u) {
        const u32 l = lens[s];
        if (l) atomicAdd(&cs.count[l], 1u);
    }
    __syncthreads();
    bool ok = true;
    {
        u32 code = 0, off = 0;
        int left = 1;
        for (u32 b = 1; b <= 15u; ++b) {
            const u32 c = cs.count[b];
            if (lane == 0) {
                cs.first[b] = code;
                cs.offs[b] = off;
            }
            code = (code + c) << 1;
            off += c;
            left = left * 2 - (int)c;
            if (left < 0) ok = false;
        }
        if (unused) *unused = left;
    }
    __syncthreads();
    if (!ok) return false;
    u32 runv = 0; // lane b: symbols of length b seen so far
    for (u32 base = 0; base < n; base += 64u) {
        const u32 s = base + lane;
        const u32 l = s < n ? lens[s] : 0u;
        u32 rank = 0;
        for (u32 b = 1; b <= 15u; ++b) {
            const unsigned long long m = __ballot(l == b);
            if (m == 0ull) continue;
            const u32 run = (u32)__builtin_amdgcn_readlane((int)runv, (int)b);
            if (l == b) rank = run + (u32)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == b) runv += (u32)__popcll(m);
        }
        if (l) {
            const u32 code = cs.first[l] + rank;
            cs.sorted[cs.offs[l] + rank] = (uint16_t)s;
            const u32 rev = __brev(code) >> (32u - l); // the order the bits arrive in
            if (l <= (u32)R) {
                const u32 e = symbol_entry(which, s, l);
                for (u32 i = rev; i < (1u << R); i += (1u << l)) table[i] = e;
            } else {
                table[rev & ((1u << R) - 1u)] = make_entry((u32)R, 0, KIND_LONG, 0);
            }
        }
    }
    __syncthreads();
    return true;
}

// Second pass over the literal/length table: where a literal's code leaves room in the index for another whole literal
// code, the entry takes both (bit 10 set, code lengths summed in bits 0-3, the first one's kept in bits 4-7, the second
// byte in bits 24-31) -- sequence and quality lines that found no match are runs of literals with short codes.
constexpr u32 PAIR_FLAG = 1u << 10;
__device__ void pair_literals(u32 *lit, u32 lane) {
    u32 upd[(1 << LIT_BITS) / 64];
#pragma unroll
    for (int r = 0; r < (1 << LIT_BITS) / 64; ++r) {
        const u32 i = (u32)r * 64u + lane;
        const u32 e1 = lit[i];
        const u32 nb1 = e1 & 15u;
        u32 out = e1;
        if (((e1 >> 8) & 3u) == KIND_LIT && nb1 != 0u && nb1 < (u32)LIT_BITS) {
            const u32 e2 = lit[i >> nb1]; // the bits behind the first code, zeros beyond the index: valid if its code ends inside it
            const u32 nb2 = e2 & 15u;
            if (((e2 >> 8) & 3u) == KIND_LIT && nb2 != 0u && nb1 + nb2 <= (u32)LIT_BITS)
                out = (nb1 + nb2) | (nb1 << 4) | (KIND_LIT << 8) | PAIR_FLAG | (((e1 >> 16) & 0xFFu) << 16) | (((e2 >> 16) & 0xFFu) << 24);
        }
        upd[r] = out;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < (1 << LIT_BITS) / 64; ++r) lit[(u32)r * 64u + lane] = upd[r];
    __syncthreads();
}

// a code longer than the table's index, bit by bit from the canonical description (v: the next bits, first bit lowest)
__device__ u32 decode_long(const CodeSet &cs, int R, int which, u32 v) {
    u32 code = __brev(v & ((1u << R) - 1u)) >> (32 - R);
    for (u32 l = (u32)R + 1u; l <= 15u; ++l) {
        code = (code << 1) | ((v >> (l - 1u)) & 1u);
        const u32 idx = code - cs.first[l];
        if (idx < cs.count[l]) return symbol_entry(which, cs.sorted[cs.offs[l] + idx], l);
    }
    return which == 0 ? LIT_STOP : 0u;
}

// The compressed bytes of one member as a bit stream.  All of it wave-uniform except win / nxt (lane i: word i of the
// current / the next 256 bytes).
struct Reader {
    const u32 *words;
    u32 n_words; // words that may be read (the member's bytes, rounded out): beyond them the stream reads as zeros
    u32 win, nxt, widx;
    u64 bb, base_bits; // base_bits: bits of the member consumed before `words` was (re)positioned
    u32 bc, skip0;
};
__device__ __forceinline__ u32 rd_load(const Reader &r, u32 w) { return w < r.n_words ? r.words[w] : 0u; }
__device__ void rd_init(Reader &r, const uint8_t *comp, u64 byte_off, u64 byte_end, u64 base_bits, u32 lane) {
    const u64 a = byte_off & ~3ull;
    r.base_bits = base_bits;
    r.words = (const u32 *)(comp + a);
    r.n_words = (u32)((byte_end - a + 3ull) >> 2);
    r.win = rd_load(r, lane);
    r.nxt = rd_load(r, 64u + lane);
    r.skip0 = (u32)(byte_off & 3ull) * 8u;
    const u32 w0 = (u32)__builtin_amdgcn_readlane((int)r.win, 0);
    r.widx = 1;
    r.bb = (u64)(w0 >> r.skip0);
    r.bc = 32u - r.skip0;
}
__device__ __forceinline__ void rd_fill(Reader &r, u32 lane) { // at least 32 bits in bb
    if (r.bc < 32u) {
        const u32 w = (u32)__builtin_amdgcn_readlane((int)r.win, (int)(r.widx & 63u));
        r.widx++;
        if ((r.widx & 63u) == 0u) {
            r.win = r.nxt;
            r.nxt = rd_load(r, r.widx + 64u + lane);
        }
        r.bb |= (u64)w << r.bc;
        r.bc += 32u;
    }
}
__device__ __forceinline__ u32 rd_take(Reader &r, u32 n) {
    const u32 v = (u32)r.bb & ((1u << n) - 1u);
    r.bb >>= n;
    r.bc -= n;
    return v;
}
__device__ __forceinline__ u64 rd_used_bits(const Reader &r) { return r.base_bits + (u64)r.widx * 32u - r.bc - r.skip0; }

// The header of a dynamic-Huffman block behind its three type bits (>= 14 bits buffered): HLIT, HDIST, the code-length
// code, and with it the hlit + hdist code lengths into L.lens.  Wave-uniform; 0 or the reason of failure.
template <class LDS>
__device__ __forceinline__ u32 read_dynamic_header(LDS &L, Reader &r, u32 lane, u32 &hlit, u32 &hdist) {
    u32 fail = 0;
    hlit = rd_take(r, 5) + 257u;
    hdist = rd_take(r, 5) + 1u;
    const u32 hclen = rd_take(r, 4) + 4u;
    if (lane < 32u) L.cl_lens[lane] = 0;
    __syncthreads();
    // the order the code-length code's own lengths come in: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
    const u64 ord_lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) |
                       (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const u64 ord_hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    for (u32 i = 0; i < hclen; ++i) {
        rd_fill(r, lane);
        const u32 v = rd_take(r, 3);
        const u32 sym = (u32)((i < 12u ? ord_lo >> (5u * i) : ord_hi >> (5u * (i - 12u))) & 31ull);
        if (lane == 0) L.cl_lens[sym] = (uint8_t)v;
    }
    __syncthreads();
    if (!build_table(L.cl_lens, 19, CL_BITS, L.dist, L.cs[1], 2, lane)) return BZ_BAD_BLOCK_;
    const u32 n = hlit + hdist;
    u32 i = 0, prev = 0;
    while (i < n && !fail) {
        rd_fill(r, lane);
        const u32 e = rfl(L.dist[(u32)r.bb & ((1u << CL_BITS) - 1u)]);
        const u32 nb = e & 15u;
        if (nb == 0u) return BZ_BAD_CODE_;
        rd_take(r, nb);
        const u32 sym = e >> 16;
        if (sym < 16u) {
            if (lane == 0) L.lens[i] = (uint8_t)sym;
            prev = sym;
            i++;
        } else {
            u32 rep, val = 0;
            if (sym == 16u) {
                if (i == 0u) return BZ_BAD_BLOCK_;
                rep = 3u + rd_take(r, 2);
                val = prev;
            } else if (sym == 17u) {
                rep = 3u + rd_take(r, 3);
            } else {
                rep = 11u + rd_take(r, 7);
            }
            if (i + rep > n) return BZ_BAD_BLOCK_;
            for (u32 j = lane; j < rep; j += 64u) L.lens[i + j] = (uint8_t)val;
            prev = val;
            i += rep;
        }
    }
    __syncthreads();
    if (L.lens[256] == 0) return BZ_BAD_BLOCK_; // no end-of-block code
    return fail;
}

// One match: `len` bytes from `dist` bytes back.  Nothing else writes either range while this runs.
__device__ __forceinline__ void copy_match(uint8_t *d, const uint8_t *s, u32 len, u32 dist) {
    if (dist >= len) { // apart: up to 64 bytes are loaded before the first of them is stored (one wait, not one per word)
        for (u32 j = 0; j < len; j += 64u) {
            const u32 n = len - j < 64u ? len - j : 64u;
            u64 r[8];
#pragma unroll
            for (u32 q = 0; q < 8u; ++q)
                if (8u * q < n) __builtin_memcpy(&r[q], s + j + 8u * q, 8); // (may read up to 7 bytes past the match: never stored)
#pragma unroll
            for (u32 q = 0; q < 8u; ++q) {
                if (8u * q + 8u <= n) {
                    __builtin_memcpy(d + j + 8u * q, &r[q], 8);
                } else if (8u * q < n) {
                    u64 w = r[q];
                    for (u32 t = 8u * q; t < n; ++t, w >>= 8) d[j + t] = (uint8_t)w;
                }
            }
        }
    } else if (dist >= 8u) { // overlapping, period >= 8: a word never reads bytes of its own store
        u32 j = 0;
        for (; j + 8u <= len; j += 8u) {
            u64 a;
            __builtin_memcpy(&a, s + j, 8);
            __builtin_memcpy(d + j, &a, 8);
        }
        for (; j < len; ++j) d[j] = s[j];
    } else { // a run of period < 8: the pattern is read once and replayed from registers
        u64 pat = 0;
        for (u32 q = 0; q < dist; ++q) pat |= (u64)s[q] << (8u * q);
        u32 ph = 0;
        for (u32 j = 0; j < len; ++j) {
            d[j] = (uint8_t)(pat >> (8u * ph));
            ph = ph + 1u == dist ? 0u : ph + 1u;
        }
    }
}

// The queued symbols of one group, one per lane, written out.  pos / info: the lane's token (info: low 9 bits match
// length, 0 = literal; high half the distance, or the literal byte -- two of them if bit 9 is set).  A lane copies when everything its match reads
// has been written: sources below `W`, the output position of the first token not yet written.
// T: bytes (BGZF), or 16-bit symbols (plain gzip: a byte, or a marker for a byte of the unknown window, copied like any other).
template <class T>
__device__ void resolve_group(T *out, u32 tpos, u32 tinfo, u32 ntok, u32 lane) {
    const u32 len = tinfo & 0x1FFu, hi = tinfo >> 16;
    bool done = lane >= ntok;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, BZ_FENCE_SCOPE); // earlier groups' bytes
    for (;;) {
        const unsigned long long pending = __ballot(!done);
        if (pending == 0ull) break;
        const u32 W = (u32)__builtin_amdgcn_readlane((int)tpos, (int)__builtin_ctzll(pending));
        if (!done) {
            if (len == 0u) {
                out[tpos] = (T)(hi & 0xFFu);
                if (tinfo & 0x200u) out[tpos + 1u] = (T)((hi >> 8) & 0xFFu);
                done = true;
            } else {
                const u32 src = tpos - hi;
                const u32 src_end = src + (len < hi ? len : hi); // (an overlapping match re-reads its own bytes)
                if (src_end <= W) {
                    copy_match((uint8_t *)(out + tpos), (const uint8_t *)(out + src), len * (u32)sizeof(T), hi * (u32)sizeof(T));
                    done = true;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, BZ_FENCE_SCOPE);
    }
}

} // namespace

// inclusive prefix sum over the 64 lanes, in registers (row shifts, then the two row broadcasts gfx9 has for this)
__device__ __forceinline__ u32 wave_scan_add(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}

// write out the queued tokens, 64 at a time (`all`: the rest too)
template <class LDS, class T>
__device__ void flush_queue(LDS &L, T *out, u32 &qn, u32 lane, bool all) {
    while (qn >= 64u || (all && qn > 0u)) {
        const u32 n = qn < 64u ? qn : 64u;
        __syncthreads();
        const u32 tpos = L.qpos[lane], tinfo = L.qinfo[lane];
        resolve_group(out, tpos, tinfo, n, lane);
        const u32 rest = qn - n;
        const u32 p1 = L.qpos[64u + lane], i1 = L.qinfo[64u + lane];
        __syncthreads();
        if (lane < rest) {
            L.qpos[lane] = p1; // (narrowed to the queue's position type)
            L.qinfo[lane] = i1;
        }
        qn = rest;
    }
    __syncthreads();
}

// The symbols of one block, 64 bit offsets at a time: every lane decodes the symbol that would start at its offset --
// literal/length lookup, extra bits, distance lookup, all of it -- and the lanes where symbols really start are found by
// following the lengths from the first one (a scalar chase of one v_readlane per symbol).  Output positions come from a
// prefix sum over those lanes; their tokens join the queue.  mbits: bits of the member consumed (in: where the block's
// symbols begin, out: behind its end-of-block code).  Returns 0 or the reason of failure.
template <class LDS, class T>
__device__ u32 block_symbols_parallel(LDS &L, const uint8_t *mbase, u64 in_bits, u32 isize, T *out, u64 &mbits_ref, u32 &pos_ref,
                                      u32 &qn_ref, u32 lane) {
    u64 mbits = mbits_ref;
    u32 pos = pos_ref, qn = qn_ref, fail = 0;
    for (;;) {
        if (mbits > in_bits + 64u) {
            fail = BZ_OVERRUN_;
            break;
        }
        const u64 my = mbits + lane;
        u64 b;
        __builtin_memcpy(&b, mbase + (my >> 3), 8); // (reads at most 24 bytes past the member: inside the buffer's padding)
        b >>= (u32)(my & 7u);
        u32 e = L.lit[(u32)b & ((1u << LIT_BITS) - 1u)];
        u32 kind = (e >> 8) & 3u;
        if (kind == KIND_LONG) {
            e = decode_long(L.cs[0], LIT_BITS, 0, (u32)b);
            kind = (e >> 8) & 3u;
        }
        const u32 nb = e & 15u;
        u32 t = nb, info = 0, adv = 0, st = 0; // st: 0 a token, 1 end of block, 2 no such code
        if (kind == KIND_LIT) {
            info = (e & 0xFFFF0000u) | ((e & PAIR_FLAG) >> 1);
            adv = 1u + ((e >> 10) & 1u);
        } else if (kind == KIND_BASE) {
            const u32 ex = (e >> 4) & 15u;
            const u32 len = (e >> 16) + ((u32)(b >> nb) & ((1u << ex) - 1u));
            t += ex;
            u32 d = L.dist[(u32)(b >> t) & ((1u << DIST_BITS) - 1u)];
            if (((d >> 8) & 3u) == KIND_LONG) d = decode_long(L.cs[1], DIST_BITS, 1, (u32)(b >> t));
            const u32 db = d & 15u, dex = (d >> 4) & 15u;
            if (db == 0u) st = 2;
            const u32 dist = (d >> 16) + ((u32)(b >> (t + db)) & ((1u << dex) - 1u));
            t += db + dex;
            info = (dist << 16) | len;
            adv = len;
        } else {
            st = nb ? 1u : 2u;
        }
        const u32 tl = st ? 128u : t;
        // the chain of symbol starts
        unsigned long long marks = 0ull;
        u32 s = 0, stop = 64u;
        while (s < 64u) {
            marks |= 1ull << s;
            const u32 v = (u32)__builtin_amdgcn_readlane((int)tl, (int)s);
            if (v & 128u) {
                stop = s;
                break;
            }
            s += v;
        }
        if (stop < 64u) marks &= ~(1ull << stop);
        const bool tok = (marks >> lane) & 1ull;
        const u32 a = tok ? adv : 0u;
        const u32 incl = wave_scan_add(a);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u32 mypos = pos + incl - a;
        if (__ballot(tok && (info & 0x1FFu) != 0u && (info >> 16) > mypos)) {
            fail = BZ_BAD_MATCH_;
            break;
        }
        if (pos + total > isize) {
            fail = BZ_BAD_SIZE_;
            break;
        }
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(marks >> 32), __builtin_amdgcn_mbcnt_lo((u32)marks, 0u));
        if (tok) {
            L.qpos[qn + rank] = mypos; // (narrowed to the queue's position type)
            L.qinfo[qn + rank] = info;
        }
        qn += (u32)__popcll(marks);
        pos += total;
        if (qn >= 64u) flush_queue(L, out, qn, lane, false);
        if (stop < 64u) {
            if ((u32)__builtin_amdgcn_readlane((int)st, (int)stop) != 1u) {
                fail = BZ_BAD_CODE_;
                break;
            }
            mbits += stop + (u32)__builtin_amdgcn_readlane((int)nb, (int)stop);
            break;
        }
        mbits += s;
    }
    mbits_ref = mbits;
    pos_ref = pos;
    qn_ref = qn;
    return fail;
}

// status[0]: 0, or (member index << 8 | reason) of a failed member (the largest such word wins)
enum BgzfFail : u32 {
    BZ_BAD_BLOCK = 1,  // block type 3, stored length check, code lengths that over-subscribe or repeat from nothing
    BZ_BAD_CODE = 2,   // a bit pattern no code of the block's tables has, or an undefined symbol
    BZ_BAD_MATCH = 3,  // distance reaches before the member's first byte
    BZ_BAD_SIZE = 4,   // the text is not `isize` bytes long
    BZ_OVERRUN = 5,    // the stream goes on past the member's last byte, or stops short of it
    BZ_BAD_CRC = 6,
};

// PAR: the symbols of a block are decoded 64 bit offsets at a time (block_symbols_parallel) instead of one by one
template <bool PAR>
__global__ __launch_bounds__(64, 5) void k_bgzf_inflate(const uint8_t *comp, const BgzfMember *members, u32 n_members,
                                                     uint8_t *text, u32 *status) {
    __shared__ Lds L;
    const u32 mi = blockIdx.x, lane = threadIdx.x;
    if (mi >= n_members) return;
    const BgzfMember m = members[mi];
    uint8_t *out = text + m.out_off;
    const u32 isize = m.isize;
    Reader r;
    rd_init(r, comp, m.in_off, (u64)m.in_off + m.in_len, 0, lane);
    u32 fail = 0, pos = 0, ntok = 0, tpos = 0, tinfo = 0, qn = 0;
    bool final_block = false;
    const u64 in_bits = (u64)m.in_len * 8u;
    while (!final_block && !fail) {
        if (rd_used_bits(r) > in_bits) {
            fail = BZ_OVERRUN;
            break;
        }
        rd_fill(r, lane);
        final_block = rd_take(r, 1) != 0u;
        const u32 type = rd_take(r, 2);
        if (type == 3u) {
            fail = BZ_BAD_BLOCK;
            break;
        }
        if (type == 0u) { // stored: LEN, ~LEN, then the bytes themselves
            rd_take(r, r.bc & 7u);
            rd_fill(r, lane);
            const u32 len = rd_take(r, 16), nlen = rd_take(r, 16);
            const u64 used = rd_used_bits(r) >> 3;
            if ((len ^ 0xFFFFu) != nlen || used + len > m.in_len || pos + len > isize) {
                fail = BZ_BAD_BLOCK;
                break;
            }
            if (PAR) flush_queue(L, out, qn, lane, true);
            else resolve_group(out, tpos, tinfo, ntok, lane);
            ntok = 0;
            const uint8_t *src = comp + m.in_off + used;
            for (u32 j = lane; j < len; j += 64u) out[pos + j] = src[j];
            pos += len;
            rd_init(r, comp, (u64)m.in_off + used + len, (u64)m.in_off + m.in_len, (used + len) * 8u, lane);
            continue;
        }
        u32 hlit = 288, hdist = 32;
        if (type == 1u) { // the fixed code
            for (u32 s = lane; s < 288u; s += 64u) L.lens[s] = s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : 8;
            if (lane < 32u) L.lens[288u + lane] = 5;
        } else {
            fail = read_dynamic_header(L, r, lane, hlit, hdist);
            if (fail) break;
        }
        __syncthreads();
        if (!build_table(L.lens, hlit, LIT_BITS, L.lit, L.cs[0], 0, lane) ||
            !build_table(L.lens + hlit, hdist, DIST_BITS, L.dist, L.cs[1], 1, lane)) {
            fail = BZ_BAD_BLOCK;
            break;
        }
        pair_literals(L.lit, lane);
        if (PAR) {
            u64 mb = rd_used_bits(r);
            fail = block_symbols_parallel(L, comp + m.in_off, in_bits, isize, out, mb, pos, qn, lane);
            if (fail) break;
            // the bit reader again, behind the end-of-block code
            rd_init(r, comp, (u64)m.in_off + (mb >> 3), (u64)m.in_off + m.in_len, (mb >> 3) * 8u, lane);
            rd_fill(r, lane);
            rd_take(r, (u32)(mb & 7u));
            continue;
        }
        for (;;) { // the block's symbols
            rd_fill(r, lane);
            u32 e = rfl(L.lit[(u32)r.bb & ((1u << LIT_BITS) - 1u)]);
            u32 kind = (e >> 8) & 3u;
            if (kind == KIND_LONG) {
                e = rfl(decode_long(L.cs[0], LIT_BITS, 0, (u32)r.bb));
                kind = (e >> 8) & 3u;
            }
            u32 info, adv;
            if (kind == KIND_LIT) { // one byte, or two (PAIR_FLAG, which becomes bit 9 of the token)
                rd_take(r, e & 15u);
                info = (e & 0xFFFF0000u) | ((e & PAIR_FLAG) >> 1);
                adv = 1u + ((e >> 10) & 1u);
            } else if (kind == KIND_BASE) {
                rd_take(r, e & 15u);
                const u32 len = (e >> 16) + rd_take(r, (e >> 4) & 15u);
                rd_fill(r, lane);
                u32 d = rfl(L.dist[(u32)r.bb & ((1u << DIST_BITS) - 1u)]);
                if (((d >> 8) & 3u) == KIND_LONG) d = rfl(decode_long(L.cs[1], DIST_BITS, 1, (u32)r.bb));
                const u32 db = d & 15u;
                if (db == 0u) {
                    fail = BZ_BAD_CODE;
                    break;
                }
                rd_take(r, db);
                const u32 dist = (d >> 16) + rd_take(r, (d >> 4) & 15u);
                if (dist > pos) {
                    fail = BZ_BAD_MATCH;
                    break;
                }
                info = (dist << 16) | len;
                adv = len;
            } else { // end of block, or a bit pattern without a code
                if ((e & 15u) == 0u) fail = BZ_BAD_CODE;
                rd_take(r, e & 15u);
                break;
            }
            tpos = writelane(pos, ntok, tpos); // the queue: token i sits in lane i
            tinfo = writelane(info, ntok, tinfo);
            pos += adv;
            if (++ntok == 64u) {
                // (nothing has been written for these tokens yet: sizes are checked once per group)
                if (pos > isize || rd_used_bits(r) > in_bits + 64u) {
                    fail = pos > isize ? BZ_BAD_SIZE : BZ_OVERRUN;
                    break;
                }
                resolve_group(out, tpos, tinfo, ntok, lane);
                ntok = 0;
            }
        }
    }
    if (!fail && pos != isize) fail = BZ_BAD_SIZE;
    if (!fail) {
        if (PAR) flush_queue(L, out, qn, lane, true);
        else resolve_group(out, tpos, tinfo, ntok, lane);
    }
    if (!fail && ((rd_used_bits(r) + 7u) >> 3) != m.in_len) fail = BZ_OVERRUN;
    if (fail && lane == 0) atomicMax(status, (mi << 8) | fail);
}

// ---------------------------------------------------------------------------------------------------------------------
// plain gzip: one DEFLATE stream, no index -- the two-pass scheme of pugz / rapidgzip (fh_pargz.h has it for host threads)
// with a wavefront per chunk.
//   k_gz_find     chunk c > 0: the first bit offset of its range that reads as the header of a non-final dynamic block --
//                 64 offsets at a time through the cheap tests (type bits, HLIT / HDIST, a complete code-length code), the
//                 survivors one by one through the real header parser, complete literal and distance codes, and first
//                 symbols that are text.
//   k_gz_inflate  every chunk with a start decodes from it into 16-bit symbols behind 32768 marker slots (what lies in front
//                 of a chunk is unknown: a match that reaches there copies markers), block by block, until a block ends
//                 exactly where a later chunk was found to begin (GZ_NEXT), the stream ends, or the bytes do.
//   k_gz_chain    one workgroup walks the chunks that really follow each other: text offsets, the 32 KiB in front of each
//                 (the previous one's tail, its markers looked up in the window before that), which text tile is whose.
//   k_gz_text     symbols narrowed to bytes, markers looked up; then CRC-32 of the batch's text (k_gz_crc_*).
// A "start" that is none costs its wavefront's work and nothing else: no chain passes through it.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ bool gz_text_byte(u32 c) { return c == '\n' || c == '\r' || c == '\t' || (c >= 32u && c < 127u); }

// Does a non-final dynamic block with complete codes, whose first symbols are text, begin at bit `pos`?  Wave-uniform.
__device__ bool gz_block_check(LdsGz &L, const uint8_t *comp, u64 n_bytes, u64 pos, u32 lane) {
    Reader r;
    rd_init(r, comp, pos >> 3, n_bytes, (pos >> 3) * 8u, lane);
    rd_fill(r, lane);
    rd_take(r, (u32)(pos & 7u));
    rd_fill(r, lane);
    rd_take(r, 3); // BFINAL = 0, BTYPE = 2: the caller has looked
    u32 hlit, hdist;
    if (read_dynamic_header(L, r, lane, hlit, hdist)) return false;
    __syncthreads();
    int free_lit = 0, free_dist = 0;
    if (!build_table(L.lens, hlit, LIT_BITS, L.lit, L.cs[0], 0, lane, &free_lit) || free_lit != 0) return false;
    if (!build_table(L.lens + hlit, hdist, DIST_BITS, L.dist, L.cs[1], 1, lane, &free_dist)) return false;
    if (free_dist != 0 && free_dist != 16384 && free_dist != 32768) return false; // complete, one code, or none
    const u64 in_bits = n_bytes * 8u;
    for (u32 sym = 0; sym < 512u; ++sym) {
        if (rd_used_bits(r) > in_bits) return false;
        rd_fill(r, lane);
        u32 e = rfl(L.lit[(u32)r.bb & ((1u << LIT_BITS) - 1u)]);
        u32 kind = (e >> 8) & 3u;
        if (kind == KIND_LONG) {
            e = rfl(decode_long(L.cs[0], LIT_BITS, 0, (u32)r.bb));
            kind = (e >> 8) & 3u;
        }
        if (kind == KIND_LIT) {
            if (!gz_text_byte(e >> 16)) return false;
            rd_take(r, e & 15u);
            continue;
        }
        if (kind != KIND_BASE) return (e & 15u) != 0u && sym > 0u; // end of block (an empty one tells nothing), or no such code
        rd_take(r, e & 15u);
        rd_take(r, (e >> 4) & 15u);
        rd_fill(r, lane);
        u32 d = rfl(L.dist[(u32)r.bb & ((1u << DIST_BITS) - 1u)]);
        if (((d >> 8) & 3u) == KIND_LONG) d = rfl(decode_long(L.cs[1], DIST_BITS, 1, (u32)r.bb));
        if ((d & 15u) == 0u) return false;
        rd_take(r, d & 15u);
        rd_take(r, (d >> 4) & 15u);
    }
    return true;
}
} // namespace

__global__ __launch_bounds__(64) void k_gz_find(const uint8_t *comp, u64 n_bytes, u64 chunk_bits, u32 n_chunks, u64 first_bit,
                                                GzChunk *recs) {
    __shared__ LdsGz L;
    const u32 ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= n_chunks) return;
    u64 found = GZ_NONE;
    if (ci == 0u) {
        found = first_bit;
    } else {
        u64 from = (u64)ci * chunk_bits;
        if (from <= first_bit) from = first_bit + 1u;
        // (a header and a few symbols have to fit behind a start: nothing is looked for in the last 64 bytes)
        const u64 all = n_bytes > 64u ? (n_bytes - 64u) * 8u : 0u;
        u64 to = (u64)(ci + 1u) * chunk_bits;
        if (to > all) to = all;
        for (u64 p0 = from; p0 < to && found == GZ_NONE; p0 += 64u) {
            const u64 pos = p0 + lane;
            bool cand = false;
            if (pos < to) {
                u64 a;
                __builtin_memcpy(&a, comp + (pos >> 3), 8);
                const u64 v = a >> (u32)(pos & 7u);
                const u32 hl = (u32)(v >> 3) & 31u, hd = (u32)(v >> 8) & 31u, hclen = ((u32)(v >> 13) & 15u) + 4u;
                if ((v & 7u) == 4u && hl <= 29u && hd <= 29u) {
                    const u64 p2 = pos + 17u;
                    u64 b;
                    __builtin_memcpy(&b, comp + (p2 >> 3), 8);
                    const u64 w = b >> (u32)(p2 & 7u); // 57 bits: 19 lengths of 3
                    u32 kraft = 0;
#pragma unroll
                    for (u32 i = 0; i < 19u; ++i) {
                        const u32 l = (u32)(w >> (3u * i)) & 7u;
                        if (i < hclen && l) kraft += 128u >> l;
                    }
                    cand = kraft == 128u; // every encoder's code-length code is complete
                }
            }
            unsigned long long mask = __ballot(cand);
            while (mask) {
                const u32 b = (u32)__builtin_ctzll(mask);
                mask &= mask - 1ull;
                if (gz_block_check(L, comp, n_bytes, p0 + b, lane)) {
                    found = p0 + b;
                    break;
                }
            }
        }
    }
    if (lane == 0) recs[ci] = GzChunk{found, found, 0u, (u32)GZ_IDLE};
}

__global__ __launch_bounds__(64, 5) void k_gz_inflate(const uint8_t *comp, u64 n_bytes, u64 chunk_bits, u32 n_chunks, GzChunk *recs,
                                                      uint16_t *sym, u64 cap) {
    __shared__ LdsGz L;
    const u32 ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= n_chunks) return;
    const u64 start = recs[ci].start_bit;
    if (start == GZ_NONE) return;
    uint16_t *out = sym + (u64)ci * cap;
    for (u32 j = lane; j < GZ_WINDOW / 2u; j += 64u) ((u32 *)out)[j] = (0x8000u | (2u * j)) | ((0x8001u | (2u * j)) << 16);
    __syncthreads();
    const u64 in_bits = n_bytes * 8u;
    // room: this chunk's share of the symbol buffer and those of the chunks behind it in which no start was found (a chunk
    // that decodes through their ranges produces their text as well)
    u32 limit;
    {
        u32 j = ci + 1u;
        while (j < n_chunks && recs[j].start_bit == GZ_NONE) j++;
        const u64 room = (u64)(j - ci) * cap;
        limit = room > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (u32)room;
    }
    Reader r;
    rd_init(r, comp, start >> 3, n_bytes, (start >> 3) * 8u, lane);
    rd_fill(r, lane);
    rd_take(r, (u32)(start & 7u));
    u32 fail = 0, state = GZ_FAILED, pos = GZ_WINDOW, qn = 0;
    u64 end_bit = start, fail_at = 0;
    u32 end_pos = pos;
    for (;;) {
        // a block boundary: what has been decoded up to here stands whatever becomes of the next block
        end_bit = rd_used_bits(r);
        end_pos = pos;
        if (end_bit + 3u > in_bits) {
            state = GZ_OUT_OF_INPUT;
            break;
        }
        rd_fill(r, lane);
        const bool final_block = rd_take(r, 1) != 0u;
        const u32 type = rd_take(r, 2);
        if (type == 3u) {
            fail = BZ_BAD_BLOCK;
            fail_at = end_bit;
            break;
        }
        if (type == 0u) { // stored: LEN, ~LEN, then the bytes themselves
            rd_take(r, r.bc & 7u);
            rd_fill(r, lane);
            const u32 len = rd_take(r, 16), nlen = rd_take(r, 16);
            const u64 used = rd_used_bits(r) >> 3;
            if (used + len > n_bytes) {
                state = GZ_OUT_OF_INPUT;
                break;
            }
            if ((len ^ 0xFFFFu) != nlen) {
                fail = BZ_BAD_BLOCK;
                fail_at = used * 8u;
                break;
            }
            if ((u64)pos + len > limit) {
                fail = BZ_BAD_SIZE;
                break;
            }
            flush_queue(L, out, qn, lane, true);
            const uint8_t *src = comp + used;
            for (u32 j = lane; j < len; j += 64u) out[pos + j] = src[j];
            pos += len;
            rd_init(r, comp, used + len, n_bytes, (used + len) * 8u, lane);
        } else {
            u32 hlit = 288, hdist = 32;
            if (type == 1u) { // the fixed code
                for (u32 s = lane; s < 288u; s += 64u) L.lens[s] = s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : 8;
                if (lane < 32u) L.lens[288u + lane] = 5;
            } else {
                fail = read_dynamic_header(L, r, lane, hlit, hdist);
                if (fail) {
                    fail_at = rd_used_bits(r);
                    break;
                }
            }
            __syncthreads();
            if (!build_table(L.lens, hlit, LIT_BITS, L.lit, L.cs[0], 0, lane) ||
                !build_table(L.lens + hlit, hdist, DIST_BITS, L.dist, L.cs[1], 1, lane)) {
                fail = BZ_BAD_BLOCK;
                fail_at = rd_used_bits(r);
                break;
            }
            pair_literals(L.lit, lane);
            u64 mb = rd_used_bits(r);
            fail = block_symbols_parallel(L, comp, in_bits, limit, out, mb, pos, qn, lane);
            if (fail) {
                fail_at = mb;
                break;
            }
            rd_init(r, comp, mb >> 3, n_bytes, (mb >> 3) * 8u, lane);
            rd_fill(r, lane);
            rd_take(r, (u32)(mb & 7u));
        }
        const u64 b = rd_used_bits(r);
        if (b > in_bits) { // the block "ended" in the padding behind the bytes
            state = GZ_OUT_OF_INPUT;
            break;
        }
        end_bit = b;
        end_pos = pos;
        if (final_block) {
            state = GZ_MEMBER_END;
            break;
        }
        const u64 j = b / chunk_bits;
        if (j > ci && j < n_chunks && recs[j].start_bit == b) {
            state = GZ_NEXT;
            break;
        }
    }
    if (fail) {
        // (what was decoded from the padding behind the last byte proves nothing: the block is cut short by the batch's end)
        if (fail != BZ_BAD_SIZE && fail_at + 512u > in_bits) state = GZ_OUT_OF_INPUT;
        else state = (u32)GZ_FAILED | (fail << 8);
    }
    flush_queue(L, out, qn, lane, true);
    if (lane == 0) {
        recs[ci].end_bit = end_bit;
        recs[ci].out_len = end_pos - GZ_WINDOW;
        recs[ci].state = state;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// CRC-32 (IEEE 802.3, reflected) of each member's text
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr u32 CRC_POLY = 0xEDB88320u;
// a(x) * b(x) mod P, reflected representation (bit 31 = x^0): the multiplication of zlib's crc32_combine
__host__ __device__ inline u32 crc_multmodp(u32 a, u32 b) {
    u32 m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0u) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
struct X2N {
    u32 t[32]; // x^(2^k) mod P
};
__device__ inline u32 crc_x2nmodp(const X2N &x, u32 n, u32 k) { // x^(n * 2^k) mod P
    u32 p = 1u << 31;
    while (n) {
        if (n & 1u) p = crc_multmodp(x.t[k & 31u], p);
        n >>= 1;
        k++;
    }
    return p;
}
} // namespace

__global__ __launch_bounds__(256) void k_bgzf_crc(const BgzfMember *members, u32 n_members, const uint8_t *text, X2N x2n,
                                                  u32 *status) {
    __shared__ u32 T[4][256];
    {
        u32 c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        T[0][threadIdx.x] = c;
    }
    __syncthreads();
    for (int t = 1; t < 4; ++t) {
        const u32 prev = T[t - 1][threadIdx.x];
        T[t][threadIdx.x] = (prev >> 8) ^ T[0][prev & 0xFFu];
        __syncthreads();
    }
    const u32 lane = threadIdx.x & 63u;
    const u32 mi = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (mi >= n_members) return;
    const BgzfMember m = members[mi];
    const u32 slice = (((m.isize + 63u) >> 6) + 3u) & ~3u;
    const u32 lo = lane * slice < m.isize ? lane * slice : m.isize;
    const u32 hi = lo + slice < m.isize ? lo + slice : m.isize;
    const uint8_t *p = text + m.out_off;
    u32 c = 0xFFFFFFFFu, i = lo;
    for (; i < hi && ((uintptr_t)(p + i) & 3u); ++i) c = (c >> 8) ^ T[0][(c ^ p[i]) & 0xFFu];
    for (; i + 4u <= hi; i += 4u) {
        c ^= *(const u32 *)(p + i);
        c = T[3][c & 0xFFu] ^ T[2][(c >> 8) & 0xFFu] ^ T[1][(c >> 16) & 0xFFu] ^ T[0][c >> 24];
    }
    for (; i < hi; ++i) c = (c >> 8) ^ T[0][(c ^ p[i]) & 0xFFu];
    c ^= 0xFFFFFFFFu;
    // crc(A || B) = crc(A) * x^(8 |B|) + crc(B): every slice shifted past what follows it
    u32 part = hi > lo ? crc_multmodp(crc_x2nmodp(x2n, m.isize - hi, 3), c) : 0u;
    for (int off = 32; off > 0; off >>= 1) part ^= __shfl_xor(part, off);
    if (lane == 0 && part != m.crc) atomicMax(status, (mi << 8) | (u32)BZ_BAD_CRC);
}

// out[0] = offset of the last header line whose line two below is a '+' line (what fh_host.cpp's reader cuts a FASTQ
// chunk at), or `total` for the last text of a file; out[1] = 1 when no such line is in the last 16.
__device__ void fastq_cut_body(const uint8_t *text, u32 total, u32 last, u32 *out) {
    const u32 lane = threadIdx.x;
    if (last || total == 0u) {
        if (lane == 0) {
            out[0] = total;
            out[1] = 0;
        }
        return;
    }
    __shared__ u32 ls[16];
    u32 n = 0, scan_hi = total;
    while (n < 16u && scan_hi > 0u) {
        const u32 lo = scan_hi >= 64u ? scan_hi - 64u : 0u;
        const u32 idx = lo + lane;
        const bool nl = idx < scan_hi && text[idx] == '\n';
        unsigned long long mask = __ballot(nl);
        while (mask && n < 16u) {
            const u32 b = 63u - (u32)__builtin_clzll(mask);
            mask &= ~(1ull << b);
            const u32 start = lo + b + 1u;
            if (start < total) {
                if (lane == 0) ls[n] = start;
                n++;
            }
        }
        scan_hi = lo;
    }
    if (n < 16u && scan_hi == 0u) {
        if (lane == 0) ls[n] = 0u;
        n++;
    }
    __syncthreads();
    if (lane == 0) {
        u32 cut = 0, bad = 1;
        for (u32 i = 2; i < n; ++i)
            if (text[ls[i]] == '@' && text[ls[i - 2]] == '+') {
                cut = ls[i];
                bad = 0;
                break;
            }
        out[0] = cut;
        out[1] = bad;
    }
}

__global__ __launch_bounds__(64) void k_fastq_cut(const uint8_t *text, u32 total, u32 last, u32 *out) { fastq_cut_body(text, total, last, out); }

// ---------------------------------------------------------------------------------------------------------------------
// plain gzip, second half: the chain of chunks, markers looked up, the text's CRC-32
// ---------------------------------------------------------------------------------------------------------------------
// live[4 * i ..]: chunk index, symbols, text offset, bytes of real text in front of it (<= GZ_WINDOW)
__global__ __launch_bounds__(1024) void k_gz_chain(GzBatch B) {
    __shared__ uint8_t W[GZ_WINDOW];
    __shared__ u32 s_stop;
    const u32 tid = threadIdx.x;
    {
        const uint4 *src = (const uint4 *)B.window;
        uint4 *dst = (uint4 *)W;
        dst[tid] = src[tid];
        dst[tid + 1024u] = src[tid + 1024u];
    }
    __syncthreads();
    u32 ci = 0, n = 0, status = 0, valid = B.valid, kind = GZ_IDLE;
    u64 off = 0, end_bit = 0;
    for (;;) {
        const GzChunk c = B.recs[ci];
        kind = c.state & 255u;
        end_bit = c.end_bit;
        if (kind == GZ_IDLE || kind == GZ_FAILED) {
            status = (ci << 8) | (kind == GZ_FAILED ? ((c.state >> 8) & 255u) : (u32)BZ_BAD_BLOCK);
            break;
        }
        if (off + c.out_len > B.text_cap) {
            status = (ci << 8) | (u32)BZ_BAD_SIZE;
            break;
        }
        { // the window in front of this chunk
            uint4 *dst = (uint4 *)(B.win_in + (u64)ci * GZ_WINDOW);
            const uint4 *src = (const uint4 *)W;
            dst[tid] = src[tid];
            dst[tid + 1024u] = src[tid + 1024u];
        }
        if (tid == 0) {
            B.live[4u * n] = ci;
            B.live[4u * n + 1u] = c.out_len;
            B.live[4u * n + 2u] = (u32)off;
            B.live[4u * n + 3u] = valid;
        }
        for (u64 t = (off + 4095u) / 4096u + tid; t < (off + c.out_len + 4095u) / 4096u; t += 1024u) B.tile_map[t] = n;
        // the window behind it: the last GZ_WINDOW of [marker slots | symbols], markers looked up
        const uint16_t *tail = B.sym + (u64)ci * B.cap + c.out_len;
        u32 packed[8];
        {
            uint4 raw[4];
#pragma unroll
            for (u32 q = 0; q < 4u; ++q) __builtin_memcpy(&raw[q], tail + tid * 32u + q * 8u, 16);
            const uint16_t *e = (const uint16_t *)raw;
#pragma unroll
            for (u32 q = 0; q < 8u; ++q) {
                u32 w = 0;
#pragma unroll
                for (u32 z = 0; z < 4u; ++z) {
                    const u32 v = e[q * 4u + z];
                    w |= (v < 256u ? v : (u32)W[v & 0x7FFFu]) << (8u * z);
                }
                packed[q] = w;
            }
        }
        __syncthreads();
#pragma unroll
        for (u32 q = 0; q < 8u; ++q) ((u32 *)W)[tid * 8u + q] = packed[q];
        __syncthreads();
        off += c.out_len;
        valid = (u64)valid + c.out_len > GZ_WINDOW ? GZ_WINDOW : valid + c.out_len;
        n++;
        if (kind != GZ_NEXT) break;
        const u64 nx = c.end_bit / B.chunk_bits;
        if (nx <= ci || nx >= B.n_chunks) { // (k_gz_inflate stops with GZ_NEXT only at a later chunk's start)
            status = (ci << 8) | (u32)BZ_BAD_BLOCK;
            break;
        }
        ci = (u32)nx;
    }
    (void)s_stop;
    {
        uint4 *dst = (uint4 *)B.window;
        const uint4 *src = (const uint4 *)W;
        dst[tid] = src[tid];
        dst[tid + 1024u] = src[tid + 1024u];
    }
    if (tid == 0) {
        u32 *S = B.summary;
        S[GZS_STATUS] = status;
        S[GZS_N_LIVE] = n;
        S[GZS_TOTAL] = (u32)off;
        S[GZS_END_STATE] = kind;
        S[GZS_END_BIT_LO] = (u32)end_bit;
        S[GZS_END_BIT_HI] = (u32)(end_bit >> 32);
        S[GZS_VALID] = valid;
        u32 have = 0, crc = 0, isize = 0, trailing = 0;
        if (kind == GZ_MEMBER_END) {
            const u64 t = (end_bit + 7u) >> 3;
            if (t + 8u <= B.n_bytes) {
                const uint8_t *p = B.comp + t;
                have = 1;
                crc = p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
                isize = p[4] | ((u32)p[5] << 8) | ((u32)p[6] << 16) | ((u32)p[7] << 24);
                trailing = (u32)(B.n_bytes - t - 8u);
            }
        }
        S[GZS_HAVE_TRAILER] = have;
        S[GZS_CRC_WANT] = crc;
        S[GZS_ISIZE_WANT] = isize;
        S[GZS_TRAILING] = trailing;
    }
}

// one workgroup per 4096 bytes of text, 16 bytes a thread
__global__ __launch_bounds__(256) void k_gz_text(GzBatch B) {
    const u32 *S = B.summary;
    if (S[GZS_STATUS]) return;
    const u32 total = S[GZS_TOTAL];
    const u32 o = blockIdx.x * 4096u + threadIdx.x * 16u;
    if (o >= total) return;
    u32 li = B.tile_map[blockIdx.x];
    u32 ci = B.live[4u * li], len = B.live[4u * li + 1u], off = B.live[4u * li + 2u], valid = B.live[4u * li + 3u];
    while (o >= off + len) { // (the tile began in an earlier chunk than this thread's bytes do)
        li++;
        ci = B.live[4u * li], len = B.live[4u * li + 1u], off = B.live[4u * li + 2u], valid = B.live[4u * li + 3u];
    }
    uint8_t res[16];
    u32 e = 0, bad = 0;
    const u32 want = total - o < 16u ? total - o : 16u;
    while (e < want) {
        const u32 x = o + e - off;
        const u32 run = want - e < len - x ? want - e : len - x;
        const uint16_t *p = B.sym + (u64)ci * B.cap + GZ_WINDOW + x;
        const uint8_t *Wn = B.win_in + (u64)ci * GZ_WINDOW;
        uint16_t v16[16];
        if (run == 16u) {
            __builtin_memcpy(v16, p, 32);
        } else {
            for (u32 q = 0; q < run; ++q) v16[q] = p[q];
        }
        for (u32 q = 0; q < 16u; ++q) {
            if (q >= run) break;
            u32 v = v16[q];
            if (v >= 256u) {
                const u32 idx = v & 0x7FFFu;
                if (idx < GZ_WINDOW - valid) bad = 1; // before the first byte of the member
                v = Wn[idx];
            }
            res[e + q] = (uint8_t)v;
        }
        e += run;
        if (e < want) {
            do {
                li++;
                ci = B.live[4u * li], len = B.live[4u * li + 1u], off = B.live[4u * li + 2u], valid = B.live[4u * li + 3u];
            } while (len == 0u);
        }
    }
    uint8_t *dst = B.text + B.left + o;
    if (want == 16u) {
        __builtin_memcpy(dst, res, 16);
    } else {
        for (u32 q = 0; q < want; ++q) dst[q] = res[q];
    }
    if (bad) atomicMax(&B.summary[GZS_STATUS], (ci << 8) | (u32)BZ_BAD_MATCH);
}

// CRC-32 of text[left, left + total): a wavefront per 64 KiB slice, then one wavefront joins the slices
__global__ __launch_bounds__(256) void k_gz_crc_slices(GzBatch B, X2N x2n) {
    __shared__ u32 T[4][256];
    {
        u32 c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        T[0][threadIdx.x] = c;
    }
    __syncthreads();
    for (int t = 1; t < 4; ++t) {
        const u32 prev = T[t - 1][threadIdx.x];
        T[t][threadIdx.x] = (prev >> 8) ^ T[0][prev & 0xFFu];
        __syncthreads();
    }
    if (B.summary[GZS_STATUS]) return;
    const u32 total = B.summary[GZS_TOTAL];
    const u32 lane = threadIdx.x & 63u;
    const u32 si = blockIdx.x * 4u + (threadIdx.x >> 6);
    if ((u64)si * 65536u >= total) return;
    const u32 size = total - si * 65536u < 65536u ? total - si * 65536u : 65536u;
    const u32 slice = (((size + 63u) >> 6) + 3u) & ~3u;
    const u32 lo = lane * slice < size ? lane * slice : size;
    const u32 hi = lo + slice < size ? lo + slice : size;
    const uint8_t *p = B.text + B.left + (u64)si * 65536u;
    u32 c = 0xFFFFFFFFu, i = lo;
    for (; i < hi && ((uintptr_t)(p + i) & 3u); ++i) c = (c >> 8) ^ T[0][(c ^ p[i]) & 0xFFu];
    for (; i + 4u <= hi; i += 4u) {
        c ^= *(const u32 *)(p + i);
        c = T[3][c & 0xFFu] ^ T[2][(c >> 8) & 0xFFu] ^ T[1][(c >> 16) & 0xFFu] ^ T[0][c >> 24];
    }
    for (; i < hi; ++i) c = (c >> 8) ^ T[0][(c ^ p[i]) & 0xFFu];
    c ^= 0xFFFFFFFFu;
    u32 part = hi > lo ? crc_multmodp(crc_x2nmodp(x2n, size - hi, 3), c) : 0u;
    for (int off = 32; off > 0; off >>= 1) part ^= __shfl_xor(part, off);
    if (lane == 0) B.crc_tmp[si] = part;
}
__global__ __launch_bounds__(64) void k_gz_crc_join(GzBatch B, X2N x2n) {
    if (B.summary[GZS_STATUS]) return;
    const u32 total = B.summary[GZS_TOTAL];
    const u32 lane = threadIdx.x;
    const u32 n = (u32)(((u64)total + 65535u) >> 16);
    const u32 per = (n + 63u) / 64u;
    const u32 lo = lane * per < n ? lane * per : n, hi = lo + per < n ? lo + per : n;
    const u32 shift64k = crc_x2nmodp(x2n, 65536u, 3);
    u32 acc = 0;
    u64 bytes = 0;
    for (u32 s = lo; s < hi; ++s) {
        const u32 size = total - s * 65536u < 65536u ? total - s * 65536u : 65536u;
        const u32 op = size == 65536u ? shift64k : crc_x2nmodp(x2n, size, 3);
        acc = crc_multmodp(op, acc) ^ B.crc_tmp[s];
        bytes += size;
    }
    // the lanes' stretches, in order
    const u32 my_op = crc_x2nmodp(x2n, (u32)bytes, 3); // (a lane holds < 2^32 bytes: the whole text does)
    u32 crc = 0;
    for (u32 l = 0; l < 64u; ++l) {
        const u32 a = (u32)__builtin_amdgcn_readlane((int)acc, (int)l), op = (u32)__builtin_amdgcn_readlane((int)my_op, (int)l);
        const u32 nb = (u32)__builtin_amdgcn_readlane((int)(u32)bytes, (int)l);
        if (nb) crc = crc_multmodp(op, crc) ^ a;
    }
    if (lane == 0) B.summary[GZS_CRC] = crc;
}
// (launch_fastq_cut for a total only the device knows)
__global__ __launch_bounds__(64) void k_gz_cut(GzBatch B) {
    const u32 *S = B.summary;
    if (S[GZS_STATUS]) return;
    fastq_cut_body(B.text, B.left + S[GZS_TOTAL], S[GZS_END_STATE] == (u32)GZ_MEMBER_END ? 1u : 0u, B.summary + GZS_CUT);
}

static const X2N &x2n_table() {
    static const X2N x2n = [] {
        X2N x;
        u32 p = 1u << 30; // x^1
        x.t[0] = p;
        for (int k = 1; k < 32; ++k) x.t[k] = p = crc_multmodp(p, p);
        return x;
    }();
    return x2n;
}

uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    const X2N &x = x2n_table();
    u32 p = 1u << 31; // x^0
    u32 k = 3;        // bits -> bytes
    for (uint64_t n = len_b; n; n >>= 1, ++k)
        if (n & 1u) p = crc_multmodp(x.t[k & 31u], p);
    return crc_multmodp(p, crc_a) ^ crc_b;
}

hipError_t launch_gzip_batch(const GzBatch &b, hipStream_t st) {
    if (b.n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gz_find, dim3(b.n_chunks), dim3(64), 0, st, b.comp, b.n_bytes, b.chunk_bits, b.n_chunks, b.first_bit, b.recs);
    hipLaunchKernelGGL(k_gz_inflate, dim3(b.n_chunks), dim3(64), 0, st, b.comp, b.n_bytes, b.chunk_bits, b.n_chunks, b.recs, b.sym, b.cap);
    hipLaunchKernelGGL(k_gz_chain, dim3(1), dim3(1024), 0, st, b);
    const u32 n_tiles = (u32)((b.text_cap + 4095u) / 4096u);
    hipLaunchKernelGGL(k_gz_text, dim3(n_tiles), dim3(256), 0, st, b);
    const u32 n_slices = (u32)((b.text_cap + 65535u) / 65536u);
    hipLaunchKernelGGL(k_gz_crc_slices, dim3((n_slices + 3u) / 4u), dim3(256), 0, st, b, x2n_table());
    hipLaunchKernelGGL(k_gz_crc_join, dim3(1), dim3(64), 0, st, b, x2n_table());
    hipLaunchKernelGGL(k_gz_cut, dim3(1), dim3(64), 0, st, b);
    return hipGetLastError();
}

hipError_t launch_bgzf_inflate(const uint8_t *comp, const BgzfMember *members, uint32_t n_members, uint8_t *text,
                               uint32_t *status, hipStream_t st) {
    if (n_members == 0) return hipSuccess;
    const X2N &x2n = x2n_table();
    static const bool serial = getenv("FH_BGZF_SERIAL") != nullptr; // A/B: one symbol at a time
    if (serial) hipLaunchKernelGGL(k_bgzf_inflate<false>, dim3(n_members), dim3(64), 0, st, comp, members, n_members, text, status);
    else hipLaunchKernelGGL(k_bgzf_inflate<true>, dim3(n_members), dim3(64), 0, st, comp, members, n_members, text, status);
    hipLaunchKernelGGL(k_bgzf_crc, dim3((n_members + 3u) / 4u), dim3(256), 0, st, members, n_members, (const uint8_t *)text,
                       x2n, status);
    return hipGetLastError();
}

hipError_t launch_fastq_cut(const uint8_t *text, uint32_t total, uint32_t last, uint32_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_fastq_cut, dim3(1), dim3(64), 0, st, text, total, last, out);
    return hipGetLastError();
}

} // namespace fh
