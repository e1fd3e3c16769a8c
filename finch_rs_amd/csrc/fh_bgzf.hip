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
#include "fh_options.h"
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
// plain gzip: a chunk's symbols, the 32768 window slots in front included -- and the last GZ_RING of them kept in LDS, where the
// matches of a group of tokens that reach back into the group itself (a header copied from the record before, a run of one
// quality value) find their source without a round trip through the L2 (resolve_group_ring)
constexpr u32 GZ_RING = 1024;
#ifndef FH_GZ_COOP_LEN
#define FH_GZ_COOP_LEN 8
#endif
constexpr u32 GZ_COOP_LEN = FH_GZ_COOP_LEN; // matches longer than this are copied by the wavefront together (resolve_group_ring)
struct LdsGz : LdsT<u32> {
    uint16_t ring[GZ_RING]; // symbol at position p: ring[p % GZ_RING], for ring_from <= p < the end of the last group written
    u32 ring_from;
};
template <class L>
struct HasRing {
    static constexpr bool value = false;
};
template <>
struct HasRing<LdsGz> {
    static constexpr bool value = true;
};

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

// The same for the chunks of a plain gzip stream, through the ring: every symbol of the group is first written to LDS -- a
// literal as it is, a match from the ring where its source is recent, from global memory where it lies further back (written
// there by earlier groups: one wait at the start covers them) -- and a lane whose source lies inside the group waits for an
// LDS round, not for stores to reach the L2 and come back; then the group goes out to global memory in one piece, 16 bytes a
// lane.  (Writing straight to global memory and waiting for it between rounds was half of k_gz_chunks' time.)  A group longer
// than the ring goes the long way and leaves the ring empty.
__device__ __noinline__ void resolve_group_ring(LdsGz &L, uint16_t *out, u32 tpos, u32 tinfo, u32 ntok, u32 lane) {
    constexpr u32 M = GZ_RING - 1u;
    const u32 len = tinfo & 0x1FFu, hi = tinfo >> 16;
    const u32 adv = len ? len : 1u + ((tinfo >> 9) & 1u);
    const u32 G0 = (u32)__builtin_amdgcn_readlane((int)tpos, 0);
    const u32 G1 = (u32)__builtin_amdgcn_readlane((int)(tpos + adv), (int)(ntok - 1u));
    u32 R0 = rfl(L.ring_from);
    if (G1 - G0 > GZ_RING || R0 > G0) {
        resolve_group(out, tpos, tinfo, ntok, lane);
        if (lane == 0) L.ring_from = G1;
        return;
    }
    if (G1 > GZ_RING && R0 < G1 - GZ_RING) R0 = G1 - GZ_RING; // (what lies further back is overwritten by this group)
    // (out of line, this function sees L through a generic pointer: the ring is addressed as what it is, LDS)
    __attribute__((address_space(3))) uint16_t *const ring = (__attribute__((address_space(3))) uint16_t *)L.ring;
    bool done = lane >= ntok;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, BZ_FENCE_SCOPE); // earlier groups' symbols are in global memory
    for (;;) {
        const unsigned long long pending = __ballot(!done);
        if (pending == 0ull) break;
        const u32 W = (u32)__builtin_amdgcn_readlane((int)tpos, (int)__builtin_ctzll(pending));
        if (!done) {
            if (len == 0u) {
                ring[tpos & M] = (uint16_t)(hi & 0xFFu);
                if (tinfo & 0x200u) ring[(tpos + 1u) & M] = (uint16_t)((hi >> 8) & 0xFFu);
                done = true;
            } else {
                const u32 src = tpos - hi;
                const u32 src_end = src + (len < hi ? len : hi); // (an overlapping match re-reads its own symbols)
                if (src_end <= W && len <= GZ_COOP_LEN) {
                    if (src_end <= R0 && hi >= len) { // all of it further back than the ring: four symbols a load
                        const uint16_t *sp = out + src;
                        for (u32 j = 0; j < len; j += 32u) {
                            const u32 n = len - j < 32u ? len - j : 32u;
                            u64 r[8];
#pragma unroll
                            for (u32 q = 0; q < 8u; ++q)
                                if (4u * q < n) __builtin_memcpy(&r[q], sp + j + 4u * q, 8); // (may read up to 3 symbols past the match: not kept)
#pragma unroll
                            for (u32 q = 0; q < 8u; ++q) {
                                u64 w = r[q];
#pragma unroll
                                for (u32 z = 0; z < 4u; ++z, w >>= 16)
                                    if (4u * q + z < n) ring[(tpos + j + 4u * q + z) & M] = (uint16_t)w;
                            }
                        }
                    } else {
                        for (u32 j = 0; j < len; ++j) {
                            const u32 p = src + j;
                            ring[(tpos + j) & M] = p >= R0 ? ring[p & M] : out[p];
                        }
                    }
                    done = true;
                }
            }
        }
        // A long match would hold the other 63 lanes up for as many steps as it has symbols (a run of one quality value: 150
        // of them in every record): the wavefront copies those together, 64 symbols a step.  Its source is complete (that is
        // what "ready" means), and where a match overlaps itself every symbol is its pattern's, src + j mod dist.
        unsigned long long longs = __ballot(!done && len > GZ_COOP_LEN && (tpos - hi) + (len < hi ? len : hi) <= W);
        while (longs) {
            const int l = (int)__builtin_ctzll(longs);
            longs &= longs - 1ull;
            const u32 t_pos = (u32)__builtin_amdgcn_readlane((int)tpos, l), t_len = (u32)__builtin_amdgcn_readlane((int)len, l);
            const u32 t_dist = (u32)__builtin_amdgcn_readlane((int)hi, l);
            const u32 t_src = t_pos - t_dist;
            for (u32 j = lane; j < t_len; j += 64u) {
                const u32 p = t_src + (t_dist >= t_len ? j : j % t_dist);
                uint16_t v;
                if (p >= R0) v = ring[p & M];
                else v = __builtin_nontemporal_load(out + p);
                ring[(t_pos + j) & M] = v;
            }
            if ((int)lane == l) done = true;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // (the LDS serves a wavefront's accesses in order: nothing to wait for)
    }
    // the group's symbols, in one piece
    const u32 span = G1 - G0;
    for (u32 e = lane * 8u; e < span; e += 512u) {
        uint16_t v[8];
#pragma unroll
        for (u32 q = 0; q < 8u; ++q) v[q] = ring[(G0 + e + q) & M];
        if (e + 8u <= span) {
            __builtin_memcpy(out + G0 + e, v, 16);
        } else {
            for (u32 q = 0; e + q < span; ++q) out[G0 + e + q] = v[q];
        }
    }
    if (lane == 0) L.ring_from = R0;
}

// write out the queued tokens, 64 at a time (`all`: the rest too)
template <class LDS, class T>
__device__ void flush_queue(LDS &L, T *out, u32 &qn, u32 lane, bool all) {
    while (qn >= 64u || (all && qn > 0u)) {
        const u32 n = qn < 64u ? qn : 64u;
        __syncthreads();
        const u32 tpos = L.qpos[lane], tinfo = L.qinfo[lane];
#ifndef FH_GZ_NO_RING // (A/B builds)
        if constexpr (HasRing<LDS>::value) resolve_group_ring(L, out, tpos, tinfo, n, lane);
        else
#endif
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
// grow(need, limit, mbits): the output would reach `need` elements, beyond `limit`: may the limit be raised (it does so)?
struct NoGrow {
    __device__ __forceinline__ bool operator()(u32, u32 &, u64) const { return false; }
};
template <class LDS, class T, class GROW = NoGrow>
__device__ u32 block_symbols_parallel(LDS &L, const uint8_t *mbase, u64 in_bits, u32 isize, T *out, u64 &mbits_ref, u32 &pos_ref,
                                      u32 &qn_ref, u32 lane, GROW grow = GROW(), u32 *limit_out = nullptr) {
    u64 mbits = mbits_ref;
    u32 pos = pos_ref, qn = qn_ref, fail = 0;
    // WINDOW (the chunks of a plain gzip stream): the wavefronts spend seven eighths of their time waiting, and the load of the
    // stream's bits -- its address hangs on the symbols of the round before -- is the longest wait of a round.  So every lane
    // keeps 24 bytes of the stream from ITS offset on, loaded one round ahead: a round moves on by at most 111 bits (63 + the
    // longest symbol), so the 64 bits a lane needs next lie inside what it loaded for this round, a shift away.
#ifdef FH_GZ_NO_WINDOW // (A/B builds)
    constexpr bool WINDOW = false;
#else
    constexpr bool WINDOW = !__is_same(GROW, NoGrow);
#endif
    u64 w0 = 0, w1 = 0, w2 = 0, wbase = mbits;
    if (WINDOW) {
        const uint8_t *p = mbase + ((mbits + lane) >> 3);
        __builtin_memcpy(&w0, p, 8);
        __builtin_memcpy(&w1, p + 8, 8);
        __builtin_memcpy(&w2, p + 16, 8);
    }
    for (;;) {
        if (mbits > in_bits + 64u) {
            fail = BZ_OVERRUN_;
            break;
        }
        const u64 my = mbits + lane;
        u64 b;
        if (WINDOW) {
            const u32 off = (u32)(mbits - wbase) + (u32)((wbase + lane) & 7u); // of this round's first bit within the lane's 24 bytes
            const u64 lo = off < 64u ? w0 : w1, hi = off < 64u ? w1 : w2;
            const u32 sh = off & 63u;
            b = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
            // the next round's, from this round's offset on
            const uint8_t *p = mbase + (my >> 3);
            __builtin_memcpy(&w0, p, 8);
            __builtin_memcpy(&w1, p + 8, 8);
            __builtin_memcpy(&w2, p + 16, 8);
            wbase = mbits;
        } else {
            __builtin_memcpy(&b, mbase + (my >> 3), 8); // (reads at most 24 bytes past the member: inside the buffer's padding)
            b >>= (u32)(my & 7u);
        }
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
        if (pos + total > isize && !grow(pos + total, isize, mbits)) {
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
    if (limit_out) *limit_out = isize;
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
// with a wavefront per chunk of ~a block's worth of compressed bytes.
//   k_gz_chunks   a chunk's wavefront first looks for a block start in its range: 512 bit offsets a step through the cheap
//                 tests (the three type bits of a non-final dynamic block, HLIT / HDIST in range, a complete code-length
//                 code: one table lookup per nine bits), the survivors collected and taken 64 at a time, a lane each, through
//                 an exact parse of the header that stops as soon as the literal or distance code is over-subscribed, and
//                 what is left (complete codes) one by one through the real tables and 512 symbols that have to be text.
//                 From the start it decodes into 16-bit symbols behind 32768 marker slots (what lies in front of a chunk is
//                 unknown: a match that reaches there copies markers), block by block, until a block boundary beyond its
//                 range passes the very same tests -- that is where the wavefront of that range began --, the stream ends,
//                 or the bytes do.
//   k_gz_chain    one workgroup follows the chunks that really continue each other (a start that was none is never reached)
//                 and lays out their text; k_gz_windows gives every one of them the 32 KiB in front of it (the tail of the
//                 one before, its markers looked up in the window before that -- composed over groups of chunks so that the
//                 groups run side by side).
//   k_gz_text     symbols narrowed to bytes, markers looked up; then CRC-32 of the batch's text (k_gz_crc_*).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr u32 GZ_QCAP = 448;    // candidates waiting for the exact header parse
constexpr u64 GZ_TAIL_GUARD = 4608; // no start is looked for in the last bytes of what is there: header (<= 563 bytes) and 512 symbols (<= 3 KiB) of one must lie inside
struct GzFindLds { // (laid over the literal table while a start is being looked for)
    uint8_t lut[512]; // Kraft sum, in 1/128, of three 3-bit code lengths
    u32 n;
    u32 queue[GZ_QCAP];
};
static_assert(sizeof(GzFindLds) <= sizeof(u32) << LIT_BITS, "the finder's scratch lies over the literal table");

__device__ __forceinline__ bool gz_text_byte(u32 c) { return c == '\n' || c == '\r' || c == '\t' || (c >= 32u && c < 127u); }
__device__ __forceinline__ u64 gz_peek(const uint8_t *comp, u64 bit) { // >= 57 bits of the stream from bit `bit`
    u64 a;
    __builtin_memcpy(&a, comp + (bit >> 3), 8);
    return a >> (u32)(bit & 7u);
}
__device__ void gz_fill_lut(GzFindLds &F, u32 lane) {
    for (u32 i = lane; i < 512u; i += 64u) {
        u32 s = 0;
        for (u32 f = 0; f < 3u; ++f) {
            const u32 l = (i >> (3u * f)) & 7u;
            if (l) s += 128u >> l;
        }
        F.lut[i] = (uint8_t)s;
    }
    if (lane == 0) F.n = 0;
    __syncthreads();
}
// The cheap tests of a bit offset whose three type bits read "non-final, dynamic": v = the stream's bits from it on, vh =
// those from its bit 64 on.  HLIT / HDIST in range, and the code-length code complete (every encoder's is).
template <bool LUT>
__device__ __forceinline__ bool gz_stage_a(const GzFindLds *F, u64 v, u64 vh) {
    const u32 hl = (u32)(v >> 3) & 31u, hd = (u32)(v >> 8) & 31u, hclen = ((u32)(v >> 13) & 15u) + 4u;
    if (hl > 29u || hd > 29u) return false;
    u64 w = (v >> 17) | (vh << 47); // the 3-bit lengths
    w &= (1ull << (3u * hclen)) - 1ull;
    u32 k = 0;
    if (LUT) {
#pragma unroll
        for (u32 f = 0; f < 7u; ++f) k += F->lut[(u32)(w >> (9u * f)) & 511u];
    } else {
        for (u32 f = 0; f < 19u; ++f) {
            const u32 l = (u32)(w >> (3u * f)) & 7u;
            if (l) k += 128u >> l;
        }
    }
    return k == 128u;
}

// The exact parse of a dynamic block's header at bit `pos`, by one lane: true if the hlit + hdist code lengths can be read
// and describe a complete literal/length code with an end-of-block symbol and a distance code that is complete, a single
// code or empty.  Gives up as soon as either code is over-subscribed (random bits are after some thirty lengths).
__device__ __noinline__ bool gz_stage_b(const uint8_t *comp, u64 pos, u64 in_bits) {
    const u64 v = gz_peek(comp, pos);
    const u32 hlit = ((u32)(v >> 3) & 31u) + 257u, hdist = ((u32)(v >> 8) & 31u) + 1u, hclen = ((u32)(v >> 13) & 15u) + 4u;
    const u64 w = gz_peek(comp, pos + 17u) & ((1ull << (3u * hclen)) - 1ull);
    // the code-length code: field i of w is the length of symbol ORD[i]
    constexpr u32 ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    u32 cl[19];
#pragma unroll
    for (u32 i = 0; i < 19u; ++i) cl[ORD[i]] = (u32)(w >> (3u * i)) & 7u;
    u32 cnt[8], first[8], offs[8];
#pragma unroll
    for (u32 l = 1; l <= 7u; ++l) {
        u32 c = 0;
#pragma unroll
        for (u32 s = 0; s < 19u; ++s) c += cl[s] == l;
        cnt[l] = c;
    }
    {
        u32 code = 0, off = 0;
#pragma unroll
        for (u32 l = 1; l <= 7u; ++l) {
            first[l] = code;
            offs[l] = off;
            code = (code + cnt[l]) << 1;
            off += cnt[l];
        }
    }
    u64 srt_lo = 0, srt_hi = 0; // the symbols in canonical order, five bits each
    {
        u32 k = 0;
#pragma unroll
        for (u32 l = 1; l <= 7u; ++l)
#pragma unroll
            for (u32 s = 0; s < 19u; ++s)
                if (cl[s] == l) {
                    if (k < 12u) srt_lo |= (u64)s << (5u * k);
                    else srt_hi |= (u64)s << (5u * (k - 12u));
                    k++;
                }
    }
    const u32 n = hlit + hdist;
    u64 bp = pos + 17u + 3u * hclen;
    u32 i = 0, prev = 0, kl = 0, kd = 0, nd = 0;
    bool eob = false;
    while (i < n) {
        if (bp + 64u > in_bits) return false;
        u64 b = gz_peek(comp, bp);
        u32 code = 0, sym = 0, used = 0;
#pragma unroll
        for (u32 l = 1; l <= 7u; ++l) {
            code = (code << 1) | ((u32)(b >> (l - 1u)) & 1u);
            const u32 idx = code - first[l];
            if (used == 0u && idx < cnt[l]) {
                const u32 k = offs[l] + idx;
                sym = k < 12u ? (u32)(srt_lo >> (5u * k)) & 31u : (u32)(srt_hi >> (5u * (k - 12u))) & 31u;
                used = l;
            }
        }
        if (used == 0u) return false;
        b >>= used;
        bp += used;
        u32 rep = 1, val = sym;
        if (sym == 16u) {
            if (i == 0u) return false;
            rep = 3u + ((u32)b & 3u);
            bp += 2u;
            val = prev;
        } else if (sym == 17u) {
            rep = 3u + ((u32)b & 7u);
            bp += 3u;
            val = 0;
        } else if (sym == 18u) {
            rep = 11u + ((u32)b & 127u);
            bp += 7u;
            val = 0;
        }
        if (i + rep > n) return false;
        if (val) {
            const u32 add = 32768u >> val;
            for (u32 r = 0; r < rep; ++r) {
                const u32 at = i + r;
                if (at < hlit) kl += add;
                else kd += add, nd++;
                if (at == 256u) eob = true;
            }
            if (kl > 32768u || kd > 32768u) return false;
        }
        prev = val;
        i += rep;
    }
    return kl == 32768u && eob && (kd == 32768u || kd == 0u || (kd == 16384u && nd == 1u));
}

// Does a non-final dynamic block with complete codes, whose first symbols are text, begin at bit `pos`?  The real tables,
// wave-uniform; overwrites the block tables (and the finder's scratch over them).
__device__ __noinline__ bool gz_block_check(LdsGz &L, const uint8_t *comp, u64 n_bytes, u64 pos, u32 lane) {
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
        // end of block, or no such code.  (A block of a few symbols tells nothing: one bit pattern in fifty reads as "a text
        // byte, then the end" under a random complete code.  Encoders close a block that short only at a flush, and a real
        // one that is turned down here is simply decoded by the chunk in front of it.)
        if (kind != KIND_BASE) return (e & 15u) != 0u && sym >= 64u;
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

__device__ __forceinline__ u32 wave_min_u32(u32 v) {
    for (int off = 32; off > 0; off >>= 1) {
        const u32 o = (u32)__shfl_xor((int)v, off);
        v = o < v ? o : v;
    }
    return v;
}

// The candidates collected so far, through the exact parse (a lane each) and the real tables (in stream order): the first
// bit offset that passes, or GZ_NONE.  `base`: what the queue's offsets count from.
__device__ __noinline__ u64 gz_flush_candidates(LdsGz &L, const uint8_t *comp, u64 n_bytes, u64 base, u32 lane) {
    GzFindLds &F = *(GzFindLds *)L.lit;
    __syncthreads();
    const u32 n = F.n < GZ_QCAP ? F.n : GZ_QCAP;
    const u64 in_bits = n_bytes * 8u;
    u32 mine = 0xFFFFFFFFu;
    for (u32 g = 0; g < n; g += 64u)
        if (g + lane < n) {
            const u32 rel = F.queue[g + lane];
            if (rel < mine && gz_stage_b(comp, base + rel, in_bits)) mine = rel;
        }
    __syncthreads();
    u64 best = GZ_NONE;
    bool clobbered = false;
    for (;;) {
        const u32 mn = wave_min_u32(mine);
        if (mn == 0xFFFFFFFFu) break;
        clobbered = true;
        if (gz_block_check(L, comp, n_bytes, base + mn, lane)) {
            best = base + mn;
            break;
        }
        if (mine == mn) mine = 0xFFFFFFFFu;
    }
    if (best == GZ_NONE) {
        __syncthreads();
        if (clobbered) gz_fill_lut(F, lane);
        else if (lane == 0) F.n = 0;
        __syncthreads();
    }
    return best;
}

// first bit offset in [from, to) at which a plausible block begins; GZ_NONE: nowhere.  (to <= (n_bytes - GZ_TAIL_GUARD) * 8)
__device__ __noinline__ u64 gz_find_start(LdsGz &L, const uint8_t *comp, u64 n_bytes, u64 from, u64 to, u32 lane) {
    if (from >= to) return GZ_NONE;
    GzFindLds &F = *(GzFindLds *)L.lit;
    gz_fill_lut(F, lane);
    const u64 base = from & ~7ull;
    u64 found = GZ_NONE;
    for (u64 b0 = from >> 3; b0 * 8u < to && found == GZ_NONE; b0 += 64u) {
        const u64 byte = b0 + lane;
        if (byte * 8u < to) {
            u64 lo, hi;
            __builtin_memcpy(&lo, comp + byte, 8);
            __builtin_memcpy(&hi, comp + byte + 8u, 8);
            u32 m = (u32)(~lo & ~(lo >> 1) & (lo >> 2)) & 0xFFu; // offsets whose three bits read 0, 0, 1: non-final, dynamic
            while (m) {
                const u32 o = (u32)__builtin_ctz(m);
                m &= m - 1u;
                const u64 pos = byte * 8u + o;
                if (pos < from || pos >= to) continue;
                const u64 v = o ? (lo >> o) | (hi << (64u - o)) : lo;
                if (gz_stage_a<true>(&F, v, hi >> o)) {
                    const u32 slot = atomicAdd(&F.n, 1u);
                    if (slot < GZ_QCAP) F.queue[slot] = (u32)(pos - base);
                }
            }
        }
        __syncthreads();
        if (F.n >= 64u) found = gz_flush_candidates(L, comp, n_bytes, base, lane);
    }
    if (found == GZ_NONE) found = gz_flush_candidates(L, comp, n_bytes, base, lane);
    __syncthreads();
    return found;
}

// The same verdict for ONE bit offset (a block boundary the decoder stands at): would gz_find_start stop here?
__device__ __noinline__ bool gz_is_start(LdsGz &L, const uint8_t *comp, u64 n_bytes, u64 pos, u32 lane) {
    u64 lo, hi;
    __builtin_memcpy(&lo, comp + (pos >> 3), 8);
    __builtin_memcpy(&hi, comp + (pos >> 3) + 8u, 8);
    const u32 o = (u32)(pos & 7u);
    const u64 v = o ? (lo >> o) | (hi << (64u - o)) : lo;
    if ((v & 7u) != 4u) return false;
    if (!gz_stage_a<false>(nullptr, v, hi >> o)) return false;
    if (!gz_stage_b(comp, pos, n_bytes * 8u)) return false;
    return gz_block_check(L, comp, n_bytes, pos, lane);
}
} // namespace

// chunks [c0, c0 + gridDim.x) of the batch; comp[0, n_bytes) is what has arrived of it (`final`: all there will ever be of the
// stream), chunk c covers bits [c, c + 1) * chunk_bits
// A chunk's symbols go to its own stretch of the symbol buffer (`cap` slots) and on into those of the chunks behind it
// whose ranges it has decoded all the way through: nothing on the chain begins in such a range (a boundary in it that
// read as a start would have stopped this chunk), so the stretch is free unless a wavefront that began at a start that is
// none has taken it -- claims[c]: 0, or 1 + the chunk that writes stretch c.
struct GzGrow {
    u32 ci, n_regions, *claims, *owned;
    u64 cap, chunk_bits;
    __device__ __forceinline__ bool operator()(u32 need, u32 &limit, u64 mbits) const {
        while (limit < need) {
            const u32 j = ci + *owned;
            if (j >= n_regions || mbits < (u64)(j + 1u) * chunk_bits || limit + cap > 0x7FFFFFF0ull) return false;
            u32 old = 0;
            if (threadIdx.x == 0) old = atomicCAS(&claims[j], 0u, ci + 1u);
            if (rfl(old) != 0u) return false;
            *owned += 1u;
            limit += (u32)cap;
        }
        return true;
    }
};

constexpr u64 GZ_WAIT_TICKS = 300000000ull; // s_memrealtime ticks (100 MHz: three seconds) a wavefront waits for its bytes before it gives up

// Wait until `need` bytes of the batch are on the device, or all of it is: how many there are (n_bytes) and whether more will
// come (state, GzFeed).  false: told to give up, or nothing moved for three seconds (a reader stalled that long: the host-side inflate takes the file).
__device__ __noinline__ bool gz_wait_bytes(const GzFeed *feed, u64 need, u64 &n_bytes, u32 &state, u32 lane) {
    u32 ok = 1, st = 0, av_lo = 0, av_hi = 0;
    if (lane == 0) {
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            if (__hip_atomic_load(&feed->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                ok = 0;
                break;
            }
            st = __hip_atomic_load(&feed->state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); // (before avail: the host writes it last)
            const u64 av = __hip_atomic_load(&feed->avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            av_lo = (u32)av, av_hi = (u32)(av >> 32);
            if (st || av >= need) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > GZ_WAIT_TICKS) {
                ok = 0;
                break;
            }
            // (every poll is a read across the PCIe link, and thousands of wavefronts may be waiting while the batch's own
            // bytes come in over it: the further away a wavefront's bytes are, the longer it sleeps -- 3.4 us per 32 KiB
            // still missing, what arrives in that time at 10 GB/s, half a millisecond at most)
            const u64 missing = need - av;
            const u32 naps = missing >> 15 > 128u ? 128u : (u32)(missing >> 15);
            for (u32 i = 0; i <= naps; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
    ok = rfl(ok), st = rfl(st), av_lo = rfl(av_lo), av_hi = rfl(av_hi);
#ifndef FH_GZ_NO_FENCE // (A/B builds)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // what the copy engine wrote, not what the caches remember of the last batch
#endif
    n_bytes = av_lo | ((u64)av_hi << 32);
    state = st;
    return ok != 0u;
}

__global__ __launch_bounds__(64, 4) void k_gz_chunks(const uint8_t *comp, const GzFeed *feed, u64 n_bytes, u64 chunk_bits, u32 c0, u64 first_bit,
                                                     u32 final, GzChunk *recs, uint16_t *sym, u64 cap, u32 *claims, u32 n_regions, u64 *times) {
    __shared__ LdsGz L;
    const u32 ci = c0 + blockIdx.x, lane = threadIdx.x;
    if (times && lane == 0) times[3u * ci] = __builtin_amdgcn_s_memrealtime(); // (FH_GZ_TIMES: dispatched / bytes there / done)
    u32 feed_state = 1; // (all of the batch is there)
    if (feed) {
        // the batch is still coming in: wait until this chunk's bytes and a block's worth behind them are there
        if (!gz_wait_bytes(feed, (u64)(ci + 1u) * (chunk_bits >> 3) + GZ_LOOKAHEAD, n_bytes, feed_state, lane)) {
            if (lane == 0) recs[ci] = GzChunk{GZ_NONE, GZ_NONE, 0u, (u32)GZ_FAILED | ((u32)BZ_OVERRUN << 8)};
            return;
        }
        final = feed_state == 2u;
        if ((u64)ci * (chunk_bits >> 3) >= n_bytes) return; // (the batch ended in front of this chunk)
    }
    u64 in_bits = n_bytes * 8u;
    u64 search_end = n_bytes > GZ_TAIL_GUARD ? (n_bytes - GZ_TAIL_GUARD) * 8u : 0u; // no start is looked for beyond
    const u64 range_end = (u64)(ci + 1u) * chunk_bits;
    if (times && lane == 0) times[3u * ci + 1u] = __builtin_amdgcn_s_memrealtime();
    u64 start;
    if (ci == 0u) {
        start = first_bit;
    } else {
        u64 from = (u64)ci * chunk_bits;
        if (from <= first_bit) from = first_bit + 1u;
        start = gz_find_start(L, comp, n_bytes, from, range_end < search_end ? range_end : search_end, lane);
    }
    if (start == GZ_NONE) {
        if (lane == 0) recs[ci] = GzChunk{GZ_NONE, GZ_NONE, 0u, (u32)GZ_IDLE};
        return;
    }
    {
        u32 old = 0;
        if (lane == 0) old = atomicCAS(&claims[ci], 0u, ci + 1u);
        if (rfl(old) != 0u) { // (its stretch of the symbol buffer holds the overflow of a chunk in front of it)
            if (lane == 0) recs[ci] = GzChunk{start, start, 0u, (u32)GZ_FAILED | ((u32)BZ_BAD_SIZE << 8)};
            return;
        }
    }
    uint16_t *out = sym + (u64)ci * cap;
    for (u32 j = lane; j < GZ_WINDOW / 2u; j += 64u) ((u32 *)out)[j] = (0x8000u | (2u * j)) | ((0x8001u | (2u * j)) << 16);
    if (lane == 0) L.ring_from = GZ_WINDOW; // (nothing in the ring yet)
    __syncthreads();
    u32 limit = cap > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (u32)cap, owned = 1;
    const GzGrow grow{ci, n_regions, claims, &owned, cap, chunk_bits};
    Reader r;
    rd_init(r, comp, start >> 3, n_bytes, (start >> 3) * 8u, lane);
    rd_fill(r, lane);
    rd_take(r, (u32)(start & 7u));
    u32 fail = 0, state = GZ_FAILED, pos = GZ_WINDOW, qn = 0;
    u64 end_bit = start, fail_at = 0;
    u32 end_pos = pos;
    bool first_block = true;
    for (;;) {
        // a block boundary: what has been decoded up to here stands whatever becomes of the next block
        end_bit = rd_used_bits(r);
        end_pos = pos;
        if (feed_state == 0u && end_bit + GZ_LOOKAHEAD * 8u > in_bits) {
            // this chunk has decoded on and on (stored blocks, boundaries that do not read as starts) to the end of what was
            // there when it set out, and the batch is still coming in: wait for more of it
            if (!gz_wait_bytes(feed, (end_bit >> 3) + GZ_LOOKAHEAD, n_bytes, feed_state, lane)) {
                fail = BZ_OVERRUN;
                fail_at = 0;
                break;
            }
            final = feed_state == 2u;
            in_bits = n_bytes * 8u;
            search_end = n_bytes > GZ_TAIL_GUARD ? (n_bytes - GZ_TAIL_GUARD) * 8u : 0u;
            rd_init(r, comp, end_bit >> 3, n_bytes, (end_bit >> 3) * 8u, lane);
            rd_fill(r, lane);
            rd_take(r, (u32)(end_bit & 7u));
        }
        if (end_bit + 3u > in_bits) {
            state = GZ_OUT_OF_INPUT;
            break;
        }
        if (!first_block && end_bit >= range_end) {
            // beyond this chunk's range: the wavefront of the range this boundary lies in began here if it reads as a start
            if (end_bit >= search_end) {
                if (!final) { // (nobody looks for starts this close to the end of what is there)
                    state = GZ_OUT_OF_INPUT;
                    break;
                }
            } else {
                flush_queue(L, out, qn, lane, true); // (the tests overwrite the tables and the queue's LDS neighbours)
                if (gz_is_start(L, comp, n_bytes, end_bit, lane)) {
                    state = GZ_NEXT;
                    break;
                }
            }
        }
        first_block = false;
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
            if ((u64)pos + len > limit && !grow(pos + len, limit, (used + len) * 8u)) {
                fail = BZ_BAD_SIZE;
                break;
            }
            flush_queue(L, out, qn, lane, true);
            const uint8_t *src = comp + used;
            for (u32 j = lane; j < len; j += 64u) out[pos + j] = src[j];
            pos += len;
            if (lane == 0) L.ring_from = pos; // (these went straight to global memory)
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
            fail = block_symbols_parallel(L, comp, in_bits, limit, out, mb, pos, qn, lane, grow, &limit);
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
    }
    if (fail) {
        // (what was decoded from the padding behind the last byte proves nothing: the block is cut short by the batch's end)
        if (fail != BZ_BAD_SIZE && fail_at + 512u > in_bits) state = GZ_OUT_OF_INPUT;
        else state = (u32)GZ_FAILED | (fail << 8);
    }
    flush_queue(L, out, qn, lane, true);
    if (lane == 0) recs[ci] = GzChunk{start, end_bit, end_pos - GZ_WINDOW, state};
    if (times && lane == 0) times[3u * ci + 2u] = __builtin_amdgcn_s_memrealtime();
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
// the same table where a kernel can index it with a register: as a kernel argument it lives in scalar registers, and an index
// that is not a constant sends the whole of it through scratch memory -- k_gz_crc_join spent 0.4 ms on 5 000 multiplications
constexpr u32 crc_multmodp_c(u32 a, u32 b) {
    u32 m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0u) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
struct X2NTable {
    u32 t[32];
    constexpr X2NTable() : t() {
        u32 p = 1u << 30; // x^1
        t[0] = p;
        for (int k = 1; k < 32; ++k) t[k] = p = crc_multmodp_c(p, p);
    }
};
__device__ __constant__ X2NTable g_x2n = X2NTable();
__device__ inline u32 crc_x2n(u32 n, u32 k) { // x^(n * 2^k) mod P
    u32 p = 1u << 31;
    while (n) {
        if (n & 1u) p = crc_multmodp(g_x2n.t[k & 31u], p);
        n >>= 1;
        k++;
    }
    return p;
}
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

// CRC-32 of p[0, size) by one wavefront, 4 KiB at a time: every lane checksums 64 consecutive bytes of the block (a load
// instruction of the wavefront then covers whole cache lines, and the next three use the rest of them -- a lane walking
// through a KiB of its own made every load fetch 64 lines for 256 bytes, sixteen times the text through the L1), the 64
// checksums are shifted past what follows them in the block and folded, and the block's joins those of the blocks before it.
// T: the four slicing tables (LDS); ops[lane]: x^(8 * 64 * (63 - lane)) mod P, op_block: x^(8 * 4096) mod P.
struct CrcLaneOps {
    u32 piece, block;
};
__device__ __forceinline__ CrcLaneOps crc_lane_ops(const X2N &x2n, u32 lane) {
    return CrcLaneOps{crc_x2n(64u * (63u - lane), 3), crc_x2n(4096u, 3)};
}
__device__ __forceinline__ u32 wave_crc32(const uint8_t *p, u32 size, const u32 (*T)[256], const X2N &x2n, const CrcLaneOps &ops, u32 lane) {
    u32 running = 0;
    for (u32 b0 = 0; b0 < size; b0 += 4096u) {
        const u32 blen = size - b0 < 4096u ? size - b0 : 4096u;
        const u32 o = lane * 64u;
        const u32 n = o < blen ? (blen - o < 64u ? blen - o : 64u) : 0u;
        const uint8_t *q = p + b0 + o;
        u32 c = 0xFFFFFFFFu;
        if (n == 64u) {
            uint4 w[4];
#pragma unroll
            for (u32 k = 0; k < 4u; ++k) __builtin_memcpy(&w[k], q + 16u * k, 16);
            const u32 *v = (const u32 *)w;
#pragma unroll
            for (u32 k = 0; k < 16u; ++k) {
                c ^= v[k];
                c = T[3][c & 0xFFu] ^ T[2][(c >> 8) & 0xFFu] ^ T[1][(c >> 16) & 0xFFu] ^ T[0][c >> 24];
            }
        } else {
            for (u32 i = 0; i < n; ++i) c = (c >> 8) ^ T[0][(c ^ q[i]) & 0xFFu];
        }
        c ^= 0xFFFFFFFFu;
        u32 part = 0;
        if (n) {
            // (a full block: the lane's operator is ready; the last, shorter one: worked out from what follows the piece)
            const u32 op = blen == 4096u ? ops.piece : crc_x2n(blen - o - n, 3);
            part = crc_multmodp(op, c);
        }
        for (int off = 32; off > 0; off >>= 1) part ^= __shfl_xor(part, off);
        running = crc_multmodp(blen == 4096u ? ops.block : crc_x2n(blen, 3), running) ^ part;
    }
    return running;
}

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
    // (a lane per sixty-fourth of the member; wave_crc32's layout -- made for the long text of a gzip batch -- is no faster here:
    // 167 against 159 us per launch of ~1500 members, profiles/r04_crc_ab.txt)
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
__device__ __forceinline__ void lds_barrier() { // (not __syncthreads: loads of the next tail stay in flight across it)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
constexpr u32 GZC_END_MEMBER = 0xFFFFFFF1u, GZC_END_INPUT = 0xFFFFFFF2u, GZC_BROKEN = 0xFFFFFFF3u, GZC_FAILED = 0xFFFFFFF4u;

__global__ __launch_bounds__(1024) void k_gz_chain(GzBatch B) {
    __shared__ u32 nxt[GZ_MAX_CHUNKS]; // the chunk that continues chunk c, or why none does
    __shared__ u32 part[1024];
    __shared__ u32 s_n, s_status, s_kind, s_last;
    const u32 tid = threadIdx.x;
    for (u32 c = tid; c < B.n_chunks; c += 1024u) {
        const GzChunk r = B.recs[c];
        const u32 kind = r.state & 255u;
        u32 v = GZC_BROKEN;
        if (kind == GZ_NEXT) {
            const u64 j = r.end_bit / B.chunk_bits;
            if (j > c && j < B.n_chunks && B.recs[j].start_bit == r.end_bit) v = (u32)j;
        } else if (kind == GZ_MEMBER_END) {
            v = GZC_END_MEMBER;
        } else if (kind == GZ_OUT_OF_INPUT) {
            v = GZC_END_INPUT;
        } else if (kind == GZ_FAILED) {
            v = GZC_FAILED;
        }
        nxt[c] = v;
    }
    __syncthreads();
    if (tid == 0) {
        u32 c = 0, n = 0, status = 0, kind = GZ_IDLE;
        for (;;) {
            B.live[4u * n] = c;
            n++;
            const u32 v = nxt[c];
            if (v < B.n_chunks) {
                c = v;
                continue;
            }
            if (v == GZC_END_MEMBER) kind = GZ_MEMBER_END;
            else if (v == GZC_END_INPUT) kind = GZ_OUT_OF_INPUT;
            else status = (c << 8) | (v == GZC_FAILED ? ((B.recs[c].state >> 8) & 255u) : (u32)BZ_BAD_BLOCK);
            break;
        }
        s_n = n;
        s_status = status;
        s_kind = kind;
        s_last = c;
    }
    __syncthreads();
    const u32 n = s_n;
    // text offsets: a prefix sum over the chain, eight chunks a thread
    u32 len[8], sum = 0;
#pragma unroll
    for (u32 q = 0; q < 8u; ++q) {
        const u32 i = tid * 8u + q;
        len[q] = i < n ? B.recs[B.live[4u * i]].out_len : 0u;
        sum += len[q];
    }
    part[tid] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {
        const u32 v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const u64 total = part[1023];
    u32 status = s_status;
    if (!status && total > B.text_cap) status = (s_last << 8) | (u32)BZ_BAD_SIZE;
    if (!status) {
        u32 off = part[tid] - sum;
#pragma unroll
        for (u32 q = 0; q < 8u; ++q) {
            const u32 i = tid * 8u + q;
            if (i < n) {
                B.live[4u * i + 1u] = len[q];
                B.live[4u * i + 2u] = off;
                const u64 valid = (u64)B.valid + off;
                B.live[4u * i + 3u] = valid > GZ_WINDOW ? GZ_WINDOW : (u32)valid;
                for (u32 t = (off + 4095u) / 4096u; t < (u32)(((u64)off + len[q] + 4095u) / 4096u); ++t) B.tile_map[t] = i;
            }
            off += len[q];
        }
    }
    if (tid == 0) {
        u32 *S = B.summary;
        const u64 end_bit = B.recs[s_last].end_bit;
        const u32 kind = s_kind;
        S[GZS_STATUS] = status;
        S[GZS_CRC] = 0; // (k_gz_crc_join XORs the slices' shares into it)
        S[GZS_N_LIVE] = n;
        S[GZS_TOTAL] = (u32)total;
        S[GZS_END_STATE] = kind;
        S[GZS_END_BIT_LO] = (u32)end_bit;
        S[GZS_END_BIT_HI] = (u32)(end_bit >> 32);
        const u64 valid = (u64)B.valid + total;
        S[GZS_VALID] = valid > GZ_WINDOW ? GZ_WINDOW : (u32)valid;
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

// The window in front of every chunk on the chain = the last GZ_WINDOW symbols of [marker slots | symbols] of the one before
// it, markers looked up in THAT one's window: a chain of 32768-entry maps.  Composed in three passes so that GZ_GROUPS
// stretches of the chain run side by side: (1) every group's chunks composed into one map from "window in front of the
// group" to "window behind it" (16-bit entries: a byte, or a marker into the group's input), (2) one workgroup walks the
// groups, (3) every group again, from its now known input window, leaving each chunk's window behind.
__device__ __forceinline__ void gz_group_range(const GzBatch &B, u32 g, u32 &lo, u32 &hi) {
    const u32 n = B.summary[GZS_N_LIVE];
    const u32 per = (n + GZ_GROUPS - 1u) / GZ_GROUPS;
    lo = g * per < n ? g * per : n;
    hi = lo + per < n ? lo + per : n;
}
// 32 consecutive tail symbols of a chunk for this thread
struct GzTail { uint4 raw[4]; };
__device__ __forceinline__ GzTail gz_load_tail(const GzBatch &B, u32 li, u32 tid) {
    const u32 ci = B.live[4u * li], len = B.live[4u * li + 1u];
    const uint16_t *tail = B.sym + (u64)ci * B.cap + len;
    GzTail t;
#pragma unroll
    for (u32 q = 0; q < 4u; ++q) __builtin_memcpy(&t.raw[q], tail + tid * 32u + q * 8u, 16);
    return t;
}

__global__ __launch_bounds__(1024) void k_gz_win_compose(GzBatch B) {
    __shared__ uint16_t M[GZ_WINDOW];
    if (B.summary[GZS_STATUS]) return;
    const u32 tid = threadIdx.x, g = blockIdx.x;
    u32 lo, hi;
    gz_group_range(B, g, lo, hi);
#pragma unroll
    for (u32 q = 0; q < 32u; ++q) M[tid * 32u + q] = (uint16_t)(0x8000u | (tid * 32u + q));
    lds_barrier();
    GzTail nx{};
    if (lo < hi) nx = gz_load_tail(B, lo, tid);
    for (u32 li = lo; li < hi; ++li) {
        const GzTail cur = nx;
        if (li + 1u < hi) nx = gz_load_tail(B, li + 1u, tid);
        const uint16_t *e = (const uint16_t *)cur.raw;
        u32 packed[16];
#pragma unroll
        for (u32 q = 0; q < 16u; ++q) {
            const u32 a = e[2u * q], b = e[2u * q + 1u];
            const u32 ra = a < 256u ? a : (u32)M[a & 0x7FFFu], rb = b < 256u ? b : (u32)M[b & 0x7FFFu];
            packed[q] = ra | (rb << 16);
        }
        lds_barrier();
#pragma unroll
        for (u32 q = 0; q < 16u; ++q) ((u32 *)M)[tid * 16u + q] = packed[q];
        lds_barrier();
    }
    uint4 *dst = (uint4 *)(B.group_map + (u64)g * GZ_WINDOW);
    const uint4 *src = (const uint4 *)M;
#pragma unroll
    for (u32 q = 0; q < 4u; ++q) dst[tid * 4u + q] = src[tid * 4u + q];
}

__global__ __launch_bounds__(1024) void k_gz_win_groups(GzBatch B) {
    __shared__ uint8_t W[GZ_WINDOW];
    if (B.summary[GZS_STATUS]) return;
    const u32 tid = threadIdx.x;
    ((uint4 *)W)[tid] = ((const uint4 *)B.window)[tid];
    ((uint4 *)W)[tid + 1024u] = ((const uint4 *)B.window)[tid + 1024u];
    lds_barrier();
    for (u32 g = 0; g < GZ_GROUPS; ++g) {
        uint4 *dst = (uint4 *)(B.group_win + (u64)g * GZ_WINDOW);
        dst[tid] = ((const uint4 *)W)[tid];
        dst[tid + 1024u] = ((const uint4 *)W)[tid + 1024u];
        uint4 raw[4];
        const uint16_t *m = B.group_map + (u64)g * GZ_WINDOW + tid * 32u;
#pragma unroll
        for (u32 q = 0; q < 4u; ++q) raw[q] = ((const uint4 *)m)[q];
        const uint16_t *e = (const uint16_t *)raw;
        u32 packed[8];
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
        lds_barrier();
#pragma unroll
        for (u32 q = 0; q < 8u; ++q) ((u32 *)W)[tid * 8u + q] = packed[q];
        lds_barrier();
    }
    ((uint4 *)B.window)[tid] = ((const uint4 *)W)[tid];
    ((uint4 *)B.window)[tid + 1024u] = ((const uint4 *)W)[tid + 1024u];
}

__global__ __launch_bounds__(1024) void k_gz_win_chunks(GzBatch B) {
    __shared__ uint8_t W[GZ_WINDOW];
    if (B.summary[GZS_STATUS]) return;
    const u32 tid = threadIdx.x, g = blockIdx.x;
    u32 lo, hi;
    gz_group_range(B, g, lo, hi);
    if (lo >= hi) return;
    {
        const uint4 *src = (const uint4 *)(B.group_win + (u64)g * GZ_WINDOW);
        ((uint4 *)W)[tid] = src[tid];
        ((uint4 *)W)[tid + 1024u] = src[tid + 1024u];
    }
    lds_barrier();
    GzTail nx = gz_load_tail(B, lo, tid);
    for (u32 li = lo; li < hi; ++li) {
        const GzTail cur = nx;
        if (li + 1u < hi) nx = gz_load_tail(B, li + 1u, tid);
        { // the window in front of this chunk
            uint4 *dst = (uint4 *)(B.win_in + (u64)B.live[4u * li] * GZ_WINDOW);
            dst[tid] = ((const uint4 *)W)[tid];
            dst[tid + 1024u] = ((const uint4 *)W)[tid + 1024u];
        }
        if (li + 1u == hi) break; // (the window behind the group's last chunk is the next group's: pass 2 has it)
        const uint16_t *e = (const uint16_t *)cur.raw;
        u32 packed[8];
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
        lds_barrier();
#pragma unroll
        for (u32 q = 0; q < 8u; ++q) ((u32 *)W)[tid * 8u + q] = packed[q];
        lds_barrier();
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
    const CrcLaneOps ops = crc_lane_ops(x2n, lane);
    const u32 crc = wave_crc32(B.text + B.left + (u64)si * 65536u, size, T, x2n, ops, lane);
    if (lane == 0) B.crc_tmp[si] = crc;
}
// crc(whole) = the XOR over the slices of crc(slice) shifted past everything behind the slice: all shifts independent.  A thread
// takes a run of consecutive slices, works out the shift of its last one from the byte count (x^(8 n) mod P by the bits of n),
// and gets from slice to slice before it with one multiplication by x^(8 * 65536).
// (sixteen wavefronts on sixteen CUs: on one they take turns, and the byte count's thirty multiplications are most of the work)
__global__ __launch_bounds__(64) void k_gz_crc_join(GzBatch B, X2N x2n) {
    if (B.summary[GZS_STATUS]) return;
    const u32 total = B.summary[GZS_TOTAL];
    const u32 tid = blockIdx.x * 64u + threadIdx.x;
    const u32 n = (u32)(((u64)total + 65535u) >> 16);
    const u32 per = (n + 1023u) / 1024u;
    const u32 lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    u32 acc = 0;
    if (lo < hi) {
        const u64 e = (u64)hi * 65536u;
        const u32 end_last = e < total ? (u32)e : total; // where the run's last slice ends
        u32 op = crc_x2n(total - end_last, 3);
        const u32 step = crc_x2n(65536u, 3);
        for (u32 i = hi; i-- > lo;) {
            acc ^= crc_multmodp(op, B.crc_tmp[i]);
            // the slice in front of slice i has slice i's bytes behind it as well
            const u32 len_i = total - i * 65536u < 65536u ? total - i * 65536u : 65536u;
            op = crc_multmodp(op, len_i == 65536u ? step : crc_x2n(len_i, 3));
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc ^= (u32)__shfl_xor((int)acc, off);
    if (threadIdx.x == 0 && acc) atomicXor(&B.summary[GZS_CRC], acc); // (k_gz_chain left it at zero)
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

hipError_t launch_gzip_chunks(const GzBatch &b, const GzFeed *feed, uint64_t avail_bytes, uint32_t c0, uint32_t n, bool final, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gz_chunks, dim3(n), dim3(64), 0, st, b.comp, feed, avail_bytes, b.chunk_bits, c0, b.first_bit, final ? 1u : 0u, b.recs,
                       b.sym, b.cap, b.claims, b.n_regions, b.times);
    return hipGetLastError();
}

hipError_t launch_gzip_batch(const GzBatch &b, hipStream_t st, hipStream_t st_crc, hipEvent_t text_done) {
    if (b.n_chunks == 0 || b.n_chunks > GZ_MAX_CHUNKS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_gz_chain, dim3(1), dim3(1024), 0, st, b);
    hipLaunchKernelGGL(k_gz_win_compose, dim3(GZ_GROUPS), dim3(1024), 0, st, b);
    hipLaunchKernelGGL(k_gz_win_groups, dim3(1), dim3(1024), 0, st, b);
    hipLaunchKernelGGL(k_gz_win_chunks, dim3(GZ_GROUPS), dim3(1024), 0, st, b);
    const u32 n_tiles = (u32)((b.text_cap + 4095u) / 4096u);
    hipLaunchKernelGGL(k_gz_text, dim3(n_tiles), dim3(256), 0, st, b);
    if (hipError_t e = hipEventRecord(text_done, st); e != hipSuccess) return e;
    if (hipError_t e = hipStreamWaitEvent(st_crc, text_done, 0); e != hipSuccess) return e;
    const u32 n_slices = (u32)((b.text_cap + 65535u) / 65536u);
    hipLaunchKernelGGL(k_gz_crc_slices, dim3((n_slices + 3u) / 4u), dim3(256), 0, st_crc, b, x2n_table());
    hipLaunchKernelGGL(k_gz_crc_join, dim3(16), dim3(64), 0, st_crc, b, x2n_table());
    hipLaunchKernelGGL(k_gz_cut, dim3(1), dim3(64), 0, st, b);
    return hipGetLastError();
}

hipError_t launch_bgzf_inflate(const uint8_t *comp, const BgzfMember *members, uint32_t n_members, uint8_t *text,
                               uint32_t *status, hipStream_t st) {
    if (n_members == 0) return hipSuccess;
    const X2N &x2n = x2n_table();
    static const bool serial = cfg("bgzf_serial") != nullptr; // A/B: one symbol at a time
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
