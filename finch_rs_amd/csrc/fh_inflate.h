// fh_inflate.h -- DEFLATE (RFC 1951) / gzip (RFC 1952) decoding for the FASTX byte sources of fh_host.cpp.
//
// needletail hands finch a decompressed stream whatever the file holds (lib.rs:60, magic-byte sniffing); most real FASTQ is
// gzip, and with the sketch kernel at hundreds of Gbases/s the inflate speed IS the end-to-end speed of such a file.
// zlib's inflate (1.2.11 in the image) decodes a byte or a match per table lookup through a 32-bit bit buffer and runs at
// 0.3 Gbases/s of FASTQ on the GPU box's host.  This decoder is built for throughput instead:
//   * a 64-bit bit buffer topped up with one unaligned 8-byte load per symbol (no per-byte loop);
//   * an 11-bit first-level litlen table whose entries carry the symbol's value AND the bits it consumes, extra bits
//     included, so a literal is one lookup + one store and a match is two lookups; up to three literals are decoded
//     per refill;
//   * matches are copied 8 bytes at a time (16 for far ones), runs of one byte by a broadcast word;
//   * output goes through a private window so that the hot loop never checks a caller's buffer end.
// gzip members are checked as zlib's wrapper checks them: CRC-32 (carry-less-multiply folding where the CPU has it,
// verified against zlib's crc32 when the library loads, zlib's otherwise) and ISIZE; truncated or corrupt input is an
// error, never silently short output.
#pragma once
#include <zlib.h> // crc32(): the fallback and the self-test reference

#include <cstdint>
#include <cstring>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

// the hot loop is compiled twice (BMI2 gives three-operand variable shifts and bit-field extracts); the dynamic linker picks
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define FH_INFLATE_CLONES __attribute__((target_clones("bmi2", "default"), noinline))
#else
#define FH_INFLATE_CLONES
#endif

namespace finch {
namespace inf {

// ---------------------------------------------------------------------------------------------------------------------
// CRC-32 (IEEE 802.3, reflected) by carry-less multiplication.  The method is Intel's ("Fast CRC Computation for Generic
// Polynomials Using PCLMULQDQ Instruction", Gopal et al., 2009): a 128-bit accumulator A stands for A(x) mod P, and moving it
// d bits up the message is  clmul(A.lo, x^(d+64) mod P) ^ clmul(A.hi, x^d mod P)  -- two multiplies and an xor with the data
// d bits further on.  Four accumulators 512 bits apart run over the bulk, are folded into one (128 bits apart), and that one
// walks the remaining 16-byte blocks.  The constants are the paper's (bit-reflected x^n mod P for n = 512+64, 512, 128+64,
// 128).  What is left at the end is 16 bytes R congruent to the whole message; instead of the paper's 128 -> 64 -> 32 bit
// Barrett steps they simply go through the ordinary byte-wise CRC once (zlib's, register 0 in, no inversions): that
// computes R(x) * x^32 mod P, which is the definition of the remainder wanted.
// ---------------------------------------------------------------------------------------------------------------------
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static inline __m128i crc_fold(__m128i acc, __m128i k, __m128i data) {
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k, 0x00), _mm_clmulepi64_si128(acc, k, 0x11)), data);
}
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_clmul(const uint8_t *buf, size_t len, uint32_t crc /* running, inverted */) {
    // len >= 64 and a multiple of 16
    const __m128i k512 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll); // {x^(512+64), x^512} mod P, reflected
    const __m128i k128 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll); // {x^(128+64), x^128} mod P, reflected
    const __m128i *p = (const __m128i *)buf;
    __m128i acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = _mm_loadu_si128(p + i);
    acc[0] = _mm_xor_si128(acc[0], _mm_cvtsi32_si128((int)crc)); // the running register joins the first four message bytes
    p += 4;
    size_t blocks = len / 16 - 4; // 16-byte blocks still to take
    for (; blocks >= 4; blocks -= 4, p += 4)
        for (int i = 0; i < 4; ++i) acc[i] = crc_fold(acc[i], k512, _mm_loadu_si128(p + i));
    __m128i a = acc[0];
    for (int i = 1; i < 4; ++i) a = crc_fold(a, k128, acc[i]);
    for (; blocks; --blocks, ++p) a = crc_fold(a, k128, _mm_loadu_si128(p));
    alignas(16) uint8_t rest[16];
    _mm_store_si128((__m128i *)rest, a);
    return ~(uint32_t)::crc32(0xFFFFFFFFu, rest, 16); // (zlib inverts on the way in and out: register 0 in, raw register out)
}
#endif

// true iff the folding routine may be used: the CPU has the instructions and the routine reproduces zlib's crc32 on a
// probe that exercises the 64-byte loop, the 16-byte loop and a non-zero starting value
static inline bool crc32_clmul_ok() {
#if defined(__x86_64__)
    static const bool ok = [] {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        uint8_t probe[64 * 5 + 48];
        uint32_t x = 0x9E3779B9u;
        for (size_t i = 0; i < sizeof probe; ++i) {
            x = x * 1664525u + 1013904223u;
            probe[i] = (uint8_t)(x >> 24);
        }
        for (uint32_t start : {0u, 0xDEADBEEFu}) {
            const uint32_t want = (uint32_t)::crc32(start, probe, (uInt)sizeof probe);
            if (~crc32_clmul(probe, sizeof probe, ~start) != want) return false;
        }
        return true;
    }();
    return ok;
#else
    return false;
#endif
}

// zlib's crc32(crc, buf, len) contract
static inline uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len) {
#if defined(__x86_64__)
    if (len >= 256 && crc32_clmul_ok()) {
        const size_t body = len & ~(size_t)15;
        crc = ~crc32_clmul(buf, body, ~crc);
        buf += body;
        len -= body;
    }
#endif
    while (len) {
        const size_t n = len < (1u << 30) ? len : (1u << 30);
        crc = (uint32_t)::crc32(crc, buf, (uInt)n);
        buf += n;
        len -= n;
    }
    return crc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Huffman decode tables
// ---------------------------------------------------------------------------------------------------------------------
// Entry (32 bits):  [31:16] value (literal byte / length or distance base / subtable start)   [15:12] kind
//                   [11:8] number of extra bits (lengths, distances) or subtable index bits   [7:0] bits the entry consumes
// For lengths and distances the consumed bits INCLUDE the extra bits, which sit right above the code in the bit buffer.
// Two literals whose codes fit the first-level index together share an entry: [23:16] the first, [31:24] the second,
// [11:8] the first one's code length (0 = the entry holds a single literal), [7:0] both lengths -- nucleotide text is
// 2-3 bits per symbol under a Huffman code, so most lookups of a FASTQ stream return two bytes.
constexpr uint32_t K_LITERAL = 1u << 12, K_EOB = 2u << 12, K_SUB = 4u << 12, K_LEN = 8u << 12; // kind 0 = invalid code
constexpr int LIT_BITS = 11, DIST_BITS = 8;
constexpr int LIT_TABLE_MAX = (1 << LIT_BITS) + 1024, DIST_TABLE_MAX = (1 << DIST_BITS) + 512; // with every possible subtable

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t bitrev(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// what a symbol decodes to, without its code length
static inline uint32_t litlen_entry(int sym) {
    if (sym < 256) return ((uint32_t)sym << 16) | K_LITERAL;
    if (sym == 256) return K_EOB;
    if (sym > 285) return 0; // 286, 287: not valid in a stream
    return ((uint32_t)LEN_BASE[sym - 257] << 16) | K_LEN | ((uint32_t)LEN_EXTRA[sym - 257] << 8) | LEN_EXTRA[sym - 257];
}
static inline uint32_t dist_entry(int sym) {
    if (sym > 29) return 0;
    return ((uint32_t)DIST_BASE[sym] << 16) | K_LEN | ((uint32_t)DIST_EXTRA[sym] << 8) | DIST_EXTRA[sym];
}

// Canonical code from lengths[0, n) (RFC 1951 3.2.2) into a two-level table.  Returns false for an over-subscribed set of
// lengths; an incomplete set leaves invalid entries behind (decoding one is an error), which covers the single-code
// distance trees encoders do emit.
template <class EntryFn>
static inline bool build_table(const uint8_t *lengths, int n, int primary_bits, uint32_t *table, int table_cap, EntryFn entry_of) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lengths[i]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false; // over-subscribed
    }
    uint32_t next[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    const int P = 1 << primary_bits;
    for (int i = 0; i < P; ++i) table[i] = 0;
    // longest code behind each primary index decides its subtable's size
    uint8_t sub_len[1 << LIT_BITS];
    memset(sub_len, 0, (size_t)P);
    uint32_t codes[288];
    for (int s = 0; s < n; ++s) {
        const int l = lengths[s];
        if (!l) continue;
        codes[s] = bitrev(next[l]++, l);
        if (l > primary_bits) {
            uint8_t &m = sub_len[codes[s] & (uint32_t)(P - 1)];
            if (l - primary_bits > m) m = (uint8_t)(l - primary_bits);
        }
    }
    int top = P;
    for (int i = 0; i < P; ++i) {
        if (!sub_len[i]) continue;
        const int sz = 1 << sub_len[i];
        if (top + sz > table_cap) return false;
        table[i] = ((uint32_t)top << 16) | K_SUB | ((uint32_t)sub_len[i] << 8) | (uint32_t)primary_bits;
        for (int j = 0; j < sz; ++j) table[top + j] = 0;
        top += sz;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lengths[s];
        if (!l) continue;
        const uint32_t e = entry_of(s);
        if (l <= primary_bits) {
            const uint32_t v = e ? e + (uint32_t)l : 0u; // (consumed bits = code length + the extra bits already in e)
            for (uint32_t i = codes[s]; i < (uint32_t)P; i += 1u << l) table[i] = v;
        } else {
            const uint32_t pe = table[codes[s] & (uint32_t)(P - 1)];
            const int sb = (int)((pe >> 8) & 15u), base = (int)(pe >> 16);
            const uint32_t v = e ? e + (uint32_t)(l - primary_bits) : 0u;
            for (uint32_t i = codes[s] >> primary_bits; i < (1u << sb); i += 1u << (l - primary_bits)) table[base + (int)i] = v;
        }
    }
    return true;
}

// second pass over the first level of a literal/length table: pair up literals (see the entry format)
static inline void pair_literals(uint32_t *table) {
    const int P = 1 << LIT_BITS;
    static thread_local uint32_t single[1 << LIT_BITS];
    memcpy(single, table, sizeof single);
    for (int i = 0; i < P; ++i) {
        const uint32_t e = single[i];
        if ((e & (K_LITERAL | K_SUB)) != K_LITERAL) continue;
        const uint32_t l1 = e & 0xFFu;
        if (l1 >= (uint32_t)LIT_BITS) continue;
        const uint32_t f = single[(uint32_t)i >> l1]; // what the bits after the first code select, zero-extended ...
        if ((f & (K_LITERAL | K_SUB)) != K_LITERAL) continue;
        const uint32_t l2 = f & 0xFFu;
        if (l1 + l2 > (uint32_t)LIT_BITS) continue;   // ... which is only meaningful if the second code lies inside the index
        table[i] = (f & 0x00FF0000u) << 8 | (e & 0x00FF0000u) | K_LITERAL | (l1 << 8) | (l1 + l2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the decoder
// ---------------------------------------------------------------------------------------------------------------------
enum Status {
    OK = 0, NEED_INPUT, NEED_OUTPUT, STREAM_END, BAD,
    MORE_SLICES, // callers that run() a buffer piecewise: not a run() result
    BLOCK_END,   // run() with stop_at_block_end: a block has just ended (the last one too: DONE is reported by the next call)
};
constexpr ptrdiff_t OUT_MARGIN = 258 + 16 + 80; // longest match + copy overshoot + the literals of one refill (<= 2 per 2 bits)

struct Decoder {
    // bit reader
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    // block state
    enum { HEADER, STORED, CODES, DONE } state = HEADER;
    bool final_block = false;
    uint32_t stored_left = 0;
    uint32_t lit[LIT_TABLE_MAX], dist[DIST_TABLE_MAX];
    // History that is not in front of the output buffer: the ext_len bytes that END at ext_end precede win_start in the
    // stream (a caller that decodes straight into successive destination buffers keeps the last 32 KiB of what it has
    // delivered here).  A match may reach into it.
    const uint8_t *ext_end = nullptr;
    size_t ext_len = 0;
    bool stop_at_block_end = false; // (fh_pargz.h: a chunk ends where the next one was found to begin)

    // start in the middle of a stream: the next bit to decode is bit (bit & 7) of *in; `in` is left behind that byte
    void start_at_bit(const uint8_t *&in, unsigned bit) {
        reset();
        bitbuf = (uint64_t)(*in++ >> (bit & 7u));
        bitcnt = 8 - (int)(bit & 7u);
    }
    // bits of the stream consumed so far, given where `in` stands relative to the start of the stream's buffer
    uint64_t bit_position(const uint8_t *in, const uint8_t *base) const { return (uint64_t)(in - base) * 8u - (uint64_t)bitcnt; }
    // the header of the next block alone (tables built, state CODES / STORED / DONE)
    Status block_header(const uint8_t *&in, const uint8_t *in_end) { return header(in, in_end); }

    void reset() {
        ext_end = nullptr;
        ext_len = 0;
        bitbuf = 0;
        bitcnt = 0;
        state = HEADER;
        final_block = false;
        stored_left = 0;
    }

    static inline uint64_t load64(const uint8_t *p) {
        uint64_t v;
        memcpy(&v, p, 8);
        return v;
    }

    // Decode into [out, out_end) from [in, in_end).  `win_start` bounds how far back a match may reach.  The caller
    // guarantees 8 readable bytes beyond in_end (padding): the bit reader loads whole words.  On NEED_INPUT / NEED_OUTPUT
    // nothing of the symbol at hand has been consumed.
    Status run(const uint8_t *&in, const uint8_t *in_end, uint8_t *&out, uint8_t *out_end, const uint8_t *win_start) {
        for (;;) {
            if (state == DONE) return STREAM_END;
            if (state == HEADER) {
                const Status s = header(in, in_end);
                if (s != OK) return s;
                if (stop_at_block_end && state != CODES && state != STORED) return BLOCK_END; // (an empty stored block)
                continue;
            }
            if (state == STORED) {
                // (byte aligned: bitcnt is a multiple of 8 and those bytes are given back to the input first)
                while (stored_left) {
                    if (bitcnt) {
                        if (out == out_end) return NEED_OUTPUT;
                        *out++ = (uint8_t)bitbuf;
                        bitbuf >>= 8;
                        bitcnt -= 8;
                        --stored_left;
                        continue;
                    }
                    size_t n = stored_left;
                    const size_t in_left = in < in_end ? (size_t)(in_end - in) : 0;
                    if (in_left < n) n = in_left;
                    if ((size_t)(out_end - out) < n) n = (size_t)(out_end - out);
                    if (n == 0) return in_left == 0 ? NEED_INPUT : NEED_OUTPUT;
                    memcpy(out, in, n);
                    in += n;
                    out += n;
                    stored_left -= (uint32_t)n;
                }
                state = final_block ? DONE : HEADER;
                if (stop_at_block_end) return BLOCK_END;
                continue;
            }
            const Status s = codes(in, in_end, out, out_end, win_start);
            if (s != OK) return s;
            if (stop_at_block_end && state != CODES) return BLOCK_END;
        }
    }

  private:
    // make at least `need` (<= 56) bits available if the input has them
    inline bool fill(const uint8_t *&in, const uint8_t *in_end, int need) {
        while (bitcnt < need) {
            if (in >= in_end) return false; // (>=: the symbol loops of fh_pargz.h read ahead into the padding behind the input)
            bitbuf |= (uint64_t)*in++ << bitcnt;
            bitcnt += 8;
        }
        return true;
    }

    Status header(const uint8_t *&in, const uint8_t *in_end) {
        // The header of a dynamic block is at most 3 + 14 + 19*3 + 320*(15+7) bits; decode it from a snapshot and commit
        // only when all of it was there.
        const uint64_t sv_buf = bitbuf;
        const int sv_cnt = bitcnt;
        const uint8_t *sv_in = in;
        auto need_input = [&]() {
            bitbuf = sv_buf;
            bitcnt = sv_cnt;
            in = sv_in;
            return NEED_INPUT;
        };
        if (!fill(in, in_end, 3)) return need_input();
        final_block = bitbuf & 1u;
        const unsigned type = (unsigned)(bitbuf >> 1) & 3u;
        bitbuf >>= 3;
        bitcnt -= 3;
        if (type == 3) return BAD;
        if (type == 0) {
            const int drop = bitcnt & 7;
            bitbuf >>= drop;
            bitcnt -= drop;
            if (!fill(in, in_end, 32)) return need_input();
            const uint32_t len = (uint32_t)bitbuf & 0xFFFFu, nlen = (uint32_t)(bitbuf >> 16) & 0xFFFFu;
            bitbuf >>= 32;
            bitcnt -= 32;
            if ((len ^ nlen) != 0xFFFFu) return BAD;
            stored_left = len;
            state = STORED;
            if (len == 0) state = final_block ? DONE : HEADER;
            return OK;
        }
        uint8_t lens[320];
        int hlit = 288, hdist = 32;
        if (type == 1) {
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else {
            if (!fill(in, in_end, 14)) return need_input();
            hlit = (int)(bitbuf & 31u) + 257;
            hdist = (int)((bitbuf >> 5) & 31u) + 1;
            const int hclen = (int)((bitbuf >> 10) & 15u) + 4;
            bitbuf >>= 14;
            bitcnt -= 14;
            if (hlit > 286 || hdist > 30) return BAD;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; ++i) {
                if (!fill(in, in_end, 3)) return need_input();
                cl[order[i]] = (uint8_t)(bitbuf & 7u);
                bitbuf >>= 3;
                bitcnt -= 3;
            }
            uint32_t cltab[128];
            if (!build_table(cl, 19, 7, cltab, 128, [](int s) { return ((uint32_t)s << 16) | K_LITERAL; })) return BAD;
            int i = 0;
            while (i < hlit + hdist) {
                // (a code is <= 7 bits, its extra bits <= 7.  With fewer than 14 bits left -- the last block of a member may end
                // within a byte or two of its header -- the entry is looked up on what there is, the missing bits reading as 0:
                // whether that was enough is decided by the entry's own length below, not before looking)
                const bool full = fill(in, in_end, 14);
                const uint32_t e = cltab[bitbuf & 127u];
                if (!(e & K_LITERAL)) {
                    if (!full && bitcnt < 7) return need_input();
                    return BAD;
                }
                const int l = (int)(e & 0xFFu), sym = (int)(e >> 16);
                const int extra = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
                if (bitcnt < l + extra) return need_input();
                bitbuf >>= l;
                bitcnt -= l;
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                    continue;
                }
                int rep;
                uint8_t val = 0;
                if (sym == 16) {
                    if (i == 0) return BAD;
                    val = lens[i - 1];
                    rep = 3 + (int)(bitbuf & 3u);
                } else if (sym == 17) {
                    rep = 3 + (int)(bitbuf & 7u);
                } else {
                    rep = 11 + (int)(bitbuf & 127u);
                }
                bitbuf >>= extra;
                bitcnt -= extra;
                if (i + rep > hlit + hdist) return BAD;
                while (rep--) lens[i++] = val;
            }
            if (lens[256] == 0) return BAD; // no end-of-block code
            // the distance lengths follow the literal/length ones directly
            memmove(lens + 288, lens + hlit, (size_t)hdist);
            memset(lens + hlit, 0, (size_t)(288 - hlit));
        }
        if (!build_table(lens, hlit, LIT_BITS, lit, LIT_TABLE_MAX, litlen_entry)) return BAD;
        pair_literals(lit);
        if (!build_table(lens + 288, hdist, DIST_BITS, dist, DIST_TABLE_MAX, dist_entry)) return BAD;
        state = CODES;
        return OK;
    }

    FH_INFLATE_CLONES Status codes(const uint8_t *&in_ref, const uint8_t *in_end, uint8_t *&out_ref, uint8_t *out_end, const uint8_t *win_start) {
        const uint8_t *in = in_ref;
        uint8_t *out = out_ref;
        uint64_t bb = bitbuf;
        int bc = bitcnt;
        Status result = OK;
        const uint32_t *const LT = lit, *const DT = dist;
        // ---- fast loop: >= 8 input bytes for every refill; room for the literals one refill can hold (<= 36), the longest
        //      match and the copy overshoot ----
        while (in_end - in >= 16 && out_end - out >= OUT_MARGIN) {
            // top up to >= 56 bits: one unaligned load; the bytes already in the buffer are re-read, not skipped
            bb |= load64(in) << bc;
            in += (63 - bc) >> 3;
            bc |= 56;
            uint32_t e = LT[bb & ((1u << LIT_BITS) - 1u)];
            // literals: as many as the buffered bits allow (nucleotide text under a Huffman code is 2-3 bits per symbol);
            // the loop is left with >= 20 bits, enough for any length code and its extra bits
            for (;;) {
                if (__builtin_expect((e & K_SUB) != 0, 0)) {
                    bb >>= LIT_BITS;
                    bc -= LIT_BITS;
                    e = LT[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
                }
                if (!(e & K_LITERAL)) break;
                out[0] = (uint8_t)(e >> 16);
                out[1] = (uint8_t)(e >> 24); // (the second literal of a paired entry; otherwise overwritten by what follows)
                out += 1 + (((e >> 8) & 15u) != 0u);
                bb >>= (e & 0xFFu);
                bc -= (int)(e & 0xFFu);
                if (bc < 20) goto next_symbol; // any length code with its extra bits still fits? else top up first
                e = LT[bb & ((1u << LIT_BITS) - 1u)];
            }
            if (__builtin_expect(!(e & K_LEN), 0)) {
                if (e & K_EOB) {
                    bb >>= (e & 0xFFu);
                    bc -= (int)(e & 0xFFu);
                    state = final_block ? DONE : HEADER;
                    goto done;
                }
                result = BAD;
                goto done;
            }
            {
                const uint32_t total = e & 0xFFu, nx = (e >> 8) & 15u;
                const uint32_t length = (e >> 16) + (uint32_t)((bb >> (total - nx)) & ((1u << nx) - 1u));
                bb >>= total;
                bc -= (int)total;
                if (bc < 28) { // a distance is <= 15 + 13 bits
                    bb |= load64(in) << bc;
                    in += (63 - bc) >> 3;
                    bc |= 56;
                }
                uint32_t d = DT[bb & ((1u << DIST_BITS) - 1u)];
                if (__builtin_expect((d & K_SUB) != 0, 0)) {
                    bb >>= DIST_BITS;
                    bc -= DIST_BITS;
                    d = DT[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))];
                }
                if (__builtin_expect(!(d & K_LEN), 0)) {
                    result = BAD;
                    goto done;
                }
                const uint32_t dtotal = d & 0xFFu, dnx = (d >> 8) & 15u;
                const uint32_t distance = (d >> 16) + (uint32_t)((bb >> (dtotal - dnx)) & ((1u << dnx) - 1u));
                bb >>= dtotal;
                bc -= (int)dtotal;
                if (__builtin_expect((size_t)(out - win_start) < distance, 0)) {
                    // reaches before this buffer: into the external history, or before the start of the stream
                    const size_t back = distance - (size_t)(out - win_start);
                    if (back > ext_len) {
                        result = BAD;
                        goto done;
                    }
                    const uint8_t *sp = ext_end - back;
                    uint32_t i = 0;
                    for (; i < length && i < back; ++i) out[i] = sp[i];
                    for (; i < length; ++i) out[i] = win_start[i - back];
                    out += length;
                    goto next_symbol;
                }
                const uint8_t *src = out - distance;
                uint8_t *const end = out + length;
                if (distance >= 16) {
                    do {
                        memcpy(out, src, 16);
                        out += 16;
                        src += 16;
                    } while (out < end);
                } else if (distance >= 8) {
                    do {
                        memcpy(out, src, 8);
                        out += 8;
                        src += 8;
                    } while (out < end);
                } else if (distance == 1) {
                    const uint64_t v = 0x0101010101010101ull * src[0];
                    do {
                        memcpy(out, &v, 8);
                        out += 8;
                    } while (out < end);
                } else {
                    do *out++ = *src++;
                    while (out < end);
                }
                out = end;
            }
        next_symbol:;
        }
        // (the word loads leave true-but-uncounted stream bits above `bc`; everything below works on a clean buffer)
        bb &= bc >= 64 ? ~0ull : ((1ull << bc) - 1ull);
        // ---- careful loop: the same decoding with every bound checked; a symbol is consumed only when all of it (its
        //      extra bits and its distance included) is in the input and fits the output ----
        for (;;) {
            const uint64_t sv_bb = bb;
            const int sv_bc = bc;
            const uint8_t *const sv_in = in;
            auto more = [&](int need) { // >= need bits, byte by byte
                while (bc < need) {
                    if (in >= in_end) return false;
                    bb |= (uint64_t)*in++ << bc;
                    bc += 8;
                }
                return true;
            };
            auto give_up = [&](Status s) {
                bb = sv_bb;
                bc = sv_bc;
                in = sv_in;
                result = s;
            };
            // the fast loop takes over again as soon as its margins are there
            if (in_end - in >= 16 && out_end - out >= OUT_MARGIN) {
                bitbuf = bb;
                bitcnt = bc;
                in_ref = in;
                out_ref = out;
                return OK;
            }
            more(15 + 5); // (whatever is there; the checks below tell whether it was enough)
            uint32_t e = LT[bb & ((1u << LIT_BITS) - 1u)];
            int used = 0;
            if (e & K_SUB) {
                used = LIT_BITS;
                e = LT[(e >> 16) + (uint32_t)((bb >> LIT_BITS) & ((1u << ((e >> 8) & 15u)) - 1u))];
            }
            if (e == 0 && bc < 15) { // an entry of an incomplete code may be "invalid" only because the bits are not there yet
                give_up(NEED_INPUT);
                break;
            }
            if (e == 0) {
                result = BAD;
                break;
            }
            used += (int)(e & 0xFFu);
            // a paired entry: the second literal only if its bits are all there and there is room for it
            int l1 = (e & K_LITERAL) ? (int)((e >> 8) & 15u) : 0;
            if (l1 && (used > bc || out_end - out < 2)) {
                used = l1;
                l1 = 0;
            }
            if (used > bc) {
                give_up(NEED_INPUT);
                break;
            }
            if (e & K_LITERAL) {
                if (out == out_end) {
                    give_up(NEED_OUTPUT);
                    break;
                }
                *out++ = (uint8_t)(e >> 16);
                if (l1) *out++ = (uint8_t)(e >> 24);
                bb >>= used;
                bc -= used;
                continue;
            }
            if (e & K_EOB) {
                bb >>= used;
                bc -= used;
                state = final_block ? DONE : HEADER;
                break;
            }
            const uint32_t nx = (e >> 8) & 15u;
            const uint32_t length = (e >> 16) + (uint32_t)((bb >> (used - (int)nx)) & ((1u << nx) - 1u));
            bb >>= used;
            bc -= used;
            more(15 + 13);
            uint32_t d = DT[bb & ((1u << DIST_BITS) - 1u)];
            int dused = 0;
            if (d & K_SUB) {
                dused = DIST_BITS;
                d = DT[(d >> 16) + (uint32_t)((bb >> DIST_BITS) & ((1u << ((d >> 8) & 15u)) - 1u))];
            }
            if (d == 0 && bc < 15) {
                give_up(NEED_INPUT);
                break;
            }
            if (!(d & K_LEN)) {
                result = BAD;
                break;
            }
            dused += (int)(d & 0xFFu);
            if (dused > bc) {
                give_up(NEED_INPUT);
                break;
            }
            const uint32_t dnx = (d >> 8) & 15u;
            const uint32_t distance = (d >> 16) + (uint32_t)((bb >> (dused - (int)dnx)) & ((1u << dnx) - 1u));
            const size_t have = (size_t)(out - win_start);
            if (have < distance && distance - have > ext_len) {
                result = BAD;
                break;
            }
            if ((size_t)(out_end - out) < length) {
                give_up(NEED_OUTPUT);
                break;
            }
            bb >>= dused;
            bc -= dused;
            if (have < distance) { // the match starts in the external history
                const size_t back = distance - have;
                const uint8_t *sp = ext_end - back;
                uint32_t i = 0;
                for (; i < length && i < back; ++i) out[i] = sp[i];
                for (; i < length; ++i) out[i] = win_start[i - back];
            } else {
                const uint8_t *src = out - distance;
                for (uint32_t i = 0; i < length; ++i) out[i] = src[i];
            }
            out += length;
        }
    done:
        bb &= bc >= 64 ? ~0ull : ((1ull << bc) - 1ull);
        bitbuf = bb;
        bitcnt = bc;
        in_ref = in;
        out_ref = out;
        return result;
    }
};

// One complete raw DEFLATE stream of known inflated size (a BGZF member): false unless it decodes to exactly out_len bytes
// and ends with the input.  `in` must have 8 readable bytes beyond in_len.
static inline bool inflate_exact(Decoder &dec, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    dec.reset();
    const uint8_t *ip = in, *const in_end = in + in_len;
    uint8_t *op = out, *const out_end = out + out_len;
    const Status s = dec.run(ip, in_end, op, out_end, out);
    if (s != STREAM_END || op != out_end) return false;
    // whole bytes the bit buffer still holds were not part of the stream
    return (size_t)(ip - in) - (size_t)(dec.bitcnt >> 3) == in_len;
}

} // namespace inf
} // namespace finch
