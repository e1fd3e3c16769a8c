// fh_pargz.h -- one gzip member decoded by several threads.
//
// Most sequencing reads are stored as plain gzip: one DEFLATE stream, no index, every block coded against the 32 KiB of
// text before it.  needletail (lib.rs:60) inflates it on one thread, and so does fh_inflate.h's sequential reader at
// ~1.1 GB/s of text -- two orders of magnitude below what the device does with the text afterwards.  This is the two-pass
// scheme published for pugz (Kerbiriou & Chikhi, 2019) and rapidgzip (Knespel & Brunst, 2023), restated for this reader:
//
//   1. The compressed bytes are cut into chunks.  In each chunk but the first, a thread looks for the start of a DEFLATE
//      block: it tries every bit offset as the header of a dynamic-Huffman block and keeps the first one whose three codes
//      are complete and whose first symbols decode to text (FASTA / FASTQ is all this reader serves).
//   2. Every thread decodes from its start.  What lies before it is unknown: a byte copied from there is a mark, 0x8000 + i
//      for "byte i of the 32 KiB window in front of this chunk", and copying a match copies marks along.  pugz and
//      rapidgzip decode to 16-bit symbols for that; on sequencing reads marks never die out (every header line is a copy of
//      a copy of ... a header in the unknown window), so here the text is decoded to bytes at once and the marks are kept on
//      the side, sparsely: one bit per 8-byte word says whether the word holds anything of the unknown window, and only such
//      words have their positions written as 16-bit symbols too.  A match whose source words are all clean -- one 64-bit
//      look at the bitmap -- is the ordinary byte copy; 5-50 % of the words are dirty, depending on the data.  A thread
//      stops at the block boundary where the next chunk was found to begin; if it runs past that offset instead, the
//      "start" was not one, the next chunk's work is dropped and the thread carries on to the one after.
//   3. In order: the window in front of chunk i is the last 32 KiB of the text up to it (a chunk's tail is put right
//      first), then every chunk looks its marks up in its window, side by side.  The text is handed out by several threads,
//      each of which checksums what it has just copied; the CRC-32s are joined with crc32_combine.
//
// The member's CRC-32 and ISIZE are checked at its end as always, so a chunk stitched wrongly cannot go unnoticed.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <memory>
#include <mutex>
#include <vector>

#include "fh_inflate.h"

namespace finch {
namespace pargz {

constexpr uint32_t WINDOW = 32768;
constexpr uint16_t MARK = 0x8000;

static inline bool text_byte(uint32_t c) { return c == '\n' || c == '\r' || c == '\t' || (c >= 32 && c < 127); }

// ---------------------------------------------------------------------------------------------------------------------
// block finder
// ---------------------------------------------------------------------------------------------------------------------
// Is bit `pos` of base[0, n) (n bytes + 16 readable bytes of padding) the start of a non-final dynamic block whose codes
// are complete and whose first symbols are text?
static inline bool plausible_block(const uint8_t *base, size_t n, uint64_t pos, inf::Decoder &scratch) {
    const uint8_t *in = base + (pos >> 3);
    if ((size_t)(in - base) + 64 > n) return false;
    const uint64_t v = inf::Decoder::load64(in) >> (pos & 7u);
    if ((v & 7u) != 4u) return false; // BFINAL = 0, BTYPE = 2
    const uint32_t hlit = (uint32_t)(v >> 3) & 31u, hdist = (uint32_t)(v >> 8) & 31u, hclen = ((uint32_t)(v >> 13) & 15u) + 4u;
    if (hlit > 29u || hdist > 29u) return false;
    { // the code-length code must be complete (every encoder's Huffman construction makes it so)
        const uint64_t p2 = pos + 17;
        const uint64_t w = inf::Decoder::load64(base + (p2 >> 3)) >> (p2 & 7u); // 57 bits: 19 lengths of 3
        uint32_t kraft = 0;
        for (uint32_t i = 0; i < hclen; ++i) {
            const uint32_t l = (uint32_t)(w >> (3u * i)) & 7u;
            if (l) kraft += 128u >> l;
        }
        if (kraft != 128u) return false;
    }
    // the real parser, on a decoder of its own
    const uint8_t *ip = base + (pos >> 3);
    scratch.start_at_bit(ip, (unsigned)(pos & 7u));
    if (scratch.block_header(ip, base + n) != inf::OK || scratch.state != inf::Decoder::CODES) return false;
    // complete literal/length code, complete (or single-symbol / empty) distance code: what the tables hold tells --
    // an incomplete code leaves invalid first-level entries behind
    for (int i = 0; i < (1 << inf::LIT_BITS); ++i)
        if (scratch.lit[i] == 0) return false;
    int dist_holes = 0;
    for (int i = 0; i < (1 << inf::DIST_BITS); ++i) dist_holes += scratch.dist[i] == 0;
    if (dist_holes != 0 && dist_holes != (1 << inf::DIST_BITS) / 2 && dist_holes != (1 << inf::DIST_BITS)) return false;
    // first symbols: text
    uint64_t bb = scratch.bitbuf;
    int bc = scratch.bitcnt;
    const uint8_t *const in_end = base + n;
    for (int sym = 0; sym < 512; ++sym) {
        if (ip > in_end) return false;
        bb |= inf::Decoder::load64(ip) << bc;
        ip += (63 - bc) >> 3;
        bc |= 56;
        uint32_t e = scratch.lit[bb & ((1u << inf::LIT_BITS) - 1u)];
        if (e & inf::K_SUB) {
            bb >>= inf::LIT_BITS;
            bc -= inf::LIT_BITS;
            e = scratch.lit[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
        }
        if (e & inf::K_LITERAL) {
            if (!text_byte((e >> 16) & 0xFFu)) return false;
            if (((e >> 8) & 15u) != 0u && !text_byte(e >> 24)) return false;
            bb >>= (e & 0xFFu);
            bc -= (int)(e & 0xFFu);
            continue;
        }
        if (e & inf::K_EOB) return sym > 0; // (an empty block tells nothing)
        if (!(e & inf::K_LEN)) return false;
        bb >>= (e & 0xFFu);
        bc -= (int)(e & 0xFFu);
        if (bc < 28) {
            bb |= inf::Decoder::load64(ip) << bc;
            ip += (63 - bc) >> 3;
            bc |= 56;
        }
        uint32_t d = scratch.dist[bb & ((1u << inf::DIST_BITS) - 1u)];
        if (d & inf::K_SUB) {
            bb >>= inf::DIST_BITS;
            bc -= inf::DIST_BITS;
            d = scratch.dist[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))];
        }
        if (!(d & inf::K_LEN)) return false;
        bb >>= (d & 0xFFu);
        bc -= (int)(d & 0xFFu);
    }
    return true;
}

// first plausible block start at or after bit `from`, below bit `to`; UINT64_MAX if there is none
static inline uint64_t find_block_start(const uint8_t *base, size_t n, uint64_t from, uint64_t to, inf::Decoder &scratch) {
    for (uint64_t pos = from; pos < to; ++pos) {
        // (the three header bits first: seven of eight offsets end here)
        if ((size_t)(pos >> 3) + 64 > n) break;
        const uint32_t b = (uint32_t)(inf::Decoder::load64(base + (pos >> 3)) >> (pos & 7u));
        if ((b & 7u) != 4u) continue;
        if (plausible_block(base, n, pos, scratch)) return pos;
    }
    return UINT64_MAX;
}

// ---------------------------------------------------------------------------------------------------------------------
// a chunk's output
// ---------------------------------------------------------------------------------------------------------------------
// Storage that grows without being zeroed or copied element by element (std::vector does both; realloc of a large block
// remaps its pages).
template <class T>
struct GrowBuf {
    T *p = nullptr;
    size_t cap = 0;
    GrowBuf() = default;
    GrowBuf(const GrowBuf &) = delete;
    GrowBuf &operator=(const GrowBuf &) = delete;
    GrowBuf(GrowBuf &&o) noexcept : p(o.p), cap(o.cap) {
        o.p = nullptr;
        o.cap = 0;
    }
    GrowBuf &operator=(GrowBuf &&o) noexcept {
        if (this != &o) {
            free(p);
            p = o.p;
            cap = o.cap;
            o.p = nullptr;
            o.cap = 0;
        }
        return *this;
    }
    ~GrowBuf() { free(p); }
    T *data() const { return p; }
    size_t size() const { return cap; }
    void reserve(size_t n) {
        if (n <= cap) return;
        T *q = (T *)realloc(p, n * sizeof(T));
        if (!q) throw std::bad_alloc();
        p = q;
        cap = n;
    }
    void release() {
        free(p);
        p = nullptr;
        cap = 0;
    }
};

template <class T, class U>
static inline GrowBuf<T> rebind(GrowBuf<U> &&b) { // the same memory as elements of another type
    GrowBuf<T> r;
    r.p = (T *)b.p;
    r.cap = b.cap * sizeof(U) / sizeof(T);
    b.p = nullptr;
    b.cap = 0;
    return r;
}

// Buffers the chunks of finished batches (and finished files) leave behind, for the next ones: fresh memory costs a
// page fault per 4 KiB when it is first written and another system call's worth when it is returned -- together about
// as much as decoding into it.  At most FINCH_PARGZ_POOL_MB (default 2048) are kept.
struct BufPool {
    std::mutex mu;
    std::vector<GrowBuf<uint8_t>> free_list;
    size_t held = 0, limit;
    BufPool() {
        const char *e = getenv("FINCH_PARGZ_POOL_MB");
        limit = (size_t)(e ? std::max(0ll, atoll(e)) : 2048ll) << 20;
    }
    static BufPool &global() {
        static BufPool p;
        return p;
    }
    GrowBuf<uint8_t> get() { // the largest one there is, or an empty one
        std::lock_guard<std::mutex> g(mu);
        if (free_list.empty()) return GrowBuf<uint8_t>();
        size_t best = 0;
        for (size_t i = 1; i < free_list.size(); ++i)
            if (free_list[i].cap > free_list[best].cap) best = i;
        GrowBuf<uint8_t> b = std::move(free_list[best]);
        free_list[best] = std::move(free_list.back());
        free_list.pop_back();
        held -= b.cap;
        return b;
    }
    void put(GrowBuf<uint8_t> &&b) {
        if (!b.cap) return;
        std::lock_guard<std::mutex> g(mu);
        if (held + b.cap > limit) return; // (b is freed by its destructor)
        held += b.cap;
        free_list.push_back(std::move(b));
    }
};

// A chunk's text.  A chunk that starts at a known window (the first of a batch) is plain bytes.  Any other one is decoded
// into bytes too, with what the unknown window shows through kept on the side: `dirty` has one bit per 8-byte word of the
// text (the WINDOW bytes in front of it included), and where a word's bit is set, `sym` holds all its positions as 16-bit
// symbols -- a byte, or MARK + i for byte i of the window.  Everything else in `sym` is never touched (no memory behind it).
struct Chunk {
    uint64_t start_bit = UINT64_MAX, end_bit = 0;
    bool known_window = false; // the first chunk of a batch
    bool ok = true;            // false: a code that cannot be, a match reaching beyond the window
    bool member_end = false;   // the final block ended at end_bit
    bool out_of_input = false; // the batch's bytes ended inside the block that starts at end_bit
    size_t lead = 0;           // bytes in front of the text in `bytes` (WINDOW when the window is unknown)
    GrowBuf<uint8_t> bytes;
    size_t n_bytes = 0;        // length of the text
    GrowBuf<uint16_t> sym;     // (same indexing as bytes)
    GrowBuf<uint8_t> dirty;    // bit w: word w of `bytes` holds something of the unknown window
    size_t n_dirty_words = 0;  // (statistics)
    std::vector<uint8_t> win_in; // the text in front of the chunk, once it is known
    size_t text_len() const { return n_bytes; }
    const uint8_t *text() const { return bytes.data() + lead; }
};

static inline bool bit_at(const uint8_t *bm, size_t i) { return (bm[i >> 3] >> (i & 7)) & 1u; }
static inline void set_bits(uint8_t *bm, size_t lo, size_t hi) { // [lo, hi]
    for (size_t i = lo; i <= hi; ++i) bm[i >> 3] |= (uint8_t)(1u << (i & 7));
}

// room for `want` positions in all three arrays (GrowBuf::size() is the capacity)
static inline void grow_sparse(Chunk &c, size_t want) {
    const size_t old = c.bytes.size();
    if (want + 64 > old) c.bytes.reserve(std::max(want + 64, old + old / 2 + 65536));
    c.sym.reserve(c.bytes.size());
    const size_t old_bm = c.dirty.size(), bm = c.bytes.size() / 64 + 64;
    if (bm > old_bm) {
        c.dirty.reserve(bm);
        memset(c.dirty.data() + old_bm, 0, c.dirty.size() - old_bm);
    }
}

static inline size_t count_dirty_words(const Chunk &c) {
    if (c.known_window) return 0;
    size_t n = 0;
    const size_t w0 = c.lead >> 3, w1 = (c.lead + c.n_bytes + 7) >> 3;
    for (size_t w = w0; w < w1; ++w) n += bit_at(c.dirty.data(), w);
    return n;
}

// The symbols of the block at hand (dec.state == CODES) into the chunk.  o: index of the next output position in
// c.bytes (lead included); fix_until: positions below it lie in a word that is already dirty, so whatever is written there
// goes into c.sym as well.  OK at the end of the block.
static inline inf::Status sparse_codes(inf::Decoder &dec, const uint8_t *&in_ref, const uint8_t *in_end, Chunk &c, size_t &o_ref, size_t &fix_ref) {
    const uint8_t *in = in_ref;
    uint64_t bb = dec.bitbuf;
    int bc = dec.bitcnt;
    const uint32_t *const LT = dec.lit, *const DT = dec.dist;
    size_t o = o_ref, fix_until = fix_ref;
    inf::Status result = inf::OK;
    for (;;) {
        if (in > in_end) { // (into the padding: the block does not end in this buffer)
            result = inf::NEED_INPUT;
            break;
        }
        if (o + 764 > c.bytes.size()) grow_sparse(c, o + 764);
        uint8_t *const o8 = c.bytes.data();
        uint16_t *const o16 = c.sym.data();
        uint8_t *const bm = c.dirty.data();
        bb |= inf::Decoder::load64(in) << bc;
        in += (63 - bc) >> 3;
        bc |= 56;
        uint32_t e = LT[bb & ((1u << inf::LIT_BITS) - 1u)];
        for (;;) { // literals while the buffered bits last
            if (__builtin_expect((e & inf::K_SUB) != 0, 0)) {
                bb >>= inf::LIT_BITS;
                bc -= inf::LIT_BITS;
                e = LT[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
            }
            if (!(e & inf::K_LITERAL)) break;
            o8[o] = (uint8_t)(e >> 16);
            o8[o + 1] = (uint8_t)(e >> 24);
            if (__builtin_expect(o < fix_until, 0)) {
                o16[o] = (uint8_t)(e >> 16);
                o16[o + 1] = (uint8_t)(e >> 24);
            }
            o += 1 + (((e >> 8) & 15u) != 0u);
            bb >>= (e & 0xFFu);
            bc -= (int)(e & 0xFFu);
            if (bc < 20) goto next_symbol;
            e = LT[bb & ((1u << inf::LIT_BITS) - 1u)];
        }
        if (__builtin_expect(!(e & inf::K_LEN), 0)) {
            if (e & inf::K_EOB) {
                bb >>= (e & 0xFFu);
                bc -= (int)(e & 0xFFu);
                dec.state = dec.final_block ? inf::Decoder::DONE : inf::Decoder::HEADER;
                break;
            }
            result = inf::BAD;
            break;
        }
        {
            const uint32_t total = e & 0xFFu, nx = (e >> 8) & 15u;
            const uint32_t length = (e >> 16) + (uint32_t)((bb >> (total - nx)) & ((1u << nx) - 1u));
            bb >>= total;
            bc -= (int)total;
            if (bc < 28) {
                bb |= inf::Decoder::load64(in) << bc;
                in += (63 - bc) >> 3;
                bc |= 56;
            }
            uint32_t d = DT[bb & ((1u << inf::DIST_BITS) - 1u)];
            if (__builtin_expect((d & inf::K_SUB) != 0, 0)) {
                bb >>= inf::DIST_BITS;
                bc -= inf::DIST_BITS;
                d = DT[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))];
            }
            if (__builtin_expect(!(d & inf::K_LEN), 0)) {
                result = inf::BAD;
                break;
            }
            const uint32_t dtotal = d & 0xFFu, dnx = (d >> 8) & 15u;
            const uint32_t distance = (d >> 16) + (uint32_t)((bb >> (dtotal - dnx)) & ((1u << dnx) - 1u));
            bb >>= dtotal;
            bc -= (int)dtotal;
            // (distance <= 32768 <= lead: the source always exists)
            const size_t src = o - distance;
            // does the part of the source that exists already touch a dirty word?  (<= 34 words: one 64-bit look)
            const uint32_t span = length < distance ? length : distance;
            const size_t ws = src >> 3, nw = ((src + span - 1) >> 3) - ws + 1;
            const uint64_t bits = (inf::Decoder::load64(bm + (ws >> 3)) >> (ws & 7u)) & ((1ull << nw) - 1ull);
            if (__builtin_expect(bits == 0, 1)) {
                const uint8_t *sp = o8 + src;
                uint8_t *op = o8 + o, *const end = op + length;
                if (distance >= 16) {
                    do {
                        memcpy(op, sp, 16);
                        op += 16;
                        sp += 16;
                    } while (op < end);
                } else if (distance >= 8) {
                    do {
                        memcpy(op, sp, 8);
                        op += 8;
                        sp += 8;
                    } while (op < end);
                } else if (distance == 1) {
                    const uint64_t v = 0x0101010101010101ull * sp[0];
                    do {
                        memcpy(op, &v, 8);
                        op += 8;
                    } while (op < end);
                } else {
                    do *op++ = *sp++;
                    while (op < end);
                }
                if (__builtin_expect(o < fix_until, 0)) {
                    const size_t lim = std::min<size_t>(o + length, fix_until);
                    for (size_t p2 = o; p2 < lim; ++p2) o16[p2] = o8[p2];
                }
                o += length;
            } else {
                const size_t w0 = o >> 3, w1 = (o + length - 1) >> 3;
                if (!bit_at(bm, w0))
                    for (size_t p2 = w0 << 3; p2 < o; ++p2) o16[p2] = o8[p2]; // what the first word holds so far
                set_bits(bm, w0, w1);
                for (uint32_t i = 0; i < length; ++i) {
                    const size_t p2 = src + i;
                    const uint16_t v = bit_at(bm, p2 >> 3) ? o16[p2] : (uint16_t)o8[p2];
                    o16[o + i] = v;
                    o8[o + i] = (uint8_t)v;
                }
                o += length;
                fix_until = (w1 + 1) << 3;
            }
        }
    next_symbol:;
    }
    bb &= bc >= 64 ? ~0ull : ((1ull << bc) - 1ull);
    dec.bitbuf = bb;
    dec.bitcnt = bc;
    in_ref = in;
    o_ref = o;
    fix_ref = fix_until;
    return result;
}

// Decode chunk `ci` of the batch base[0, n) (n bytes + 64 bytes of zero padding) from its start_bit.  It stops at the first
// block boundary where a later chunk was found to begin (start_bit; UINT64_MAX: nowhere); starts it runs past were not
// starts.  With a known window (the first chunk of a batch) `window` holds the text in front of it.
static inline void decode_chunk(const uint8_t *base, size_t n, std::vector<Chunk> &chunks, size_t ci, const uint8_t *window, size_t window_len) {
    Chunk &c = chunks[ci];
    std::unique_ptr<inf::Decoder> dec(new inf::Decoder());
    const uint8_t *in = base + (c.start_bit >> 3);
    const uint8_t *const in_end = base + n;
    dec->start_at_bit(in, (unsigned)(c.start_bit & 7u));
    dec->stop_at_block_end = true;
    size_t next = ci + 1;
    auto at_boundary = [&](uint64_t pos) { // true: this is where a later chunk begins (the ones passed over began nowhere)
        while (next < chunks.size() && (chunks[next].start_bit == UINT64_MAX || chunks[next].start_bit < pos)) next++;
        return next < chunks.size() && chunks[next].start_bit == pos;
    };
    // (a chunk that began at a false start decodes noise until a code fails or this much has come out of it)
    const size_t text_cap = std::max<size_t>((size_t)64 << 20, (n / std::max<size_t>(1, chunks.size())) * 256);
    c.end_bit = c.start_bit;
    const size_t guess = std::max<size_t>((size_t)1 << 20, (n / std::max<size_t>(1, chunks.size())) * 6);
    size_t o = 0, fix_until = 0;
    if (c.known_window) {
        c.lead = 0;
        c.bytes.reserve(guess);
        dec->ext_end = window + window_len;
        dec->ext_len = window_len;
    } else {
        c.lead = WINDOW;
        c.dirty.release();
        grow_sparse(c, WINDOW + guess);
        memset(c.dirty.data(), 0, c.dirty.size());
        memset(c.dirty.data(), 0xFF, WINDOW / 64); // every word of the window is "dirty"
        for (uint32_t i = 0; i < WINDOW; ++i) c.sym.data()[i] = (uint16_t)(MARK | i);
        memset(c.bytes.data(), 0, WINDOW);
        o = WINDOW;
        fix_until = WINDOW;
    }
    for (;;) {
        // ---- one block ----
        const uint64_t block_start = dec->bit_position(in, base);
        const size_t sv_o = o, sv_bytes = c.n_bytes, sv_fix = fix_until;
        inf::Status s;
        if (!c.known_window) {
            s = dec->block_header(in, in_end);
            if (s == inf::OK && dec->state == inf::Decoder::CODES) {
                s = sparse_codes(*dec, in, in_end, c, o, fix_until);
            } else if (s == inf::OK && dec->state == inf::Decoder::STORED) { // literal bytes, byte aligned
                while (dec->stored_left && s == inf::OK) {
                    if (o + 128 > c.bytes.size()) grow_sparse(c, o + 65536);
                    uint8_t b;
                    if (dec->bitcnt) {
                        b = (uint8_t)(dec->bitbuf & 0xFFu);
                        dec->bitbuf >>= 8;
                        dec->bitcnt -= 8;
                    } else if (in >= in_end) {
                        s = inf::NEED_INPUT;
                        break;
                    } else {
                        b = *in++;
                    }
                    c.bytes.data()[o] = b;
                    if (o < fix_until) c.sym.data()[o] = b;
                    o++;
                    dec->stored_left--;
                }
                if (s == inf::OK) dec->state = dec->final_block ? inf::Decoder::DONE : inf::Decoder::HEADER;
            }
            if (s == inf::OK) s = inf::BLOCK_END;
            c.n_bytes = o - WINDOW;
        } else {
            for (;;) {
                if (c.bytes.size() - c.n_bytes < (size_t)1 << 20) c.bytes.reserve(c.bytes.size() + c.bytes.size() / 2);
                uint8_t *op = c.bytes.data() + c.n_bytes;
                s = dec->run(in, in_end, op, c.bytes.data() + c.bytes.size(), c.bytes.data());
                c.n_bytes = (size_t)(op - c.bytes.data());
                if (s != inf::NEED_OUTPUT) break;
            }
            if (s == inf::STREAM_END) s = inf::BLOCK_END; // (cannot happen: DONE is checked below)
        }
        if (s == inf::NEED_INPUT || (s == inf::BLOCK_END && dec->bit_position(in, base) > (uint64_t)n * 8u)) {
            // the batch ends inside this block: it belongs to the next batch.  (What the block wrote stays behind the
            // text's end; dirty bits it set there are beyond what anyone looks at.)
            o = sv_o;
            fix_until = sv_fix;
            c.n_bytes = sv_bytes;
            c.end_bit = block_start;
            c.out_of_input = true;
            return;
        }
        if (s != inf::BLOCK_END) {
            c.ok = false;
            return;
        }
        const uint64_t pos = dec->bit_position(in, base);
        c.end_bit = pos;
        if (dec->state == inf::Decoder::DONE) {
            c.member_end = true;
            return;
        }
        if (at_boundary(pos)) return;
        if (c.text_len() > text_cap) {
            c.ok = false;
            return;
        }
    }
}

// The window marks of text positions [lo, hi) of chunk c put right, given the window in front of it (window_len bytes
// ending at window_end).  false: a mark that points before the start of the stream.
static inline bool resolve_range(Chunk &c, size_t lo, size_t hi, const uint8_t *window_end, size_t window_len) {
    if (c.known_window || lo >= hi) return true;
    bool ok = true;
    uint8_t *const o8 = c.bytes.data();
    const uint16_t *const o16 = c.sym.data();
    const uint8_t *const bm = c.dirty.data();
    const size_t a = c.lead + lo, b = c.lead + hi;
    for (size_t w = a >> 3; w <= (b - 1) >> 3; ++w) {
        if ((w & 7) == 0 && bm[w >> 3] == 0 && w + 8 <= ((b - 1) >> 3)) { // eight clean words at once
            w += 7;
            continue;
        }
        if (!bit_at(bm, w)) continue;
        const size_t p0 = std::max(a, w << 3), p1 = std::min(b, (w + 1) << 3);
        for (size_t p = p0; p < p1; ++p) {
            const uint16_t v = o16[p];
            if (v & MARK) {
                const size_t back = WINDOW - (size_t)(v & 0x7FFFu); // 1 = the byte right in front of the chunk
                if (back > window_len) ok = false;
                else o8[p] = window_end[-(ptrdiff_t)back];
            }
        }
    }
    return ok;
}

// The window behind chunk c -- the last WINDOW bytes of (window in front of it + its text).  Puts the marks of the chunk's
// own last WINDOW bytes right on the way.
static inline bool window_behind(Chunk &c, const std::vector<uint8_t> &win_in, std::vector<uint8_t> &win_out) {
    const size_t total = c.text_len();
    const size_t take = std::min<size_t>(WINDOW, total);
    const bool ok = resolve_range(c, total - take, total, win_in.data() + win_in.size(), win_in.size());
    const size_t old = take < WINDOW ? std::min<size_t>(WINDOW - take, win_in.size()) : 0; // (a short chunk: the older window shows through)
    std::vector<uint8_t> w(old + take);
    if (old) memcpy(w.data(), win_in.data() + win_in.size() - old, old);
    if (take) memcpy(w.data() + old, c.text() + (total - take), take);
    win_out.swap(w);
    return ok;
}

} // namespace pargz
} // namespace finch
