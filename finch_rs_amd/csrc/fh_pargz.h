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
//   2. Every thread decodes from its start.  What lies before it is unknown, so the output is 16-bit symbols: a byte, or
//      0x8000 + i for "byte i of the 32 KiB window in front of this chunk"; copying a match copies such markers along.
//      As soon as the last 32 KiB produced hold no marker, everything after them is independent of the unknown window and
//      the thread switches to the ordinary byte decoder (fh_inflate.h, full speed).  A thread stops at the block boundary
//      where the next chunk was found to begin; if it runs past that offset instead, the "start" was not one, the next
//      chunk's work is dropped and the thread carries on to the one after.
//   3. In order: the window in front of chunk i is the last 32 KiB of the text up to it (taken from the chunk tails alone).
//      The markers are looked up when the text is handed out: symbols are narrowed straight into the caller's buffer by
//      several threads, each of which checksums what it has just written; the CRC-32s are joined with crc32_combine.
//
// The member's CRC-32 and ISIZE are checked at its end as always, so a chunk stitched wrongly cannot go unnoticed.
#pragma once
#include "fh_options.h"
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
using fh::cfg;

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
        const char *e = cfg("pargz_pool_mb");
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

struct Chunk {
    uint64_t start_bit = UINT64_MAX, end_bit = 0;
    bool known_window = false; // the first chunk of a batch: decoded as bytes from the start
    bool ok = true;            // false: a code that cannot be, a match reaching beyond the window
    bool member_end = false;   // the final block ended at end_bit
    bool out_of_input = false; // the batch's bytes ended inside the block that starts at end_bit
    GrowBuf<uint16_t> sym;     // symbols while the unknown window can still show through (WINDOW marker slots in front)
    size_t n_sym = 0;          // symbols behind the WINDOW slots
    std::vector<uint8_t> win_in; // (pass 3) the text in front of the chunk: what its markers point into
    GrowBuf<uint8_t> bytes;
    size_t n_bytes = 0;
    size_t text_len() const { return n_sym + n_bytes; }
};

// Symbols of the block at hand (dec.state == CODES) as 16-bit values behind c.sym[WINDOW + c.n_sym).  last_marker: index
// (in symbols behind the window slots) just past the latest marker written.  OK at the end of the block.
static inline inf::Status marker_codes(inf::Decoder &dec, const uint8_t *&in_ref, const uint8_t *in_end, Chunk &c, size_t &last_marker) {
    const uint8_t *in = in_ref;
    uint64_t bb = dec.bitbuf;
    int bc = dec.bitcnt;
    const uint32_t *const LT = dec.lit, *const DT = dec.dist;
    size_t o = WINDOW + c.n_sym;
    inf::Status result = inf::OK;
    for (;;) {
        if (in > in_end) { // (into the padding: the block does not end in this buffer)
            result = inf::NEED_INPUT;
            break;
        }
        if (o + 600 > c.sym.size()) c.sym.reserve(c.sym.size() + c.sym.size() / 2 + 65536);
        uint16_t *const out = c.sym.data();
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
            out[o] = (uint16_t)((e >> 16) & 0xFFu);
            out[o + 1] = (uint16_t)(e >> 24);
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
            // (distance <= 32768 = the marker slots in front: the source always exists)
            const uint16_t *src = out + o - distance;
            uint16_t any = 0;
#if defined(__SSE2__)
            if (distance >= 8) { // eight symbols a step (a step may run up to seven past the match: room is kept, and what
                                 // it drags along can only make `any` see a marker early)
                __m128i acc = _mm_setzero_si128();
                uint16_t *dst = out + o;
                for (uint32_t i = 0; i < length; i += 8) {
                    const __m128i v = _mm_loadu_si128((const __m128i *)(src + i));
                    _mm_storeu_si128((__m128i *)(dst + i), v);
                    acc = _mm_or_si128(acc, v);
                }
                any = (uint16_t)((_mm_movemask_epi8(acc) & 0xAAAA) ? MARK : 0);
            } else
#endif
            {
                for (uint32_t i = 0; i < length; ++i) {
                    out[o + i] = src[i];
                    any |= src[i];
                }
            }
            o += length;
            if (any & MARK) last_marker = o - WINDOW;
        }
    next_symbol:;
    }
    bb &= bc >= 64 ? ~0ull : ((1ull << bc) - 1ull);
    dec.bitbuf = bb;
    dec.bitcnt = bc;
    in_ref = in;
    c.n_sym = o - WINDOW;
    return result;
}

// Decode chunk `ci` of the batch base[0, n) (n bytes + 64 bytes of zero padding) from its start_bit.  It stops at the first block
// boundary where a later chunk was found to begin (start_bit; UINT64_MAX: nowhere); starts it runs past were not starts.  With a known
// window (the first chunk of a batch) `window` holds the text in front of it.
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
    bool clean = c.known_window;
    uint8_t pre[WINDOW]; // the WINDOW bytes in front of `bytes` once the markers have faded
    size_t last_marker = 0;
    if (!clean) {
        c.sym.reserve(WINDOW + std::max<size_t>((size_t)1 << 20, (n / std::max<size_t>(1, chunks.size())) * 6));
        for (uint32_t i = 0; i < WINDOW; ++i) c.sym.data()[i] = (uint16_t)(MARK | i);
        last_marker = 0;
        // (the window slots count as markers at "position 0": clean once WINDOW symbols without one have been produced)
    } else {
        dec->ext_end = window + window_len;
        dec->ext_len = window_len;
    }
    c.bytes.reserve((size_t)4 << 20);
    for (;;) {
        // ---- one block ----
        const uint64_t block_start = dec->bit_position(in, base);
        const size_t sv_sym = c.n_sym, sv_bytes = c.n_bytes;
        inf::Status s;
        if (!clean) {
            s = dec->block_header(in, in_end);
            if (s == inf::OK && dec->state == inf::Decoder::CODES) {
                s = marker_codes(*dec, in, in_end, c, last_marker);
            } else if (s == inf::OK && dec->state == inf::Decoder::STORED) { // literal bytes, byte aligned
                while (dec->stored_left && s == inf::OK) {
                    if (WINDOW + c.n_sym + 8 > c.sym.size()) c.sym.reserve(c.sym.size() + c.sym.size() / 2 + 65536);
                    if (dec->bitcnt) {
                        c.sym.data()[WINDOW + c.n_sym++] = (uint16_t)(dec->bitbuf & 0xFFu);
                        dec->bitbuf >>= 8;
                        dec->bitcnt -= 8;
                        dec->stored_left--;
                    } else if (in >= in_end) {
                        s = inf::NEED_INPUT;
                    } else {
                        c.sym.data()[WINDOW + c.n_sym++] = *in++;
                        dec->stored_left--;
                    }
                }
                if (s == inf::OK) dec->state = dec->final_block ? inf::Decoder::DONE : inf::Decoder::HEADER;
            }
            if (s == inf::OK) s = inf::BLOCK_END;
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
            // the batch ends inside this block: it belongs to the next batch
            c.n_sym = sv_sym;
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
        // (a guard against false starts -- a speculative start that happens to decode keeps producing garbage.  A chunk
        //  whose window is known starts where its predecessor really ended: whatever it inflates to, a run of 300 MB of 'N'
        //  included, is the stream's own text)
        if (!c.known_window && c.text_len() > text_cap) {
            c.ok = false;
            return;
        }
        if (!clean && c.n_sym >= last_marker + WINDOW) { // no marker in the last WINDOW symbols: bytes from here on
            const uint16_t *tail = c.sym.data() + WINDOW + c.n_sym - WINDOW;
            for (uint32_t i = 0; i < WINDOW; ++i) pre[i] = (uint8_t)tail[i];
            dec->ext_end = pre + WINDOW;
            dec->ext_len = WINDOW;
            clean = true;
        }
    }
}

// one symbol of a chunk given the window in front of it (window_len bytes ending at window_end)
static inline bool resolve_symbol(uint16_t v, const uint8_t *window_end, size_t window_len, uint8_t *out) {
    if (!(v & MARK)) {
        *out = (uint8_t)v;
        return true;
    }
    const size_t back = WINDOW - (size_t)(v & 0x7FFFu); // 1 = the byte right in front of the chunk
    if (back > window_len) { // a stream that reaches before its own start
        *out = 0;
        return false;
    }
    *out = window_end[-(ptrdiff_t)back];
    return true;
}

// n symbols narrowed to bytes, markers looked up in the window (window_len bytes ending at window_end)
static inline bool resolve_span(const uint16_t *s, size_t n, const uint8_t *window_end, size_t window_len, uint8_t *h) {
    bool ok = true;
    size_t i = 0;
#if defined(__SSE2__)
    // sixteen symbols at a time: narrowed as they are when none of them is a marker (markers are few and far between
    // once a chunk is a few hundred kilobytes in)
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 8));
        if (__builtin_expect(_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xAAAA, 0)) { // a top bit set: a marker among them
            for (size_t j = i; j < i + 16; ++j) ok = resolve_symbol(s[j], window_end, window_len, h + j) && ok;
        } else {
            _mm_storeu_si128((__m128i *)(h + i), _mm_packus_epi16(a, b));
        }
    }
#endif
    for (; i < n; ++i) ok = resolve_symbol(s[i], window_end, window_len, h + i) && ok;
    return ok;
}

// The window behind chunk c -- the last WINDOW bytes of (window in front of it + its text) -- without resolving all of it.
static inline bool window_behind(const Chunk &c, const std::vector<uint8_t> &win_in, std::vector<uint8_t> &win_out) {
    const size_t total = c.n_sym + c.n_bytes;
    const size_t take = std::min<size_t>(WINDOW, total);
    const size_t old = take < WINDOW ? std::min<size_t>(WINDOW - take, win_in.size()) : 0; // (a short chunk: the older window shows through)
    std::vector<uint8_t> w(old + take);
    if (old) memcpy(w.data(), win_in.data() + win_in.size() - old, old);
    const size_t first = total - take; // of the chunk's text
    const size_t from_sym = first < c.n_sym ? c.n_sym - first : 0;
    const bool ok = resolve_span(c.sym.data() + WINDOW + first, from_sym, win_in.data() + win_in.size(), win_in.size(), w.data() + old);
    if (take > from_sym) memcpy(w.data() + old + from_sym, c.bytes.data() + (first + from_sym - c.n_sym), take - from_sym);
    win_out.swap(w);
    return ok;
}

} // namespace pargz
} // namespace finch
