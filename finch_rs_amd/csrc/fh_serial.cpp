// fh_serial.cpp -- the sketch file formats of finch next to the `.sk` writer in fh_host.cpp:
//
//   .bsk  write_finch_file / read_finch_file   lib/src/serialization/mod.rs:123-166, 168-222   schema finch.capnp
//   .msh  write_mash_file  / read_mash_file    lib/src/serialization/mash.rs:12-58, 60-135     schema mash.capnp
//   .sk   MultiSketch::to_sketches (reader)    lib/src/serialization/json.rs:92-139, 160-262, filtering.rs:110-134
//   open_sketch_file                           lib/src/lib.rs:96-118
//
// The reference serialises through the capnp crate (`capnp::serialize::write_message`: the standard UNPACKED stream
// framing).  There is no Cap'n Proto runtime in this image, and none is needed: the wire encoding of a fixed schema is a
// public specification.  The writers below lay the two schemas out by hand as single-segment messages; the struct layouts
// (data / pointer section sizes and every field offset) are the ones capnpc computed for the reference and committed in
// its generated code -- lib/src/serialization/finch_capnp.rs:80-97,201,253-278,398,450-473,591,643-690,844,979 and
// mash_capnp.rs:53-107,307,441,492-550,743 -- cited at each struct.  The readers take any valid message (several
// segments, far pointers, list-of-struct upgrades), because files written by the reference itself are multi-segment.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "fh_host_model.h"

namespace finch {

// =====================================================================================================================
// Cap'n Proto encoding (https://capnproto.org/encoding.html), the subset the two schemas need
// =====================================================================================================================
namespace capnp {

constexpr uint64_t MAX_WORDS = 1ull << 29; // pointer offsets are 30-bit signed word counts

// ---- builder: one segment, bump allocation ----
struct Builder {
    std::vector<uint64_t> w;
    bool too_big = false;
    size_t alloc(size_t n) {
        const size_t at = w.size();
        if (at + n > MAX_WORDS) {
            too_big = true;
            return at;
        }
        w.resize(at + n, 0);
        return at;
    }
    // struct pointer at word `at` -> struct at `target` (finch_capnp.rs STRUCT_SIZE values are passed by the callers)
    void struct_ptr(size_t at, size_t target, uint16_t data_words, uint16_t ptr_words) {
        const int64_t off = (int64_t)target - (int64_t)at - 1;
        w[at] = ((uint64_t)((uint32_t)(off << 2) | 0u)) | ((uint64_t)data_words << 32) | ((uint64_t)ptr_words << 48);
    }
    // list pointer: elem = 2 (bytes), 4 (u32), 5 (u64), 7 (composite: count = words after the tag)
    void list_ptr(size_t at, size_t target, unsigned elem, uint32_t count) {
        const int64_t off = (int64_t)target - (int64_t)at - 1;
        w[at] = ((uint64_t)((uint32_t)(off << 2) | 1u)) | ((uint64_t)elem << 32) | ((uint64_t)count << 35);
    }
    void bytes(size_t ptr_at, const void *p, size_t n, bool nul_terminated) {
        const size_t count = n + (nul_terminated ? 1 : 0);
        const size_t at = alloc((count + 7) / 8);
        if (too_big) return;
        if (n) memcpy((uint8_t *)(w.data() + at), p, n);
        list_ptr(ptr_at, at, 2, (uint32_t)count);
    }
    void text(size_t ptr_at, const std::string &s) { bytes(ptr_at, s.data(), s.size(), true); }   // Text: NUL counted
    void data(size_t ptr_at, const std::string &s) { bytes(ptr_at, s.data(), s.size(), false); }  // Data
    // List(struct): tag word (element count in the offset field + per-element sizes), then the elements
    size_t struct_list(size_t ptr_at, uint32_t n, uint16_t data_words, uint16_t ptr_words) {
        const size_t per = (size_t)data_words + ptr_words;
        const size_t at = alloc(1 + per * n);
        if (too_big) return at;
        w[at] = ((uint64_t)(n << 2)) | ((uint64_t)data_words << 32) | ((uint64_t)ptr_words << 48);
        list_ptr(ptr_at, at, 7, (uint32_t)(per * n));
        return at + 1;
    }
    // the stream framing of capnp::serialize::write_message: segment count - 1, segment sizes, padding, segments
    void frame(std::string &out) const {
        const uint32_t hdr[2] = {0u, (uint32_t)w.size()};
        out.assign((const char *)hdr, 8);
        out.append((const char *)w.data(), w.size() * 8);
    }
};

static uint64_t f64_bits(double v) {
    uint64_t b;
    memcpy(&b, &v, 8);
    return b;
}
static double bits_f64(uint64_t b) {
    double v;
    memcpy(&v, &b, 8);
    return v;
}

// ---- reader ----
struct Segment {
    const uint64_t *p;
    uint64_t n;
};

struct Message {
    std::vector<uint64_t> store; // the message copied to aligned storage
    std::vector<Segment> segs;
    std::string err;

    bool fail(const char *what) {
        if (err.empty()) err = what;
        return false;
    }
    bool parse(const uint8_t *data, uint64_t len) {
        if (len < 8) return fail("message shorter than its header");
        uint32_t nseg_m1;
        memcpy(&nseg_m1, data, 4);
        const uint64_t nseg = (uint64_t)nseg_m1 + 1;
        if (nseg > 512) return fail("too many segments");
        const uint64_t hdr_bytes = ((4 + 4 * nseg) + 7) / 8 * 8;
        if (len < hdr_bytes) return fail("truncated segment table");
        std::vector<uint32_t> sizes(nseg);
        memcpy(sizes.data(), data + 4, 4 * nseg);
        uint64_t total = 0;
        for (uint32_t s : sizes) total += s;
        if (len < hdr_bytes + total * 8) return fail("message shorter than its segment table says");
        store.resize(total ? total : 1);
        memcpy(store.data(), data + hdr_bytes, total * 8);
        uint64_t off = 0;
        for (uint32_t s : sizes) {
            segs.push_back(Segment{store.data() + off, s});
            off += s;
        }
        return true;
    }
};

struct StructR {
    Message *m = nullptr;
    uint32_t seg = 0;
    uint64_t data = 0, ptrs = 0; // word offsets inside the segment
    uint32_t data_bits = 0;      // (a list element may be narrower than a word)
    uint16_t n_ptrs = 0;
    uint8_t byte_shift = 0;      // a 1- / 2- / 4-byte list element read as a struct: where in its word it starts
    bool null() const { return m == nullptr; }
    // the first word of the data section with the bits that are not the struct's cleared (a whole word when data_bits >= 64)
    uint64_t first() const {
        const uint64_t w = m->segs[seg].p[data] >> (8 * byte_shift);
        return data_bits >= 64 ? w : (w & ((1ull << data_bits) - 1));
    }
    uint64_t word(unsigned i) const { // 64-bit field i of the data section; fields beyond it read as 0 (schema evolution)
        if (!m || (uint64_t)(i + 1) * 64 > data_bits) return 0;
        return m->segs[seg].p[data + i];
    }
    uint32_t u32(unsigned i) const { // 32-bit field i
        if (!m || (uint64_t)(i + 1) * 32 > data_bits) return 0;
        if (data_bits < 64) return (uint32_t)first();
        return (uint32_t)(m->segs[seg].p[data + i / 2] >> (32 * (i & 1)));
    }
    uint16_t u16(unsigned i) const {
        if (!m || (uint64_t)(i + 1) * 16 > data_bits) return 0;
        if (data_bits < 64) return (uint16_t)(first() >> (16 * i));
        return (uint16_t)(m->segs[seg].p[data + i / 4] >> (16 * (i & 3)));
    }
    uint8_t u8(unsigned i) const {
        if (!m || (uint64_t)(i + 1) * 8 > data_bits) return 0;
        if (data_bits < 64) return (uint8_t)(first() >> (8 * i));
        return (uint8_t)(m->segs[seg].p[data + i / 8] >> (8 * (i & 7)));
    }
    bool bit(unsigned i) const {
        if (!m || i >= data_bits) return false;
        if (data_bits < 64) return (first() >> i) & 1u;
        return (m->segs[seg].p[data + i / 64] >> (i & 63)) & 1u;
    }
};

struct ListR {
    Message *m = nullptr;
    uint32_t seg = 0;
    uint64_t at = 0;     // first element (word offset)
    uint32_t count = 0;
    unsigned elem = 0;   // size code
    uint16_t data_words = 0, n_ptrs = 0; // composite elements
};

static bool valid_utf8(const unsigned char *s, size_t n) {
    size_t i = 0;
    while (i < n) {
        const unsigned char c = s[i];
        size_t len;
        uint32_t cp;
        if (c < 0x80) { ++i; continue; }
        else if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1Fu; }
        else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0Fu; }
        else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07u; }
        else return false;
        if (i + len > n) return false;
        for (size_t j = 1; j < len; ++j) {
            if ((s[i + j] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (s[i + j] & 0x3Fu);
        }
        if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF))
            return false;
        i += len;
    }
    return true;
}

struct Walker {
    Message &m;
    explicit Walker(Message &m_) : m(m_) {}
    // A pointer word at (seg, at): follow far pointers until a struct / list pointer and its base are known.
    // Returns false for a null pointer (ok stays true) or on malformed input (ok = false).
    bool resolve(uint32_t seg, uint64_t at, uint64_t &word, uint32_t &tseg, uint64_t &target, bool &ok) {
        ok = true;
        if (seg >= m.segs.size() || at >= m.segs[seg].n) return ok = m.fail("pointer outside its segment");
        uint64_t w = m.segs[seg].p[at];
        if (w == 0) return false;
        if ((w & 3) == 2) { // far pointer
            const bool dbl = (w >> 2) & 1;
            const uint64_t off = (uint32_t)w >> 3;
            const uint32_t sid = (uint32_t)(w >> 32);
            if (sid >= m.segs.size() || off + (dbl ? 2 : 1) > m.segs[sid].n) return ok = m.fail("far pointer out of bounds");
            if (!dbl) { // the landing pad is an ordinary pointer, relative to itself
                seg = sid;
                at = off;
                w = m.segs[seg].p[at];
                if (w == 0) return false;
                if ((w & 3) == 2) return ok = m.fail("far pointer to a far pointer");
            } else { // pad[0]: far pointer to the content, pad[1]: tag with the type and sizes
                const uint64_t p0 = m.segs[sid].p[off], tag = m.segs[sid].p[off + 1];
                if ((p0 & 3) != 2 || ((p0 >> 2) & 1)) return ok = m.fail("bad double-far landing pad");
                tseg = (uint32_t)(p0 >> 32);
                target = (uint32_t)p0 >> 3;
                if (tseg >= m.segs.size()) return ok = m.fail("far pointer out of bounds");
                word = tag;
                return true;
            }
        }
        if ((w & 3) == 3) return ok = m.fail("capability pointer in a data file");
        const int64_t off = (int64_t)((int32_t)(uint32_t)w >> 2);
        const int64_t t = (int64_t)at + 1 + off;
        if (t < 0 || (uint64_t)t > m.segs[seg].n) return ok = m.fail("pointer target out of bounds");
        word = w;
        tseg = seg;
        target = (uint64_t)t;
        return true;
    }
    bool get_struct(uint32_t seg, uint64_t at, StructR &out) { // false = malformed; out.null() = null pointer
        out = StructR{};
        uint64_t w, target;
        uint32_t tseg;
        bool ok;
        if (!resolve(seg, at, w, tseg, target, ok)) return ok;
        if ((w & 3) != 0) return m.fail("expected a struct pointer");
        const uint16_t dw = (uint16_t)(w >> 32), pw = (uint16_t)(w >> 48);
        if (target + dw + pw > m.segs[tseg].n) return m.fail("struct out of bounds");
        out.m = &m;
        out.seg = tseg;
        out.data = target;
        out.data_bits = (uint32_t)dw * 64;
        out.ptrs = target + dw;
        out.n_ptrs = pw;
        return true;
    }
    bool get_list(uint32_t seg, uint64_t at, ListR &out) {
        out = ListR{};
        uint64_t w, target;
        uint32_t tseg;
        bool ok;
        if (!resolve(seg, at, w, tseg, target, ok)) return ok;
        if ((w & 3) != 1) return m.fail("expected a list pointer");
        out.m = &m;
        out.seg = tseg;
        out.elem = (unsigned)((w >> 32) & 7);
        const uint64_t cnt = w >> 35;
        if (out.elem == 7) {
            if (target + 1 + cnt > m.segs[tseg].n) return m.fail("list out of bounds");
            const uint64_t tag = m.segs[tseg].p[target];
            if ((tag & 3) != 0) return m.fail("bad composite list tag");
            out.count = (uint32_t)tag >> 2;
            out.data_words = (uint16_t)(tag >> 32);
            out.n_ptrs = (uint16_t)(tag >> 48);
            if ((uint64_t)out.count * ((uint64_t)out.data_words + out.n_ptrs) > cnt) return m.fail("composite list larger than its pointer says");
            out.at = target + 1;
        } else {
            static const unsigned bits[7] = {0, 1, 8, 16, 32, 64, 64};
            const uint64_t words = (cnt * bits[out.elem] + 63) / 64;
            if (target + words > m.segs[tseg].n) return m.fail("list out of bounds");
            out.count = (uint32_t)cnt;
            out.at = target;
        }
        return true;
    }
    // pointer field i of a struct
    bool field_struct(const StructR &s, unsigned i, StructR &out) {
        out = StructR{};
        if (s.null() || i >= s.n_ptrs) return true;
        return get_struct(s.seg, s.ptrs + i, out);
    }
    bool field_list(const StructR &s, unsigned i, ListR &out) {
        out = ListR{};
        if (s.null() || i >= s.n_ptrs) return true;
        return get_list(s.seg, s.ptrs + i, out);
    }
    bool field_bytes(const StructR &s, unsigned i, bool text, std::string &out, bool *present = nullptr) {
        out.clear();
        ListR l;
        if (!field_list(s, i, l)) return false;
        if (present) *present = l.m != nullptr;
        if (!l.m) return true;
        if (l.elem != 2) return m.fail("expected a byte list");
        size_t n = l.count;
        const char *p = (const char *)(m.segs[l.seg].p + l.at);
        if (text) {
            if (n == 0 || p[n - 1] != 0) return m.fail("text without its terminator");
            --n;
            if (!valid_utf8((const unsigned char *)p, n)) return m.fail("text is not UTF-8"); // (the capnp crate's text::Reader is a &str)
        }
        out.assign(p, n);
        return true;
    }
    // A list about to be read as List(struct): every encoding but a bit list can be read that way (the capnp runtime upgrades
    // void, byte, 2- / 4- / 8-byte, pointer and composite lists alike; fields beyond the element read as defaults), and elements
    // that occupy no words at all (void, or composite with an empty struct) are charged one word each against a traversal budget, as the capnp runtime does: a
    // 24-byte message must not be able to claim 2^29 elements and have the reader allocate for them.
    static constexpr uint64_t TRAVERSAL_WORDS = 8ull << 20; // capnp's default ReaderOptions::traversal_limit_in_words
    bool struct_list(const ListR &l) {
        if (!l.m) return true;
        if (l.elem == 1) return m.fail("list of structs stored as a bit list"); // (the one upgrade the capnp runtime refuses)
        if (l.elem == 0 || (l.elem == 7 && (uint32_t)l.data_words + l.n_ptrs == 0)) {
            amplified += l.count;
            if (amplified > TRAVERSAL_WORDS) return m.fail("read limit exceeded (zero-sized list elements)");
        }
        return true;
    }
    uint64_t amplified = 0;
    // element i of a List(struct); lists of primitives / pointers upgraded to structs read through the same view
    bool element(const ListR &l, uint32_t i, StructR &out) {
        out = StructR{};
        out.m = &m;
        out.seg = l.seg;
        if (l.elem == 7) {
            out.data = l.at + (uint64_t)i * ((uint64_t)l.data_words + l.n_ptrs);
            out.data_bits = (uint32_t)l.data_words * 64;
            out.ptrs = out.data + l.data_words;
            out.n_ptrs = l.n_ptrs;
        } else if (l.elem == 6) {
            out.ptrs = l.at + i;
            out.n_ptrs = 1;
        } else if (l.elem == 5) {
            out.data = l.at + i;
            out.data_bits = 64;
        } else if (l.elem == 0) { // void elements: every field reads as its default
        } else if (l.elem >= 2 && l.elem <= 4) {
            // byte / 2-byte / 4-byte elements upgraded to structs (the capnp runtime allows it): the element is the start of
            // the data section, fields beyond it read as defaults
            const uint32_t eb = 1u << (l.elem - 2);
            const uint64_t off = (uint64_t)i * eb;
            out.data = l.at + off / 8;
            out.byte_shift = (uint8_t)(off & 7u);
            out.data_bits = 8 * eb;
        } else {
            return m.fail("list of structs stored as a bit list");
        }
        return true;
    }
    bool u64_list(const ListR &l, std::vector<uint64_t> &out) {
        out.clear();
        if (!l.m) return true;
        if (l.elem != 5) return m.fail("expected a list of 64-bit values");
        out.assign(m.segs[l.seg].p + l.at, m.segs[l.seg].p + l.at + l.count);
        return true;
    }
    bool u32_list(const ListR &l, std::vector<uint32_t> &out) {
        out.clear();
        if (!l.m) return true;
        if (l.elem != 4) return m.fail("expected a list of 32-bit values");
        out.resize(l.count);
        memcpy(out.data(), m.segs[l.seg].p + l.at, (size_t)l.count * 4);
        return true;
    }
};

} // namespace capnp

// =====================================================================================================================
// .bsk -- finch.capnp
// =====================================================================================================================
// Struct sizes (finch_capnp.rs): FilterParams {data 4, ptrs 0} :201, SketchParams {5, 0} :398, KmerCount {2, 2} :591,
// Sketch {2, 5} :844, Multisketch {0, 1} :979.
static int write_bsk(const std::vector<Sketch> &sketches, std::string &out) {
    capnp::Builder b;
    const size_t root = b.alloc(1);
    const size_t ms = b.alloc(1); // Multisketch: pointer 0 = sketches
    b.struct_ptr(root, ms, 0, 1);
    if (sketches.size() > (1u << 28)) return hfail(FH_ERR_UNSUPPORTED, "too many sketches for one message");
    const size_t list = b.struct_list(ms, (uint32_t)sketches.size(), 2, 5);
    for (size_t i = 0; i < sketches.size() && !b.too_big; ++i) {
        const Sketch &s = sketches[i];
        const size_t at = list + 7 * i, ptrs = at + 2;
        // Sketch (finch_capnp.rs:643-690): data[0] seqLength, data[1] numValidKmers; pointers: 0 name, 1 comment, 2 hashes,
        // 3 filterParams, 4 sketchParams.  Same call order as write_finch_file (mod.rs:130-162).
        b.text(ptrs + 0, s.name);
        b.w[at + 0] = s.seq_length;
        b.w[at + 1] = s.num_valid_kmers;
        b.text(ptrs + 1, s.comment);
        if (s.hashes.size() > (1u << 27)) return hfail(FH_ERR_UNSUPPORTED, "sketch of %zu hashes does not fit one message", s.hashes.size());
        const size_t hl = b.struct_list(ptrs + 2, (uint32_t)s.hashes.size(), 2, 2);
        for (size_t j = 0; j < s.hashes.size() && !b.too_big; ++j) {
            // KmerCount (finch_capnp.rs:450-473): u64 field 0 hash; u32 field 2 count, u32 field 3 extraCount (= the two
            // halves of data word 1); pointers: 0 kmer, 1 label
            const KmerCount &h = s.hashes[j];
            const size_t e = hl + 4 * j;
            b.w[e] = h.hash;
            b.bytes(e + 2, h.kmer.data(), h.kmer.size(), false);
            b.w[e + 1] = (uint64_t)h.count | ((uint64_t)h.extra_count << 32);
            if (h.label) b.data(e + 3, *h.label);
        }
        // FilterParams (finch_capnp.rs:80-97): bit 0 filtered, u32 field 1 lowAbunFilter, u32 field 2 highAbunFilter, f64
        // field 2 errFilter, f64 field 3 strandFilter; values as mod.rs:150-156 sets them
        const finch_filter_params &fp = s.filter_params;
        const size_t f = b.alloc(4);
        if (b.too_big) break;
        b.struct_ptr(ptrs + 3, f, 4, 0);
        b.w[f + 0] = (uint64_t)(fp.filter_on == 1 ? 1u : 0u) | ((uint64_t)(fp.has_abun_lo ? fp.abun_lo : 0u) << 32);
        b.w[f + 1] = (uint64_t)(fp.has_abun_hi ? fp.abun_hi : UINT32_MAX);
        b.w[f + 2] = capnp::f64_bits(fp.err_filter);
        b.w[f + 3] = capnp::f64_bits(fp.strand_filter);
        // SketchParams (finch_capnp.rs:253-278): u16 field 0 sketchMethod, u8 field 2 kmerLength, u64 fields 1..3
        // kmersToSketch / hashSeed / finalSize, bit 24 noStrict, f64 field 4 scale; set_sketch_params (mod.rs:57-92) writes
        // only the fields of the variant at hand
        const finch_sketch_params &sp = s.sketch_params;
        const size_t p = b.alloc(5);
        if (b.too_big) break;
        b.struct_ptr(ptrs + 4, p, 5, 0);
        const uint64_t method = sp.kind == 0 ? 0u : sp.kind == 1 ? 1u : 2u; // murmurHash3 / murmurHash3Scaled / none
        b.w[p + 0] = method | ((uint64_t)(uint8_t)sp.kmer_length << 16);
        if (sp.kind == 0) {
            b.w[p + 0] |= (uint64_t)(sp.no_strict ? 1u : 0u) << 24;
            b.w[p + 1] = sp.kmers_to_sketch;
            b.w[p + 2] = sp.hash_seed;
            b.w[p + 3] = sp.final_size;
        } else if (sp.kind == 1) {
            b.w[p + 1] = sp.kmers_to_sketch;
            b.w[p + 2] = sp.hash_seed;
            b.w[p + 4] = capnp::f64_bits(sp.scale);
        }
    }
    if (b.too_big) return hfail(FH_ERR_UNSUPPORTED, "sketches exceed the 4 GiB a single-segment message can address");
    b.frame(out);
    return FH_OK;
}

static int read_bsk(const uint8_t *data, uint64_t len, std::vector<Sketch> &out) {
    capnp::Message m;
    capnp::Walker w(m);
    auto bad = [&]() { return hfail(FH_ERR_INVALID, "not a valid .bsk message: %s", m.err.empty() ? "malformed" : m.err.c_str()); };
    if (!m.parse(data, len)) return bad();
    capnp::StructR root;
    if (!w.get_struct(0, 0, root)) return bad();
    capnp::ListR sl;
    if (!w.field_list(root, 0, sl) || !w.struct_list(sl)) return bad();
    out.clear();
    out.resize(sl.m ? sl.count : 0);
    for (uint32_t i = 0; i < out.size(); ++i) {
        capnp::StructR cs;
        if (!w.element(sl, i, cs)) return bad();
        Sketch &s = out[i];
        if (!w.field_bytes(cs, 0, true, s.name) || !w.field_bytes(cs, 1, true, s.comment)) return bad();
        s.seq_length = cs.word(0);
        s.num_valid_kmers = cs.word(1);
        capnp::ListR hl;
        if (!w.field_list(cs, 2, hl) || !w.struct_list(hl)) return bad();
        s.hashes.resize(hl.m ? hl.count : 0);
        for (uint32_t j = 0; j < s.hashes.size(); ++j) {
            capnp::StructR ch;
            if (!w.element(hl, j, ch)) return bad();
            KmerCount &h = s.hashes[j];
            h.hash = ch.word(0);
            h.count = ch.u32(2);
            h.extra_count = ch.u32(3);
            std::string kmer_bytes;
            std::string label_bytes;
            bool has_label = false;
            if (!w.field_bytes(ch, 0, false, kmer_bytes) || !w.field_bytes(ch, 1, false, label_bytes, &has_label)) return bad();
            if (has_label) h.label = std::make_shared<const std::string>(std::move(label_bytes));
            h.kmer = kmer_bytes;
        }
        capnp::StructR sp, fp;
        if (!w.field_struct(cs, 4, sp) || !w.field_struct(cs, 3, fp)) return bad();
        // get_sketch_params (mod.rs:94-121)
        finch_sketch_params &q = s.sketch_params;
        memset(&q, 0, sizeof q);
        const uint16_t method = sp.u16(0);
        if (method > 2) return hfail(FH_ERR_INVALID, "sketch method %u is not in the schema", method);
        q.kind = method;
        q.kmer_length = sp.u8(2);
        if (method == 0) {
            q.kmers_to_sketch = sp.word(1);
            q.hash_seed = sp.word(2);
            q.final_size = sp.word(3);
            q.no_strict = sp.bit(24);
        } else if (method == 1) {
            q.kmers_to_sketch = sp.word(1);
            q.hash_seed = sp.word(2);
            q.scale = capnp::bits_f64(sp.word(4));
        }
        // mod.rs:196-210: 0 / u32::MAX stand for "no bound"
        finch_filter_params &f = s.filter_params;
        memset(&f, 0, sizeof f);
        f.filter_on = fp.bit(0) ? 1 : 0;
        const uint32_t lo = fp.u32(1), hi = fp.u32(2);
        f.has_abun_lo = lo != 0;
        f.abun_lo = lo;
        f.has_abun_hi = hi != UINT32_MAX;
        f.abun_hi = f.has_abun_hi ? hi : 0;
        f.err_filter = capnp::bits_f64(fp.word(2));
        f.strand_filter = capnp::bits_f64(fp.word(3));
    }
    return FH_OK;
}

// =====================================================================================================================
// .msh -- mash.capnp
// =====================================================================================================================
// MinHash {data 3, ptrs 4} (mash_capnp.rs:307): u32 field 0 kmerSize, 1 windowSize, 2 minHashesPerWindow; bits 96 concatenated,
// 97 noncanonical, 98 preserveCase; f32 field 4 error; u32 field 5 hashSeed XOR 42 (its schema default); pointers 0
// referenceListOld, 1 locusList, 2 alphabet, 3 referenceList (mash_capnp.rs:53-107).  ReferenceList {0, 1} :441.
// Reference {data 3, ptrs 7} :743: u32 field 0 length, u64 field 1 length64, u64 field 2 numValidKmers; pointers 0 sequence,
// 1 quality, 2 name, 3 comment, 4 hashes32, 5 hashes64, 6 counts32 (mash_capnp.rs:492-550).
static int write_msh(const std::vector<Sketch> &sketches, std::string &out) {
    if (int rc = check_compatible(sketches)) return rc; // SketchParams::from_sketches (mash.rs:13)
    const finch_sketch_params &sp = sketches[0].sketch_params;
    capnp::Builder b;
    const size_t root = b.alloc(1);
    const size_t mh = b.alloc(3 + 4);
    b.struct_ptr(root, mh, 3, 4);
    size_t largest = 0; // mash.rs:25: max hashes.len()  (unwrap_or(1) only covers an empty list, which from_sketches rejects)
    for (const Sketch &s : sketches) largest = std::max(largest, s.hashes.size());
    const uint32_t k = sp.kmer_length;
    const uint32_t seed32 = sp.kind == 2 ? 0u : (uint32_t)sp.hash_seed; // hash_info().2 as u32 (mash.rs:19)
    // setters in the order of mash.rs:18-28 (plain stores, no allocation, except the alphabet text)
    b.w[mh + 0] = (uint64_t)k | ((uint64_t)k << 32);                    // kmerSize, windowSize
    b.w[mh + 1] = (uint64_t)(uint32_t)largest | ((uint64_t)1u << 32);   // minHashesPerWindow; concatenated = bit 96; noncanonical,
                                                                        // preserveCase (bits 97, 98) false
    b.w[mh + 2] = (uint64_t)0u /* error 0.0f */ | ((uint64_t)(seed32 ^ 42u) << 32);
    b.text(mh + 3 + 2, "ACGT");
    const size_t rl = b.alloc(1); // ReferenceList
    b.struct_ptr(mh + 3 + 3, rl, 0, 1);
    const size_t refs = b.struct_list(rl, (uint32_t)sketches.size(), 3, 7);
    for (size_t i = 0; i < sketches.size() && !b.too_big; ++i) {
        const Sketch &s = sketches[i];
        const size_t at = refs + 10 * i, ptrs = at + 3;
        if (s.hashes.size() > (1u << 28)) return hfail(FH_ERR_UNSUPPORTED, "sketch of %zu hashes does not fit one message", s.hashes.size());
        b.text(ptrs + 2, s.name);
        b.text(ptrs + 3, s.comment);
        b.w[at + 1] = s.seq_length;       // length64
        b.w[at + 2] = s.num_valid_kmers;  // numValidKmers
        const uint32_t n = (uint32_t)s.hashes.size();
        const size_t h = b.alloc(n);
        if (b.too_big) break;
        for (uint32_t j = 0; j < n; ++j) b.w[h + j] = s.hashes[j].hash;
        b.list_ptr(ptrs + 5, h, 5, n);
        const size_t c = b.alloc(((size_t)n + 1) / 2);
        if (b.too_big) break;
        uint32_t *cp = (uint32_t *)(b.w.data() + c);
        for (uint32_t j = 0; j < n; ++j) cp[j] = s.hashes[j].count;
        b.list_ptr(ptrs + 6, c, 4, n);
    }
    if (b.too_big) return hfail(FH_ERR_UNSUPPORTED, "sketches exceed the 4 GiB a single-segment message can address");
    b.frame(out);
    return FH_OK;
}

static int read_msh(const uint8_t *data, uint64_t len, std::vector<Sketch> &out) {
    capnp::Message m;
    capnp::Walker w(m);
    auto bad = [&]() { return hfail(FH_ERR_INVALID, "not a valid .msh message: %s", m.err.empty() ? "malformed" : m.err.c_str()); };
    if (!m.parse(data, len)) return bad();
    capnp::StructR mh;
    if (!w.get_struct(0, 0, mh)) return bad();
    // mash.rs:66-74: Mash{kmers_to_sketch 0, final_size 0, no_strict true, seed, k}
    finch_sketch_params sp;
    memset(&sp, 0, sizeof sp);
    sp.kind = 0;
    sp.no_strict = 1;
    sp.hash_seed = mh.u32(5) ^ 42u;
    sp.kmer_length = (uint8_t)mh.u32(0);
    finch_filter_params fp; // FilterParams::default()
    memset(&fp, 0, sizeof fp);
    capnp::StructR rl_new, rl_old;
    if (!w.field_struct(mh, 3, rl_new) || !w.field_struct(mh, 0, rl_old)) return bad();
    capnp::ListR refs;
    if (!w.field_list(rl_new, 0, refs)) return bad();
    if (!refs.m) // mash.rs:85-89: has_references() ? new : old
        if (!w.field_list(rl_old, 0, refs)) return bad();
    if (!w.struct_list(refs)) return bad();
    out.clear();
    out.resize(refs.m ? refs.count : 0);
    for (uint32_t i = 0; i < out.size(); ++i) {
        capnp::StructR r;
        if (!w.element(refs, i, r)) return bad();
        Sketch &s = out[i];
        capnp::ListR hl, cl;
        std::vector<uint64_t> hs;
        std::vector<uint32_t> cs;
        if (!w.field_list(r, 5, hl) || !w.field_list(r, 6, cl) || !w.u64_list(hl, hs) || !w.u32_list(cl, cs)) return bad();
        s.hashes.resize(hs.size());
        // mash.rs:95-121: no counts -> (1, 0); else zip(hashes, counts) -> (c, c / 2)
        if (cs.empty()) {
            for (size_t j = 0; j < hs.size(); ++j) s.hashes[j] = KmerCount{hs[j], KmerBytes(), 1u, 0u};
        } else {
            s.hashes.resize(std::min(hs.size(), cs.size()));
            for (size_t j = 0; j < s.hashes.size(); ++j) s.hashes[j] = KmerCount{hs[j], KmerBytes(), cs[j], cs[j] / 2};
        }
        if (!w.field_bytes(r, 2, true, s.name) || !w.field_bytes(r, 3, true, s.comment)) return bad();
        s.seq_length = r.word(1);
        s.num_valid_kmers = r.word(2);
        s.sketch_params = sp;
        s.filter_params = fp;
    }
    return FH_OK;
}

// =====================================================================================================================
// .sk reader -- MultiSketch / JsonSketch deserialisation (json.rs:92-139, 160-262) with serde_json's rules for the field
// types at hand: unknown keys ignored, Option fields may be missing or null, everything else must be there
// =====================================================================================================================
namespace json {

struct Value {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    std::string s; // Str: the decoded text; Num: the literal
    std::vector<Value> a;
    std::vector<std::pair<std::string, Value>> o;
    const Value *get(const char *key) const { // serde: the last duplicate wins is an error there; first match is enough here
        for (const auto &kv : o)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct Parser {
    const char *p, *end;
    std::string err;
    int depth = 0;
    bool fail(const char *what) {
        if (err.empty()) err = what;
        return false;
    }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    static void utf8(std::string &o, uint32_t c) {
        if (c < 0x80) o.push_back((char)c);
        else if (c < 0x800) {
            o.push_back((char)(0xC0 | (c >> 6)));
            o.push_back((char)(0x80 | (c & 0x3F)));
        } else if (c < 0x10000) {
            o.push_back((char)(0xE0 | (c >> 12)));
            o.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
            o.push_back((char)(0x80 | (c & 0x3F)));
        } else {
            o.push_back((char)(0xF0 | (c >> 18)));
            o.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
            o.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
            o.push_back((char)(0x80 | (c & 0x3F)));
        }
    }
    bool hex4(uint32_t &v) {
        if (end - p < 4) return fail("truncated \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string &o) {
        o.clear();
        ++p; // opening quote
        for (;;) {
            if (p >= end) return fail("unterminated string");
            const unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in a string");
            if (c != '\\') {
                o.push_back((char)c);
                continue;
            }
            if (p >= end) return fail("unterminated string");
            const char e = *p++;
            switch (e) {
            case '"': o.push_back('"'); break;
            case '\\': o.push_back('\\'); break;
            case '/': o.push_back('/'); break;
            case 'b': o.push_back('\b'); break;
            case 'f': o.push_back('\f'); break;
            case 'n': o.push_back('\n'); break;
            case 'r': o.push_back('\r'); break;
            case 't': o.push_back('\t'); break;
            case 'u': {
                uint32_t v;
                if (!hex4(v)) return false;
                if (v >= 0xD800 && v < 0xDC00) { // surrogate pair
                    uint32_t lo;
                    if (end - p < 2 || p[0] != '\\' || p[1] != 'u') return fail("lone surrogate");
                    p += 2;
                    if (!hex4(lo)) return false;
                    if (lo < 0xDC00 || lo > 0xDFFF) return fail("lone surrogate");
                    v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
                } else if (v >= 0xDC00 && v <= 0xDFFF) {
                    return fail("lone surrogate");
                }
                utf8(o, v);
                break;
            }
            default: return fail("bad escape");
            }
        }
    }
    bool value(Value &v) {
        ws();
        if (p >= end) return fail("unexpected end of input");
        if (++depth > 64) return fail("nesting too deep");
        bool ok = true;
        const char c = *p;
        if (c == '{') {
            v.kind = Value::Obj;
            ++p;
            ws();
            if (p < end && *p == '}') ++p;
            else
                for (;;) {
                    ws();
                    if (p >= end || *p != '"') { ok = fail("expected an object key"); break; }
                    std::string k;
                    if (!string(k)) { ok = false; break; }
                    ws();
                    if (p >= end || *p != ':') { ok = fail("expected ':'"); break; }
                    ++p;
                    v.o.emplace_back(std::move(k), Value());
                    if (!value(v.o.back().second)) { ok = false; break; }
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == '}') { ++p; break; }
                    ok = fail("expected ',' or '}'");
                    break;
                }
        } else if (c == '[') {
            v.kind = Value::Arr;
            ++p;
            ws();
            if (p < end && *p == ']') ++p;
            else
                for (;;) {
                    v.a.emplace_back();
                    if (!value(v.a.back())) { ok = false; break; }
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; break; }
                    ok = fail("expected ',' or ']'");
                    break;
                }
        } else if (c == '"') {
            v.kind = Value::Str;
            ok = string(v.s);
        } else if (c == 't' && end - p >= 4 && !memcmp(p, "true", 4)) {
            v.kind = Value::Bool;
            v.b = true;
            p += 4;
        } else if (c == 'f' && end - p >= 5 && !memcmp(p, "false", 5)) {
            v.kind = Value::Bool;
            p += 5;
        } else if (c == 'n' && end - p >= 4 && !memcmp(p, "null", 4)) {
            p += 4;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            v.kind = Value::Num;
            const char *q = p;
            if (*q == '-') ++q;
            const char *digits = q;
            while (q < end && *q >= '0' && *q <= '9') ++q;
            if (q == digits || (q - digits > 1 && *digits == '0')) ok = fail("bad number");
            if (ok && q < end && *q == '.') {
                const char *f = ++q;
                while (q < end && *q >= '0' && *q <= '9') ++q;
                if (q == f) ok = fail("bad number");
            }
            if (ok && q < end && (*q == 'e' || *q == 'E')) {
                ++q;
                if (q < end && (*q == '+' || *q == '-')) ++q;
                const char *x = q;
                while (q < end && *q >= '0' && *q <= '9') ++q;
                if (q == x) ok = fail("bad number");
            }
            v.s.assign(p, q);
            p = q;
        } else {
            ok = fail("unexpected character");
        }
        --depth;
        return ok;
    }
};

// an unsigned integer JSON number that fits `max` (what serde accepts for u8 / u16 / u32 / u64 fields)
static bool as_uint(const Value *v, uint64_t max, uint64_t &out) {
    if (!v || v->kind != Value::Num || v->s.empty() || v->s[0] == '-') return false;
    if (v->s.find_first_of(".eE") != std::string::npos) return false;
    errno = 0;
    char *e = nullptr;
    const unsigned long long x = strtoull(v->s.c_str(), &e, 10);
    if (errno || *e || x > max) return false;
    out = x;
    return true;
}

} // namespace json

// str::parse::<u32 / u64>: optional '+', decimal digits only
static bool rust_parse_uint(const std::string &s, uint64_t max, uint64_t &out) {
    size_t i = 0;
    if (i < s.size() && s[i] == '+') ++i;
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (max - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

// str::parse::<f64>: decimal / exponent forms, "inf", "infinity", "nan" (any case), optional sign; no surrounding blanks
static bool rust_parse_f64(const std::string &s, double &out) {
    if (s.empty() || s[0] == ' ' || s.find_first_of("xXpP") != std::string::npos) return false;
    char *e = nullptr;
    errno = 0;
    const double v = strtod(s.c_str(), &e);
    if (e == s.c_str() || *e) return false;
    out = v;
    return true;
}

// FilterParams::from_serialized (filtering.rs:110-134)
static int filters_from_map(const json::Value *f, finch_filter_params &out) {
    memset(&out, 0, sizeof out);
    size_t n = 0;
    auto get = [&](const char *k) -> const std::string * {
        if (!f) return nullptr;
        const json::Value *v = f->get(k);
        return v ? &v->s : nullptr;
    };
    if (f) n = f->o.size();
    uint64_t u;
    if (const std::string *s = get("minCopies")) {
        if (!rust_parse_uint(*s, UINT32_MAX, u)) return hfail(FH_ERR_INVALID, "invalid digit found in string");
        out.has_abun_lo = 1;
        out.abun_lo = (uint32_t)u;
    }
    if (const std::string *s = get("maxCopies")) {
        if (!rust_parse_uint(*s, UINT32_MAX, u)) return hfail(FH_ERR_INVALID, "invalid digit found in string");
        out.has_abun_hi = 1;
        out.abun_hi = (uint32_t)u;
    }
    out.filter_on = n ? 1 : 0;
    if (const std::string *s = get("errFilter"))
        if (!rust_parse_f64(*s, out.err_filter)) return hfail(FH_ERR_INVALID, "invalid float literal");
    if (const std::string *s = get("strandFilter"))
        if (!rust_parse_f64(*s, out.strand_filter)) return hfail(FH_ERR_INVALID, "invalid float literal");
    return FH_OK;
}

static int read_sk(const uint8_t *data, uint64_t len, std::vector<Sketch> &out) {
    json::Parser ps{(const char *)data, (const char *)data + len, std::string()};
    json::Value root;
    auto bad = [&](const char *what) { return hfail(FH_ERR_INVALID, "Error parsing sketch JSON: %s", what); };
    if (!ps.value(root)) return bad(ps.err.c_str());
    ps.ws();
    if (ps.p != ps.end) return bad("trailing characters");
    if (root.kind != json::Value::Obj) return bad("expected an object");
    // struct MultiSketch (json.rs:141-158): every field but `scale` is required
    uint64_t kmer, sketch_size, hash_bits, hash_seed;
    if (!json::as_uint(root.get("kmer"), 255, kmer)) return bad("kmer");
    const json::Value *alphabet = root.get("alphabet"), *pc = root.get("preserveCase"), *canon = root.get("canonical");
    if (!alphabet || alphabet->kind != json::Value::Str) return bad("alphabet");
    if (!pc || pc->kind != json::Value::Bool) return bad("preserveCase");
    if (!canon || canon->kind != json::Value::Bool) return bad("canonical");
    if (!json::as_uint(root.get("sketchSize"), UINT32_MAX, sketch_size)) return bad("sketchSize");
    const json::Value *ht = root.get("hashType");
    if (!ht || ht->kind != json::Value::Str) return bad("hashType");
    if (!json::as_uint(root.get("hashBits"), 65535, hash_bits)) return bad("hashBits");
    if (!json::as_uint(root.get("hashSeed"), UINT64_MAX, hash_seed)) return bad("hashSeed");
    const json::Value *sc = root.get("scale");
    bool has_scale = false;
    double scale = 0.0;
    if (sc && sc->kind != json::Value::Null) {
        if (sc->kind != json::Value::Num) return bad("scale");
        scale = strtod(sc->s.c_str(), nullptr);
        has_scale = true;
    }
    const json::Value *sks = root.get("sketches");
    if (!sks || sks->kind != json::Value::Arr) return bad("sketches");
    std::vector<Sketch> res(sks->a.size());
    for (size_t i = 0; i < res.size(); ++i) {
        const json::Value &js = sks->a[i];
        if (js.kind != json::Value::Obj) return bad("sketch");
        Sketch &s = res[i];
        const json::Value *name = js.get("name");
        if (!name || name->kind != json::Value::Str) return bad("sketch name");
        s.name = name->s;
        auto opt_u64 = [&](const char *k, uint64_t &v) { // Option<u64>: missing / null -> unwrap_or(0) in to_sketches
            const json::Value *x = js.get(k);
            v = 0;
            return !x || x->kind == json::Value::Null || json::as_uint(x, UINT64_MAX, v);
        };
        if (!opt_u64("seqLength", s.seq_length)) return bad("seqLength");
        if (!opt_u64("numValidKmers", s.num_valid_kmers)) return bad("numValidKmers");
        if (const json::Value *c = js.get("comment")) {
            if (c->kind == json::Value::Str) s.comment = c->s;
            else if (c->kind != json::Value::Null) return bad("comment");
        }
        const json::Value *filters = js.get("filters");
        if (filters && filters->kind == json::Value::Null) filters = nullptr;
        if (filters) {
            if (filters->kind != json::Value::Obj) return bad("filters");
            for (const auto &kv : filters->o)
                if (kv.second.kind != json::Value::Str) return bad("filters");
        }
        const json::Value *hashes = js.get("hashes"), *kmers = js.get("kmers"), *counts = js.get("counts");
        if (!hashes || hashes->kind != json::Value::Arr) return bad("hashes");
        if (kmers && kmers->kind == json::Value::Null) kmers = nullptr;
        if (counts && counts->kind == json::Value::Null) counts = nullptr;
        if (kmers && kmers->kind != json::Value::Arr) return bad("kmers");
        if (counts && counts->kind != json::Value::Arr) return bad("counts");
        const size_t n = hashes->a.size();
        // (the reference indexes kmers[i] / counts[i] and panics on a short array: an error here)
        if ((kmers && kmers->a.size() < n) || (counts && counts->a.size() < n)) return bad("kmers / counts shorter than hashes");
        s.hashes.resize(n);
        for (size_t j = 0; j < n; ++j) {
            KmerCount &h = s.hashes[j];
            uint64_t hv, cv = 1;
            // QuotedU64 (json.rs:264-292): a string holding a u64
            if (hashes->a[j].kind != json::Value::Str || !rust_parse_uint(hashes->a[j].s, UINT64_MAX, hv)) return bad("hash");
            if (kmers) {
                if (kmers->a[j].kind != json::Value::Str) return bad("kmer");
                h.kmer = kmers->a[j].s;
            }
            if (counts && !json::as_uint(&counts->a[j], UINT32_MAX, cv)) return bad("count");
            h.hash = hv;
            h.count = (uint32_t)cv;
            h.extra_count = (uint32_t)cv / 2; // json.rs:124
        }
        if (int rc = filters_from_map(filters, s.filter_params)) return rc;
        // MultiSketch::get_params (json.rs:160-201)
        finch_sketch_params &q = s.sketch_params;
        memset(&q, 0, sizeof q);
        q.kmer_length = (uint32_t)kmer;
        if (ht->s == "MurmurHash3_x64_128") {
            if (hash_bits != 64) return hfail(FH_ERR_INVALID, "Multisketch has incompatible hash size (%llu != 64)", (unsigned long long)hash_bits);
            q.hash_seed = hash_seed;
            q.kmers_to_sketch = sketch_size;
            if (!has_scale) {
                q.kind = 0;
                q.final_size = sketch_size;
                q.no_strict = 1;
            } else {
                q.kind = 1;
                q.scale = scale;
            }
        } else if (ht->s == "None") {
            q.kind = 2;
        } else {
            return hfail(FH_ERR_INVALID, "%s sketch type is not supported", ht->s.c_str());
        }
    }
    out.swap(res);
    return FH_OK;
}

static bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

static int read_whole_file(const char *path, std::string &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return hfail(FH_ERR_INVALID, "Error opening \"%s\"", path); // lib.rs:102
    char buf[1 << 16];
    size_t g;
    out.clear();
    while ((g = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, g);
    const bool err = ferror(f) != 0;
    fclose(f);
    if (err) return hfail(FH_ERR_INVALID, "Error reading \"%s\"", path);
    return FH_OK;
}

static std::string file_name_of(const std::string &path) {
    const size_t s = path.find_last_of('/');
    return s == std::string::npos ? path : path.substr(s + 1);
}

static int bytes_out(const std::string &s, uint8_t **out, uint64_t *len) {
    uint8_t *p = (uint8_t *)malloc(s.size() ? s.size() : 1);
    if (!p) return hfail(FH_ERR_INVALID, "out of memory");
    memcpy(p, s.data(), s.size());
    *out = p;
    if (len) *len = s.size();
    return FH_OK;
}

} // namespace finch

using namespace finch;

extern "C" {

int finch_sketches_to_bsk(const finch_sketches *s, uint8_t **out, uint64_t *len) try {
    if (!s || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::string o;
    if (int rc = write_bsk(s->v, o)) return rc;
    return bytes_out(o, out, len);
} FINCH_CATCH

int finch_sketches_to_msh(const finch_sketches *s, uint8_t **out, uint64_t *len) try {
    if (!s || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::string o;
    if (int rc = write_msh(s->v, o)) return rc;
    return bytes_out(o, out, len);
} FINCH_CATCH

void finch_free_bytes(uint8_t *p) { free(p); }

static int wrap(std::vector<Sketch> &v, finch_sketches **out) {
    auto res = std::make_unique<finch_sketches>();
    res->v.swap(v);
    *out = res.release();
    return FH_OK;
}

int finch_sketches_from_bsk(const uint8_t *data, uint64_t len, finch_sketches **out) try {
    if ((!data && len) || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::vector<Sketch> v;
    if (int rc = read_bsk(data, len, v)) return rc;
    return wrap(v, out);
} FINCH_CATCH

int finch_sketches_from_msh(const uint8_t *data, uint64_t len, finch_sketches **out) try {
    if ((!data && len) || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::vector<Sketch> v;
    if (int rc = read_msh(data, len, v)) return rc;
    return wrap(v, out);
} FINCH_CATCH

int finch_sketches_from_json(const uint8_t *data, uint64_t len, finch_sketches **out) try {
    if ((!data && len) || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::vector<Sketch> v;
    if (int rc = read_sk(data, len, v)) return rc;
    return wrap(v, out);
} FINCH_CATCH

// open_sketch_file (lib.rs:96-118): the format is taken from the file name
int finch_open_sketch_file(const char *path, finch_sketches **out) try {
    if (!path || !out) return hfail(FH_ERR_INVALID, "null argument");
    const std::string fn = file_name_of(path);
    if (fn.empty()) return hfail(FH_ERR_INVALID, "Path does not have a filename: \"%s\"", path);
    const bool msh = ends_with(fn, ".msh"), bsk = ends_with(fn, ".bsk"), sk = ends_with(fn, ".sk") || ends_with(fn, ".json");
    std::string bytes;
    if (int rc = read_whole_file(path, bytes)) return rc; // (the reference opens the file before it looks at the suffix)
    if (!msh && !bsk && !sk) return hfail(FH_ERR_INVALID, "File suffix is not *.bsk, *.msh, or *.sk");
    std::vector<Sketch> v;
    int rc;
    if (msh) rc = read_msh((const uint8_t *)bytes.data(), bytes.size(), v);
    else if (bsk) rc = read_bsk((const uint8_t *)bytes.data(), bytes.size(), v);
    else {
        rc = read_sk((const uint8_t *)bytes.data(), bytes.size(), v);
        if (rc != FH_OK && g_host_err.rfind("Error parsing sketch JSON", 0) == 0) {
            const std::string why = g_host_err;
            rc = hfail(FH_ERR_INVALID, "Error parsing \"%s\" (%s)", path, why.c_str()); // lib.rs:112
        }
    }
    if (rc != FH_OK) return rc;
    return wrap(v, out);
} FINCH_CATCH

// the `sketch` subcommand's output step (cli/src/main.rs:53-70, 225-231): binary / Mash / JSON by file name
int finch_write_sketch_file(const finch_sketches *s, const char *path) try {
    if (!s || !path) return hfail(FH_ERR_INVALID, "null argument");
    const std::string fn = file_name_of(path);
    std::string o;
    if (ends_with(fn, ".bsk")) {
        if (int rc = write_bsk(s->v, o)) return rc;
    } else if (ends_with(fn, ".msh")) {
        if (int rc = write_msh(s->v, o)) return rc;
    } else if (ends_with(fn, ".sk") || ends_with(fn, ".json")) {
        char *js = nullptr;
        uint64_t n = 0;
        if (int rc = finch_sketches_to_json(s, &js, &n)) return rc;
        o.assign(js, n);
        finch_free_string(js);
    } else {
        return hfail(FH_ERR_INVALID, "File suffix is not *.bsk, *.msh, or *.sk");
    }
    FILE *f = fopen(path, "wb");
    if (!f) return hfail(FH_ERR_INVALID, "%s: %s (os error %d)", path, strerror(errno), errno);
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    if (fclose(f) != 0 || !ok) return hfail(FH_ERR_INVALID, "%s: write failed", path);
    return FH_OK;
} FINCH_CATCH

int finch_sketch_params_of(const finch_sketches *s, uint32_t i, finch_sketch_params *out) try {
    if (!s || i >= s->v.size() || !out) return hfail(FH_ERR_INVALID, "bad argument");
    *out = s->v[i].sketch_params;
    return FH_OK;
} FINCH_CATCH

const char *finch_sketch_comment(const finch_sketches *s, uint32_t i) { return (s && i < s->v.size()) ? s->v[i].comment.c_str() : ""; }

int finch_sketch_set_comment(finch_sketches *s, uint32_t i, const char *comment) try {
    if (!s || i >= s->v.size()) return hfail(FH_ERR_INVALID, "bad argument");
    s->v[i].comment = comment ? comment : "";
    return FH_OK;
} FINCH_CATCH

// Vec<Sketch> concatenation (the CLI collects the sketches of all inputs before it writes one file, main.rs:60-70)
int finch_sketches_append(finch_sketches *dst, const finch_sketches *src) try {
    if (!dst || !src) return hfail(FH_ERR_INVALID, "null argument");
    dst->v.insert(dst->v.end(), src->v.begin(), src->v.end());
    return FH_OK;
} FINCH_CATCH

// FilterParams::filter_sketch (filtering.rs:20-54) as the reference has it: the sketch's filter parameters take the
// stricter of their own and `filters`' values; the hashes are NOT touched (the reference computes the filtered list
// and drops it, filtering.rs:24).
int finch_filter_sketch(finch_sketches *s, uint32_t i, const finch_filter_params *filters) try {
    if (!s || i >= s->v.size() || !filters) return hfail(FH_ERR_INVALID, "bad argument");
    finch_filter_params &fp = s->v[i].filter_params;
    const finch_filter_params &f = *filters;
    fp.filter_on = f.filter_on;
    const uint32_t cur_lo = fp.has_abun_lo ? fp.abun_lo : 0u, cur_hi = fp.has_abun_hi ? fp.abun_hi : UINT32_MAX;
    fp.has_abun_lo = f.has_abun_lo;
    fp.abun_lo = f.has_abun_lo ? std::max(f.abun_lo, cur_lo) : 0u;
    fp.has_abun_hi = f.has_abun_hi;
    fp.abun_hi = f.has_abun_hi ? std::min(f.abun_hi, cur_hi) : 0u;
    fp.err_filter = std::max(fp.err_filter, f.err_filter);
    fp.strand_filter = std::max(fp.strand_filter, f.strand_filter);
    return FH_OK;
} FINCH_CATCH

} // extern "C"
