// fh_fqstrip.h -- plain 4-line FASTQ text -> the packed sequence stream, on the host, by a team of threads.
//
// needletail hands finch one record at a time (lib.rs:60-68) and finch sketches its sequence() bytes (mash.rs:72-76); the
// device only ever needs those: 151 of the ~316 bytes a 150-base record's text has.  The device-side splitter (fh_text.hip)
// takes the whole text over the PCIe link -- 2.1 bytes per base -- to throw headers, '+' lines and quality strings away
// there; with enough read threads the host drops them BEFORE the link and 1.007 bytes per base cross it.
//
// The text of a chunk begins at a record and ends behind one (the reader cuts it so).  T threads share it by bytes:
//   1. every thread counts the newlines of its stretch; a prefix sum gives every stretch the index of its first line, hence
//      (index mod 4) where its first RECORD begins -- found by skipping at most three lines;
//   2. every thread walks the records that begin in its stretch once: header must begin with '@', the third line with '+',
//      sequence and quality must be equally long (CR before the line end not counted) -- what needletail checks -- and notes
//      where each sequence line lies; a prefix sum over the threads' packed sizes places their output;
//   3. every thread copies its sequence lines (blanks dropped as normalize(false) does, fh_strip.h) and puts the breaker
//      behind each.
// Anything else than plain 4-line FASTQ (a blank line between records, a sequence over several lines, a length mismatch) makes
// the whole chunk "not for this path": the caller reads the input again through the parser that is the judge of what
// needletail accepts (parse_fastx), exactly as it does when the device-side splitter refuses a text.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "fh_strip.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fqstrip {

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline size_t count_nl_avx2(const uint8_t *p, size_t n) {
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t i = 0, c = 0;
    for (; i + 32 <= n; i += 32)
        c += (size_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i)), nl)));
    for (; i < n; ++i) c += p[i] == '\n';
    return c;
}
#endif
inline size_t count_nl(const uint8_t *p, size_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return count_nl_avx2(p, n);
#endif
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) c += p[i] == '\n';
    return c;
}

// position of the first '\n' in p[from, n), or n
inline size_t next_nl(const uint8_t *p, size_t from, size_t n) {
    if (from >= n) return n;
    const void *q = memchr(p + from, '\n', n - from);
    return q ? (size_t)((const uint8_t *)q - p) : n;
}

struct SeqLine {
    uint64_t off;  // where the record's sequence line begins in the text
    uint32_t len;  // its bytes without the line end (and without a CR in front of it)
};

struct Piece { // one thread's share of a chunk
    size_t first_line = 0, first_idx = 0; // the first line that begins in the stretch, and its index in the chunk
    size_t rec_begin = 0;                 // the first record that begins in the stretch (or where the next piece's does)
    std::vector<SeqLine> lines;
    uint64_t packed = 0, bases = 0;
    bool bad = false;
};

// One record at `pos`: header, sequence, '+' line, quality (whose newline the input's last record may lack), given the
// positions of its four newlines (q_end == n: the last line of the input without its newline).  false: not 4-line FASTQ.
inline bool take_record(const uint8_t *text, size_t pos, size_t n, size_t h_end, size_t s_end, size_t p_end, size_t q_end, Piece &me) {
    if (text[pos] != '@' || p_end >= n || text[s_end + 1] != '+') return false; // (p_end < n: all three newlines are there)
    const size_t s0 = h_end + 1, q0 = p_end + 1;
    size_t sl = s_end - s0, ql = q_end - q0;
    if (sl && text[s_end - 1] == '\r') --sl;
    if (ql && text[q_end - 1] == '\r') --ql;
    if (sl != ql || sl > 0xFFFFFFF0u) return false;
    me.lines.push_back(SeqLine{(uint64_t)s0, (uint32_t)sl});
    me.bases += sl;
    return true;
}

// the records of text[pos, end) (text has n bytes); -> where the walk stopped (== end if all is well; me.bad otherwise)
inline size_t walk_records_scalar(const uint8_t *text, size_t pos, size_t end, size_t n, Piece &me) {
    while (pos < end) {
        const size_t h_end = next_nl(text, pos, n), s_end = next_nl(text, h_end + 1, n), p_end = next_nl(text, s_end + 1, n);
        const size_t q_end = next_nl(text, p_end + 1, n);
        if (!take_record(text, pos, n, h_end, s_end, p_end, q_end, me)) {
            me.bad = true;
            return pos;
        }
        pos = q_end < n ? q_end + 1 : n;
    }
    return pos;
}
#if defined(__x86_64__)
// The same with the newlines found 32 bytes a step, a step's newlines as a bit mask: a record of 150 bases is ten steps and
// four bits (~20 ns) where four memchr calls take ~80.
__attribute__((target("avx2"))) inline size_t walk_records_avx2(const uint8_t *text, size_t pos, size_t end, size_t n, Piece &me) {
    const __m256i nlv = _mm256_set1_epi8('\n');
    size_t base = pos; // the step the mask is of
    uint32_t mask = 0; // its newlines not yet handed out
    bool loaded = false;
    // the next newline at or behind the scan position, or n
#define FQ_NEXT_NL(var)                                                                                                     \
    for (;;) {                                                                                                              \
        if (!loaded) {                                                                                                      \
            if (base >= n) {                                                                                                \
                var = n;                                                                                                    \
                break;                                                                                                      \
            }                                                                                                               \
            if (base + 32 <= n) mask = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(text + base)), nlv)); \
            else {                                                                                                          \
                mask = 0;                                                                                                   \
                for (size_t i_ = base; i_ < n; ++i_)                                                                        \
                    if (text[i_] == '\n') mask |= 1u << (i_ - base);                                                        \
            }                                                                                                               \
            loaded = true;                                                                                                  \
        }                                                                                                                   \
        if (mask) {                                                                                                         \
            var = base + (size_t)__builtin_ctz(mask);                                                                       \
            mask &= mask - 1;                                                                                               \
            break;                                                                                                          \
        }                                                                                                                   \
        base += 32;                                                                                                         \
        loaded = false;                                                                                                     \
    }
    while (pos < end) {
        size_t h_end, s_end, p_end, q_end;
        FQ_NEXT_NL(h_end)
        FQ_NEXT_NL(s_end)
        FQ_NEXT_NL(p_end)
        FQ_NEXT_NL(q_end)
        if (!take_record(text, pos, n, h_end, s_end, p_end, q_end, me)) {
            me.bad = true;
            return pos;
        }
        pos = q_end < n ? q_end + 1 : n;
    }
#undef FQ_NEXT_NL
    return pos;
}
#endif

// a sense-reversing barrier for a fixed team (spins with yield: the phases are a fraction of a millisecond apart)
struct Barrier {
    explicit Barrier(unsigned n) : n_(n) {}
    void wait() {
        const unsigned g = gen_.load(std::memory_order_acquire);
        if (arrived_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            arrived_.store(0, std::memory_order_relaxed);
            gen_.store(g + 1, std::memory_order_release);
        } else {
            while (gen_.load(std::memory_order_acquire) == g) std::this_thread::yield();
        }
    }
    unsigned n_;
    std::atomic<unsigned> arrived_{0}, gen_{0};
};

// What thread t of T does for the chunk text[0, n) (whole records; the last line may lack its '\n').  All T threads call it
// with the same arguments and their own t; pieces has T entries; out has room for n / 2 + 64 bytes.  On return (every
// thread): *ok = the text is plain 4-line FASTQ; then out[0, *m) is its packed stream, *n_rec / *bases its records and the
// sum of their sequence lengths (mash.rs:72).  Thread 0 writes the four results.
inline void strip_chunk(unsigned t, unsigned T, const uint8_t *text, size_t n, uint8_t *out, std::vector<Piece> &pieces, Barrier &bar,
                        bool *ok, uint64_t *m, uint64_t *n_rec, uint64_t *bases) {
    Piece &me = pieces[t];
    const size_t b0 = n * t / T, b1 = n * (t + 1) / T;
    me.lines.clear();
    me.packed = me.bases = 0;
    me.bad = false;
    // 1. newlines of the stretch -> (after the barrier) index of the first line that begins in it
    const size_t my_nl = count_nl(text + b0, b1 - b0);
    me.first_idx = my_nl; // (borrowed: the count, until thread 0 has turned the counts into indices)
    bar.wait();
    if (t == 0) {
        size_t before = 0; // newlines in text[0, b_i)
        for (unsigned i = 0; i < T; ++i) {
            const size_t cnt = pieces[i].first_idx, bi = n * i / T;
            // a line begins at bi iff bi == 0 or the byte in front is a newline; otherwise the stretch's first line begins behind
            // its first newline
            if (bi == 0 || text[bi - 1] == '\n') {
                pieces[i].first_line = bi;
                pieces[i].first_idx = before;
            } else {
                const size_t q = next_nl(text, bi, n);
                pieces[i].first_line = q < n ? q + 1 : n;
                pieces[i].first_idx = before + 1;
            }
            before += cnt;
        }
    }
    bar.wait();
    // the first RECORD that begins in the stretch: skip to the next line whose index is a multiple of four
    {
        size_t pos = me.first_line, idx = me.first_idx;
        while (pos < n && (idx & 3u)) {
            const size_t q = next_nl(text, pos, n);
            pos = q < n ? q + 1 : n;
            ++idx;
        }
        me.rec_begin = pos < b1 ? pos : (size_t)-1; // (a record that begins in a later stretch is that stretch's; b1 of the last one is n)
    }
    bar.wait();
    // where my records end: at the next piece's first record
    size_t end = n;
    for (unsigned i = t + 1; i < T; ++i)
        if (pieces[i].rec_begin != (size_t)-1) {
            end = pieces[i].rec_begin;
            break;
        }
    // 2. walk my records once
    if (me.rec_begin != (size_t)-1) {
        size_t pos = me.rec_begin;
#if defined(__x86_64__)
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) pos = walk_records_avx2(text, pos, end, n, me);
        else
#endif
            pos = walk_records_scalar(text, pos, end, n, me);
        if (!me.bad && pos != end) me.bad = true; // (a record ran over where the next stretch's first one begins: not 4-line text)
    }
    bar.wait();
    // 3. copy.  A line's packed size is only known once its blanks are gone (they are rare: one pass that strips into place
    // and, if any line shrank, the pieces behind it would not be contiguous) -- so blanks are looked for first, cheaply, per
    // line; lines without any (all of them, in practice) are plain copies.
    {
        uint64_t sz = 0;
        for (const SeqLine &L : me.lines) sz += (uint64_t)L.len + 1;
        me.packed = sz; // upper bound: exact if no line holds a blank
    }
    bar.wait();
    bool any_bad = false;
    uint64_t my_off = 0, total = 0, recs = 0, bs = 0;
    for (unsigned i = 0; i < T; ++i) {
        any_bad |= pieces[i].bad;
        if (i < t) my_off += pieces[i].packed;
        total += pieces[i].packed;
        recs += pieces[i].lines.size();
        bs += pieces[i].bases;
    }
    bool shrank = false;
    if (!any_bad) {
        uint8_t *o = out + my_off;
        const size_t n_lines = me.lines.size();
        for (size_t li = 0; li < n_lines; ++li) {
            const SeqLine &L = me.lines[li];
            // (fh_strip::strip stores whole vectors: what it writes behind a line is overwritten by the next line -- except behind
            // the piece's last line, where the next piece's output begins: that one goes byte-exact)
            const size_t kept = li + 1 < n_lines ? fh_strip::strip(o, text + L.off, L.len) : fh_strip::strip_scalar(o, text + L.off, L.len);
            if (kept != L.len) { // blanks inside a sequence line: the breaker bytes fill what they leave (k-mers never span a breaker)
                memset(o + kept, 0, L.len - kept);
                shrank = true;
            }
            o[L.len] = 0;
            o += (size_t)L.len + 1;
        }
    }
    (void)shrank;
    bar.wait();
    if (t == 0) {
        *ok = !any_bad;
        *m = total;
        *n_rec = recs;
        *bases = bs;
    }
}

} // namespace fqstrip
