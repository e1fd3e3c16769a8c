// fh_fqstrip.h -- plain 4-line FASTQ text -> the packed sequence stream, on the host, by a team of threads.
//
// needletail hands finch one record at a time (lib.rs:60-68) and finch sketches its sequence() bytes (mash.rs:72-76); the
// device only ever needs those: 151 of the ~316 bytes a 150-base record's text has.  The device-side splitter (fh_text.hip)
// takes the whole text over the PCIe link -- 2.1 bytes per base -- to throw headers, '+' lines and quality strings away
// there; with enough read threads the host drops them BEFORE the link and ~1.05 bytes per base cross it.
//
// The text of a chunk begins at a record and ends behind one (the reader cuts it so).  T threads share it by bytes, and each
// reads its stretch ONCE (the strip is bound by memory, not by instructions: a pass that counted the newlines first, to know
// every stretch's line phase, cost as much as the pass that does the work):
//   1. thread t > 0 GUESSES the first record that begins in its stretch: the first line that begins with '@', whose second
//      line below begins with '+' and whose fourth is the end of the text or begins with '@' (a quality line may begin with
//      '@' too: its second line below is a sequence line);
//   2. every thread walks the records from its start to the next thread's: header must begin with '@', the third line with
//      '+', sequence and quality must be equally long (a CR in front of the line end not counted) -- what needletail checks
//      -- and copies each sequence line (blanks dropped as normalize(false) does, fh_strip.h), one breaker byte behind it,
//      to the output;
//   3. the guesses are PROVED by the walks: thread 0 starts at a record (the chunk does), so what it walks are records, and it
//      must land exactly on thread 1's start -- which is then a record's start too -- and so on down the chain.  A walk that
//      does not land on the next start (a wrong guess, or text that is not 4-line FASTQ) fails the chunk.
// Where a thread's output goes cannot wait for the threads in front of it to know how much they keep: thread t writes from
// out[ceil(start_t / 2)] on -- a record's sequence and breaker are less than half of its text, so the regions cannot overlap
// -- and fills what it leaves of its region with breaker bytes: a few per cent of the stream (the headers' share) are
// breakers between records, which no k-mer spans and no window is valid on.
// A chunk that fails is "not for this path": the caller reads the input again through the paths that are the judges of
// what needletail accepts (the device-side splitter, then parse_fastx), exactly as before.
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

// position of the first '\n' in p[from, n), or n
inline size_t next_nl(const uint8_t *p, size_t from, size_t n) {
    if (from >= n) return n;
    const void *q = memchr(p + from, '\n', n - from);
    return q ? (size_t)((const uint8_t *)q - p) : n;
}

struct Piece { // one thread's share of a chunk
    size_t rec_begin = 0; // the first record that begins in the stretch ((size_t)-1: none)
    uint64_t out_end = 0; // where its output ends (without the filler)
    uint64_t recs = 0, bases = 0;
    bool bad = false;
};
constexpr size_t NONE = (size_t)-1;

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

// the guess: the first line at or behind `from` that looks like a record's header (see above), NONE if there is none in [from, to)
inline size_t guess_record(const uint8_t *text, size_t from, size_t to, size_t n) {
    size_t ls = from;
    if (from > 0 && text[from - 1] != '\n') {
        const size_t q = next_nl(text, from, n);
        ls = q < n ? q + 1 : n;
    }
    while (ls < to && ls < n) {
        const size_t e0 = next_nl(text, ls, n), l1 = e0 < n ? e0 + 1 : n;
        if (text[ls] == '@') {
            const size_t e1 = next_nl(text, l1, n), l2 = e1 < n ? e1 + 1 : n;
            if (l2 < n && text[l2] == '+') {
                const size_t e2 = next_nl(text, l2, n), l3 = e2 < n ? e2 + 1 : n;
                const size_t e3 = next_nl(text, l3, n), l4 = e3 < n ? e3 + 1 : n;
                if (l4 >= n || text[l4] == '@') return ls;
            }
        }
        ls = l1;
    }
    return NONE;
}

// One record at `pos` given the positions of its four newlines (q_end == n: the input's last line without its newline): checked,
// its sequence copied to o (blanks dropped, the bytes they leave and the breaker filled with 0).  -> bytes written, 0 = not
// 4-line FASTQ.  o_limit: where the region ends -- nothing at all is written at or behind it.
inline size_t take_record(const uint8_t *text, size_t pos, size_t n, size_t h_end, size_t s_end, size_t p_end, size_t q_end, uint8_t *o,
                          const uint8_t *o_limit, Piece &me) {
    if (text[pos] != '@' || p_end >= n || text[s_end + 1] != '+') return 0; // (p_end < n: all three newlines are there)
    const size_t s0 = h_end + 1, q0 = p_end + 1;
    size_t sl = s_end - s0, ql = q_end - q0;
    if (sl && text[s_end - 1] == '\r') --sl;
    if (ql && text[q_end - 1] == '\r') --ql;
    if (sl != ql) return 0;
    // (fh_strip::strip stores whole vectors: up to 32 bytes behind the line are scratch -- overwritten by the next record, but not to
    // be written where the next piece's region begins)
    const bool exact = o + sl + 33 > o_limit;
    const size_t kept = exact ? fh_strip::strip_scalar(o, text + s0, sl) : fh_strip::strip(o, text + s0, sl);
    if (kept != sl) memset(o + kept, 0, sl - kept); // blanks inside a sequence line: breakers where they leave room (no k-mer spans one)
    o[sl] = 0;
    me.recs++;
    me.bases += sl;
    return sl + 1;
}

// the records of text[pos, end) copied to o; -> where the walk stopped (== end if all is well; me.bad otherwise), *o_end = the output's end
inline size_t walk_copy_scalar(const uint8_t *text, size_t pos, size_t end, size_t n, uint8_t *o, const uint8_t *o_limit, uint8_t **o_end, Piece &me) {
    while (pos < end) {
        auto after = [n](size_t e) { return e < n ? e + 1 : n; };
        const size_t h_end = next_nl(text, pos, n), s_end = next_nl(text, after(h_end), n), p_end = next_nl(text, after(s_end), n);
        const size_t q_end = next_nl(text, after(p_end), n);
        const size_t nxt = q_end < n ? q_end + 1 : n;
        const size_t w = take_record(text, pos, n, h_end, s_end, p_end, q_end, o, o_limit, me);
        if (!w) {
            me.bad = true;
            break;
        }
        o += w;
        pos = nxt;
    }
    *o_end = o;
    return pos;
}
#if defined(__x86_64__)
// The same with the newlines found 32 bytes a step, a step's newlines as a bit mask: a record of 150 bases is ten steps and
// four bits where four memchr calls take four times as long.
__attribute__((target("avx2"))) inline size_t walk_copy_avx2(const uint8_t *text, size_t pos, size_t end, size_t n, uint8_t *o, const uint8_t *o_limit,
                                                             uint8_t **o_end, Piece &me) {
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
        const size_t nxt = q_end < n ? q_end + 1 : n;
        const size_t w = take_record(text, pos, n, h_end, s_end, p_end, q_end, o, o_limit, me);
        if (!w) {
            me.bad = true;
            break;
        }
        o += w;
        pos = nxt;
    }
#undef FQ_NEXT_NL
    *o_end = o;
    return pos;
}
#endif

// What thread t of T does for the chunk text[0, n) (whole records; the last line may lack its '\n').  All T threads call it
// with the same arguments and their own t; pieces has T entries; out has room for n / 2 + 64 bytes.  On return (every
// thread): *ok = the text is plain 4-line FASTQ and every guess was proved; then out[0, *m) is its packed stream (with
// breaker bytes between the threads' regions), *n_rec / *bases its records and the sum of their sequence lengths
// (mash.rs:72).  Thread 0 writes the four results.
inline void strip_chunk(unsigned t, unsigned T, const uint8_t *text, size_t n, uint8_t *out, std::vector<Piece> &pieces, Barrier &bar,
                        bool *ok, uint64_t *m, uint64_t *n_rec, uint64_t *bases) {
    Piece &me = pieces[t];
    const size_t b0 = n * t / T, b1 = n * (t + 1) / T;
    me.recs = me.bases = 0;
    me.bad = false;
    me.out_end = 0;
    // 1. where my first record begins
    me.rec_begin = t == 0 ? (n ? 0 : NONE) : guess_record(text, b0, b1, n);
    bar.wait();
    // 2. my records: up to the next piece's first one
    size_t end = n;
    for (unsigned i = t + 1; i < T; ++i)
        if (pieces[i].rec_begin != NONE) {
            end = pieces[i].rec_begin;
            break;
        }
    if (me.rec_begin != NONE) {
        uint8_t *const o0 = out + (me.rec_begin + 1) / 2, *o_end = o0;
        const uint8_t *const o_limit = out + (end + 1) / 2 + (end >= n ? 32 : 0); // (the buffer's slack behind the last piece: 64 bytes)
        size_t pos;
#if defined(__x86_64__)
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) pos = walk_copy_avx2(text, me.rec_begin, end, n, o0, o_limit, &o_end, me);
        else
#endif
            pos = walk_copy_scalar(text, me.rec_begin, end, n, o0, o_limit, &o_end, me);
        if (!me.bad && pos != end) me.bad = true; // (did not land on the next piece's start: that guess, or the text, is wrong)
        me.out_end = (uint64_t)(o_end - out);
        // what is left of my region is breakers (the last piece's output simply ends)
        if (!me.bad && end < n) {
            uint8_t *const region_end = out + (end + 1) / 2;
            if (o_end < region_end) memset(o_end, 0, (size_t)(region_end - o_end));
        }
    }
    bar.wait();
    if (t == 0) {
        bool any_bad = false;
        uint64_t total = 0, recs = 0, bs = 0;
        for (unsigned i = 0; i < T; ++i) {
            any_bad |= pieces[i].bad;
            if (pieces[i].rec_begin != NONE) total = pieces[i].out_end; // (the last piece that has records says where the stream ends)
            recs += pieces[i].recs;
            bs += pieces[i].bases;
        }
        *ok = !any_bad;
        *m = total;
        *n_rec = recs;
        *bases = bs;
    }
    bar.wait(); // (thread 0's four results are there when anybody returns)
}

} // namespace fqstrip
