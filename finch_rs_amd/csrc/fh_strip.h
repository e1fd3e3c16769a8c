// fh_strip.h -- copy bytes dropping ' ', '\t', '\r', '\n': what needletail's normalize(false) removes from a sequence
// (mash.rs:73), done while the host stages record bytes for the device so that device positions are contiguous.
//
// A sequence line of a FASTA file is 60-80 clean bytes and one newline, a FASTQ sequence line 100-250 and one: the copy is
// memory-bound if it moves vectors and pays for the rare blank by storing what follows it once more, one byte further left
// (AVX2: 32 bytes per step, one extra overlapping store per blank).  The 8-byte SWAR loop this replaces ran at 1.3 GB/s per
// thread on 70-column FASTA and was what bound a batch of small genomes read through the host parser.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fh_strip {

inline bool is_blank(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

// portable form; dst and src may be the same buffer if dst + 8 <= src
inline size_t strip_scalar(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t i = 0, m = 0;
    while (i + 8 <= n) {
        uint64_t x;
        memcpy(&x, src + i, 8);
        if (((x - 0x2121212121212121ull) & ~x & 0x8080808080808080ull) == 0) { // no byte < 0x21
            memcpy(dst + m, &x, 8);
            m += 8;
        } else {
            for (int j = 0; j < 8; ++j) {
                const uint8_t c = src[i + j];
                if (!is_blank(c)) dst[m++] = c;
            }
        }
        i += 8;
    }
    for (; i < n; ++i)
        if (!is_blank(src[i])) dst[m++] = src[i];
    return m;
}

#if defined(__x86_64__)
// dst must have room for the bytes kept PLUS 32 (stores are whole vectors; what lies behind the kept bytes is scratch).
// Packing towards the front of the same buffer is fine if dst + 32 <= src (a store of 32 bytes then never reaches bytes that
// are still to be loaded); closer than that the buffers must not overlap.
__attribute__((target("avx2"))) inline size_t strip_avx2(uint8_t *dst, const uint8_t *src, size_t n) {
    const __m256i sp = _mm256_set1_epi8(' '), tb = _mm256_set1_epi8('\t'), cr = _mm256_set1_epi8('\r'), nl = _mm256_set1_epi8('\n');
    // (the four blanks are all below 0x21: ONE signed compare says "no blank here" for a vector of sequence letters -- seven
    // instructions fewer per vector than the four equalities, which only a vector that has such a byte, or a byte >= 0x80, goes
    // through.  The per-record copy of fh_process is bound by instructions as much as by memory: 150-base records, one call each.)
    const __m256i lim = _mm256_set1_epi8(0x21);
    size_t i = 0, m = 0;
    while (i + 32 <= n) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
        _mm256_storeu_si256((__m256i *)(dst + m), v);
        if (__builtin_expect(_mm256_movemask_epi8(_mm256_cmpgt_epi8(lim, v)) == 0, 1)) {
            m += 32;
            i += 32;
            continue;
        }
        const __m256i b = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, sp), _mm256_cmpeq_epi8(v, tb)),
                                          _mm256_or_si256(_mm256_cmpeq_epi8(v, cr), _mm256_cmpeq_epi8(v, nl)));
        uint32_t mask = (uint32_t)_mm256_movemask_epi8(b);
        if (mask == 0) {
            m += 32;
            i += 32;
            continue;
        }
        // keep the bytes in front of the first blank, then go on right behind it (a fresh 32-byte step from there: a line
        // has one blank, the next step is clean again)
        const unsigned p = (unsigned)__builtin_ctz(mask);
        m += p;
        i += p + 1;
    }
    if (i == n) return m;
    // Fewer than 32 bytes are left.  With a whole vector behind them in src, load the LAST 32 bytes of src once more: if there
    // is no blank among them, the n - i new ones follow in dst exactly where the vector's older bytes already lie (nothing was
    // dropped between them), so one store that overlaps those finishes the copy -- a read's 150 or 250 bases end in such a
    // tail, and the byte loop it replaces took as long as the vectors in front of it
    if (n >= 32 && m >= 32 - (n - i)) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + n - 32));
        bool clean = _mm256_movemask_epi8(_mm256_cmpgt_epi8(lim, v)) == 0;
        if (!clean) {
            const __m256i b = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, sp), _mm256_cmpeq_epi8(v, tb)),
                                              _mm256_or_si256(_mm256_cmpeq_epi8(v, cr), _mm256_cmpeq_epi8(v, nl)));
            clean = _mm256_movemask_epi8(b) == 0;
        }
        if (clean) {
            const size_t r = n - i;
            _mm256_storeu_si256((__m256i *)(dst + m - (32 - r)), v);
            return m + r;
        }
    }
    return m + strip_scalar(dst + m, src + i, n - i);
}
#endif

// how many of src[0, n) would be kept
inline size_t count_kept_scalar(const uint8_t *src, size_t n) {
    size_t blanks = 0;
    for (size_t i = 0; i < n; ++i) blanks += is_blank(src[i]) ? 1 : 0;
    return n - blanks;
}
#if defined(__x86_64__)
__attribute__((target("avx2,popcnt"))) inline size_t count_kept_avx2(const uint8_t *src, size_t n) {
    const __m256i sp = _mm256_set1_epi8(' '), tb = _mm256_set1_epi8('\t'), cr = _mm256_set1_epi8('\r'), nl = _mm256_set1_epi8('\n');
    size_t i = 0, blanks = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
        const __m256i b = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, sp), _mm256_cmpeq_epi8(v, tb)),
                                          _mm256_or_si256(_mm256_cmpeq_epi8(v, cr), _mm256_cmpeq_epi8(v, nl)));
        blanks += (size_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(b));
    }
    for (; i < n; ++i) blanks += is_blank(src[i]) ? 1 : 0;
    return n - blanks;
}
#endif
inline size_t count_kept(const uint8_t *src, size_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
    if (avx2) return count_kept_avx2(src, n);
#endif
    return count_kept_scalar(src, n);
}

// -> bytes written.  dst needs room for n + 32 bytes.
inline size_t strip(uint8_t *dst, const uint8_t *src, size_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return strip_avx2(dst, src, n);
#endif
    return strip_scalar(dst, src, n);
}

} // namespace fh_strip
