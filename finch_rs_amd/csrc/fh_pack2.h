// fh_pack2.h -- the batch sketcher's TWO-BIT input form (include/finch_hip.h, fh_batch_submit_packed), written by the host
// while it stages a file.
//
// A worker of finch::sketch_files (lib/src/lib.rs:29-49) moves every byte of a genome over the host-to-device link, and
// with many files per launch that link is what bounds configs[4] (docs/MEASUREMENTS_r06.md 1).  What the sketch kernel
// makes of a tile of 2048 sequence bytes in its phase A is 768 bytes: per lane the 2-bit codes of its 32 positions (A 0, C 1,
// G 2, T/U 3 -- needletail's normalize(false) + canonical_kmers as fh_core.h classify4 restates them: ACGT, acgt, U/u are
// bases, every other byte breaks k-mers) and one "is a base" bit per position.  Producing exactly that on the host, in the
// pass that strips the line ends anyway, puts 0.375 bytes per position on the link instead of 1 and leaves the kernel's
// phase A two loads.
//
// Layout of a file's region (64-byte aligned), TILE_BYTES = 768 per tile of 2048 positions:
//     tile t at region + 768 t:   [ 64 x u64 codes | 64 x u32 good ]
//     group g = position / 32 of the tile: codes[g] bits [2 i, 2 i + 2) = code of position 32 g + i, good[g] bit i = it is a base
//     (the code of a position that is no base is unspecified: the kernel masks every window that holds one)
// ceil(len / 2048) tiles hold the file; ONE more tile of zeroes follows (the kernel loads the tile behind the one it hashes:
// the halo of its last lane), and positions behind `len` in the last tile are zero (not a base).
//
// Test infrastructure does not live here: tests/test_batch_packed.py checks pack32 against the numpy restatement of the
// classification for every byte value and the AVX2 form against the scalar one.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "fh_strip.h"

namespace fh_pack2 {

constexpr uint64_t TILE_POS = 2048, TILE_BYTES = 768, CODES_BYTES = 512;

inline uint64_t region_bytes(uint64_t len) { return ((len + TILE_POS - 1) / TILE_POS + 1) * TILE_BYTES; }

// one byte -> its code and base bit (fh_core.h classify4: the byte's low three bits pick the letter it has to be)
inline void classify1(uint8_t b, uint64_t &code, uint32_t &good) {
    static const uint8_t EXPECT[8] = {0xFF, 'A', 0xFF, 'C', 'T', 'U', 0xFF, 'G'};
    static const uint8_t CODE[8] = {0, 0, 0, 1, 3, 3, 0, 2};
    code = CODE[b & 7];
    good = (uint8_t)(b & 0xDF) == EXPECT[b & 7];
}

// 32 bytes -> their codes and base bits
inline void pack32_scalar(const uint8_t *p, uint64_t &codes, uint32_t &good) {
    uint64_t c = 0;
    uint32_t g = 0;
    for (int i = 0; i < 32; ++i) {
        uint64_t ci;
        uint32_t gi;
        classify1(p[i], ci, gi);
        c |= ci << (2 * i);
        g |= gi << i;
    }
    codes = c;
    good = g;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void pack32_avx2(const uint8_t *p, uint64_t &codes, uint32_t &good) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)p);
    const __m256i idx = _mm256_and_si256(v, _mm256_set1_epi8(7));
    const __m256i expect_tbl = _mm256_setr_epi8((char)0xFF, 'A', (char)0xFF, 'C', 'T', 'U', (char)0xFF, 'G', 0, 0, 0, 0, 0, 0, 0, 0, //
                                                (char)0xFF, 'A', (char)0xFF, 'C', 'T', 'U', (char)0xFF, 'G', 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i code_tbl = _mm256_setr_epi8(0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i expect = _mm256_shuffle_epi8(expect_tbl, idx);
    const __m256i code = _mm256_shuffle_epi8(code_tbl, idx);
    good = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(v, _mm256_set1_epi8((char)0xDF)), expect));
    // four codes -> one byte: c0 + 4 c1 in every 16-bit lane, then (that) + 16 (its neighbour) in every 32-bit lane
    const __m256i x16 = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0401));
    const __m256i x32 = _mm256_madd_epi16(x16, _mm256_set1_epi32(0x00100001));
    const __m256i y = _mm256_packus_epi32(x32, x32);
    const __m256i z = _mm256_packus_epi16(y, y); // the low dword of each 128-bit half: the codes of its 16 bases
    codes = (uint64_t)(uint32_t)_mm256_extract_epi32(z, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(z, 4) << 32);
}
#endif

inline bool have_avx2() {
#if defined(__x86_64__)
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
#else
    return false;
#endif
}

// whole groups of 32 bytes, the first one being group `g0` of the region
inline void pack_groups(const uint8_t *src, size_t n_groups, uint8_t *region, uint64_t g0, bool avx2) {
    for (size_t i = 0; i < n_groups; ++i) {
        const uint64_t g = g0 + i;
        uint8_t *const tile = region + (g >> 6) * TILE_BYTES;
        uint64_t c;
        uint32_t gd;
#if defined(__x86_64__)
        if (avx2) pack32_avx2(src + 32 * i, c, gd);
        else
#endif
            pack32_scalar(src + 32 * i, c, gd);
        memcpy(tile + 8 * (g & 63), &c, 8);
        memcpy(tile + CODES_BYTES + 4 * (g & 63), &gd, 4);
    }
}

inline bool have_bmi2() {
#if defined(__x86_64__)
    static const bool v = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt");
    return v;
#else
    return false;
#endif
}

// A file's region filled piece by piece.  The caller has checked region_bytes(most positions it may add) against the room it has.
//
// Two forms of text(): with AVX2 + BMI2 ONE pass over the text -- every 32 bytes are classified as they stand, the blanks'
// codes and base bits are squeezed out of the two words (pext) and what is left is appended to a bit accumulator that emits a
// group whenever it holds 32 positions; nothing is written but the region.  Without BMI2 (or with `fused` cleared) the text is
// stripped into a buffer the cache keeps (fh_strip.h) and whole groups of 32 bytes leave it for the region.  The two never mix
// within a file.
struct Packer {
    static constexpr size_t PIECE = 16384; // bytes of text stripped into the buffer at a time
    uint8_t *region = nullptr;
    uint64_t groups = 0; // whole groups written
    size_t fill = 0;     // bytes waiting in tmp
    bool avx2 = have_avx2();
    bool fused = have_bmi2();
    // the fused form's accumulator: `na` (< 32) positions that wait for their group to fill up
    unsigned __int128 acc_c = 0;
    uint64_t acc_g = 0;
    unsigned na = 0;
    alignas(64) uint8_t tmp[PIECE + 128];

    // tests: 1 = the portable form, 2 = two passes with AVX2 (what a CPU without BMI2 runs), anything else = the best there is
    void force_form(unsigned f) {
        avx2 = have_avx2() && f != 1;
        fused = have_bmi2() && f != 1 && f != 2;
    }
    void begin(uint8_t *r) { region = r, groups = 0, fill = 0, acc_c = 0, acc_g = 0, na = 0; }
    void emit(uint64_t c, uint32_t g) {
        uint8_t *const tile = region + (groups >> 6) * TILE_BYTES;
        memcpy(tile + 8 * (groups & 63), &c, 8);
        memcpy(tile + CODES_BYTES + 4 * (groups & 63), &g, 4);
        ++groups;
    }
    // cnt (1..32) positions: their codes in the low 2 cnt bits of c, their base bits in the low cnt bits of g, nothing above
    void append(uint64_t c, uint32_t g, unsigned cnt) {
        acc_c |= (unsigned __int128)c << (2 * na);
        acc_g |= (uint64_t)g << na;
        na += cnt;
        if (na >= 32) {
            emit((uint64_t)acc_c, (uint32_t)acc_g);
            acc_c >>= 64;
            acc_g >>= 32;
            na -= 32;
        }
    }
#if defined(__x86_64__)
    __attribute__((target("avx2,bmi2,popcnt"))) void text_fused(const uint8_t *src, size_t n) {
        const __m256i blank_tbl = _mm256_setr_epi8(' ', -1, -1, -1, -1, -1, -1, -1, -1, '\t', '\n', -1, -1, '\r', -1, -1, //
                                                   ' ', -1, -1, -1, -1, -1, -1, -1, -1, '\t', '\n', -1, -1, '\r', -1, -1);
        const __m256i expect_tbl = _mm256_setr_epi8((char)0xFF, 'A', (char)0xFF, 'C', 'T', 'U', (char)0xFF, 'G', 0, 0, 0, 0, 0, 0, 0, 0, //
                                                    (char)0xFF, 'A', (char)0xFF, 'C', 'T', 'U', (char)0xFF, 'G', 0, 0, 0, 0, 0, 0, 0, 0);
        const __m256i seven = _mm256_set1_epi8(7), fold = _mm256_set1_epi8((char)0xDF);
        // (the accumulator lives in registers for the length of the call: as members every step would wait for the stores of the one before)
        uint64_t lo = (uint64_t)acc_c, gg = acc_g, grp = groups; // (acc_c holds fewer than 64 bits between calls: na < 32)
        unsigned a = na;
        uint8_t *const reg = region;
        size_t i = 0;
        uint8_t last[32];
        for (;;) {
            const uint8_t *p = src + i;
            if (i + 32 > n) {
                if (i >= n) break;
                memset(last, '\n', 32); // (a blank: dropped)
                memcpy(last, src + i, n - i);
                p = last;
            }
            const __m256i v = _mm256_loadu_si256((const __m256i *)p);
            // is a base: the case-folded byte is the letter its low three bits pick (pack32_avx2, fh_core.h classify4)
            const __m256i expect = _mm256_shuffle_epi8(expect_tbl, _mm256_and_si256(v, seven));
            const uint32_t g = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(v, fold), expect));
            // the code of a base from two bits of its byte: A ..001, C ..011, G ..111, T ..100, U ..101 -> high bit = bit 2, low bit =
            // bit 1 ^ bit 2 (A 0, C 1, G 2, T/U 3).  Where the byte is no base the code is whatever that gives: nobody looks at it.
            const uint32_t m1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 6)); // bit 1 of every byte
            const uint32_t m2 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 5)); // bit 2
            // (' ', '\t', '\n', '\r' have the low nibbles 0, 9, 10, 13: a byte is a blank iff it equals the table entry its low nibble picks;
            // bytes >= 0x80 pick zero, which they are not.  No branch on "is there a blank": a 70-column line puts one into every other
            // vector, in no pattern a predictor learns)
            const __m256i b = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(blank_tbl, v), v);
            const uint32_t keep = ~(uint32_t)_mm256_movemask_epi8(b);
            // squeeze the blanks out of the three masks, then interleave the code bits
            const uint64_t c_lo = _pext_u32(m1 ^ m2, keep), c_hi = _pext_u32(m2, keep);
            const uint64_t cc = _pdep_u64(c_lo, 0x5555555555555555ull) | _pdep_u64(c_hi, 0xAAAAAAAAAAAAAAAAull);
            const uint64_t gc = _pext_u32(g, keep);
            // append: a < 32 positions wait in (lo, gg); cc has at most 64 bits, so lo | cc << 2a spills into hi
            const unsigned sh = 2 * a;
            lo |= cc << sh;
            const uint64_t hi = sh ? cc >> (64 - sh) : 0; // what of cc does not fit the word
            gg |= gc << a;
            a += (unsigned)_mm_popcnt_u32(keep);
            if (a >= 32) {
                uint8_t *const tile = reg + (grp >> 6) * TILE_BYTES;
                const uint32_t g32 = (uint32_t)gg;
                memcpy(tile + 8 * (grp & 63), &lo, 8);
                memcpy(tile + CODES_BYTES + 4 * (grp & 63), &g32, 4);
                ++grp;
                lo = hi;
                gg >>= 32;
                a -= 32;
            }
            i += 32;
        }
        acc_c = lo, acc_g = gg, na = a, groups = grp;
    }
#endif
    // where the next bytes go: at least PIECE + 32 bytes of room behind it
    uint8_t *room() {
        if (fill >= 64) drain();
        return tmp + fill;
    }
    void added(size_t n) { fill += n; }
    void drain() {
        const size_t n32 = fill / 32;
        pack_groups(tmp, n32, region, groups, avx2);
        groups += n32;
        const size_t rest = fill - 32 * n32;
        if (n32 && rest) memcpy(tmp, tmp + 32 * n32, rest); // (rest < 32 <= 32 n32: no overlap)
        fill = rest;
    }
    void byte(uint8_t b) {
        if (fused) {
            uint64_t c;
            uint32_t g;
            classify1(b, c, g);
            append(c, g, 1);
            return;
        }
        if (fill >= PIECE) drain();
        tmp[fill++] = b;
    }
    // [src, src + n) without its blanks (fh_strip.h: ' ', '\t', '\r', '\n') into the region
    void text(const uint8_t *src, size_t n) {
#if defined(__x86_64__)
        if (fused) return text_fused(src, n);
#endif
        for (size_t o = 0; o < n;) {
            const size_t step = n - o < PIECE - 64 ? n - o : PIECE - 64;
            uint8_t *d = room(); // fill < 64 now: fill + step + 32 <= PIECE + 32
            added(fh_strip::strip(d, src + o, step));
            o += step;
        }
    }
    // -> positions of the file; the region is complete (tail of the last tile and the tile behind it zeroed)
    uint64_t finish() {
        if (fused) {
            fill = 0;
            if (na) { // the last, partial group: what lies above its positions is zero (not a base)
                const uint64_t len_f = groups * 32 + na;
                emit((uint64_t)acc_c, (uint32_t)acc_g);
                acc_c = 0, acc_g = 0, na = 0;
                return close_region(len_f);
            }
            return close_region(groups * 32);
        }
        drain();
        const uint64_t len = groups * 32 + fill;
        if (fill) {
            memset(tmp + fill, 0, 32 - fill);
            pack_groups(tmp, 1, region, groups, avx2);
            ++groups;
            fill = 0;
        }
        return close_region(len);
    }
    uint64_t close_region(uint64_t len) {
        const uint64_t n_tiles = (len + TILE_POS - 1) / TILE_POS;
        // groups behind the last one of the last tile
        for (uint64_t g = groups; g < n_tiles * 64; ++g) {
            uint8_t *const tile = region + (g >> 6) * TILE_BYTES;
            memset(tile + 8 * (g & 63), 0, 8);
            memset(tile + CODES_BYTES + 4 * (g & 63), 0, 4);
        }
        memset(region + n_tiles * TILE_BYTES, 0, TILE_BYTES);
        return len;
    }
};

} // namespace fh_pack2
