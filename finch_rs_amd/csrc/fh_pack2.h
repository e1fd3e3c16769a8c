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

// 32 bytes -> their codes and base bits (fh_core.h classify4: the byte's low three bits pick the letter it has to be)
inline void pack32_scalar(const uint8_t *p, uint64_t &codes, uint32_t &good) {
    static const uint8_t EXPECT[8] = {0xFF, 'A', 0xFF, 'C', 'T', 'U', 0xFF, 'G'};
    static const uint8_t CODE[8] = {0, 0, 0, 1, 3, 3, 0, 2};
    uint64_t c = 0;
    uint32_t g = 0;
    for (int i = 0; i < 32; ++i) {
        const uint8_t b = p[i];
        c |= (uint64_t)CODE[b & 7] << (2 * i);
        g |= (uint32_t)((uint8_t)(b & 0xDF) == EXPECT[b & 7]) << i;
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

// A file's region filled piece by piece: bytes go through a buffer the cache keeps (room / added), whole groups of 32 leave it
// for the region.  The caller has checked region_bytes(most positions it may add) against the room it has.
struct Packer {
    static constexpr size_t PIECE = 16384; // bytes of text stripped into the buffer at a time
    uint8_t *region = nullptr;
    uint64_t groups = 0; // whole groups written
    size_t fill = 0;     // bytes waiting in tmp
    bool avx2 = have_avx2();
    alignas(64) uint8_t tmp[PIECE + 128];

    void begin(uint8_t *r) { region = r, groups = 0, fill = 0; }
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
        if (fill >= PIECE) drain();
        tmp[fill++] = b;
    }
    // strip [src, src + n) of its blanks (fh_strip.h) into the region
    void text(const uint8_t *src, size_t n) {
        for (size_t o = 0; o < n;) {
            const size_t step = n - o < PIECE - 64 ? n - o : PIECE - 64;
            uint8_t *d = room(); // fill < 64 now: fill + step + 32 <= PIECE + 32
            added(fh_strip::strip(d, src + o, step));
            o += step;
        }
    }
    // -> positions of the file; the region is complete (tail of the last tile and the tile behind it zeroed)
    uint64_t finish() {
        drain();
        const uint64_t len = groups * 32 + fill;
        if (fill) {
            memset(tmp + fill, 0, 32 - fill);
            pack_groups(tmp, 1, region, groups, avx2);
            ++groups;
            fill = 0;
        }
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
