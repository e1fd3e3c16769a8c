// fh_core.h -- per-lane arithmetic of the sketch kernel (K2), written once so that the exact same
// code is compiled by hipcc for gfx950 and (for logic tests on a GPU-less box) by g++ for the host.
//
// What it restates (reference file:line, relative to the finch-rs tree):
//   classify4/classify_chunk : needletail 0.5.0 Sequence::normalize(false) restricted to the
//                              k-mer-relevant alphabet (mash.rs:73): {A,C,G,T,a,c,g,t,u,U} -> 2-bit codes
//                              (u/U -> T), every other byte breaks k-mers.  Whitespace never reaches the
//                              device (stripped while staging, see fh_api).
//   canonical selection      : needletail canonical_kmers (mash.rs:76): fwd < rc ? (fwd,false) : (rc,true)
//   murmur_h1_lut<K>         : murmurhash3 0.0.5 murmurhash3_x64_128(kmer, seed).0 (hashing.rs:10-12) on the
//                              ASCII bytes of the canonical k-mer.
//
// Layouts.  A k-mer is held as a 2-bit word, A=0 C=1 G=2 T=3 (ASCII order == code order):
//   "m-form" (MSB-first): first base in the most significant used digit -> numeric order == lexicographic
//   "l-form" (LSB-first): base i at bits [2i+1:2i]
// For a forward window with l-form F and m-form Fm:  m-form of its reverse complement == ~F & mask.
//
// murmur3 via lookup tables: the three "first stage" products k1*c1, k2*c2 (block) and the tail products
// are linear in the key bytes, so  (ascii bytes of a 4-base group) * c  mod 2^64 is tabulated per group
// value (256 entries x 8 B per constant) and summed; only the 7 (k=21) / 8 (k=31) data-dependent
// 64-bit multiplies of the later stages remain.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FH_HD __host__ __device__ __forceinline__
#define FH_HDM __host__ __device__ __forceinline__
#else
#define FH_HD static inline
#define FH_HDM inline
#endif

namespace fh {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr u64 MURMUR_C1 = 0x87c37b91114253d5ULL;
constexpr u64 MURMUR_C2 = 0x4cf5ad432745937fULL;
constexpr u64 EMPTY64 = ~0ULL;

FH_HD u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }

FH_HD u64 fmix64(u64 k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

// v_perm_b32: byte i of the result = byte sel.byte[i] of the 8-byte value {s0 (bytes 4..7), s1 (bytes 0..3)}
FH_HD u32 perm_b32(u32 s0, u32 s1, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(s0, s1, sel);
#else
    u64 src = ((u64)s0 << 32) | s1;
    u32 r = 0;
    for (int i = 0; i < 4; ++i) {
        u32 s = (sel >> (8 * i)) & 0xff;
        u32 b = (s < 8) ? (u32)((src >> (8 * s)) & 0xff) : (s == 12 ? 0u : 0xffu);
        r |= b << (8 * i);
    }
    return r;
#endif
}

// 4 ASCII bytes -> q8: 2-bit codes, base i at bits [2i+1:2i];  good4: bit i set iff byte i is in ACGTUacgtu
FH_HD void classify4(u32 d, u32 &q8, u32 &good4) {
    // (c >> 1) & 3 : A->0 C->1 G->3 T/U->2 ; x ^ (x >> 1) : 0 1 2 3
    u32 x = (d >> 1) & 0x03030303u;
    u32 code = x ^ ((x >> 1) & 0x01010101u);
    // the letter this code stands for, as an 8-entry byte LUT in a register
    u32 expect = perm_b32(0u, 0x54474341u /* 'T','G','C','A' */, code);
    u32 upper = d & 0xDFDFDFDFu; // fold case
    u32 tmask = (code & (code >> 1)) & 0x01010101u; // 1 where code == 3: accept U (0x55) for T (0x54)
    u32 diff = (upper ^ expect) & ~tmask;
    // per byte: bit 7 set iff diff byte != 0
    u32 nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;
    u32 ok = nz ^ 0x80808080u; // bit 7 of byte i set iff good
    u32 t = code | (code >> 6);
    t = t | (t >> 12);
    q8 = t & 0xFFu;
    u32 g = ok >> 7; // bits 0,8,16,24
    g = g | (g >> 7);
    g = g | (g >> 14);
    good4 = g & 0xFu;
}

// 16 bytes (4 dwords, little endian) -> 16 codes (32 bits, l-form) + 16 good bits
FH_HD void classify_chunk(u32 d0, u32 d1, u32 d2, u32 d3, u32 &codes, u32 &good) {
    u32 q0, q1, q2, q3, g0, g1, g2, g3;
    classify4(d0, q0, g0);
    classify4(d1, q1, g1);
    classify4(d2, q2, g2);
    classify4(d3, q3, g3);
    codes = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
    good = g0 | (g1 << 4) | (g2 << 8) | (g3 << 12);
}

FH_HD u64 kmask(int K) { return K >= 32 ? ~0ULL : ((1ULL << (2 * K)) - 1ULL); }

// reverse the order of the 32 2-bit digits of a 64-bit word
FH_HD u64 pairrev64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo = __builtin_bitreverse32((u32)x), hi = __builtin_bitreverse32((u32)(x >> 32));
    u64 r = ((u64)lo << 32) | hi; // full bit reversal
#else
    u64 r = 0;
    for (int i = 0; i < 64; ++i) r |= ((x >> i) & 1ULL) << (63 - i);
#endif
    return ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
}

// ASCII bytes (little endian, first base in byte 0) of an nb-base group given in m-form
FH_HD u32 ascii_group(u32 q, int nb) {
    u32 w = 0;
    for (int i = 0; i < nb; ++i) {
        u32 code = (q >> (2 * (nb - 1 - i))) & 3u;
        u32 ch = code == 0 ? 0x41u : code == 1 ? 0x43u : code == 2 ? 0x47u : 0x54u;
        w |= ch << (8 * i);
    }
    return w;
}

// Geometry of the murmur3 key for a K-byte k-mer, group g = key bytes [4g, 4g+nb)
struct GroupGeom {
    int nb;     // bases in the group (1..4)
    int shift;  // right shift of the m-form canonical word that brings the group to bit 0
    int word;   // 0.. : (block b -> 2b,2b+1), tail -> 2*NB, 2*NB+1 ; odd = k2 word
    bool hi;    // group sits in the high 32 bits of its 64-bit word
    bool is_k2; // multiplied by c2 (k2 words) else c1 (k1 words)
};

constexpr GroupGeom group_geom(int K, int g) {
    GroupGeom r{};
    int B = 4 * g;
    int NB = K / 16;
    r.nb = (K - B) < 4 ? (K - B) : 4;
    r.shift = 2 * (K - B - r.nb);
    int o = (B < 16 * NB) ? (B % 16) : (B - 16 * NB);
    int base = (B < 16 * NB) ? 2 * (B / 16) : 2 * NB;
    r.is_k2 = o >= 8;
    r.word = base + (r.is_k2 ? 1 : 0);
    r.hi = (o % 8) >= 4;
    return r;
}

// the constant the partial (last, nb<4) group is multiplied by
constexpr u64 partial_const(int K) { return group_geom(K, (K + 3) / 4 - 1).is_k2 ? MURMUR_C2 : MURMUR_C1; }
constexpr int partial_nb(int K) { return K & 3; }

// Table entry builders (run once per workgroup into LDS / once on the host for tests)
FH_HD u64 lut_entry(u32 q, int nb, u64 c) { return (u64)ascii_group(q, nb) * c; }

// murmurhash3_x64_128(ascii(canonical k-mer), seed).0 given the m-form canonical word.
// T1[q] = ascii_group(q,4)*c1, T2[q] = ascii_group(q,4)*c2 (256 entries each),
// TP[q] = ascii_group(q, K&3) * partial_const(K) (4^(K&3) entries, unused if K%4==0).
template <int K>
FH_HD u64 murmur_h1_lut(u64 cm, u64 seed, const u64 *T1, const u64 *T2, const u64 *TP) {
    constexpr int NB = K / 16, TAIL = K & 15, NG = (K + 3) / 4;
    u64 wc[2 * NB + 2];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 2 * NB + 2; ++i) wc[i] = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int g = 0; g < NG; ++g) {
        const GroupGeom gg = group_geom(K, g);
        u32 q = (u32)(cm >> gg.shift) & ((1u << (2 * gg.nb)) - 1u);
        const u64 *T = (gg.nb == 4) ? (gg.is_k2 ? T2 : T1) : TP;
#if defined(FH_ABL_NOLDS)
        (void)T;
        wc[gg.word] += gg.hi ? ((u64)q << 32) : (u64)q * 0x9E3779B1ull; // ablation: no table lookups
#else
        if (gg.hi) {
            u32 plo = ((const u32 *)T)[2 * q]; // low dword of the 64-bit entry (little endian)
            wc[gg.word] += (u64)plo << 32;
        } else {
            wc[gg.word] += T[q];
        }
#endif
    }
    u64 h1 = seed, h2 = seed;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < NB; ++b) {
        u64 k1 = rotl64(wc[2 * b], 31) * MURMUR_C2;
        h1 ^= k1;
        h1 = rotl64(h1, 27) + h2;
        h1 = h1 * 5 + 0x52dce729ULL;
        u64 k2 = rotl64(wc[2 * b + 1], 33) * MURMUR_C1;
        h2 ^= k2;
        h2 = rotl64(h2, 31) + h1;
        h2 = h2 * 5 + 0x38495ab5ULL;
    }
    if (TAIL > 8) {
        u64 k2 = rotl64(wc[2 * NB + 1], 33) * MURMUR_C1;
        h2 ^= k2;
    }
    if (TAIL > 0) {
        u64 k1 = rotl64(wc[2 * NB], 31) * MURMUR_C2;
        h1 ^= k1;
    }
    h1 ^= (u64)K;
    h2 ^= (u64)K;
    h1 += h2;
    h2 += h1;
    h1 = fmix64(h1);
    h2 = fmix64(h2);
    return h1 + h2;
}

// ---- rolling window state of one lane ----
template <int K>
struct Roll {
    u64 Fm; // m-form of the forward window
    u64 F;  // l-form of the forward window
    u32 run; // number of consecutive good bases ending at the newest base

    // state after the first K-1 bases of the lane segment; s64 = l-form code stream (bases 0..31),
    // good = good bits of bases 0..31
    FH_HDM void init(u64 s64, u32 good) {
        if (K == 1) {
            Fm = 0; F = 0; run = 0;
            return;
        }
        const u64 mask = kmask(K);
        F = (s64 << 2) & mask;
        // keep bases 0..K-2, pair-reverse, align so that base K-2 is digit 0
        u64 low = (K - 1 >= 32) ? s64 : (s64 & ((1ULL << (2 * (K - 1))) - 1ULL));
        Fm = pairrev64(low) >> (64 - 2 * (K - 1));
        u32 bad = ~good & (u32)((1ULL << (K - 1)) - 1ULL);
        if (bad == 0) run = K - 1;
        else {
            const int msb = 31 - __builtin_clz(bad);
            run = (u32)(K - 2 - msb);
        }
    }
    // roll in one base (code c, good bit g)
    FH_HDM void push(u32 c, u32 g) {
        const u64 mask = kmask(K);
        Fm = ((Fm << 2) | c) & mask;
        F = (F >> 2) | ((u64)c << (2 * (K - 1)));
        run = (run + 1u) * g;
    }
    FH_HDM bool valid() const { return run >= (u32)K; }
    // canonical m-form and strand (true = reverse complement retained), canonical_kmers semantics:
    // (fwd < rc) ? fwd : rc   -- ties (even-k palindromes) report rc
    FH_HDM u64 canonical(bool &is_rc) const {
        const u64 rcm = ~F & kmask(K);
        is_rc = !(Fm < rcm);
        return is_rc ? rcm : Fm;
    }
};

// ---- synthetic data generator : splitmix64 counter RNG ----
FH_HD u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

FH_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

FH_HD uint8_t synth_genome_base(u64 seed, u64 i) {
    const u64 h = splitmix64(splitmix64(seed ^ 0x67656e6f6d65ULL /* "genome" */) + i);
    return (uint8_t) "ACGT"[h >> 62];
}

FH_HD uint8_t comp_base(uint8_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }

// byte j (0..read_len, j == read_len is the '\0' breaker) of read r
FH_HD uint8_t synth_read_byte(const uint8_t *genome, u64 genome_len, u64 r, u32 j, u32 read_len, u64 seed,
                              u32 sub_ppm, u32 n_ppm) {
    if (j >= read_len) return 0;
    const u64 h0 = splitmix64(splitmix64(seed ^ 0x7265616473ULL /* "reads" */) + r);
    const u64 span = genome_len - read_len + 1;
    const u64 start = mulhi64(splitmix64(h0), span);
    const bool rev = (h0 >> 63) != 0;
    uint8_t b = rev ? comp_base(genome[start + (read_len - 1 - j)]) : genome[start + j];
    const u64 hj = splitmix64(h0 + 0x632BE59BD9B4E019ULL * (u64)(j + 1));
    const u32 u_sub = (u32)(mulhi64(hj, 1000000ULL));                       // uniform 0..999999
    const u32 u_n = (u32)(mulhi64(splitmix64(hj), 1000000ULL));
    if (u_n < n_ppm) return 'N';
    if (u_sub < sub_ppm) {
        const u32 code = b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : 3u;
        const u32 add = 1u + (u32)((((hj >> 20) & 0xFFFFFULL) * 3ULL) >> 20); // uniform 1..3
        b = (uint8_t) "ACGT"[(code + add) & 3u];
    }
    return b;
}

} // namespace fh
