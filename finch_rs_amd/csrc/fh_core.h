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
    // gather the four 2-bit codes / four flag bits with one multiply each: the partial products land on
    // disjoint bits, the wanted ones adjacent at the top of the word
    q8 = (code * 0x01041040u) >> 24;    // c0@24 c1@26 c2@28 c3@30
    good4 = (ok * 0x00204081u) >> 28;   // b0@28 b1@29 b2@30 b3@31
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

// K = 4m+1 with the last full 4-base group in the low half of its 64-bit key word: that group and the single
// trailing base are five consecutive key bytes of ONE word, so one 1024-entry table replaces two lookups.
constexpr bool tail_merge5(int K) { return (K & 3) == 1 && K >= 5 && group_geom(K, (K + 3) / 4 - 1).hi; }

// ASCII bytes of a 5-base group (m-form, first base in byte 0) as a 40-bit value
FH_HD u64 ascii_group5(u32 q) {
    u64 w = 0;
    for (int i = 0; i < 5; ++i) {
        u32 code = (q >> (2 * (4 - i))) & 3u;
        u64 ch = code == 0 ? 0x41u : code == 1 ? 0x43u : code == 2 ? 0x47u : 0x54u;
        w |= ch << (8 * i);
    }
    return w;
}
FH_HD u64 lut_entry5(u32 q, u64 c) { return ascii_group5(q) * c; }

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
// Fm  : m-form of the forward window (bottom aligned, masked)
// Rcm : m-form of the reverse complement of the window == l-form of the complemented codes, which rolls
//       by (x >> 2) | (cbar << 2(K-1)) with no mask (the oldest digit falls off the bottom)
template <int K>
struct Roll {
    u64 Fm;
    u64 Rcm;

    // state after the first K-1 bases of the lane segment; s64 = l-form code stream of bases 0..31
    FH_HDM void init(u64 s64) {
        if (K == 1) {
            Fm = 0;
            Rcm = 0;
            return;
        }
        const u64 mask = kmask(K);
        Rcm = ((~s64) << 2) & mask;
        const u64 low = s64 & ((1ULL << (2 * (K - 1))) - 1ULL); // K-1 <= 31
        Fm = pairrev64(low) >> (64 - 2 * (K - 1));
    }
    // roll in one base (code c)
    FH_HDM void push(u32 c) {
        const u64 mask = kmask(K);
        Fm = ((Fm << 2) | c) & mask;
        Rcm = (Rcm >> 2) | ((u64)(c ^ 3u) << (2 * (K - 1)));
    }
    // canonical m-form and strand (true = reverse complement retained), canonical_kmers semantics:
    // (fwd < rc) ? fwd : rc   -- ties (even-k palindromes) report rc
    FH_HDM u64 canonical(bool &is_rc) const {
        is_rc = !(Fm < Rcm);
        return is_rc ? Rcm : Fm;
    }
};

// bit j of the result: the K-base window starting at base j lies entirely in good bases (g64 bit b = base
// b is one of ACGT).  Log-doubling AND of shifted masks; runs once per lane per tile.
template <int K>
FH_HD u32 window_valid_mask(u64 g64) {
    u64 A[6]; // A[p][j] = AND of 2^p consecutive good bits starting at j
    A[0] = g64;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int p = 1; p < 6; ++p) A[p] = A[p - 1] & (A[p - 1] >> (1 << (p - 1)));
    u64 W = ~0ULL;
    int off = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int p = 5; p >= 0; --p) {
        if (K & (1 << p)) {
            W &= (A[p] >> off);
            off += (1 << p);
        }
    }
    return (u32)W;
}

// ---- 32-bit split lookup tables for the LUT murmur (device layout: u32 T[4][256] + partial tables) ----
// TQ[0] = lo(ascii4*c1), TQ[1] = hi(ascii4*c1), TQ[2] = lo(ascii4*c2), TQ[3] = hi(ascii4*c2)
// TP[0] = lo(ascii_nb*cp), TP[1] = hi(ascii_nb*cp)  (64 entries each, nb = K & 3)
struct U64H {
    u32 lo, hi;
};

FH_HD U64H make64(u64 x) { return U64H{(u32)x, (u32)(x >> 32)}; }
FH_HD u64 join64(U64H x) { return ((u64)x.hi << 32) | x.lo; }

FH_HD u32 alignbit_b32(u32 hi, u32 lo, u32 sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (u32)(((((u64)hi) << 32) | lo) >> (sh & 31));
#endif
}

template <int R>
FH_HD U64H rotl64h(U64H x) {
    if (R == 32) return U64H{x.hi, x.lo};
    if (R < 32) return U64H{alignbit_b32(x.lo, x.hi, 32 - R), alignbit_b32(x.hi, x.lo, 32 - R)};
    return U64H{alignbit_b32(x.hi, x.lo, 64 - R), alignbit_b32(x.lo, x.hi, 64 - R)};
}

// 64-bit add / (x*5 + c) on the device as single v_lshl_add_u64 instructions
FH_HD u64 add64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a + b;
#endif
}

FH_HD u64 mul5_add(u64 h, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 t, r;
    asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(t) : "v"(h));
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(t), "s"(c));
    return r;
#else
    return h * 5 + c;
#endif
}

FH_HD u64 mul64c(U64H a, u64 C) {
    const u32 cl = (u32)C, ch = (u32)(C >> 32);
    u64 p = (u64)a.lo * cl;
    const u32 cross = a.lo * ch + a.hi * cl;
    return p + ((u64)cross << 32);
}

// murmurhash3_x64_128(ascii(canonical k-mer), seed).0 from the m-form canonical word, 32-bit split tables.
// SEED0: compile-time knowledge that seed == 0 (the default; drops three 64-bit ops).
// The first-stage products of the key words, as gathered from the tables: words [2b], [2b+1] = k1*c1, k2*c2 of
// block b; [2NB], [2NB+1] = the tail's.  Split from the rest of the hash so that the kernel can issue the
// lookups of the next position before it runs the dependent multiply chain of the current one.
template <int K>
struct KeyWords {
    static constexpr int N = 2 * (K / 16) + 2;
    u32 lo[N], hi[N];
};

// T5 (only if tail_merge5(K)): [0..1023] = lo, [1024..2047] = hi of ascii_group5 * partial_const(K)
template <int K>
FH_HD void murmur_lookup(u64 cm, const u32 *TQ, const u32 *TP, const u32 *T5, KeyWords<K> &w) {
    constexpr int NG_ALL = (K + 3) / 4;
    constexpr bool M5 = tail_merge5(K);
    constexpr int NG = M5 ? NG_ALL - 2 : NG_ALL; // groups looked up one by one
    const u32 cml = (u32)cm, cmh = (u32)(cm >> 32);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < KeyWords<K>::N; ++i) w.lo[i] = w.hi[i] = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int g = 0; g < NG; ++g) {
        const GroupGeom gg = group_geom(K, g);
        // byte offset (index * 4) of the group in a 32-bit table
        const int sh = gg.shift - 2; // want ((cm >> shift) & m) << 2
        const u32 fm = ((1u << (2 * gg.nb)) - 1u) << 2;
        u32 idx4;
        if (sh >= 32) idx4 = (cmh >> (sh - 32)) & fm;
        else if (sh >= 0 && sh + 2 * gg.nb + 2 <= 32) idx4 = (cml >> sh) & fm;
        else if (sh >= 0) idx4 = alignbit_b32(cmh, cml, (u32)sh) & fm;
        else idx4 = (cml << (-sh)) & fm;
        const u32 *T = (gg.nb == 4) ? (TQ + (gg.is_k2 ? 512 : 0)) : TP;
        const u32 hi_off = (gg.nb == 4) ? 256u : 64u;
        const u32 plo = *(const u32 *)((const char *)T + idx4);
        if (gg.hi) {
            w.hi[gg.word] += plo;
        } else {
            const u32 phi = *(const u32 *)((const char *)(T + hi_off) + idx4);
            w.lo[gg.word] += plo; // at most one lo-half group per word: no carry
            w.hi[gg.word] += phi;
        }
    }
    if (M5) {
        const GroupGeom gq = group_geom(K, NG_ALL - 2); // the lo-half quad; the merged group ends at bit 0
        const u32 idx4 = (cml & 0x3FFu) << 2;
        w.lo[gq.word] += *(const u32 *)((const char *)T5 + idx4);
        w.hi[gq.word] += *(const u32 *)((const char *)(T5 + 1024) + idx4);
    }
}

// SEED0: compile-time knowledge that seed == 0 (the default; drops three 64-bit ops).
// The hash is produced in two steps so that the hot loop can reject on high words alone.  `HashParts` are the two
// fmix64 states short of their last multiply and xor-shift:  a = ka * M2, b = kb * M2,
// hash = (a ^ a>>33) + (b ^ b>>33).  The final xor-shifts only touch the low words, so hi(hash) is hi(a) + hi(b)
// or that plus one, and
//     hash <= tau  =>  hi(a) + hi(b) + 1  (mod 2^32)  <=  hi(tau) + 1
// (for hi(tau) != 2^32-1; the wrap of the left side to 0 covers hi(a)+hi(b) = 2^32-1 with a carry).  Multiplication
// by M2 is linear mod 2^32 in the cross terms, so the sum of the two high words needs two mul_hi and two mul_lo
// instead of two full 64-bit products:
//     hi(a) + hi(b) = mulhi(ka.lo, M2.lo) + mulhi(kb.lo, M2.lo) + (ka.lo + kb.lo) * M2.hi + (ka.hi + kb.hi) * M2.lo
struct HashParts {
    u64 ka, kb;
};
constexpr u64 FMIX_M1 = 0xff51afd7ed558ccdULL, FMIX_M2 = 0xc4ceb9fe1a85ec53ULL;
FH_HD u64 fmix64_head(u64 k) {
    k ^= k >> 33;
    k *= FMIX_M1;
    k ^= k >> 33;
    return k;
}
FH_HD u32 mulhi32(u32 x, u32 y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(x, y);
#else
    return (u32)(((u64)x * y) >> 32);
#endif
}
FH_HD u64 parts_hash(HashParts p) {
    const u64 a = p.ka * FMIX_M2, b = p.kb * FMIX_M2;
    return add64(a ^ (a >> 33), b ^ (b >> 33));
}
FH_HD u32 parts_hi_plus1(HashParts p) {
    constexpr u32 M2L = (u32)FMIX_M2, M2H = (u32)(FMIX_M2 >> 32);
    const u32 al = (u32)p.ka, ah = (u32)(p.ka >> 32), bl = (u32)p.kb, bh = (u32)(p.kb >> 32);
    return mulhi32(al, M2L) + mulhi32(bl, M2L) + (al + bl) * M2H + (ah + bh) * M2L + 1u;
}
FH_HD u32 tau_hi_bound(u64 tau) { return (u32)(tau >> 32) == 0xFFFFFFFFu ? 0xFFFFFFFFu : (u32)(tau >> 32) + 1u; }

template <int K, bool SEED0>
FH_HD HashParts murmur_finish_parts(const KeyWords<K> &w, u64 seed) {
    constexpr int NB = K / 16, TAIL = K & 15;
    u64 h1 = seed, h2 = seed;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < NB; ++b) {
        u64 k1 = mul64c(rotl64h<31>(U64H{w.lo[2 * b], w.hi[2 * b]}), MURMUR_C2);
        if (SEED0 && b == 0) h1 = k1;
        else h1 ^= k1;
        h1 = join64(rotl64h<27>(make64(h1)));
        if (!(SEED0 && b == 0)) h1 = add64(h1, h2);
        h1 = mul5_add(h1, 0x52dce729ULL);
        u64 k2 = mul64c(rotl64h<33>(U64H{w.lo[2 * b + 1], w.hi[2 * b + 1]}), MURMUR_C1);
        if (SEED0 && b == 0) h2 = k2;
        else h2 ^= k2;
        h2 = join64(rotl64h<31>(make64(h2)));
        h2 = add64(h2, h1);
        h2 = mul5_add(h2, 0x38495ab5ULL);
    }
    if (TAIL > 8) {
        u64 k2 = mul64c(rotl64h<33>(U64H{w.lo[2 * NB + 1], w.hi[2 * NB + 1]}), MURMUR_C1);
        if (SEED0 && NB == 0) h2 = k2;
        else h2 ^= k2;
    }
    if (TAIL > 0) {
        u64 k1 = mul64c(rotl64h<31>(U64H{w.lo[2 * NB], w.hi[2 * NB]}), MURMUR_C2);
        if (SEED0 && NB == 0) h1 = k1;
        else h1 ^= k1;
    }
    h1 ^= (u64)K;
    h2 ^= (u64)K;
    h1 = add64(h1, h2);
    h2 = add64(h2, h1);
    return HashParts{fmix64_head(h1), fmix64_head(h2)};
}

template <int K, bool SEED0>
FH_HD u64 murmur_finish(const KeyWords<K> &w, u64 seed) {
    return parts_hash(murmur_finish_parts<K, SEED0>(w, seed));
}

// murmurhash3_x64_128(ascii(canonical k-mer), seed).0 from the m-form canonical word, 32-bit split tables.
template <int K, bool SEED0>
FH_HD u64 murmur_h1_fast(u64 cm, u64 seed, const u32 *TQ, const u32 *TP, const u32 *T5) {
    KeyWords<K> w;
    murmur_lookup<K>(cm, TQ, TP, T5, w);
    return murmur_finish<K, SEED0>(w, seed);
}

// table slot key: admitted hashes are small numbers, so mix before scaling to the table size
FH_HD u32 slot_key(u64 h) { return (u32)((h * 0x9E3779B97F4A7C15ULL) >> 32); }

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
