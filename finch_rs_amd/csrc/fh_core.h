// fh_core.h -- per-lane arithmetic of the sketch kernel (K2), written once so that the exact same
// code is compiled by hipcc for gfx950 and (for logic tests on a GPU-less box) by g++ for the host.
//
// What it restates (reference file:line, relative to the finch-rs tree):
//   classify4/classify_chunk : needletail 0.5.0 Sequence::normalize(false) restricted to the
//                              k-mer-relevant alphabet (mash.rs:73): {A,C,G,T,a,c,g,t,u,U} -> 2-bit codes
//                              (u/U -> T), every other byte breaks k-mers.  Whitespace never reaches the
//                              device (stripped while staging, see fh_api).
//   Windows<K>::canonical    : needletail canonical_kmers (mash.rs:76): fwd < rc ? (fwd,false) : (rc,true)
//   murmur_h1_fast<K>        : murmurhash3 0.0.5 murmurhash3_x64_128(kmer, seed).0 (hashing.rs:10-12) on the
//                              ASCII bytes of the canonical k-mer -- the kernel's form (lookup tables with both
//                              key-word stages folded in, see "lookup tables" below)
//   murmur_h1_lut<K>         : the same hash from plain first-stage tables; kept as an independent form that the
//                              host logic test cross-checks the kernel's form against (not used on the device)
//
// Layouts.  A k-mer is held as a 2-bit word, A=0 C=1 G=2 T=3 (ASCII order == code order):
//   "m-form" (MSB-first): first base in the most significant used digit -> numeric order == lexicographic
//   "l-form" (LSB-first): base i at bits [2i+1:2i]
// For a forward window with l-form F and m-form Fm:  m-form of its reverse complement == ~F & mask.
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

// Classification of 4 ASCII bytes (needletail normalize(false) + canonical_kmers: ACGT, acgt and U/u are bases,
// every other byte breaks k-mers).  The low three bits of the five accepted upper-case letters are distinct
// (A 001, C 011, T 100, U 101, G 111), so they index two 8-entry byte tables held in registers (v_perm_b32): the
// letter the byte has to be, and its 2-bit code (A 0, C 1, G 2, T/U 3).
//   qtop : the four codes gathered into the TOP byte (base i at bits [24+2i, 26+2i)); lower bytes are scrap
//   vf   : bit 8i+7 set iff byte i IS a base; all other bits clear
// diff = the case-folded byte xor the letter it has to be: zero iff the byte is a base.  Its zero bytes are found with
// (diff - 0x01010101) & ~diff & 0x80808080 -- two instructions where the carry-free form ((x & 0x7f..) + 0x7f.. | x) takes
// three.  That test is wrong only for a byte 0x01 above a zero byte (the borrow), and a byte of diff is never 0x01: where
// the table holds a letter the low three bits of diff are zero (the letter was picked by them), where it holds 0xFF bit 5
// of diff is set (the fold cleared it in the byte).  tests/test_core_logic_host.py walks every pair of neighbouring bytes.
FH_HD void classify4(u32 d, u32 &qtop, u32 &vf) {
    const u32 sel = d & 0x07070707u;
    const u32 expect = perm_b32(0x47FF5554u, 0x43FF41FFu, sel); // idx 0..7: -, A, -, C, T, U, -, G  (0xFF = none)
    const u32 code = perm_b32(0x02000303u, 0x01000000u, sel);   // idx 0..7: 0, 0, 0, 1, 3, 3, 0, 2
    const u32 diff = (d & 0xDFDFDFDFu) ^ expect;                // case folded; a byte is 0 iff it is a base
    vf = (diff - 0x01010101u) & ~diff & 0x80808080u;
    // gather with one multiply: the partial products land on disjoint bits, the wanted ones adjacent on top
    qtop = code * 0x01041040u; // c0@24 c1@26 c2@28 c3@30
}

// 16 bytes (4 dwords, little endian) -> 16 codes (32 bits, l-form) + 16 good bits
FH_HD void classify_chunk(u32 d0, u32 d1, u32 d2, u32 d3, u32 &codes, u32 &good) {
    u32 q0, q1, q2, q3, v0, v1, v2, v3;
    classify4(d0, q0, v0);
    classify4(d1, q1, v1);
    classify4(d2, q2, v2);
    classify4(d3, q3, v3);
    // pack the four top bytes: selector 3 / 7 = top byte of the second / first operand, 0x0c = zero
    codes = perm_b32(q1, q0, 0x0c0c0703u) | perm_b32(q3, q2, 0x07030c0cu);
    // the flags of TWO dwords, interleaved four bits apart (byte j of the first at bit 8j+3, of the second at 8j+7), are
    // gathered by ONE multiply: 2^21 + 2^14 + 2^7 + 1 brings them to bits 24+j and 28+j, and no two of the 32 partial
    // products meet (8j' - 7j + {0, 4}, j != j', lies outside 0..7 and the two families differ by 4 mod 7 != 0)
    const u32 w0 = (v0 >> 4) | v1, w1 = (v2 >> 4) | v3;
    good = ((w0 * 0x00204081u) >> 24) | (((w1 * 0x00204081u) >> 16) & 0xFF00u);
}

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

// reverse the order of the 16 2-bit digits of a dword
FH_HD u32 pairrev32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 r = __builtin_bitreverse32(x);
#else
    u32 r = 0;
    for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
#endif
    return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
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

// Entry of the plain first-stage tables (64-bit; used by murmur_h1_lut, the host-side cross-check form)
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
        if (gg.hi) {
            u32 plo = ((const u32 *)T)[2 * q]; // low dword of the 64-bit entry (little endian)
            wc[gg.word] += (u64)plo << 32;
        } else {
            wc[gg.word] += T[q];
        }
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

// funnel shift right: low 32 bits of {hi:lo} >> sh
FH_HD u32 alignbit_b32(u32 hi, u32 lo, u32 sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (u32)(((((u64)hi) << 32) | lo) >> (sh & 31));
#endif
}

// bit field extract (x >> off) & (2^width - 1), width < 32: one v_bfe_u32 (left to itself the compiler widens the
// shift to 64 bits)
FH_HD u32 bfe_u32(u32 x, u32 off, u32 width) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(x, off, width);
#else
    return (x >> off) & ((1u << width) - 1u);
#endif
}

// ---- the K-base windows of one lane ----
// A lane owns 32 start positions and sees 64 bases (its own 32 + the neighbour's) as 2-bit codes in l-form
// (base b at bits [2b, 2b+2) of clo | chi << 64).  Both strands' m-form words of every window are bit fields of
// two 128-bit strings built once per tile, so a window costs two funnel shifts per strand and no window depends
// on the previous one:
//   nC = ~codes            : window j's reverse complement in m-form is bits [2j, 2j + 2K) of nC
//                            (complement = 3 - c; reversing the order turns l-form into m-form)
//   D  = digit-reversed    : base b at digit 63 - b; window j's forward m-form is bits [2(64-K-j), +2K) of D
// canonical(): (fwd < rc) ? fwd : rc -- ties (even-k palindromes) report rc, as needletail's canonical_kmers.
// The canonical word leaves here shifted left by pre_shift(K) bits (its low pre_shift bits are scrap): that puts every
// 4-base group of the murmur3 key on a byte boundary of the word, and a byte of a register times the record size is ONE
// instruction (v_lshlrev_b32_sdwa) where a bit field is a shift and a mask or a funnel shift and a mask.
// For K = 23..32 the shift is the whole room the word leaves, 64 - 2K (the same modulo 8): the window then fills its two
// registers to the top, each half is ONE funnel shift of the string, and the mask of the upper half (a v_bfe_u32, or a
// funnel shift and a v_and_b32, or the 64-bit shift the compiler makes of them) is gone: 1.7 VALU instructions per position
// fewer at k = 21 in the ISA.  Measured (profiles/r04j_ab_pre_wide.txt): k = 25 +1.8 %, k = 23, 24, 28 +0.5...0.7 %.  K = 17..22
// run their 32 positions as one unrolled pass, which spills 10-23 registers with the wide shift, and in two rounds of 16 the
// gain is what the rounds cost: they keep the smallest shift, as do shorter k-mers (their strings would move by 32 bits and
// more) and K = 33..64 (WindowsW).
// Round 5: where the pre-shifted word leaves the top two bits of its register pair clear (2K + PRE <= 62: every K <= 28 with
// the SMALLEST shift), min(fwd, rc) is ONE v_min_f64 on the bit patterns -- sign 0, exponent never all ones, so the doubles
// order exactly as the integers do (denormals are kept: the kernels run with the default f64 denormal mode; tools/ubench.hip
// checks the instruction against the integer minimum bit for bit) -- instead of v_cmp_lt_u64 and two v_cndmask_b32.  The
// strand flag is not needed to hash: the (rare) admit path works it out again (Windows::strand_of).  That is worth more than
// the wide shift's missing masks, so K = 23..28 go back to the smallest shift (FH_MINF64=0: round 4's windows).
#ifndef FH_MINF64
#define FH_MINF64 1
#endif
#ifndef FH_PRE_WIDE_FROM
#define FH_PRE_WIDE_FROM (FH_MINF64 ? 29 : 23)
#endif
constexpr int pre_shift(int K) { // 2K + pre_shift <= 64 for every K <= 32
    return (K >= FH_PRE_WIDE_FROM && K >= 17 && K <= 32) ? 64 - 2 * K : (8 - (2 * K) % 8) % 8;
}
// min of two 64-bit words whose top two bits are clear
FH_HD u64 min_u62(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a < b ? a : b;
#endif
}

// DOFF (fh_k2s.hip's long rounds): the digit-reversed view reaches DOFF bases further (base b at digit 63 + DOFF - b, the
// view's fifth word in use), so that the forward window's bit field -- whose low PRE bits are the bases behind the window --
// stays inside the string for every window the 64-base view holds, j <= 64 - K, when DOFF = PRE / 2.
template <int K, int DOFF = 0>
struct Windows {
    static constexpr int PRE = pre_shift(K), NB = 2 * K + PRE;
    static_assert(DOFF == 0 || 2 * DOFF == PRE, "the long view reaches just as far as the window's scrap bits");
    u32 nC[5], D[5];

    FH_HDM void init(u64 clo, u64 chi) {
        const u32 c0 = ~(u32)clo, c1 = ~(u32)(clo >> 32), c2 = ~(u32)chi, c3 = ~(u32)(chi >> 32);
        if (PRE == 0) {
            nC[0] = c0;
            nC[1] = c1;
            nC[2] = c2;
            nC[3] = c3;
            nC[4] = 0;
        } else { // the reverse-complement string is kept shifted left by PRE bits, so its windows start at >= 0
            nC[0] = c0 << PRE;
            nC[1] = alignbit_b32(c1, c0, 32 - PRE);
            nC[2] = alignbit_b32(c2, c1, 32 - PRE);
            nC[3] = alignbit_b32(c3, c2, 32 - PRE);
            nC[4] = c3 >> (32 - PRE);
        }
        const u64 rl = pairrev64(chi), rh = pairrev64(clo);
        D[0] = (u32)rl;
        D[1] = (u32)(rl >> 32);
        D[2] = (u32)rh;
        D[3] = (u32)(rh >> 32);
        D[4] = 0;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FH_D_FOLDABLE)
        // The words are made opaque: knowing that D[2], D[3] are the halves of ONE 64-bit value, the compiler folds the upper-half
        // bit field of a forward window into a 64-bit shift of the pair (v_lshrrev_b64 + v_and_b32 where one v_bfe_u32 does:
        // k = 21 58.8 -> 58.0 VALU instructions per position; k = 17 +1.2 %, k = 21 +0.5 %, k = 22 +0.7 %, profiles/r04k_ab_d_opaque.txt)
        for (int i = 0; i < 4; ++i) asm("" : "+v"(D[i]));
#endif
    }
    // The two strings given as they are (fh_k2s.hip cuts them out of a tile's strings in LDS): nc = the lane's 64-base view
    // complemented and shifted left by PRE bits (bit i of nc = bit i - PRE of ~codes; 5 words), d = the view digit-reversed
    // (base b at digit 63 - b; 4 words).
    // With DOFF the view has a fifth word d[4] (its low 2 DOFF bits are looked at).
    FH_HDM void init_words(const u32 *nc, const u32 *d) {
        for (int i = 0; i < 5; ++i) nC[i] = nc[i];
        for (int i = 0; i < 4; ++i) D[i] = d[i];
        D[4] = DOFF ? d[4] : 0;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FH_D_FOLDABLE)
        for (int i = 0; i < (DOFF ? 5 : 4); ++i) asm("" : "+v"(D[i]));
#endif
    }
    // bits [off, off + NB) of the string W (off, K compile-time after unrolling)
    static FH_HDM u64 field(const u32 *W, int off) {
        const int w = off >> 5, s = off & 31, nbits = NB;
        u32 lo = s ? alignbit_b32(W[w + 1], W[w], (u32)s) : W[w];
        u32 hi = 0;
        if (nbits < 32) lo &= (1u << nbits) - 1u;
        if (nbits > 32) {
            const int hb = nbits - 32; // 2..32
            const u32 hm = hb >= 32 ? 0xFFFFFFFFu : ((1u << hb) - 1u);
            if (s + hb <= 32) hi = (hb >= 32) ? W[w + 1] : bfe_u32(W[w + 1], (u32)s, (u32)hb);
            else hi = alignbit_b32(W[w + 2], W[w + 1], (u32)s) & hm;
        }
        return ((u64)hi << 32) | lo;
    }
    FH_HDM u64 fwd(int j) const { return field(D, 2 * (64 + DOFF - K - j) - PRE); } // low PRE bits: scrap (later bases)
    // A k-mer can equal its reverse complement only for even K.  There the reverse complement's scrap bits are cleared
    // (the forward word's are not), so fwd' < rc' exactly when fwd < rc and equal words compare as "not less": the tie
    // goes to rc, as it must.  For odd K the words differ above the scrap bits and the scrap cannot matter.
    FH_HDM u64 rc(int j) const {
        const u64 r = field(nC, 2 * j);
        return (K % 2 == 0) ? (r & ~((1ULL << PRE) - 1ULL)) : r;
    }
    // Move the lane's view R positions on (R = 8 or 16): afterwards window j of this object is what window j + R was.  The
    // kernels of K >= 25 run 32 / R rounds of R unrolled positions, so the bit-field offsets stay compile-time constants
    // while only R positions' worth of lookups and hash states are in flight (register budget: fh_k2.hip).  nC moves down 2R
    // bits, D up 2R bits; window j reads bits below 128 - 2j of D and nothing a later window needs leaves at the top.
    template <int R>
    FH_HDM void advance() {
        static_assert(R == 8 || R == 16, "rounds of 8 or 16 positions");
        if (R == 16) {
            nC[0] = nC[1], nC[1] = nC[2], nC[2] = nC[3], nC[3] = nC[4], nC[4] = 0;
            D[3] = D[2], D[2] = D[1], D[1] = D[0], D[0] = 0;
        } else {
            for (int i = 0; i < 4; ++i) nC[i] = alignbit_b32(nC[i + 1], nC[i], 16);
            nC[4] >>= 16;
            for (int i = 3; i > 0; --i) D[i] = alignbit_b32(D[i], D[i - 1], 16);
            D[0] <<= 16;
        }
    }
    // the canonical word is formed without the strand flag (canonical_word) and the flag recovered on the admit path (strand_of)
    static constexpr bool MINF64 = FH_MINF64 && NB <= 62;
    // the canonical m-form word << PRE alone
    FH_HDM u64 canonical_word(int j) const {
#ifdef FH_EXP_FWD_ONLY // measurement only -- WRONG sketches: the ceiling of any cheaper strand decision (profiles/r03_*)
        return fwd(j);
#endif
        const u64 f = fwd(j), r = rc(j);
        if (MINF64) return min_u62(f, r); // (even K: rc's scrap bits are cleared, so equal k-mers give rc's word -- either is the k-mer)
        return (f < r) ? f : r;
    }
    // Was window j's canonical word the reverse complement's (ties -> yes)?  Worked out again from the strings, from copies the
    // compiler cannot connect with the hot loop's own windows: what it could share it would keep alive from the loop into the
    // branch, in registers the loop does not have.
    FH_HDM bool strand_of(int j) const {
        u32 d[5], c[5];
        for (int i = 0; i < 5; ++i) d[i] = D[i], c[i] = nC[i];
#if defined(__HIP_DEVICE_COMPILE__)
        for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(d[i]), "+v"(c[i]));
#endif
        const u64 f = field(d, 2 * (64 + DOFF - K - j) - PRE);
        u64 r = field(c, 2 * j);
        if (K % 2 == 0) r &= ~((1ULL << PRE) - 1ULL);
        return !(f < r);
    }
    // the canonical m-form word << PRE
    FH_HDM u64 canonical(int j, bool &is_rc) const {
#ifdef FH_EXP_FWD_ONLY
        is_rc = false;
        return fwd(j);
#endif
        if (MINF64) {
            is_rc = strand_of(j);
            return canonical_word(j);
        }
        const u64 f = fwd(j), r = rc(j);
        is_rc = !(f < r);
        return is_rc ? r : f;
    }
};

// bit j of the result: the K-base window starting at base j lies entirely in good bases (g64 bit b = base
// b is one of ACGT).  Log-doubling AND of shifted masks; runs once per lane per tile.
template <int K>
FH_HD u64 window_valid_mask64(u64 g64) {
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
    return W; // (bits j > 64 - K are clear: their windows reach behind the 64 bases)
}
template <int K>
FH_HD u32 window_valid_mask(u64 g64) { return (u32)window_valid_mask64<K>(g64); }

// ---- the segment kernel's views (fh_k2s.hip) ----
// A wave's tile of NCH 16-byte chunks lies in LDS as three strings: Fc = the complemented codes (chunk i at word 1 + i, word 0
// and the words behind the string zero), Rv = the digit-reversed codes (chunk i as pairrev32 at word NCH - 1 - i: the base at
// tile position p is digit 16 NCH - 1 - p), Gd = the good bits (chunk i at half-word i).  A lane's round looks at the 64
// bases from tile position p0 on (p0 + 63 + DOFF < 16 NCH): nc / d are what Windows<K, DOFF>::init_words takes, seg_good_bits
// what window_valid_mask64 takes.  Positions per round: as many windows as the view holds (65 - K, at most 48) for K <= 24 and 26,
// 16 for longer k-mers (their register budget: fh_k2.hip).
#ifndef FH_SEG_LONG
#define FH_SEG_LONG 1 // 0: rounds of 32 for K <= 22 (round 5's first form)
#endif
// (long rounds wherever the unrolled pass keeps inside 128 VGPRs: K = 25, 27, 28 spill 18-30 registers with it, K = 29..32 32)
#ifndef FH_SEG_LONG_MAXK
#define FH_SEG_LONG_MAXK 24
#endif
constexpr bool seg_long(int K) { return FH_SEG_LONG && (K <= FH_SEG_LONG_MAXK || (K == 26 && FH_SEG_LONG_MAXK == 24)); }
constexpr int seg_round(int K) { return seg_long(K) ? (65 - K < 48 ? 65 - K : 48) : (K >= 23 ? 16 : 32); }
constexpr int seg_doff(int K) { return seg_long(K) ? pre_shift(K) / 2 : 0; }

template <int K>
FH_HD void seg_cut_views(const u32 *Fc, const u32 *Rv, u32 NCH, u32 p0, u32 *nc, u32 *d) {
    constexpr int PRE = pre_shift(K), DOFF = seg_doff(K), ND = DOFF ? 5 : 4;
    const int gc = (int)(2u * p0) - PRE; // first bit of the complemented string's view, shifted left by PRE
    const u32 ic = (u32)((gc >> 5) + 1), sc = (u32)gc & 31u;
    u32 f[6];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int w = 0; w < 6; ++w) f[w] = Fc[ic + (u32)w];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int w = 0; w < 5; ++w) nc[w] = alignbit_b32(f[w + 1], f[w], sc);
    const u32 gd = 2u * (16u * NCH - 64u - (u32)DOFF - p0); // first bit of the digit-reversed string's view (its base 63 + DOFF)
    const u32 id = gd >> 5, sd = gd & 31u;
    u32 r[ND + 1];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int w = 0; w < ND + 1; ++w) r[w] = Rv[id + (u32)w];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int w = 0; w < ND; ++w) d[w] = alignbit_b32(r[w + 1], r[w], sd);
}
FH_HD u64 seg_good_bits(const u32 *Gd, u32 p0) {
    const u32 gi = p0 >> 5, gs = p0 & 31u;
    const u32 g0 = Gd[gi], g1 = Gd[gi + 1u], g2 = Gd[gi + 2u];
    return (u64)alignbit_b32(g1, g0, gs) | ((u64)alignbit_b32(g2, g1, gs) << 32);
}

// ---- lookup tables with murmur3's second stage folded in ----
// A key word x (8 key bytes: group A = low 4 bytes, group B = high 4 bytes, either possibly short) enters the
// hash as  kx = rotl(x * c, R) * C  with (c, R, C) = (c1, 31, c2) for k1 words and (c2, 33, c1) for k2 words.
// x * c = u + (v << 32) with u = ascii(A) * c (64 bit) and v = lo32(ascii(B) * c): the first multiply is linear in
// the groups, so u and v come from tables indexed by the 2-bit codes of A and B.  The rotate and second multiply
// are folded into the tables as far as linearity reaches; w = hi(u) + v (mod 2^32) is the only non-linear coupling:
//   R = 33:  rotl(x,33) = (u << 33) + 2w + (lo(u) >> 31)            =>  kx = U2[A] + w * (2C)
//            U2[A] = ((u << 33) + (lo(u) >> 31)) * C
//   R = 31:  rotl(x,31) = (u << 31) + (v << 63) + (w >> 1)          =>  kx = U1[A] + ((v << 31) << 32) + (w >> 1) * C
//            U1[A] = (u << 31) * C          (C odd: (v << 63) * C = v << 63)
//   no B  :  kx = F[A] straight from the table (A up to five bases: a lone trailing base is merged into A).
// That is one 32x32->64 multiply-add, one mul_lo and a few adds per two-group word instead of a rotate (2 alignbit)
// and a full 64-bit multiply (4 multiplier ops), and nothing at all for the single-group tail word.
// Records are laid out for one LDS access per group: A records 16 B {lo(U), hi(U), hi(u), -}, B records 8 B
// {v, v << 31}, single-group records 8 B {lo(kx), hi(kx)} (an 8-byte LDS read costs what a 4-byte one does).
struct alignas(16) Rec4 {
    u32 x, y, z, w;
};
struct alignas(8) Rec2 {
    u32 x, y;
};

struct U64H {
    u32 lo, hi;
};

FH_HD U64H make64(u64 x) { return U64H{(u32)x, (u32)(x >> 32)}; }
FH_HD u64 join64(U64H x) { return ((u64)x.hi << 32) | x.lo; }

template <int R>
FH_HD U64H rotl64h(U64H x) {
    if (R == 32) return U64H{x.hi, x.lo};
    if (R < 32) return U64H{alignbit_b32(x.lo, x.hi, 32 - R), alignbit_b32(x.hi, x.lo, 32 - R)};
    return U64H{alignbit_b32(x.hi, x.lo, 64 - R), alignbit_b32(x.lo, x.hi, 64 - R)};
}

// 64-bit add and h * 5 + c as v_lshl_add_u64.  The compiler selects the instruction for plain adds by itself, but the
// pinned form schedules measurably better in the unrolled position loop (A/B on MI355X: +1.4 %); h * 5 + c would be
// turned into two v_mad_u64_u32 and a move.
FH_HD u64 add64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a + b;
#endif
}

// a ^ b ^ c with c confined to the low word
FH_HD u64 xor3_lo(u64 a, u64 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(lo) : "v"((u32)a), "v"((u32)b), "v"(c));
    return ((u64)((u32)(a >> 32) ^ (u32)(b >> 32)) << 32) | lo;
#else
    return a ^ b ^ (u64)c;
#endif
}

FH_HD u64 mul5_add(u64 h, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 t;
    asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(t) : "v"(h));
    return t + c;
#else
    return h * 5 + c;
#endif
}

// geometry of key word i of a K-byte key (words 2b, 2b+1 = k1, k2 of block b; 2NB, 2NB+1 = the tail's)
struct WordGeom {
    int kind;   // 0 = absent, 1 = single group (table holds the finished word), 2 = two groups
    int nbA, nbB; // bases in the low / high group (nbA = 5: trailing base merged into the low group)
    int shiftA, shiftB; // right shift of the m-form canonical word that brings the group to bit 0
    bool is_k2;
    bool partial; // the word's last group is looked up in the per-K table P
};
constexpr WordGeom word_geom(int K, int i) {
    WordGeom r{};
    const int NB = K / 16;
    const int off = (i < 2 * NB) ? 8 * i : 16 * NB + 8 * (i - 2 * NB);
    const int rem = K - off;
    r.is_k2 = (i & 1) != 0;
    r.nbA = rem <= 0 ? 0 : (rem < 4 ? rem : 4);
    r.nbB = rem <= 4 ? 0 : (rem - 4 < 4 ? rem - 4 : 4);
    if (r.nbB == 1) {
        r.nbA = 5;
        r.nbB = 0;
    }
    r.shiftA = 2 * (K - off - r.nbA);
    r.shiftB = r.nbB ? 2 * (K - off - 4 - r.nbB) : 0;
    r.kind = r.nbA == 0 ? 0 : (r.nbB == 0 ? 1 : 2);
    r.partial = r.kind == 1 || (r.kind == 2 && r.nbB < 4);
    return r;
}
constexpr int n_key_words(int K) { return 2 * (K / 16) + 2; }
// the one word of a key that uses the per-K table P (always the last one present), -1 if none (K % 8 == 0)
constexpr int partial_word(int K) {
    for (int i = 0; i < n_key_words(K); ++i)
        if (word_geom(K, i).kind != 0 && word_geom(K, i).partial) return i;
    return -1;
}
// entries of P: finished words of a single group (4^nbA) or B records of a short high group (4^nbB)
constexpr int partial_entries(int K) {
    const int i = partial_word(K);
    if (i < 0) return 1;
    const WordGeom g = word_geom(K, i);
    return g.kind == 1 ? (1 << (2 * g.nbA)) : (1 << (2 * g.nbB));
}
constexpr bool has_pair_word(int K, bool k2) {
    for (int i = 0; i < n_key_words(K); ++i)
        if (word_geom(K, i).kind == 2 && word_geom(K, i).is_k2 == k2) return true;
    return false;
}

// ASCII bytes of an nb-base group (nb <= 5), first base in byte 0, from its m-form digits
FH_HD u64 ascii_group_n(u32 q, int nb) {
    u64 w = 0;
    for (int i = 0; i < nb; ++i) {
        const u32 code = (q >> (2 * (nb - 1 - i))) & 3u;
        const u64 ch = code == 0 ? 0x41u : code == 1 ? 0x43u : code == 2 ? 0x47u : 0x54u;
        w |= ch << (8 * i);
    }
    return w;
}
FH_HD u64 rotl64c(u64 x, int r) { return (x << r) | (x >> (64 - r)); }

// table builders (once per workgroup into LDS; on the host for the logic tests)
FH_HD Rec4 lut_rec_A(u32 q, bool k2) {
    const u64 u = ascii_group_n(q, 4) * (k2 ? MURMUR_C2 : MURMUR_C1);
    const u64 U = k2 ? ((u << 33) + (u64)((u32)u >> 31)) * MURMUR_C1 : (u << 31) * MURMUR_C2;
    return Rec4{(u32)U, (u32)(U >> 32), (u32)(u >> 32), 0u};
}
FH_HD Rec2 lut_rec_B(u32 q, int nb, bool k2) {
    const u32 v = (u32)(ascii_group_n(q, nb) * (k2 ? MURMUR_C2 : MURMUR_C1));
    // k1 words need v << 31 as the addend of a multiply-add (key_word_mix): it comes first, so that the register pair the
    // record is loaded into is that instruction's accumulator as it stands; k2 words only ever read v
    return k2 ? Rec2{v, v << 31} : Rec2{v << 31, v};
}
// xr: a constant folded in by xor (the key length, which murmur3 xors into h1 / h2 right after the tail words)
FH_HD Rec2 lut_rec_S(u32 q, int nb, bool k2, u64 xr) {
    const u64 x = ascii_group_n(q, nb) * (k2 ? MURMUR_C2 : MURMUR_C1);
    const u64 kx = (k2 ? rotl64c(x, 33) * MURMUR_C1 : rotl64c(x, 31) * MURMUR_C2) ^ xr;
    return Rec2{(u32)kx, (u32)(kx >> 32)};
}
// the single-group word of a key is a tail word (K % 16 != 0 there), so `h ^= K` can ride in its table
constexpr bool len_folded(int K, bool k2) {
    const int i = 2 * (K / 16) + (k2 ? 1 : 0);
    return word_geom(K, i).kind == 1;
}
template <int K>
FH_HD Rec2 lut_rec_P(u32 q) {
    constexpr int i = partial_word(K);
    if (i < 0) return Rec2{0u, 0u};
    constexpr WordGeom g = word_geom(K, i < 0 ? 0 : i);
    return g.kind == 1 ? lut_rec_S(q, g.nbA, g.is_k2, (u64)K) : lut_rec_B(q, g.nbB, g.is_k2);
}

struct LutTables {
    const Rec4 *A1, *A2; // 256 A records for k1 / k2 words
    const Rec2 *B1, *B2; // 256 B records for k1 / k2 words (full 4-base high group)
    const Rec2 *P;       // partial_entries(K) records for the key's last word
    // A1 sixteen-fold replicated (murmur_lookup<K, 16>): entry q's replica r sits at byte (q << 8) + (r << 4), and a lane
    // reads replica (lane & 15) -- a1_lane_off = (lane & 15) << 4.  The sixteen lanes of every lane group of a ds_read_b128
    // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) differ in lane & 15, so they read sixteen different 16-byte slots
    // of the 256-byte bank row whatever their indices: no bank conflicts (a random-index 16-byte lookup costs 9.6 LDS cycles
    // per wave otherwise, 4 then).  64 KB instead of 4: affordable once one 1024-thread workgroup per CU shares the tables.
    u32 a1_lane_off = 0;
};

// What the lookups of one position return, kept raw so that the kernel can issue the loads of the next position
// before it runs the dependent arithmetic of the current one.
template <int K>
struct KeyWords {
    static constexpr int N = n_key_words(K);
    u32 a0[N], a1[N], a2[N], b0[N], b1[N];
};

// (byte `b` of x) << lg in one instruction
FH_HD u32 byte_shl(u32 x, int b, int lg) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 r;
    const u32 amount = (u32)lg;
    if (b == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(amount), "v"(x));
    else if (b == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(amount), "v"(x));
    else if (b == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(amount), "v"(x));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(amount), "v"(x));
    return r;
#else
    return ((x >> (8 * b)) & 0xFFu) << lg;
#endif
}

// ((byte `b` of x) << 8) | low: the offset of a sixteen-fold replicated 16-byte record, `low` = the lane's replica offset
// (< 256) in byte 0 and, in byte 2, bits 16-23 of the table's address -- ONE instruction like byte_shl (v_perm_b32 picks the byte into bits 8-15 and the low byte from the second source)
FH_HD u32 byte_shl8_or(u32 x, int b, u32 low) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_perm_b32 D, S0, S1, sel: bytes 0-3 of the pool are S1's, 4-7 S0's; selector byte 0x0C = constant 0
    // (byte 2 of the result is byte 2 of `low`: the caller may park the table's 64 KB-aligned base there)
    return __builtin_amdgcn_perm(x, low, 0x0C020000u | ((u32)(4 + b) << 8));
#else
    return (((x >> (8 * b)) & 0xFFu) << 8) | (low & 0x00FF00FFu);
#endif
}

// byte offset ((cm >> shift) & (4^nb - 1)) << lg of a record; cm = canonical word << pre_shift(K), shift includes it
FH_HD u32 field_off(u32 cml, u32 cmh, int shift, int nb, int lg) {
    if (nb == 4 && (shift & 7) == 0) return byte_shl(shift < 32 ? cml : cmh, (shift >> 3) & 3, lg);
    const int sh = shift - lg;
    const u32 fm = ((1u << (2 * nb)) - 1u) << lg;
    if (sh >= 32) return (cmh >> (sh - 32)) & fm;
    if (sh >= 0 && shift + 2 * nb <= 32) return (cml >> sh) & fm;
    if (sh >= 0) return alignbit_b32(cmh, cml, (u32)sh) & fm;
    return (cml << (-sh)) & fm;
}

// one A record as ONE 16-byte load (a vector load: the compiler splits a load of the struct into an 8- and a 4-byte -- or two
// 8-byte -- LDS reads when it sees that the fourth dword is unused, and every one of those costs what the 16-byte read does)
FH_HD Rec4 load_rec4(const Rec4 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(const u32x4 *)p;
    return Rec4{v.x, v.y, v.z, v.w};
#else
    return *p;
#endif
}

template <int K, int A1REP = 1>
FH_HD void murmur_lookup(u64 cm, const LutTables &T, KeyWords<K> &w) { // cm = canonical word << pre_shift(K)
    static_assert(A1REP == 1 || A1REP == 16, "A1 is plain or sixteen-fold replicated");
    constexpr int PRE = pre_shift(K);
    const u32 cml = (u32)cm, cmh = (u32)(cm >> 32);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < KeyWords<K>::N; ++i) {
        const WordGeom g = word_geom(K, i);
        w.a0[i] = w.a1[i] = w.a2[i] = w.b0[i] = w.b1[i] = 0;
#ifdef FH_EXP_NO_LDS // measurement only -- WRONG hashes: the records are made of their offsets, no table is read (what the lookups cost the loop)
        if (g.kind == 1) {
            w.a0[i] = field_off(cml, cmh, g.shiftA + PRE, g.nbA, 3);
            w.a1[i] = cmh;
        } else if (g.kind == 2) {
            w.a0[i] = field_off(cml, cmh, g.shiftA + PRE, 4, 4);
            w.a1[i] = cml;
            w.a2[i] = cmh;
            w.b0[i] = field_off(cml, cmh, g.shiftB + PRE, g.nbB, 3);
            w.b1[i] = cml;
        }
        continue;
#endif
        if (g.kind == 1) {
            const Rec2 r = *(const Rec2 *)((const char *)T.P + field_off(cml, cmh, g.shiftA + PRE, g.nbA, 3));
            w.a0[i] = r.x;
            w.a1[i] = r.y;
        } else if (g.kind == 2) {
            u32 a_off;
            if (A1REP == 16 && !g.is_k2) {
                const int sh = g.shiftA + PRE; // (a full group of the pre-shifted word is a byte of a register)
                a_off = byte_shl8_or(sh < 32 ? cml : cmh, (sh >> 3) & 3, T.a1_lane_off);
            } else {
                a_off = field_off(cml, cmh, g.shiftA + PRE, 4, 4);
            }
            const Rec4 ra = load_rec4((const Rec4 *)((const char *)(g.is_k2 ? T.A2 : T.A1) + a_off));
#if defined(__HIP_DEVICE_COMPILE__)
            // A 16-byte LDS read costs 9.5 cycles per wave, the 12-byte read the compiler narrows this to costs 16
            // (tools/ubench_lds.hip), so the unused fourth dword is kept "live".  Measured: k = 14-16 +3 %, k = 21 +2.7 %,
            // k = 24 +12 %, k = 31 +10 % (all of these kernels are close to the LDS pipe's limit).
            asm volatile("" ::"v"(ra.w));
#endif
            const Rec2 *TB = g.partial ? T.P : (g.is_k2 ? T.B2 : T.B1);
            const Rec2 rb = *(const Rec2 *)((const char *)TB + field_off(cml, cmh, g.shiftB + PRE, g.nbB, 3));
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FH_NO_B64_KEEP)
            // likewise an 8-byte read (6.5 cycles per wave under random-index bank conflicts) beats the 4-byte one (9) the
            // compiler narrows a k2 word's B record to -- only its first dword is used there (tools/ubench_lds.hip)
            if (g.is_k2) asm volatile("" ::"v"(rb.y));
#endif
            w.a0[i] = ra.x;
            w.a1[i] = ra.y;
            w.a2[i] = ra.z;
            w.b0[i] = g.is_k2 ? rb.x : rb.y;
            w.b1[i] = g.is_k2 ? rb.y : rb.x;
        }
    }
}

// a * b + c (32 x 32 + 64 -> 64): one v_mad_u64_u32, pinned so that the addend stays the register pair the table
// record was loaded into
FH_HD u64 mad64(u32 a, u32 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "s"(b), "v"(c));
    return r;
#else
    return (u64)a * b + c;
#endif
}

// kx = rotl(x * c, R) * C of key word i from its records (see the derivation above)
template <int K>
FH_HD u64 key_word_mix(const KeyWords<K> &w, int i) {
    const WordGeom g = word_geom(K, i);
    const u64 acc = ((u64)w.a1[i] << 32) | w.a0[i];
    if (g.kind == 1) return acc;
    const u32 ww = w.a2[i] + w.b0[i];
    if (g.is_k2) {
        constexpr u64 M = MURMUR_C1 << 1;
        const u64 t = mad64(ww, (u32)M, acc);
        return ((u64)((u32)(t >> 32) + ww * (u32)(M >> 32)) << 32) | (u32)t; // only the high word takes the cross term
    }
    const u32 y = ww >> 1;
#if defined(__HIP_DEVICE_COMPILE__)
    // high word = hi(y * C.lo + acc) + y * C.hi + b1: the last two are the low word of a second multiply-add on the B record
    const u64 t = mad64(y, (u32)MURMUR_C2, acc);
    const u64 q = mad64(y, (u32)(MURMUR_C2 >> 32), ((u64)w.b0[i] << 32) | w.b1[i]);
    return ((u64)((u32)(t >> 32) + (u32)q) << 32) | (u32)t;
#else
    return mad64(y, (u32)MURMUR_C2, acc) + ((u64)(y * (u32)(MURMUR_C2 >> 32) + w.b1[i]) << 32);
#endif
}

// SEED0: compile-time knowledge that seed == 0 (the default; drops three 64-bit ops).
// The hash is produced in two steps so that the hot loop can reject on high words alone.  `HashParts` are the two
// fmix64 states short of their last multiply and xor-shift:  a = ka * M2, b = kb * M2,
// hash = (a ^ a>>33) + (b ^ b>>33).  The final xor-shifts only touch the low words, so hi(hash) is hi(a) + hi(b)
// or that plus one (parts_hi_plus1 is the test the hot loop makes).
struct HashParts {
    u64 ka, kb;
};
constexpr u64 FMIX_M1 = 0xff51afd7ed558ccdULL, FMIX_M2 = 0xc4ceb9fe1a85ec53ULL;
FH_HD u64 fmix64_head(u64 k) {
#if defined(__HIP_DEVICE_COMPILE__)
    // k * M1 as three multiplier instructions and a plain add: the cross terms lo * M1.hi + hi * M1.lo are the low word
    // of a v_mad_u64_u32 whose accumulator carries the first product (its high half is whatever the pair holds)
    constexpr u32 ML = (u32)FMIX_M1, MH = (u32)(FMIX_M1 >> 32);
    const u32 hi = (u32)(k >> 32), lo = (u32)k ^ (hi >> 1);
    const u64 t = (u64)lo * ML;
    const u64 q = mad64(hi, ML, ((u64)hi << 32) | (lo * MH));
    const u32 nh = (u32)(t >> 32) + (u32)q;
    return ((u64)nh << 32) | ((u32)t ^ (nh >> 1));
#else
    k ^= k >> 33;
    k *= FMIX_M1;
    k ^= k >> 33;
    return k;
#endif
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
    // a + b = (ka + kb) * M2 =: P (mod 2^64), and hi(a) + hi(b) = hi(P) - carry(a.lo + b.lo): one 64-bit add in front of
    // ONE high-word product instead of two.  hi(hash) is hi(P) - 1, hi(P) or hi(P) + 1, so
    //     hash <= tau  =>  hi(P) + 1  (mod 2^32)  <=  hi(tau) + 2
    constexpr u32 M2L = (u32)FMIX_M2, M2H = (u32)(FMIX_M2 >> 32);
    const u64 s = add64(p.ka, p.kb);
    const u32 sl = (u32)s, sh = (u32)(s >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    u64 z, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 1" : "=v"(z), "=s"(carry) : "v"(sl), "s"(M2H)); // low word: sl * M2H + 1
    return mulhi32(sl, M2L) + (u32)mad64(sh, M2L, z);
#else
    return mulhi32(sl, M2L) + sl * M2H + sh * M2L + 1u;
#endif
}
FH_HD u32 tau_hi_bound(u64 tau) { return (u32)(tau >> 32) >= 0xFFFFFFFEu ? 0xFFFFFFFFu : (u32)(tau >> 32) + 2u; }

template <int K, bool SEED0>
FH_HD HashParts murmur_finish_parts(const KeyWords<K> &w, u64 seed) {
    constexpr int NB = K / 16, TAIL = K & 15;
    u64 h1 = seed, h2 = seed;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < NB; ++b) {
        u64 k1 = key_word_mix<K>(w, 2 * b);
        if (SEED0 && b == 0) h1 = k1;
        else h1 ^= k1;
        h1 = join64(rotl64h<27>(make64(h1)));
        if (!(SEED0 && b == 0)) h1 = add64(h1, h2);
        h1 = mul5_add(h1, 0x52dce729ULL);
        u64 k2 = key_word_mix<K>(w, 2 * b + 1);
        if (SEED0 && b == 0) h2 = k2;
        else h2 ^= k2;
        h2 = join64(rotl64h<31>(make64(h2)));
        h2 = add64(h2, h1);
        h2 = mul5_add(h2, 0x38495ab5ULL);
    }
    // the tail words and the key length go in by xor: where both apply to a state, one three-input v_bitop3_b32 on
    // the low word (the length is < 2^32) instead of two v_xor_b32
    bool len1 = !len_folded(K, false), len2 = !len_folded(K, true);
    if (TAIL > 8) {
        u64 k2 = key_word_mix<K>(w, 2 * NB + 1);
        if (SEED0 && NB == 0) h2 = k2;
        else if (len2) h2 = xor3_lo(h2, k2, (u32)K), len2 = false;
        else h2 ^= k2;
    }
    if (TAIL > 0) {
        u64 k1 = key_word_mix<K>(w, 2 * NB);
        if (SEED0 && NB == 0) h1 = k1;
        else if (len1) h1 = xor3_lo(h1, k1, (u32)K), len1 = false;
        else h1 ^= k1;
    }
    if (len1) h1 ^= (u64)K;
    if (len2) h2 ^= (u64)K;
    h1 = add64(h1, h2);
    h2 = add64(h2, h1);
    return HashParts{fmix64_head(h1), fmix64_head(h2)};
}

template <int K, bool SEED0>
FH_HD u64 murmur_finish(const KeyWords<K> &w, u64 seed) {
    return parts_hash(murmur_finish_parts<K, SEED0>(w, seed));
}

// murmurhash3_x64_128(ascii(canonical k-mer), seed).0 from the m-form canonical word
template <int K, bool SEED0>
FH_HD u64 murmur_h1_fast(u64 cm, u64 seed, const LutTables &T) {
    KeyWords<K> w;
    murmur_lookup<K>(cm, T, w);
    return murmur_finish<K, SEED0>(w, seed);
}

// ------------------------------------------------------------------------------------------------------------------
// K = 33..64: k-mers of two 64-bit words.  finch's kmer_length is a u8 (sketch_schemes/mod.rs:54-71) and the reference
// hashes whatever length it is given; beyond 32 bases the canonical word no longer fits a register pair, so these K use
// their own window type.  Everything else -- classification, the lookup tables (a K-byte murmur3 key is K/16 blocks of
// two 8-byte words + a tail, each word two 4-base groups: word_geom, key_word_mix, murmur_finish_parts are written for any
// K) -- is shared with the narrow path.
//   * a lane still owns 32 start positions, but a window now reaches up to 63 bases beyond its start: the lane sees 96
//     bases (its own 32 and the two following lanes' 32 each) as three l-form code words;
//   * the reverse-complement string nC (~codes << PRE) and the digit-reversed string D (base b at digit 95 - b) are 192 bits
//     long; window j is bits [2j, +NB) of nC and bits [2(96 - K - j) - PRE, +NB) of D, NB = 2K + PRE <= 128;
//   * the canonical word (<< PRE) is four dwords; every 4-base group of the murmur3 key is a byte of one of them.
// ------------------------------------------------------------------------------------------------------------------
struct U128 {
    u64 lo, hi;
};
FH_HD bool less128(U128 a, U128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

template <int K>
struct WindowsW {
    static_assert(K > 32 && K <= 64, "WindowsW serves K = 33..64");
    static constexpr int PRE = pre_shift(K), NB = 2 * K + PRE; // 72..128
    u32 nC[8], D[8];

    FH_HDM void init(u64 c0, u64 c1, u64 c2) {
        const u32 w[6] = {~(u32)c0, ~(u32)(c0 >> 32), ~(u32)c1, ~(u32)(c1 >> 32), ~(u32)c2, ~(u32)(c2 >> 32)};
        if (PRE == 0) {
            for (int i = 0; i < 6; ++i) nC[i] = w[i];
            nC[6] = 0;
        } else {
            nC[0] = w[0] << PRE;
            for (int i = 1; i < 6; ++i) nC[i] = alignbit_b32(w[i], w[i - 1], 32 - PRE);
            nC[6] = w[5] >> (32 - PRE);
        }
        nC[7] = 0;
        const u64 r0 = pairrev64(c2), r1 = pairrev64(c1), r2 = pairrev64(c0);
        D[0] = (u32)r0;
        D[1] = (u32)(r0 >> 32);
        D[2] = (u32)r1;
        D[3] = (u32)(r1 >> 32);
        D[4] = (u32)r2;
        D[5] = (u32)(r2 >> 32);
        D[6] = D[7] = 0;
    }
    // bits [off, off + NB) of the string W, as four dwords d[0..3]
    static FH_HDM void field(const u32 *W, int off, u32 *d) {
        const int w = off >> 5, s = off & 31;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i < 4; ++i) {
            if (32 * i >= NB) d[i] = 0;
            else {
                u32 x = s ? alignbit_b32(W[w + i + 1], W[w + i], (u32)s) : W[w + i];
                if (NB - 32 * i < 32) x &= (1u << (NB - 32 * i)) - 1u;
                d[i] = x;
            }
        }
    }
    FH_HDM void fwd(int j, u32 *d) const { field(D, 2 * (96 - K - j) - PRE, d); }
    // (even K: a k-mer may equal its reverse complement; the scrap bits are cleared on this side so that the tie goes to
    //  rc, exactly as in Windows<K>::rc)
    FH_HDM void rc(int j, u32 *d) const {
        field(nC, 2 * j, d);
        if (K % 2 == 0 && PRE) d[0] &= ~((1u << PRE) - 1u);
    }
    // Move the lane's view eight positions on: afterwards window j of this object is what window j + 8 was.  (The kernel
    // runs 4 rounds of 8 fully unrolled positions: bit-field offsets stay compile-time constants while the code is a
    // quarter of the 32-position unroll, which needed all 256 VGPRs.)  nC moves down 16 bits, D up 16 bits: D's 192 bits
    // sit in 256, so three moves lose nothing a later window reads.
    FH_HDM void advance8() {
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i < 7; ++i) nC[i] = alignbit_b32(nC[i + 1], nC[i], 16);
        nC[7] >>= 16;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 7; i > 0; --i) D[i] = alignbit_b32(D[i], D[i - 1], 16);
        D[0] <<= 16;
    }
    // the canonical m-form word << PRE as four dwords; tie -> rc (needletail canonical_kmers)
    FH_HDM void canonical(int j, u32 *cm, bool &is_rc) const {
        u32 f[4], r[4];
        fwd(j, f);
        rc(j, r);
        const U128 F{((u64)f[1] << 32) | f[0], ((u64)f[3] << 32) | f[2]}, R{((u64)r[1] << 32) | r[0], ((u64)r[3] << 32) | r[2]};
        is_rc = !less128(F, R);
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i < 4; ++i) cm[i] = is_rc ? r[i] : f[i];
    }
};

// the canonical k-mer itself (m-form, 2K bits) from the pre-shifted dwords
template <int K>
FH_HD U128 kmer_words_w(const u32 *cm) {
    constexpr int PRE = pre_shift(K);
    const u64 lo = ((u64)cm[1] << 32) | cm[0], hi = ((u64)cm[3] << 32) | cm[2];
    if (PRE == 0) return U128{lo, hi};
    return U128{(lo >> PRE) | (hi << (64 - PRE)), hi >> PRE};
}

// bit j: the K-base window starting at base j (0..31) lies entirely in good bases; g0 / g1 / g2 = good bits of the lane's
// own 32 bases and of the next two lanes'
template <int K>
FH_HD u32 window_valid_mask_w(u32 g0, u32 g1, u32 g2) {
    typedef unsigned __int128 u128;
    u128 A[7];
    A[0] = (u128)g0 | ((u128)g1 << 32) | ((u128)g2 << 64);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int p = 1; p < 7; ++p) A[p] = A[p - 1] & (A[p - 1] >> (1 << (p - 1)));
    u128 W = ~(u128)0;
    int off = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int p = 6; p >= 0; --p) {
        if (K & (1 << p)) {
            W &= (A[p] >> off);
            off += (1 << p);
        }
    }
    return (u32)W;
}

// byte offset of a group's record: full groups are bytes of one of the four dwords (shift is a multiple of 8, see
// pre_shift); the key's last, short group sits in the low bits of dword 0
FH_HD u32 field_off_w(const u32 *cm, int shift, int nb, int lg) {
    if (nb == 4 && (shift & 7) == 0) return byte_shl(cm[shift >> 5], (shift >> 3) & 3, lg);
    const u32 fm = ((1u << (2 * nb)) - 1u) << lg;
    const int sh = shift - lg;
    return sh >= 0 ? (cm[0] >> sh) & fm : (cm[0] << (-sh)) & fm;
}

template <int K>
FH_HD void murmur_lookup_w(const u32 *cm, const LutTables &T, KeyWords<K> &w) { // cm = canonical word << pre_shift(K), 4 dwords
    constexpr int PRE = pre_shift(K);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < KeyWords<K>::N; ++i) {
        const WordGeom g = word_geom(K, i);
        w.a0[i] = w.a1[i] = w.a2[i] = w.b0[i] = w.b1[i] = 0;
        if (g.kind == 1) {
            const Rec2 r = *(const Rec2 *)((const char *)T.P + field_off_w(cm, g.shiftA + PRE, g.nbA, 3));
            w.a0[i] = r.x;
            w.a1[i] = r.y;
        } else if (g.kind == 2) {
            const Rec4 ra = load_rec4((const Rec4 *)((const char *)(g.is_k2 ? T.A2 : T.A1) + field_off_w(cm, g.shiftA + PRE, 4, 4)));
            const Rec2 *TB = g.partial ? T.P : (g.is_k2 ? T.B2 : T.B1);
            const Rec2 rb = *(const Rec2 *)((const char *)TB + field_off_w(cm, g.shiftB + PRE, g.nbB, 3));
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FH_NO_B64_KEEP)
            // one 16-byte and one 8-byte LDS read per word, as in murmur_lookup: left alone the compiler splits the A record
            // into an 8- and a 4-byte read (its fourth dword is unused) and narrows a k2 word's B record to 4 bytes -- 13 LDS
            // instructions per position at K = 33 on a kernel that lives off the LDS pipe (profiles/r03_k33_*)
            asm volatile("" ::"v"(ra.w));
            if (g.is_k2) asm volatile("" ::"v"(rb.y));
#endif
            w.a0[i] = ra.x;
            w.a1[i] = ra.y;
            w.a2[i] = ra.z;
            w.b0[i] = g.is_k2 ? rb.x : rb.y;
            w.b1[i] = g.is_k2 ? rb.y : rb.x;
        }
    }
}

template <int K>
FH_HD u64 murmur_h1_fast_w(const u32 *cm, u64 seed, const LutTables &T) {
    KeyWords<K> w;
    murmur_lookup_w<K>(cm, T, w);
    return murmur_finish<K, false>(w, seed);
}

// ASCII of a k-mer held as two m-form words (hi: the first K - 32 bases when K > 32)
FH_HD uint8_t kmer_base(u64 lo, u64 hi, int k, int b) {
    const int d = k - 1 - b; // digit, counted from the last base
    const u32 code = (u32)(((d >= 32) ? (hi >> (2 * (d - 32))) : (lo >> (2 * d))) & 3u);
    return (uint8_t) "ACGT"[code];
}

// Quarter-octave index of a hash value (floor(4 log2 x), x >= 4): the sampling pre-pass of large sketches histograms the
// sample's hashes by it (fh_kernels.hip, k_live_count_hist), and the host turns a bucket back into its upper edge
FH_HD u32 qoct_index(u64 x) {
    if (x < 4) return (u32)x; // (below the first quarter-octave: one bucket per value)
#if defined(__HIP_DEVICE_COMPILE__)
    const int e = 63 - __clzll((long long)x);
#else
    const int e = 63 - __builtin_clzll(x);
#endif
    return (u32)(4 * e) + (u32)((x >> (e - 2)) & 3u);
}
FH_HD u64 qoct_upper_edge(u32 q) { // largest x with qoct_index(x) <= q
    if (q < 8) return q < 3 ? (u64)q : 3ULL;
    const u32 e = (q + 1) >> 2, m = (q + 1) & 3u;
    if (e >= 64) return ~0ULL;
    return ((1ULL << e) + ((u64)m << (e - 2))) - 1ULL;
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
