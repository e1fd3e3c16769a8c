// Host build of the per-lane kernel arithmetic (finch_rs_amd/csrc/fh_core.h) for logic tests on a
// GPU-less machine.  Test infrastructure: it is NOT part of libfinch_hip.so and nothing in the
// product path calls it.  It walks a byte stream exactly the way one lane of the sketch kernel does
// (32 start positions per lane segment, classification of 16-byte chunks, state initialised from the
// first K-1 bases, LUT-based murmur3) and reports per-position results.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../finch_rs_amd/csrc/fh_core.h"

using namespace fh;

template <int K>
static int run(const uint8_t *seq, uint64_t len, uint64_t seed, uint64_t *hashes, uint8_t *valid, uint8_t *isrc,
               uint64_t *canon) {
    std::vector<uint8_t> buf(((len + 31) / 32) * 32 + 96, 0);
    memcpy(buf.data(), seq, len);
    std::vector<u64> T1(256), T2(256), TP(64, 0);
    for (u32 q = 0; q < 256; ++q) {
        T1[q] = lut_entry(q, 4, MURMUR_C1);
        T2[q] = lut_entry(q, 4, MURMUR_C2);
    }
    const int pnb = partial_nb(K);
    for (u32 q = 0; q < (1u << (2 * pnb)); ++q)
        if (pnb) TP[q] = lut_entry(q, pnb, partial_const(K));
    // the kernel's tables (second stage folded in), built exactly as the workgroup prologue builds them
    std::vector<Rec4> A1(256), A2(256);
    std::vector<Rec2> B1(256), B2(256), P(partial_entries(K));
    for (u32 q = 0; q < 256; ++q) {
        A1[q] = lut_rec_A(q, false);
        A2[q] = lut_rec_A(q, true);
        B1[q] = lut_rec_B(q, 4, false);
        B2[q] = lut_rec_B(q, 4, true);
    }
    for (u32 q = 0; q < (u32)partial_entries(K); ++q) P[q] = lut_rec_P<K>(q);
    const LutTables LT{A1.data(), A2.data(), B1.data(), B2.data(), P.data()};
    for (uint64_t s = 0; s < len; s += 32) {
        u32 cw[4], gw[4];
        for (int c = 0; c < 4; ++c) {
            u32 d[4];
            memcpy(d, buf.data() + s + 16 * c, 16);
            classify_chunk(d[0], d[1], d[2], d[3], cw[c], gw[c]);
        }
        const u64 clo = (u64)cw[0] | ((u64)cw[1] << 32), chi = (u64)cw[2] | ((u64)cw[3] << 32);
        const u64 g64 = (u64)gw[0] | ((u64)gw[1] << 16) | ((u64)gw[2] << 32) | ((u64)gw[3] << 48);
        Windows<K> win;
        win.init(clo, chi);
        const u32 W = window_valid_mask<K>(g64 & 0x7FFFFFFFFFFFFFFFull);
        for (int j = 0; j < 32; ++j) {
            const uint64_t p = s + j;
            if (p >= len) break;
            bool rc;
            const u64 cm = win.canonical(j, rc);
            valid[p] = (W >> j) & 1u;
            isrc[p] = rc ? 1 : 0;
            canon[p] = cm >> pre_shift(K);
            const u64 h_ref = murmur_h1_lut<K>(cm >> pre_shift(K), seed, T1.data(), T2.data(), TP.data());
            const u64 h_fast = murmur_h1_fast<K, false>(cm, seed, LT);
            if (h_fast != h_ref) return -2;
            if (seed == 0 && murmur_h1_fast<K, true>(cm, 0, LT) != h_ref) return -3;
            hashes[p] = h_fast;
        }
    }
    // The segment kernel's form (fh_k2s.hip): the buffer as ONE tile's three strings, a round's view cut at EVERY position p0
    // (a lane's p0 = stride * lane + round * seg_round(K) is any number), Windows<K, seg_doff(K)> and the 64-bit validity mask:
    // every window of the round against what the walk above found at that position.
    {
        const u32 NCH = (u32)(buf.size() / 16);
        std::vector<u32> Fc(NCH + 3, 0u), Rv(NCH + 2, 0u), Gd(NCH / 2 + 3, 0u);
        for (u32 i = 0; i < NCH; ++i) {
            u32 d[4], q, g;
            memcpy(d, buf.data() + 16 * (size_t)i, 16);
            classify_chunk(d[0], d[1], d[2], d[3], q, g);
            Fc[1 + i] = ~q;
            Rv[NCH - 1 - i] = pairrev32(q);
            reinterpret_cast<unsigned short *>(Gd.data())[i] = (unsigned short)g;
        }
        constexpr int R = seg_round(K), DOFF = seg_doff(K);
        static_assert(R + K - 1 <= 64, "a round's windows lie inside the 64-base view");
        for (uint64_t p0 = 0; p0 + 64 + DOFF <= 16ull * NCH && p0 < len; ++p0) {
            u32 nc[5], d[5];
            seg_cut_views<K>(Fc.data(), Rv.data(), NCH, (u32)p0, nc, d);
            Windows<K, DOFF> win;
            win.init_words(nc, d);
            const u64 W = window_valid_mask64<K>(seg_good_bits(Gd.data(), (u32)p0));
            for (int j = 0; j < R; ++j) {
                const uint64_t p = p0 + j;
                if (p >= len) break;
                if (((W >> j) & 1u) != valid[p]) return -5;
                if (!valid[p]) continue;
                bool rc;
                const u64 cm = win.canonical(j, rc);
                if ((cm >> pre_shift(K)) != canon[p] || (rc ? 1 : 0) != isrc[p]) return -6;
                if (win.canonical_word(j) != cm || win.strand_of(j) != rc) return -7;
            }
        }
    }
    return 0;
}

template <int K>
static int dispatch(int k, const uint8_t *seq, uint64_t len, uint64_t seed, uint64_t *hashes, uint8_t *valid,
                    uint8_t *isrc, uint64_t *canon) {
    if (k == K) return run<K>(seq, len, seed, hashes, valid, isrc, canon);
    if constexpr (K > 1) return dispatch<K - 1>(k, seq, len, seed, hashes, valid, isrc, canon);
    return -1;
}

extern "C" int fhcore_positions(const uint8_t *seq, uint64_t len, int k, uint64_t seed, uint64_t *hashes,
                                uint8_t *valid, uint8_t *isrc, uint64_t *canon) {
    if (k < 1 || k > 32) return -1;
    return dispatch<32>(k, seq, len, seed, hashes, valid, isrc, canon);
}

// K = 33..64 (WindowsW): the same walk with three code words per lane segment; canon receives two words per position
// (low, high) of the canonical k-mer in m-form
template <int K>
static int run_w(const uint8_t *seq, uint64_t len, uint64_t seed, uint64_t *hashes, uint8_t *valid, uint8_t *isrc,
                 uint64_t *canon) {
    std::vector<uint8_t> buf(((len + 31) / 32) * 32 + 128, 0);
    memcpy(buf.data(), seq, len);
    std::vector<Rec4> A1(256), A2(256);
    std::vector<Rec2> B1(256), B2(256), P(partial_entries(K));
    for (u32 q = 0; q < 256; ++q) {
        A1[q] = lut_rec_A(q, false);
        A2[q] = lut_rec_A(q, true);
        B1[q] = lut_rec_B(q, 4, false);
        B2[q] = lut_rec_B(q, 4, true);
    }
    for (u32 q = 0; q < (u32)partial_entries(K); ++q) P[q] = lut_rec_P<K>(q);
    const LutTables LT{A1.data(), A2.data(), B1.data(), B2.data(), P.data()};
    for (uint64_t s = 0; s < len; s += 32) {
        u32 cw[6], gw[6];
        for (int c = 0; c < 6; ++c) {
            u32 d[4];
            memcpy(d, buf.data() + s + 16 * c, 16);
            classify_chunk(d[0], d[1], d[2], d[3], cw[c], gw[c]);
        }
        const u64 c0 = (u64)cw[0] | ((u64)cw[1] << 32), c1 = (u64)cw[2] | ((u64)cw[3] << 32), c2 = (u64)cw[4] | ((u64)cw[5] << 32);
        WindowsW<K> win;
        win.init(c0, c1, c2);
        const u32 W = window_valid_mask_w<K>(gw[0] | (gw[1] << 16), gw[2] | (gw[3] << 16), gw[4] | (gw[5] << 16));
        for (int j = 0; j < 32; ++j) {
            const uint64_t p = s + j;
            if (p >= len) break;
            bool rc;
            u32 cm[4];
            win.canonical(j, cm, rc);
            valid[p] = (W >> j) & 1u;
            isrc[p] = rc ? 1 : 0;
            const U128 km = kmer_words_w<K>(cm);
            canon[2 * p] = km.lo;
            canon[2 * p + 1] = km.hi;
            hashes[p] = murmur_h1_fast_w<K>(cm, seed, LT);
        }
        // the kernel's form: four rounds of eight positions, the strings moved on between rounds
        WindowsW<K> adv;
        adv.init(c0, c1, c2);
        for (int c = 0; c < 4; ++c) {
            for (int u = 0; u < 8; ++u) {
                bool rc0, rc1;
                u32 a[4], b[4];
                adv.canonical(u, a, rc0);
                win.canonical(8 * c + u, b, rc1);
                if (rc0 != rc1 || memcmp(a, b, 16) != 0) return -4;
            }
            adv.advance8();
        }
    }
    return 0;
}

template <int K>
static int dispatch_w(int k, const uint8_t *seq, uint64_t len, uint64_t seed, uint64_t *hashes, uint8_t *valid,
                      uint8_t *isrc, uint64_t *canon) {
    if (k == K) return run_w<K>(seq, len, seed, hashes, valid, isrc, canon);
    if constexpr (K > 33) return dispatch_w<K - 1>(k, seq, len, seed, hashes, valid, isrc, canon);
    return -1;
}

extern "C" int fhcore_positions_w(const uint8_t *seq, uint64_t len, int k, uint64_t seed, uint64_t *hashes,
                                  uint8_t *valid, uint8_t *isrc, uint64_t *canon) {
    if (k < 33 || k > 64) return -1;
    return dispatch_w<64>(k, seq, len, seed, hashes, valid, isrc, canon);
}

extern "C" uint32_t fhcore_qoct_index(uint64_t x) { return qoct_index(x); }
extern "C" uint64_t fhcore_qoct_upper_edge(uint32_t q) { return qoct_upper_edge(q); }

extern "C" void fhcore_synth(uint8_t *genome, uint64_t glen, uint8_t *reads, uint64_t first, uint64_t n, uint32_t rl,
                             uint64_t seed, uint32_t sub_ppm, uint32_t n_ppm) {
    for (uint64_t i = 0; i < glen; ++i) genome[i] = synth_genome_base(seed, i);
    for (uint64_t r = 0; r < n; ++r)
        for (uint32_t j = 0; j <= rl; ++j)
            reads[r * (rl + 1) + j] = synth_read_byte(genome, glen, first + r, j, rl, seed, sub_ppm, n_ppm);
}

// high-word prefilter (fh_core.h HashParts): returns the number of violations of
//   parts_hash(p) <= tau  =>  parts_hi_plus1(p) <= tau_hi_bound(tau)
// over the given (a, b, tau) triples, and writes the prefilter's pass count (selectivity check)
extern "C" uint64_t fhcore_prefilter_check(const uint64_t *a, const uint64_t *b, const uint64_t *tau, uint64_t n,
                                           uint64_t *n_pass, uint64_t *n_true) {
    uint64_t bad = 0, pass = 0, tr = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const HashParts p{a[i], b[i]}; // (ka, kb)
        const bool truth = parts_hash(p) <= tau[i];
        const bool cand = parts_hi_plus1(p) <= tau_hi_bound(tau[i]);
        if (truth && !cand) ++bad;
        pass += cand;
        tr += truth;
    }
    *n_pass = pass;
    *n_true = tr;
    return bad;
}

// classification of n 16-byte chunks: codes (32 bits, base i at bits [2i, 2i+2)) and good bits per chunk
extern "C" void fhcore_classify(const uint8_t *bytes, uint64_t n_chunks, uint32_t *codes, uint32_t *good) {
    for (uint64_t c = 0; c < n_chunks; ++c) {
        u32 d[4];
        memcpy(d, bytes + 16 * c, 16);
        classify_chunk(d[0], d[1], d[2], d[3], codes[c], good[c]);
    }
}
