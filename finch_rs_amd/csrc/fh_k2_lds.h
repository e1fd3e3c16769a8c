// fh_k2_lds.h -- what the tile kernels of K = 1..32 share about their workgroups: waves per workgroup, positions per unrolled
// round, and where a workgroup's LDS lives (lookup tables, the waves' two-tile rings, admit queues).  Used by k2_sketch
// (fh_k2.hip) and by the batch kernel (fh_k2b.hip: many files per launch), which runs the very same per-tile code.
#pragma once
#include <hip/hip_runtime.h>

#include "fh_core.h"
#include "fh_device.h"
#include "fh_kernels.h"
#include "fh_k2_common.h"

namespace fh {

// Register budget: four waves per SIMD (<= 128 VGPRs) for every K, with NO spilled register (tools/k2_regs.py checks the
// shipped objects).  Everything wave-uniform is kept scalar (threshold, wave index, queue bookkeeping); K <= 22 run the
// lane's 32 positions as one unrolled pass (k = 21: 127 VGPRs), K >= 23 -- K >= 25 would take 150-190 registers that way --
// as two rounds of 16 (102-116 VGPRs; profiles/r03_ab_rounds.txt).  Those kernels do 6-8 table lookups per position and
// live off the LDS pipe, where a fourth wave is worth more than a longer unrolled pass (measured k = 31: 373 Gbases/s at
// 2 waves, 427 at 3, 443 at 4, 272 at 5).  What must never be spilled is anything the admit path reads: a reload there
// stalls ~40 % of the wave-iterations of a launch that admits 1 %.
#ifndef FH_MINW_BIG
#define FH_MINW_BIG 4
#endif
#ifndef FH_MINW_SMALL
#define FH_MINW_SMALL 4
#endif
constexpr int k2_min_waves(int K) { return K <= 21 ? FH_MINW_SMALL : FH_MINW_BIG; }
// positions per unrolled round of the lane's 32 (see the loop): 32 = one pass
#ifndef FH_ROUND_BIG
#define FH_ROUND_BIG 16
#endif
#ifndef FH_ROUND_FROM
#define FH_ROUND_FROM 23 // (k = 23, 24 as one pass of 32 spill 2-7 registers at the 128-VGPR limit; in two rounds they take 109)
#endif
constexpr int k2_round(int K) { return K >= FH_ROUND_FROM ? FH_ROUND_BIG : 32; }
// K >= 25 live off the LDS pipe (8 random table lookups per position at k = 31: the CU's LDS array was busy 310 of the 325
// cycles a wave-iteration took, two thirds of that bank conflicts -- profiles/r03b_k31_pmc_k2.json).  There ONE workgroup of
// sixteen waves per CU shares the tables instead of four workgroups holding a set each, and the LDS that frees holds the k1
// words' A table sixteen times over, laid out so that its 16-byte reads cannot conflict (fh_core.h, LutTables): two of
// k = 31's four A lookups drop from 9.6 to 4 LDS cycles.
#ifndef FH_SHARE_FROM
#define FH_SHARE_FROM 25
#endif
constexpr int k2_wpb_of(int K) { return K >= FH_SHARE_FROM ? 16 : WAVES_PER_BLOCK; } // waves per workgroup (fh_kernels.h: k2_waves_per_block)
constexpr int k2_a1_rep(int K) { return K >= FH_SHARE_FROM && has_pair_word(K, false) ? 16 : 1; }
// where a workgroup's LDS lives: the lookup tables, and per wave the two-tile ring of classified bases and the admit queue
struct K2Lds {
    Rec4 *A1, *A2;
    Rec2 *B1, *B2, *P;
    u32 *codes, *good; // [WPB][256], [WPB][128]
    AdmitQueueT<false> *queue_lo, *queue_hi; // waves [0, Q_SPLIT) and [Q_SPLIT, WPB)
    const Rec4 *a1_lookup_base;              // what the hot loop adds its A1 offsets to (the hand-laid block's first byte)
};
constexpr int K2_Q_SPLIT = 11;
// four workgroups of four waves per CU, a table set each: separate arrays, as the K <= 24 kernels were tuned with
template <int K>
__device__ __forceinline__ K2Lds k2_lds_plain() {
    __shared__ Rec4 sA1[has_pair_word(K, false) ? 256 : 1];
    __shared__ Rec4 sA2[has_pair_word(K, true) ? 256 : 1];
    __shared__ Rec2 sB1[has_pair_word(K, false) ? 256 : 1];
    __shared__ Rec2 sB2[has_pair_word(K, true) ? 256 : 1];
    __shared__ Rec2 sP[partial_entries(K)];
    __shared__ __attribute__((aligned(16))) u32 sCodes[WAVES_PER_BLOCK][256];
    __shared__ __attribute__((aligned(16))) u32 sGood[WAVES_PER_BLOCK][128];
    __shared__ __attribute__((aligned(16))) AdmitQueueT<false> sQueue[WAVES_PER_BLOCK];
    return K2Lds{sA1, sA2, sB1, sB2, sP, &sCodes[0][0], &sGood[0][0], sQueue, sQueue, sA1};
}
// One workgroup of sixteen waves per CU: ONE block of LDS laid out by hand.  Everything the hot loop reaches through an
// instruction's 16-bit offset field -- the small tables -- sits in the first 64 KB; the replicated A1 table is the second
// 64 KB exactly, so that its address (0x10000 | index byte << 8 | replica << 4) still comes out of the ONE v_perm_b32 that
// forms the offset (fh_core.h, byte_shl8_or: the 0x01 of bits 16-23 rides in the lane constant); the admit queues of the
// last five waves follow behind it.  (Left to the compiler's layout, tables beyond 64 KB cost an address add per lookup:
// +5.6 VALU instructions per position at k = 31, which ate what the conflict-free reads gave.)
constexpr u32 K2S_A2 = 0, K2S_B1 = 4096, K2S_B2 = 6144, K2S_P = 8192, K2S_CODES = 16384, K2S_GOOD = 32768, K2S_QLO = 40960,
              K2S_A1 = 65536, K2S_QHI = 131072, K2S_BYTES = K2S_QHI + (16 - K2_Q_SPLIT) * (u32)sizeof(AdmitQueueT<false>);
static_assert(K2S_QLO + K2_Q_SPLIT * sizeof(AdmitQueueT<false>) <= K2S_A1, "the low admit queues end below the replicated table");
static_assert(K2S_BYTES <= 160 * 1024, "one workgroup's LDS");
template <int K>
__device__ __forceinline__ K2Lds k2_lds_shared() {
    static_assert(partial_entries(K) * sizeof(Rec2) <= K2S_CODES - K2S_P, "the key's last-word table fits its slot");
    __shared__ __attribute__((aligned(65536))) unsigned char blob[K2S_BYTES];
    return K2Lds{(Rec4 *)(blob + K2S_A1), (Rec4 *)(blob + K2S_A2), (Rec2 *)(blob + K2S_B1), (Rec2 *)(blob + K2S_B2), (Rec2 *)(blob + K2S_P),
                 (u32 *)(blob + K2S_CODES), (u32 *)(blob + K2S_GOOD), (AdmitQueueT<false> *)(blob + K2S_QLO),
                 (AdmitQueueT<false> *)(blob + K2S_QHI) - K2_Q_SPLIT, (const Rec4 *)blob};
}

} // namespace fh
