// fh_device.h -- device-resident sketch state shared by the kernels (fh_kernels.hip) and the host
// side of the C ABI (fh_api.hip).
//
// The per-sketcher state replaces the reference's BinaryHeap + HashMap (mash.rs:10-18):
//   * an open-addressing hash table in HBM keyed by the 64-bit k-mer hash, each entry holding the
//     occurrence count, the reverse-strand count, the stream position of the first occurrence and the
//     2-bit packed canonical k-mer;
//   * a monotone non-increasing admit threshold tau: a k-mer occurrence is upserted iff hash <= tau.
// tau only ever drops to the n-th smallest distinct hash seen so far, so no occurrence of a final
// bottom-n member is ever dropped => counts are exact and the result is order independent
// (SURVEY.md 8e).  "Pruning" = pick the new tau and shrink the live list; table slots of pruned
// entries are left behind as garbage that can never match again (their hash > tau).
#pragma once
#include <stdint.h>

namespace fh {

struct Entry {          // 40 B
    uint64_t hash;      // EMPTY64 = free slot
    uint64_t kmer;      // m-form canonical k-mer, EMPTY64 until claimed
    uint64_t pos;       // smallest stream position of an occurrence (atomicMin)
    uint64_t count;     // occurrences            (clamped to u32::MAX on output, mash.rs:46-49)
    uint64_t extra;     // occurrences on the rc strand
};

struct CollRec {        // an occurrence whose k-mer differs from the slot's k-mer (64-bit hash collision)
    uint64_t hash, kmer, pos;
};

struct Ctl {
    uint64_t tau;           // admit iff hash <= tau
    uint64_t total_kmers;   // number of valid k-mer windows seen (mash.rs:35)
    uint32_t n_live;        // entries in the live list
    uint32_t overflow;      // table probe limit hit / live list full (capacity error)
    uint32_t n_coll;        // collision log entries
    uint32_t need_big;      // the single-workgroup prune met more live entries than it can sort
    uint32_t sorted;        // live list currently sorted ascending by hash
    uint32_t launches_skipped;
    uint32_t n_dead;        // entries in the dropped-slot list (0xFFFFFFFF = list overflowed)
    uint32_t pad0;
    // the one hash value that cannot be a table key (== EMPTY64)
    uint64_t sp_count, sp_extra, sp_pos, sp_kmer;
};

constexpr int TILE_POS = 2048;   // k-mer start positions per wavefront tile (64 lanes x 32)
constexpr int LANE_POS = 32;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int SMALL_MAX = 8192;  // live entries the single-workgroup prune can sort in LDS
constexpr int MAX_PROBE = 4096;

struct SketchArgs {
    const uint8_t *seq;   // packed stream (device), 16-byte aligned
    uint64_t len_total;   // readable bytes
    uint64_t p_begin;     // first k-mer start position of this launch (multiple of TILE_POS)
    uint64_t p_end;       // one past the last k-mer start position
    uint64_t base_pos;    // stream coordinate of seq[0]
    uint64_t seed;
    uint64_t hash_mask;   // ~0 unless the test hook is on
    Entry *table;
    uint32_t cap;
    uint32_t *live;
    uint32_t live_cap;
    Ctl *ctl;
    CollRec *clog;
    uint32_t clog_cap;
    uint32_t tiles_total;
    uint32_t tiles_per_wave;
};

} // namespace fh
