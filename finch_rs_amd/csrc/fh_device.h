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
    uint64_t count;     // occurrences whose canonical k-mer is the forward window   } one atomic add per occurrence;
    uint64_t extra;     // occurrences whose canonical k-mer is the reverse complement } output: count = sum, extra_count
                        // = extra, each clamped to u32::MAX (mash.rs:46-49)
};

struct CollRec {        // an occurrence whose k-mer differs from the slot's k-mer (64-bit hash collision)
    uint64_t hash, kmer, pos;
    uint64_t kmer_hi;   // K > 32: the first K - 32 bases (0 otherwise)
};

struct Ctl {
    uint64_t tau;           // admit iff hash <= tau
    uint32_t n_live;        // entries in the live list
    uint32_t overflow;      // table probe limit hit / live list full (capacity error)
    uint32_t n_coll;        // collision log entries
    uint32_t need_big;      // the single-workgroup prune met more live entries than it can sort
    uint32_t sorted;        // live list currently sorted ascending by hash
    uint32_t spec_ok;       // verdict of the last speculative range, taken on the device (k_small_epilogue): 1 = the range ran
                            // dry and left at least `size` hashes at or below the guessed threshold.  Launches marked
                            // `gate` (SketchArgs, k_queue_reset) do nothing unless it is 1, so the host can queue the
                            // speculated range, the verdict and the rest of the input without a round trip in between.
    uint32_t n_dead;        // entries in the dropped-slot list (0xFFFFFFFF = list overflowed)
    uint32_t hist_on;       // 1: every NEW hash is counted in hist[] and the admit path refreshes tau from it (below)
    uint32_t left_in_pos;   // next unread entry of the leftover list handed to this launch
    uint32_t n_left_out;    // leftover tile ranges recorded by waves that stopped mid-chunk
    uint32_t soft_limit;    // the inserter that takes n_live to this value raises `stopped`
    uint32_t read_first;    // admit path of this launch reads an entry before it issues atomics on it (fh_k2.hip, upsert)
    // the one hash value that cannot be a table key (== EMPTY64)
    uint64_t sp_count, sp_extra, sp_pos, sp_kmer;
    // In-launch threshold refresh.  A persistent wave used to keep the threshold it read at its start, so a launch over
    // 50 Gbases admitted at the rate of its first microsecond, filled the live set and had to be stopped, pruned and
    // relaunched three times per pass.  Now every hash that is NEW to the table is counted by quarter-octave of its value
    // (fh_core.h qoct_index; each distinct hash is inserted at most once between resets, tau being monotone), and the
    // upper edge of the first bucket at which the running total reaches sel_size is a valid threshold whatever the table
    // still holds: at least sel_size distinct hashes at or below it have been seen, so no final member lies above it
    // (SURVEY.md 8e).  The wave whose insert is the 16th of its shard list recomputes it (refresh_tau, fh_k2_common.h) and
    // lowers tau with an atomic min; waves re-read tau once per tile.  sel_size = 0: off.  tau_floor: a scaled sketch
    // keeps everything <= max_hash.
    uint64_t sel_size, tau_floor;
    // K > 32: a k-mer is two words.  The table entry keeps the low one (the last 32 bases); the first K - 32 bases of slot i
    // sit in kmer_hi[i], an array of its own so that the entry layout and the K <= 32 kernels do not change.  null otherwise.
    uint64_t *kmer_hi;
    // number of valid k-mer windows seen (mash.rs:35), spread over many words so that the one atomic each
    // wave issues at its end does not serialise on a single L2 address (~12 ns per same-address atomic)
    uint64_t kmer_counts[256];
    uint64_t text_bases;    // sequence bytes emitted by the device-side FASTQ packer (fh_text.hip)
    uint64_t inserted_total; // new hashes inserted since the last reset (k_live_commit adds each launch's appends)
    // where the table lives (read on the rare admit path only, so that the hot loop does not have to keep
    // these in scalar registers); written by the host whenever the table is (re)allocated
    Entry *table;
    uint32_t *live;
    CollRec *clog;
    uint32_t cap, live_cap, clog_cap, shard_cap;
    // New inserts are appended to N_SHARDS per-shard lists (shard = wave id mod N_SHARDS) whose cursors sit
    // on separate 128-byte lines: a single cursor would serialise every insert in L2 (~14 ns each, i.e. 0.1 s
    // per 8 M inserts during the warm-up of a 2 M-hash sketch).  k_live_flatten appends the shard lists to
    // `live` after every sketch launch, so the selection kernels only ever see one flat list.
    uint32_t *shard_cnt;    // N_SHARDS cursors, stride SHARD_STRIDE dwords
    uint32_t *shard_buf;    // N_SHARDS x shard_cap slot indices
    uint32_t shard_soft;    // a shard reaching this many appends raises `stopped`
    uint32_t pad1;
    // Tile queue of the current range (dynamic scheduling; survives a stopped launch).  Every wave hits
    // these two words once per pull, and same-line atomics serialise in L2 (~12-16 ns each), so each gets a
    // 128-byte line of its own, away from n_live / tau that the admit path updates.
    uint32_t pad3[32];
    uint32_t next_unit;     // next unit of UNIT_TILES tiles nobody has pulled yet
    uint32_t pad4[31];
    uint32_t stopped;       // raised when the live set reaches soft_limit (or a wave exhausts its budget)
    uint32_t pad5[31];
    // -DFH_PROFILE_FLUSH builds only: wave-cycles spent inside flush_queue, number of flushes, entries flushed
    uint64_t dbg_flush_cycles, dbg_flush_calls, dbg_flush_entries, dbg_wave_cycles;
    uint32_t hist[256];     // new hashes since the last reset per quarter-octave of hash value (hist_on)
};

constexpr int TILE_POS = 2048;   // k-mer start positions per wavefront tile (64 lanes x 32)
constexpr int LANE_POS = 32;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int SMALL_MAX = 12288; // live entries the single-workgroup prune can select from in LDS (8 B of key each: 96 KB of the 160)
constexpr int SMALL_SORT_MAX = 4096; // survivors it can sort there (12 B each)
constexpr int HIST_REFRESH = 16; // a shard list's every HIST_REFRESH-th insert recomputes the threshold from Ctl::hist
constexpr int MAX_PROBE = 4096;
constexpr int N_SHARDS = 256;
constexpr int SHARD_STRIDE = 32; // dwords between shard cursors (one 128-byte line each)
#ifndef FH_UNIT_TILES
#define FH_UNIT_TILES 2
#endif
#ifndef FH_MAX_UNITS
#define FH_MAX_UNITS 8
#endif
constexpr int UNIT_TILES = FH_UNIT_TILES; // queue granularity (SketchArgs::unit_tiles); a pull takes 1..MAX_UNITS consecutive units
constexpr int MAX_UNITS = FH_MAX_UNITS;   // (guided self-scheduling: big pulls first, single units at the end)
// What a wave can insert between two looks at its budget: a tile's positions (k2_sketch, k2_sketch_w: 2048) or a round's (the
// segment kernels: 64 lanes x up to 48 positions, fh_k2s.hip k2s_round).  Table, shard lists and budgets are sized with it.
// (+ QCAP = 64: up to QCAP / 2 - 1 candidates parked before the round are upserted after the wave last looked at its budget)
constexpr int WAVE_OVERSHOOT = 3072 + 64;
constexpr int WAVE_BUDGET = 2048; // + at most TILE_POS-1 overshoot inside the tile that crosses it   // tiles a wave pulls from the queue at a time (contiguous: halo reuse)

// The segment kernel (fh_k2s.hip): a lane owns one SEGMENT of seg_stride consecutive k-mer start positions instead of 32, a
// tile is 64 segments.  With the stride of fixed-length records (read length + 1) the windows that cross a record's breaker
// sit at the end of every lane's segment and are skipped for the whole wave -- 21 of 151 positions at k = 21, 31 at k = 31;
// the result does not depend on the stride (a round of positions is skipped only when no lane has a valid window in it).
constexpr uint32_t SEG_MIN_STRIDE = 40, SEG_MAX_STRIDE = 168; // (what a wave's share of the 160 KB of LDS holds two strings of)
// Records longer than a lane's segment may be (2 x 250 / 2 x 300 reads: strides 251, 301) are shared by TWO or FOUR lanes
// (SketchArgs::seg_sub): a tile is then 32 or 16 records, its strings the same 10.7 KB at most, and the record's valid windows
// are dealt out evenly -- the first lanes take ceil((stride - K) / sub) start positions each, the last lane the rest with the K
// positions behind the last window -- so that every lane of the wave runs out of valid windows in the same round.
constexpr uint32_t SEG_MAX_RECORD = 4 * SEG_MAX_STRIDE;
constexpr uint32_t seg_sub_for(uint32_t stride) { return stride <= SEG_MAX_STRIDE ? 1u : stride <= 2 * SEG_MAX_STRIDE ? 2u : stride <= SEG_MAX_RECORD ? 4u : 0u; }
// RAGGED records (trimmed reads: no stride fits): SketchArgs::seg_sub == SEG_RAGGED with seg_stride == SEG_RAGGED_STRIDE.  A lane
// looks at SEG_RAGGED_STRIDE positions of the tile in cells of 32; a cell that holds valid windows becomes a WORK ITEM (from its
// first to its last valid window), the items are ordered by size class in the wave's LDS and dealt out 64 a round -- a round
// ends behind the longest of its items, and cells without a window cost nothing (fh_k2s.hip).  K = 25, 27..32 only (rounds of 32
// positions; the K that live off the LDS pipe, where every window not hashed is lookups saved: docs/MEASUREMENTS_r06.md 4).
constexpr uint32_t SEG_RAGGED = 0x100u, SEG_RAGGED_STRIDE = 128u;
constexpr bool seg_ragged_k(int k) { return k == 25 || (k >= 27 && k <= 32); }
constexpr uint32_t seg_tile_pos(uint32_t stride, uint32_t sub) { return sub == SEG_RAGGED ? 64u * stride : (64u / sub) * stride; } // start positions of a tile (a multiple of 16)
// (the segment kernels' leftover lists hold TRIPLES (t0, t1, c0): tiles [t0, t1), t0 from round c0 on -- a wave may stop inside a tile)

struct SketchArgs {
    const uint8_t *seq;   // packed stream (device), 16-byte aligned
    uint64_t len_total;   // readable bytes
    uint64_t p_begin;     // first k-mer start position of this launch (multiple of TILE_POS)
    uint64_t p_end;       // one past the last k-mer start position
    uint64_t base_pos;    // stream coordinate of seq[0]
    uint64_t seed;
    uint64_t hash_mask;   // ~0 unless the test hook is on
    uint64_t tau_lo;      // HASLO launches admit only hashes > tau_lo (second pass after a speculative threshold)
    Ctl *ctl;
    uint32_t tiles_total;
    uint32_t n_units;        // ceil(tiles_total / UNIT_TILES)
    uint32_t n_waves;        // waves of this launch (guides the pull size)
    uint32_t wave_budget;    // new hashes one wave may insert per launch before it stops (hard capacity guard)
    uint32_t n_left_in;      // leftover tile ranges from the previous (stopped) launch of this range
    uint32_t gate;           // 1: the launch does nothing unless ctl->spec_ok (the speculation before it succeeded)
    uint32_t unit_tiles;     // tiles per queue unit (UNIT_TILES; 1 for inputs too small to fill the chip with units of 2)
    uint32_t first_units;    // > 0 (first launch of a range): wave w starts on units [w, w + 1) x first_units without asking the
                             // queue, which the host has set to begin behind them.  Every pull is an atomic on ONE address
                             // (~14 ns each, serialised in L2): the 2 x 1953 of a 4 Mb genome's waves were 55 us of a launch
                             // that hashes for 10, the 12 000 of a 32 M-position prefix 100 us of 150.
    uint32_t static_only;    // the first units cover the whole range: no wave pulls from the queue
    uint32_t seg_stride;     // != 0: the segment kernel, tiles of 64 / seg_sub x seg_stride positions (p_begin a multiple of 16)
    uint32_t seg_sub;        // lanes per record of the segment kernel: 1 (strides <= 168), 2 (<= 336) or 4 (<= 672); K > 32: 1
    uint32_t max_units;      // segment kernel: units a pull takes at most (k2_sketch: MAX_UNITS; its tiles are a fifth the size)
    const uint32_t *left_in; // pairs (t0, t1); segment kernels: triples (t0, t1, c0)
    uint32_t *left_out;      // the same, capacity >= number of waves
};

// Many files per launch (fh_k2b.hip, fh_batch.hip): one descriptor per file of a batch.  Every file has a control block,
// table partition and shard lists of its own; the batch's tiles form one tile space that the launch's waves cut into equal
// contiguous stretches.
struct BatchFile {
    const uint8_t *seq;  // the file's packed stream (device), 16-byte aligned; bytes behind `len` are never read as sequence.
                         // BatchArgs.two_bit: the file's region in the two-bit form (fh_pack2.h), 64-byte aligned
    uint64_t len;        // k-mer start positions (= bytes of the packed stream)
    Ctl *ctl;            // the file's control block
    uint64_t tau;        // the one threshold the file is sketched at (EMPTY64: everything is admitted)
    uint32_t tile0;      // first tile of the file in the batch's tile space
    uint32_t n_tiles;    // ceil(len / TILE_POS)
};
struct BatchArgs {
    const BatchFile *files;
    uint32_t n_files, tiles_total, tiles_per_wave;
    uint32_t two_bit; // 1: the files are staged in the two-bit form (768 bytes per tile: phase A is two loads)
    uint64_t seed;
};
constexpr uint32_t TWO_BIT_TILE_BYTES = 768, TWO_BIT_CODES_BYTES = 512;

} // namespace fh
