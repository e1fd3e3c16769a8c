// fh_kernels.h -- launchers of the gfx950 kernels (fh_kernels.hip), used by the C ABI (fh_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "fh_device.h"

#define FH_NPARTS 4 // fh_k2.hip is compiled this many times, each part instantiating 32/FH_NPARTS values of K

namespace fh {

// (`blocks` is not used any more: every kernel launches ceil(a.n_waves / k2_waves_per_block(k)) workgroups)
hipError_t launch_k2(int k, const SketchArgs &a, int blocks, hipStream_t st);
// waves per workgroup of the sketch kernel for k-mer length k: 16 where one workgroup per CU shares the lookup tables
// (fh_k2.hip: K = 25..32), 4 otherwise.  The host needs it to know how many waves a launch really has.
#ifndef FH_SHARE_FROM
#define FH_SHARE_FROM 25
#endif
constexpr int k2_waves_per_block(int k) { return (k >= FH_SHARE_FROM && k <= 32) ? 16 : WAVES_PER_BLOCK; }
hipError_t launch_k2_part0(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2_part1(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2_part2(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2_part3(int k, const SketchArgs &a, int blocks, hipStream_t st);
// fh_k2s.hip: the segment form of the sketch kernel (SketchArgs::seg_stride != 0; K = 1..32, any seed, no test mask, no lower
// threshold; strides up to SEG_MAX_RECORD with seg_sub lanes per record), sixteen waves per workgroup whatever K
hipError_t launch_k2s_part0(int k, const SketchArgs &a, hipStream_t st);
hipError_t launch_k2s_part1(int k, const SketchArgs &a, hipStream_t st);
hipError_t launch_k2s_part2(int k, const SketchArgs &a, hipStream_t st);
hipError_t launch_k2s_part3(int k, const SketchArgs &a, hipStream_t st);
constexpr int K2S_WAVES_PER_BLOCK = 16;
// fh_k2ws.hip: the segment form for K = 33..64 (any seed / mask / lower threshold, like fh_k2w.hip), WAVES_PER_BLOCK waves per workgroup
hipError_t launch_k2ws_part0(int k, const SketchArgs &a, hipStream_t st);
hipError_t launch_k2ws_part1(int k, const SketchArgs &a, hipStream_t st);
hipError_t launch_k2ws_part2(int k, const SketchArgs &a, hipStream_t st);
hipError_t launch_k2ws_part3(int k, const SketchArgs &a, hipStream_t st);
constexpr int seg_waves_per_block(int k) { return k > 32 ? WAVES_PER_BLOCK : K2S_WAVES_PER_BLOCK; }
// Is the packed stream made of records of one length?  One wavefront looks at its first bytes and at records spread over
// the block: out[0] = the records' stride (length + 1, within [SEG_MIN_STRIDE, SEG_MAX_RECORD]) if every byte looked at where a
// breaker should be is one, else 0.  (A tuning hint only: the segment kernel is exact for any stride.)  `out`: device or pinned host memory.
hipError_t launch_seg_probe(const uint8_t *seq, uint64_t len, uint32_t *out, hipStream_t st);
// fh_k2w.hip: K = 33..64, again in FH_NPARTS translation units
hipError_t launch_k2w_part0(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2w_part1(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2w_part2(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2w_part3(int k, const SketchArgs &a, int blocks, hipStream_t st);
// fh_k2b.hip: the batch form of the tile kernel (K = 1..32, any seed): every file of a.files sketched at its own threshold
// into its own control block, n_waves waves of tiles_per_wave tiles each
hipError_t launch_k2b_part0(int k, const BatchArgs &a, uint32_t n_waves, hipStream_t st);
hipError_t launch_k2b_part1(int k, const BatchArgs &a, uint32_t n_waves, hipStream_t st);
hipError_t launch_k2b_part2(int k, const BatchArgs &a, uint32_t n_waves, hipStream_t st);
hipError_t launch_k2b_part3(int k, const BatchArgs &a, uint32_t n_waves, hipStream_t st);
hipError_t launch_k2b(int k, const BatchArgs &a, uint32_t n_waves, hipStream_t st);
constexpr int FH_MAX_K = 64;
hipError_t launch_prune_small(Entry *table, uint32_t *live, uint32_t *dead, uint32_t dead_cap, Ctl *ctl, uint32_t kind,
                              uint64_t size, uint64_t max_hash, uint32_t trigger, uint32_t force, uint32_t sort_out,
                              hipStream_t st);
// the fused epilogue of small sketches (fh_kernels.hip, k_small_epilogue)
constexpr uint32_t EPI_GATED = 1u;         // do nothing (but mirror the control block) unless ctl->spec_ok
constexpr uint32_t EPI_FLATTEN = 2u;       // append the shard lists of new inserts to the live list
constexpr uint32_t EPI_PRUNE_TRIGGER = 4u; // select if the live list is longer than `trigger`
constexpr uint32_t EPI_PRUNE_FORCE = 8u;   // select whatever its length
constexpr uint32_t EPI_SORT = 16u;         // ... and leave the survivors sorted (to_vec order)
constexpr uint32_t EPI_VERDICT = 32u;      // set ctl->spec_ok for the speculative range of n_units units just run
constexpr uint32_t EPI_GATHER = 64u;       // write the sorted sketch's columns to `out` (needs EPI_SORT)
constexpr uint32_t EPI_NEED_SPEC = 128u;   // EPI_GATHER: a speculation is unverified -- the sketch only counts if ctl->spec_ok
constexpr uint32_t EPI_RESET = 256u;       // EPI_GATHER: if the sketch counts, leave the handle as fh_reset would (slots cleared,
                                           // control block re-initialised) behind the mirror of the final control block
// what the gather epilogue reports in the mirrored control block's `sorted` word
constexpr uint32_t FIN_OK_RESET = 2u;      // everything queued did what it was queued for, the sketch is in `out`, the handle is reset
constexpr uint32_t FIN_OK = 3u;            // ... the handle is not reset (dropped-slot list overflowed: fh_reset sweeps the table)
struct EpiArgs {
    Entry *table;
    uint32_t *live, *dead;
    uint32_t dead_cap;
    Ctl *ctl;
    uint32_t kind;
    uint64_t size, max_hash;
    uint32_t trigger, flags, n_units;
    uint32_t check_units; // EPI_GATHER: units of the range still pending that must all have been pulled (0 = no range pending)
    uint64_t tau0;        // EPI_RESET: what init_ctl gets
    uint32_t hist_on;
    uint64_t *out;       // EPI_GATHER: hash | k-mer | first position | [high k-mer word] | count | extra columns, out_stride
    uint32_t out_stride; //   entries apart (device or pinned host memory)
    uint32_t wide;       // K > 32
    Ctl *h_ctl;          // pinned host mirror of the control block, written last (or null)
};
hipError_t launch_small_epilogue(const EpiArgs &a, hipStream_t st);
// the epilogue of a batch (fh_batch.hip): workgroup f runs the fused epilogue with args[f] -- flatten, select, sort, the
// sketch's columns and the mirrored control block to the host, the partition left reset -- and, should that not end in
// FIN_OK_RESET (an overflow somewhere), sweeps file f's whole partition so that the next batch finds it clean all the same.
// read_first: what the reset control block's admit path is told (Ctl::read_first).
hipError_t launch_batch_epilogue(const EpiArgs *args, uint32_t n_files, uint32_t read_first, hipStream_t st);
// one workgroup per file: table partition filled with empty entries, control block initialised and pointed at the partition
struct BatchPartition {
    Ctl *ctl;
    Entry *table;
    uint32_t *live, *shard_cnt, *shard_buf;
    CollRec *clog;
    uint32_t cap, live_cap, clog_cap, shard_cap;
};
hipError_t launch_batch_init(const BatchPartition *parts, uint32_t n_files, uint64_t size, uint32_t read_first, hipStream_t st);
hipError_t launch_reset_small(Entry *table, const uint32_t *live, const uint32_t *dead, Ctl *ctl, uint64_t tau0, uint64_t sel_size,
                              uint64_t tau_floor, uint32_t hist_on, hipStream_t st);
hipError_t launch_clear_slots(Entry *table, uint64_t cap, const uint32_t *live, const uint32_t *dead, const Ctl *ctl,
                              hipStream_t st);
// (o_kmer_hi: K > 32 only, else null)
hipError_t launch_gather(const Entry *table, const uint32_t *live, const Ctl *ctl, int k, uint64_t *o_hash,
                         uint32_t *o_count, uint32_t *o_extra, uint64_t *o_kmer, uint64_t *o_kmer_hi, uint64_t *o_pos,
                         uint32_t cap_out, hipStream_t st);
hipError_t launch_gather_rows(const uint64_t *hash, const uint64_t *kmer, const uint64_t *kmer_hi, const uint32_t *rows, uint32_t n,
                              uint64_t *out, hipStream_t st);
// sampling pre-pass of large sketches (fh_kernels.hip): tile runs [i * stride, i * stride + run_tiles) as a leftover list for the
// sketch kernel; histograms of the live entries by quarter-octave of hash value (3 x 256: entries, one occurrence, two)
hipError_t launch_fill_tile_runs(uint32_t *list, uint32_t n_runs, uint32_t stride, uint32_t run_tiles, uint32_t tiles_total,
                                 hipStream_t st);
hipError_t launch_live_count_hist(const Entry *table, const uint32_t *live, const Ctl *ctl, uint32_t *hist, hipStream_t st);
// fh_big.hip
hipError_t big_sort_tmp_bytes(uint32_t M, size_t *bytes);
hipError_t launch_big_prune(Entry *table, uint32_t *live, uint32_t *dead, uint32_t dead_cap, Ctl *ctl, uint32_t M,
                            uint32_t n_dead_now, uint32_t kind, uint64_t size, uint64_t max_hash, uint64_t *keys_a,
                            uint64_t *keys_b, uint32_t *slots_a, uint32_t *slots_b, void *tmp, size_t tmp_bytes,
                            uint32_t *keep_dev, hipStream_t st);
// in-stream prune without the sort (radix select + partition); scratch >= SEL_SCRATCH_BYTES
constexpr size_t SEL_SCRATCH_BYTES = 16384;
hipError_t launch_big_prune_select(Entry *table, uint32_t *live, uint32_t *dead, uint32_t dead_cap, Ctl *ctl, uint32_t M,
                                   uint32_t n_dead_now, uint32_t kind, uint64_t size, uint64_t max_hash, uint64_t *keys,
                                   uint32_t *slots, void *scratch, uint32_t *keep_dev, hipStream_t st);
hipError_t launch_rehash(const Entry *src, const uint32_t *src_live, uint32_t M, Entry *dst, uint32_t dst_cap,
                         uint32_t *dst_live, Ctl *ctl, const uint64_t *src_hi, uint64_t *dst_hi, hipStream_t st);
// fh_text.hip
// (line_end: scratch for one u32 per text line, line_cap entries; more lines than that -> the error flag)
hipError_t launch_fastq_pack(const uint8_t *text, uint64_t len, uint8_t *out, uint32_t *blk_a, uint32_t *blk_b,
                             uint32_t *totals, Ctl *ctl, uint32_t *err, uint32_t *line_end, uint32_t line_cap, hipStream_t st);
hipError_t launch_fasta_pack(const uint8_t *text, uint64_t len, uint32_t start_state, uint8_t *out, uint32_t *blk_a,
                             uint32_t *blk_b, uint32_t *totals, hipStream_t st);
// fh_bgzf.hip: the members of a BGZF batch inflated and CRC-checked, one wavefront each.  in_off / in_len: the member's
// DEFLATE bytes within `comp` (4-byte aligned buffer; reads stay inside the member); out_off: where its isize bytes of
// text go in `text`.  status[0] stays 0 or becomes (member << 8 | reason).
struct BgzfMember {
    uint32_t in_off, in_len, out_off, isize, crc;
};
hipError_t launch_bgzf_inflate(const uint8_t *comp, const BgzfMember *members, uint32_t n_members, uint8_t *text,
                               uint32_t *status, hipStream_t st);
// out[0]: where the last whole FASTQ record of text[0, total) ends (`last`: total); out[1]: 1 if no boundary was found
hipError_t launch_fastq_cut(const uint8_t *text, uint32_t total, uint32_t last, uint32_t *out, hipStream_t st);
// fh_bgzf.hip: plain gzip -- ONE DEFLATE stream, cut into chunks of `chunk_bits`, a wavefront each.  comp[0, n_bytes) are the
// stream's bytes from bit `first_bit` of comp[0] on (+ 4096 readable bytes behind them).  A chunk's record: where its
// first block was found (GZ_NONE: nowhere in its range), the block boundary it stopped at, the symbols it produced and why
// it stopped.  Symbols are 16-bit: a byte, or 0x8000 + i = "byte i of the 32 KiB in front of this chunk"; chunk c's go to
// sym[c * cap + GZ_WINDOW ...) behind GZ_WINDOW marker slots.
constexpr uint32_t GZ_WINDOW = 32768;
constexpr uint32_t GZ_MAX_CHUNKS = 8192; // chunks of one batch
constexpr uint32_t GZ_GROUPS = 64;      // stretches of the chain of chunks whose windows are worked out side by side
constexpr uint64_t GZ_NONE = ~0ull;
enum GzState : uint32_t { // low byte of GzChunk::state (the reason of a failure above it: BgzfFail in fh_bgzf.hip)
    GZ_IDLE = 0,         // never decoded (no block start in its range)
    GZ_NEXT = 1,         // stopped where a later chunk begins
    GZ_MEMBER_END = 2,   // the final block ended at end_bit
    GZ_OUT_OF_INPUT = 3, // the bytes ended inside the block that begins at end_bit
    GZ_FAILED = 4,
};
struct GzChunk {
    uint64_t start_bit, end_bit;
    uint32_t out_len, state;
};
// what the pass over the chain of chunks that really follow each other leaves behind (u32 words)
enum GzSummaryWord : uint32_t {
    GZS_STATUS = 0,   // 0, or (chunk << 8 | reason)
    GZS_N_LIVE = 1,   // chunks on the chain
    GZS_TOTAL = 2,    // bytes of text they hold (< 2^32)
    GZS_END_STATE = 3, // the last one's GzState
    GZS_END_BIT_LO = 4, GZS_END_BIT_HI = 5,
    GZS_HAVE_TRAILER = 6, GZS_CRC_WANT = 7, GZS_ISIZE_WANT = 8, // MEMBER_END: the eight bytes behind the stream, if they are there
    GZS_TRAILING = 9, // bytes behind the trailer
    GZS_VALID = 10,   // bytes of the member's text known in front of the next batch (<= GZ_WINDOW)
    GZS_CRC = 11,     // CRC-32 of this batch's text
    GZS_CUT = 12, GZS_CUT_BAD = 13, // (launch_fastq_cut's two words)
    GZS_WORDS = 16,
};
// scratch (all device memory): recs[n_chunks], sym[n_chunks * cap], win_in[n_chunks * GZ_WINDOW], live[4 * n_chunks], group_map, group_win,
// tile_map[text_cap / 4096 + 2], crc_tmp[text_cap / 65536 + 2].  window: the text in front of the batch (GZ_WINDOW bytes, the last `valid`
// of them real) on entry, in front of the next batch on return.  The batch's text goes to text[left ...) (`left` bytes in
// front of it are the partial record the previous batch ended with); GZS_CUT is launch_fastq_cut's answer for text[0, left + total).
struct GzBatch {
    const uint8_t *comp;
    uint64_t n_bytes, first_bit, chunk_bits;
    uint32_t n_chunks;
    uint64_t cap;
    GzChunk *recs;
    uint16_t *sym;
    uint32_t *claims;   // [n_regions]: who writes each chunk's stretch of `sym` (zero before a batch's first chunks are launched)
    uint32_t n_regions; // stretches there are (>= n_chunks)
    uint64_t *times;    // nullptr, or 3 x n_regions: when every chunk's wavefront was dispatched / had its bytes / was done (100 MHz ticks)
    uint8_t *win_in, *window;
    uint16_t *group_map; // GZ_GROUPS x GZ_WINDOW
    uint8_t *group_win;  // GZ_GROUPS x GZ_WINDOW
    uint32_t valid;
    uint32_t *live, *tile_map, *crc_tmp;
    uint8_t *text;
    uint32_t left;
    uint64_t text_cap; // room behind text + left
    uint32_t *summary;
};
// How much of a batch has arrived, in pinned host memory the chunks' wavefronts poll (a batch handed over in pieces is decoded
// by ONE launch that is there from the first piece on: a wavefront waits until its chunk and GZ_LOOKAHEAD bytes behind it
// are on the device).  The host writes `avail`, then `state`.
constexpr uint64_t GZ_LOOKAHEAD = 1ull << 20; // (no block is that long)
struct GzFeed {
    uint64_t avail; // bytes of the batch on the device, the carried ones in front included
    uint32_t state; // 0: more to come, 1: that is the whole batch, 2: ... and the stream ends with it
    uint32_t abort; // give up (the batch is abandoned)
};
// the chunks [c0, c0 + n) of the batch decoded (k_gz_chunks; feed != nullptr: avail_bytes / final come from there, and chunks
// that begin behind the batch's end drop out), with avail_bytes of the batch's bytes there (`final`: all the stream
// will ever have); then, once every chunk has been: the chain, the windows, the text, its CRC-32 and where its last record ends
hipError_t launch_gzip_chunks(const GzBatch &b, const GzFeed *feed, uint64_t avail_bytes, uint32_t c0, uint32_t n, bool final, hipStream_t st);
// (the CRC-32 kernels go to st_crc, behind `text_done` recorded on st once the text is there)
hipError_t launch_gzip_batch(const GzBatch &b, hipStream_t st, hipStream_t st_crc, hipEvent_t text_done);
// crc(A || B) from crc(A), crc(B) and |B| (zlib's crc32_combine)
uint32_t crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);
hipError_t launch_fill_table(Entry *table, uint64_t cap, hipStream_t st);
// (keep_text_bases: everything but the count of sequence bytes the text packers have emitted so far)
// (sel_size / tau_floor / hist_on: the in-launch threshold refresh, Ctl in fh_device.h)
hipError_t launch_init_ctl(Ctl *ctl, uint64_t tau0, hipStream_t st, bool keep_text_bases = false, uint64_t sel_size = 0,
                           uint64_t tau_floor = 0, bool hist_on = false);
hipError_t launch_set_tau(Ctl *ctl, uint64_t tau, hipStream_t st);
hipError_t launch_debug_add_counts(Entry *table, const uint32_t *live, const Ctl *ctl, uint64_t add_count, uint64_t add_extra, hipStream_t st);
hipError_t launch_set_table(Ctl *ctl, Entry *table, uint32_t *live, CollRec *clog, uint32_t cap, uint32_t live_cap,
                            uint32_t clog_cap, uint32_t *shard_cnt, uint32_t *shard_buf, uint32_t shard_cap, uint64_t *kmer_hi,
                            hipStream_t st);
hipError_t launch_live_flatten(Ctl *ctl, hipStream_t st);
hipError_t launch_queue_reset(Ctl *ctl, uint32_t new_range, uint32_t soft_limit, uint32_t read_first, hipStream_t st,
                              bool set_tau = false, uint64_t tau = 0, bool gate = false, uint32_t first_total = 0);
hipError_t launch_read_probe(const void *p, uint64_t bytes, uint32_t *sink, hipStream_t st);
hipError_t launch_synth_genome(uint8_t *out, uint64_t len, uint64_t seed, hipStream_t st);
hipError_t launch_synth_reads(uint8_t *out, const uint8_t *genome, uint64_t genome_len, uint64_t first_read,
                              uint64_t n_reads, uint32_t read_len, uint64_t seed, uint32_t sub_ppm, uint32_t n_ppm,
                              hipStream_t st);

} // namespace fh
