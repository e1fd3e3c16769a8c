/*
 * finch_hip.h -- C ABI of libfinch_hip.so, the MI355X (gfx950) MinHash sketching engine that sits
 * behind finch's sketching interface.
 *
 * This is the drop-in boundary for exactly one path of onecodex/finch-rs (citations relative to the
 * reference tree):
 *
 *     finch::sketch_files()                      lib/src/lib.rs:29-49
 *       -> sketch_stream()                       lib/src/lib.rs:51-94
 *         -> SketchScheme::process()             lib/src/sketch_schemes/mash.rs:67-80, scaled.rs:65-78
 *           -> MashSketcher/ScaledSketcher::push mash.rs:34-63, scaled.rs:37-61
 *         -> total_bases_and_kmers(), to_vec()   mash.rs:82-102, scaled.rs:80-100
 *
 * A Rust `impl SketchScheme for HipSketcher` binds these symbols 1:1:
 *     SketchParams::create_sketcher()  (mod.rs:86-113)  -> fh_new
 *     SketchScheme::process()                          -> fh_push_block (record bytes + 1 breaker byte)
 *     SketchScheme::total_bases_and_kmers()            -> fh_finish (total_kmers; total_bases is a host counter)
 *     SketchScheme::to_vec()                           -> fh_finish + fh_copy_out (ascending hash)
 *     drop                                             -> fh_free
 *
 * Plain C: pointers and sizes only, no exceptions cross the boundary.  Every function returns
 * FH_OK (0) or a negative FH_ERR_* code; fh_last_error() returns a thread-local message.
 * A handle owns one device, its HIP streams and all device memory; a handle is NOT thread-safe,
 * different handles are independent (one per file / per rayon worker / per GPU).
 * There is no CPU fallback: if no HIP device is usable, fh_new fails with FH_ERR_NO_DEVICE.
 */
#ifndef FINCH_HIP_H
#define FINCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FH_OK 0
#define FH_ERR_INVALID (-1)     /* bad argument (null pointer, unknown kind, ...) */
#define FH_ERR_NO_DEVICE (-2)   /* no usable HIP device / device index out of range */
#define FH_ERR_HIP (-3)         /* a HIP runtime call failed (message has the HIP error string) */
#define FH_ERR_STATE (-4)       /* call not valid in the handle's current state */
#define FH_ERR_CAPACITY (-5)    /* device table / collision log capacity exceeded */
#define FH_ERR_UNSUPPORTED (-6) /* valid request this build cannot serve on the device */

#define FH_MAX_KMER_LENGTH 64

#define FH_KIND_MASH 0   /* SketchParams::Mash   (mod.rs:55-61)  -> MashSketcher::new(size, k, seed)          */
#define FH_KIND_SCALED 1 /* SketchParams::Scaled (mod.rs:62-67)  -> ScaledSketcher::new(size, scale, k, seed) */

/* POD mirror of the sketcher constructor arguments (mash.rs:21, scaled.rs:22). */
typedef struct fh_params {
    uint32_t kind;        /* FH_KIND_MASH | FH_KIND_SCALED */
    uint32_t k;           /* kmer_length, 1..64 on the device (FH_MAX_KMER_LENGTH; beyond: FH_ERR_UNSUPPORTED).  k <= 32 runs the
                             single-word kernels the roofline is quoted on, 33..64 the two-word ones (fh_k2w.hip) */
    uint64_t size;        /* kmers_to_sketch */
    uint64_t seed;        /* hash_seed */
    double scale;         /* scaled only; max_hash = u64::MAX / ((1/scale) as u64) */
    uint64_t max_launch;  /* 0 = default (16 M); max k-mer start positions in flight at once (bounds the launched waves):
                             bounds the grid and sizes the device table (table slots ~ 2 x (this + 4 x size)) */
    uint64_t hash_mask;   /* 0 = none (all bits); test hook: AND every hash with this mask (forces collisions) */
    uint64_t stage_bytes; /* 0 = default (64 MiB); size of each of the two pinned/device staging buffers of fh_push_block */
} fh_params;

typedef struct fh_sketcher fh_sketcher;

/* number of visible HIP devices (0 if none / runtime unusable) */
int fh_device_count(void);
const char *fh_last_error(void);
/* library/ABI version, bumped on any change of this header's functions (5: the batch sketcher with its two-bit input form, fh_set_option; the round-5
 * additions fh_set_record_stride, fh_debug_segments, fh_process_records_in, fh_debug_add_counts, fh_debug_gzip_feed_timeouts) */
#define FH_ABI_VERSION 5
int fh_abi_version(void);

/* --- configuration: ONE surface ---
 * Everything that selects between the library's (all exact) code paths, sizes a buffer for a test, sizes thread teams and
 * pools or switches a trace on is a named option: set here, process-wide, or listed in the ONE environment variable the
 * library reads, FH_DEBUG="name=value,name=value" (a name alone means "1"; an explicit fh_set_option wins).  No option
 * changes a sketch.  value == NULL: back to "not set".  An option is looked at when the thing it configures is created or
 * first used (a sketcher's options at fh_new; most A/B switches once per process), so set options before the first
 * fh_new.  fh_option_list(): "name<TAB>what it does<NEWLINE>" for every option -- the authoritative list (README.md prints
 * it); fh_get_option: the value in force, NULL if not set (or no such option).  FH_ERR_INVALID: no such option. */
int fh_set_option(const char *name, const char *value);
const char *fh_get_option(const char *name);
const char *fh_option_list(void);

/* create_sketcher: allocate the device-resident sketch state on `device`. NULL on error. */
fh_sketcher *fh_new(const fh_params *params, int device);
/* Drop a sketcher.  finch creates one per file and drops it after to_vec (lib.rs:58-79); since a sketcher owns
 * gigabytes of device memory, fh_free resets it and keeps up to `pool` (option, default 64, 0 = never) of them
 * parked -- `pool_bytes` of device memory at most (option; default 24 GiB or a tenth of the device, whichever is less; an
 * embedding process that shares the device with another allocator calls fh_release_cached) -- and fh_new hands a parked one back
 * when the parameters and the device match (~0.1 ms instead of ~5 ms).  fh_release_cached frees what is parked; the
 * library does so itself before any of its own allocations fails for lack of memory. */
void fh_free(fh_sketcher *s);
void fh_release_cached(void);
/* forget everything pushed so far (state as after fh_new); keeps device memory */
int fh_reset(fh_sketcher *s);

/* Global stream coordinate of the next pushed byte.  Only matters for sharded inputs: it makes
 * "first occurrence" (which k-mer bytes are retained on a 64-bit hash collision, mash.rs:52-56)
 * well defined across shards.  Default 0; advanced automatically by pushes. */
int fh_set_stream_offset(fh_sketcher *s, uint64_t offset);

/* process(): push one block of *sequence bytes* (host memory).  The block holds whole records, raw /
 * un-normalised, each record followed by one breaker byte (any byte that is not whitespace and not in
 * ACGTUacgtu, e.g. '\0').  The library performs normalize(false) (whitespace skipped, case folded, U->T,
 * everything else breaks k-mers), reverse-complement, canonical k-mers, murmurhash3_x64_128 and the
 * bottom-n / scaled admission on the device.  k-mers never span two pushed blocks.
 * Asynchronous: returns once the bytes are staged; the caller may reuse `bytes` immediately. */
int fh_push_block(fh_sketcher *s, const uint8_t *bytes, uint64_t len);
/* process() as the trait has it (mash.rs:67-80), ONE record per call: `seq` = the record's raw sequence() bytes, no breaker.
 * The library copies them -- blanks dropped on the way, as normalize(false) would -- straight into its pinned staging buffer,
 * puts the breaker behind them and commits a full buffer by itself; fh_finish / fh_sync / any other push commit what is
 * waiting.  This is the binding of choice at the trait level: one call and one copy per record,
 * nothing to buffer on the caller's side.  fh_process_records does the same for n records of one buffer
 * (base + offsets[i], lens[i]); fh_total_bases returns the sum of the lengths fh_process has seen since the last reset
 * (total_bases, mash.rs:72 -- the host counter of SURVEY B2, kept here so that the binding need not). */
int fh_process(fh_sketcher *s, const uint8_t *seq, uint64_t len);
int fh_process_records(fh_sketcher *s, const uint8_t *base, const uint64_t *offsets, const uint64_t *lens, uint64_t n);
/* fh_process_records for records the caller has not checked: a record that does not lie inside base[0, base_len) ends the call
 * with FH_ERR_INVALID (the records in front of it have been taken); *bases (may be NULL) = the lengths taken, summed. */
int fh_process_records_in(fh_sketcher *s, const uint8_t *base, uint64_t base_len, const uint64_t *offsets, const uint64_t *lens,
                          uint64_t n, uint64_t *bases);
int fh_total_bases(fh_sketcher *s, uint64_t *total_bases);
/* Same with flags.  FH_PUSH_CONTINUE: this block continues the record the previous push ended in (a
 * record longer than the caller's buffer, e.g. a chromosome): k-mers span the boundary of the two pushes. */
#define FH_PUSH_CONTINUE 1u
int fh_push_block_ex(fh_sketcher *s, const uint8_t *bytes, uint64_t len, uint32_t flags);

/* Device-side FASTQ parsing: the caller fills the handle's pinned staging buffer with RAW
 * plain 4-line FASTQ text cut at a record boundary (it starts with a '@' header line and ends after a quality
 * line) and the library finds the sequence lines on the device (line index mod 4), drops CR, turns each
 * sequence line's newline into the record breaker and sketches the result.  Replaces needletail's record
 * splitting for that format (lib.rs:60-68); FASTA has fh_push_fasta_text, anything else (blank lines between FASTQ
 * records, multi-line FASTQ) goes through fh_push_block.  fh_text_buffer hands out the buffer to fill next (capacity = stage_bytes); fh_push_fastq_text
 * consumes its first `len` bytes.  FH_ERR_INVALID if the text is not 4-line FASTQ. */
int fh_text_buffer(fh_sketcher *s, uint8_t **buf, uint64_t *cap);
/* Both staging buffers at once, for a reader that fills one while the library works on the other: bufs[*next] is the one
 * the next fh_push_fastq_text / fh_push_fasta_text consumes, after that the two alternate.  A buffer may be refilled as
 * soon as the push that consumed it has returned (those two calls are done with the host copy by then), also while the
 * push of the other buffer is still running on another thread; the handle itself stays single-threaded. */
int fh_text_buffers(fh_sketcher *s, uint8_t *bufs[2], uint64_t *cap, int *next);
int fh_push_fastq_text(fh_sketcher *s, uint64_t len);
/* Optional: start the host-to-device copy of staging buffer `slot` (0 / 1 of fh_text_buffers, filled with `len` bytes) right
 * away, on the handle's copy stream -- the one call a SECOND thread (the reader that has just filled the buffer) may make
 * while a push of the other buffer is running.  The fh_push_fastq_text / fh_push_fasta_text that later consumes the slot with the same `len`
 * finds its text already on the way; without the call the push copies by itself.  This is what keeps the PCIe link busy
 * across pushes (a push also waits for its record-splitting kernel). */
int fh_text_prefetch(fh_sketcher *s, int slot, uint64_t len);
/* BGZF-compressed FASTQ, inflated on the device (one wavefront per member, CRC-32 checked there too).  The caller fills
 * the text buffer of fh_text_buffers with n_members fh_bgzf_member records followed by the members' raw DEFLATE bytes
 * (`bytes` in all): in_off / in_len = a member's DEFLATE data counted from the start of the buffer (after the 18-byte BGZF
 * header, before the 8-byte trailer), out_off = the running sum of the isize before it, isize and crc32 = its trailer.
 * The inflated text of a batch need not end with a record: what follows the last whole one waits on the device for
 * the next push (so a batch's text plus one record must fit fh_bgzf_text_capacity); FH_BGZF_LAST = the file ends here.  Replaces
 * the decompress-then-parse step of needletail's reader (lib.rs:60) for bgzip'd reads.  FH_ERR_INVALID: damaged
 * member (named in fh_last_error) or text that is not plain 4-line FASTQ; the sketcher has to be reset then. */
typedef struct fh_bgzf_member {
    uint32_t in_off, in_len, out_off, isize, crc32;
} fh_bgzf_member;
#define FH_BGZF_LAST 1u
#define FH_BGZF_MORE 2u /* only copy these members over: they are inflated together with those of the following pushes, by the
                         * first one without this flag (a pinned buffer of poorly compressed members holds too few of them to
                         * fill the device); the table entries in the buffer are rewritten by the call */
int fh_push_bgzf_fastq(fh_sketcher *s, uint64_t bytes, uint32_t n_members, uint32_t flags);
/* Plain gzip (one DEFLATE stream, no index: what `gzip` and most sequencers' pipelines write) of FASTQ text, inflated on the
 * device.  The caller puts the next bytes of the member's DEFLATE stream -- what follows the RFC 1952 header -- into the
 * text buffer of fh_text_buffers; FH_GZ_FIRST: they are the stream's first, FH_GZ_LAST: the input ends with them.  A batch
 * is what the pushes up to and including the first one without FH_GZ_MORE have put into ONE buffer, each push its `bytes`
 * behind those of the push before (fh_gzip_batch_capacity in all, at most); with FH_GZ_MORE a push only has its piece
 * copied over and the chunks in front of it decoded, so the device works while the caller reads on.  The batch is cut into
 * chunks, a wavefront each: every chunk is decoded from the first block start found in it into 16-bit symbols (a byte, or
 * "byte i of the 32 KiB in front of this chunk"), the chunks that really continue each other are chained and their markers
 * looked up (the two-pass scheme of pugz / rapidgzip).  What lies behind the last block boundary reached stays on the device
 * for the next batch, like the partial FASTQ record the text ended with.  *member_done: the final block has been decoded
 * and the text's CRC-32 and size are those of the trailer; *trailing: bytes of this batch behind the trailer (another
 * member, or garbage: the caller's business).  Replaces the decompress-then-parse step of needletail's reader (lib.rs:60)
 * for gzip'd reads.  FH_ERR_INVALID: a damaged stream, a block longer than a batch, text more than about 8 x its DEFLATE
 * bytes, or text that is not plain 4-line FASTQ; the sketcher has to be reset then (the host-side inflate is the judge of such
 * files). */
#define FH_GZ_FIRST 1u
#define FH_GZ_LAST 2u
#define FH_GZ_MORE 4u
int fh_push_gzip_fastq(fh_sketcher *s, uint64_t bytes, uint32_t flags, uint32_t *member_done, uint64_t *trailing);
/* bytes one batch of fh_push_gzip_fastq may hold */
int fh_gzip_batch_capacity(fh_sketcher *s, uint64_t *cap);
/* Text one batch may inflate to, the carried-over partial record included (8 x stage_bytes, at most 1 GiB: a wavefront
 * per member only fills the device with thousands of members in flight). */
int fh_bgzf_text_capacity(fh_sketcher *s, uint64_t *cap);
/* Device-side FASTA parsing: the staged text is raw (multi-line) FASTA.  A line that begins with '>' is a header
 * (dropped; it ends the previous record: one breaker byte is emitted), every other line is sequence: its bytes are
 * kept except ' ', '\t', '\r' and the newline itself, so k-mers span line breaks exactly as they do after
 * needletail's FASTA reader + normalize(false) (lib.rs:60-68, mash.rs:73).  `start_state` says what the chunk begins
 * in the middle of: 0 = a line start (chunks are normally cut after a newline), 1 = a sequence line, 2 = a header
 * line.  With FH_PUSH_CONTINUE the chunk continues the previous fh_push_fasta_text chunk (k-mers span the cut).
 * total_bases (raw sequence-region lengths, mash.rs:72) is the caller's to count: it needs no per-base work. */
int fh_push_fasta_text(fh_sketcher *s, uint64_t len, uint32_t start_state, uint32_t flags);
/* One FASTA input split over several sketchers (finch_sketch_file_sharded): the chunk the NEXT fh_push_fasta_text
 * pushes continues a record whose preceding text went to another sketcher.  `halo` holds the last n (<= k-1) packed
 * sequence bytes before the cut (whitespace already dropped): k-mers that span the cut are formed here, k-mers inside
 * the halo are the other sketcher's.  Their stream coordinates are the n coordinates below the chunk's
 * (fh_set_stream_offset).  Applies to one push; excludes FH_PUSH_CONTINUE. */
int fh_set_text_halo(fh_sketcher *s, const uint8_t *halo, uint32_t n);
/* Zero-copy form of fh_push_block_ex: the caller writes packed-stream bytes (sequence bytes + one breaker byte per
 * record, whitespace already removed) straight into the buffer handed out by fh_text_buffer / fh_text_buffers and commits
 * the first `len` of them.  Same flags as fh_push_block_ex.  A filler thread that has called fh_text_prefetch(slot, len) on the
 * buffer finds its copy honoured (a push without FH_PUSH_CONTINUE): the link works while the previous push is sketched.  The
 * buffer of the push BEFORE the one that has just returned is free to be filled again (this push waited for that one's
 * launches, which had waited for its copy). */
int fh_push_staged(fh_sketcher *s, uint64_t len, uint32_t flags);
/* sequence bytes seen by fh_push_fastq_text so far (what total_bases counts, mash.rs:72); valid after fh_finish */
int fh_text_bases(fh_sketcher *s, uint64_t *total_bases);

/* Same for a block that is already resident in this device's HBM (16-byte aligned, no whitespace
 * bytes: the packed stream produced by fh_push_block's staging or by fh_synth_reads_device).
 * The memory must stay valid until fh_finish/fh_sync returns. */
int fh_push_device(fh_sketcher *s, const void *dev_bytes, uint64_t len);

/* What the caller knows about the records of the packed streams it pushes (a property of its data: it survives fh_reset).
 * stride = record length + 1 (the breaker): every record of every block pushed from now on has that length -- reads of one
 * length, which is what sequencers write; 1 = records are not of one length, do not look; 0 (the default) = not known: the
 * library looks itself.  A block of 16 MiB or more is probed (one wavefront: its first bytes and 128 records spread over it)
 * BEHIND its own launches, without a wait, and the NEXT block of the handle goes by the newest answer that has arrived (the
 * blocks of one input and the passes over one buffer share their read length; an answer is read only once its probe's event
 * has completed).  Only a handle that has no answer yet waits for one, and only for a block of 256 MiB or more (0.1-0.3 ms,
 * once) -- so a handle's first block below 256 MiB runs the tile kernel.  With a stride the block is sketched by a kernel that
 * does not hash the k positions of every record whose window crosses its breaker (fh_k2s.hip; the reference's canonical_kmers
 * yields len - k + 1 windows per record, mash.rs:76): strides 40..168 one lane per record, 169..336 two, 337..672 four
 * (k > 32: 40..168 only); any seed.  A TUNING hint: the sketch is the same bit for bit whatever is said here, also when it is
 * wrong or stale (tests/test_gpu_segments.py).  Strides outside those ranges are taken as 1. */
int fh_set_record_stride(fh_sketcher *s, uint32_t stride);
/* debug / tests: launches of the segment kernel, blocks probed for a stride, the stride of the last block (0: none) */
int fh_debug_segments(fh_sketcher *s, uint64_t *launches, uint64_t *probes, uint32_t *stride);

/* wait for all pushed work; surfaces deferred device errors */
int fh_sync(fh_sketcher *s);

/* to_vec() part 1: finalise (bottom-n select + sort ascending) and report the sizes.
 * n_out = number of retained hashes, total_kmers = number of valid k-mers pushed (mash.rs:35). */
int fh_finish(fh_sketcher *s, uint64_t *n_out, uint64_t *total_kmers);
/* to_vec() part 2: copy the sketch out, ascending by hash.  Any pointer may be NULL.
 * counts/extra are saturated at u32::MAX (mash.rs:46-49).  kmers receives n_out*k ASCII bytes
 * (uppercase canonical k-mer of the first occurrence).  first_pos = stream coordinate of that occurrence. */
int fh_copy_out(fh_sketcher *s, uint64_t *hashes, uint32_t *counts, uint32_t *extra_counts, uint8_t *kmers,
                uint64_t *first_pos);
/* The same with (hash, count, extra_count) interleaved as the reference's KmerCount holds them
 * (sketch_schemes/mod.rs:16-22, minus the k-mer bytes, which go to `kmers` as above): to_vec() fills its
 * Vec<KmerCount> from one array.  Any pointer may be NULL. */
typedef struct fh_kmer_count {
    uint64_t hash;
    uint32_t count;
    uint32_t extra_count;
} fh_kmer_count;
int fh_copy_out_records(fh_sketcher *s, fh_kmer_count *records, uint8_t *kmers, uint64_t *first_pos);

/* The k-mer bytes of selected records only (rows = indices into the ascending sketch): a caller that filters a 2 M-hash
 * oversketch down to 10 000 hashes (filter_counts + truncate, lib.rs:82-83) needs the bytes of the survivors, not of all. */
int fh_copy_out_kmers(fh_sketcher *s, const uint32_t *rows, uint64_t n_rows, uint8_t *kmers);
/* The same for whole records: records[i] / kmers[i*k..] = row rows[i] of the result (either may be NULL). */
int fh_copy_out_rows(fh_sketcher *s, const uint32_t *rows, uint64_t n_rows, fh_kmer_count *records, uint8_t *kmers);
/* The finished result's count columns where they lie (n entries each, ascending by hash like every copy-out; valid until
 * the next fh_reset / fh_free / merge): what a filter pass over a 2 M-hash oversketch reads, without a copy of it.
 * FH_ERR_STATE after a merge (the result is a record vector then: use fh_copy_out). */
int fh_result_counts(fh_sketcher *s, const uint32_t **counts, const uint32_t **extra_counts, uint64_t *n);

/* Host-side merge of partial sketches (multi-GPU read-block sharding): union, counts
 * summed (saturating), k-mer of the smallest first_pos, re-select per kind.  Both must be finished.
 * After the call dst holds the merged sketch (fh_copy_out works on it); dst's total_kmers += src's. */
int fh_merge(fh_sketcher *dst, const fh_sketcher *src);
/* Same, from raw arrays (a partial sketch received from another process). */
int fh_merge_arrays(fh_sketcher *dst, uint64_t n, const uint64_t *hashes, const uint32_t *counts,
                    const uint32_t *extra_counts, const uint8_t *kmers, const uint64_t *first_pos,
                    uint64_t total_kmers);

/* Handle-free form of the same merge (pure host code, no device needed): merges partial sketch B into
 * partial sketch A.  out_* must hold nA+nB records; *n_out receives the merged count.  `scale` is only
 * read for FH_KIND_SCALED. */
int fh_merge_partials(uint32_t kind, uint64_t size, double scale, uint32_t k, uint64_t nA, const uint64_t *hashesA,
                      const uint32_t *countsA, const uint32_t *extraA, const uint8_t *kmersA, const uint64_t *posA,
                      uint64_t nB, const uint64_t *hashesB, const uint32_t *countsB, const uint32_t *extraB,
                      const uint8_t *kmersB, const uint64_t *posB, uint64_t *n_out, uint64_t *out_hashes,
                      uint32_t *out_counts, uint32_t *out_extra, uint8_t *out_kmers, uint64_t *out_pos);

/* N-way form over the sharding wire format (finch_rs_amd/sharding.py pack_partial; what the ranks of a sharded job
 * gather on rank 0): one buffer of int64 words per partial sketch,
 *   [0] n   [1] total_kmers   then, each padded to pad_n entries: hashes (u64), counts, extra_counts (as i64),
 *   first positions (u64), k-mers (ASCII, ceil(k/8)*8 bytes each).
 * out_* must hold the sum of the partial sizes (mash: `size` records are enough).  Same result as chaining
 * fh_merge_partials. */
int fh_merge_wire(uint32_t kind, uint64_t size, double scale, uint32_t k, uint64_t pad_n, uint32_t n_parts,
                  const int64_t *const *bufs, uint64_t *n_out, uint64_t *out_hashes, uint32_t *out_counts,
                  uint32_t *out_extra, uint8_t *out_kmers, uint64_t *out_pos, uint64_t *total_kmers);

/* ONE input resident as n read blocks in the HBM of n devices -> one sketch (north_star: read-block sharding with a
 * host-side merge and no collective; the fan-out of sketch_files, lib.rs:34-36, applied to the blocks of a single file).
 * Handle i -- a sketcher on the device that holds block i -- is reset and sketches dev_blocks[i] (as fh_push_device:
 * 16-byte aligned, lens[i] bytes of packed stream whose first byte has stream coordinate stream_offsets[i]) on a thread
 * of its own, kept by the library between calls; the calling thread runs block 0 and then merges the n partial sketches
 * into handles[0] by fh_merge's rule.  On return every handle is finished: handles[0] holds the merged sketch (fh_finish
 * reports its size and the k-mer total of all blocks, fh_copy_out* deliver it), the others their partial ones.  The
 * handles must be distinct and have the same sketch parameters; several may share a device.  EVERY handle is reset on entry
 * (whatever the caller had pushed into it is discarded: the call sketches the blocks and nothing else).  The first block that
 * fails decides the return code, and fh_last_error() names it ("block i (device d): ..."); the handles of the other blocks are
 * left finished with their partial sketches, handles[0] then holds block 0's alone.  The caller's current HIP device is the
 * same on return as on entry, whichever way the call ends.  The library's threads (not the caller's) run on the CPUs of the
 * NUMA node their device is attached to where sysfs names one (option no_numa_pin: wherever the scheduler puts them). */
int fh_sketch_device_blocks(fh_sketcher *const *handles, const void *const *dev_blocks, const uint64_t *lens,
                            const uint64_t *stream_offsets, uint32_t n);

/* --- many sketches per launch: a BATCH of files (finch::sketch_files, lib/src/lib.rs:29-49, whose par_iter over the files
 * -- lib.rs:34-36 -- is the one place the reference is parallel) ---
 * Through an fh_sketcher a file costs a host-to-device copy, three kernel launches and a synchronisation of its own:
 * ~150 us of latency-bound device time for the ~6 us a 4 Mb genome takes to hash.  A batch handle sketches ALL the files its
 * caller has staged with one copy, one launch of the sketch kernel over the files' tiles (fh_k2b.hip), one launch of the
 * epilogue (a workgroup per file: select, sort, to_vec straight into pinned host memory, state left reset) and one
 * synchronisation.  Mash sketches of 1..3000 hashes, k = 1..32, any seed; everything else -- and every file the batch path
 * cannot vouch for -- goes through an fh_sketcher.
 *
 *   fh_batch_new(params, device, max_files, stage_bytes)   two slots, each a pinned staging buffer of stage_bytes
 *   fh_batch_stage(b, slot, &buf, &cap)                    where the caller writes the PACKED streams of its files (the
 *                                                          format of fh_push_block: sequence bytes, one breaker byte
 *                                                          behind every record), each at a 16-byte aligned offset
 *   fh_batch_submit(b, slot, offsets, lens, n)             asynchronous; file i = buf[offsets[i], +lens[i]), offsets ascending
 *   fh_batch_wait(b, slot, status)                         status[i] = 0: sketch i is ready (fh_batch_result /
 *                                                          fh_batch_copy_out*); 1: NOT TAKEN -- the file holds fewer than
 *                                                          `size` distinct k-mers below the threshold it was sketched at
 *                                                          (low-complexity or tiny input), or two of its k-mers share a
 *                                                          64-bit hash, or a capacity was exceeded: sketch it through an
 *                                                          fh_sketcher, which handles all of that.  Never an approximate
 *                                                          sketch: a taken file's hashes, counts and k-mer bytes are the
 *                                                          reference's (mash.rs:34-63, 86-102), bit for bit.
 *   fh_batch_submit_packed(b, slot, offsets, lens, n)      the same with every file staged in the TWO-BIT form: 0.375 bytes
 *                                                          per position on the link instead of 1 (the link is what bounds a
 *                                                          batch of genomes).  File i occupies fh_batch_packed_bytes(lens[i])
 *                                                          bytes at the 64-byte aligned offsets[i]; lens[i] = its positions.
 *                                                          Per tile of 2048 positions 768 bytes: 64 little-endian u64 of codes
 *                                                          (position 32 g + i of the tile at bits [2 i, 2 i + 2) of word g:
 *                                                          A 0, C 1, G 2, T/U 3, either case) and 64 u32 of "is a base" bits
 *                                                          (bit i of word g; clear = the byte breaks k-mers, as every byte
 *                                                          but ACGTUacgtu and every record's breaker does); positions behind
 *                                                          the file's last are zero and one all-zero tile follows the last.
 *                                                          fh_batch_pack writes that form from a packed byte stream.
 * The two slots alternate: fill slot 1 while slot 0 is in flight.  A batch handle is single-threaded like an fh_sketcher;
 * different handles are independent (one per worker thread). */
typedef struct fh_batch fh_batch;
fh_batch *fh_batch_new(const fh_params *params, int device, uint32_t max_files, uint64_t stage_bytes);
void fh_batch_free(fh_batch *b);
int fh_batch_stage(fh_batch *b, int slot, uint8_t **buf, uint64_t *cap);
int fh_batch_submit(fh_batch *b, int slot, const uint64_t *offsets, const uint64_t *lens, uint32_t n_files);
int fh_batch_submit_packed(fh_batch *b, int slot, const uint64_t *offsets, const uint64_t *lens, uint32_t n_files);
uint64_t fh_batch_packed_bytes(uint64_t len);
int fh_batch_pack(const uint8_t *stream, uint64_t len, uint8_t *region, uint64_t region_cap);
int fh_batch_wait(fh_batch *b, int slot, uint8_t *status);
/* file i of the batch last waited for in `slot`: hashes retained and valid k-mers seen (mash.rs:35); then the sketch,
 * ascending by hash, as fh_copy_out / fh_copy_out_records deliver it (any pointer may be NULL) */
int fh_batch_result(fh_batch *b, int slot, uint32_t i, uint64_t *n_out, uint64_t *total_kmers);
int fh_batch_copy_out(fh_batch *b, int slot, uint32_t i, uint64_t *hashes, uint32_t *counts, uint32_t *extra_counts, uint8_t *kmers,
                      uint64_t *first_pos);
int fh_batch_copy_out_records(fh_batch *b, int slot, uint32_t i, fh_kmer_count *records, uint8_t *kmers);
/* measurement: HIP events around every batch's sketch launch; their sum, the launches and the positions they covered since
 * the last call; files taken / not taken since fh_batch_new */
int fh_batch_set_profiling(fh_batch *b, int enable);
int fh_batch_kernel_time(fh_batch *b, double *total_ms, uint64_t *launches, uint64_t *positions);
int fh_batch_counters(fh_batch *b, uint64_t *taken, uint64_t *not_taken);

/* --- measurement support (bench.py) --- */
/* when enabled, every sketch-kernel launch is bracketed by HIP events on the handle's stream */
int fh_set_profiling(fh_sketcher *s, int enable);
/* sum of event-measured durations (ms) and number of sketch-kernel launches since the last reset,
 * and the k-mer start positions those launches covered */
int fh_kernel_time(fh_sketcher *s, double *total_ms, uint64_t *launches, uint64_t *positions);

/* diagnostics: sketch-kernel launches, relaunches after a capacity stop, device-wide selections */
int fh_debug_counters(fh_sketcher *s, uint64_t *launches, uint64_t *relaunches, uint64_t *big_prunes);
/* gzip batches of this process whose piece-fed decoding launch gave up waiting for its bytes (fh_push_gzip_fastq with FH_GZ_MORE;
 * the handle then decodes its batches only once they are complete).  0 on every system this was run on: a benchmark that sees
 * anything else is measuring the fallback. */
uint64_t fh_debug_gzip_feed_timeouts(void);
/* test hook: add_count / add_extra are added to the forward-strand / reverse-strand counters of every hash the sketcher holds
 * at this moment (everything pushed so far is sketched first), so that a test can drive the reported u32 counts into their
 * saturation (mash.rs:45-50: count.0 / count.1 are saturating adds) without 2^32 occurrences of a k-mer */
int fh_debug_add_counts(fh_sketcher *s, uint64_t add_count, uint64_t add_extra);

/* diagnostics: blocks sketched with a speculative threshold, and how many of them needed the second pass */
int fh_debug_speculation(fh_sketcher *s, uint64_t *first_pass, uint64_t *second_pass);

/* diagnostics of the single-synchronisation path of small sketches (kmers_to_sketch <= 3000): speculative ranges whose
 * verdict was taken on the device and read later, how many of those had to be finished step by step after all, and
 * fh_finish calls served by the one fused launch */
int fh_debug_fast_path(fh_sketcher *s, uint64_t *deferred, uint64_t *recovered, uint64_t *fused_finishes);

/* streaming-read bandwidth of this box's HBM over [dev_bytes, dev_bytes+bytes) (16 B/lane loads, best of `reps`):
 * the measured counterpart of the 8 TB/s spec peak that bench.py prints next to the roofline */
int fh_measure_read_bandwidth(int device, const void *dev_bytes, uint64_t bytes, int reps, double *gb_per_s);

/* --- device memory helpers so callers need no HIP/torch binding (tests, bench) --- */
int fh_device_alloc(int device, uint64_t bytes, void **out);
int fh_device_free(int device, void *p);
int fh_copy_to_device(int device, void *dst, const void *src, uint64_t bytes);
int fh_copy_from_device(int device, void *dst, const void *src, uint64_t bytes);

/* --- synthetic inputs: counter-based, identical on host and device --- */
/* genome: `len` uniform ACGT bytes */
int fh_synth_genome_host(uint8_t *out, uint64_t len, uint64_t seed);
int fh_synth_genome_device(int device, void *dev_out, uint64_t len, uint64_t seed);
/* reads [first_read, first_read+n_reads): each read_len bases sampled from the genome (random start and
 * strand, per-base substitution and N rates as parts-per-million) followed by one '\0' breaker byte.
 * out receives n_reads*(read_len+1) bytes. */
int fh_synth_reads_host(uint8_t *out, const uint8_t *genome, uint64_t genome_len, uint64_t first_read,
                        uint64_t n_reads, uint32_t read_len, uint64_t seed, uint32_t sub_ppm, uint32_t n_ppm);
int fh_synth_reads_device(int device, void *dev_out, const void *dev_genome, uint64_t genome_len,
                          uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t seed,
                          uint32_t sub_ppm, uint32_t n_ppm);

#ifdef __cplusplus
}
#endif
#endif /* FINCH_HIP_H */
