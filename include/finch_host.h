/*
 * finch_host.h -- host-side mirror of finch's library entry points for the accelerated path, exported
 * by libfinch_hip.so next to the device ABI (finch_hip.h).  C++ implementation (the reference is
 * compiled code; no Rust toolchain in this image), C linkage so that any host language can bind it.
 *
 * Reference items mirrored (relative to the finch-rs tree):
 *   finch_sketch_files      finch::sketch_files        lib/src/lib.rs:29-49   (one Sketch per file, input order;
 *                                                       rayon par_iter over files -> worker threads, each with its
 *                                                       own device sketcher; files are mapped round-robin to GPUs)
 *   finch_sketch_buffer     finch::sketch_stream       lib/src/lib.rs:51-94   (in-memory FASTA/FASTQ[.gz] image)
 *   FASTX reading           needletail 0.5.0 parse_fastx_reader (lib.rs:60-68): gz / bz2 / xz sniffed by magic bytes,
 *                           '>' FASTA (multi-line), '@' FASTQ (4-line records)
 *   filtering               FilterParams::filter_counts lib/src/filtering.rs:60-87 (strand -> err -> abundance)
 *   post filter             SketchParams::process_post_filter lib/src/sketch_schemes/mod.rs:115-128
 *   .sk writer              MultiSketch / JsonSketch   lib/src/serialization/json.rs:64-89,141-158,199-218
 *
 * The per-base work (normalize, canonical k-mers, murmur3, bottom-n) is done by the device engine; this
 * layer parses, stages, filters (O(n) on <= n records) and serialises.
 */
#ifndef FINCH_HOST_H
#define FINCH_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SketchParams (mod.rs:54-71); AllCounts is not on the accelerated path */
typedef struct finch_sketch_params {
    uint32_t kind;            /* 0 = Mash, 1 = Scaled (2 = AllCounts: serialisation only) */
    uint32_t kmer_length;
    uint64_t kmers_to_sketch;
    uint64_t final_size;      /* Mash only */
    uint32_t no_strict;       /* Mash only */
    uint32_t pad;
    uint64_t hash_seed;
    double scale;             /* Scaled only */
} finch_sketch_params;

/* FilterParams (filtering.rs:11-16) */
typedef struct finch_filter_params {
    int32_t filter_on;        /* -1 = None, 0 = Some(false), 1 = Some(true) */
    uint32_t has_abun_lo, abun_lo;
    uint32_t has_abun_hi, abun_hi;
    uint32_t pad;
    double err_filter;
    double strand_filter;
} finch_filter_params;

/* Vec<Sketch> (serialization/mod.rs:46-55) */
typedef struct finch_sketches finch_sketches;

const char *finch_last_error(void);

/* SketchParams::default() (mod.rs:73-83) and FilterParams::default() (filtering.rs:136-145) */
void finch_default_sketch_params(finch_sketch_params *out);
void finch_default_filter_params(finch_filter_params *out);

/* sketch_files: `devices` lists the HIP devices to use (NULL/0 = device 0); n_threads = worker threads
 * (0 = one per hardware thread the process may use, at least 4 and at most 16 per device).  "-" reads stdin.  Returns 0 or a negative FH_ERR_* code (message via
 * finch_last_error; the first failing file wins, as in the reference's collect()). */
int finch_sketch_files(const char *const *filenames, uint32_t n_files, const finch_sketch_params *sketch_params,
                       const finch_filter_params *filters, const int *devices, uint32_t n_devices, uint32_t n_threads,
                       finch_sketches **out);
/* sketch_stream over an in-memory file image */
int finch_sketch_buffer(const uint8_t *data, uint64_t len, const char *name, const finch_sketch_params *sketch_params,
                        const finch_filter_params *filters, int device, finch_sketches **out);
/* ONE input across several devices (north_star; beyond the reference, which parallelises over files only,
 * lib.rs:34-36): a reader cuts the decompressed FASTQ / FASTA text into record- (FASTQ) or line-aligned (FASTA) chunks of
 * about `chunk_bytes` (0 = 32 MiB) and deals them round-robin to one sketcher per entry of `devices` (an entry may
 * repeat: several handles on one GPU); every handle splits and sketches its chunks on its device at the chunks' own
 * stream offsets, FASTA records that span a cut hand their last k-1 bases over as a halo; the partial sketches are
 * merged on the host (fh_merge), then filters / post filter as in sketch_stream.  The result is the Sketch
 * finch_sketch_files returns for the same file.  FASTQ the device-side splitter refuses (not plain 4-line FASTQ: blank
 * lines between records, a record longer than a chunk, ...) is read again through ONE handle and the host parser, which is
 * the judge of what needletail accepts (stdin, which cannot be read twice: FH_ERR_INVALID). */
int finch_sketch_file_sharded(const char *filename, const finch_sketch_params *sketch_params, const finch_filter_params *filters,
                              const int *devices, uint32_t n_devices, uint64_t chunk_bytes, finch_sketches **out);
int finch_sketch_buffer_sharded(const uint8_t *data, uint64_t len, const char *name, const finch_sketch_params *sketch_params,
                                const finch_filter_params *filters, const int *devices, uint32_t n_devices, uint64_t chunk_bytes,
                                finch_sketches **out);
/* The tail of sketch_stream (lib.rs:70-93) for a caller that fed a sketcher of include/finch_hip.h itself: fh_finish, to_vec,
 * filter_counts, process_post_filter -> one Sketch.  format: 1 FASTA, 2 FASTQ (the filtering default, lib.rs:70-76). */
struct fh_sketcher;
int finch_sketch_from_sketcher(struct fh_sketcher *h, const char *name, uint64_t seq_length, int format,
                               const finch_sketch_params *sketch_params, const finch_filter_params *filters, finch_sketches **out);
void finch_sketches_free(finch_sketches *s);

uint32_t finch_sketches_len(const finch_sketches *s);
const char *finch_sketch_name(const finch_sketches *s, uint32_t i);
uint64_t finch_sketch_seq_length(const finch_sketches *s, uint32_t i);
uint64_t finch_sketch_num_valid_kmers(const finch_sketches *s, uint32_t i);
uint64_t finch_sketch_n_hashes(const finch_sketches *s, uint32_t i);
/* the (possibly updated) filter params of sketch i (lib.rs:70-76, filtering.rs:69-80) */
int finch_sketch_filter_params(const finch_sketches *s, uint32_t i, finch_filter_params *out);
/* hashes ascending; kmers = n*k ASCII bytes; any pointer may be NULL */
int finch_sketch_copy(const finch_sketches *s, uint32_t i, uint64_t *hashes, uint32_t *counts, uint32_t *extra_counts,
                      uint8_t *kmers);

/* MultiSketch::from_sketches + serde_json::to_string (the `.sk` format).  *out is malloc'ed; free with
 * finch_free_string. */
int finch_sketches_to_json(const finch_sketches *s, char **out, uint64_t *len);
void finch_free_string(char *p);

/* ---- the other sketch file formats (lib/src/serialization/) ----
 * .bsk = write_finch_file / read_finch_file (mod.rs:123-222, schema finch.capnp), .msh = write_mash_file / read_mash_file
 * (mash.rs:12-135, schema mash.capnp): Cap'n Proto messages in the standard unpacked stream framing, as the reference's
 * capnp::serialize::write_message emits them.  The writers produce single-segment messages (at most 4 GiB; beyond that
 * FH_ERR_UNSUPPORTED); the readers take any valid message, the reference's multi-segment files included.  *out is
 * malloc'ed: finch_free_bytes. */
int finch_sketches_to_bsk(const finch_sketches *s, uint8_t **out, uint64_t *len);
int finch_sketches_to_msh(const finch_sketches *s, uint8_t **out, uint64_t *len);
void finch_free_bytes(uint8_t *p);
int finch_sketches_from_bsk(const uint8_t *data, uint64_t len, finch_sketches **out);
int finch_sketches_from_msh(const uint8_t *data, uint64_t len, finch_sketches **out);
/* MultiSketch::to_sketches over serde_json::from_slice (json.rs:92-139, 160-262; filtering.rs:110-134): the `.sk` reader */
int finch_sketches_from_json(const uint8_t *data, uint64_t len, finch_sketches **out);
/* open_sketch_file (lib.rs:96-118): *.msh / *.bsk / *.sk, *.json by file name */
int finch_open_sketch_file(const char *path, finch_sketches **out);
/* the `sketch` subcommand's output step (cli/src/main.rs:53-70): format by file name */
int finch_write_sketch_file(const finch_sketches *s, const char *path);
/* SketchParams of sketch i; kind 2 = AllCounts (only ever seen in files that were read: it is not on the accelerated path) */
int finch_sketch_params_of(const finch_sketches *s, uint32_t i, finch_sketch_params *out);
const char *finch_sketch_comment(const finch_sketches *s, uint32_t i);
int finch_sketch_set_comment(finch_sketches *s, uint32_t i, const char *comment);
/* dst.extend(src): the CLI collects the sketches of all its inputs before it writes one file (main.rs:60-70) */
int finch_sketches_append(finch_sketches *dst, const finch_sketches *src);
/* FilterParams::filter_sketch (filtering.rs:20-54) exactly as the reference has it: sketch i's filter parameters take
 * the stricter of their own and `filters`' values; its hashes are left alone (the reference drops the filtered list). */
int finch_filter_sketch(finch_sketches *s, uint32_t i, const finch_filter_params *filters);

/* distance (lib/src/distance.rs:9-47): compares sketch ia of `a` (query) with sketch ib of `b` (reference).
 * raw_distance (distance.rs:66-126) unless old_mode (old_distance, distance.rs:136-157). */
typedef struct finch_distance_out {
    double containment, jaccard, mash_distance;
    uint64_t common_hashes, total_hashes;
} finch_distance_out;
int finch_distance(const finch_sketches *a, uint32_t ia, const finch_sketches *b, uint32_t ib, int old_mode,
                   finch_distance_out *out);
/* raw_distance on bare ascending hash arrays */
int finch_raw_distance(const uint64_t *query, uint64_t nq, const uint64_t *ref, uint64_t nr, double scale,
                       finch_distance_out *out);

/* ---- pieces that need no GPU (unit-testable on the host) ---- */
/* Build a one-sketch result from arrays (to exercise filtering / serialisation without a device).  FH_ERR_INVALID for
 * records no sketcher can emit: count == 0 or extra_count > count (mash.rs:45-56). */
int finch_sketches_from_arrays(const char *name, uint64_t seq_length, uint64_t num_valid_kmers, uint64_t n,
                               const uint64_t *hashes, const uint32_t *counts, const uint32_t *extra_counts,
                               const uint8_t *kmers, const finch_sketch_params *sketch_params,
                               const finch_filter_params *filters, finch_sketches **out);
/* filter_counts (filtering.rs:60-87) + process_post_filter (mod.rs:115-128) applied in place to sketch i;
 * `filters` is updated exactly as the reference updates its FilterParams. */
int finch_apply_filters(finch_sketches *s, uint32_t i, finch_filter_params *filters);
/* statistics.rs: cardinality (8-23: k-minimum-values estimate of the number of distinct k-mers, the reference's f32
 * arithmetic) and hist (30-47: out[c - 1] = hashes with count c; out == NULL only reports *n = the largest count) */
int finch_sketch_cardinality(const finch_sketches *s, uint32_t i, uint64_t *out);
int finch_sketch_hist(const finch_sketches *s, uint32_t i, uint64_t *out, uint64_t cap, uint64_t *n);
/* guess_filter_threshold (filtering.rs:154-195); counts must be >= 1.  Returns 0 (never a valid threshold) and sets
 * finch_last_error for a null array or a zero count. */
uint32_t finch_guess_filter_threshold(const uint32_t *counts, uint64_t n, double filter_level);
/* FASTX scan only: number of records, sum of sequence() lengths (what total_bases counts, mash.rs:72) and
 * the format of the first record (1 FASTA, 2 FASTQ).  gz images are inflated first. */
int finch_fastx_scan(const uint8_t *data, uint64_t len, uint64_t *n_records, uint64_t *total_bases, int *format);

/* Record count / total_bases exactly as the device-side FASTA path keeps them (the host reads raw text into the staging
 * buffer in chunks cut after a newline and only locates the header lines); test hook: must agree with finch_fastx_scan. */
int finch_fasta_count_chunked(const uint8_t *data, uint64_t len, uint64_t chunk, uint64_t *n_records, uint64_t *total_bases);

/* Test hook: read `path` the way the text paths read plain files (requests of `chunk` bytes; requests of >= 16 MiB on
 * a regular file are split over `read_threads` threads) into dst[0, cap); *got = bytes delivered. */
int finch_read_file_probe(const char *path, uint64_t chunk, uint32_t read_threads, uint8_t *dst, uint64_t cap, uint64_t *got);

/* Test hook: the byte stream the parsers see for an input image (magic-byte sniffing, gzip / BGZF / bzip2 / xz
 * decompression), read in requests of `chunk` bytes into dst[0, cap); *got = bytes delivered. */
int finch_source_probe(const uint8_t *data, uint64_t len, uint64_t chunk, uint8_t *dst, uint64_t cap, uint64_t *got);

/* Test hook (no device): the chunks finch_sketch_*_sharded's reader deals out for an input image.  Per chunk: meta[4i..] =
 * (offset in the decompressed text, length, FASTA start state, halo length), halos[64i..] = the halo bytes.  FASTA only:
 * n_records / total_bases as the reader counts them. */
int finch_shard_probe(const uint8_t *data, uint64_t len, uint32_t k, uint64_t chunk_bytes, uint64_t max_chunks, uint64_t *meta,
                      uint8_t *halos, uint64_t *n_chunks, uint64_t *n_records, uint64_t *total_bases);

/* Test hook (no device): the batches finch_sketch_files' reader hands to fh_push_bgzf_fastq for a BGZF image -- member
 * tables and bytes in a buffer of buf_bytes, at most max_members / text_budget bytes of text per batch --, inflated on
 * the host exactly where the tables say; text_out receives the whole text, *first_byte the reader's probe of it. */
int finch_bgzf_batch_probe(const uint8_t *data, uint64_t len, uint64_t buf_bytes, uint32_t max_members, uint64_t text_budget,
                           uint8_t *text_out, uint64_t text_cap, uint64_t *text_len, uint64_t *n_batches, int *first_byte);

/* Test hook (no device): what finch_sketch_files' reader hands to fh_push_gzip_fastq for a plain gzip image -- the probe of
 * the file's first member (*first_byte: the first byte of its text, -1 if the member is BGZF, no gzip member, or nothing can
 * be decoded from its first 64 KiB; *hdr_len: the length of its RFC 1952 header) and the bytes behind the header as the
 * reader takes them, piece_bytes at a time (*deflate_bytes of them in all, *crc_of_pieces their running CRC-32). */
int finch_gzip_probe(const uint8_t *data, uint64_t len, uint64_t piece_bytes, uint64_t *hdr_len, int *first_byte, uint64_t *deflate_bytes,
                     uint32_t *crc_of_pieces);

/* Test hook: inputs this process has sketched with the BGZF inflate on the device (finch_sketch_files /
 * finch_sketch_buffer: bgzip'd FASTQ unless option device_inflate is 0), and how many of them it had to read again through the
 * host-side inflate because the device pass refused them. */
void finch_debug_device_inflate(uint64_t *files_on_device, uint64_t *files_reread);
/* the same for plain gzip files (fh_push_gzip_fastq) */
void finch_debug_device_gzip(uint64_t *files_on_device, uint64_t *files_reread);
/* Measurement hook (bench.py --workload c5): reports the sketch kernel's own time (HIP events on every worker's stream), its
 * launches and the k-mer start positions they covered over the inputs sketched since it was switched on, then: enable = 1
 * switches it on and zeroes the sums, 0 switches it off, -1 leaves it as it is. */
void finch_debug_kernel_times(int enable, double *kernel_ms, uint64_t *launches, uint64_t *positions);
/* files of this process's finch_sketch_files calls that were sketched many-per-launch (fh_batch_*, include/finch_hip.h) and
 * files the batch path handed to a sketcher of their own instead (not taken: too few distinct k-mers below the batch's
 * threshold, ...; files that never qualified -- FASTQ, compressed, stdin, huge -- count in neither) */
void finch_debug_file_batch(uint64_t *taken, uint64_t *not_taken);
/* inputs of this process whose FASTQ text was stripped to the packed sequence stream on the host (fh_fqstrip.h: text in host
 * memory and >= 8 read threads -- headers, '+' lines and quality strings never cross the PCIe link) */
uint64_t finch_debug_fastq_host_strip(void);
/* test hook: text[0, len) -- whole records of plain 4-line FASTQ -- through that strip on `threads` threads: out (cap >= len / 2 + 64)
 * receives the packed stream (each record's sequence, blanks dropped, one 0 byte behind it); FH_ERR_INVALID if the text is not
 * plain 4-line FASTQ (what the caller then hands to the parser that is the judge of it) */
int finch_fastq_strip_probe(const uint8_t *text, uint64_t len, uint32_t threads, uint8_t *out, uint64_t cap, uint64_t *packed,
                            uint64_t *n_records, uint64_t *total_bases);

/* test hook: FASTA text (text[0] == '>') through the walk a worker of finch_sketch_files stages a genome with -- read in pieces,
 * line ends stripped, records closed by a breaker, written in the batch sketcher's two-bit form (finch_hip.h
 * fh_batch_submit_packed) -- `piece` bytes at a time; region: cap >= fh_batch_packed_bytes(len) */
int finch_fasta_two_bit_probe(const uint8_t *text, uint64_t len, uint64_t piece, uint8_t *region, uint64_t cap, uint64_t *positions,
                              uint64_t *n_records, uint64_t *total_bases);

#ifdef __cplusplus
}
#endif
#endif /* FINCH_HOST_H */
