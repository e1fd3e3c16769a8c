// fh_api.hip -- host side of the C ABI declared in include/finch_hip.h.
//
// One handle = one device + one HIP stream + the device-resident sketch state (fh_device.h).
// The host only stages bytes, sizes launches and copies the final <= n records back; every
// per-base operation of the reference's process()/push() (mash.rs:34-80) runs in fh_kernels.hip.
// There is deliberately no CPU implementation of the sketching path in this library.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/finch_hip.h"
#include "fh_core.h"
#include "fh_strip.h"
#include "fh_device.h"
#include "fh_kernels.h"
#include "fh_internal.h"
#include "fh_options.h"

using namespace fh;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(FH_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// Device / pinned allocations made while a handle is in use.  fh_free parks reset handles (see "handle cache" below) and
// those keep their device memory; before an allocation is allowed to fail for lack of memory the parked handles are
// given back and it is tried once more.
template <class T>
hipError_t dev_malloc(T **p, size_t bytes) {
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        int dev = 0;
        (void)hipGetDevice(&dev);
        fh_release_cached();
        (void)hipSetDevice(dev);
        e = hipMalloc((void **)p, bytes);
    }
    return e;
}
template <class T>
hipError_t host_malloc(T **p, size_t bytes) {
    hipError_t e = hipHostMalloc((void **)p, bytes, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        int dev = 0;
        (void)hipGetDevice(&dev);
        fh_release_cached();
        (void)hipSetDevice(dev);
        e = hipHostMalloc((void **)p, bytes, hipHostMallocDefault);
    }
    return e;
}

constexpr uint64_t DEFAULT_MAX_LAUNCH = 256ull * 32 * TILE_POS; // k-mer start positions in flight (upper bound on the waves)
constexpr uint64_t FIRST_LAUNCH = 4096;
constexpr uint64_t SMALL_N_MAX = 3000; // largest kmers_to_sketch served by the in-LDS selection alone
constexpr uint32_t CLOG_CAP = 65536;
const uint64_t STAGE_BYTES_ENV = [] {
    const char *e = cfg("stage_bytes"); // test knob: force blocks to span staging slices
    const uint64_t v = e ? strtoull(e, nullptr, 10) : 0;
    return v >= 4096 ? v : 0ull;
}();
constexpr uint64_t STAGE_BYTES_DEFAULT = 64ull << 20;
constexpr uint64_t STAGE_HEADROOM = 64; // bytes in front of a staging buffer's data: fh_push_staged puts the K-1 <= 63 carried bytes there
constexpr int N_STAGE = 2;

struct ResultRec {
    uint64_t hash;
    uint32_t count, extra;
    uint64_t kmer; // m-form (K > 32: the last 32 bases)
    uint64_t pos;
    uint64_t kmer_hi = 0; // K > 32: the first K - 32 bases
};

} // namespace

struct fh_sketcher {
    fh_params p{};
    int device = 0;
    uint64_t max_hash = 0; // scaled
    uint64_t max_launch = 0;
    hipStream_t stream = nullptr;

    // device state
    Entry *table = nullptr;
    uint32_t cap = 0;
    uint32_t *live = nullptr;
    uint32_t live_cap = 0;
    uint32_t *shard_cnt = nullptr, *shard_buf = nullptr; // per-shard append lists of new inserts
    uint32_t shard_cap = 0;
    uint32_t *dead = nullptr; // slots dropped from the live list (garbage to clear on reset)
    uint32_t dead_cap = 0;
    Ctl *ctl = nullptr;
    CollRec *clog = nullptr;
    // gather outputs (device), capacity out_cap (grown on demand)
    uint64_t *o_hash = nullptr, *o_kmer = nullptr, *o_pos = nullptr;
    uint64_t *o_kmer_hi = nullptr; // K > 32 only
    uint64_t *kmer_hi = nullptr;   // K > 32 only: high words of the table's k-mers, one per slot (fh_device.h)
    uint32_t *o_count = nullptr, *o_extra = nullptr;
    uint32_t out_cap = 0;
    // (the six arrays are slices of ONE allocation, out_stride entries apart: a small sketch goes to the host in one copy)
    void *o_block = nullptr;
    size_t out_stride = 0, o_block_bytes = 0;
    // A large Mash sketch's wide columns (hash, k-mer, first position: 24 B per record) stay on the device after fh_finish
    // until somebody asks for them: the filters of configs[2] look at the counts of all 2 M records and then want 10 000 rows.
    bool wide_pending = false;
    uint32_t *d_rows = nullptr;   // fh_copy_out_rows by device-side gather: row indices, gathered words
    uint64_t *d_rows_out = nullptr, *h_rows_out = nullptr;
    uint32_t *h_rows = nullptr;
    size_t rows_cap = 0;
    // device-wide selection (fh_big.hip), allocated on first use
    bool big_mode = false;        // live sets beyond the in-LDS sort (large kmers_to_sketch, scaled)
    uint64_t live_target = 0;     // prune when the live list reaches this
    uint64_t *keys_a = nullptr, *keys_b = nullptr;
    uint32_t *slots_a = nullptr, *slots_b = nullptr, *keep_dev = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint32_t big_cap = 0;
    uint64_t n_big_prunes = 0;
    // the range of k-mer start positions currently on the device (one at a time); completion is checked
    // lazily (drain) so that pushes stay asynchronous
    struct Pending {
        bool active = false;
        const uint8_t *seq = nullptr;
        uint64_t len = 0, base_pos = 0, p_begin = 0, p_end = 0;
        uint32_t tiles_total = 0, n_units = 0, n_left_in = 0;
        uint32_t unit_tiles = UNIT_TILES; // queue granularity of this range (1 for inputs that would not fill the chip with 2)
        uint32_t first_units = 0, grid_waves = 0; // the range's first launch: units every wave starts on unasked, and its waves
        uint32_t seg = 0; // != 0: the range runs through the segment kernel with this stride (tiles of 64 / seg_sub x seg positions)
        uint32_t seg_sub = 1; // lanes per record there (fh_device.h, seg_sub_for)
        uint32_t max_units = MAX_UNITS; // units a pull takes at most
        int left_cur = 0;
        double admit_at_start = 1.0; // admit rate the range started with (for the novelty estimate)
        bool gated = false;   // queued behind an unverified speculation: its launches run only if Ctl::spec_ok
        bool verdict = false; // the speculative range itself: its epilogue sets Ctl::spec_ok
    } pend;
    // Small sketches (kmers_to_sketch <= 3000): the launches that follow a sketch launch are one fused kernel
    // (k_small_epilogue), the threshold is refreshed inside the launch (Ctl::hist), and a speculative first range is NOT
    // waited for -- its verdict is taken on the device, whatever is queued behind it is gated on that verdict, and the host
    // looks at the outcome at its next synchronisation (the next push, or fh_finish): one round trip per file / per pass
    // instead of five.  FH_NO_FAST=1 keeps the step-by-step path (A/B, tests); FH_NO_HIST=1 only the refresh off.
    bool fast = false, hist = false;
    struct Spec {
        bool pending = false; // a speculative range whose verdict the host has not read yet
        Pending range;        // that range (to finish it the slow way if the speculation failed)
        uint64_t tau = 0;     // the guessed threshold
        uint64_t n_pos = 0;   // positions it covered (counted into positions_done in advance)
    } spec;
    uint64_t n_fast_finish = 0, n_spec_deferred = 0, n_spec_recovered = 0;
    // the epilogue of the last sketch launch, not launched yet: whatever touches the control block next launches it first
    // (flush_epilogue) -- and fh_finish folds it into its own, so a file of a batch gets ONE epilogue launch, not two
    uint32_t epi_pending = 0, epi_units = 0;
    // fh_finish's fused epilogue left the device side as fh_reset would (EPI_RESET): the next fh_reset is host bookkeeping only
    // fh_process: the staging buffer records are being written to (null: none taken yet), how much of it is filled, whether
    // the block continues a record an earlier commit cut, whether a record is open, and mash.rs:72's total_bases
    uint8_t *proc_buf = nullptr;
    uint64_t proc_cap = 0, proc_fill = 0, proc_total_bases = 0;
    bool proc_continuing = false, proc_in_record = false, proc_in_cut = false;
    bool device_clean = false;
    uint64_t final_text_bases = 0; // Ctl::text_bases as of fh_finish (fh_text_bases stays valid after it)
    uint32_t *left_buf[2] = {nullptr, nullptr}; // leftover tile ranges of a stopped launch (pairs; two per wave: fh_k2s.hip)
    // The segment kernel (fh_k2s.hip): seg_hint is what the caller said about the records (fh_set_record_stride: 0 = find out,
    // 1 = do not use it, else the stride), blk_seg the stride of the block being sketched (0: k2_sketch), gran what the block's
    // ranges are cut at (a tile of the kernel that runs it)
    uint32_t seg_hint = 0, blk_seg = 0, blk_sub = 1; // (blk_sub: lanes per record, or SEG_RAGGED)
    uint64_t gran = TILE_POS;
    uint32_t *h_probe = nullptr; // pinned: launch_seg_probe's answer
    bool probe_seen = false;     // a block of this handle has been asked (probe_answer is its answer, or a later block's)
    hipEvent_t probe_ev = nullptr; // recorded behind the newest probe: h_probe[0] is read only once it has completed
    uint32_t probe_answer = 0;     // the newest answer read that way
    uint32_t probe_breakers = 0;   // ... and how many of the 4096 bytes it looked at all over the block were no bases
    uint64_t n_seg_launches = 0, n_seg_probes = 0;
    uint64_t max_waves = 0;
    uint64_t max_range = 0; // test knob: cap on positions per range
    uint64_t tau_lo = 0;    // != 0 while a block is re-read for the hashes above a speculative threshold
    bool no_spec = false;   // test knob: disable the speculative first pass
    uint64_t n_spec = 0, n_spec_fallback = 0, n_spec_rescaled = 0;
    // sampling pre-pass of large sketches (fh_kernels.hip, k_sample_hashes): buffers allocated on first use
    uint32_t *smp_list = nullptr, *smp_hist = nullptr, *h_smp_hist = nullptr; // tile runs; histograms (h_: pinned)
    uint64_t smp_list_cap = 0;
    uint64_t n_sampled = 0; // blocks whose threshold came from a sample
    // observed novelty: new hashes per position over the last completed range (admitted occurrences of hashes that are
    // already in the table cost time, but they do not fill it)
    uint64_t ins_seen = 0;
    double novelty = 1.0; // new hashes per ADMITTED occurrence, over the last completed range (1 = assume the worst)
    uint64_t n_launches = 0, n_relaunches = 0;
    // staging
    uint8_t *h_stage[N_STAGE] = {nullptr, nullptr};
    uint8_t *d_stage[N_STAGE] = {nullptr, nullptr};
    hipEvent_t stage_done[N_STAGE] = {nullptr, nullptr};
    // fh_text_prefetch: the copy stream (created on first use, by the reader's thread) and, per slot, the length whose
    // host-to-device copy is already under way (0 = none)
    hipStream_t copy_stream = nullptr;
    uint64_t stage_prefetched[N_STAGE] = {0, 0};
    bool stage_busy[N_STAGE] = {false, false};
    uint64_t stage_cap[N_STAGE] = {0, 0}; // bytes a slot can hold (<= stage_bytes; slots grow with the pushes they serve)
    int stage_next = 0;
    uint64_t stage_bytes = 0;
    // device-side FASTQ packing (fh_text.hip): packed output + block scan scratch per staging slot
    // fh_push_bgzf_fastq: the batch as it came (member table + DEFLATE bytes), and text / packed buffers of its own:
    // a wavefront per member only fills the chip with thousands of members in flight, i.e. hundreds of MB of text per batch
    uint8_t *d_comp = nullptr;
    BgzfMember *d_bz_members = nullptr;
    uint64_t bz_comp_cap = 0, bz_acc_bytes = 0, bz_acc_text = 0; // FH_BGZF_MORE: what has been collected for the next launch
    uint32_t bz_acc_n = 0;
    uint64_t bz_batch_left = 0; // bytes of the previous launch's text that lead this one's
    // every push of a batch inflates on a side stream of its own while the host reads the next members; the push
    // without FH_BGZF_MORE joins them
    static constexpr int BZ_STREAMS = 4;
    hipStream_t bz_stream[BZ_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t bz_done[BZ_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
    bool bz_used[BZ_STREAMS] = {false, false, false, false};
    int bz_q = 0;
    uint32_t *d_bz_status = nullptr, *h_bz_status = nullptr;
    uint64_t bz_text_cap = 0;
    uint8_t *bz_text[2] = {nullptr, nullptr}, *bz_packed[2] = {nullptr, nullptr};
    uint32_t *bz_blk_a[2] = {nullptr, nullptr}, *bz_blk_b[2] = {nullptr, nullptr}, *bz_lines = nullptr;
    uint32_t bz_line_cap = 0;
    int bz_next = 0;
    const uint8_t *bgzf_left_ptr = nullptr; // text behind the last whole record of the previous batch (in the other bz_text)
    uint64_t bgzf_left_len = 0;
    // fh_push_gzip_fastq (plain gzip: one DEFLATE stream cut into chunks, fh_bgzf.hip): symbol buffers and the pass over the
    // chain of chunks; the text, packed bytes and scan scratch are the BGZF path's
    uint16_t *gz_sym = nullptr;
    uint64_t gz_sym_elems = 0;
    GzChunk *gz_recs = nullptr;
    uint8_t *gz_win_in = nullptr, *gz_window = nullptr;
    uint32_t *gz_live = nullptr, *gz_tile_map = nullptr, *gz_crc_tmp = nullptr, *gz_summary = nullptr, *h_gz_summary = nullptr;
    uint32_t gz_chunks_cap = 0; // chunks there is room for
    uint64_t gz_chunk_alloc = 0; // the chunk size the buffers were sized for
    uint64_t gz_cap = 0, gz_acc = 0;              // symbol slots per chunk; bytes of the batch being collected (FH_GZ_MORE)
    uint16_t *gz_group_map = nullptr;
    uint32_t *gz_claims = nullptr;
    uint64_t *gz_times = nullptr; // (FH_GZ_TIMES: per-chunk timestamps of the decoding launch)
    GzFeed *h_gz_feed = nullptr; // (pinned) how much of the batch being collected has arrived: the decoding launch polls it
    bool gz_feeding = false;     // such a launch is out
    double gz_t0 = 0;            // when the batch being collected saw its first push
    bool gz_fed = false;         // the batch being collected is decoded by a launch that waits for its pieces
    bool gz_no_feed = false;     // such a launch has timed out on this handle: batches are decoded once they are complete
    uint8_t *gz_group_win = nullptr;
    uint64_t gz_base = 0;     // where a push's bytes land in d_comp: what the previous push left undecoded sits in front of them
    uint64_t gz_tail_len = 0; // ... that many bytes, decoding resumes at bit gz_bit of the first
    uint32_t gz_bit = 0, gz_valid = 0, gz_crc = 0;
    uint64_t gz_total = 0;
    bool gz_open = false;
    uint8_t *d_packed[N_STAGE] = {nullptr, nullptr};
    uint32_t *d_blk_a[N_STAGE] = {nullptr, nullptr}, *d_blk_b[N_STAGE] = {nullptr, nullptr};
    uint32_t *d_lines = nullptr; // fh_push_fastq_text: where every text line of the chunk ends (one u32 per line)
    uint32_t line_cap = 0;
    const uint8_t *dprev_ptr = nullptr; // fh_push_fasta_text: the previous chunk's packed range (device), for the
    uint64_t dprev_len = 0;             // K-1 bytes a k-mer may span across chunks
    uint32_t *d_text_tot = nullptr; // [0] newlines, [1] packed bytes, [2] error flag
    uint32_t *h_text_tot = nullptr; // pinned
    uint64_t text_bases = 0;
    uint8_t carry[64] = {0}; // last K-1 staged bytes: k-mers span staging slices (and FH_PUSH_CONTINUE pushes)
    uint32_t carry_len = 0;
    uint8_t halo[64] = {0};  // fh_set_text_halo: packed bytes that precede the next fh_push_fasta_text chunk
    uint32_t halo_len = 0;
    Ctl *h_ctl = nullptr; // pinned
    void *h_out = nullptr; // pinned staging of the finished sketch (fh_finish)
    size_t h_out_bytes = 0;

    // host bookkeeping
    uint64_t stream_off = 0;     // stream coordinate of the next byte
    uint64_t positions_done = 0; // k-mer start positions processed since reset
    uint32_t trigger = 0;
    bool open_loop = false;      // threshold tight enough that launches need no host feedback
    uint64_t last_tau = ~0ull;   // as of the last status readback
    uint32_t last_live = 0;
    bool finished = false;
    bool dirty = false; // device table holds entries from this run

    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    double prof_ms = 0.0;
    uint64_t prof_launches = 0, prof_positions = 0;

    // finished result (host), ascending by hash
    // fh_finish leaves the result as arrays in the pinned D2H buffer (r_* point into h_out); the record vector is
    // only built when a merge needs it (2 M records: 10 ms of host time that fh_copy_out does not need)
    uint64_t *r_hash = nullptr, *r_kmer = nullptr, *r_pos = nullptr, *r_kmer_hi = nullptr;
    uint32_t *r_count = nullptr, *r_extra = nullptr;
    size_t r_n = 0;
    bool res_built = false;
    std::vector<ResultRec> res;
    uint64_t total_kmers = 0;
};

namespace {

uint64_t scaled_max_hash(double scale) {
    // scaled.rs:23,31 : iscale = (1. / scale) as u64 (saturating cast) ; max_hash = u64::MAX / iscale
    double inv = 1.0 / scale;
    uint64_t iscale;
    if (!(inv == inv) || inv <= 0.0) iscale = 0;
    else if (inv >= 18446744073709551616.0) iscale = UINT64_MAX;
    else iscale = (uint64_t)inv;
    return iscale ? UINT64_MAX / iscale : UINT64_MAX;
}

uint64_t initial_tau(const fh_sketcher *s) {
    if (s->p.kind == FH_KIND_SCALED && s->p.size == 0) return s->max_hash;
    return EMPTY64;
}

int set_device(const fh_sketcher *s) {
    HIP_TRY(hipSetDevice(s->device));
    return FH_OK;
}

// the inflate launches of an abandoned batch may still be running
// a batch of gzip bytes abandoned half-way: its decoding launch is still waiting for the rest
static void gzip_quiesce(fh_sketcher *s) {
    if (s->gz_feeding && s->h_gz_feed) {
        __atomic_store_n(&s->h_gz_feed->abort, 1u, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(s->stream);
    }
    s->gz_feeding = false;
    s->gz_acc = 0;
}
static void bgzf_quiesce(fh_sketcher *s) {
    for (int q = 0; q < fh_sketcher::BZ_STREAMS; ++q)
        if (s->bz_stream[q] && s->bz_used[q]) {
            (void)hipStreamSynchronize(s->bz_stream[q]);
            s->bz_used[q] = false;
        }
    s->bz_acc_n = 0;
    s->bz_acc_bytes = s->bz_acc_text = 0;
}
// (device_part = false: the caller has queued a kernel that re-initialises the control block itself)
int init_state(fh_sketcher *s, bool device_part = true) {
    s->epi_pending = 0; // (callers that could have one pending -- fh_reset -- have launched it: it rewinds the shard cursors)
    s->proc_buf = nullptr;
    s->proc_fill = s->proc_total_bases = 0;
    s->proc_continuing = s->proc_in_record = s->proc_in_cut = false;
    if (device_part) HIP_TRY(launch_init_ctl(s->ctl, initial_tau(s), s->stream, false, s->p.size, 0ull, s->hist));
    s->spec.pending = false;
    s->stream_off = 0;
    s->ins_seen = 0;
    s->novelty = 1.0;
    s->positions_done = 0;
    s->open_loop = false;
    s->last_tau = initial_tau(s);
    s->last_live = 0;
    s->pend.active = false;
    s->tau_lo = 0;
    s->carry_len = 0;
    s->dprev_len = 0;
    s->bgzf_left_len = 0;
    s->gz_open = false;
    gzip_quiesce(s);
    bgzf_quiesce(s);
    for (int i = 0; i < N_STAGE; ++i) // a copy fh_text_prefetch started for a stream that was then abandoned
        if (s->stage_prefetched[i]) {
            (void)hipEventSynchronize(s->stage_done[i]);
            s->stage_prefetched[i] = 0;
        }
    s->halo_len = 0;
    s->live_target = s->big_mode ? std::max<uint64_t>(4 * s->p.size, 1ull << 16) : (uint64_t)SMALL_MAX;
    s->finished = false;
    s->dirty = false;
    s->res.clear();
    s->res_built = false;
    s->wide_pending = false;
    s->r_n = 0;
    s->total_kmers = 0;
    s->prof_used = 0;
    s->prof_ms = 0.0;
    s->prof_launches = 0;
    s->prof_positions = 0;
    return FH_OK;
}

double admit_rate(uint64_t tau) { return tau == EMPTY64 ? 1.0 : ((double)tau + 1.0) / 18446744073709551616.0; }
// expected new hashes per position: what fills the live set.  Occurrences of hashes already in the table are admitted
// too, but only cost time; on low-diversity input (small k, deep coverage) they are nearly all there is, and sizing
// ranges by the admit rate alone kept such streams in closed-loop mode for good (k = 11, 10 Gbase: 1400 ranges per pass).
// How the admit path of the next launch touches an entry (fh_k2.hip, upsert).  Reading it first turns the five
// atomics of an occurrence whose hash is already in the table into three loads and one or two adds; the device
// sustains ~25 G atomics/s but twice that in loads (tools/ubench_atomics.hip).  Measured per 10 Gbase pass
// (bench.py, 8 steps, three runs each): k = 31, n = 2 M 57.7 -> 54.3 ms; k = 21, n = 200 000 20.8 -> 20.2 ms; Scaled
// 0.008 24.5 -> 24.2 ms; nothing to gain or lose at n = 1000.  The exception is a stream whose admitted occurrences
// all land on a handful of hot entries (k = 8: 1000 entries take 3 % of all positions): there the loads queue behind
// the atomics on the same lines (103 -> 112 ms) and the launch keeps the plain form; the observed novelty tells.
uint32_t read_first_of(const fh_sketcher *s) {
    if (const char *e = cfg("read_first")) return atoi(e) ? 1u : 0u; // A/B and tests: 0 = never, 1 = always
    return s->novelty >= 0.02 ? 1u : 0u;
}

double fill_rate(const fh_sketcher *s) { return admit_rate(s->last_tau) * std::min(1.0, 2.0 * s->novelty); }

// Warm-up is closed-loop: while the admit threshold is loose, a launch may insert up to one new hash
// per position, so its size is chosen from the threshold read back after the previous launch such that
// the live set stays inside what the in-LDS prune can sort.  tau only ever decreases, so once a
// maximum-size launch is safe it stays safe and launches go open-loop (no host feedback).
// each shard serves ceil(waves / N_SHARDS) waves, each of which inserts at most WAVE_BUDGET + WAVE_OVERSHOOT new
// hashes per launch, plus its share of the room below the soft limit
uint32_t shard_cap_for(const fh_sketcher *s, uint64_t live_target) {
    const uint64_t waves_per_shard = (s->max_waves + N_SHARDS - 1) / N_SHARDS;
    return (uint32_t)(waves_per_shard * (WAVE_BUDGET + WAVE_OVERSHOOT) + live_target / N_SHARDS + 1024);
}

int alloc_shards(fh_sketcher *s, uint64_t live_target) {
    const uint32_t cap = shard_cap_for(s, live_target);
    if (s->shard_buf && cap <= s->shard_cap) return FH_OK;
    if (s->shard_buf) (void)hipFree(s->shard_buf);
    s->shard_buf = nullptr;
    if (!s->shard_cnt) HIP_TRY(dev_malloc(&s->shard_cnt, (size_t)N_SHARDS * SHARD_STRIDE * sizeof(uint32_t)));
    HIP_TRY(dev_malloc(&s->shard_buf, (size_t)N_SHARDS * cap * sizeof(uint32_t)));
    s->shard_cap = cap;
    return FH_OK;
}

uint32_t soft_limit_of(const fh_sketcher *s) {
    // stop pulling work when the live set reaches this; the table holds this + everything in flight
    if (s->big_mode) return (uint32_t)std::min<uint64_t>(s->live_target, 0xFFFFFFF0ull);
    return (uint32_t)(SMALL_MAX - 1024);
}

// Warm-up is closed-loop: while the admit threshold is loose, a range may insert up to one new hash
// per position, so its size is chosen from the threshold read back after the previous range such that
// the live set stays small.  Once (positions in flight x admit rate) is small, one launch takes the
// whole remaining input: waves stop by themselves if the table ever nears its guarded size.
uint64_t next_range_size(const fh_sketcher *s, uint64_t remaining) {
    uint64_t P;
    const uint64_t G = s->gran; // ranges are cut at whole tiles of the kernel that runs the block (TILE_POS, or 64 segments)
    if (s->open_loop) {
        P = ((remaining + G - 1) / G) * G; // everything, incl. the last partial tile
    } else {
        const double room = (double)s->live_target - (double)std::min<uint64_t>(s->last_live, s->live_target);
        // small mode keeps half the room in reserve (the live set has to fit the in-LDS prune); in big mode the
        // soft-limit stop makes overshoot harmless, and halving the room every range cost 5-11 launches per prune
        const double fr = fill_rate(s);
        double p = fr > 0.0 ? (s->big_mode ? 1.0 : 0.5) * room / fr : 1e19;
        P = p >= 1e18 ? remaining + G : (uint64_t)p; // (rounded down below; capped at the rounded-up remainder)
    }
    if (s->max_range) P = std::min<uint64_t>(P, s->max_range);
    P = std::max<uint64_t>((P / G) * G, G);
    return std::min<uint64_t>(P, ((remaining + G - 1) / G) * G);
}

int check_ctl(fh_sketcher *s);
int flush_epilogue(fh_sketcher *s);
int recover_spec(fh_sketcher *s);
int reread_above(fh_sketcher *s, const uint8_t *seq, uint64_t len, uint64_t base_pos, uint64_t p_begin, uint64_t p_end, uint64_t lo);
int sketch_positions(fh_sketcher *s, const uint8_t *d_seq, uint64_t len, uint64_t base_pos, uint64_t pos, uint64_t n_pos, uint64_t lo_end);
int big_prune(fh_sketcher *s, bool sorted = true);
int grow_table(fh_sketcher *s, uint64_t new_live_cap);

int collect_profile(fh_sketcher *s) {
    for (size_t i = 0; i < s->prof_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->prof_events[i].first, s->prof_events[i].second));
        s->prof_ms += ms;
    }
    s->prof_used = 0;
    return FH_OK;
}

EpiArgs epi_args(const fh_sketcher *s, uint32_t flags) {
    EpiArgs e{};
    e.table = s->table;
    e.live = s->live;
    e.dead = s->dead;
    e.dead_cap = s->dead_cap;
    e.ctl = s->ctl;
    e.kind = s->p.kind;
    e.size = s->p.size;
    e.max_hash = s->max_hash;
    e.trigger = s->trigger;
    e.flags = flags;
    e.wide = s->p.k > 32 ? 1u : 0u;
    return e;
}

int flush_epilogue(fh_sketcher *s) {
    if (!s->epi_pending) return FH_OK;
    EpiArgs e = epi_args(s, s->epi_pending);
    e.n_units = s->epi_units;
    s->epi_pending = 0;
    HIP_TRY(launch_small_epilogue(e, s->stream));
    return FH_OK;
}

// one launch of the persistent sketch kernel over the pending range's queue (+ the in-stream prune)
int launch_pending(fh_sketcher *s) {
    fh_sketcher::Pending &r = s->pend;
    SketchArgs a{};
    a.seq = r.seq;
    a.len_total = r.len;
    a.p_begin = r.p_begin;
    a.p_end = r.p_end;
    a.base_pos = r.base_pos;
    a.seed = s->p.seed;
    a.hash_mask = s->p.hash_mask ? s->p.hash_mask : ~0ull;
    a.tau_lo = s->tau_lo;
    a.ctl = s->ctl;
    a.tiles_total = r.tiles_total;
    a.n_units = r.n_units;
    a.n_left_in = r.n_left_in;
    a.gate = r.gated ? 1u : 0u;
    a.unit_tiles = r.unit_tiles;
    a.first_units = r.first_units; // (start_range's launch only: a relaunch takes what is left through the queue)
    a.static_only = r.first_units && (uint64_t)r.first_units * r.grid_waves >= r.n_units ? 1u : 0u;
    a.seg_stride = r.seg;
    a.seg_sub = r.seg_sub;
    a.max_units = r.max_units;
    r.first_units = 0;
    a.left_in = s->left_buf[r.left_cur];
    a.left_out = s->left_buf[r.left_cur ^ 1];
    const uint64_t work_units = (uint64_t)r.n_units + r.n_left_in;
    const uint64_t wpb = r.seg ? (uint64_t)seg_waves_per_block((int)s->p.k) : (uint64_t)k2_waves_per_block((int)s->p.k);
    if (r.seg) s->n_seg_launches++;
    // (whole workgroups run: that many waves pull, insert and may leave a leftover entry; max_waves is a multiple of wpb)
    const uint64_t waves = (std::max<uint64_t>(1, std::min<uint64_t>(work_units, s->max_waves)) + wpb - 1) / wpb * wpb;
    a.n_waves = (uint32_t)waves;
    // per-wave insert budget: the table and the shard lists are sized for max_waves waves inserting
    // WAVE_BUDGET + WAVE_OVERSHOOT new hashes each, so a launch with fewer waves may let each of them insert
    // proportionally more (warm-up ranges at a loose threshold would otherwise stop after one tile per wave
    // and be relaunched for the rest)
    {
        const uint64_t per_wave = (uint64_t)WAVE_BUDGET + WAVE_OVERSHOOT;
        const uint64_t wps_max = (s->max_waves + N_SHARDS - 1) / N_SHARDS, wps = (waves + N_SHARDS - 1) / N_SHARDS;
        const uint64_t b_table = s->max_waves * per_wave / waves, b_shard = wps_max * per_wave / wps;
        a.wave_budget = (uint32_t)std::min<uint64_t>(std::min(b_table, b_shard) - WAVE_OVERSHOOT, 1u << 30);
    }
    const int blocks = (int)((waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->profiling) {
        if (s->prof_used == s->prof_events.size()) {
            hipEvent_t a0, a1;
            HIP_TRY(hipEventCreate(&a0));
            HIP_TRY(hipEventCreate(&a1));
            s->prof_events.emplace_back(a0, a1);
        }
        e0 = s->prof_events[s->prof_used].first;
        e1 = s->prof_events[s->prof_used].second;
        s->prof_used++;
        HIP_TRY(hipEventRecord(e0, s->stream));
    }
    HIP_TRY(launch_k2((int)s->p.k, a, blocks, s->stream));
    if (s->profiling) {
        HIP_TRY(hipEventRecord(e1, s->stream));
        s->prof_launches++;
    }
    if (s->fast) {
        // shard lists -> live list, the in-stream selection and (a speculative range) the verdict: one launch
        // (not launched yet: flush_epilogue, or folded into fh_finish's)
        s->epi_pending = EPI_FLATTEN | (s->open_loop && !r.verdict ? EPI_PRUNE_TRIGGER : EPI_PRUNE_FORCE) |
                         (r.gated ? EPI_GATED : 0u) | (r.verdict ? EPI_VERDICT : 0u);
        s->epi_units = r.n_units;
    } else {
        HIP_TRY(launch_live_flatten(s->ctl, s->stream)); // shard lists -> flat live list, n_live
        if (!s->big_mode)
            HIP_TRY(launch_prune_small(s->table, s->live, s->dead, s->dead_cap, s->ctl, s->p.kind, s->p.size, s->max_hash,
                                       s->trigger, s->open_loop ? 0u : 1u, 0u, s->stream));
    }
    s->n_launches++;
    return FH_OK;
}

// wait for the pending range; if its launch stopped early (table near its guarded size), prune and
// relaunch until the queue is dry
int drain(fh_sketcher *s) {
    if (s->spec.pending && !s->pend.active) { // nothing but the verdict of a speculative range to wait for
        if (int rc = check_ctl(s)) return rc;
        if (!s->h_ctl->spec_ok) return recover_spec(s);
        s->spec.pending = false;
        s->last_tau = s->h_ctl->tau;
        s->last_live = s->h_ctl->n_live;
        return FH_OK;
    }
    while (s->pend.active) {
        if (int rc = check_ctl(s)) return rc;
        if (s->spec.pending) { // the range waited for was queued behind a speculation
            if (!s->h_ctl->spec_ok) {
                if (int rc = recover_spec(s)) return rc;
                continue;
            }
            s->spec.pending = false;
            s->pend.gated = false;
        }
        const Ctl c = *s->h_ctl;
        s->last_tau = c.tau;
        s->last_live = c.n_live;
        const bool remaining = c.next_unit < s->pend.n_units || c.n_left_out > 0;
        static const bool trace = cfg("trace") != nullptr; // per-launch outcome on stderr (debug aid)
        if (trace)
            fprintf(stderr, "[fh] launch %llu: units %u/%u left_in %u left_out %u n_live %u stopped %u tau %.3e soft %u\n",
                    (unsigned long long)s->n_launches, c.next_unit, s->pend.n_units, s->pend.n_left_in, c.n_left_out,
                    c.n_live, c.stopped, (double)c.tau, soft_limit_of(s));
        if (trace && c.dbg_wave_cycles)
            fprintf(stderr, "[fh]   (cumulative) flush: %.3g wave-cycles in %llu calls, %llu entries; waves alive %.3g cycles => %.1f %% of wave time, %.0f cycles per call\n",
                    (double)c.dbg_flush_cycles, (unsigned long long)c.dbg_flush_calls, (unsigned long long)c.dbg_flush_entries,
                    (double)c.dbg_wave_cycles, 100.0 * (double)c.dbg_flush_cycles / (double)c.dbg_wave_cycles,
                    (double)c.dbg_flush_cycles / (double)std::max<uint64_t>(c.dbg_flush_calls, 1));
        if (c.need_big || (s->big_mode && 2 * (uint64_t)c.n_live >= s->live_target) ||
            (remaining && c.n_live >= soft_limit_of(s) / 2)) {
            if (s->big_mode || c.need_big || c.n_live > (uint32_t)SMALL_MAX) {
                if (int rc = big_prune(s, false)) return rc;
            } else {
                HIP_TRY(launch_prune_small(s->table, s->live, s->dead, s->dead_cap, s->ctl, s->p.kind, s->p.size,
                                           s->max_hash, 0u, 1u, 0u, s->stream));
            }
        }
        if (!remaining) {
            // novelty of this range: new hashes per admitted occurrence (the threshold only went down while it ran, so
            // dividing by the admit rate it started with errs low by at most the stops it had; a short or stopped
            // range is not trusted to say "nothing new")
            const uint64_t dpos = s->pend.p_end - s->pend.p_begin;
            const uint64_t dins = c.inserted_total - s->ins_seen;
            s->ins_seen = c.inserted_total;
            const double expected = (double)dpos * std::max(0.0, s->pend.admit_at_start - (s->tau_lo ? admit_rate(s->tau_lo) : 0.0));
            if (expected >= 200.0) s->novelty = std::min(1.0, (double)(dins + 1) / expected);
            else if (dins) s->novelty = 1.0;
            s->pend.active = false;
            break;
        }
        s->pend.n_left_in = c.n_left_out;
        s->pend.left_cur ^= 1;
        s->pend.gated = s->pend.verdict = false; // (a relaunch is ordinary work: whatever it was queued behind has been looked at)
        HIP_TRY(launch_queue_reset(s->ctl, 0u, soft_limit_of(s), read_first_of(s), s->stream));
        s->n_relaunches++;
        if (int rc = launch_pending(s)) return rc;
    }
    return FH_OK;
}

// sketch [0,len) of a device-resident packed stream whose first byte has stream coordinate base_pos
// (spec_tau != 0: a speculative range -- the threshold rides on the queue reset and the epilogue takes the verdict; gated:
//  the range is queued behind a speculation nobody has looked at yet)
int start_range(fh_sketcher *s, const uint8_t *d_seq, uint64_t len, uint64_t base_pos, uint64_t pos, uint64_t end,
                uint64_t spec_tau = 0, bool gated = false) {
    fh_sketcher::Pending &r = s->pend;
    r.seq = d_seq;
    r.len = len;
    r.base_pos = base_pos;
    r.p_begin = pos;
    r.p_end = end;
    // the segment kernel where the block has a stride and the launch is the plain one (any seed; no test mask, no lower threshold)
    // (K > 32: fh_k2ws.hip takes seed, mask and lower threshold at run time, as fh_k2w.hip does)
    r.seg = (s->blk_seg && (s->p.k > 32 || (!s->p.hash_mask && !s->tau_lo)) && pos % s->gran == 0) ? s->blk_seg : 0u;
    r.seg_sub = r.seg ? s->blk_sub : 1u;
    const uint64_t tile = r.seg ? (uint64_t)seg_tile_pos(r.seg, r.seg_sub) : (uint64_t)TILE_POS;
    const uint64_t tiles = (end - pos + tile - 1) / tile;
    if (tiles >= (1ull << 31)) return fail(FH_ERR_INVALID, "block too large for one range");
    r.tiles_total = (uint32_t)tiles;
    // an input of a few megabases does not fill the chip with units of two tiles (configs[4]: 4 Mb = 977 of them for 1024
    // SIMDs): single tiles put twice as many waves to work, each for half as long
    static const uint32_t unit_knob = [] {
        const char *e = cfg("unit_tiles"); // A/B knob: 0 = by size
        return e ? (uint32_t)atoi(e) : 0u;
    }();
    r.unit_tiles = unit_knob ? unit_knob : (tiles < (uint64_t)UNIT_TILES * 2 * s->max_waves ? 1u : (uint32_t)UNIT_TILES);
    r.max_units = MAX_UNITS;
    if (r.seg) {
        // a segment tile is 64 x stride positions, about five of k2_sketch's: single tiles are the units, and a pull takes about
        // what eight of k2_sketch's units are (the chip's reads stay inside one moving window of a few hundred megabytes)
        static const uint32_t seg_pull = [] {
            const char *e = cfg("seg_pull_pos"); // A/B knob: positions a pull takes at most
            return e ? (uint32_t)atoi(e) : (uint32_t)MAX_UNITS * UNIT_TILES * TILE_POS;
        }();
        if (!unit_knob) r.unit_tiles = 1u;
        r.max_units = std::max<uint32_t>(1u, seg_pull / (uint32_t)(tile * r.unit_tiles));
    }
    r.n_units = (uint32_t)((tiles + r.unit_tiles - 1) / r.unit_tiles);
    r.n_left_in = 0;
    r.left_cur = 0;
    r.gated = gated;
    r.verdict = spec_tau != 0;
    r.admit_at_start = admit_rate(s->last_tau);
    // The first launch's waves (launch_pending: min(units, max_waves), rounded up to whole workgroups) start on units of
    // their own, no atomic: all of a short range's units (at most MAX_UNITS per wave: the work per tile is uniform, there is
    // nothing to balance), the first guided pull's worth of a long one -- the queue, which begins behind them, hands out the
    // rest.  (Handing a long range out statically is NOT faster, on the contrary: with 4096 waves each streaming through a
    // 12 MB stretch of its own the launch ran at 505-525 Gbases/s against 619 through the queue, which keeps the chip's
    // reads inside one moving window of a few hundred megabytes -- profiles/r04_ab_static_units.txt.)
    static const bool no_static = cfg("no_static_units") != nullptr; // A/B knob
    {
        const uint64_t waves = std::max<uint64_t>(1, std::min<uint64_t>(r.n_units, s->max_waves));
        const uint64_t wpb = r.seg ? (uint64_t)seg_waves_per_block((int)s->p.k) : (uint64_t)k2_waves_per_block((int)s->p.k);
        const uint64_t grid = (waves + wpb - 1) / wpb * wpb;
        const uint64_t fu = std::min<uint64_t>((r.n_units + waves - 1) / waves, r.max_units);
        r.first_units = no_static ? 0u : (uint32_t)fu;
        r.grid_waves = (uint32_t)grid;
    }
    const uint32_t first_total = (uint32_t)std::min<uint64_t>(r.n_units, (uint64_t)r.first_units * r.grid_waves);
    if (int rc = flush_epilogue(s)) return rc;
    HIP_TRY(launch_queue_reset(s->ctl, 1u, soft_limit_of(s), read_first_of(s), s->stream, spec_tau != 0, spec_tau, gated, first_total));
    if (int rc = launch_pending(s)) return rc;
    r.active = true;
    if (s->profiling) s->prof_positions += end - pos;
    s->dirty = true;
    return FH_OK;
}

int set_tau(fh_sketcher *s, uint64_t tau) {
    if (int rc = flush_epilogue(s)) return rc;
    HIP_TRY(launch_set_tau(s->ctl, tau, s->stream));
    s->last_tau = tau;
    return FH_OK;
}

// First block of a fresh sketcher: instead of warming the threshold up through a series of small closed-loop ranges
// (half a dozen to a dozen host round trips, which dominate a file of a few Mb and still cost ~1 ms of a 10 Gbase
// pass), guess it from the length (about 4 x size hashes expected below it if every k-mer were distinct) and sketch
// the block -- or, for a block above 64 M positions, its first 32 M positions -- in one go.  As soon as `size` distinct
// hashes <= the guess have been seen the usual argument holds (everything not recorded is larger than the size-th
// smallest so far) and the rest of the input continues from a tight threshold.  If the speculated part ends with
// fewer -- low-complexity input -- it is read a second time for the hashes above the guess only (HASLO launches),
// through the normal loop; a wrong guess therefore costs at most one extra pass over 64 M positions.
constexpr uint64_t SPEC_MAX_POS = 64ull << 20, SPEC_PREFIX_POS = 32ull << 20; // multiples of TILE_POS
int speculative_first_block(fh_sketcher *s, const uint8_t *d_seq, uint64_t len, uint64_t base_pos, uint64_t n_pos,
                            bool *done) {
    *done = false;
    const double want = 4.0 * (double)std::max<uint64_t>(s->p.size, 1);
    if (s->no_spec || s->max_range || s->p.hash_mask || n_pos < 32768 || n_pos > SPEC_MAX_POS ||
        want * 2.0 >= (double)n_pos)
        return FH_OK;
    if (s->p.kind == FH_KIND_SCALED && s->p.size == 0) return FH_OK; // threshold is max_hash from the start
    uint64_t tau_spec = (uint64_t)(want / (double)n_pos * 18446744073709551616.0);
    if (s->p.kind == FH_KIND_SCALED) tau_spec = std::max(tau_spec, s->max_hash);
    if (tau_spec == 0 || tau_spec >= EMPTY64 - 1) return FH_OK;
    if (s->fast && !s->big_mode && s->p.kind == FH_KIND_MASH && s->p.size > 0) {
        // Not waited for: the range's epilogue takes the verdict on the device (Ctl::spec_ok), what the caller queues behind
        // it is gated on that, and the host reads the outcome at its next synchronisation (drain / fh_finish).  Until
        // then the bookkeeping assumes success: at least `size` hashes at or below the guess.
        s->n_spec++;
        s->n_spec_deferred++;
        const bool was_open = s->open_loop;
        s->open_loop = true;
        const int rc = start_range(s, d_seq, len, base_pos, 0, n_pos, tau_spec, false);
        s->open_loop = was_open;
        if (rc) return rc;
        s->spec.pending = true;
        s->spec.range = s->pend;
        s->spec.tau = tau_spec;
        s->spec.n_pos = n_pos;
        s->pend.active = false;
        s->last_tau = tau_spec;
        s->last_live = (uint32_t)s->p.size;
        s->positions_done += n_pos;
        *done = true;
        return FH_OK;
    }
    if (int rc = set_tau(s, tau_spec)) return rc;
    s->n_spec++;
    const bool was_open = s->open_loop;
    s->open_loop = true; // one range for the whole block; waves stop by themselves if the live set fills up
    if (int rc = start_range(s, d_seq, len, base_pos, 0, n_pos)) return rc;
    if (int rc = drain(s)) return rc;
    s->open_loop = was_open;
    // settle the live set and see whether the guess captured `size` distinct hashes
    if (s->big_mode || s->last_live > (uint32_t)SMALL_MAX) {
        if (int rc = big_prune(s, false)) return rc;
    } else {
        HIP_TRY(launch_prune_small(s->table, s->live, s->dead, s->dead_cap, s->ctl, s->p.kind, s->p.size, s->max_hash, 0u,
                                   1u, 0u, s->stream));
        if (int rc = check_ctl(s)) return rc;
        s->last_tau = s->h_ctl->tau;
        s->last_live = s->h_ctl->n_live;
    }
    if ((uint64_t)s->last_live >= s->p.size) {
        s->positions_done += n_pos;
        *done = true;
        return FH_OK;
    }
    // too few: everything <= tau_spec is in the table with exact counts; re-read the block for the rest
    s->n_spec_fallback++;
    s->tau_lo = tau_spec;
    if (int rc = set_tau(s, EMPTY64)) return rc;
    return FH_OK;
}

// Large sketches (kmers_to_sketch in the tens of thousands to millions: the CLI's oversketch) on a large first block: take
// the threshold from a sample of the block instead of discovering it by filling the table (fh_kernels.hip,
// "sampling pre-pass").  The sample pass is the sketch kernel itself over one run of SAMPLE_RUN_TILES tiles in every
// 64 runs' worth (1/64 of the positions, ~1.6 % of a pass), at a cap threshold; the table then holds the sample's hashes
// with their sample multiplicities, is read out into three histograms, and cleared again.  On success the whole block is
// sketched in ONE launch at a threshold a little above its final one.  If the estimate was too tight -- fewer than `size`
// hashes live at the end -- everything at or below it is in the table with exact counts and the block is re-read for the
// hashes above it, exactly like a failed speculation.
constexpr uint32_t SAMPLE_RUN_TILES_DEFAULT = 8, SAMPLE_ONE_IN_DEFAULT = 64;
static uint32_t sample_knob(const char *name, uint32_t dflt) { // measurement knobs (options sample_run_tiles, sample_one_in)
    const char *e = cfg(name);
    const long v = e ? atol(e) : 0;
    return v > 0 && v < 65536 ? (uint32_t)v : dflt;
}
int sampled_first_block(fh_sketcher *s, const uint8_t *d_seq, uint64_t len, uint64_t base_pos, uint64_t n_pos, bool *attempted,
                        bool *done) {
    *attempted = *done = false;
    static const bool off = cfg("no_sample") != nullptr;
    static const uint64_t min_pos = [] {
        const char *e = cfg("sample_min_pos"); // test knob
        return e ? strtoull(e, nullptr, 10) : (256ull << 20);
    }();
    static const double scale_knob = [] {
        const char *e = cfg("sample_scale"); // test knob: multiply the estimated threshold (< 1 forces the repair pass)
        return e ? atof(e) : 1.0;
    }();
    if (off || s->no_spec || s->max_range || s->p.hash_mask || !s->big_mode || s->p.kind != FH_KIND_MASH || s->p.size < 16384 ||
        n_pos < min_pos || (double)n_pos < 64.0 * (double)s->p.size)
        return FH_OK;
    static const uint32_t run_tiles_knob = sample_knob("sample_run_tiles", 0);
    static const uint32_t SAMPLE_ONE_IN = sample_knob("sample_one_in", SAMPLE_ONE_IN_DEFAULT);
    static const double cap_knob = [] {
        const char *e = cfg("sample_cap_scale"); // measurement knob: scales the sample pass's cap threshold
        return e ? atof(e) : 1.0;
    }();
    const uint64_t tiles = (n_pos + TILE_POS - 1) / TILE_POS;
    // One run per resident wave where the block is large enough: the runs are dealt out one at a time, and 9 600 runs of 8 tiles
    // on 4 096 waves are three rounds of which the last is a third full (0.78 ms for a 10 Gbase block; 4 100 runs of 19 tiles:
    // 0.51 ms, profiles/r05_sample_runs.txt).  Between 8 and 128 tiles a run; the sampled share stays one tile in SAMPLE_ONE_IN.
    const uint32_t SAMPLE_RUN_TILES = run_tiles_knob ? run_tiles_knob
        : (uint32_t)std::min<uint64_t>(128, std::max<uint64_t>(SAMPLE_RUN_TILES_DEFAULT, (tiles + (uint64_t)SAMPLE_ONE_IN * s->max_waves - 1) /
                                                                                             ((uint64_t)SAMPLE_ONE_IN * s->max_waves)));
    const uint32_t stride = SAMPLE_RUN_TILES * SAMPLE_ONE_IN;
    const uint64_t n_runs = tiles / stride;
    if (n_runs < 64 || tiles >= (1ull << 31)) return FH_OK;
    if (!s->smp_hist) {
        HIP_TRY(dev_malloc(&s->smp_hist, 768 * sizeof(uint32_t)));
        HIP_TRY(host_malloc(&s->h_smp_hist, 768 * sizeof(uint32_t)));
    }
    if (n_runs > s->smp_list_cap) {
        (void)hipFree(s->smp_list);
        s->smp_list = nullptr;
        s->smp_list_cap = 0;
        HIP_TRY(dev_malloc(&s->smp_list, (size_t)n_runs * 2 * sizeof(uint32_t)));
        s->smp_list_cap = n_runs;
    }
    // cap: the estimate has to reach `want` distinct k-mers BELOW the cap, and an occurrence sample sees a k-mer of
    // multiplicity m with probability ~m / 64 only.  The pass is bound by its admits, not by the positions it reads, so it
    // first runs at a cap that lets about size / 2 sample occurrences into the table (reads with sequencing errors: most
    // distinct k-mers are singletons, and the estimate crosses `want` far below even that); only if the estimate does not
    // get there is the pass repeated at 2 x size occurrences (half the table's soft limit).
    const double n_samples = (double)n_runs * SAMPLE_RUN_TILES * TILE_POS;
    uint64_t tau_guess = 0, tau_cap = 0;
    double S = 0, c1 = 0, c2 = 0;
    for (const double cap_occ : {0.5, 2.0}) {
    const double cap_frac = std::min(0.25, cap_knob * cap_occ * (double)s->p.size / n_samples);
    tau_cap = (uint64_t)(cap_frac * 18446744073709551616.0);
    // ---- the sample pass: the sketch kernel over the tile runs, handed over as a "leftover" list ----
    if (int rc = set_tau(s, tau_cap)) return rc;
    HIP_TRY(launch_fill_tile_runs(s->smp_list, (uint32_t)n_runs, stride, SAMPLE_RUN_TILES, (uint32_t)tiles, s->stream));
    HIP_TRY(launch_queue_reset(s->ctl, 1u, soft_limit_of(s), 1u, s->stream));
    {
        SketchArgs a{};
        a.seq = d_seq;
        a.len_total = len;
        a.p_begin = 0;
        a.p_end = n_pos;
        a.base_pos = base_pos;
        a.seed = s->p.seed;
        a.hash_mask = ~0ull;
        a.tau_lo = 0;
        a.ctl = s->ctl;
        a.tiles_total = (uint32_t)tiles;
        a.n_units = 0; // nothing but the list
        a.n_left_in = (uint32_t)n_runs;
        a.left_in = s->smp_list;
        a.left_out = s->left_buf[0];
        const uint64_t wpb = (uint64_t)k2_waves_per_block((int)s->p.k);
        const uint64_t waves = (std::max<uint64_t>(1, std::min<uint64_t>(n_runs, s->max_waves)) + wpb - 1) / wpb * wpb;
        a.n_waves = (uint32_t)waves;
        a.wave_budget = WAVE_BUDGET;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (s->profiling) {
            if (s->prof_used == s->prof_events.size()) {
                hipEvent_t a0, a1;
                HIP_TRY(hipEventCreate(&a0));
                HIP_TRY(hipEventCreate(&a1));
                s->prof_events.emplace_back(a0, a1);
            }
            e0 = s->prof_events[s->prof_used].first;
            e1 = s->prof_events[s->prof_used].second;
            s->prof_used++;
            HIP_TRY(hipEventRecord(e0, s->stream));
        }
        HIP_TRY(launch_k2((int)s->p.k, a, (int)((waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), s->stream));
        if (s->profiling) {
            HIP_TRY(hipEventRecord(e1, s->stream));
            s->prof_launches++;
            s->prof_positions += (uint64_t)n_samples;
        }
        s->n_launches++;
    }
    HIP_TRY(launch_live_flatten(s->ctl, s->stream));
    HIP_TRY(launch_live_count_hist(s->table, s->live, s->ctl, s->smp_hist, s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_smp_hist, s->smp_hist, 768 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    if (int rc = check_ctl(s)) return rc; // (synchronises)
    const bool sample_clean = !s->h_ctl->stopped && s->h_ctl->n_left_out == 0;
    // the table goes back to empty: the sample's counts are sample counts
    HIP_TRY(launch_clear_slots(s->table, s->cap, s->live, s->dead, s->ctl, s->stream));
    HIP_TRY(launch_init_ctl(s->ctl, initial_tau(s), s->stream, true));
    s->last_tau = initial_tau(s);
    s->last_live = 0;
    if (!sample_clean) return FH_OK; // (a stream of few, very frequent k-mers filled a wave's budget: nothing to estimate)
    // The threshold below which the whole block is estimated to hold `want` distinct hashes: the first quarter-octave
    // bucket whose upper edge reaches the estimate, and within it the point where the estimate -- linear in the threshold
    // inside a bucket, hashes being uniform -- crosses `want`.  (Taking the bucket's upper edge, a factor 2^(1/4) above its
    // lower one, left up to 1.49 x size entries live where 1.25 x was asked for: round 2's configs[2].)  The estimator
    // (Chao1: seen + singletons^2 / 2 doubletons) errs low, i.e. towards a threshold that is too high, never too tight.
    static const double want_factor = [] {
        const char *e = cfg("sample_want"); // A/B knob
        return e ? std::max(1.0, atof(e)) : 1.15;
    }();
    const uint32_t *H = s->h_smp_hist;
    const double want = want_factor * (double)s->p.size;
    double est_prev = -1.0;
    S = c1 = c2 = 0;
    const uint32_t q_cap = qoct_index(tau_cap);
    for (uint32_t q = 0; q < 256 && q < q_cap; ++q) { // (the bucket that holds the cap is only partly sampled)
        S += H[q];
        c1 += H[256 + q];
        c2 += H[512 + q];
        // too few doubletons to say anything about the unseen (the low-cap pass asks for more of them: its estimate rests on a
        // quarter of the sample, and a threshold that comes out too tight costs a second pass over the block)
        if (S < 1024 || c2 < (cap_occ < 1.0 ? 128 : 32)) continue;
        const double est = S + c1 * c1 / (2.0 * c2);
        if (est >= want) {
            const uint64_t hi = qoct_upper_edge(q), lo = q ? qoct_upper_edge(q - 1) : 0;
            tau_guess = hi;
            if (est_prev >= 0.0 && est > est_prev && want > est_prev)
                tau_guess = lo + (uint64_t)((double)(hi - lo) * std::min(1.0, (want - est_prev) / (est - est_prev)));
            break;
        }
        est_prev = est;
    }
    if (tau_guess || cap_frac >= 0.25) break;
    }
    static const bool trace = cfg("trace") != nullptr;
    if (trace)
        fprintf(stderr, "[fh] sample: %llu runs of %u tiles, cap %.3e: S %.0f c1 %.0f c2 %.0f -> tau %.3e (%s)\n",
                (unsigned long long)n_runs, SAMPLE_RUN_TILES, (double)tau_cap, S, c1, c2, (double)tau_guess, tau_guess ? "guess" : "none");
    if (!tau_guess) return FH_OK;
    if (scale_knob != 1.0) tau_guess = (uint64_t)std::min(1.8e19, std::max(1.0, (double)tau_guess * scale_knob));
    *attempted = true;
    s->n_sampled++;
    if (int rc = set_tau(s, tau_guess)) return rc;
    const bool was_open = s->open_loop;
    s->open_loop = true; // one range for the whole block; waves stop by themselves if the live set fills up after all
    if (int rc = start_range(s, d_seq, len, base_pos, 0, n_pos)) return rc;
    if (int rc = drain(s)) return rc;
    s->open_loop = was_open;
    if (int rc = big_prune(s, false)) return rc;
    if ((uint64_t)s->last_live >= s->p.size) {
        s->positions_done += n_pos;
        *done = true;
        return FH_OK;
    }
    // too tight: everything <= tau_guess is in the table with exact counts; re-read the block for the rest
    s->n_spec_fallback++;
    s->tau_lo = tau_guess;
    if (int rc = set_tau(s, EMPTY64)) return rc;
    return FH_OK;
}

int sketch_device_range(fh_sketcher *s, const uint8_t *d_seq, uint64_t len, uint64_t base_pos) {
    if (int rc = drain(s)) return rc;
    if (len < s->p.k) return FH_OK;
    const uint64_t n_pos = len - s->p.k + 1; // windows that fit
    // Records of one length?  Then the segment kernel skips what every record's last k positions cannot hold (fh_k2s.hip).  The
    // caller may have said so (fh_set_record_stride); a large block is asked itself -- one wavefront, one round trip (~20 us:
    // worth it from a few milliseconds of sketching on).  The stride only decides how fast, never what comes out.
    s->blk_seg = 0;
    s->blk_sub = 1;
    s->gran = TILE_POS;
    bool probe_behind = false;
    {
        static const bool seg_off = cfg("no_seg") != nullptr; // A/B knob
        // A block is asked for its stride BEHIND its own launches, without a wait (the answer lands in pinned memory): what
        // the last block this handle asked said is what the next one goes by -- the blocks of one input, or the passes over one
        // buffer, share their read length, and a stale or wrong stride costs speed, never the result.  Only a handle that has
        // not asked yet waits for the answer (one wavefront and a round trip, ~0.1-0.3 ms), and only for a block large enough
        // not to notice.
        static const uint64_t probe_min = [] {
            const char *e = cfg("seg_probe_min"); // test knob: blocks from this size on are asked (behind their launches)
            return e ? strtoull(e, nullptr, 10) : (16ull << 20);
        }();
        static const uint64_t probe_wait_min = [] {
            const char *e = cfg("seg_probe_wait_min"); // test knob: ... and waited for, if the handle has no answer yet
            return e ? strtoull(e, nullptr, 10) : (256ull << 20);
        }();
        static const uint32_t seg_env = [] {
            const char *e = cfg("seg_stride"); // test knob: every block through the segment kernel with this stride
            return e ? (uint32_t)atoi(e) : 0u;
        }();
        uint32_t S = seg_env ? seg_env : s->seg_hint;
        if (!seg_off && S != 1u && (s->p.k > 32 || !s->p.hash_mask)) {
            if (S == 0 && len >= probe_min) {
                if (!s->h_probe) {
                    HIP_TRY(host_malloc(&s->h_probe, 64));
                    s->h_probe[0] = 0;
                }
                if (s->probe_seen) {
                    // (the newest probe may still be running -- it sits behind its block's launches: its answer is taken once
                    // its event has completed, until then the one before it stands)
                    if (s->probe_ev && hipEventQuery(s->probe_ev) == hipSuccess) s->probe_answer = s->h_probe[0], s->probe_breakers = s->h_probe[1];
                    else (void)hipGetLastError();
                    S = s->probe_answer;
                } else if (len >= probe_wait_min) {
                    if (int rc = flush_epilogue(s)) return rc;
                    HIP_TRY(launch_seg_probe(d_seq, len, s->h_probe, s->stream));
                    HIP_TRY(hipStreamSynchronize(s->stream));
                    S = s->probe_answer = s->h_probe[0];
                    s->probe_breakers = s->h_probe[1];
                    s->n_seg_probes++;
                    s->probe_seen = true;
                }
                probe_behind = true;
            }
            // (records of up to SEG_MAX_RECORD - 1 bases, two or four lanes to a record beyond SEG_MAX_STRIDE; the two-word kernels
            // take a lane per record only)
            const uint32_t s_max = s->p.k > 32 ? SEG_MAX_STRIDE : SEG_MAX_RECORD;
            if (S >= SEG_MIN_STRIDE && S <= s_max && S > s->p.k && n_pos >= 64ull * S) {
                s->blk_seg = S;
                s->blk_sub = seg_sub_for(S);
                s->gran = seg_tile_pos(S, s->blk_sub);
            } else if (seg_ragged_k((int)s->p.k) && n_pos >= 64ull * SEG_RAGGED_STRIDE) {
                // no one stride, but records of a few hundred bases at most (a breaker in every 256 bytes the probe looked at, or
                // more): the work-item form.  Option seg_ragged: 0 = never, 1 = whatever the block holds (tests)
                static const int rag_opt = cfg("seg_ragged") ? atoi(cfg("seg_ragged")) : -1;
                const bool dense = S == 0 && s->seg_hint == 0 && probe_behind && s->probe_seen && s->probe_breakers >= 16u;
                if (rag_opt == 1 || (rag_opt != 0 && dense)) {
                    s->blk_seg = SEG_RAGGED_STRIDE;
                    s->blk_sub = SEG_RAGGED;
                    s->gran = seg_tile_pos(SEG_RAGGED_STRIDE, SEG_RAGGED);
                }
            }
        }
    }
    struct ProbeBehind { // (on every way out of this function that has launched the block's work)
        fh_sketcher *s;
        const uint8_t *seq;
        uint64_t len;
        bool on;
        ~ProbeBehind() {
            if (!on || !s->h_probe) return;
            if (launch_seg_probe(seq, len, s->h_probe, s->stream) == hipSuccess) {
                if (!s->probe_ev && hipEventCreateWithFlags(&s->probe_ev, hipEventDisableTiming) != hipSuccess) s->probe_ev = nullptr;
                if (s->probe_ev) (void)hipEventRecord(s->probe_ev, s->stream);
                s->probe_seen = true;
                s->n_seg_probes++;
            }
        }
    } probe_guard{s, d_seq, len, probe_behind};
    uint64_t pos = 0;
    uint64_t lo_end = 0; // a failed speculation re-reads [0, lo_end) for the hashes above its guess only
    bool sampled = false;
    if (s->positions_done == 0 && s->tau_lo == 0) {
        bool done = false;
        if (int rc = sampled_first_block(s, d_seq, len, base_pos, n_pos, &sampled, &done)) return rc;
        if (sampled && done) return FH_OK;
        if (sampled) lo_end = n_pos; // (tau_lo is set: the loop below re-reads the block for the hashes above the estimate)
    }
    if (!sampled && s->positions_done == 0 && s->tau_lo == 0) {
        // a large first block speculates on its first 32 M positions only: a wrong guess then costs a second pass
        // over that prefix (0.1 ms), a right one replaces the ten closed-loop warm-up ranges and their round trips
        // (option spec_prefix_pos: the positions a large first block speculates on -- measurement knob)
        static const uint64_t prefix_pos = cfg("spec_prefix_pos") ? std::max<uint64_t>(1ull << 20, strtoull(cfg("spec_prefix_pos"), nullptr, 10)) : SPEC_PREFIX_POS;
        const uint64_t spec_pos = n_pos <= SPEC_MAX_POS ? n_pos : std::max<uint64_t>(prefix_pos / s->gran, 1) * s->gran;
        bool done = false;
        if (int rc = speculative_first_block(s, d_seq, len, base_pos, spec_pos, &done)) return rc;
        if (done) {
            if (spec_pos == n_pos) return FH_OK;
            pos = spec_pos;
        } else if (s->tau_lo) {
            if (int rc = reread_above(s, d_seq, len, base_pos, 0, spec_pos, s->tau_lo)) return rc;
            if (spec_pos == n_pos) return FH_OK;
            pos = spec_pos;
        }
    }
    return sketch_positions(s, d_seq, len, base_pos, pos, n_pos, lo_end);
}

// positions [pos, n_pos) of a device-resident block through the range loop; with s->tau_lo set, [pos, lo_end) are re-read
// for the hashes above it only (a speculation that found too few)
int sketch_positions(fh_sketcher *s, const uint8_t *d_seq, uint64_t len, uint64_t base_pos, uint64_t pos, uint64_t n_pos, uint64_t lo_end) {
    struct LoGuard { // the lower bound only applies inside this call
        fh_sketcher *s;
        ~LoGuard() { s->tau_lo = 0; }
    } lo_guard{s};
    while (pos < n_pos) {
        // (a speculation whose verdict is still out does not hold the next range up: that one is queued gated)
        if (s->pend.active)
            if (int rc = drain(s)) return rc;
        if (s->tau_lo && pos >= lo_end) s->tau_lo = 0; // the re-read of the speculated prefix is complete (drained)
        if (!s->open_loop) {
            // status of everything before this range is known (drained): decide whether the threshold is
            // tight enough to let launches run to completion on their own
            // (positions the resident waves hold at once: a tile each -- of the kernel that runs the block, s->gran: a segment tile is
            // about five of k2_sketch's)
            const double inflight = (double)(s->max_waves * std::max<uint64_t>(s->gran, (uint64_t)TILE_POS));
            const double room = (double)s->live_target -
                                (double)std::min<uint64_t>(std::max<uint64_t>(s->p.size, s->last_live), s->live_target);
            if (s->positions_done > 0 && inflight * fill_rate(s) <= 0.25 * room) s->open_loop = true;
            // with the threshold refreshed inside the launch (Ctl::hist) what a launch inserts no longer grows with its
            // length: ~size x ln(distinct after / distinct before), and the waves stop by themselves should that fill the
            // live set after all
            if (s->hist && s->positions_done > 0 && (s->spec.pending || (uint64_t)s->last_live >= s->p.size)) s->open_loop = true;
        }
        const uint64_t limit = s->tau_lo ? lo_end : n_pos;
        const uint64_t P = next_range_size(s, limit - pos);
        const uint64_t end = std::min<uint64_t>(limit, pos + P);
        if (int rc = start_range(s, d_seq, len, base_pos, pos, end, 0, s->spec.pending)) return rc;
        s->positions_done += end - pos;
        pos = end;
        if (!s->open_loop || s->tau_lo)
            if (int rc = drain(s)) return rc; // closed loop: the next range is sized from this one's outcome
    }
    return FH_OK;
}

// Positions [p_begin, p_end) hold fewer than `size` distinct hashes at or below `lo` (all of them in the table, counts exact):
// a speculative threshold that assumed every k-mer distinct, on input that repeats itself -- reads at thirty-fold coverage
// with few errors hold a thirtieth of the distinct k-mers their length suggests.  The range is read again for the hashes above
// `lo`: not at once for all of them (that fills the table with every distinct k-mer of the range, selection after selection:
// 12 ms for 40 M positions where the pass itself takes 0.1), but up to a threshold scaled by how far short the count fell
// -- the density of distinct hashes is known now --, and only if that too comes up short for everything.
int reread_above(fh_sketcher *s, const uint8_t *seq, uint64_t len, uint64_t base_pos, uint64_t p_begin, uint64_t p_end, uint64_t lo) {
    const bool no_scale = cfg("no_spec_rescale") != nullptr; // A/B (read per call: a rare path)
    for (int attempt = 0;; ++attempt) {
        uint64_t hi = EMPTY64;
        const uint64_t have = s->last_live;
        if (!no_scale && attempt < 2 && !s->big_mode && have >= 16 && have < s->p.size) {
            const double t = (double)lo * 4.0 * (double)s->p.size / (double)have; // about 4 x size hashes at or below it
            if (t < 1.7e19) hi = (uint64_t)t;
        }
        s->tau_lo = lo;
        if (int rc = set_tau(s, hi)) return rc;
        if (int rc = sketch_positions(s, seq, len, base_pos, p_begin, p_end, p_end)) return rc;
        if (int rc = drain(s)) return rc;
        if (hi == EMPTY64) return FH_OK;
        s->n_spec_rescaled++;
        static const bool trace = cfg("trace") != nullptr;
        if (trace) fprintf(stderr, "[fh] speculation fell short (%llu of %llu hashes at or below %.3e): range read again up to %.3e\n", (unsigned long long)have,
                           (unsigned long long)s->p.size, (double)lo, (double)hi);
        HIP_TRY(launch_prune_small(s->table, s->live, s->dead, s->dead_cap, s->ctl, s->p.kind, s->p.size, s->max_hash, 0u, 1u, 0u, s->stream));
        if (int rc = check_ctl(s)) return rc;
        s->last_tau = s->h_ctl->tau;
        s->last_live = s->h_ctl->n_live;
        if (s->h_ctl->need_big)
            if (int rc = big_prune(s, false)) return rc;
        if ((uint64_t)s->last_live >= s->p.size) return FH_OK;
        s->positions_done -= p_end - p_begin; // (the next reading of the range counts it again)
        lo = hi;
    }
}

// The verdict of a deferred speculation came back negative (Ctl::spec_ok == 0 after its epilogue): the speculative range
// stopped early or found fewer than `size` hashes below its guess, and nothing that was queued behind it has run.  Take
// the bookkeeping back, finish the range the step-by-step way (relaunches, then the re-read for the hashes above the
// guess if they are still too few) and sketch what was queued behind it through the ordinary loop.
int recover_spec(fh_sketcher *s) {
    const fh_sketcher::Pending queued = s->pend; // (gated; did nothing)
    const fh_sketcher::Spec sp = s->spec;
    s->spec.pending = false;
    s->n_spec_recovered++;
    s->positions_done -= sp.n_pos;
    if (queued.active) {
        s->positions_done -= queued.p_end - queued.p_begin;
        if (s->profiling) s->prof_positions -= queued.p_end - queued.p_begin;
    }
    s->open_loop = false;
    s->pend = sp.range;
    s->pend.active = true;
    s->pend.gated = s->pend.verdict = false;
    if (int rc = drain(s)) return rc; // (relaunches a range that stopped early until its queue is dry)
    HIP_TRY(launch_prune_small(s->table, s->live, s->dead, s->dead_cap, s->ctl, s->p.kind, s->p.size, s->max_hash, 0u, 1u, 0u, s->stream));
    if (int rc = check_ctl(s)) return rc;
    s->last_tau = s->h_ctl->tau;
    s->last_live = s->h_ctl->n_live;
    if (s->h_ctl->need_big)
        if (int rc = big_prune(s, false)) return rc;
    if ((uint64_t)s->last_live >= s->p.size) {
        s->positions_done += sp.n_pos;
    } else {
        // too few: everything <= the guess is in the table with exact counts; re-read the range for the rest
        s->n_spec_fallback++;
        if (int rc = reread_above(s, sp.range.seq, sp.range.len, sp.range.base_pos, sp.range.p_begin, sp.range.p_end, sp.tau)) return rc;
    }
    if (queued.active)
        if (int rc = sketch_positions(s, queued.seq, queued.len, queued.base_pos, queued.p_begin, queued.p_end, 0)) return rc;
    return FH_OK;
}

int check_ctl(fh_sketcher *s) {
    if (int rc = flush_epilogue(s)) return rc;
    HIP_TRY(hipMemcpyAsync(s->h_ctl, s->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->h_ctl->overflow == 1) return fail(FH_ERR_CAPACITY, "device hash table capacity exceeded");
    if (s->h_ctl->overflow == 2) return fail(FH_ERR_CAPACITY, "hash collision log capacity exceeded");
    return FH_OK;
}

int ensure_big_buffers(fh_sketcher *s, uint32_t M) {
    if (M <= s->big_cap) return FH_OK;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->keys_a) { (void)hipFree(s->keys_a); (void)hipFree(s->keys_b); (void)hipFree(s->slots_a); (void)hipFree(s->slots_b); (void)hipFree(s->sort_tmp); }
    s->keys_a = s->keys_b = nullptr; s->slots_a = s->slots_b = nullptr; s->sort_tmp = nullptr;
    const uint32_t cap = (uint32_t)std::min<uint64_t>((uint64_t)M + M / 2 + 1024, s->live_cap);
    HIP_TRY(dev_malloc(&s->keys_a, (size_t)cap * 8));
    HIP_TRY(dev_malloc(&s->keys_b, (size_t)cap * 8));
    HIP_TRY(dev_malloc(&s->slots_a, (size_t)cap * 4));
    HIP_TRY(dev_malloc(&s->slots_b, (size_t)cap * 4));
    HIP_TRY(big_sort_tmp_bytes(cap, &s->sort_tmp_bytes));
    HIP_TRY(dev_malloc(&s->sort_tmp, s->sort_tmp_bytes ? s->sort_tmp_bytes : 16));
    if (!s->keep_dev) HIP_TRY(dev_malloc(&s->keep_dev, 64 + SEL_SCRATCH_BYTES)); // [0,64) keep count, then select scratch
    s->big_cap = cap;
    return FH_OK;
}

// device-wide bottom-n selection (fh_big.hip); host-driven because it runs a handful of times per stream
// sorted = false (between launches): only the new threshold and the partition of the live list are needed, found by
// a radix select; fh_finish asks for the sorted live list (to_vec order)
int big_prune(fh_sketcher *s, bool sorted) {
    if (int rc = check_ctl(s)) return rc;
    const uint32_t M = s->h_ctl->n_live;
    if (int rc = ensure_big_buffers(s, M)) return rc;
    static const bool no_select = cfg("no_select") != nullptr; // A/B and debugging: always sort
    if (!sorted && !no_select && s->p.size >= 1)
        HIP_TRY(launch_big_prune_select(s->table, s->live, s->dead, s->dead_cap, s->ctl, M, s->h_ctl->n_dead, s->p.kind,
                                        s->p.size, s->max_hash, s->keys_a, s->slots_a, (char *)s->keep_dev + 64,
                                        s->keep_dev, s->stream));
    else
        HIP_TRY(launch_big_prune(s->table, s->live, s->dead, s->dead_cap, s->ctl, M, s->h_ctl->n_dead, s->p.kind,
                                 s->p.size, s->max_hash, s->keys_a, s->keys_b, s->slots_a, s->slots_b, s->sort_tmp,
                                 s->sort_tmp_bytes, s->keep_dev, s->stream));
    if (M == 0) {
        // nothing to sort; still clear the flag
        uint32_t zero = 0;
        HIP_TRY(hipMemcpyAsync(&s->ctl->need_big, &zero, 4, hipMemcpyHostToDevice, s->stream));
    }
    if (int rc = check_ctl(s)) return rc;
    s->last_tau = s->h_ctl->tau;
    s->last_live = s->h_ctl->n_live;
    s->n_big_prunes++;
    // scaled sketches keep everything <= max_hash: the live set itself grows with the input
    if (2 * (uint64_t)s->last_live > s->live_target) {
        s->live_target = 2 * (uint64_t)s->last_live;
        const uint64_t need = s->live_target + s->max_waves * (uint64_t)(WAVE_BUDGET + WAVE_OVERSHOOT) + 4096;
        if (need > s->live_cap) {
            if (int rc = grow_table(s, need + need / 2)) return rc;
        } else if (shard_cap_for(s, s->live_target) > s->shard_cap) {
            if (int rc = alloc_shards(s, s->live_target + s->live_target / 2)) return rc;
            HIP_TRY(launch_set_table(s->ctl, s->table, s->live, s->clog, s->cap, s->live_cap, CLOG_CAP, s->shard_cnt,
                                     s->shard_buf, s->shard_cap, s->kmer_hi, s->stream));
        }
    }
    return FH_OK;
}

int grow_table(fh_sketcher *s, uint64_t new_live_cap) {
    const uint64_t new_cap = 2 * new_live_cap;
    if (new_cap >= (1ull << 32)) return fail(FH_ERR_CAPACITY, "sketch state would exceed 2^32 table slots");
    Entry *nt = nullptr;
    uint32_t *nl = nullptr, *nd = nullptr;
    HIP_TRY(dev_malloc(&nt, new_cap * sizeof(Entry)));
    HIP_TRY(dev_malloc(&nl, new_live_cap * sizeof(uint32_t)));
    HIP_TRY(dev_malloc(&nd, new_live_cap * sizeof(uint32_t)));
    uint64_t *nh = nullptr;
    if (s->p.k > 32) {
        HIP_TRY(dev_malloc(&nh, new_cap * sizeof(uint64_t)));
        HIP_TRY(hipMemsetAsync(nh, 0xFF, new_cap * sizeof(uint64_t), s->stream)); // EMPTY64 = not written
    }
    HIP_TRY(launch_fill_table(nt, new_cap, s->stream));
    HIP_TRY(launch_rehash(s->table, s->live, s->last_live, nt, (uint32_t)new_cap, nl, s->ctl, s->kmer_hi, nh, s->stream));
    uint32_t zero = 0;
    HIP_TRY(hipMemcpyAsync(&s->ctl->n_dead, &zero, 4, hipMemcpyHostToDevice, s->stream)); // garbage stayed behind
    HIP_TRY(hipStreamSynchronize(s->stream));
    (void)hipFree(s->table);
    (void)hipFree(s->live);
    (void)hipFree(s->dead);
    (void)hipFree(s->kmer_hi);
    s->kmer_hi = nh;
    s->table = nt;
    s->live = nl;
    s->dead = nd;
    s->cap = (uint32_t)new_cap;
    s->live_cap = (uint32_t)new_live_cap;
    s->dead_cap = (uint32_t)new_live_cap;
    if (int rc = alloc_shards(s, s->live_target)) return rc;
    HIP_TRY(launch_set_table(s->ctl, s->table, s->live, s->clog, s->cap, s->live_cap, CLOG_CAP, s->shard_cnt, s->shard_buf,
                             s->shard_cap, s->kmer_hi, s->stream));
    return check_ctl(s);
}

int ensure_out(fh_sketcher *s, uint32_t n) {
    if (n <= s->out_cap) return FH_OK;
    HIP_TRY(hipStreamSynchronize(s->stream));
    (void)hipFree(s->o_block);
    // (the row-gather buffers of fh_copy_out_rows do not depend on out_cap: they live until destroy_handle)
    s->o_block = nullptr;
    s->o_hash = s->o_kmer = s->o_pos = s->o_kmer_hi = nullptr; s->o_count = s->o_extra = nullptr;
    s->out_cap = 0;
    const uint32_t cap = std::max<uint32_t>(n, (uint32_t)SMALL_MAX);
    const bool wide = s->p.k > 32;
    const size_t stride = ((size_t)cap + 2) & ~(size_t)1; // one spare record (the special hash, host side), even
    const size_t bytes = stride * (wide ? 40 : 32);
    HIP_TRY(dev_malloc(&s->o_block, bytes));
    s->o_hash = (uint64_t *)s->o_block;
    s->o_kmer = s->o_hash + stride;
    s->o_pos = s->o_kmer + stride;
    s->o_kmer_hi = wide ? s->o_pos + stride : nullptr;
    s->o_count = (uint32_t *)((wide ? s->o_kmer_hi : s->o_pos) + stride);
    s->o_extra = s->o_count + stride;
    s->out_stride = stride;
    s->o_block_bytes = bytes;
    s->out_cap = cap;
    return FH_OK;
}

uint32_t sat_add(uint32_t a, uint32_t b);

// Large sketches (kmers_to_sketch in the millions: the CLI's oversketch x200) make the per-record host loops of
// fh_copy_out tens of MB of decode work; split it over a few threads (2 M records, k = 31: 7.7 ms with two, 3.9 with
// four, 2.1 with eight, 1.3 with sixteen; the plain gather loop in fh_finish got slower with threads and stays
// inline).  Sketches below 128 k records run inline.
template <class F>
void parallel_for(size_t n, F f) {
    const size_t MIN_PER_THREAD = 1u << 16;
    static const unsigned cap = [] {
        const char *e = cfg("host_threads"); // 1 = always inline
        const unsigned v = e ? (unsigned)atoi(e) : 8u;
        return v ? v : 1u;
    }();
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = std::min<size_t>(std::min<size_t>(hw ? hw : 1u, cap), n / MIN_PER_THREAD);
    if (nt <= 1) {
        f((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    const size_t per = (n + nt - 1) / nt;
    size_t started = 1;
    try {
        for (; started < nt; ++started) {
            const size_t t = started;
            th.emplace_back([=] { f(std::min(n, t * per), std::min(n, (t + 1) * per)); });
        }
    } catch (...) { // no thread to be had (EAGAIN): the rest runs here -- joinable threads must never unwind
    }
    f((size_t)0, std::min(n, per));
    for (size_t t = started; t < nt; ++t) f(std::min(n, t * per), std::min(n, (t + 1) * per));
    for (auto &x : th) x.join();
}

void select_final_p(uint32_t kind, uint64_t size, uint64_t max_hash, std::vector<ResultRec> &v) {
    // v ascending by hash and distinct.  mash.rs:57-60 / scaled.rs:41-58 net effect.
    if (kind == FH_KIND_MASH) {
        if (v.size() > size) v.resize(size);
    } else {
        size_t n_le = 0;
        while (n_le < v.size() && v[n_le].hash <= max_hash) ++n_le;
        size_t keep = std::max<size_t>(n_le, std::min<size_t>(v.size(), size));
        v.resize(keep);
    }
}

void select_final(const fh_sketcher *s, std::vector<ResultRec> &v) { select_final_p(s->p.kind, s->p.size, s->max_hash, v); }

// union of two ascending partial sketches: counts summed (saturating, mash.rs:46-49), k-mer of the smaller
// first position (mash.rs:52-56 keeps the first occurrence's bytes)
int merge_sorted(const std::vector<ResultRec> &a, const std::vector<ResultRec> &b, std::vector<ResultRec> &out) {
    out.clear();
    out.reserve(a.size() + b.size());
    size_t i = 0, j = 0;
    while (i < a.size() || j < b.size()) {
        if (i > 0 && i < a.size() && a[i].hash <= a[i - 1].hash) return fail(FH_ERR_INVALID, "merge: input not ascending");
        if (j > 0 && j < b.size() && b[j].hash <= b[j - 1].hash) return fail(FH_ERR_INVALID, "merge: input not ascending");
        if (j >= b.size() || (i < a.size() && a[i].hash < b[j].hash)) out.push_back(a[i++]);
        else if (i >= a.size() || b[j].hash < a[i].hash) out.push_back(b[j++]);
        else {
            ResultRec r = a[i];
            const ResultRec &d = b[j];
            r.count = sat_add(r.count, d.count);
            r.extra = sat_add(r.extra, d.extra);
            if (d.pos < r.pos) {
                r.pos = d.pos;
                r.kmer = d.kmer;
                r.kmer_hi = d.kmer_hi;
            }
            out.push_back(r);
            ++i;
            ++j;
        }
    }
    return FH_OK;
}

// m-form 2-bit k-mer -> ASCII, four bases per table lookup
struct Ascii4Lut {
    uint32_t v[256];
    Ascii4Lut() {
        for (uint32_t q = 0; q < 256; ++q) v[q] = ascii_group(q, 4);
    }
};
const Ascii4Lut g_ascii4;

// (mhi: the first k - 32 bases of a k-mer longer than 32)
void kmer_ascii(uint64_t m, uint64_t mhi, int k, uint8_t *out) {
    if (k > 32) {
        const int nh = k - 32;
        for (int b = 0; b < nh; ++b) out[b] = (uint8_t) "ACGT"[(mhi >> (2 * (nh - 1 - b))) & 3u];
        out += nh;
        k = 32;
    }
    int b = 0;
    for (; b + 4 <= k; b += 4) {
        const uint32_t w = g_ascii4.v[(m >> (2 * (k - b - 4))) & 0xFFu];
        memcpy(out + b, &w, 4);
    }
    for (; b < k; ++b) out[b] = (uint8_t) "ACGT"[(m >> (2 * (k - 1 - b))) & 3u];
}

uint64_t ascii_kmer(const uint8_t *in, int k, uint64_t *mhi) {
    uint64_t m = 0, h = 0;
    for (int b = 0; b < k; ++b) {
        const uint8_t c = in[b];
        const uint64_t code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
        h = (h << 2) | (m >> 62);
        m = (m << 2) | code;
    }
    *mhi = h;
    return m;
}

ResultRec make_rec(uint64_t hash, uint32_t count, uint32_t extra, const uint8_t *kmer_ascii_bytes, int k, uint64_t pos) {
    ResultRec r{hash, count, extra, 0, pos};
    r.kmer = ascii_kmer(kmer_ascii_bytes, k, &r.kmer_hi);
    return r;
}

uint32_t sat_add(uint32_t a, uint32_t b) {
    const uint32_t r = a + b;
    return r < a ? UINT32_MAX : r;
}

} // namespace

namespace fh { // fh_internal.h
int api_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
hipError_t api_dev_malloc(void **p, size_t bytes) { return dev_malloc(p, bytes); }
hipError_t api_host_malloc(void **p, size_t bytes) { return host_malloc(p, bytes); }
void api_kmer_ascii(uint64_t m, uint64_t mhi, int k, uint8_t *out) { kmer_ascii(m, mhi, k, out); }
} // namespace fh

extern "C" {

int fh_set_option(const char *name, const char *value) {
    if (!name) return fail(FH_ERR_INVALID, "null option name");
    if (cfg_assign(name, value) != 0) return fail(FH_ERR_INVALID, "no such option: %s (fh_option_list names them)", name);
    return FH_OK;
}
const char *fh_get_option(const char *name) { return name && cfg_known(name) ? cfg(name) : nullptr; }
const char *fh_option_list(void) { return cfg_list(); }

int fh_abi_version(void) { return FH_ABI_VERSION; } // (include/finch_hip.h says what each version added)

const char *fh_last_error(void) { return g_err.c_str(); }

int fh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- handle cache ----
// finch creates one sketcher per file and drops it after to_vec (lib.rs:58-79), and a sketcher here owns gigabytes of
// table plus pinned staging memory: creating and freeing one costs ~5 ms, a 5 Mb genome ~1 ms to sketch.  fh_free
// therefore resets the handle and parks it; the next fh_new with the same parameters on the same device takes it
// over.  At most pool_max() handles (FH_POOL, default 64; 0 = off) holding at most pool_max_bytes() (FH_POOL_BYTES,
// default 24 GiB) together stay parked; fh_release_cached frees them, and so does any allocation of the library that
// would otherwise run out of memory.
namespace {
std::mutex g_pool_mu;
std::vector<fh_sketcher *> g_pool;
size_t pool_max() {
    static const size_t v = [] {
        const char *e = cfg("pool");
        return e ? (size_t)strtoul(e, nullptr, 10) : (size_t)64;
    }();
    return v;
}
// device memory the parked handles may hold together (option pool_bytes; default 24 GiB or a tenth of the device, whichever is
// less: room for the sixteen worker sketchers
// per GPU of a finch_sketch_files batch, small next to 288 GB, and given back the moment any allocation of the library
// would otherwise fail -- dev_malloc above; an embedding process that wants it all back calls fh_release_cached)
uint64_t pool_max_bytes() {
    static const uint64_t v = [] {
        if (const char *e = cfg("pool_bytes")) return (uint64_t)strtoull(e, nullptr, 10);
        // (a tenth of the current device at most: an embedding process -- PyTorch, RCCL in bench.py -- shares the device with
        // an allocator that cannot ask this library to let go)
        size_t free_b = 0, total_b = 0;
        uint64_t cap = 24ull << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) cap = std::min<uint64_t>(cap, (uint64_t)total_b / 10);
        else (void)hipGetLastError();
        return cap;
    }();
    return v;
}
uint64_t handle_bytes(const fh_sketcher *s) {
    uint64_t b = (uint64_t)s->cap * (sizeof(Entry) + (s->kmer_hi ? 8 : 0)) + (uint64_t)s->live_cap * 8 + (uint64_t)s->shard_cap * N_SHARDS * 4;
    for (int i = 0; i < N_STAGE; ++i) b += 2 * s->stage_cap[i];
    if (s->bz_text_cap) b += s->bz_comp_cap + 4 * s->bz_text_cap + s->bz_text_cap / 2;
    b += 2 * s->gz_sym_elems + (uint64_t)s->gz_chunks_cap * (GZ_WINDOW + 48) + (s->gz_summary ? (uint64_t)GZ_GROUPS * GZ_WINDOW * 3 : 0);
    return b + (uint64_t)s->out_cap * 32 + (uint64_t)s->big_cap * 24;
}
bool same_params(const fh_params &a, const fh_params &b) {
    return a.kind == b.kind && a.k == b.k && a.size == b.size && a.seed == b.seed && memcmp(&a.scale, &b.scale, sizeof(double)) == 0 &&
           a.max_launch == b.max_launch && a.hash_mask == b.hash_mask && a.stage_bytes == b.stage_bytes;
}
void destroy_handle(fh_sketcher *s);
} // namespace
static void free_gzip_buffers(fh_sketcher *s);

void fh_release_cached(void) {
    std::vector<fh_sketcher *> victims;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        victims.swap(g_pool);
    }
    for (fh_sketcher *s : victims) destroy_handle(s);
    fh::batch_release_cached(); // (parked batch handles, fh_batch.hip)
}

static fh_sketcher *new_handle(const fh_params *params, int device);

fh_sketcher *fh_new(const fh_params *params, int device) {
    if (params && pool_max()) {
        std::lock_guard<std::mutex> g(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); ++i) {
            fh_sketcher *s = g_pool[i];
            if (s->device == device && same_params(s->p, *params)) {
                g_pool.erase(g_pool.begin() + (long)i);
                // fh_free left it reset; only the per-handle statistics are still the previous owner's
                s->n_launches = s->n_relaunches = s->n_big_prunes = 0;
                s->n_spec = s->n_spec_fallback = s->n_sampled = s->n_spec_rescaled = 0;
                s->profiling = false;
                // the environment knobs a handle reads at creation are the new owner's to set
                s->no_spec = cfg("no_spec") != nullptr;
                s->n_fast_finish = s->n_spec_deferred = s->n_spec_recovered = 0;
                s->n_seg_launches = s->n_seg_probes = 0;
                s->seg_hint = 0;
                s->probe_seen = false; // (the previous owner's reads say nothing about the new one's)
                s->probe_answer = 0;
                s->probe_breakers = 0;
                s->gz_no_feed = false; // (nor does a feed that timed out on its files)
                {
                    const bool fast = !s->big_mode && cfg("no_fast") == nullptr;
                    const bool hist = fast && s->p.size > 0 && cfg("no_hist") == nullptr;
                    if (fast != s->fast || hist != s->hist) { // (the control block was initialised for the previous owner's setting)
                        s->fast = fast;
                        s->hist = hist;
                        if (hipSetDevice(s->device) != hipSuccess || init_state(s) != FH_OK) {
                            destroy_handle(s);
                            break; // (a fresh handle below)
                        }
                    }
                }
                {
                    const char *mr = cfg("max_range");
                    s->max_range = mr ? strtoull(mr, nullptr, 10) : 0;
                }
                return s;
            }
        }
    }
    fh_sketcher *s = new_handle(params, device);
    if (!s && pool_max()) { // the parked handles may be what exhausted the device memory
        bool any;
        {
            std::lock_guard<std::mutex> g(g_pool_mu);
            any = !g_pool.empty();
        }
        if (any) {
            fh_release_cached();
            s = new_handle(params, device);
        }
    }
    return s;
}

static fh_sketcher *new_handle(const fh_params *params, int device) {
    if (!params) {
        fail(FH_ERR_INVALID, "params is NULL");
        return nullptr;
    }
    if (params->k < 1 || params->k > (uint32_t)FH_MAX_K) {
        fail(FH_ERR_UNSUPPORTED, "kmer_length %u outside the device range 1..%d", params->k, FH_MAX_K);
        return nullptr;
    }
    if (params->kind != FH_KIND_MASH && params->kind != FH_KIND_SCALED) {
        fail(FH_ERR_INVALID, "unknown sketch kind %u", params->kind);
        return nullptr;
    }
    if (params->kind == FH_KIND_SCALED && !(params->scale > 0.0 && params->scale <= 1.0)) {
        fail(FH_ERR_INVALID, "scale must be in (0, 1]");
        return nullptr;
    }
    int ndev = fh_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) {
        fail(FH_ERR_NO_DEVICE, "no usable HIP device (requested %d of %d)", device, ndev);
        return nullptr;
    }
    fh_sketcher *s = new fh_sketcher();
    s->p = *params;
    s->device = device;
    s->max_hash = params->kind == FH_KIND_SCALED ? scaled_max_hash(params->scale) : 0;
    s->stage_bytes = STAGE_BYTES_ENV ? STAGE_BYTES_ENV : (params->stage_bytes >= 4096 ? params->stage_bytes : STAGE_BYTES_DEFAULT);
    s->max_launch = params->max_launch ? params->max_launch : DEFAULT_MAX_LAUNCH;
    s->max_launch = std::max<uint64_t>((s->max_launch / TILE_POS) * TILE_POS, FIRST_LAUNCH);
    if (s->max_launch > (1ull << 30)) s->max_launch = 1ull << 30;
    // prune when the live list has doubled (or is about to outgrow the in-LDS sort)
    {
        uint64_t n = params->size;
        uint64_t t = std::min<uint64_t>(2 * n, (n + SMALL_MAX) / 2);
        if (params->kind == FH_KIND_SCALED) t = std::min<uint64_t>(t, SMALL_MAX / 2);
        s->trigger = (uint32_t)std::min<uint64_t>(t, SMALL_MAX - 1);
    }
    auto bail = [&](const char *what, hipError_t e) -> fh_sketcher * {
        fail(FH_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        destroy_handle(s);
        return nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    s->big_mode = params->kind == FH_KIND_SCALED || params->size > SMALL_N_MAX;
    s->fast = !s->big_mode && cfg("no_fast") == nullptr;
    s->hist = s->fast && params->size > 0 && cfg("no_hist") == nullptr;
    s->live_target = s->big_mode ? std::max<uint64_t>(4 * params->size, 1ull << 16) : (uint64_t)SMALL_MAX;
    // the table can never fill: waves stop pulling work at soft_limit (<= live_target) and each of the
    // max_waves resident waves can insert at most WAVE_OVERSHOOT (a tile's or a long round's positions) after that
    {
        static const uint64_t waves_per_cu = [] {
            const char *e = cfg("waves_per_cu"); // tuning knob
            return e ? (uint64_t)atoi(e) : 16ull; // what is resident at 4 waves per SIMD; 32 measured 1.4 % slower
        }();
        s->max_waves = std::max<uint64_t>(1, std::min<uint64_t>(256ull * waves_per_cu, s->max_launch / TILE_POS));
        // whole workgroups: a launch is rounded up to them (16 waves for K = 25..32, fh_k2.hip), and everything sized by
        // max_waves -- table, shard lists, the leftover lists a stopped launch writes one entry per wave into -- has to hold
        // what really runs
        // (... and the segment kernels' workgroups, which a block of equal reads brings in whatever K: sixteen waves for K <= 32)
        const uint64_t wpb = std::max<uint64_t>((uint64_t)k2_waves_per_block((int)params->k), (uint64_t)seg_waves_per_block((int)params->k));
        s->max_waves = std::max<uint64_t>(wpb, s->max_waves / wpb * wpb);
        const char *mr = cfg("max_range"); // test knob: force many ranges per push
        s->max_range = mr ? strtoull(mr, nullptr, 10) : 0;
    }
    const uint64_t live_cap = s->live_target + s->max_waves * (uint64_t)(WAVE_BUDGET + WAVE_OVERSHOOT) + 4096;
    const uint64_t cap = 2 * live_cap;
    if (cap >= (1ull << 32)) {
        fail(FH_ERR_INVALID, "max_launch too large");
        destroy_handle(s);
        return nullptr;
    }
    s->cap = (uint32_t)cap;
    s->live_cap = (uint32_t)live_cap;
    if ((e = hipMalloc(&s->table, cap * sizeof(Entry))) != hipSuccess) return bail("hipMalloc(table)", e);
    if (params->k > 32) { // two-word k-mers: the high words live beside the table (fh_device.h, Ctl::kmer_hi)
        if ((e = hipMalloc(&s->kmer_hi, cap * sizeof(uint64_t))) != hipSuccess) return bail("hipMalloc(kmer_hi)", e);
        if ((e = hipMemsetAsync(s->kmer_hi, 0xFF, cap * sizeof(uint64_t), s->stream)) != hipSuccess) return bail("hipMemset(kmer_hi)", e);
    }
    if ((e = hipMalloc(&s->live, live_cap * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(live)", e);
    s->dead_cap = (uint32_t)live_cap;
    if ((e = hipMalloc(&s->dead, (size_t)s->dead_cap * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(dead)", e);
    if ((e = hipMalloc(&s->ctl, sizeof(Ctl))) != hipSuccess) return bail("hipMalloc(ctl)", e);
    for (int i = 0; i < 2; ++i)
        if ((e = hipMalloc(&s->left_buf[i], (size_t)s->max_waves * 4 * sizeof(uint32_t) + 64)) != hipSuccess)
            return bail("hipMalloc(left)", e);
    if ((e = hipMalloc(&s->clog, CLOG_CAP * sizeof(CollRec))) != hipSuccess) return bail("hipMalloc(clog)", e);
    if ((e = hipHostMalloc(&s->h_ctl, sizeof(Ctl), hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    s->no_spec = cfg("no_spec") != nullptr;

    if ((e = launch_fill_table(s->table, cap, s->stream)) != hipSuccess) return bail("fill_table", e);
    if (alloc_shards(s, s->live_target) != FH_OK) {
        destroy_handle(s);
        return nullptr;
    }
    if ((e = launch_set_table(s->ctl, s->table, s->live, s->clog, s->cap, s->live_cap, CLOG_CAP, s->shard_cnt, s->shard_buf,
                              s->shard_cap, s->kmer_hi, s->stream)) != hipSuccess)
        return bail("set_table", e);
    if (init_state(s) != FH_OK) {
        destroy_handle(s);
        return nullptr;
    }
    if ((e = hipStreamSynchronize(s->stream)) != hipSuccess) return bail("hipStreamSynchronize", e);
    return s;
}

void fh_free(fh_sketcher *s) {
    if (!s) return;
    if (pool_max() && s->ctl && fh_reset(s) == FH_OK) {
        std::lock_guard<std::mutex> g(g_pool_mu);
        uint64_t held = handle_bytes(s);
        for (const fh_sketcher *q : g_pool) held += handle_bytes(q);
        if (g_pool.size() < pool_max() && held <= pool_max_bytes()) {
            g_pool.push_back(s);
            return;
        }
    }
    destroy_handle(s);
}

namespace {
void destroy_handle(fh_sketcher *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto &pe : s->prof_events) {
        (void)hipEventDestroy(pe.first);
        (void)hipEventDestroy(pe.second);
    }
    for (int i = 0; i < N_STAGE; ++i) {
        if (s->h_stage[i]) (void)hipHostFree(s->h_stage[i]);
        if (s->d_stage[i]) (void)hipFree(s->d_stage[i]);
        if (s->stage_done[i]) (void)hipEventDestroy(s->stage_done[i]);
        (void)hipFree(s->d_packed[i]);
        (void)hipFree(s->d_blk_a[i]);
        (void)hipFree(s->d_blk_b[i]);
    }
    (void)hipFree(s->d_text_tot);
    (void)hipFree(s->d_lines);
    for (int q = 0; q < fh_sketcher::BZ_STREAMS; ++q) {
        if (s->bz_stream[q]) {
            (void)hipStreamSynchronize(s->bz_stream[q]);
            (void)hipStreamDestroy(s->bz_stream[q]);
        }
        if (s->bz_done[q]) (void)hipEventDestroy(s->bz_done[q]);
    }
    (void)hipFree(s->d_comp);
    (void)hipFree(s->d_bz_members);
    (void)hipFree(s->d_bz_status);
    (void)hipFree(s->bz_lines);
    gzip_quiesce(s);
    free_gzip_buffers(s);
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(s->bz_text[i]);
        (void)hipFree(s->bz_packed[i]);
        (void)hipFree(s->bz_blk_a[i]);
        (void)hipFree(s->bz_blk_b[i]);
    }
    if (s->h_bz_status) (void)hipHostFree(s->h_bz_status);
    if (s->h_text_tot) (void)hipHostFree(s->h_text_tot);
    if (s->h_ctl) (void)hipHostFree(s->h_ctl);
    if (s->h_out) (void)hipHostFree(s->h_out);
    (void)hipFree(s->left_buf[0]);
    (void)hipFree(s->left_buf[1]);
    (void)hipFree(s->keys_a);
    (void)hipFree(s->keys_b);
    (void)hipFree(s->slots_a);
    (void)hipFree(s->slots_b);
    (void)hipFree(s->sort_tmp);
    (void)hipFree(s->keep_dev);
    (void)hipFree(s->table);
    (void)hipFree(s->live);
    (void)hipFree(s->shard_cnt);
    (void)hipFree(s->shard_buf);
    (void)hipFree(s->dead);
    (void)hipFree(s->ctl);
    (void)hipFree(s->clog);
    (void)hipFree(s->o_block);
    (void)hipFree(s->d_rows);
    (void)hipFree(s->d_rows_out);
    if (s->h_rows) (void)hipHostFree(s->h_rows);
    if (s->h_rows_out) (void)hipHostFree(s->h_rows_out);
    (void)hipFree(s->kmer_hi);
    (void)hipFree(s->smp_list);
    (void)hipFree(s->smp_hist);
    if (s->h_smp_hist) (void)hipHostFree(s->h_smp_hist);
    if (s->h_probe) (void)hipHostFree(s->h_probe);
    if (s->probe_ev) (void)hipEventDestroy(s->probe_ev);
    if (s->copy_stream) {
        (void)hipStreamSynchronize(s->copy_stream);
        (void)hipStreamDestroy(s->copy_stream);
    }
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}
} // namespace

int fh_reset(fh_sketcher *s) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (int rc = set_device(s)) return rc;
    gzip_quiesce(s); // (a launch that waits for the rest of an abandoned batch holds the stream until it is told to give up)
    if (s->device_clean && s->finished && !s->pend.active && !s->epi_pending) { // fh_finish's epilogue has done the device side
        s->device_clean = false;
        return init_state(s, false);
    }
    s->device_clean = false;
    if (s->spec.pending && !s->pend.active) s->spec.pending = false; // (nobody will ask for its outcome; the clear below is stream-ordered behind it)
    if (int rc = drain(s)) return rc;
    if (int rc = flush_epilogue(s)) return rc; // (it also rewinds the shard cursors of the last launch)
    if (s->dirty) {
        // clear only the slots this run touched (live + dropped); fall back to a full refill if the
        // dropped-slot list overflowed
        if (s->fast && s->finished && s->h_ctl->n_dead != 0xFFFFFFFFu && (uint64_t)s->h_ctl->n_live + s->h_ctl->n_dead <= 65536u) {
            // (a finished sketch: the host has its final control block, and the two lists are short -- one
            // single-workgroup launch clears them and re-initialises the control block)
            HIP_TRY(launch_reset_small(s->table, s->live, s->dead, s->ctl, initial_tau(s), s->p.size, 0ull, s->hist ? 1u : 0u, s->stream));
            return init_state(s, false);
        }
        HIP_TRY(launch_clear_slots(s->table, s->cap, s->live, s->dead, s->ctl, s->stream));
    }
    return init_state(s);
}

int fh_set_stream_offset(fh_sketcher *s, uint64_t offset) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    s->stream_off = offset;
    return FH_OK;
}

// ---- SketchScheme::process itself (mash.rs:67-80), one record per call ----
// The record's bytes go straight into the pinned staging buffer, blanks dropped on the way (the ONE host-side copy), one
// breaker byte behind them; a full buffer is committed as fh_push_staged does.  A binding at the trait level is this call
// and nothing else per record -- no block of its own to append to and hand over (a second copy of every base).
static int text_buffer_impl(fh_sketcher *s, uint8_t **buf, uint64_t *cap);
static int proc_acquire(fh_sketcher *s) {
    if (s->proc_buf) return FH_OK;
    uint8_t *buf = nullptr;
    uint64_t cap = 0;
    if (int rc = text_buffer_impl(s, &buf, &cap)) return rc;
    s->proc_buf = buf;
    s->proc_cap = cap;
    s->proc_fill = 0;
    return FH_OK;
}
static int proc_commit(fh_sketcher *s) {
    uint8_t *const buf = s->proc_buf;
    s->proc_buf = nullptr; // (fh_push_staged hands the slot on; whatever happens, this block is done with)
    if (!buf || s->proc_fill == 0) return FH_OK;
    const uint64_t n = s->proc_fill;
    s->proc_fill = 0;
    const int rc = fh_push_staged(s, n, s->proc_continuing ? FH_PUSH_CONTINUE : 0u);
    s->proc_continuing = s->proc_in_record;
    return rc;
}
// records written by fh_process that no push has committed yet: every other way of feeding the sketcher, and everything that
// looks at the result, commits them first
static int proc_flush(fh_sketcher *s) { return s->proc_buf ? proc_commit(s) : FH_OK; }

// The common case of fh_process in one piece: a staging buffer is open and has room for the whole record, its breaker and the
// copy's slack.  (A read is 100-250 bytes and the call is made once per read, on one thread: mash.rs:67-80 -- at 16 GB/s a
// record has 9 ns, of which the calls, checks and the loop of the general path below were two.)
#if defined(__x86_64__)
__attribute__((target("avx2"))) static inline bool process_fast(fh_sketcher *s, const uint8_t *seq, uint64_t len) {
    uint8_t *const b = s->proc_buf + s->proc_fill;
    const size_t m = fh_strip::strip_avx2(b, seq, (size_t)len);
    b[m] = 0; // the breaker: k-mers never span records
    s->proc_fill += m + 1;
    s->proc_total_bases += len; // mash.rs:72: the raw sequence() slice, blanks included
    return true;
}
static const bool g_have_avx2 = __builtin_cpu_supports("avx2");
#define FH_PROCESS_FAST(s, seq, len) \
    (g_have_avx2 && (s)->proc_buf && !(s)->finished && (s)->proc_cap - (s)->proc_fill > (len) + 130 && process_fast((s), (seq), (len)))
#else
#define FH_PROCESS_FAST(s, seq, len) false
#endif

int fh_process(fh_sketcher *s, const uint8_t *seq, uint64_t len) {
    if (!s || (!seq && len)) return fail(FH_ERR_INVALID, "null argument");
    if (FH_PROCESS_FAST(s, seq, len)) return FH_OK;
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    s->proc_total_bases += len; // mash.rs:72: the raw sequence() slice, blanks included
    s->proc_in_record = true;
    while (len) {
        if (int rc = proc_acquire(s)) return rc;
        const uint64_t room = s->proc_cap - s->proc_fill;
        // (the strip stores whole vectors: 32 bytes of slack behind what it keeps.)  A record that does not fit what is left
        // of the block but would fit an empty one starts the next block: blocks then hold whole records -- no k-mer spans
        // them, and a block of equal reads is one the stride probe recognises (fh_k2s.hip)
        if (room <= 64 || (room < len + 130 && !s->proc_in_cut && s->proc_fill > 0 && len + 130 <= s->proc_cap)) {
            s->proc_in_record = s->proc_in_cut; // (the block ends inside a record only if bytes of this one are in it)
            const int rc = proc_commit(s);
            s->proc_in_record = true;
            if (rc) return rc;
            continue;
        }
        s->proc_in_cut = true; // (bytes of this record are in the block: from here on it is cut where the block ends)
        const uint64_t take = std::min<uint64_t>(len, room - 64);
        s->proc_fill += fh_strip::strip(s->proc_buf + s->proc_fill, seq, take);
        seq += take;
        len -= take;
    }
    if (int rc = proc_acquire(s)) return rc;
    s->proc_buf[s->proc_fill++] = 0; // the breaker: k-mers never span records
    s->proc_in_record = false;
    s->proc_in_cut = false;
    if (s->proc_cap - s->proc_fill <= 64) return proc_commit(s);
    return FH_OK;
}

int fh_process_records(fh_sketcher *s, const uint8_t *base, const uint64_t *offsets, const uint64_t *lens, uint64_t n) {
    if (!s || (n && (!base || !offsets || !lens))) return fail(FH_ERR_INVALID, "null argument");
    for (uint64_t i = 0; i < n; ++i) {
        if (FH_PROCESS_FAST(s, base + offsets[i], lens[i])) continue;
        if (int rc = fh_process(s, base + offsets[i], lens[i])) return rc;
    }
    return FH_OK;
}

int fh_process_records_in(fh_sketcher *s, const uint8_t *base, uint64_t base_len, const uint64_t *offsets, const uint64_t *lens,
                          uint64_t n, uint64_t *bases) {
    if (!s || (n && (!base || !offsets || !lens))) return fail(FH_ERR_INVALID, "null argument");
    uint64_t sum = 0;
    int rc = FH_OK;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t o = offsets[i], l = lens[i];
        if (o > base_len || l > base_len - o) { // (no sum of the two: it wraps)
            char msg[96];
            snprintf(msg, sizeof msg, "record %llu lies outside the buffer", (unsigned long long)i);
            rc = fail(FH_ERR_INVALID, msg);
            break;
        }
        if (!FH_PROCESS_FAST(s, base + o, l) && (rc = fh_process(s, base + o, l))) break;
        sum += l;
    }
    if (bases) *bases = sum;
    return rc;
}

int fh_total_bases(fh_sketcher *s, uint64_t *total_bases) {
    if (!s || !total_bases) return fail(FH_ERR_INVALID, "null argument");
    *total_bases = s->proc_total_bases;
    return FH_OK;
}

int fh_set_record_stride(fh_sketcher *s, uint32_t stride) {
    if (!s) return fail(FH_ERR_INVALID, "null argument");
    if (stride > 1u && (stride < SEG_MIN_STRIDE || stride > (s->p.k > 32 ? SEG_MAX_STRIDE : SEG_MAX_RECORD))) stride = 1u; // (nothing the segment kernel takes)
    s->seg_hint = stride;
    return FH_OK;
}

int fh_debug_segments(fh_sketcher *s, uint64_t *launches, uint64_t *probes, uint32_t *stride) {
    if (!s) return fail(FH_ERR_INVALID, "null argument");
    if (launches) *launches = s->n_seg_launches;
    if (probes) *probes = s->n_seg_probes;
    if (stride) *stride = s->blk_seg;
    return FH_OK;
}

int fh_push_device(fh_sketcher *s, const void *dev_bytes, uint64_t len) {
    if (!s || (!dev_bytes && len)) return fail(FH_ERR_INVALID, "null argument");
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    if (int rc = proc_flush(s)) return rc;
    if (((uintptr_t)dev_bytes & 15u) != 0) return fail(FH_ERR_INVALID, "device block must be 16-byte aligned");
    if (int rc = set_device(s)) return rc;
    s->carry_len = 0;
    s->dprev_len = 0;
    int rc = sketch_device_range(s, (const uint8_t *)dev_bytes, len, s->stream_off);
    s->stream_off += len;
    return rc;
}

// copy src[0, n) to dst dropping ' ', '\t', '\r', '\n' (what normalize(false) removes: fh_strip.h); dst has room for n + 32
// bytes.  A block of megabytes is split over a few threads: a first pass counts what every share keeps, the second copies
// every share to its place (one thread strips ~5 GB/s, a third of what the copy to the device behind it moves).
static uint64_t strip_block(uint8_t *dst, const uint8_t *src, uint64_t n) {
    static const unsigned cap = [] {
        const char *e = cfg("host_threads"); // 1 = always inline
        const unsigned v = e ? (unsigned)atoi(e) : 8u;
        return v ? v : 1u;
    }();
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nt = std::min<size_t>(std::min<size_t>(hw, cap), n >> 21);
    if (nt <= 1) return fh_strip::strip(dst, src, n);
    const size_t per = (n + nt - 1) / nt;
    std::vector<size_t> kept(nt + 1, 0);
    auto lo_of = [&](size_t t) { return std::min<size_t>(n, t * per); };
    {
        std::vector<std::thread> th;
        th.reserve(nt);
        size_t started = 0;
        try {
            for (; started < nt - 1; ++started) {
                const size_t t = started + 1;
                th.emplace_back([&, t] { kept[t + 1] = fh_strip::count_kept(src + lo_of(t), lo_of(t + 1) - lo_of(t)); });
            }
        } catch (...) {
        }
        kept[1] = fh_strip::count_kept(src, lo_of(1));
        for (size_t t = started + 1; t < nt; ++t) kept[t + 1] = fh_strip::count_kept(src + lo_of(t), lo_of(t + 1) - lo_of(t));
        for (auto &x : th) x.join();
    }
    for (size_t t = 0; t < nt; ++t) kept[t + 1] += kept[t]; // -> where share t's bytes begin
    {
        // (the exact scalar form: a share must not write a byte behind what it keeps -- the next share's bytes begin there)
        std::vector<std::thread> th;
        th.reserve(nt);
        size_t started = 0;
        try {
            for (; started < nt - 1; ++started) {
                const size_t t = started + 1;
                th.emplace_back([&, t] { (void)fh_strip::strip_scalar(dst + kept[t], src + lo_of(t), lo_of(t + 1) - lo_of(t)); });
            }
        } catch (...) {
        }
        (void)fh_strip::strip_scalar(dst, src, lo_of(1));
        for (size_t t = started + 1; t < nt; ++t) (void)fh_strip::strip_scalar(dst + kept[t], src + lo_of(t), lo_of(t + 1) - lo_of(t));
        for (auto &x : th) x.join();
    }
    return kept[nt];
}

static int ensure_stage(fh_sketcher *s);
static int ensure_slot(fh_sketcher *s, int i, uint64_t want);

int fh_push_block_ex(fh_sketcher *s, const uint8_t *bytes, uint64_t len, uint32_t flags) {
    if (!s || (!bytes && len)) return fail(FH_ERR_INVALID, "null argument");
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    if (int rc = proc_flush(s)) return rc;
    if (int rc = set_device(s)) return rc;
    // normalize(false) drops whitespace (needletail; mash.rs:73): strip it while staging so that device
    // positions are contiguous.  k-mers may span staging slices of one block: carry K-1 bytes over.
    const uint32_t K = s->p.k;
    if (!(flags & FH_PUSH_CONTINUE)) s->carry_len = 0;
    s->dprev_len = 0; // a device-side FASTA chunk chain ends here
    uint64_t in = 0;
    while (in < len) {
        const int b = s->stage_next;
        const uint32_t carry_len = s->carry_len;
        if (int rc = ensure_slot(s, b, len - in + carry_len + 64)) return rc;
        if (s->stage_busy[b]) {
            HIP_TRY(hipEventSynchronize(s->stage_done[b]));
            s->stage_busy[b] = false;
        }
        uint8_t *dst = s->h_stage[b];
        memcpy(dst, s->carry, carry_len);
        // (the slot was allocated 128 bytes larger than its capacity: room for the vector stores of the strip)
        const uint64_t consumed = std::min<uint64_t>(len - in, s->stage_cap[b] - carry_len);
        const uint64_t fresh = strip_block(dst + carry_len, bytes + in, consumed);
        in += consumed;
        const uint64_t m = carry_len + fresh;
        const uint64_t base = s->stream_off - carry_len;
        if (fresh) {
            HIP_TRY(hipMemcpyAsync(s->d_stage[b], dst, m, hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipEventRecord(s->stage_done[b], s->stream));
            s->stage_busy[b] = true;
            if (int rc = sketch_device_range(s, s->d_stage[b], m, base)) return rc;
            s->stream_off += fresh;
            // the next H2D into d_stage[b] is stream-ordered after these kernels
            s->carry_len = (uint32_t)std::min<uint64_t>(K - 1, m);
            memcpy(s->carry, dst + m - s->carry_len, s->carry_len);
            s->stage_next = (b + 1) % N_STAGE;
        }
    }
    return FH_OK;
}

// Staging slot i able to hold `want` bytes.  Pinning memory is slow (2 x 64 MiB: ~40 ms), so a slot is only as large
// as the pushes it has served: a sketcher that sees one 5 MB record block never pins more than that.
static int ensure_slot(fh_sketcher *s, int i, uint64_t want) {
    want = std::min<uint64_t>(std::max<uint64_t>(want, 1ull << 20), s->stage_bytes);
    if (!s->stage_done[i]) HIP_TRY(hipEventCreateWithFlags(&s->stage_done[i], hipEventDisableTiming));
    if (s->stage_cap[i] >= want) return FH_OK;
    if (s->stage_cap[i]) {
        const uint64_t old_cap = s->stage_cap[i];
        // growing: whatever still reads the old buffers has to finish first
        if (int rc = drain(s)) return rc;
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->stage_busy[i] = false;
        (void)hipHostFree(s->h_stage[i]);
        (void)hipFree(s->d_stage[i]);
        s->h_stage[i] = s->d_stage[i] = nullptr;
        s->stage_cap[i] = 0;
        want = std::min<uint64_t>(std::max<uint64_t>(want, 2 * old_cap), s->stage_bytes); // grow geometrically
    }
    HIP_TRY(host_malloc((void **)&s->h_stage[i], want + 128));
    HIP_TRY(dev_malloc((void **)&s->d_stage[i], want + 128));
    s->stage_cap[i] = want;
    return FH_OK;
}

// the zero-copy paths hand the caller a buffer of the full staging size
static int ensure_stage(fh_sketcher *s) {
    for (int i = 0; i < N_STAGE; ++i)
        if (int rc = ensure_slot(s, i, s->stage_bytes)) return rc;
    return FH_OK;
}

static int text_buffer_impl(fh_sketcher *s, uint8_t **buf, uint64_t *cap);
int fh_text_buffer(fh_sketcher *s, uint8_t **buf, uint64_t *cap) {
    if (!s || !buf || !cap) return fail(FH_ERR_INVALID, "null argument");
    if (int rc = proc_flush(s)) return rc; // (records of fh_process sit in the very buffer this hands out)
    return text_buffer_impl(s, buf, cap);
}
static int text_buffer_impl(fh_sketcher *s, uint8_t **buf, uint64_t *cap) {
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_stage(s)) return rc;
    const int b = s->stage_next;
    if (s->stage_busy[b] || s->stage_prefetched[b]) {
        HIP_TRY(hipEventSynchronize(s->stage_done[b]));
        s->stage_busy[b] = false;
        s->stage_prefetched[b] = 0;
    }
    *buf = s->h_stage[b] + STAGE_HEADROOM; // room in front for the K-1 carry bytes of fh_push_staged
    *cap = s->stage_bytes;
    return FH_OK;
}

int fh_text_buffers(fh_sketcher *s, uint8_t *bufs[2], uint64_t *cap, int *next) {
    if (!s || !bufs || !cap || !next) return fail(FH_ERR_INVALID, "null argument");
    if (int rc = proc_flush(s)) return rc;
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_stage(s)) return rc;
    for (int i = 0; i < N_STAGE; ++i) {
        if (s->stage_busy[i] || s->stage_prefetched[i]) { // (a prefetch nobody consumed: an aborted stream)
            HIP_TRY(hipEventSynchronize(s->stage_done[i]));
            s->stage_busy[i] = false;
            s->stage_prefetched[i] = 0;
        }
        bufs[i] = s->h_stage[i] + STAGE_HEADROOM;
    }
    *cap = s->stage_bytes;
    *next = s->stage_next;
    return FH_OK;
}

int fh_push_staged(fh_sketcher *s, uint64_t len, uint32_t flags) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    if (len > s->stage_bytes) return fail(FH_ERR_INVALID, "block longer than the staging buffer");
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_stage(s)) return rc;
    if (!(flags & FH_PUSH_CONTINUE)) s->carry_len = 0;
    s->dprev_len = 0; // a device-side FASTA chunk chain ends here
    if (len == 0) return FH_OK;
    const int b = s->stage_next;
    const uint32_t K = s->p.k;
    const uint32_t carry_len = s->carry_len;
    static_assert(STAGE_HEADROOM >= sizeof(fh_sketcher::carry) - 1, "the carried bytes must fit in front of the staged data");
    uint8_t *src = s->h_stage[b] + STAGE_HEADROOM - carry_len;
    memcpy(src, s->carry, carry_len);
    const uint64_t m = carry_len + len;
    const uint64_t base = s->stream_off - carry_len;
    // the device buffer of this slot may still feed a pending range
    if (int rc = drain(s)) return rc;
    if (carry_len == 0 && s->stage_prefetched[b] == len) { // the filler's thread has the copy under way (fh_text_prefetch)
        HIP_TRY(hipStreamWaitEvent(s->stream, s->stage_done[b], 0));
    } else {
        if (s->stage_prefetched[b]) HIP_TRY(hipEventSynchronize(s->stage_done[b])); // (a prefetch of something else: let it land first)
        HIP_TRY(hipMemcpyAsync(s->d_stage[b], src, m, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipEventRecord(s->stage_done[b], s->stream));
    }
    s->stage_prefetched[b] = 0;
    s->stage_busy[b] = true;
    if (int rc = sketch_device_range(s, s->d_stage[b], m, base)) return rc;
    s->stream_off += len;
    s->carry_len = (uint32_t)std::min<uint64_t>(K - 1, m);
    memcpy(s->carry, src + m - s->carry_len, s->carry_len);
    s->stage_next = (b + 1) % N_STAGE;
    return FH_OK;
}

// scratch the device-side FASTQ splitter needs for slot b
static int ensure_fastq_scratch(fh_sketcher *s, int b) {
    const uint64_t nblk = (s->stage_bytes + 4095) / 4096 + 1;
    if (!s->d_packed[b]) {
        HIP_TRY(dev_malloc((void **)&s->d_packed[b], s->stage_bytes + 64));
        HIP_TRY(dev_malloc((void **)&s->d_blk_a[b], 5 * nblk * sizeof(uint32_t))); // (FASTQ: newlines + four class counts per block)
        HIP_TRY(dev_malloc((void **)&s->d_blk_b[b], nblk * sizeof(uint32_t)));
    }
    if (!s->d_text_tot) {
        HIP_TRY(dev_malloc((void **)&s->d_text_tot, 4 * sizeof(uint32_t)));
        HIP_TRY(host_malloc((void **)&s->h_text_tot, 4 * sizeof(uint32_t)));
    }
    if (!s->d_lines) { // a record is four lines; below 8 bytes per line on average the host parser takes the file
        s->line_cap = (uint32_t)std::min<uint64_t>(s->stage_bytes / 8 + 64, 0x7FFFFFFFull);
        HIP_TRY(dev_malloc(&s->d_lines, (size_t)s->line_cap * sizeof(uint32_t)));
    }
    return FH_OK;
}

// text[0, len) holds whole 4-line FASTQ records (or is being filled on the stream): split, check, sketch
static int fastq_text_on_device(fh_sketcher *s, const uint8_t *text, uint64_t len, uint8_t *packed, uint32_t *blk_a, uint32_t *blk_b,
                                uint32_t *lines, uint32_t line_cap) {
    HIP_TRY(hipMemsetAsync(s->d_text_tot, 0, 4 * sizeof(uint32_t), s->stream));
    HIP_TRY(launch_fastq_pack(text, len, packed, blk_a, blk_b, s->d_text_tot, s->ctl, s->d_text_tot + 2, lines, line_cap, s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_text_tot, s->d_text_tot, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->h_text_tot[2])
        return fail(FH_ERR_INVALID, "not plain 4-line FASTQ text (header without '@', separator without '+', blanks inside a "
                                    "sequence line, or sequence and quality lengths differ)");
    const uint64_t n_packed = s->h_text_tot[1];
    s->carry_len = 0; // every sequence line ends with its breaker: nothing spans chunks
    s->dprev_len = 0;
    const int rc = sketch_device_range(s, packed, n_packed, s->stream_off);
    s->stream_off += n_packed;
    return rc;
}

int fh_push_fastq_text(fh_sketcher *s, uint64_t len) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    if (s->proc_buf) return fail(FH_ERR_STATE, "records of fh_process are waiting in the staging buffer: fh_sync first");
    if (len > s->stage_bytes) return fail(FH_ERR_INVALID, "text longer than the staging buffer");
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_stage(s)) return rc;
    s->bgzf_left_len = 0;
    if (len == 0) return FH_OK;
    const int b = s->stage_next;
    if (int rc = ensure_fastq_scratch(s, b)) return rc;
    // the packed buffer of this slot may still feed a pending range
    if (int rc = drain(s)) return rc;
    if (s->stage_prefetched[b] == len) { // the reader's thread has the copy under way (fh_text_prefetch)
        HIP_TRY(hipStreamWaitEvent(s->stream, s->stage_done[b], 0));
    } else {
        if (s->stage_prefetched[b]) HIP_TRY(hipEventSynchronize(s->stage_done[b])); // (a prefetch of something else: let it land first)
        HIP_TRY(hipMemcpyAsync(s->d_stage[b], s->h_stage[b] + STAGE_HEADROOM, len, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipEventRecord(s->stage_done[b], s->stream));
    }
    s->stage_prefetched[b] = 0;
    s->stage_busy[b] = true;
    s->stage_next = (b + 1) % N_STAGE;
    return fastq_text_on_device(s, s->d_stage[b], len, s->d_packed[b], s->d_blk_a[b], s->d_blk_b[b], s->d_lines, s->line_cap);
}

// (called by the reader's thread: touches nothing but slot `slot`'s copy state and the copy stream.  The slot's device
//  buffer is free: the push that last used it returned after its record-splitting kernel had read it.)
int fh_text_prefetch(fh_sketcher *s, int slot, uint64_t len) {
    if (!s || slot < 0 || slot >= N_STAGE) return fail(FH_ERR_INVALID, "bad argument");
    if (len == 0 || len > s->stage_cap[slot] || !s->d_stage[slot] || !s->stage_done[slot] || s->stage_prefetched[slot]) return FH_OK; // the push will copy
    if (int rc = set_device(s)) return rc;
    if (!s->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipMemcpyAsync(s->d_stage[slot], s->h_stage[slot] + STAGE_HEADROOM, len, hipMemcpyHostToDevice, s->copy_stream));
    HIP_TRY(hipEventRecord(s->stage_done[slot], s->copy_stream));
    s->stage_prefetched[slot] = len;
    return FH_OK;
}

static_assert(sizeof(fh_bgzf_member) == sizeof(BgzfMember) && offsetof(fh_bgzf_member, crc32) == offsetof(BgzfMember, crc),
              "the ABI's member record is what the kernels read");
// BGZF members of FASTQ text, inflated on the device (fh_bgzf.hip).  The caller has put, into the text buffer of
// fh_text_buffers, `n_members` fh_bgzf_member records followed by the members' DEFLATE bytes (`bytes` in all; in_off
// counts from the start of the buffer).  The text of a batch rarely ends with a record: what follows its last whole
// record stays on the device and leads the text of the next push; FH_BGZF_LAST says there is no next push.
constexpr uint32_t BZ_MAX_MEMBERS = 1u << 16; // members one launch may take (8 GB of text at bgzip's member size)
// buffers of the device-side BGZF path: text of up to 8 staging buffers' worth per batch (1 GiB at most)
static int ensure_bgzf_buffers(fh_sketcher *s) {
    if (s->bz_text_cap) return FH_OK;
    const uint64_t cap = std::min<uint64_t>(8 * s->stage_bytes, 1ull << 30);
    const uint64_t nblk = (cap + 4095) / 4096 + 1;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(dev_malloc((void **)&s->bz_text[i], cap + 128));
        HIP_TRY(dev_malloc((void **)&s->bz_packed[i], cap + 64));
        HIP_TRY(dev_malloc((void **)&s->bz_blk_a[i], 5 * nblk * sizeof(uint32_t)));
        HIP_TRY(dev_malloc((void **)&s->bz_blk_b[i], nblk * sizeof(uint32_t)));
    }
    s->bz_line_cap = (uint32_t)std::min<uint64_t>(cap / 8 + 64, 0x7FFFFFFFull);
    HIP_TRY(dev_malloc((void **)&s->bz_lines, (size_t)s->bz_line_cap * sizeof(uint32_t)));
    // (DEFLATE never grows data by more than a few bytes per block: a batch's members are about as large as its text at worst)
    s->bz_comp_cap = cap + cap / 64 + s->stage_bytes;
    HIP_TRY(dev_malloc((void **)&s->d_comp, s->bz_comp_cap + 4096));
    HIP_TRY(dev_malloc((void **)&s->d_bz_members, (size_t)BZ_MAX_MEMBERS * sizeof(BgzfMember)));
    HIP_TRY(dev_malloc((void **)&s->d_bz_status, 4 * sizeof(uint32_t)));
    HIP_TRY(host_malloc((void **)&s->h_bz_status, 4 * sizeof(uint32_t)));
    if (!s->d_text_tot) {
        HIP_TRY(dev_malloc((void **)&s->d_text_tot, 4 * sizeof(uint32_t)));
        HIP_TRY(host_malloc((void **)&s->h_text_tot, 4 * sizeof(uint32_t)));
    }
    for (int q = 0; q < fh_sketcher::BZ_STREAMS; ++q) {
        HIP_TRY(hipStreamCreateWithFlags(&s->bz_stream[q], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->bz_done[q], hipEventDisableTiming));
    }
    s->bz_text_cap = cap;
    return FH_OK;
}


int fh_bgzf_text_capacity(fh_sketcher *s, uint64_t *cap) {
    if (!s || !cap) return fail(FH_ERR_INVALID, "null argument");
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_bgzf_buffers(s)) return rc;
    *cap = s->bz_text_cap;
    return FH_OK;
}

int fh_push_bgzf_fastq(fh_sketcher *s, uint64_t bytes, uint32_t n_members, uint32_t flags) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    if (s->proc_buf) return fail(FH_ERR_STATE, "records of fh_process are waiting in the staging buffer: fh_sync first");
    if (bytes > s->stage_bytes) return fail(FH_ERR_INVALID, "batch longer than the staging buffer");
    if ((uint64_t)n_members * sizeof(fh_bgzf_member) > bytes) return fail(FH_ERR_INVALID, "member table longer than the batch");
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_stage(s)) return rc;
    if (int rc = ensure_bgzf_buffers(s)) return rc;
    const int b = s->stage_next, t = s->bz_next;
    fh_bgzf_member *mt = (fh_bgzf_member *)(s->h_stage[b] + STAGE_HEADROOM);
    uint64_t text = 0;
    for (uint32_t i = 0; i < n_members; ++i) {
        const fh_bgzf_member &m = mt[i];
        if (m.in_off < (uint64_t)n_members * sizeof(fh_bgzf_member) || (uint64_t)m.in_off + m.in_len > bytes || m.isize > 65536u ||
            m.out_off != text)
            return fail(FH_ERR_INVALID, "BGZF member %u: bad table entry", i);
        text += m.isize;
    }
    const bool first_of_batch = s->bz_acc_n == 0 && s->bz_acc_bytes == 0;
    const uint64_t left = first_of_batch ? s->bgzf_left_len : s->bz_batch_left;
    if (left + s->bz_acc_text + text > s->bz_text_cap || left + s->bz_acc_text + text >= (1ull << 31))
        return fail(FH_ERR_INVALID, "a FASTQ record and a batch of BGZF text do not fit the text buffer together");
    if (s->bz_acc_bytes + bytes > s->bz_comp_cap || (uint64_t)s->bz_acc_n + n_members > BZ_MAX_MEMBERS)
        return fail(FH_ERR_INVALID, "too many BGZF members collected with FH_BGZF_MORE");
    if (first_of_batch) {
        // the packed buffer of this slot may still feed a pending range
        if (int rc = drain(s)) return rc;
        s->bz_batch_left = left;
        s->bgzf_left_len = 0;
        HIP_TRY(hipMemsetAsync(s->d_bz_status, 0, 4 * sizeof(uint32_t), s->stream));
        if (left) HIP_TRY(hipMemcpyAsync(s->bz_text[t], s->bgzf_left_ptr, left, hipMemcpyDeviceToDevice, s->stream));
    }
    if (n_members) {
        // the members join those of the batch's earlier pushes: their offsets now count from the start of the device-side
        // byte and text areas; they are copied and inflated on a side stream while the caller reads on
        for (uint32_t i = 0; i < n_members; ++i) {
            mt[i].in_off += (uint32_t)s->bz_acc_bytes;
            mt[i].out_off += (uint32_t)s->bz_acc_text;
        }
        const int q = s->bz_q;
        s->bz_q = (q + 1) % fh_sketcher::BZ_STREAMS;
        hipStream_t st = s->bz_stream[q];
        // (the copies go through the main stream, which is idle while a batch collects: behind the side stream's previous
        // kernel they would hold the caller's buffer for as long as that runs)
        HIP_TRY(hipMemcpyAsync(s->d_bz_members + s->bz_acc_n, mt, (size_t)n_members * sizeof(fh_bgzf_member), hipMemcpyHostToDevice,
                               s->stream));
        HIP_TRY(hipMemcpyAsync(s->d_comp + s->bz_acc_bytes, s->h_stage[b] + STAGE_HEADROOM, bytes, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipEventRecord(s->stage_done[b], s->stream));
        HIP_TRY(hipStreamWaitEvent(st, s->stage_done[b], 0));
        s->stage_busy[b] = true;
        s->stage_next = (b + 1) % N_STAGE;
        HIP_TRY(launch_bgzf_inflate(s->d_comp, s->d_bz_members + s->bz_acc_n, n_members, s->bz_text[t] + left, s->d_bz_status, st));
        HIP_TRY(hipEventRecord(s->bz_done[q], st));
        s->bz_used[q] = true;
        s->bz_acc_n += n_members;
        s->bz_acc_bytes += (bytes + 255) & ~255ull;
        s->bz_acc_text += text;
    }
    if (flags & FH_BGZF_MORE) {
        if (flags & FH_BGZF_LAST) return fail(FH_ERR_INVALID, "FH_BGZF_MORE and FH_BGZF_LAST exclude each other");
        if (s->stage_busy[b]) { // the caller refills this buffer next
            HIP_TRY(hipEventSynchronize(s->stage_done[b]));
            s->stage_busy[b] = false;
        }
        return FH_OK;
    }
    const uint64_t total = left + s->bz_acc_text;
    s->bz_acc_n = 0;
    s->bz_acc_bytes = s->bz_acc_text = 0;
    for (int q = 0; q < fh_sketcher::BZ_STREAMS; ++q)
        if (s->bz_used[q]) {
            HIP_TRY(hipStreamWaitEvent(s->stream, s->bz_done[q], 0));
            s->bz_used[q] = false;
        }
    if (total == 0) return FH_OK;
    s->bz_next = t ^ 1;
    HIP_TRY(launch_fastq_cut(s->bz_text[t], (uint32_t)total, (flags & FH_BGZF_LAST) ? 1u : 0u, s->d_bz_status + 1, s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_bz_status, s->d_bz_status, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (const uint32_t st = s->h_bz_status[0]) {
        static const char *const why[] = {"", "bad block header", "invalid code", "distance too far back", "size differs from ISIZE",
                                          "stream longer or shorter than the member", "CRC-32 differs"};
        return fail(FH_ERR_INVALID, "BGZF member %u of the batch: %s", st >> 8, (st & 255u) < 7u ? why[st & 255u] : "corrupt");
    }
    if (s->h_bz_status[2]) return fail(FH_ERR_INVALID, "no FASTQ record boundary at the end of a batch of BGZF text");
    const uint64_t cut = s->h_bz_status[1];
    s->bgzf_left_ptr = s->bz_text[t] + cut;
    s->bgzf_left_len = total - cut;
    if (cut == 0) return FH_OK; // one record longer than the text so far: keep collecting (it moves on to the other buffer)
    return fastq_text_on_device(s, s->bz_text[t], cut, s->bz_packed[t], s->bz_blk_a[t], s->bz_blk_b[t], s->bz_lines, s->bz_line_cap);
}

// Plain gzip of FASTQ text -- one DEFLATE stream, no index -- inflated on the device (fh_bgzf.hip: k_gz_chunks, k_gz_chain,
// k_gz_win_*, k_gz_text).  A batch is the bytes of the pushes up to and including the first one without FH_GZ_MORE: they are
// cut into chunks of ~a block's worth, a wavefront decodes each from the first block start found in it, and the chain of
// chunks that really continue each other gives the text.  A batch that comes in pieces is decoded by ONE launch that is there
// from the first piece on: its wavefronts wait for their bytes on a word in pinned host memory that every push moves on
// (GzFeed), so the device decodes while the caller reads on.  The bytes behind the last block boundary
// reached stay on the device and lead the next batch, as do the 32 KiB of text a match may reach back into and the partial
// FASTQ record the text ended with.
constexpr uint64_t GZ_CHUNK_BYTES = 8192;   // compressed bytes per chunk, at least (12 KiB with the default buffers: GZ_MAX_CHUNKS chunks have to cover a
                                            // batch): below a block of level-1 output, so that a wavefront seldom decodes more than one block --
                                            // every chunk of a batch is resident at once, and the launch lasts as long as its longest chunk (a
                                            // chunk without a block start costs a scan of its range)
constexpr uint64_t GZ_SYM_PER_BYTE = 16;    // symbol slots per byte of a chunk: text up to eight times its DEFLATE bytes, for the chunk's own range
                                            // and the one behind it -- only once a chunk has decoded all the way THROUGH a range behind it may it
                                            // take that range's slots over as well (k_gz_chunks)
constexpr uint64_t GZ_CARRY_MAX = 4ull << 20; // undecoded bytes a batch may leave for the next
static std::atomic<uint64_t> g_gz_feed_timeouts{0};
static void free_gzip_buffers(fh_sketcher *s) {
    (void)hipFree(s->gz_sym);
    (void)hipFree(s->gz_group_map);
    (void)hipFree(s->gz_group_win);
    (void)hipFree(s->gz_claims);
    (void)hipFree(s->gz_times);
    s->gz_claims = nullptr;
    s->gz_times = nullptr;
    (void)hipFree(s->gz_recs);
    (void)hipFree(s->gz_win_in);
    (void)hipFree(s->gz_window);
    (void)hipFree(s->gz_live);
    (void)hipFree(s->gz_tile_map);
    (void)hipFree(s->gz_crc_tmp);
    (void)hipFree(s->gz_summary);
    if (s->h_gz_summary) (void)hipHostFree(s->h_gz_summary);
    if (s->h_gz_feed) (void)hipHostFree(s->h_gz_feed);
    s->h_gz_feed = nullptr;
    s->gz_sym = nullptr, s->gz_group_map = nullptr, s->gz_group_win = nullptr, s->gz_recs = nullptr;
    s->gz_win_in = s->gz_window = nullptr;
    s->gz_live = s->gz_tile_map = s->gz_crc_tmp = s->gz_summary = s->h_gz_summary = nullptr;
    s->gz_sym_elems = 0;
    s->gz_chunks_cap = 0;
}
static uint64_t gz_chunk_bytes(const fh_sketcher *s) {
    if (const char *e = cfg("gz_chunk")) return std::max<uint64_t>(1024, strtoull(e, nullptr, 10) & ~7ull); // (tests: many chunks in a small input)
    const uint64_t most = s->gz_base + s->stage_bytes;
    return std::max<uint64_t>(GZ_CHUNK_BYTES, ((most + GZ_MAX_CHUNKS - 1) / GZ_MAX_CHUNKS + 4095) & ~(uint64_t)4095);
}
// bytes one batch may hold: what a staging buffer does, and no more than the text buffer takes at eight times the size
// (reads with real quality strings inflate three- to fourfold, with constant ones five- to sixfold)
static uint64_t gz_batch_capacity(const fh_sketcher *s) {
    return std::min<uint64_t>(s->stage_bytes, std::max<uint64_t>((uint64_t)1 << 16, s->bz_text_cap / 8));
}
static int ensure_gzip_buffers(fh_sketcher *s) {
    if (int rc = ensure_bgzf_buffers(s)) return rc;
    s->gz_base = std::min<uint64_t>(GZ_CARRY_MAX, s->bz_text_cap / 4) & ~(uint64_t)255;
    const uint64_t chunk = gz_chunk_bytes(s);
    if (s->gz_summary && s->gz_chunk_alloc == chunk) return FH_OK;
    if (s->gz_summary) { // (the FH_GZ_CHUNK test knob has changed under a pooled handle: its buffers are sized by the chunk)
        if (s->gz_acc) return fail(FH_ERR_STATE, "FH_GZ_CHUNK changed in the middle of a batch");
        HIP_TRY(hipStreamSynchronize(s->stream));
        HIP_TRY(hipStreamSynchronize(s->copy_stream));
        free_gzip_buffers(s);
    }
    s->gz_chunk_alloc = chunk;
    // (gz_summary, allocated last, is what says "the buffers are there": an allocation that fails on the way frees what the
    // ones before it got -- a second call would allocate over their pointers)
    struct Undo {
        fh_sketcher *s;
        ~Undo() {
            if (!s->gz_summary) free_gzip_buffers(s);
        }
    } undo{s};
    const uint32_t n = (uint32_t)std::min<uint64_t>(GZ_MAX_CHUNKS, (s->gz_base + gz_batch_capacity(s) + chunk - 1) / chunk + 1);
    const uint64_t cap = (GZ_WINDOW + std::max<uint64_t>(GZ_SYM_PER_BYTE * chunk, 1u << 16) + 7) & ~(uint64_t)7;
    HIP_TRY(dev_malloc((void **)&s->gz_sym, (size_t)n * cap * sizeof(uint16_t) + 64));
    s->gz_sym_elems = (uint64_t)n * cap;
    s->gz_cap = cap;
    HIP_TRY(dev_malloc((void **)&s->gz_recs, (size_t)n * sizeof(GzChunk)));
    HIP_TRY(dev_malloc((void **)&s->gz_win_in, (size_t)n * GZ_WINDOW));
    HIP_TRY(dev_malloc((void **)&s->gz_live, (size_t)n * 4 * sizeof(uint32_t)));
    HIP_TRY(dev_malloc((void **)&s->gz_claims, (size_t)n * sizeof(uint32_t)));
    if (cfg("gz_times")) HIP_TRY(dev_malloc((void **)&s->gz_times, (size_t)n * 3 * sizeof(uint64_t)));
    s->gz_chunks_cap = n;
    HIP_TRY(dev_malloc((void **)&s->gz_group_map, (size_t)GZ_GROUPS * GZ_WINDOW * sizeof(uint16_t)));
    HIP_TRY(dev_malloc((void **)&s->gz_group_win, (size_t)GZ_GROUPS * GZ_WINDOW));
    HIP_TRY(dev_malloc((void **)&s->gz_window, GZ_WINDOW));
    HIP_TRY(dev_malloc((void **)&s->gz_tile_map, (size_t)(s->bz_text_cap / 4096 + 2) * sizeof(uint32_t)));
    HIP_TRY(dev_malloc((void **)&s->gz_crc_tmp, (size_t)(s->bz_text_cap / 65536 + 2) * sizeof(uint32_t)));
    HIP_TRY(host_malloc((void **)&s->h_gz_summary, (GZS_WORDS + 4) * sizeof(uint32_t)));
    HIP_TRY(host_malloc((void **)&s->h_gz_feed, sizeof(GzFeed)));
    if (!s->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    HIP_TRY(dev_malloc((void **)&s->gz_summary, GZS_WORDS * sizeof(uint32_t)));
    return FH_OK;
}

uint64_t fh_debug_gzip_feed_timeouts(void) { return g_gz_feed_timeouts.load(); }

int fh_gzip_batch_capacity(fh_sketcher *s, uint64_t *cap) {
    if (!s || !cap) return fail(FH_ERR_INVALID, "null argument");
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_stage(s)) return rc;
    if (int rc = ensure_bgzf_buffers(s)) return rc;
    *cap = gz_batch_capacity(s);
    return FH_OK;
}

int fh_push_gzip_fastq(fh_sketcher *s, uint64_t bytes, uint32_t flags, uint32_t *member_done, uint64_t *trailing) {
    if (!s || !member_done || !trailing) return fail(FH_ERR_INVALID, "null argument");
    *member_done = 0;
    *trailing = 0;
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    if (s->proc_buf) return fail(FH_ERR_STATE, "records of fh_process are waiting in the staging buffer: fh_sync first");
    if ((flags & FH_GZ_MORE) && (flags & FH_GZ_LAST)) return fail(FH_ERR_INVALID, "FH_GZ_MORE and FH_GZ_LAST exclude each other");
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_stage(s)) return rc;
    if (int rc = ensure_gzip_buffers(s)) return rc;
    if (flags & FH_GZ_FIRST) {
        if (s->gz_acc) return fail(FH_ERR_STATE, "FH_GZ_FIRST in the middle of a batch");
        s->gz_tail_len = 0;
        s->gz_bit = s->gz_valid = s->gz_crc = 0;
        s->gz_total = 0;
        s->gz_open = true;
        s->bgzf_left_len = 0;
    } else if (!s->gz_open) {
        return fail(FH_ERR_STATE, "no gzip member is open: the first push of one carries FH_GZ_FIRST");
    }
    if (s->gz_acc + bytes > gz_batch_capacity(s)) return fail(FH_ERR_INVALID, "batch longer than fh_gzip_batch_capacity");
    const int b = s->stage_next, t = s->bz_next;
    const uint64_t chunk_bytes = gz_chunk_bytes(s);
    // (the carried bytes begin at a multiple of four, where fh_bgzf.hip's bit reader wants its words; the new ones follow them)
    uint8_t *const comp = s->d_comp + s->gz_base - ((s->gz_tail_len + 3) & ~(uint64_t)3);
    GzBatch B{};
    B.comp = comp;
    B.first_bit = s->gz_bit;
    B.chunk_bits = chunk_bytes * 8u;
    B.cap = s->gz_cap;
    B.recs = s->gz_recs;
    B.sym = s->gz_sym;
    B.claims = s->gz_claims;
    B.times = s->gz_times;
    B.n_regions = s->gz_chunks_cap;
    const bool batch_start = s->gz_acc == 0;
    // From here on a failure leaves a launch behind that waits for bytes: the caller has to reset the sketcher (which tells
    // it to give up, gzip_quiesce).
    if (batch_start) {
        if (int rc = drain(s)) return rc; // the packed buffer of this slot may still feed a pending range
        HIP_TRY(hipMemsetAsync(s->gz_claims, 0, (size_t)s->gz_chunks_cap * sizeof(uint32_t), s->copy_stream));
        s->gz_fed = false;
        s->gz_t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        if ((flags & FH_GZ_MORE) && !s->gz_no_feed) {
            // the batch comes in pieces: ONE launch for all the chunks it may have, there from the first piece on; its
            // wavefronts wait for their bytes (GzFeed) -- launches of their own per piece would queue up behind each other
            HIP_TRY(hipStreamSynchronize(s->copy_stream)); // (the claims are clear, the carried bytes in place)
            GzFeed *f = s->h_gz_feed;
            f->avail = 0;
            f->state = 0;
            f->abort = 0;
            __atomic_thread_fence(__ATOMIC_SEQ_CST);
            const uint32_t most = (uint32_t)std::min<uint64_t>(s->gz_chunks_cap, (s->gz_tail_len + gz_batch_capacity(s) + chunk_bytes - 1) / chunk_bytes);
            HIP_TRY(launch_gzip_chunks(B, f, 0, 0, most, false, s->stream));
            s->gz_feeding = true;
            s->gz_fed = true;
        }
    }
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(comp + s->gz_tail_len + s->gz_acc, s->h_stage[b] + STAGE_HEADROOM + s->gz_acc, bytes, hipMemcpyHostToDevice, s->copy_stream));
        s->gz_acc += bytes;
    }
    const uint64_t n_bytes = s->gz_tail_len + s->gz_acc;
    const bool more = (flags & FH_GZ_MORE) != 0;
    // (zeros behind the last byte, for tidiness: what is decoded from there is taken back in any case.  Not while a launch
    // waits for bytes: a fill is a kernel, and might queue up behind that launch)
    if (!more && !s->gz_feeding) HIP_TRY(hipMemsetAsync(comp + n_bytes, 0, 256, s->copy_stream));
    if (s->gz_feeding) {
        // the piece is on the device: say so
        HIP_TRY(hipStreamSynchronize(s->copy_stream));
        GzFeed *f = s->h_gz_feed;
        __atomic_store_n(&f->avail, n_bytes, __ATOMIC_RELEASE);
        if (!more) __atomic_store_n(&f->state, (flags & FH_GZ_LAST) ? 2u : 1u, __ATOMIC_RELEASE);
        static const bool trace_pieces = cfg("trace") != nullptr;
        if (trace_pieces) {
            const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            fprintf(stderr, "[fh] gzip piece: %llu bytes of the batch on the device %.2f ms after its first push%s\n", (unsigned long long)n_bytes,
                    (now - s->gz_t0) * 1e3, more ? "" : " (complete)");
        }
    }
    if (more) return FH_OK;
    // ---- the batch is complete ----
    s->gz_acc = 0;
    if (n_bytes == 0) {
        gzip_quiesce(s);
        if (flags & FH_GZ_LAST) return fail(FH_ERR_INVALID, "gzip: the stream ends before its final block");
        return FH_OK;
    }
    const uint32_t n_chunks = (uint32_t)((n_bytes + chunk_bytes - 1) / chunk_bytes);
    if (n_chunks > s->gz_chunks_cap) return fail(FH_ERR_INVALID, "gzip: more chunks in a batch than there is room for");
    B.n_bytes = n_bytes;
    B.n_chunks = n_chunks;
    HIP_TRY(hipEventRecord(s->stage_done[b], s->copy_stream));
    s->stage_busy[b] = true;
    s->stage_next = (b + 1) % N_STAGE;
    if (!s->gz_feeding) { // the whole batch in one push: an ordinary launch
        HIP_TRY(hipStreamWaitEvent(s->stream, s->stage_done[b], 0));
        HIP_TRY(launch_gzip_chunks(B, nullptr, n_bytes, 0, n_chunks, (flags & FH_GZ_LAST) != 0, s->stream));
    }
    s->gz_feeding = false; // (it runs to its end now: every chunk has its bytes)
    const uint64_t left = s->bgzf_left_len;
    s->bgzf_left_len = 0;
    if (left) HIP_TRY(hipMemcpyAsync(s->bz_text[t], s->bgzf_left_ptr, left, hipMemcpyDeviceToDevice, s->stream));
    B.win_in = s->gz_win_in;
    B.window = s->gz_window;
    B.group_map = s->gz_group_map;
    B.group_win = s->gz_group_win;
    B.valid = s->gz_valid;
    B.live = s->gz_live;
    B.tile_map = s->gz_tile_map;
    B.crc_tmp = s->gz_crc_tmp;
    B.text = s->bz_text[t];
    B.left = (uint32_t)left;
    B.text_cap = std::min<uint64_t>(s->bz_text_cap - left, (1ull << 31) - 1 - left);
    B.summary = s->gz_summary;
    // (the text's CRC-32 on a side stream: it is only needed once the text has been split and queued for sketching)
    hipStream_t crc_stream = s->bz_stream[0];
    HIP_TRY(launch_gzip_batch(B, s->stream, crc_stream, s->bz_done[0]));
    HIP_TRY(hipMemcpyAsync(s->h_gz_summary, s->gz_summary, GZS_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_gz_summary + GZS_WORDS, s->gz_summary + GZS_CRC, sizeof(uint32_t), hipMemcpyDeviceToHost, crc_stream));
    struct CrcGuard { // (whatever way this call ends, the side stream is idle again)
        hipStream_t st;
        bool waited = false;
        ~CrcGuard() {
            if (!waited) (void)hipStreamSynchronize(st);
        }
    } crc_guard{crc_stream};
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->stage_busy[b] = false;
    const uint32_t *S = s->h_gz_summary;
    if (const uint32_t st = S[GZS_STATUS]) {
        static const char *const why[] = {"", "bad block header, or no chunk begins where the one before it stopped", "invalid code",
                                          "distance reaches before the start of the stream", "more text than the buffers hold",
                                          "stream longer or shorter than its bytes", "CRC-32 differs"};
        s->gz_open = false;
        // A launch that waits for its pieces depends on the copies getting past it (they do: the copy engines; if a runtime ever
        // ran them as kernels behind the waiting launch, every wavefront would give up after GZ_WAIT_TICKS, three seconds).  A
        // batch that failed after that long is taken as that: this handle decodes its batches once they are complete from
        // now on (the waiting launch never returns; the caller reads the file again through the host-side inflate as for any
        // other refusal), and the process counts it (fh_debug_gzip_feed_timeouts).
        if (s->gz_fed && std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - s->gz_t0 > 2.5) {
            s->gz_no_feed = true;
            g_gz_feed_timeouts++;
        }
        return fail(FH_ERR_INVALID, "gzip: chunk %u of the batch: %s", st >> 8, (st & 255u) < 7u ? why[st & 255u] : "corrupt");
    }
    const uint64_t total = S[GZS_TOTAL];
    const uint64_t end_bit = S[GZS_END_BIT_LO] | ((uint64_t)S[GZS_END_BIT_HI] << 32);
    const uint32_t end_state = S[GZS_END_STATE];
    static const bool trace = cfg("trace") != nullptr;
    if (trace)
        fprintf(stderr, "[fh] gzip batch: text there %.2f ms after the batch's first push\n",
                (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - s->gz_t0) * 1e3);
    if (trace)
        fprintf(stderr, "[fh] gzip batch: %llu bytes (%llu carried) in %u chunks of %llu: %u on the chain, %llu bytes of text, stopped at bit %llu (%s)\n",
                (unsigned long long)n_bytes, (unsigned long long)s->gz_tail_len, n_chunks, (unsigned long long)chunk_bytes, S[GZS_N_LIVE],
                (unsigned long long)total, (unsigned long long)end_bit, end_state == GZ_MEMBER_END ? "end of the member" : "out of input");
    if (s->gz_times) { // when the chunks' wavefronts were dispatched, had their bytes, were done: deciles over the batch's chunks
        std::vector<uint64_t> tm((size_t)n_chunks * 3);
        HIP_TRY(hipMemcpy(tm.data(), s->gz_times, tm.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        uint64_t t0 = ~0ull;
        for (uint32_t c = 0; c < n_chunks; ++c) t0 = std::min(t0, tm[3 * c]);
        fprintf(stderr, "[fh] gzip chunk times (ms after the first dispatch): chunk index: dispatched, bytes there, done\n");
        for (int q = 0; q <= 10; ++q) {
            const uint32_t c = (uint32_t)std::min<uint64_t>(n_chunks - 1, (uint64_t)n_chunks * q / 10);
            fprintf(stderr, "[fh]   %5u: %7.2f %7.2f %7.2f\n", c, (tm[3 * c] - t0) / 1e5, (tm[3 * c + 1] - t0) / 1e5, (tm[3 * c + 2] - t0) / 1e5);
        }
    }
    s->gz_total += total;
    s->gz_valid = S[GZS_VALID];
    const bool member_end = end_state == GZ_MEMBER_END;
    if (member_end) {
        s->gz_open = false;
        if (!S[GZS_HAVE_TRAILER]) return fail(FH_ERR_INVALID, "gzip: the member's trailer is cut short");
        if (S[GZS_ISIZE_WANT] != (uint32_t)s->gz_total) return fail(FH_ERR_INVALID, "gzip: size differs from the member's trailer");
        s->gz_tail_len = 0;
    } else {
        if (flags & FH_GZ_LAST) {
            s->gz_open = false;
            return fail(FH_ERR_INVALID, "gzip: the stream ends inside a block");
        }
        const uint64_t from = (end_bit >> 3) & ~(uint64_t)3;
        const uint64_t rest = n_bytes - from;
        // (no block boundary beyond the carried bytes: a block longer than a batch, or bytes that are no DEFLATE stream)
        if (from < s->gz_tail_len || (from == 0 && total == 0) || rest + 4 > s->gz_base) {
            s->gz_open = false;
            return fail(FH_ERR_INVALID, "gzip: no block boundary within a batch");
        }
        // (on the copy stream: the next batch's bytes, copied there too, land right behind these)
        if (rest) HIP_TRY(hipMemcpyAsync(s->d_comp + s->gz_base - ((rest + 3) & ~(uint64_t)3), comp + from, rest, hipMemcpyDeviceToDevice, s->copy_stream));
        s->gz_tail_len = rest;
        s->gz_bit = (uint32_t)(end_bit - from * 8u);
    }
    const uint64_t text_total = left + total;
    if (text_total) {
        s->bz_next = t ^ 1;
        if (S[GZS_CUT_BAD]) return fail(FH_ERR_INVALID, "no FASTQ record boundary at the end of a batch of gzip text");
        const uint64_t cut = S[GZS_CUT];
        s->bgzf_left_ptr = s->bz_text[t] + cut;
        s->bgzf_left_len = text_total - cut;
        // (cut == 0: one record longer than the text so far: keep collecting, it moves on to the other buffer)
        if (cut)
            if (int rc = fastq_text_on_device(s, s->bz_text[t], cut, s->bz_packed[t], s->bz_blk_a[t], s->bz_blk_b[t], s->bz_lines, s->bz_line_cap)) return rc;
    }
    // the checksum: of this batch's text, joined to that of the member's text before it
    HIP_TRY(hipStreamSynchronize(crc_stream));
    crc_guard.waited = true;
    s->gz_crc = crc32_join(s->gz_crc, s->h_gz_summary[GZS_WORDS], total);
    if (member_end) {
        if (S[GZS_CRC_WANT] != s->gz_crc) return fail(FH_ERR_INVALID, "gzip: CRC-32 differs from the member's trailer");
        *member_done = 1;
        *trailing = S[GZS_TRAILING];
    }
    return FH_OK;
}

// Device-side FASTA: the staged chunk is raw file text (header lines, wrapped sequence lines); which bytes are
// sequence is decided by launch_fasta_pack.  A record's sequence spans lines and chunks, so like FH_PUSH_CONTINUE
// pushes the packed stream of this chunk is preceded by the last K-1 packed bytes of the previous one (copied
// device to device).  start_state: what the chunk begins in the middle of (0 line start, 1 sequence line, 2 header).
int fh_push_fasta_text(fh_sketcher *s, uint64_t len, uint32_t start_state, uint32_t flags) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    if (s->proc_buf) return fail(FH_ERR_STATE, "records of fh_process are waiting in the staging buffer: fh_sync first");
    if (start_state > 2u) return fail(FH_ERR_INVALID, "bad start_state");
    if (len > s->stage_bytes || len >= (1ull << 30)) return fail(FH_ERR_INVALID, "text longer than the staging buffer");
    if (int rc = set_device(s)) return rc;
    if (int rc = ensure_stage(s)) return rc;
    if (!(flags & FH_PUSH_CONTINUE)) s->dprev_len = 0;
    const uint32_t halo_len = s->halo_len; // applies to this chunk only
    s->halo_len = 0;
    if (halo_len && (flags & FH_PUSH_CONTINUE)) return fail(FH_ERR_STATE, "fh_set_text_halo and FH_PUSH_CONTINUE exclude each other");
    if (len == 0) return FH_OK;
    const int b = s->stage_next;
    const uint64_t nblk = (s->stage_bytes + 4095) / 4096 + 1;
    if (!s->d_packed[b]) {
        HIP_TRY(dev_malloc((void **)&s->d_packed[b], s->stage_bytes + 64));
        HIP_TRY(dev_malloc((void **)&s->d_blk_a[b], 5 * nblk * sizeof(uint32_t))); // (FASTQ: newlines + four class counts per block)
        HIP_TRY(dev_malloc((void **)&s->d_blk_b[b], nblk * sizeof(uint32_t)));
    }
    if (!s->d_text_tot) {
        HIP_TRY(dev_malloc((void **)&s->d_text_tot, 4 * sizeof(uint32_t)));
        HIP_TRY(host_malloc((void **)&s->h_text_tot, 4 * sizeof(uint32_t)));
    }
    // the packed buffer of this slot may still feed a pending range
    if (int rc = drain(s)) return rc;
    const uint64_t K = s->p.k;
    const uint64_t carry_len = halo_len ? halo_len : std::min<uint64_t>(K - 1, s->dprev_len);
    uint8_t *dst = s->d_packed[b];
    HIP_TRY(hipMemsetAsync(s->d_text_tot, 0, 4 * sizeof(uint32_t), s->stream));
    if (s->stage_prefetched[b] == len) { // the reader's thread has the copy under way (fh_text_prefetch)
        HIP_TRY(hipStreamWaitEvent(s->stream, s->stage_done[b], 0));
    } else {
        if (s->stage_prefetched[b]) HIP_TRY(hipEventSynchronize(s->stage_done[b])); // (a prefetch of something else: let it land first)
        HIP_TRY(hipMemcpyAsync(s->d_stage[b], s->h_stage[b] + STAGE_HEADROOM, len, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipEventRecord(s->stage_done[b], s->stream));
    }
    s->stage_prefetched[b] = 0;
    s->stage_busy[b] = true;
    if (halo_len) // (the call synchronises the stream below, before s->halo can change)
        HIP_TRY(hipMemcpyAsync(dst, s->halo, halo_len, hipMemcpyHostToDevice, s->stream));
    else if (carry_len)
        HIP_TRY(hipMemcpyAsync(dst, s->dprev_ptr + s->dprev_len - carry_len, carry_len, hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(launch_fasta_pack(s->d_stage[b], len, start_state, dst + carry_len, s->d_blk_a[b], s->d_blk_b[b], s->d_text_tot,
                              s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_text_tot, s->d_text_tot, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const uint64_t n_packed = s->h_text_tot[1];
    s->stage_next = (b + 1) % N_STAGE;
    s->carry_len = 0; // the host-side carry belongs to the fh_push_block / fh_push_staged paths
    const int rc = sketch_device_range(s, dst, carry_len + n_packed, s->stream_off - carry_len);
    s->stream_off += n_packed;
    s->dprev_ptr = dst;
    s->dprev_len = carry_len + n_packed;
    return rc;
}

int fh_set_text_halo(fh_sketcher *s, const uint8_t *halo, uint32_t n) {
    if (!s || (n && !halo)) return fail(FH_ERR_INVALID, "null argument");
    if (n >= s->p.k || n > sizeof s->halo) return fail(FH_ERR_INVALID, "halo of %u bytes (at most k-1 = %u)", n, s->p.k - 1);
    memcpy(s->halo, halo, n);
    s->halo_len = n;
    return FH_OK;
}

int fh_text_bases(fh_sketcher *s, uint64_t *total_bases) {
    if (!s || !total_bases) return fail(FH_ERR_INVALID, "null argument");
    if (s->finished) { // (the device side may have been reset already: fh_finish kept the count)
        *total_bases = s->final_text_bases;
        return FH_OK;
    }
    if (int rc = set_device(s)) return rc;
    if (int rc = drain(s)) return rc;
    if (int rc = check_ctl(s)) return rc;
    *total_bases = s->h_ctl->text_bases;
    return FH_OK;
}

int fh_push_block(fh_sketcher *s, const uint8_t *bytes, uint64_t len) { return fh_push_block_ex(s, bytes, len, 0u); }

int fh_sync(fh_sketcher *s) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (int rc = set_device(s)) return rc;
    if (int rc = proc_flush(s)) return rc;
    if (int rc = drain(s)) return rc;
    if (int rc = check_ctl(s)) return rc;
    return collect_profile(s);
}

static size_t result_count(const fh_sketcher *s) { return s->res_built ? s->res.size() : s->r_n; }

// the wide columns of a large sketch, if fh_finish left them on the device
static int ensure_wide(fh_sketcher *s) {
    if (!s->wide_pending) return FH_OK;
    if (int rc = set_device(s)) return rc;
    const size_t n = s->r_n;
    HIP_TRY(hipMemcpyAsync(s->r_hash, s->o_hash, n * 8ull, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(s->r_kmer, s->o_kmer, n * 8ull, hipMemcpyDeviceToHost, s->stream));
    if (s->r_kmer_hi) HIP_TRY(hipMemcpyAsync(s->r_kmer_hi, s->o_kmer_hi, n * 8ull, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(s->r_pos, s->o_pos, n * 8ull, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->wide_pending = false;
    return FH_OK;
}

// the record form of a finished sketch (merges work on it); built from the arrays fh_finish left behind
static int ensure_records(fh_sketcher *s) {
    if (s->res_built) return FH_OK;
    // (a failed copy leaves wide_pending set: the arrays still hold an earlier sketch's columns and must not be merged)
    int prev_dev = -1;
    const bool restore = s->wide_pending && hipGetDevice(&prev_dev) == hipSuccess;
    const int rc = ensure_wide(s);
    if (restore && prev_dev != s->device) (void)hipSetDevice(prev_dev); // fh_merge(dst, src): the caller's device stays current
    if (rc) return rc;
    s->res.resize(s->r_n);
    for (size_t i = 0; i < s->r_n; ++i)
        s->res[i] = ResultRec{s->r_hash[i], s->r_count[i], s->r_extra[i], s->r_kmer[i], s->r_pos[i], s->r_kmer_hi ? s->r_kmer_hi[i] : 0ull};
    s->res_built = true;
    return FH_OK;
}

int fh_finish(fh_sketcher *s, uint64_t *n_out, uint64_t *total_kmers) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (int rc = set_device(s)) return rc;
    static const bool trace = cfg("trace") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    auto t1 = t0, t2 = t0, t3 = t0;
    if (!s->finished)
        if (int rc = proc_flush(s)) return rc;
    if (!s->finished) {
        const bool wide = s->p.k > 32;
        auto ensure_h_out = [&](size_t need) -> int {
            if (need <= s->h_out_bytes) return FH_OK;
            if (s->h_out) (void)hipHostFree(s->h_out);
            s->h_out = nullptr;
            s->h_out_bytes = 0;
            HIP_TRY(host_malloc(&s->h_out, need + need / 4));
            s->h_out_bytes = need + need / 4;
            return FH_OK;
        };
        // Small sketches: the final selection, the sort, to_vec and the control block's way to the host are ONE launch
        // queued behind whatever is still running, and ONE synchronisation -- the sorted sketch lands in the pinned result
        // buffer without a copy.  Afterwards the mirrored control block says whether everything queued did what it was
        // queued for; if not (speculation failed, a launch stopped early, more live entries than the LDS holds) the
        // step-by-step path below takes over from whatever state that left (a selection at any time is harmless).
        bool fused = false;
        if (s->fast) {
            if (int rc = ensure_out(s, (uint32_t)std::min<uint64_t>(s->p.size + 1, SMALL_MAX))) return rc;
            if (int rc = ensure_h_out(s->out_stride * (wide ? 40 : 32) + 64)) return rc;
            static const bool no_fold = cfg("no_reset_fold") != nullptr; // A/B knob
            // (the last launch's own epilogue folded in; EPI_RESET: if all went as queued the handle is left reset)
            EpiArgs e = epi_args(s, s->epi_pending | EPI_PRUNE_FORCE | EPI_SORT | EPI_GATHER | (s->spec.pending ? EPI_NEED_SPEC : 0u) |
                                        (no_fold ? 0u : EPI_RESET));
            e.n_units = s->epi_units;
            s->epi_pending = 0;
            e.check_units = s->pend.active ? s->pend.n_units : 0u;
            e.tau0 = initial_tau(s);
            e.hist_on = s->hist ? 1u : 0u;
            e.out = (uint64_t *)s->h_out;
            e.out_stride = (uint32_t)s->out_stride;
            e.h_ctl = s->h_ctl;
            HIP_TRY(launch_small_epilogue(e, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            const Ctl &c = *s->h_ctl;
            if (c.overflow == 1) return fail(FH_ERR_CAPACITY, "device hash table capacity exceeded");
            if (c.overflow == 2) return fail(FH_ERR_CAPACITY, "hash collision log capacity exceeded");
            // (the kernel asked the control block what the host would: speculation held, range ran dry, nothing too large)
            fused = c.sorted == FIN_OK || c.sorted == FIN_OK_RESET;
            s->device_clean = c.sorted == FIN_OK_RESET; // (the collision log, if any, stays readable: only its counter was reset)
            if (fused) {
                s->spec.pending = false;
                s->pend.active = false;
                s->last_tau = c.tau;
                s->last_live = c.n_live;
                s->ins_seen = c.inserted_total;
                s->n_fast_finish++;
            }
        }
        if (!fused) {
        if (int rc = drain(s)) return rc;
        if (int rc = check_ctl(s)) return rc;
        if (!s->big_mode && s->h_ctl->n_live <= (uint32_t)SMALL_MAX && !s->h_ctl->need_big) {
            HIP_TRY(launch_prune_small(s->table, s->live, s->dead, s->dead_cap, s->ctl, s->p.kind, s->p.size,
                                       s->max_hash, 0u, 1u, 1u, s->stream));
            if (int rc = check_ctl(s)) return rc;
        } else {
            if (int rc = big_prune(s)) return rc;
        }
        }
        t1 = now(); // drained + pruned
        if (!fused) {
            if (int rc = ensure_out(s, s->h_ctl->n_live)) return rc;
            HIP_TRY(launch_gather(s->table, s->live, s->ctl, (int)s->p.k, s->o_hash, s->o_count, s->o_extra, s->o_kmer,
                                  s->o_kmer_hi, s->o_pos, s->out_cap, s->stream));
        }
        // (the control block read back after the prune is final: the gather only reads it)
        if (int rc = collect_profile(s)) return rc;
        const Ctl c = *s->h_ctl;
        const uint32_t n = c.n_live;
        // D2H through one pinned staging area (pageable destinations crawl at a few GB/s); one spare record for the
        // special hash.  A small sketch (the whole block of arrays under 1 MiB: n = 1000 is 32 KB) crosses in ONE copy, the
        // host arrays the device's stride apart; a large one array by array, n entries each.
        const bool one_copy = s->o_block_bytes <= ((size_t)1 << 20);
        const size_t cap = one_copy ? s->out_stride : (size_t)n + 1;
        if (int rc = ensure_h_out(cap * (wide ? 40 : 32) + 64)) return rc;
        uint64_t *hh = (uint64_t *)s->h_out, *kk = hh + cap, *pp = kk + cap;
        uint64_t *kh = wide ? pp + cap : nullptr;
        uint32_t *cc = (uint32_t *)(pp + cap + (wide ? cap : 0)), *ee = cc + cap;
        // (nothing on the host side of this function needs the wide columns of a Mash sketch without collisions and without
        //  the special hash: the final selection is "the first `size`")
        static const bool lazy_off = cfg("no_lazy_copyout") != nullptr; // A/B knob
        s->wide_pending = !lazy_off && !one_copy && n >= (1u << 17) && s->p.kind == FH_KIND_MASH && c.n_coll == 0 && c.sp_count == 0;
        if (fused) {
            // (the columns are there already)
        } else if (n && one_copy) {
            HIP_TRY(hipMemcpyAsync(hh, s->o_block, s->o_block_bytes, hipMemcpyDeviceToHost, s->stream));
        } else if (n) {
            if (!s->wide_pending) {
                HIP_TRY(hipMemcpyAsync(hh, s->o_hash, n * 8ull, hipMemcpyDeviceToHost, s->stream));
                HIP_TRY(hipMemcpyAsync(kk, s->o_kmer, n * 8ull, hipMemcpyDeviceToHost, s->stream));
                if (wide) HIP_TRY(hipMemcpyAsync(kh, s->o_kmer_hi, n * 8ull, hipMemcpyDeviceToHost, s->stream));
                HIP_TRY(hipMemcpyAsync(pp, s->o_pos, n * 8ull, hipMemcpyDeviceToHost, s->stream));
            }
            HIP_TRY(hipMemcpyAsync(cc, s->o_count, n * 4ull, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipMemcpyAsync(ee, s->o_extra, n * 4ull, hipMemcpyDeviceToHost, s->stream));
        }
        std::vector<CollRec> coll(std::min<uint32_t>(c.n_coll, CLOG_CAP));
        if (!coll.empty())
            HIP_TRY(hipMemcpyAsync(coll.data(), s->clog, coll.size() * sizeof(CollRec), hipMemcpyDeviceToHost, s->stream));
        if (!fused || !coll.empty()) HIP_TRY(hipStreamSynchronize(s->stream));
        t2 = now(); // gathered + copied to the host
        size_t m = n;
        // the hash value that cannot be a table key, if it occurred (it sorts last)
        if (c.sp_count) {
            hh[m] = EMPTY64;
            cc[m] = (uint32_t)std::min<uint64_t>(c.sp_count, UINT32_MAX);
            ee[m] = (uint32_t)std::min<uint64_t>(c.sp_extra, UINT32_MAX);
            kk[m] = c.sp_kmer;
            if (wide) kh[m] = 0; // (two-word k-mers of this hash are all in the collision log, see upsert)
            pp[m] = c.sp_pos;
            ++m;
        }
        // final selection (mash.rs:57-60 / scaled.rs:41-58 net effect) on the ascending hash array
        if (s->p.kind == FH_KIND_MASH) {
            m = std::min<size_t>(m, s->p.size);
        } else {
            const size_t n_le = (size_t)(std::upper_bound(hh, hh + m, s->max_hash) - hh);
            m = std::max<size_t>(n_le, std::min<size_t>(m, s->p.size));
        }
        // 64-bit hash collisions between distinct k-mers: the reference keeps the bytes of the first
        // occurrence (mash.rs:52-56).  Occurrences whose k-mer differed from the slot's were logged.
        for (const CollRec &cr : coll) {
            const uint64_t *it = std::lower_bound(hh, hh + m, cr.hash);
            if (it != hh + m && *it == cr.hash && pp[it - hh] == cr.pos) {
                kk[it - hh] = cr.kmer;
                if (wide) kh[it - hh] = cr.kmer_hi;
            }
        }
        s->r_hash = hh; s->r_kmer = kk; s->r_pos = pp; s->r_count = cc; s->r_extra = ee;
        s->r_kmer_hi = kh;
        s->r_n = m;
        s->res.clear();
        s->res_built = false;
        s->total_kmers = 0;
        for (int i = 0; i < 256; ++i) s->total_kmers += c.kmer_counts[i];
        s->final_text_bases = c.text_bases;
        s->finished = true;
        t3 = now();
        if (trace)
            fprintf(stderr, "[fh] finish: drain+prune %.2f ms, gather+D2H %.2f ms, host records %.2f ms (n=%zu)\n", ms(t0, t1),
                    ms(t1, t2), ms(t2, t3), s->r_n);
    }
    if (n_out) *n_out = result_count(s);
    if (total_kmers) *total_kmers = s->total_kmers;
    return FH_OK;
}

int fh_copy_out(fh_sketcher *s, uint64_t *hashes, uint32_t *counts, uint32_t *extra_counts, uint8_t *kmers,
                uint64_t *first_pos) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (!s->finished) return fail(FH_ERR_STATE, "fh_copy_out before fh_finish");
    const int k = (int)s->p.k;
    if (s->res_built) { // after a merge the record vector is the result
        const ResultRec *res = s->res.data();
        parallel_for(s->res.size(), [=](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const ResultRec &r = res[i];
                if (hashes) hashes[i] = r.hash;
                if (counts) counts[i] = r.count;
                if (extra_counts) extra_counts[i] = r.extra;
                if (kmers) kmer_ascii(r.kmer, r.kmer_hi, k, kmers + i * (size_t)k);
                if (first_pos) first_pos[i] = r.pos;
            }
        });
        return FH_OK;
    }
    if (int rc = ensure_wide(s)) return rc;
    const uint64_t *hh = s->r_hash, *kk = s->r_kmer, *pp = s->r_pos, *kh = s->r_kmer_hi;
    const uint32_t *cc = s->r_count, *ee = s->r_extra;
    parallel_for(s->r_n, [=](size_t lo, size_t hi) {
        const size_t cnt = hi - lo;
        if (hashes) memcpy(hashes + lo, hh + lo, cnt * 8);
        if (counts) memcpy(counts + lo, cc + lo, cnt * 4);
        if (extra_counts) memcpy(extra_counts + lo, ee + lo, cnt * 4);
        if (first_pos) memcpy(first_pos + lo, pp + lo, cnt * 8);
        if (kmers)
            for (size_t i = lo; i < hi; ++i) kmer_ascii(kk[i], kh ? kh[i] : 0ull, k, kmers + i * (size_t)k);
    });
    return FH_OK;
}

int fh_copy_out_records(fh_sketcher *s, fh_kmer_count *records, uint8_t *kmers, uint64_t *first_pos) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (!s->finished) return fail(FH_ERR_STATE, "fh_copy_out before fh_finish");
    if (records) {
        if (s->res_built) {
            const ResultRec *res = s->res.data();
            parallel_for(s->res.size(), [=](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) records[i] = fh_kmer_count{res[i].hash, res[i].count, res[i].extra};
            });
        } else {
            if (int rc = ensure_wide(s)) return rc;
            const uint64_t *hh = s->r_hash;
            const uint32_t *cc = s->r_count, *ee = s->r_extra;
            parallel_for(s->r_n, [=](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) records[i] = fh_kmer_count{hh[i], cc[i], ee[i]};
            });
        }
    }
    if (!kmers && !first_pos) return FH_OK;
    return fh_copy_out(s, nullptr, nullptr, nullptr, kmers, first_pos);
}

int fh_copy_out_kmers(fh_sketcher *s, const uint32_t *rows, uint64_t n_rows, uint8_t *kmers) {
    if (!s || (n_rows && (!rows || !kmers))) return fail(FH_ERR_INVALID, "null argument");
    if (!s->finished) return fail(FH_ERR_STATE, "fh_copy_out before fh_finish");
    const int k = (int)s->p.k;
    const size_t n = result_count(s);
    if (!s->res_built)
        if (int rc = ensure_wide(s)) return rc;
    for (uint64_t i = 0; i < n_rows; ++i) {
        const size_t r = rows[i];
        if (r >= n) return fail(FH_ERR_INVALID, "row %zu of a sketch of %zu hashes", r, n);
        if (s->res_built) kmer_ascii(s->res[r].kmer, s->res[r].kmer_hi, k, kmers + i * (size_t)k);
        else kmer_ascii(s->r_kmer[r], s->r_kmer_hi ? s->r_kmer_hi[r] : 0ull, k, kmers + i * (size_t)k);
    }
    return FH_OK;
}

int fh_copy_out_rows(fh_sketcher *s, const uint32_t *rows, uint64_t n_rows, fh_kmer_count *records, uint8_t *kmers) {
    if (!s || (n_rows && !rows)) return fail(FH_ERR_INVALID, "null argument");
    if (!s->finished) return fail(FH_ERR_STATE, "fh_copy_out before fh_finish");
    const int k = (int)s->p.k;
    const size_t n = result_count(s);
    if (!s->res_built && s->wide_pending) {
        // the wide columns are still on the device: a few rows of them are gathered there (2 M records: 48 MB stay where
        // they are, 10 000 rows cross as 160 KB); most of a sketch: the columns come over after all
        if (n_rows > n / 8 || n_rows > (1u << 22)) {
            if (int rc = ensure_wide(s)) return rc;
        } else if (n_rows) {
            for (uint64_t i = 0; i < n_rows; ++i)
                if (rows[i] >= n) return fail(FH_ERR_INVALID, "row %zu of a sketch of %zu hashes", (size_t)rows[i], n);
            if (int rc = set_device(s)) return rc;
            if (n_rows > s->rows_cap) {
                (void)hipFree(s->d_rows); (void)hipFree(s->d_rows_out);
                if (s->h_rows) (void)hipHostFree(s->h_rows);
                if (s->h_rows_out) (void)hipHostFree(s->h_rows_out);
                s->d_rows = nullptr; s->d_rows_out = nullptr; s->h_rows = nullptr; s->h_rows_out = nullptr;
                s->rows_cap = 0;
                const size_t cap = std::max<size_t>(n_rows, 16384);
                HIP_TRY(dev_malloc(&s->d_rows, cap * 4));
                HIP_TRY(dev_malloc(&s->d_rows_out, cap * 24));
                HIP_TRY(host_malloc(&s->h_rows, cap * 4));
                HIP_TRY(host_malloc(&s->h_rows_out, cap * 24));
                s->rows_cap = cap;
            }
            memcpy(s->h_rows, rows, n_rows * 4);
            HIP_TRY(hipMemcpyAsync(s->d_rows, s->h_rows, n_rows * 4, hipMemcpyHostToDevice, s->stream));
            HIP_TRY(launch_gather_rows(s->o_hash, s->o_kmer, s->r_kmer_hi ? s->o_kmer_hi : nullptr, s->d_rows, (uint32_t)n_rows, s->d_rows_out, s->stream));
            HIP_TRY(hipMemcpyAsync(s->h_rows_out, s->d_rows_out, n_rows * (s->r_kmer_hi ? 24 : 16), hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            const uint64_t *gh = s->h_rows_out, *gk = gh + n_rows, *gkh = s->r_kmer_hi ? gk + n_rows : nullptr;
            for (uint64_t i = 0; i < n_rows; ++i) {
                const size_t r = rows[i];
                if (records) records[i] = fh_kmer_count{gh[i], s->r_count[r], s->r_extra[r]};
                if (kmers) kmer_ascii(gk[i], gkh ? gkh[i] : 0ull, k, kmers + i * (size_t)k);
            }
            return FH_OK;
        }
    }
    for (uint64_t i = 0; i < n_rows; ++i) {
        const size_t r = rows[i];
        if (r >= n) return fail(FH_ERR_INVALID, "row %zu of a sketch of %zu hashes", r, n);
        if (s->res_built) {
            if (records) records[i] = fh_kmer_count{s->res[r].hash, s->res[r].count, s->res[r].extra};
            if (kmers) kmer_ascii(s->res[r].kmer, s->res[r].kmer_hi, k, kmers + i * (size_t)k);
        } else {
            if (records) records[i] = fh_kmer_count{s->r_hash[r], s->r_count[r], s->r_extra[r]};
            if (kmers) kmer_ascii(s->r_kmer[r], s->r_kmer_hi ? s->r_kmer_hi[r] : 0ull, k, kmers + i * (size_t)k);
        }
    }
    return FH_OK;
}

int fh_result_counts(fh_sketcher *s, const uint32_t **counts, const uint32_t **extra_counts, uint64_t *n) {
    if (!s || !counts || !extra_counts || !n) return fail(FH_ERR_INVALID, "null argument");
    if (!s->finished) return fail(FH_ERR_STATE, "fh_result_counts before fh_finish");
    if (s->res_built) return fail(FH_ERR_STATE, "the result of a merge is a record vector: use fh_copy_out");
    *counts = s->r_count;
    *extra_counts = s->r_extra;
    *n = s->r_n;
    return FH_OK;
}

int fh_merge_arrays(fh_sketcher *dst, uint64_t n, const uint64_t *hashes, const uint32_t *counts,
                    const uint32_t *extra_counts, const uint8_t *kmers, const uint64_t *first_pos,
                    uint64_t total_kmers) {
    if (!dst || (n && (!hashes || !counts || !extra_counts || !kmers || !first_pos)))
        return fail(FH_ERR_INVALID, "null argument");
    if (!dst->finished) return fail(FH_ERR_STATE, "fh_merge: dst not finished");
    const int k = (int)dst->p.k;
    std::vector<ResultRec> src(n), out;
    for (uint64_t j = 0; j < n; ++j)
        src[j] = make_rec(hashes[j], counts[j], extra_counts[j], kmers + j * (size_t)k, k, first_pos[j]);
    if (int rc = ensure_records(dst)) return rc;
    if (int rc = merge_sorted(dst->res, src, out)) return rc;
    select_final(dst, out);
    dst->res.swap(out);
    dst->total_kmers += total_kmers;
    return FH_OK;
}

int fh_merge_partials(uint32_t kind, uint64_t size, double scale, uint32_t k, uint64_t nA, const uint64_t *hashesA,
                      const uint32_t *countsA, const uint32_t *extraA, const uint8_t *kmersA, const uint64_t *posA,
                      uint64_t nB, const uint64_t *hashesB, const uint32_t *countsB, const uint32_t *extraB,
                      const uint8_t *kmersB, const uint64_t *posB, uint64_t *n_out, uint64_t *out_hashes,
                      uint32_t *out_counts, uint32_t *out_extra, uint8_t *out_kmers, uint64_t *out_pos) {
    if (!n_out || k < 1 || k > (uint32_t)FH_MAX_K || (kind != FH_KIND_MASH && kind != FH_KIND_SCALED))
        return fail(FH_ERR_INVALID, "bad argument");
    if ((nA && (!hashesA || !countsA || !extraA || !kmersA || !posA)) || (nB && (!hashesB || !countsB || !extraB || !kmersB || !posB)))
        return fail(FH_ERR_INVALID, "null argument");
    std::vector<ResultRec> a(nA), b(nB), out;
    for (uint64_t j = 0; j < nA; ++j)
        a[j] = make_rec(hashesA[j], countsA[j], extraA[j], kmersA + j * (size_t)k, (int)k, posA[j]);
    for (uint64_t j = 0; j < nB; ++j)
        b[j] = make_rec(hashesB[j], countsB[j], extraB[j], kmersB + j * (size_t)k, (int)k, posB[j]);
    if (int rc = merge_sorted(a, b, out)) return rc;
    select_final_p(kind, size, kind == FH_KIND_SCALED ? scaled_max_hash(scale) : 0, out);
    *n_out = out.size();
    for (size_t j = 0; j < out.size(); ++j) {
        if (out_hashes) out_hashes[j] = out[j].hash;
        if (out_counts) out_counts[j] = out[j].count;
        if (out_extra) out_extra[j] = out[j].extra;
        if (out_kmers) kmer_ascii(out[j].kmer, out[j].kmer_hi, (int)k, out_kmers + j * (size_t)k);
        if (out_pos) out_pos[j] = out[j].pos;
    }
    return FH_OK;
}

// N partial sketches in the sharding wire format (finch_hip.h) -> one: a k-way walk over the sorted hash lists.
// The same rule as merge_sorted applied pairwise (sums are formed in 64 bits and clamped once, which equals a chain of
// saturating adds; the k-mer of the smallest first position wins), without building records or decoding k-mers.
int fh_merge_wire(uint32_t kind, uint64_t size, double scale, uint32_t k, uint64_t pad_n, uint32_t n_parts,
                  const int64_t *const *bufs, uint64_t *n_out, uint64_t *out_hashes, uint32_t *out_counts,
                  uint32_t *out_extra, uint8_t *out_kmers, uint64_t *out_pos, uint64_t *total_kmers) {
    if (!bufs || !n_out || !out_hashes || !out_counts || !out_extra || !out_kmers || !out_pos || k < 1 || k > (uint32_t)FH_MAX_K ||
        (kind != FH_KIND_MASH && kind != FH_KIND_SCALED))
        return fail(FH_ERR_INVALID, "bad argument");
    const uint64_t kmw = (k + 7) / 8;
    struct Part {
        const uint64_t *h, *pos;
        const int64_t *cnt, *ext;
        const uint8_t *km;
        uint64_t n, i;
    };
    std::vector<Part> parts(n_parts);
    uint64_t tk = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const int64_t *b = bufs[p];
        if (!b) return fail(FH_ERR_INVALID, "null partial");
        const uint64_t n = (uint64_t)b[0];
        if (n > pad_n) return fail(FH_ERR_INVALID, "partial sketch larger than its padding");
        tk += (uint64_t)b[1];
        const int64_t *q = b + 2;
        parts[p] = Part{(const uint64_t *)q, (const uint64_t *)(q + 3 * pad_n), q + pad_n, q + 2 * pad_n,
                        (const uint8_t *)(q + 4 * pad_n), n, 0};
        for (uint64_t j = 1; j < n; ++j)
            if (parts[p].h[j] <= parts[p].h[j - 1]) return fail(FH_ERR_INVALID, "merge: input not ascending");
    }
    const uint64_t max_hash = kind == FH_KIND_SCALED ? scaled_max_hash(scale) : 0;
    uint64_t m = 0;
    for (;;) {
        // smallest head
        bool any = false;
        uint64_t hmin = 0;
        for (const Part &pt : parts)
            if (pt.i < pt.n && (!any || pt.h[pt.i] < hmin)) {
                hmin = pt.h[pt.i];
                any = true;
            }
        if (!any) break;
        // (mash: nothing beyond the size-th distinct hash can be kept; scaled needs the count <= max_hash first)
        if (kind == FH_KIND_MASH && m >= size) break;
        uint64_t c = 0, e = 0, best_pos = 0;
        const uint8_t *best_km = nullptr;
        for (Part &pt : parts)
            if (pt.i < pt.n && pt.h[pt.i] == hmin) {
                c += (uint64_t)pt.cnt[pt.i];
                e += (uint64_t)pt.ext[pt.i];
                if (!best_km || pt.pos[pt.i] < best_pos) {
                    best_pos = pt.pos[pt.i];
                    best_km = pt.km + pt.i * kmw * 8;
                }
                ++pt.i;
            }
        out_hashes[m] = hmin;
        out_counts[m] = (uint32_t)std::min<uint64_t>(c, UINT32_MAX);
        out_extra[m] = (uint32_t)std::min<uint64_t>(e, UINT32_MAX);
        memcpy(out_kmers + m * (size_t)k, best_km, k);
        out_pos[m] = best_pos;
        ++m;
    }
    if (kind == FH_KIND_SCALED) {
        const uint64_t n_le = (uint64_t)(std::upper_bound(out_hashes, out_hashes + m, max_hash) - out_hashes);
        m = std::max<uint64_t>(n_le, std::min<uint64_t>(m, size));
    }
    *n_out = m;
    if (total_kmers) *total_kmers = tk;
    return FH_OK;
}

int fh_merge(fh_sketcher *dst, const fh_sketcher *src) {
    if (!dst || !src) return fail(FH_ERR_INVALID, "null handle");
    if (!src->finished) return fail(FH_ERR_STATE, "fh_merge: src not finished");
    if (dst->p.k != src->p.k || dst->p.kind != src->p.kind || dst->p.seed != src->p.seed || dst->p.size != src->p.size)
        return fail(FH_ERR_INVALID, "fh_merge: incompatible sketch parameters");
    if (int rc = ensure_records(const_cast<fh_sketcher *>(src))) return rc;
    const size_t n = src->res.size();
    const int k = (int)src->p.k;
    std::vector<uint64_t> hh(n), pp(n);
    std::vector<uint32_t> cc(n), ee(n);
    std::vector<uint8_t> km(n * (size_t)k + 1);
    for (size_t i = 0; i < n; ++i) {
        hh[i] = src->res[i].hash;
        cc[i] = src->res[i].count;
        ee[i] = src->res[i].extra;
        pp[i] = src->res[i].pos;
        kmer_ascii(src->res[i].kmer, src->res[i].kmer_hi, k, km.data() + i * (size_t)k);
    }
    return fh_merge_arrays(dst, n, hh.data(), cc.data(), ee.data(), km.data(), pp.data(), src->total_kmers);
}

// ---- N resident read blocks on N devices -> one sketch ----
// The fan-out of sketch_files (lib.rs:34-36: one rayon worker per file) applied to the read blocks of ONE input: a team of
// library threads, one per handle, kept between calls (starting and joining a std::thread costs 50-100 us; a rank's share of
// configs[3] on 8 GPUs is 10 ms).  The threads only ever wait on the condition variable between calls, so the team's
// destructor at process exit finds them idle.
} // extern "C"
namespace {
class BlockTeam {
    struct Job {
        fh_sketcher *h = nullptr;
        const void *block = nullptr;
        uint64_t len = 0, off = 0;
        int rc = FH_OK;
        std::string err;
    };
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> th;
    std::vector<Job> jobs;
    uint64_t generation = 0;
    size_t n_jobs = 0, n_done = 0;
    bool quit = false;

    // A worker thread moves next to its device: the CPUs of the NUMA node the GPU's PCIe slot hangs off (launch and
    // completion paths of eight devices then do not all cross one socket), as far as the process is allowed on them; nothing
    // changes if sysfs does not say (numa_node -1 or absent), and the caller's own thread is never touched.
    static void sit_near(int device) {
        static thread_local int sitting = -1;
        // what the thread was allowed BEFORE its first pin: a later pin to another node is cut out of this mask, not out of the
        // one the previous pin left (whose intersection with another node's CPUs is empty)
        static thread_local cpu_set_t allowed;
        static thread_local bool have_allowed = false;
        if (sitting == device || cfg("no_numa_pin")) return;
        if (!have_allowed) {
            if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
            have_allowed = true;
        }
        char bdf[32] = {0};
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return;
        for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
        char path[160];
        snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
        int node = -1;
        if (FILE *f = fopen(path, "r")) {
            if (fscanf(f, "%d", &node) != 1) node = -1;
            fclose(f);
        }
        if (node < 0) return;
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        char list[4096] = {0};
        if (FILE *f = fopen(path, "r")) {
            if (!fgets(list, (int)sizeof(list), f)) list[0] = 0;
            fclose(f);
        }
        cpu_set_t want;
        CPU_ZERO(&want);
        int n = 0;
        for (char *p = list; *p;) { // "0-63,128-191"
            char *e;
            const long a = strtol(p, &e, 10);
            if (e == p) break;
            long b = a;
            if (*e == '-') b = strtol(e + 1, &e, 10);
            for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
                if (CPU_ISSET((int)c, &allowed)) {
                    CPU_SET((int)c, &want);
                    ++n;
                }
            p = (*e == ',') ? e + 1 : e;
            if (*e != ',') break;
        }
        // (recorded only once the thread really sits there: a pin that did not happen is tried again at the next job)
        if (n > 0 && sched_setaffinity(0, sizeof(want), &want) == 0) sitting = device;
    }
    static void run_one(Job &j) {
        int rc = fh_reset(j.h);
        if (rc == FH_OK) rc = fh_set_stream_offset(j.h, j.off);
        if (rc == FH_OK) rc = fh_push_device(j.h, j.block, j.len);
        if (rc == FH_OK) rc = fh_finish(j.h, nullptr, nullptr);
        j.rc = rc;
        if (rc != FH_OK) j.err = fh_last_error();
    }
    void worker(size_t me) {
        uint64_t seen = 0;
        for (;;) {
            Job *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quit || (generation != seen && me < n_jobs); });
                if (quit) return;
                seen = generation;
                j = &jobs[me];
            }
            sit_near(j->h->device);
            run_one(*j);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (++n_done == n_jobs) cv_done.notify_all();
            }
        }
    }

public:
    std::mutex call_mu; // one call at a time uses the team
    ~BlockTeam() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_work.notify_all();
        for (auto &t : th)
            if (t.joinable()) t.join();
    }
    // block i on handle i, each on its own thread (the caller's runs block 0); -> first error
    int run(fh_sketcher *const *handles, const void *const *blocks, const uint64_t *lens, const uint64_t *offs, uint32_t n, std::string &err) {
        size_t have = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            jobs.assign(n, Job{});
            for (uint32_t i = 0; i < n; ++i) jobs[i] = Job{handles[i], blocks[i], lens[i], offs[i], FH_OK, {}};
            try { // threads for blocks 1 .. n-1 (worker w serves jobs[w + 1]); what cannot get a thread runs on the caller's
                th.reserve(n);
                while (th.size() + 1 < n) {
                    const size_t me = th.size() + 1;
                    th.emplace_back([this, me] { worker(me); });
                }
            } catch (...) {
            }
            have = std::min<size_t>(th.size() + 1, n);
            n_jobs = have;
            n_done = 1; // (job 0 is the caller's)
            ++generation;
        }
        cv_work.notify_all();
        run_one(jobs[0]);
        for (size_t i = have; i < n; ++i) run_one(jobs[i]);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return n_done >= n_jobs; });
            n_jobs = 0;
        }
        for (uint32_t i = 0; i < n; ++i)
            if (jobs[i].rc != FH_OK) { // (the first block that failed; the others' handles hold their partial sketches, finished)
                err = "block " + std::to_string(i) + " (device " + std::to_string(jobs[i].h->device) + "): " + jobs[i].err;
                return jobs[i].rc;
            }
        return FH_OK;
    }
};
BlockTeam &block_team() {
    static BlockTeam t;
    return t;
}
} // namespace
extern "C" {

int fh_sketch_device_blocks(fh_sketcher *const *handles, const void *const *dev_blocks, const uint64_t *lens,
                            const uint64_t *stream_offsets, uint32_t n) {
    if (!handles || !dev_blocks || !lens || !stream_offsets || n == 0) return fail(FH_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < n; ++i) {
        if (!handles[i]) return fail(FH_ERR_INVALID, "null handle");
        for (uint32_t j = 0; j < i; ++j)
            if (handles[j] == handles[i]) return fail(FH_ERR_INVALID, "the same handle for two blocks");
        const fh_params &a = handles[0]->p, &b = handles[i]->p;
        if (a.k != b.k || a.kind != b.kind || a.seed != b.seed || a.size != b.size || memcmp(&a.scale, &b.scale, sizeof(double)) != 0)
            return fail(FH_ERR_INVALID, "fh_sketch_device_blocks: incompatible sketch parameters");
    }
    // the caller's current device is the caller's again on EVERY way out (job 0 and the merge run on its thread and set others)
    struct DeviceGuard {
        int prev = -1;
        DeviceGuard() { (void)hipGetDevice(&prev); }
        ~DeviceGuard() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } device_guard;
    std::string err;
    int rc;
    try {
        BlockTeam &team = block_team();
        std::lock_guard<std::mutex> call(team.call_mu);
        rc = team.run(handles, dev_blocks, lens, stream_offsets, n, err);
    } catch (const std::exception &e) {
        return fail(FH_ERR_STATE, "fh_sketch_device_blocks: %s", e.what());
    }
    if (rc != FH_OK) return fail(rc, "fh_sketch_device_blocks: %s", err.c_str());
    if (n == 1) return FH_OK;
    // the host-side merge: union of the ascending partial sketches, counts summed (saturating), k-mer of
    // the smallest first position, re-selection -- on the records, without the detour through ASCII k-mers fh_merge takes
    fh_sketcher *dst = handles[0];
    if (int r = ensure_records(dst)) return r;
    std::vector<ResultRec> out;
    for (uint32_t i = 1; i < n; ++i) {
        if (int r = ensure_records(handles[i])) return r;
        if (int r = merge_sorted(dst->res, handles[i]->res, out)) return r;
        select_final(dst, out);
        dst->res.swap(out);
        dst->total_kmers += handles[i]->total_kmers;
    }
    return FH_OK;
}

int fh_set_profiling(fh_sketcher *s, int enable) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    s->profiling = enable != 0;
    return FH_OK;
}

int fh_kernel_time(fh_sketcher *s, double *total_ms, uint64_t *launches, uint64_t *positions) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (int rc = set_device(s)) return rc;
    if (int rc = drain(s)) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (int rc = collect_profile(s)) return rc;
    if (total_ms) *total_ms = s->prof_ms;
    if (launches) *launches = s->prof_launches;
    if (positions) *positions = s->prof_positions;
    return FH_OK;
}

int fh_debug_add_counts(fh_sketcher *s, uint64_t add_count, uint64_t add_extra) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (s->finished) return fail(FH_ERR_STATE, "sketcher already finished; call fh_reset");
    if (int rc = proc_flush(s)) return rc;
    if (int rc = set_device(s)) return rc;
    if (int rc = drain(s)) return rc; // (everything pushed so far is in the table, its new entries on the live list)
    if (int rc = flush_epilogue(s)) return rc;
    if (!s->fast) HIP_TRY(launch_live_flatten(s->ctl, s->stream));
    HIP_TRY(launch_debug_add_counts(s->table, s->live, s->ctl, add_count, add_extra, s->stream));
    return FH_OK;
}

int fh_debug_counters(fh_sketcher *s, uint64_t *launches, uint64_t *relaunches, uint64_t *big_prunes) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (launches) *launches = s->n_launches;
    if (relaunches) *relaunches = s->n_relaunches;
    if (big_prunes) *big_prunes = s->n_big_prunes;
    return FH_OK;
}

int fh_debug_fast_path(fh_sketcher *s, uint64_t *deferred, uint64_t *recovered, uint64_t *fused_finishes) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (deferred) *deferred = s->n_spec_deferred;
    if (recovered) *recovered = s->n_spec_recovered;
    if (fused_finishes) *fused_finishes = s->n_fast_finish;
    return FH_OK;
}

int fh_debug_speculation(fh_sketcher *s, uint64_t *first_pass, uint64_t *second_pass) {
    if (!s) return fail(FH_ERR_INVALID, "null handle");
    if (first_pass) *first_pass = s->n_spec + s->n_sampled; // (blocks sketched at a guessed threshold: by length or by sample)
    if (second_pass) *second_pass = s->n_spec_fallback;
    return FH_OK;
}

int fh_measure_read_bandwidth(int device, const void *dev_bytes, uint64_t bytes, int reps, double *gb_per_s) {
    if (!dev_bytes || !gb_per_s || bytes < 16) return fail(FH_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    uint32_t *sink = nullptr;
    HIP_TRY(dev_malloc(&sink, 16));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    double best = 0.0;
    for (int r = 0; r < std::max(reps, 1) + 1; ++r) { // first pass warms up
        HIP_TRY(hipEventRecord(e0, nullptr));
        HIP_TRY(launch_read_probe(dev_bytes, bytes, sink, nullptr));
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms > 0.f) best = std::max(best, (double)(bytes / 16 * 16) / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    *gb_per_s = best;
    return FH_OK;
}

int fh_device_alloc(int device, uint64_t bytes, void **out) {
    if (!out) return fail(FH_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(dev_malloc(out, bytes ? bytes : 16));
    return FH_OK;
}

int fh_device_free(int device, void *p) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(p));
    return FH_OK;
}

int fh_copy_to_device(int device, void *dst, const void *src, uint64_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return FH_OK;
}

int fh_copy_from_device(int device, void *dst, const void *src, uint64_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return FH_OK;
}

int fh_synth_genome_host(uint8_t *out, uint64_t len, uint64_t seed) {
    if (!out && len) return fail(FH_ERR_INVALID, "null argument");
    for (uint64_t i = 0; i < len; ++i) out[i] = synth_genome_base(seed, i);
    return FH_OK;
}

int fh_synth_genome_device(int device, void *dev_out, uint64_t len, uint64_t seed) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(launch_synth_genome((uint8_t *)dev_out, len, seed, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return FH_OK;
}

int fh_synth_reads_host(uint8_t *out, const uint8_t *genome, uint64_t genome_len, uint64_t first_read,
                        uint64_t n_reads, uint32_t read_len, uint64_t seed, uint32_t sub_ppm, uint32_t n_ppm) {
    if ((!out || !genome) && n_reads) return fail(FH_ERR_INVALID, "null argument");
    if (genome_len < read_len) return fail(FH_ERR_INVALID, "genome shorter than a read");
    const uint64_t rec = (uint64_t)read_len + 1;
    for (uint64_t r = 0; r < n_reads; ++r)
        for (uint32_t j = 0; j <= read_len; ++j)
            out[r * rec + j] = synth_read_byte(genome, genome_len, first_read + r, j, read_len, seed, sub_ppm, n_ppm);
    return FH_OK;
}

int fh_synth_reads_device(int device, void *dev_out, const void *dev_genome, uint64_t genome_len,
                          uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t seed,
                          uint32_t sub_ppm, uint32_t n_ppm) {
    if (genome_len < read_len) return fail(FH_ERR_INVALID, "genome shorter than a read");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(launch_synth_reads((uint8_t *)dev_out, (const uint8_t *)dev_genome, genome_len, first_read, n_reads,
                               read_len, seed, sub_ppm, n_ppm, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return FH_OK;
}

} // extern "C"
