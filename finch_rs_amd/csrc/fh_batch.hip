// fh_batch.hip -- host side of the batch sketcher (include/finch_hip.h, "many sketches per launch"): the files a worker of
// finch::sketch_files (lib/src/lib.rs:29-49) has staged are sketched by ONE launch of k2_batch (fh_k2b.hip) and finished by
// ONE launch of k_batch_epilogue (fh_kernels.hip, a workgroup per file), behind ONE host-to-device copy and in front of ONE
// synchronisation -- where a file through an fh_sketcher costs a copy, three launches and a synchronisation of its own.
//
// What is resident per batch handle: max_files partitions (control block, table partition of PART_CAP entries, live / dead
// lists, 256 shard lists), and per slot (two: the caller fills one while the device works on the other) a pinned staging
// buffer, its device twin, the pinned result columns and mirrored control blocks of max_files sketches.  The staging
// buffer begins with the batch's file descriptors (BatchFile, fh_device.h), so descriptors and sequence cross the link
// in one copy.
//
// Exactness: a file is sketched at ONE threshold below which E = 4 n (3 n for n > 2000) of its positions' hashes are
// expected; the epilogue keeps the n smallest of what was admitted.  That IS the reference's sketch (mash.rs:34-63: the
// bottom n distinct hashes with their exact occurrence counts) whenever at least n distinct hashes lie at or below the
// threshold -- or the threshold admitted everything -- and nothing overflowed and no two k-mers shared a 64-bit hash; in
// every other case the file is reported as not taken (status 1) and the caller sketches it through an fh_sketcher.  There
// is no other outcome: the batch path never returns an approximate sketch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/finch_hip.h"
#include "fh_core.h"
#include "fh_device.h"
#include "fh_internal.h"
#include "fh_kernels.h"
#include "fh_options.h"
#include "fh_pack2.h"

using namespace fh;

namespace {

#define BHIP_TRY(expr)                                                                                      \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) return api_fail(FH_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

constexpr uint32_t PART_CAP = 32768;      // table entries per file: at most SMALL_MAX = 12288 hashes are ever wanted in one
constexpr uint32_t PART_LIVE = 16384;     // live / dead list entries per file (> SMALL_MAX)
constexpr uint32_t PART_SHARD_CAP = 512;  // entries per shard list (inserts are dealt over the 256 lists drain by drain)
constexpr uint32_t PART_CLOG = 64;        // collision records per file (any collision sends the file the long way)
constexpr uint64_t BATCH_MAX_N = 3000;    // kmers_to_sketch the in-LDS selection serves (fh_api.hip SMALL_N_MAX)
constexpr uint32_t BATCH_MAX_FILES = 4096;
constexpr uint64_t BATCH_MAX_WAVES = 4096; // 16 per CU x 256 CUs

uint64_t expected_below(uint64_t n) { return n <= 2000 ? 4 * n : 3 * n; }

} // namespace

struct fh_batch {
    fh_params p{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t max_files = 0;
    uint64_t data_bytes = 0, header_bytes = 0;
    uint32_t out_stride = 0; // entries between the columns of one sketch
    size_t out_words = 0;    // u64 words of one sketch's columns
    // partitions
    Ctl *ctls = nullptr;
    Entry *tables = nullptr;
    uint32_t *live = nullptr, *dead = nullptr, *shard_cnt = nullptr, *shard_buf = nullptr;
    CollRec *clog = nullptr;
    BatchPartition *d_parts = nullptr;
    struct Slot {
        uint8_t *h_stage = nullptr, *d_stage = nullptr;
        EpiArgs *d_epi = nullptr;
        Ctl *h_ctl = nullptr;
        uint64_t *h_out = nullptr;
        hipEvent_t done = nullptr, k0 = nullptr, k1 = nullptr;
        bool in_flight = false, waited = false;
        uint32_t n_files = 0;
        std::vector<uint64_t> tau, len;
        std::vector<uint8_t> status;
        uint64_t positions = 0;
    } slot[2];
    bool profiling = false;
    double prof_ms = 0.0;
    uint64_t prof_launches = 0, prof_positions = 0;
    uint64_t n_taken = 0, n_not_taken = 0;
};

namespace {

void destroy(fh_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (auto &s : b->slot) {
        if (s.h_stage) (void)hipHostFree(s.h_stage);
        if (s.d_stage) (void)hipFree(s.d_stage);
        if (s.d_epi) (void)hipFree(s.d_epi);
        if (s.h_ctl) (void)hipHostFree(s.h_ctl);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.k0) (void)hipEventDestroy(s.k0);
        if (s.k1) (void)hipEventDestroy(s.k1);
    }
    if (b->ctls) (void)hipFree(b->ctls);
    if (b->tables) (void)hipFree(b->tables);
    if (b->live) (void)hipFree(b->live);
    if (b->dead) (void)hipFree(b->dead);
    if (b->shard_cnt) (void)hipFree(b->shard_cnt);
    if (b->shard_buf) (void)hipFree(b->shard_buf);
    if (b->clog) (void)hipFree(b->clog);
    if (b->d_parts) (void)hipFree(b->d_parts);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int build(fh_batch *b) {
    const uint32_t F = b->max_files;
    BHIP_TRY(hipSetDevice(b->device));
    BHIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    BHIP_TRY(api_dev_malloc((void **)&b->ctls, (size_t)F * sizeof(Ctl)));
    BHIP_TRY(api_dev_malloc((void **)&b->tables, (size_t)F * PART_CAP * sizeof(Entry)));
    BHIP_TRY(api_dev_malloc((void **)&b->live, (size_t)F * PART_LIVE * sizeof(uint32_t)));
    BHIP_TRY(api_dev_malloc((void **)&b->dead, (size_t)F * PART_LIVE * sizeof(uint32_t)));
    BHIP_TRY(api_dev_malloc((void **)&b->shard_cnt, (size_t)F * N_SHARDS * SHARD_STRIDE * sizeof(uint32_t)));
    BHIP_TRY(api_dev_malloc((void **)&b->shard_buf, (size_t)F * N_SHARDS * PART_SHARD_CAP * sizeof(uint32_t)));
    BHIP_TRY(api_dev_malloc((void **)&b->clog, (size_t)F * PART_CLOG * sizeof(CollRec)));
    BHIP_TRY(api_dev_malloc((void **)&b->d_parts, (size_t)F * sizeof(BatchPartition)));
    std::vector<BatchPartition> parts(F);
    for (uint32_t f = 0; f < F; ++f) {
        BatchPartition &q = parts[f];
        q.ctl = b->ctls + f;
        q.table = b->tables + (size_t)f * PART_CAP;
        q.live = b->live + (size_t)f * PART_LIVE;
        q.shard_cnt = b->shard_cnt + (size_t)f * N_SHARDS * SHARD_STRIDE;
        q.shard_buf = b->shard_buf + (size_t)f * N_SHARDS * PART_SHARD_CAP;
        q.clog = b->clog + (size_t)f * PART_CLOG;
        q.cap = PART_CAP;
        q.live_cap = PART_LIVE;
        q.clog_cap = PART_CLOG;
        q.shard_cap = PART_SHARD_CAP;
    }
    BHIP_TRY(hipMemcpyAsync(b->d_parts, parts.data(), (size_t)F * sizeof(BatchPartition), hipMemcpyHostToDevice, b->stream));
    BHIP_TRY(launch_batch_init(b->d_parts, F, b->p.size, 1u, b->stream));
    // a sketch's columns as fh_finish lays them out: hash | k-mer | first position | count | extra, out_stride entries apart
    b->out_stride = (uint32_t)(((size_t)std::min<uint64_t>(b->p.size + 1, (uint64_t)SMALL_MAX) + 2) & ~(size_t)1);
    b->out_words = (size_t)b->out_stride * 4; // 3 x 8 + 2 x 4 bytes per entry
    b->header_bytes = (((uint64_t)F * sizeof(BatchFile)) + 4095) & ~4095ull;
    for (auto &s : b->slot) {
        BHIP_TRY(api_host_malloc((void **)&s.h_stage, b->header_bytes + b->data_bytes + 64));
        BHIP_TRY(api_dev_malloc((void **)&s.d_stage, b->header_bytes + b->data_bytes + 64));
        BHIP_TRY(api_dev_malloc((void **)&s.d_epi, (size_t)F * sizeof(EpiArgs)));
        BHIP_TRY(api_host_malloc((void **)&s.h_ctl, (size_t)F * sizeof(Ctl)));
        BHIP_TRY(api_host_malloc((void **)&s.h_out, (size_t)F * b->out_words * sizeof(uint64_t)));
        BHIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        BHIP_TRY(hipEventCreate(&s.k0));
        BHIP_TRY(hipEventCreate(&s.k1));
        std::vector<EpiArgs> epi(F);
        for (uint32_t f = 0; f < F; ++f) {
            EpiArgs &e = epi[f];
            e = EpiArgs{};
            e.table = parts[f].table;
            e.live = parts[f].live;
            e.dead = b->dead + (size_t)f * PART_LIVE;
            e.dead_cap = PART_LIVE;
            e.ctl = parts[f].ctl;
            e.kind = FH_KIND_MASH;
            e.size = b->p.size;
            e.max_hash = 0;
            e.trigger = 0;
            e.flags = EPI_FLATTEN | EPI_PRUNE_FORCE | EPI_SORT | EPI_GATHER | EPI_RESET;
            e.n_units = 0;
            e.check_units = 0;
            e.tau0 = EMPTY64;
            e.hist_on = 0;
            e.out = s.h_out + (size_t)f * b->out_words;
            e.out_stride = b->out_stride;
            e.wide = 0;
            e.h_ctl = s.h_ctl + f;
        }
        BHIP_TRY(hipMemcpyAsync(s.d_epi, epi.data(), (size_t)F * sizeof(EpiArgs), hipMemcpyHostToDevice, b->stream));
        BHIP_TRY(hipStreamSynchronize(b->stream)); // (epi / parts are locals)
        s.tau.resize(F);
        s.len.resize(F);
        s.status.resize(F);
    }
    BHIP_TRY(hipStreamSynchronize(b->stream));
    return FH_OK;
}

std::mutex g_pool_mu;
std::vector<fh_batch *> g_pool;
constexpr size_t BATCH_POOL_MAX = 64;

} // namespace

namespace fh {
void batch_release_cached() {
    std::vector<fh_batch *> v;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        v.swap(g_pool);
    }
    for (fh_batch *b : v) destroy(b);
}
} // namespace fh

extern "C" {

fh_batch *fh_batch_new(const fh_params *params, int device, uint32_t max_files, uint64_t stage_bytes) {
    if (!params) {
        api_fail(FH_ERR_INVALID, "null params");
        return nullptr;
    }
    if (params->kind != FH_KIND_MASH || params->k < 1 || params->k > 32 || params->size < 1 || params->size > BATCH_MAX_N ||
        params->hash_mask != 0) {
        api_fail(FH_ERR_UNSUPPORTED, "the batch sketcher serves Mash sketches of 1..%llu hashes, k = 1..32, no test mask",
                 (unsigned long long)BATCH_MAX_N);
        return nullptr;
    }
    if (max_files < 1 || max_files > BATCH_MAX_FILES || stage_bytes < 4096 || stage_bytes > (1ull << 36)) {
        api_fail(FH_ERR_INVALID, "max_files 1..%u, stage_bytes 4 KiB..64 GiB", BATCH_MAX_FILES);
        return nullptr;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        api_fail(FH_ERR_NO_DEVICE, "no usable HIP device (this library has no CPU path)");
        return nullptr;
    }
    if (device < 0 || device >= n_dev) {
        api_fail(FH_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n_dev);
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); ++i) {
            fh_batch *c = g_pool[i];
            if (c->device == device && c->max_files == max_files && c->data_bytes == ((stage_bytes + 4095) & ~4095ull) &&
                c->p.k == params->k && c->p.size == params->size && c->p.seed == params->seed) {
                g_pool.erase(g_pool.begin() + (long)i);
                return c;
            }
        }
    }
    fh_batch *b = new (std::nothrow) fh_batch;
    if (!b) {
        api_fail(FH_ERR_CAPACITY, "out of host memory");
        return nullptr;
    }
    b->p = *params;
    b->device = device;
    b->max_files = max_files;
    b->data_bytes = (stage_bytes + 4095) & ~4095ull;
    try {
        if (build(b) != FH_OK) {
            destroy(b);
            return nullptr;
        }
    } catch (...) {
        destroy(b);
        api_fail(FH_ERR_CAPACITY, "out of host memory");
        return nullptr;
    }
    return b;
}

// A batch handle owns ~130 MiB of pinned and ~200 MiB of device memory, and pinning alone takes tens of milliseconds:
// like fh_free, fh_batch_free parks an idle handle (state clean: every partition is left reset by its epilogue) and
// fh_batch_new hands a parked one back when parameters, device and sizes match.  fh_release_cached frees what is parked.
void fh_batch_free(fh_batch *b) {
    if (!b) return;
    bool idle = !b->slot[0].in_flight && !b->slot[1].in_flight;
    if (idle) {
        std::lock_guard<std::mutex> g(g_pool_mu);
        if (g_pool.size() < BATCH_POOL_MAX) {
            b->profiling = false;
            b->prof_ms = 0.0;
            b->prof_launches = b->prof_positions = 0;
            b->n_taken = b->n_not_taken = 0;
            b->slot[0].waited = b->slot[1].waited = false;
            g_pool.push_back(b);
            return;
        }
    }
    destroy(b);
}

int fh_batch_stage(fh_batch *b, int slot, uint8_t **buf, uint64_t *cap) {
    if (!b || slot < 0 || slot > 1 || !buf || !cap) return api_fail(FH_ERR_INVALID, "bad argument");
    if (b->slot[slot].in_flight) return api_fail(FH_ERR_STATE, "slot %d is in flight: fh_batch_wait first", slot);
    *buf = b->slot[slot].h_stage + b->header_bytes;
    *cap = b->data_bytes;
    return FH_OK;
}

static int batch_submit(fh_batch *b, int slot, const uint64_t *offsets, const uint64_t *lens, uint32_t n_files, bool two_bit) try {
    if (!b || slot < 0 || slot > 1 || (n_files && (!offsets || !lens))) return api_fail(FH_ERR_INVALID, "bad argument");
    fh_batch::Slot &s = b->slot[slot];
    if (s.in_flight) return api_fail(FH_ERR_STATE, "slot %d is in flight: fh_batch_wait first", slot);
    if (n_files > b->max_files) return api_fail(FH_ERR_INVALID, "%u files in a batch of at most %u", n_files, b->max_files);
    BHIP_TRY(hipSetDevice(b->device));
    BatchFile *hd = reinterpret_cast<BatchFile *>(s.h_stage);
    uint64_t end = 0, tiles = 0, positions = 0, prev_end = 0;
    for (uint32_t f = 0; f < n_files; ++f) {
        // bytes the file occupies in the staging buffer: its stream, or its region in the two-bit form (fh_pack2.h)
        if (two_bit && lens[f] > (1ull << 40)) return api_fail(FH_ERR_INVALID, "file %u: %llu positions", f, (unsigned long long)lens[f]);
        const uint64_t bytes = two_bit ? fh_pack2::region_bytes(lens[f]) : lens[f];
        const uint64_t align = two_bit ? 63u : 15u;
        if ((offsets[f] & align) || offsets[f] > b->data_bytes || bytes > b->data_bytes - offsets[f])
            return api_fail(FH_ERR_INVALID, "file %u: [%llu, +%llu) is not a %u-byte aligned range of the staging buffer", f,
                            (unsigned long long)offsets[f], (unsigned long long)bytes, (unsigned)align + 1u);
        if (f && offsets[f] < prev_end) return api_fail(FH_ERR_INVALID, "file %u overlaps file %u (offsets ascend)", f, f - 1);
        prev_end = offsets[f] + bytes;
        const uint64_t n_tiles = (lens[f] + TILE_POS - 1) / TILE_POS;
        if (tiles + n_tiles > 0xFFFFFFF0ull) return api_fail(FH_ERR_INVALID, "batch too large");
        BatchFile &d = hd[f];
        d.seq = s.d_stage + b->header_bytes + offsets[f];
        d.len = lens[f];
        d.ctl = b->ctls + f;
        // the threshold below which E of the file's positions' hashes are expected (every position a distinct k-mer, hashes uniform)
        const uint64_t E = expected_below(b->p.size);
        uint64_t tau = EMPTY64;
        if (lens[f] > E) {
            tau = (uint64_t)((((unsigned __int128)E) << 64) / lens[f]);
            if (tau >= EMPTY64 - 1) tau = EMPTY64;
        }
        d.tau = tau;
        d.tile0 = (uint32_t)tiles;
        d.n_tiles = (uint32_t)n_tiles;
        s.tau[f] = tau;
        s.len[f] = lens[f];
        tiles += n_tiles;
        positions += lens[f];
        end = std::max(end, prev_end);
    }
    s.n_files = n_files;
    s.positions = positions;
    s.waited = false;
    if (n_files == 0) {
        s.in_flight = true;
        BHIP_TRY(hipEventRecord(s.done, b->stream));
        return FH_OK;
    }
    BHIP_TRY(hipMemcpyAsync(s.d_stage, s.h_stage, b->header_bytes + ((end + 15) & ~15ull), hipMemcpyHostToDevice, b->stream));
    if (tiles) {
        BatchArgs a{};
        a.files = reinterpret_cast<const BatchFile *>(s.d_stage);
        a.n_files = n_files;
        a.tiles_total = (uint32_t)tiles;
        const uint64_t wpb = (uint64_t)k2_waves_per_block((int)b->p.k);
        a.tiles_per_wave = (uint32_t)std::max<uint64_t>(1, (tiles + BATCH_MAX_WAVES - 1) / BATCH_MAX_WAVES);
        const uint64_t waves = ((tiles + a.tiles_per_wave - 1) / a.tiles_per_wave + wpb - 1) / wpb * wpb;
        a.seed = b->p.seed;
        a.two_bit = two_bit ? 1u : 0u;
        if (b->profiling) BHIP_TRY(hipEventRecord(s.k0, b->stream));
        BHIP_TRY(launch_k2b((int)b->p.k, a, (uint32_t)waves, b->stream));
        if (b->profiling) BHIP_TRY(hipEventRecord(s.k1, b->stream));
    }
    BHIP_TRY(launch_batch_epilogue(s.d_epi, n_files, 1u, b->stream));
    BHIP_TRY(hipEventRecord(s.done, b->stream));
    s.in_flight = true;
    return FH_OK;
} catch (...) {
    return api_fail(FH_ERR_CAPACITY, "out of host memory");
}

int fh_batch_submit(fh_batch *b, int slot, const uint64_t *offsets, const uint64_t *lens, uint32_t n_files) {
    return batch_submit(b, slot, offsets, lens, n_files, false);
}

int fh_batch_submit_packed(fh_batch *b, int slot, const uint64_t *offsets, const uint64_t *lens, uint32_t n_files) {
    return batch_submit(b, slot, offsets, lens, n_files, true);
}

uint64_t fh_batch_packed_bytes(uint64_t len) { return fh_pack2::region_bytes(len); }

int fh_batch_pack(const uint8_t *stream, uint64_t len, uint8_t *region, uint64_t region_cap) {
    if ((len && !stream) || !region) return api_fail(FH_ERR_INVALID, "bad argument");
    if (len > (1ull << 40) || fh_pack2::region_bytes(len) > region_cap)
        return api_fail(FH_ERR_CAPACITY, "%llu positions take %llu bytes in the two-bit form, the region has %llu", (unsigned long long)len,
                        (unsigned long long)fh_pack2::region_bytes(len), (unsigned long long)region_cap);
    const bool avx2 = fh_pack2::have_avx2() && !cfg_on("pack_scalar");
    const uint64_t whole = len / 32;
    fh_pack2::pack_groups(stream, (size_t)whole, region, 0, avx2);
    uint64_t groups = whole;
    if (len & 31u) {
        uint8_t last[32] = {0};
        memcpy(last, stream + 32 * whole, (size_t)(len & 31u));
        fh_pack2::pack_groups(last, 1, region, groups++, avx2);
    }
    const uint64_t n_tiles = (len + TILE_POS - 1) / TILE_POS;
    for (uint64_t g = groups; g < n_tiles * 64; ++g) {
        uint8_t *const tile = region + (g >> 6) * fh_pack2::TILE_BYTES;
        memset(tile + 8 * (g & 63), 0, 8);
        memset(tile + fh_pack2::CODES_BYTES + 4 * (g & 63), 0, 4);
    }
    memset(region + n_tiles * fh_pack2::TILE_BYTES, 0, fh_pack2::TILE_BYTES);
    return FH_OK;
}

int fh_batch_wait(fh_batch *b, int slot, uint8_t *status) {
    if (!b || slot < 0 || slot > 1) return api_fail(FH_ERR_INVALID, "bad argument");
    fh_batch::Slot &s = b->slot[slot];
    if (!s.in_flight) return api_fail(FH_ERR_STATE, "slot %d has nothing in flight", slot);
    BHIP_TRY(hipSetDevice(b->device));
    BHIP_TRY(hipEventSynchronize(s.done));
    s.in_flight = false;
    if (!s.waited) {
        s.waited = true;
        if (b->profiling && s.n_files && s.positions) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.k0, s.k1) == hipSuccess) {
                b->prof_ms += ms;
                b->prof_launches++;
                b->prof_positions += s.positions;
            } else {
                (void)hipGetLastError();
            }
        }
        for (uint32_t f = 0; f < s.n_files; ++f) {
            const Ctl &c = s.h_ctl[f];
            // taken iff the epilogue finished the sketch and left the partition reset, the guess held (or admitted everything),
            // no two k-mers shared a hash and the one hash value that cannot be a table key did not occur
            const bool fin = c.sorted == FIN_OK_RESET && c.overflow == 0 && c.need_big == 0;
            const bool full = s.tau[f] == EMPTY64 || c.inserted_total >= b->p.size;
            const bool ok = fin && full && c.n_coll == 0 && c.sp_count == 0 && (uint64_t)c.n_live <= b->p.size;
            s.status[f] = ok ? 0 : 1;
            if (ok) b->n_taken++;
            else b->n_not_taken++;
        }
    }
    if (status) memcpy(status, s.status.data(), s.n_files);
    return FH_OK;
}

static int batch_file(fh_batch *b, int slot, uint32_t i, const Ctl **c, const uint64_t **cols) {
    if (!b || slot < 0 || slot > 1) return api_fail(FH_ERR_INVALID, "bad argument");
    const fh_batch::Slot &s = b->slot[slot];
    if (s.in_flight || !s.waited) return api_fail(FH_ERR_STATE, "slot %d: fh_batch_wait first", slot);
    if (i >= s.n_files) return api_fail(FH_ERR_INVALID, "file %u of a batch of %u", i, s.n_files);
    if (s.status[i] != 0) return api_fail(FH_ERR_STATE, "file %u was not taken by the batch path", i);
    *c = &s.h_ctl[i];
    *cols = s.h_out + (size_t)i * b->out_words;
    return FH_OK;
}

int fh_batch_result(fh_batch *b, int slot, uint32_t i, uint64_t *n_out, uint64_t *total_kmers) {
    const Ctl *c = nullptr;
    const uint64_t *cols = nullptr;
    if (int rc = batch_file(b, slot, i, &c, &cols)) return rc;
    if (n_out) *n_out = c->n_live;
    if (total_kmers) {
        uint64_t t = 0;
        for (int j = 0; j < 256; ++j) t += c->kmer_counts[j];
        *total_kmers = t;
    }
    return FH_OK;
}

int fh_batch_copy_out(fh_batch *b, int slot, uint32_t i, uint64_t *hashes, uint32_t *counts, uint32_t *extra_counts, uint8_t *kmers,
                      uint64_t *first_pos) {
    const Ctl *c = nullptr;
    const uint64_t *cols = nullptr;
    if (int rc = batch_file(b, slot, i, &c, &cols)) return rc;
    const size_t n = c->n_live, st = b->out_stride;
    const uint64_t *hh = cols, *kk = hh + st, *pp = kk + st;
    const uint32_t *cc = reinterpret_cast<const uint32_t *>(pp + st), *ee = cc + st;
    if (hashes) memcpy(hashes, hh, n * 8);
    if (counts) memcpy(counts, cc, n * 4);
    if (extra_counts) memcpy(extra_counts, ee, n * 4);
    if (first_pos) memcpy(first_pos, pp, n * 8);
    if (kmers)
        for (size_t j = 0; j < n; ++j) api_kmer_ascii(kk[j], 0, (int)b->p.k, kmers + j * b->p.k);
    return FH_OK;
}

int fh_batch_copy_out_records(fh_batch *b, int slot, uint32_t i, fh_kmer_count *records, uint8_t *kmers) {
    const Ctl *c = nullptr;
    const uint64_t *cols = nullptr;
    if (int rc = batch_file(b, slot, i, &c, &cols)) return rc;
    const size_t n = c->n_live, st = b->out_stride;
    const uint64_t *hh = cols, *kk = hh + st, *pp = kk + st;
    const uint32_t *cc = reinterpret_cast<const uint32_t *>(pp + st), *ee = cc + st;
    if (records)
        for (size_t j = 0; j < n; ++j) records[j] = fh_kmer_count{hh[j], cc[j], ee[j]};
    if (kmers)
        for (size_t j = 0; j < n; ++j) api_kmer_ascii(kk[j], 0, (int)b->p.k, kmers + j * b->p.k);
    return FH_OK;
}

int fh_batch_set_profiling(fh_batch *b, int enable) {
    if (!b) return api_fail(FH_ERR_INVALID, "null handle");
    b->profiling = enable != 0;
    return FH_OK;
}

int fh_batch_kernel_time(fh_batch *b, double *total_ms, uint64_t *launches, uint64_t *positions) {
    if (!b) return api_fail(FH_ERR_INVALID, "null handle");
    if (total_ms) *total_ms = b->prof_ms;
    if (launches) *launches = b->prof_launches;
    if (positions) *positions = b->prof_positions;
    b->prof_ms = 0.0;
    b->prof_launches = b->prof_positions = 0;
    return FH_OK;
}

int fh_batch_counters(fh_batch *b, uint64_t *taken, uint64_t *not_taken) {
    if (!b) return api_fail(FH_ERR_INVALID, "null handle");
    if (taken) *taken = b->n_taken;
    if (not_taken) *not_taken = b->n_not_taken;
    return FH_OK;
}

} // extern "C"
