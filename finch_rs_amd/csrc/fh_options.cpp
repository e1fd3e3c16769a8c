// fh_options.cpp -- the option table behind fh_options.h and the ONE place the library reads its environment.
#include "fh_options.h"

#include <cassert>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace fh {
namespace {

struct OptDef {
    const char *name, *help;
};
// (kept sorted by subject; README.md's table is this list)
const OptDef DEFS[] = {
    // --- traces (stderr) ---
    {"trace", "per-launch / per-phase timings and decisions on stderr"},
    {"trace_pargz", "the parallel gzip reader's chunk decisions on stderr"},
    {"gz_times", "per-chunk timestamps of the device gzip launch (tools/gz_bench.py)"},
    // --- which sketch kernel runs (all exact; A/B measurements and tests) ---
    {"no_seg", "segment kernels off: every block through the tile kernels"},
    {"seg_stride", "every block through the segment kernels with this stride, whatever its records are (tests)"},
    {"seg_probe_min", "blocks from this many bytes on are probed for a record stride (default 16 MiB)"},
    {"seg_probe_wait_min", "a handle's first block of at least this many bytes waits for its probe (default 256 MiB)"},
    {"seg_pull_pos", "positions a segment-kernel pull takes at most (default 32 K)"},
    {"seg_ragged", "0 = never the work-item form of the segment kernels (records of many lengths, k = 25, 27..32), 1 = for every block without a stride (tests); default: where the probe of a block finds a record end in every 256 bytes or more"},
    {"unit_tiles", "tiles per queue unit (0 = by size)"},
    {"no_static_units", "no statically dealt first units: every unit through the queue"},
    {"waves_per_cu", "persistent waves per compute unit (default 16)"},
    {"read_first", "0 / 1: the admit path never / always reads an entry before its atomics (default: by observed novelty)"},
    // --- scheduling of small sketches ---
    {"no_fast", "small sketches step by step: no fused epilogue, no deferred speculation"},
    {"no_hist", "no in-launch threshold refresh from the histogram of new hashes"},
    {"no_spec", "no speculative first block"},
    {"no_spec_rescale", "a failed speculation re-reads its block for everything above the guess at once"},
    {"no_reset_fold", "fh_finish's epilogue does not leave the handle reset"},
    {"max_range", "cap on k-mer start positions per range (tests: many ranges per push)"},
    {"spec_prefix_pos", "positions of a large first block that are sketched at the speculative threshold (default 32 M)"},
    // --- large sketches ---
    {"no_sample", "no sampling pre-pass for large sketches"},
    {"sample_min_pos", "smallest first block that is sampled (tests)"},
    {"sample_scale", "multiplies the sampled threshold (tests: < 1 forces the repair pass)"},
    {"sample_cap_scale", "scales the sample pass's cap threshold (measurement)"},
    {"sample_want", "target live-set size of the sampled threshold, in units of n (default 1.15)"},
    {"sample_run_tiles", "tiles per run of the sample (default: one run per wave)"},
    {"sample_one_in", "the sample takes one tile in this many (default 64)"},
    {"no_select", "device-wide prunes always sort instead of radix-selecting"},
    {"no_lazy_copyout", "a large sketch's wide columns cross to the host at fh_finish"},
    // --- buffers, pools, threads ---
    {"stage_bytes", "size of the staging buffers whatever fh_params says (tests: blocks span staging slices)"},
    {"pool", "sketchers fh_free keeps parked (default 64, 0 = none)"},
    {"pool_bytes", "device memory the parked sketchers may hold together (default: the smaller of 24 GiB and a tenth of the device)"},
    {"host_threads", "threads of the host-side passes over large results (1 = inline)"},
    {"no_numa_pin", "the library's block threads stay where the scheduler puts them"},
    {"read_threads", "threads the reads of one finch_sketch_files / finch_sketch_buffer call may use together"},
    {"bgzf_threads", "threads that inflate BGZF / gzip members on the host"},
    {"block_bytes", "size of the host parser's blocks (tests: records span blocks)"},
    {"max_launch", "fh_params.max_launch for the sketchers of the host layer"},
    // --- compressed input ---
    {"gz_chunk", "compressed bytes per chunk of the device gzip pass (tests: many chunks in a small input)"},
    {"bgzf_serial", "the BGZF inflate kernel decodes one symbol at a time (A/B)"},
    {"device_parse", "1 = text is split on the device or not at all, 0 = always the host parser (default: device first, host parser as the judge of what it refuses)"},
    {"device_inflate", "0 = compressed input is always inflated on the host"},
    {"device_gzip", "0 = plain gzip on the host, 1 = on the device or not at all"},
    {"gz_front", "many-file calls: every worker's gzip file through the device pass too (measurement)"},
    {"gzip_piece", "bytes per piece handed to the device gzip pass (default 16 MiB)"},
    {"pargz", "0 = the host gzip reader is the sequential one"},
    {"pargz_chunk", "compressed bytes per chunk of the parallel host gzip reader"},
    {"pargz_pool_mb", "memory the parallel host gzip reader may hold (MiB)"},
    {"zlib_inflate", "1 = zlib's inflate instead of the library's own (A/B)"},
    // --- batches of files ---
    {"no_small_sketcher", "no final_size-sized sketcher for unfiltered oversketched Mash input"},
    {"small_fasta_host", "0 = small FASTA files are split on the device, not packed by the worker"},
    {"file_batch", "0 = every file of a finch_sketch_files batch through a sketcher of its own (no many-per-launch groups)"},
    {"batch_two_bit", "0 = a group's files cross the link as bytes, not in the two-bit form (fh_batch_submit_packed)"},
    {"batch_read_piece", "bytes of a file a worker reads and packs at a time (default 256 KiB: stays in the core's L2; tests: many pieces)"},
    {"pack_scalar", "the two-bit packer's form (tests): 1 = portable, 2 = two passes with AVX2 (no BMI2); default: one pass with AVX2 + BMI2 where the CPU has them"},
    // --- FASTQ text in host memory ---
    {"fastq_host_strip", "0 = FASTQ text always goes to the device-side splitter; 1 = stripped on the host whatever the read threads (default: from 8 read threads on)"},
    {"fastq_strip_chunk", "bytes of text per chunk of the host-side FASTQ strip (tests: many chunks)"},
};
constexpr int N_OPTS = (int)(sizeof(DEFS) / sizeof(DEFS[0]));

std::mutex g_mu;
// values live for ever (a handful of short strings per process): a pointer handed out by cfg() stays valid whatever is set later
std::vector<std::string *> g_keep;
const char *g_explicit[N_OPTS], *g_env[N_OPTS];
std::string g_env_seen;
bool g_env_any = false;

int index_of(const char *name) {
    for (int i = 0; i < N_OPTS; ++i)
        if (strcmp(DEFS[i].name, name) == 0) return i;
    return -1;
}

const char *keep(const std::string &v) {
    g_keep.push_back(new std::string(v));
    return g_keep.back()->c_str();
}

// THE place the environment is read: FH_DEBUG="name=value,name=value" (separators: comma, space, semicolon).  Looked at on
// every query -- one getenv and one string compare -- so that a process that changes the variable (tests do) is followed.
void refresh_env() {
    const char *e = getenv("FH_DEBUG");
    if (!e) e = "";
    if (g_env_seen == e && (g_env_any || *e == 0)) return;
    g_env_seen = e;
    g_env_any = true;
    for (int i = 0; i < N_OPTS; ++i) g_env[i] = nullptr;
    std::string s(e);
    size_t p = 0;
    while (p < s.size()) {
        size_t q = s.find_first_of(", ;", p);
        if (q == std::string::npos) q = s.size();
        if (q > p) {
            const std::string item = s.substr(p, q - p);
            const size_t eq = item.find('=');
            const std::string name = item.substr(0, eq), val = eq == std::string::npos ? "1" : item.substr(eq + 1);
            const int i = index_of(name.c_str());
            if (i >= 0) g_env[i] = keep(val); // (unknown names are ignored: FH_DEBUG of another version)
        }
        p = q + 1;
    }
}

} // namespace

const char *cfg(const char *name) {
    std::lock_guard<std::mutex> g(g_mu);
    const int i = index_of(name);
    assert(i >= 0 && "cfg(): no such option -- add it to fh_options.cpp");
    if (i < 0) return nullptr;
    if (g_explicit[i]) return g_explicit[i];
    refresh_env();
    return g_env[i];
}

bool cfg_on(const char *name) {
    const char *v = cfg(name);
    return v && v[0] != '0';
}

uint64_t cfg_u64(const char *name, uint64_t dflt) {
    const char *v = cfg(name);
    return v ? (uint64_t)strtoull(v, nullptr, 10) : dflt;
}

int cfg_assign(const char *name, const char *value) {
    if (!name) return -1;
    std::lock_guard<std::mutex> g(g_mu);
    const int i = index_of(name);
    if (i < 0) return -1;
    g_explicit[i] = value ? keep(value) : nullptr;
    return 0;
}

bool cfg_known(const char *name) { return name && index_of(name) >= 0; }

const char *cfg_list() {
    static const std::string all = [] {
        std::string s;
        for (int i = 0; i < N_OPTS; ++i) s += std::string(DEFS[i].name) + "\t" + DEFS[i].help + "\n";
        return s;
    }();
    return all.c_str();
}

} // namespace fh
