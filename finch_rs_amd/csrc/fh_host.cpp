// fh_host.cpp -- host-side mirror of finch's library entry points for the accelerated path
// (C ABI in include/finch_host.h).  Everything per-base happens on the device through the fh_* ABI;
// this file parses FASTA/FASTQ, stages record bytes, applies the O(n) filters and serialises.
//
// Reference items mirrored (relative to the finch-rs tree) are cited at each function.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <functional>
#include <future>
#include <thread>
#include <vector>

#include "fh_host_model.h"
#include "fh_options.h"
#include "fh_inflate.h"
#include "fh_pargz.h"
#include "fh_strip.h"
#include "fh_fqstrip.h"
#include "fh_pack2.h"

// job(t) for t = 0 .. n - 1, one thread each (the caller's runs job(0)).  A thread that cannot be created (EAGAIN under a
// thread limit) must not take the process down -- a vector of joinable threads that unwinds calls std::terminate -- so its
// share runs on the calling thread instead.
template <class J>
static void fork_join(unsigned n, J job) {
    std::vector<std::thread> th;
    th.reserve(n ? n - 1 : 0); // (before the first thread exists: may throw freely)
    unsigned started = 1;
    try {
        for (; started < n; ++started) th.emplace_back(job, started);
    } catch (...) {
    }
    if (n) job(0u);
    for (unsigned t = started; t < n; ++t) job(t);
    for (auto &x : th) x.join();
}

// Up to 31 helper threads parked on a condition variable between teams.  run(n, job, mine): job(0..n-1) on n helpers while
// the caller runs mine(); returns false -- nothing was run -- if the pool is in use or cannot have n threads (the caller then
// starts threads of its own).  The helpers live as long as the process.
class TeamPool {
public:
    static TeamPool &instance() {
        static TeamPool *p = new TeamPool; // (never destroyed: its threads may outlive static destructors)
        return *p;
    }
    template <class J, class M>
    bool run(unsigned n, J job, M mine) {
        if (n == 0 || n > MAX) return false;
        if (!busy_.try_lock()) return false;
        std::function<void(unsigned)> fn = job;
        {
            std::unique_lock<std::mutex> lk(mu_);
            try {
                while (threads_ < n) {
                    std::thread(&TeamPool::worker, this, threads_).detach();
                    ++threads_;
                }
            } catch (...) {
                busy_.unlock();
                return false;
            }
            job_ = &fn;
            want_ = n;
            done_ = 0;
            ++gen_;
        }
        cv_.notify_all();
        mine();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return done_ == want_; });
            job_ = nullptr;
        }
        busy_.unlock();
        return true;
    }

private:
    static constexpr unsigned MAX = 31;
    void worker(unsigned id) {
        unsigned seen = 0;
        for (;;) {
            std::function<void(unsigned)> *j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (id >= want_) continue; // (a smaller team than there are helpers)
                j = job_;
            }
            (*j)(id);
            {
                std::lock_guard<std::mutex> g(mu_);
                ++done_;
            }
            cv_done_.notify_one();
        }
    }
    std::mutex busy_, mu_;
    std::condition_variable cv_, cv_done_;
    std::function<void(unsigned)> *job_ = nullptr;
    unsigned threads_ = 0, want_ = 0, done_ = 0, gen_ = 0;
};


namespace finch {
using fh::cfg;
using fh::cfg_on;
using fh::cfg_u64;

thread_local std::string g_host_err;

int hfail(int code, const char *fmt, ...) {
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_host_err = buf;
    return code;
}

// statistics.rs:30-47
template <class KC>
static std::vector<uint64_t> hist(const std::vector<KC> &sketch) {
    uint64_t max_count = 0;
    for (const auto &k : sketch) max_count = std::max<uint64_t>(max_count, k.count);
    std::vector<uint64_t> counts(max_count, 0);
    // (a count of 0 cannot come out of a sketcher -- push() starts every entry at 1, mash.rs:53 -- and the C ABI
    // rejects caller-supplied arrays that hold one; skipped here as well so that no input can index below the array)
    for (const auto &k : sketch)
        if (k.count) counts[k.count - 1] += 1;
    return counts;
}

// filtering.rs:154-195, from the histogram on
static uint32_t guess_filter_threshold_hist(const std::vector<uint64_t> &hist_data, double filter_level);
template <class KC>
static uint32_t guess_filter_threshold(const std::vector<KC> &sketch, double filter_level) {
    return guess_filter_threshold_hist(hist(sketch), filter_level);
}
static uint32_t guess_filter_threshold_hist(const std::vector<uint64_t> &hist_data, double filter_level) {
    uint64_t total = 0;
    for (size_t i = 0; i < hist_data.size(); ++i) total += (uint64_t)(i + 1) * hist_data[i];
    const double total_counts = (double)total;
    const double cutoff_amt = filter_level * total_counts;
    size_t wgt_cutoff = 0;
    uint64_t cum_count = 0;
    for (uint64_t count : hist_data) {
        cum_count += (uint64_t)wgt_cutoff * count;
        if ((double)cum_count > cutoff_amt) break;
        wgt_cutoff += 1;
    }
    if (wgt_cutoff == 0) return 1;
    const size_t win_size = std::max<size_t>(1, wgt_cutoff / 20);
    uint64_t sum = 0;
    for (size_t i = 0; i < win_size; ++i) sum += hist_data[i];
    uint64_t lowest_val = sum;
    size_t lowest_idx = win_size - 1;
    for (size_t i = 0, j = win_size; j < wgt_cutoff; ++i, ++j) {
        if (sum <= lowest_val) {
            lowest_val = sum;
            lowest_idx = j;
        }
        sum -= hist_data[i];
        sum += hist_data[j];
    }
    return (uint32_t)lowest_idx + 1;
}

// filtering.rs:413-432
// (in place: a 2 M-record oversketch is filtered without a second copy of it)
template <class KC>
static void filter_strands(std::vector<KC> &sketch, double ratio_cutoff) {
    sketch.erase(std::remove_if(sketch.begin(), sketch.end(),
                                [&](const KC &kmer) {
                                    if (kmer.count < 16) return false;
                                    // (extra_count <= count for everything a sketcher emits and everything the C ABI lets in)
                                    const uint32_t lowest = std::min(kmer.extra_count, kmer.count - std::min(kmer.extra_count, kmer.count));
                                    return !(((double)lowest / (double)kmer.count) >= ratio_cutoff);
                                }),
                 sketch.end());
}

// filtering.rs:329-343
// `keep_at_most`: the caller truncates to this many records afterwards (process_post_filter on a Mash sketch): records are
// ascending and the test is per record, so nothing behind that many survivors needs to be looked at
template <class KC>
static void filter_abundance(std::vector<KC> &sketch, bool has_lo, uint32_t lo, bool has_hi, uint32_t hi, size_t keep_at_most = SIZE_MAX) {
    const uint32_t lo_t = has_lo ? lo : 0u, hi_t = has_hi ? hi : UINT32_MAX;
    size_t m = 0;
    for (size_t i = 0; i < sketch.size() && m < keep_at_most; ++i)
        if (lo_t <= sketch[i].count && sketch[i].count <= hi_t) {
            if (m != i) sketch[m] = std::move(sketch[i]);
            ++m;
        }
    sketch.resize(m);
}

// FilterParams::filter_counts (filtering.rs:60-87); updates `fp` like the reference updates self
template <class KC>
static std::vector<KC> filter_counts(finch_filter_params &fp, std::vector<KC> hashes, size_t keep_at_most = SIZE_MAX) { // (by value: callers that are done with it move it in)
    const bool filter_on = fp.filter_on == 1;
    std::vector<KC> filtered = std::move(hashes);
    if (filter_on && fp.strand_filter > 0.0) filter_strands(filtered, fp.strand_filter);
    if (filter_on && fp.err_filter > 0.0) {
        const uint32_t cutoff = guess_filter_threshold(filtered, fp.err_filter);
        if (fp.has_abun_lo) {
            if (cutoff > fp.abun_lo) fp.abun_lo = cutoff;
        } else {
            fp.has_abun_lo = 1;
            fp.abun_lo = cutoff;
        }
    }
    if (filter_on && (fp.has_abun_lo || fp.has_abun_hi))
        filter_abundance(filtered, fp.has_abun_lo, fp.abun_lo, fp.has_abun_hi, fp.abun_hi, keep_at_most);
    return filtered;
}

// SketchParams::process_post_filter (mod.rs:115-128)
template <class KC>
static int process_post_filter(const finch_sketch_params &sp, std::vector<KC> &kmers, const std::string &name) {
    if (sp.kind == 0) {
        if (kmers.size() > sp.final_size) kmers.resize(sp.final_size);
        if (!sp.no_strict && kmers.size() < sp.final_size)
            return hfail(FH_ERR_INVALID, "%s had too few kmers (%zu) to sketch", name.c_str(), kmers.size());
    }
    return FH_OK;
}

// distance.rs:66-126 (raw_distance)
static void raw_distance(const uint64_t *q, size_t nq, const uint64_t *r, size_t nr, double scale, double &containment,
                         double &jaccard, uint64_t &common_out, uint64_t &total_out) {
    size_t i = 0, j = 0;
    uint64_t common = 0;
    while (i < nq && j < nr) {
        if (q[i] < r[j]) i++;
        else if (q[i] > r[j]) j++;
        else {
            common++;
            i++;
            j++;
        }
    }
    if (scale > 0.) {
        // u64::MAX / scale.recip() as u64   (saturating float->int cast)
        const double rec = 1.0 / scale;
        uint64_t irec = rec >= 18446744073709551616.0 ? UINT64_MAX : (rec <= 0.0 || rec != rec ? 0 : (uint64_t)rec);
        const uint64_t max_hash = irec ? UINT64_MAX / irec : UINT64_MAX;
        while (i < nq && q[i] < max_hash) i++;
        while (j < nr && r[j] < max_hash) j++;
    }
    containment = j == 0 ? 0. : (double)common / (double)j;
    const uint64_t total = (uint64_t)i - common + (uint64_t)j;
    jaccard = total == 0 ? 1. : (double)common / (double)total;
    common_out = common;
    total_out = total;
}

// distance.rs:136-157 (old_distance); the reference indexes query_sketch[0] unconditionally
static int old_distance(const uint64_t *q, size_t nq, const uint64_t *r, size_t nr, double &containment, double &jaccard,
                        uint64_t &common_out, uint64_t &total_out) {
    if (nq == 0 && nr > 0) return hfail(FH_ERR_INVALID, "old_distance: empty query sketch");
    size_t i = 0;
    uint64_t common = 0, total = 0;
    for (size_t t = 0; t < nr; ++t) {
        while (q[i] < r[t] && i < nq - 1) i++;
        if (q[i] == r[t]) common++;
        total++;
    }
    containment = (double)common / (double)total;
    jaccard = (double)common / (double)(common + 2 * (total - common));
    common_out = common;
    total_out = total;
    return FH_OK;
}

// ---------------------------------------------------------------------------------------------
// byte sources: plain memory / FILE*, optionally through zlib (needletail sniffs 1F 8B)
// ---------------------------------------------------------------------------------------------
struct ByteSource {
    virtual ~ByteSource() {}
    virtual size_t read(uint8_t *dst, size_t cap) = 0; // 0 = EOF
    virtual bool failed() const { return false; }
    virtual bool can_rewind() const { return false; } // regular files, memory
    virtual bool rewind() { return false; }           // back to the first byte
    virtual unsigned threads_hint() const { return 1; } // host threads the reader of this source may use
    virtual uint64_t remaining_hint() const { return UINT64_MAX; } // bytes still to come, if the source knows (files, memory)
    // the source's WHOLE content as one contiguous read-only range, if it is memory (wherever its read position is)
    virtual bool whole_view(const uint8_t **, size_t *) const { return false; }
};

static unsigned read_threads_total(const char *env);

struct MemSource : ByteSource {
    const uint8_t *p;
    size_t n, off = 0;
    unsigned n_thr; // threads a large read may use (finch_sketch_buffer: FINCH_READ_THREADS)
    MemSource(const uint8_t *p_, size_t n_, unsigned read_threads = 1) : p(p_), n(n_), n_thr(std::max(1u, read_threads)) {}
    bool can_rewind() const override { return true; }
    bool rewind() override {
        off = 0;
        return true;
    }
    // A large read is a copy into a pinned staging buffer: one thread moves ~10 GB/s, less than half of what the PCIe link
    // behind it takes, so the 64 MiB chunks of the device-side text paths are copied by a few threads (as FileSource reads).
    size_t read(uint8_t *dst, size_t cap) override {
        const size_t m = std::min(cap, n - off);
        const size_t PAR_MIN = (size_t)16 << 20;
        if (m < PAR_MIN || n_thr < 2) {
            memcpy(dst, p + off, m);
        } else {
            const size_t per = ((m + n_thr - 1) / n_thr + 4095) & ~(size_t)4095;
            const uint8_t *src = p + off;
            auto job = [=](unsigned t) {
                const size_t lo = std::min(m, (size_t)t * per), hi = std::min(m, lo + per);
                if (lo < hi) memcpy(dst + lo, src + lo, hi - lo);
            };
            fork_join(n_thr, job);
        }
        off += m;
        return m;
    }
    unsigned threads_hint() const override { return n_thr; }
    uint64_t remaining_hint() const override { return n - off; }
    bool whole_view(const uint8_t **pp, size_t *nn) const override {
        *pp = p;
        *nn = n;
        return true;
    }
};

struct FileSource : ByteSource {
    FILE *f;
    bool own;
    unsigned n_thr; // threads a large read may use (the caller divides FINCH_READ_THREADS among its workers)
    FileSource(FILE *f_, bool own_, unsigned read_threads = 1) : f(f_), own(own_), n_thr(std::max(1u, read_threads)) {}
    ~FileSource() override {
        if (own && f) fclose(f);
    }
    // Large reads from a regular file (the 64 MiB staging chunks of the device-side text paths) are split over a few
    // threads: one thread copies ~7 GB/s out of the page cache into pinned memory, which is what bounded a FASTQ file
    // end to end (1.25 GB: 6.8 GB/s of text with one thread, 15.5 with four, 19.9 with eight); the stdio position is carried along so that small reads and rewind() keep working.
    size_t read(uint8_t *dst, size_t cap) override {
        const size_t PAR_MIN = (size_t)16 << 20;
        struct stat sb;
        if (cap < PAR_MIN || n_thr < 2 || !f || fstat(fileno(f), &sb) != 0 || !S_ISREG(sb.st_mode)) return fread(dst, 1, cap, f);
        const off_t pos = ftello(f);
        if (pos < 0 || pos >= sb.st_size) return fread(dst, 1, cap, f);
        const size_t want = (size_t)std::min<uint64_t>(cap, (uint64_t)(sb.st_size - pos));
        if (want < PAR_MIN) return fread(dst, 1, cap, f);
        const int fd = fileno(f);
        const size_t per = ((want + n_thr - 1) / n_thr + 4095) & ~(size_t)4095;
        std::vector<size_t> got(n_thr, 0);
        auto job = [&](unsigned t) {
            const size_t lo = std::min(want, (size_t)t * per), hi = std::min(want, lo + per);
            size_t done = 0;
            while (lo + done < hi) {
                const ssize_t r = pread(fd, dst + lo + done, hi - lo - done, pos + (off_t)(lo + done));
                if (r <= 0) break; // error or the file shrank: the caller sees a short read
                done += (size_t)r;
            }
            got[t] = done;
        };
        fork_join(n_thr, job);
        size_t total = 0; // contiguous prefix that was read
        for (unsigned t = 0; t < n_thr; ++t) {
            const size_t lo = std::min(want, (size_t)t * per), hi = std::min(want, lo + per);
            total += got[t];
            if (got[t] < hi - lo) break;
        }
        if (fseeko(f, pos + (off_t)total, SEEK_SET) != 0) return 0;
        return total;
    }
    bool can_rewind() const override {
        struct stat sb;
        return own && f && fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode);
    }
    bool rewind() override { return can_rewind() && fseek(f, 0, SEEK_SET) == 0; }
    unsigned threads_hint() const override { return n_thr; }
    uint64_t remaining_hint() const override {
        struct stat sb;
        if (!f || fstat(fileno(f), &sb) != 0 || !S_ISREG(sb.st_mode)) return UINT64_MAX;
        const off_t pos = ftello(f);
        return pos < 0 || pos > sb.st_size ? UINT64_MAX : (uint64_t)(sb.st_size - pos);
    }
};

// prepends already-consumed sniff bytes
struct PrefixedSource : ByteSource {
    std::vector<uint8_t> prefix;
    size_t off = 0;
    std::unique_ptr<ByteSource> inner;
    size_t read(uint8_t *dst, size_t cap) override {
        if (off < prefix.size()) {
            const size_t m = std::min(cap, prefix.size() - off);
            memcpy(dst, prefix.data() + off, m);
            off += m;
            return m;
        }
        return inner->read(dst, cap);
    }
    bool can_rewind() const override { return inner->can_rewind(); }
    bool rewind() override {
        if (!inner->rewind()) return false;
        off = prefix.size(); // the inner source delivers the sniffed bytes itself again
        return true;
    }
    unsigned threads_hint() const override { return inner->threads_hint(); }
    bool failed() const override { return inner->failed(); }
    uint64_t remaining_hint() const override {
        const uint64_t r = inner->remaining_hint();
        return r == UINT64_MAX ? r : r + (prefix.size() - std::min(off, prefix.size()));
    }
    bool whole_view(const uint8_t **pp, size_t *nn) const override { return inner->whole_view(pp, nn); } // (the prefix is its first bytes)
};

struct GzSource : ByteSource {
    std::unique_ptr<ByteSource> inner;
    z_stream zs{};
    std::vector<uint8_t> inbuf;
    bool eof = false, bad = false, init = false, mid_member = false;
    explicit GzSource(std::unique_ptr<ByteSource> in) : inner(std::move(in)), inbuf(1 << 20) {
        init = inflateInit2(&zs, 15 + 32) == Z_OK; // gzip/zlib auto-detect
        bad = !init;
    }
    ~GzSource() override {
        if (init) inflateEnd(&zs);
    }
    bool failed() const override { return bad; }
    bool can_rewind() const override { return init && inner->can_rewind(); }
    bool rewind() override { // inflate again from the first byte
        if (!can_rewind() || !inner->rewind() || inflateReset(&zs) != Z_OK) return false;
        zs.avail_in = 0;
        eof = bad = mid_member = false;
        return true;
    }
    size_t read(uint8_t *dst, size_t cap) override {
        if (eof || bad) return 0;
        zs.next_out = dst;
        zs.avail_out = (uInt)std::min<size_t>(cap, 1u << 30);
        while (zs.avail_out > 0) {
            if (zs.avail_in == 0) {
                const size_t got = inner->read(inbuf.data(), inbuf.size());
                if (got == 0) {
                    eof = true;
                    if (mid_member) bad = true; // the input ends inside a member: truncated (flate2: UnexpectedEof)
                    break;
                }
                zs.next_in = inbuf.data();
                zs.avail_in = (uInt)got;
            }
            mid_member = true;
            const int rc = inflate(&zs, Z_NO_FLUSH);
            if (rc == Z_STREAM_END) {
                mid_member = false;
                // concatenated gzip members (bgzip) continue with a fresh header
                if (zs.avail_in == 0) {
                    const size_t got = inner->read(inbuf.data(), inbuf.size());
                    if (got == 0) {
                        eof = true;
                        break;
                    }
                    zs.next_in = inbuf.data();
                    zs.avail_in = (uInt)got;
                }
                inflateReset(&zs);
                continue;
            }
            if (rc != Z_OK && rc != Z_BUF_ERROR) {
                bad = true;
                break;
            }
        }
        return (size_t)(zs.next_out - dst);
    }
};

// gzip through fh_inflate.h's decoder (the default; FINCH_ZLIB_INFLATE=1 selects the zlib-based GzSource above for A/B runs).
// Same contract: concatenated members, every member's CRC-32 and ISIZE checked, input that ends inside a member or
// garbage after a member is an error (flate2 / zlib behaviour), never short output.
struct FastGzSource : ByteSource {
    std::unique_ptr<ByteSource> inner;
    std::unique_ptr<inf::Decoder> dec;
    std::vector<uint8_t> inbuf; // compressed bytes [in_lo, in_hi), 16 bytes of zero padding behind in_hi
    size_t in_lo = 0, in_hi = 0;
    bool in_eof = false;
    // Large requests are decoded straight into the caller's buffer (the 64 MiB pinned staging buffers of the device-side
    // text paths, the parser's 8 MiB buffer); what a request leaves over (< one match) and small requests go through
    // `win`.  Either way the last 32 KiB of the stream so far are kept in `hist`: that is where a match may reach when it
    // starts before the buffer at hand.
    static constexpr size_t HIST = 32768, WIN = (size_t)1 << 18, DIRECT_MIN = (size_t)1 << 16, CRC_SLICE = (size_t)1 << 18;
    std::vector<uint8_t> win, hist;
    size_t w_have = 0, w_out = 0, hist_len = 0;
    uint64_t member_out = 0; // bytes of the current member produced so far (a match may not reach before its start)
    enum { GZ_HEADER, BODY, TRAILER, END } st = GZ_HEADER;
    uint32_t crc = 0;
    bool bad = false, any_member = false;

    explicit FastGzSource(std::unique_ptr<ByteSource> in)
        : inner(std::move(in)), dec(new inf::Decoder()), inbuf(((size_t)1 << 20) + 16), win(WIN), hist(HIST) {}
    bool failed() const override { return bad; }
    bool can_rewind() const override { return inner->can_rewind(); }
    bool rewind() override {
        if (!inner->rewind()) return false;
        in_lo = in_hi = 0;
        in_eof = bad = any_member = false;
        w_have = w_out = hist_len = 0;
        member_out = 0;
        st = GZ_HEADER;
        return true;
    }
    unsigned threads_hint() const override { return inner->threads_hint(); }

    // more compressed bytes behind [in_lo, in_hi); false at the end of the input
    bool refill() {
        if (in_eof) return false;
        if (in_lo) {
            memmove(inbuf.data(), inbuf.data() + in_lo, in_hi - in_lo);
            in_hi -= in_lo;
            in_lo = 0;
        }
        const size_t cap = inbuf.size() - 16;
        if (in_hi == cap) return false; // (cannot happen: no single item is that long)
        const size_t got = inner->read(inbuf.data() + in_hi, cap - in_hi);
        if (got == 0) {
            in_eof = true;
            return false;
        }
        in_hi += got;
        memset(inbuf.data() + in_hi, 0, 16);
        return true;
    }
    // RFC 1952 member header at in_lo: 1 = parsed, 0 = need more input, -1 = not a gzip header
    int parse_header() {
        const uint8_t *p = inbuf.data() + in_lo;
        const size_t n = in_hi - in_lo;
        if (n < 10) return 0;
        if (p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 || (p[3] & 0xE0)) return -1;
        const uint8_t flg = p[3];
        size_t off = 10;
        if (flg & 4) { // FEXTRA
            if (n < off + 2) return 0;
            const size_t xlen = p[off] | ((size_t)p[off + 1] << 8);
            off += 2;
            if (n < off + xlen) return 0;
            off += xlen;
        }
        for (int bit : {8, 16}) // FNAME, FCOMMENT: zero-terminated
            if (flg & bit) {
                const void *z = memchr(p + off, 0, n - off);
                if (!z) return (n > inbuf.size() - 64) ? -1 : 0;
                off = (size_t)((const uint8_t *)z - p) + 1;
            }
        if (flg & 2) { // FHCRC
            if (n < off + 2) return 0;
            off += 2;
        }
        in_lo += off;
        return 1;
    }
    void push_hist(const uint8_t *p, size_t n) { // the stream went on by p[0, n)
        if (n >= HIST) {
            memcpy(hist.data(), p + n - HIST, HIST);
            hist_len = HIST;
            return;
        }
        const size_t keep = std::min(hist_len, HIST - n);
        memmove(hist.data(), hist.data() + hist_len - keep, keep);
        memcpy(hist.data() + keep, p, n);
        hist_len = keep + n;
    }
    // inflate into buf[0, room): as much as fits / as the input holds; the bytes join the stream (hist) on return
    size_t produce_into(uint8_t *buf, size_t room) {
        uint8_t *op = buf;
        uint8_t *const oe = buf + room;
        const uint8_t *floor = buf;                                           // matches reach back to here in this buffer ...
        size_t ext = (size_t)std::min<uint64_t>(hist_len, member_out);        // ... and this far into hist before it
        for (;;) {
            if (st == END || bad) break;
            if (st == GZ_HEADER) {
                if (in_hi == in_lo && !refill()) { // clean end of input between members
                    if (!any_member) bad = true;  // (an empty file is not gzip)
                    st = END;
                    break;
                }
                const int r = parse_header();
                if (r < 0) {
                    bad = true;
                    break;
                }
                if (r == 0) {
                    if (!refill()) bad = true; // the input ends inside a header
                    continue;
                }
                dec->reset();
                crc = 0;
                member_out = 0;
                floor = op;
                ext = 0;
                any_member = true;
                st = BODY;
            }
            if (st == BODY) {
                dec->ext_end = hist.data() + hist_len;
                dec->ext_len = ext;
                const uint8_t *ip = inbuf.data() + in_lo;
                uint8_t *const before = op;
                // (in slices of 256 KiB, so that the checksum reads what the decoder has just written from the cache: the
                // CRC of a multi-megabyte span long after it was produced costs a tenth of the inflate time)
                uint8_t *const slice_end = (size_t)(oe - op) > CRC_SLICE + inf::OUT_MARGIN ? op + CRC_SLICE + inf::OUT_MARGIN : oe;
                inf::Status s = dec->run(ip, inbuf.data() + in_hi, op, slice_end, floor);
                if (s == inf::NEED_OUTPUT && slice_end != oe) s = inf::MORE_SLICES;
                in_lo = (size_t)(ip - inbuf.data());
                const size_t fresh = (size_t)(op - before);
                if (fresh) {
                    crc = inf::crc32_fast(crc, before, fresh);
                    member_out += fresh;
                }
                if (s == inf::BAD) {
                    bad = true;
                } else if (s == inf::STREAM_END) {
                    in_lo -= (size_t)(dec->bitcnt >> 3); // whole bytes still in the bit buffer belong to the trailer
                    st = TRAILER;
                } else if (s == inf::NEED_INPUT) {
                    if (!refill()) bad = true; // the input ends inside a member: truncated
                } else if (s == inf::NEED_OUTPUT) {
                    break;
                } // (MORE_SLICES: go on where the slice ended)
                continue;
            }
            if (st == TRAILER) {
                if (in_hi - in_lo < 8) {
                    if (!refill()) bad = true;
                    continue;
                }
                const uint8_t *p = inbuf.data() + in_lo;
                const uint32_t want_crc = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                const uint32_t want_len = p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
                in_lo += 8;
                if (want_crc != crc || want_len != (uint32_t)member_out) bad = true;
                st = GZ_HEADER;
            }
        }
        const size_t n = (size_t)(op - buf);
        if (n) push_hist(buf, n);
        return n;
    }
    size_t read(uint8_t *dst, size_t cap) override {
        size_t n = 0;
        while (n < cap) {
            if (w_out < w_have) {
                const size_t m = std::min(cap - n, w_have - w_out);
                memcpy(dst + n, win.data() + w_out, m);
                w_out += m;
                n += m;
                continue;
            }
            if (st == END || bad) break;
            if (cap - n >= DIRECT_MIN) {
                const size_t got = produce_into(dst + n, cap - n);
                n += got;
                if (got == 0 && st != END && !bad) { // the next symbol needs more room than is left: through the window
                    w_have = produce_into(win.data(), win.size());
                    w_out = 0;
                    if (w_have == 0) break;
                }
            } else {
                w_have = produce_into(win.data(), win.size());
                w_out = 0;
                if (w_have == 0) break;
            }
        }
        return n;
    }
};

// Threads one input may use for large reads and BGZF members: FINCH_READ_THREADS, else a sixteenth of the machine's
// hardware threads (a GPU node has ~32 cores per GPU and other ranks beside this one), between 8 and 16.
// hardware threads this process may really use: the affinity mask capped by the cgroup's CPU quota (a container that sees 256
// processors and is granted 16 runs 24 busy workers slower than 16)
static unsigned usable_cpus() {
    static const unsigned v = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
            char q[64];
            long long per = 0;
            if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max(1ll, atoll(q) / per));
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // cgroup v1
            long long quota = -1, per = 0;
            const bool ok = fscanf(g, "%lld", &quota) == 1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (ok && fscanf(h, "%lld", &per) == 1 && quota > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max(1ll, quota / per));
                fclose(h);
            }
        }
        return n;
    }();
    return v;
}

static unsigned read_threads_total(const char *env) {
    if (env) return (unsigned)std::min(64, std::max(1, atoi(env)));
    const unsigned hw = std::thread::hardware_concurrency();
    return std::min(16u, std::max(8u, hw / 16u));
}
static bool use_zlib_inflate() {
    static const bool v = [] {
        const char *e = cfg("zlib_inflate");
        return e && e[0] == '1';
    }();
    return v;
}

// BGZF (bgzip / htslib): a series of gzip members of at most 64 KiB each whose header states the member's own size
// (extra subfield 'B','C').  The members are independent deflate streams, so a batch of them is inflated by several
// threads at once -- what a single zlib stream cannot offer (plain gzip stays at inflate speed, ~0.3 Gbases/s).
// Every member's CRC-32 and length are checked as zlib's gzip wrapper would; a member without the subfield hands the
// rest of the input to the sequential GzSource.
// A byte buffer whose pages are not touched until they are used (a std::vector would zero all of it first).
struct RawBuf {
    uint8_t *p = nullptr;
    size_t n = 0;
    explicit RawBuf(size_t bytes) : p((uint8_t *)malloc(bytes)), n(bytes) {
        if (!p) throw std::bad_alloc();
    }
    ~RawBuf() { free(p); }
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    uint8_t *data() const { return p; }
    size_t size() const { return n; }
    void swap(RawBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
};

// (defined behind ParGzSource) the reader of gzip input that is not BGZF, given `threads` to work with
static std::unique_ptr<ByteSource> make_gzip_reader(std::unique_ptr<ByteSource> in, unsigned threads);

struct BgzfSource : ByteSource {
    std::unique_ptr<ByteSource> inner;
    unsigned n_thr;
    RawBuf cbuf; // compressed bytes [c_lo, c_hi) not yet consumed
    size_t c_lo = 0, c_hi = 0;
    std::vector<uint8_t> obuf; // inflated bytes [o_lo, o_hi) not yet delivered
    size_t o_lo = 0, o_hi = 0;
    bool in_eof = false, bad = false;
    std::unique_ptr<ByteSource> tail; // the sequential gzip reader, once a member that is not BGZF turns up
    static constexpr size_t BATCH = 512; // members per round (<= 32 MiB inflated)

    bool confirmed = false;              // the first member was BGZF: the batch buffer is worth allocating
    BgzfSource(std::unique_ptr<ByteSource> in, unsigned threads) : inner(std::move(in)), n_thr(std::max(1u, threads)), cbuf(1 << 16) {}
    bool failed() const override { return bad || (tail && tail->failed()); }
    // (once another gzip reader has taken over it owns the input; rewound, it starts at the first byte of the file and
    // reads the BGZF members in front of its own as the gzip members they are)
    bool can_rewind() const override { return tail ? tail->can_rewind() : (inner && inner->can_rewind()); }
    bool rewind() override {
        if (!can_rewind()) return false;
        if (tail) {
            if (!tail->rewind()) return false;
            o_lo = o_hi = 0;
            bad = false;
            return true;
        }
        join_prefetch();
        if (!inner->rewind()) return false;
        c_lo = c_hi = o_lo = o_hi = 0;
        last_scan = 0;
        in_eof = bad = false;
        return true;
    }

    // The compressed bytes of the NEXT batch are read by a helper thread while this batch inflates: it appends behind
    // c_hi (the inflate threads only read below it), and refill() collects it before it looks at the buffer again.
    struct Prefetched { size_t got; bool eof; };
    std::future<Prefetched> pending;
    size_t last_scan = 0; // compressed bytes the previous batch took
    void join_prefetch() {
        if (!pending.valid()) return;
        const Prefetched r = pending.get();
        c_hi += r.got;
        if (r.eof) in_eof = true;
    }
    size_t batch_target() const { // compressed bytes worth having ahead: two batches at the ratio seen so far
        return std::min(cbuf.size() / 2, std::max<size_t>(2 * last_scan, (size_t)8 << 20));
    }
    void start_prefetch(size_t scan) {
        if (in_eof || n_thr < 2) return;
        const size_t ahead = c_hi - c_lo - scan, target = batch_target();
        if (ahead >= target) return;
        const size_t want = std::min(cbuf.size() - c_hi, target - ahead);
        if (want < ((size_t)1 << 20)) return; // no room behind c_hi: the next refill compacts and reads in line
        uint8_t *p = cbuf.data() + c_hi;
        ByteSource *src = inner.get();
        pending = std::async(std::launch::async, [p, want, src]() {
            Prefetched r{0, false};
            while (r.got < want) {
                const size_t g = src->read(p + r.got, want - r.got);
                if (g == 0) { r.eof = true; break; }
                r.got += g;
            }
            return r;
        });
    }

    // total size of the BGZF member starting at p (0 = not a BGZF member); needs 18 readable bytes
    static uint32_t member_size(const uint8_t *p, uint32_t *hdr_len) {
        if (p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 || p[3] != 4) return 0; // FLG must be exactly FEXTRA
        const uint32_t xlen = p[10] | ((uint32_t)p[11] << 8);
        if (xlen != 6 || p[12] != 'B' || p[13] != 'C' || p[14] != 2 || p[15] != 0) return 0; // what bgzip writes
        *hdr_len = 12 + xlen;
        return (p[16] | ((uint32_t)p[17] << 8)) + 1u;
    }
    bool fill_compressed(size_t need) { // make [c_lo, c_hi) hold at least `need` bytes if the input has them
        if (c_hi - c_lo >= need) return true;
        // (move the rest down as soon as that costs less than what has been consumed: the buffer is sized for the
        // worst case and most of it should never be touched)
        if (c_lo && (cbuf.size() - c_lo < need || c_hi - c_lo <= c_lo)) {
            memmove(cbuf.data(), cbuf.data() + c_lo, c_hi - c_lo);
            c_hi -= c_lo;
            c_lo = 0;
        }
        while (!in_eof && c_hi - c_lo < need) {
            const size_t got = inner->read(cbuf.data() + c_hi, std::min(cbuf.size() - c_hi, need - (c_hi - c_lo)));
            if (got == 0) in_eof = true;
            c_hi += got;
        }
        return c_hi - c_lo >= need;
    }
    // ---- the device-side inflate (fh_push_bgzf_fastq) takes the members as they are ----
    // First byte of the file's text without consuming anything; -1: first member not BGZF, empty or damaged.
    int peek_first_text_byte() {
        join_prefetch();
        if (!fill_compressed(18)) return -1;
        uint32_t hdr = 0;
        const uint32_t tot = member_size(cbuf.data() + c_lo, &hdr);
        if (tot < hdr + 10u || !fill_compressed(tot)) return -1;
        const uint8_t *p = cbuf.data() + c_lo;
        const uint32_t isize = p[tot - 4] | ((uint32_t)p[tot - 3] << 8) | ((uint32_t)p[tot - 2] << 16) | ((uint32_t)p[tot - 1] << 24);
        if (isize == 0 || isize > 65536u) return -1;
        std::vector<uint8_t> text(isize);
        std::unique_ptr<inf::Decoder> dec(new inf::Decoder());
        if (!inf::inflate_exact(*dec, p + hdr, tot - hdr - 8, text.data(), isize)) return -1;
        return text[0];
    }
    // ---- plain gzip on the device (fh_push_gzip_fastq) takes the DEFLATE bytes as they are ----
    // The file starts with a gzip member that is not BGZF: the length of its header (*hdr_len) and the first byte of its
    // text, nothing consumed.  -1: it does not, or nothing can be decoded from its first 64 KiB.
    int peek_plain_gzip(size_t *hdr_len) {
        join_prefetch();
        if (tail || c_lo) return -1; // (only for a reader that has handed out nothing yet)
        fill_compressed(std::min<size_t>(cbuf.size(), 65536));
        const uint8_t *p = cbuf.data();
        const size_t n = c_hi;
        if (n < 18 + 16) return -1;
        uint32_t bh = 0;
        if (member_size(p, &bh)) return -1; // BGZF: the other path
        if (p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 || (p[3] & 0xE0)) return -1;
        const uint8_t flg = p[3];
        size_t off = 10;
        if (flg & 4) {
            off += 2 + (p[off] | ((size_t)p[off + 1] << 8));
            if (off >= n) return -1;
        }
        for (int bit : {8, 16})
            if (flg & bit) {
                const void *z = memchr(p + off, 0, n - off);
                if (!z) return -1;
                off = (size_t)((const uint8_t *)z - p) + 1;
            }
        if (flg & 2) off += 2;
        if (off + 16 >= n) return -1;
        std::unique_ptr<inf::Decoder> dec(new inf::Decoder());
        dec->reset();
        uint8_t out[2048];
        const uint8_t *ip = p + off;
        uint8_t *op = out;
        const inf::Status st = dec->run(ip, p + n - 8, op, out + sizeof(out), out); // (8 readable bytes behind the end given)
        if (st == inf::BAD || op == out) return -1;
        *hdr_len = off;
        return out[0];
    }
    // the file's bytes as they are, from where the reader stands (what the probes above buffered first)
    size_t raw_read(uint8_t *dst, size_t cap) {
        join_prefetch();
        size_t n = 0;
        if (c_hi > c_lo) {
            n = std::min(cap, c_hi - c_lo);
            memcpy(dst, cbuf.data() + c_lo, n);
            c_lo += n;
            if (c_lo == c_hi) c_lo = c_hi = 0;
        }
        while (n < cap && !in_eof) {
            const size_t got = inner->read(dst + n, cap - n);
            if (got == 0) in_eof = true;
            n += got;
        }
        return n;
    }
    // Whole members into dst: a table of n records at the front (room for max_members), the members behind it exactly as
    // they lie in the file -- the file is read straight into dst, large reads split over the source's threads, and the
    // table points at the DEFLATE bytes between each member's header and trailer.  Members are taken while the table, dst
    // and the text they inflate to (text_budget; *budget_hit: that was what stopped it) have room; what was read beyond the
    // last one taken waits in cbuf for the next call.  *eof: the input ended with the last member taken.  false: a member
    // that is not BGZF, a truncated one, or one too large.
    uint64_t raw_text_seen = 0, raw_comp_seen = 0;
    bool raw_batch(uint8_t *dst, size_t dst_cap, uint32_t max_members, uint64_t text_budget, fh_bgzf_member *table, uint32_t *n_out,
                   uint64_t *bytes_out, uint64_t *text_out, bool *eof, bool *budget_hit) {
        *budget_hit = false;
        *eof = false;
        *n_out = 0;
        *bytes_out = *text_out = 0;
        join_prefetch();
        const size_t w0 = ((size_t)max_members * sizeof(fh_bgzf_member) + 255) & ~(size_t)255;
        if (dst_cap < w0 + 2 * 65536 + 64) return false;
        // what an earlier call (or the first-byte probe) read but did not hand out comes first
        size_t fill = c_hi - c_lo;
        if (fill > dst_cap - w0) return false;
        memcpy(dst + w0, cbuf.data() + c_lo, fill);
        c_lo = c_hi = 0;
        uint32_t n = 0;
        uint64_t text = 0;
        size_t off = 0; // bytes of dst + w0 consumed by the members taken
        bool stop = false;
        while (!stop) {
            // members completely in [off, fill)
            while (fill - off >= 18) {
                const uint8_t *p = dst + w0 + off;
                uint32_t hdr = 0;
                const uint32_t tot = member_size(p, &hdr);
                if (tot < hdr + 10u) return false;
                if (fill - off < tot) break;
                const uint32_t isize = p[tot - 4] | ((uint32_t)p[tot - 3] << 8) | ((uint32_t)p[tot - 2] << 16) | ((uint32_t)p[tot - 1] << 24);
                if (isize > 65536u) return false;
                if (text + isize > text_budget) {
                    *budget_hit = true;
                    stop = true;
                    break;
                }
                if (n == max_members) {
                    stop = true;
                    break;
                }
                table[n].in_off = (uint32_t)(w0 + off + hdr);
                table[n].in_len = tot - hdr - 8;
                table[n].out_off = (uint32_t)text;
                table[n].isize = isize;
                table[n].crc32 = p[tot - 8] | ((uint32_t)p[tot - 7] << 8) | ((uint32_t)p[tot - 6] << 16) | ((uint32_t)p[tot - 5] << 24);
                text += isize;
                n++;
                off += tot;
                raw_text_seen += isize;
                raw_comp_seen += tot;
            }
            if (stop) break;
            // more of the file: all dst has room for, in one read (a member needs at most 64 KiB + a header's worth)
            const size_t room = dst_cap - w0 - fill;
            if (in_eof || room < 65536 + 64) break;
            // (read about what the text budget will take, going by the members seen so far: what is read beyond it has to be
            // carried over to the next call)
            const double per_text = raw_text_seen ? (double)raw_comp_seen / (double)raw_text_seen : 0.35;
            const uint64_t text_left = text_budget > text ? text_budget - text : 0;
            const size_t want = (size_t)std::min<uint64_t>(room, (uint64_t)((double)text_left * per_text * 1.03) + (256u << 10));
            const size_t got = inner->read(dst + w0 + fill, want);
            if (got == 0) in_eof = true;
            fill += got;
        }
        if (in_eof && off == fill) *eof = true;
        else if (in_eof && !stop && fill - off > 0 && n == 0) return false; // a truncated member or stray bytes at the end
        else if (in_eof && !stop && fill - off > 0) { /* the tail is looked at again by the next call, which fails as above */ }
        // the rest goes back to cbuf
        const size_t rest = fill - off;
        if (rest) {
            if (cbuf.size() < rest) {
                RawBuf big(std::max(rest, (size_t)1 << 20));
                cbuf.swap(big);
            }
            memcpy(cbuf.data(), dst + w0 + off, rest);
            c_lo = 0;
            c_hi = rest;
        }
        *n_out = n;
        *bytes_out = w0 + off;
        *text_out = text;
        return n > 0 || *eof || *budget_hit;
    }

    struct Member { size_t in_off, in_len, out_off; uint32_t isize, crc; };
    // inflate the next batch of members into out[0, out_cap) (>= 64 KiB: room for any one member); false = end of
    // input or error; *produced may be 0 for a batch of empty members
    // FH_TRACE: where a BGZF reader's time goes (compressed reads / member scan / parallel inflate), printed at the end
    double t_read = 0, t_scan = 0, t_inflate = 0;
    uint64_t n_batches = 0, n_members = 0;
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    ~BgzfSource() override {
        if (pending.valid()) pending.wait();
        static const bool trace = cfg("trace") != nullptr;
        if (trace && n_batches)
            fprintf(stderr, "[finch] bgzf: %llu batches, %llu members, %u threads: read %.1f ms, scan %.1f ms, inflate %.1f ms\n",
                    (unsigned long long)n_batches, (unsigned long long)n_members, n_thr, t_read * 1e3, t_scan * 1e3, t_inflate * 1e3);
    }
    bool refill(uint8_t *out, size_t out_cap, size_t *produced) {
        *produced = 0;
        const double tr0 = now_s();
        join_prefetch();
        std::vector<Member> ms;
        size_t out_total = 0, scan = 0; // scan: offset from c_lo of the next member header
        // (a plain gzip file also lands here when several threads are available: look at its first header before
        // buffering 32 MiB of it)
        if (!confirmed && fill_compressed(18)) {
            uint32_t hdr = 0;
            if (member_size(cbuf.data() + c_lo, &hdr)) {
                confirmed = true;
                RawBuf big(2 * (BATCH * 65536 + 65536)); // a worst-case batch, and the next one behind it
                memcpy(big.data(), cbuf.data() + c_lo, c_hi - c_lo);
                c_hi -= c_lo;
                c_lo = 0;
                cbuf.swap(big);
            }
        }
        // top the buffer up once, then take the members that are completely in it
        if (confirmed) fill_compressed(batch_target());
        const double tr1 = now_s();
        t_read += tr1 - tr0;
        while (ms.size() < BATCH) {
            if (c_hi - c_lo - scan < 18) {
                if (scan == 0 && c_hi - c_lo > 0 && in_eof) bad = true; // trailing garbage / truncated header
                break;
            }
            const uint8_t *p = cbuf.data() + c_lo + scan;
            uint32_t hdr = 0;
            const uint32_t tot = member_size(p, &hdr);
            if (tot == 0) {
                if (scan) break; // deliver what precedes it first
                // not BGZF from here on: the sequential reader takes over (it sees the buffered bytes first)
                auto pre = std::make_unique<PrefixedSource>();
                pre->prefix.assign(cbuf.data() + c_lo, cbuf.data() + c_hi);
                pre->inner = std::move(inner);
                c_lo = c_hi = 0;
                tail = make_gzip_reader(std::move(pre), n_thr);
                return true;
            }
            if (tot < hdr + 8u + 2u) { bad = true; return false; }
            if (c_hi - c_lo - scan < tot) {
                if (in_eof) { bad = true; return false; } // truncated member
                if (scan == 0) { // a single member must fit after a refill
                    if (!fill_compressed(tot)) { bad = true; return false; }
                    continue;
                }
                break;
            }
            Member m;
            m.in_off = c_lo + scan + hdr;
            m.in_len = tot - hdr - 8;
            m.crc = p[tot - 8] | ((uint32_t)p[tot - 7] << 8) | ((uint32_t)p[tot - 6] << 16) | ((uint32_t)p[tot - 5] << 24);
            m.isize = p[tot - 4] | ((uint32_t)p[tot - 3] << 8) | ((uint32_t)p[tot - 2] << 16) | ((uint32_t)p[tot - 1] << 24);
            if (m.isize > 65536u) { bad = true; return false; }
            if (out_total + m.isize > out_cap) break; // the destination is full
            m.out_off = out_total;
            out_total += m.isize;
            ms.push_back(m);
            scan += tot;
        }
        if (ms.empty()) return false;
        const double tr2 = now_s();
        t_scan += tr2 - tr1;
        n_batches++;
        n_members += ms.size();
        last_scan = scan;
        start_prefetch(scan);
        std::atomic<bool> ok{true};
        const unsigned nt = (unsigned)std::min<size_t>(n_thr, ms.size());
        static const bool zl = use_zlib_inflate();
        auto job = [&](unsigned t) {
            if (!zl) { // fh_inflate.h: one decoder per thread, members are whole DEFLATE streams of known size
                std::unique_ptr<inf::Decoder> dec(new inf::Decoder());
                for (size_t i = t; i < ms.size() && ok; i += nt) {
                    const Member &m = ms[i];
                    // (8 readable bytes behind the compressed data: the member's own CRC-32 / ISIZE trailer)
                    if (!inf::inflate_exact(*dec, cbuf.data() + m.in_off, m.in_len, out + m.out_off, m.isize) ||
                        inf::crc32_fast(0, out + m.out_off, m.isize) != m.crc)
                        ok = false;
                }
                return;
            }
            z_stream zs{};
            if (inflateInit2(&zs, -15) != Z_OK) { ok = false; return; }
            for (size_t i = t; i < ms.size() && ok; i += nt) {
                const Member &m = ms[i];
                inflateReset(&zs);
                zs.next_in = cbuf.data() + m.in_off;
                zs.avail_in = (uInt)m.in_len;
                zs.next_out = out + m.out_off;
                zs.avail_out = m.isize;
                const int rc = inflate(&zs, Z_FINISH);
                const bool done = (rc == Z_STREAM_END) && zs.avail_out == 0 && zs.avail_in == 0;
                if (!done || (uint32_t)crc32(crc32(0L, Z_NULL, 0), out + m.out_off, m.isize) != m.crc) ok = false;
            }
            inflateEnd(&zs);
        };
        fork_join(nt, job);
        t_inflate += now_s() - tr2;
        if (!ok) { bad = true; return false; }
        c_lo += scan;
        *produced = out_total;
        return true;
    }
    size_t read(uint8_t *dst, size_t cap) override {
        size_t n = 0;
        while (n < cap && !bad) {
            if (tail) {
                const size_t g = tail->read(dst + n, cap - n);
                n += g;
                if (g == 0) break;
                continue;
            }
            if (o_lo == o_hi) {
                size_t got = 0;
                if (cap - n >= ((size_t)1 << 20)) { // room for a batch: inflate straight into the caller's buffer
                    if (!refill(dst + n, cap - n, &got)) break; // end of input or error
                    n += got;
                    continue;
                }
                const size_t own = BATCH * 65536;
                if (obuf.size() < own) obuf.resize(own);
                o_lo = o_hi = 0;
                if (!refill(obuf.data(), own, &got)) break;
                o_hi = got;
                continue; // (tail set, an empty batch, or data to hand out: the loop sorts it out)
            }
            const size_t m = std::min(cap - n, o_hi - o_lo);
            memcpy(dst + n, obuf.data() + o_lo, m);
            o_lo += m;
            n += m;
        }
        return n;
    }
};

// bzip2 / xz: the image ships the runtime libraries (libbz2.so.1, liblzma.so.5) but not their headers, so the few
// entry points needed are declared here and bound with dlopen.  Layouts follow the public bzlib.h / lzma.h ABIs.
struct BzStream {
    char *next_in;
    unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out;
    unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int);
    void (*bzfree)(void *, void *);
    void *opaque;
};

struct Bz2Api {
    int (*init)(BzStream *, int, int) = nullptr;
    int (*decompress)(BzStream *) = nullptr;
    int (*end)(BzStream *) = nullptr;
    bool ok = false;
    Bz2Api() {
        void *h = dlopen("libbz2.so.1", RTLD_NOW);
        if (!h) h = dlopen("libbz2.so.1.0", RTLD_NOW);
        if (!h) return;
        init = (int (*)(BzStream *, int, int))dlsym(h, "BZ2_bzDecompressInit");
        decompress = (int (*)(BzStream *))dlsym(h, "BZ2_bzDecompress");
        end = (int (*)(BzStream *))dlsym(h, "BZ2_bzDecompressEnd");
        ok = init && decompress && end;
    }
};

struct Bz2Source : ByteSource {
    static const Bz2Api &api() {
        static Bz2Api a;
        return a;
    }
    std::unique_ptr<ByteSource> inner;
    BzStream bs{};
    std::vector<uint8_t> inbuf;
    bool eof = false, bad = false, init = false, mid_stream = false;
    explicit Bz2Source(std::unique_ptr<ByteSource> in) : inner(std::move(in)), inbuf(1 << 20) {
        init = api().ok && api().init(&bs, 0, 0) == 0;
        bad = !init;
    }
    ~Bz2Source() override {
        if (init) api().end(&bs);
    }
    bool failed() const override { return bad; }
    size_t read(uint8_t *dst, size_t cap) override {
        if (eof || bad) return 0;
        bs.next_out = (char *)dst;
        bs.avail_out = (unsigned int)std::min<size_t>(cap, 1u << 30);
        while (bs.avail_out > 0) {
            if (bs.avail_in == 0) {
                const size_t got = inner->read(inbuf.data(), inbuf.size());
                if (got == 0) {
                    eof = true;
                    if (mid_stream) bad = true; // the input ends inside a stream: truncated
                    break;
                }
                bs.next_in = (char *)inbuf.data();
                bs.avail_in = (unsigned int)got;
            }
            mid_stream = true;
            const int rc = api().decompress(&bs);
            if (rc == 4 /* BZ_STREAM_END */) {
                mid_stream = false;
                // concatenated streams: start over if more input follows
                api().end(&bs);
                init = api().init(&bs, 0, 0) == 0;
                if (!init) {
                    bad = true;
                    break;
                }
                char *no = bs.next_out;
                (void)no;
                continue;
            }
            if (rc != 0 /* BZ_OK */) {
                bad = true;
                break;
            }
        }
        return (size_t)((uint8_t *)bs.next_out - dst);
    }
};

// lzma_stream of liblzma 5.x (public ABI: see lzma/base.h)
struct LzmaStream {
    const uint8_t *next_in;
    size_t avail_in;
    uint64_t total_in;
    uint8_t *next_out;
    size_t avail_out;
    uint64_t total_out;
    const void *allocator;
    void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2;
    size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};

struct LzmaApi {
    int (*decoder)(LzmaStream *, uint64_t, uint32_t) = nullptr;
    int (*code)(LzmaStream *, int) = nullptr;
    void (*end)(LzmaStream *) = nullptr;
    bool ok = false;
    LzmaApi() {
        void *h = dlopen("liblzma.so.5", RTLD_NOW);
        if (!h) return;
        decoder = (int (*)(LzmaStream *, uint64_t, uint32_t))dlsym(h, "lzma_stream_decoder");
        code = (int (*)(LzmaStream *, int))dlsym(h, "lzma_code");
        end = (void (*)(LzmaStream *))dlsym(h, "lzma_end");
        ok = decoder && code && end;
    }
};

struct XzSource : ByteSource {
    static const LzmaApi &api() {
        static LzmaApi a;
        return a;
    }
    std::unique_ptr<ByteSource> inner;
    LzmaStream ls{};
    std::vector<uint8_t> inbuf;
    bool eof = false, bad = false, init = false, in_eof = false;
    explicit XzSource(std::unique_ptr<ByteSource> in) : inner(std::move(in)), inbuf(1 << 20) {
        init = api().ok && api().decoder(&ls, UINT64_MAX, 0x08 /* LZMA_CONCATENATED */) == 0;
        bad = !init;
    }
    ~XzSource() override {
        if (init) api().end(&ls);
    }
    bool failed() const override { return bad; }
    size_t read(uint8_t *dst, size_t cap) override {
        if (eof || bad) return 0;
        ls.next_out = dst;
        ls.avail_out = cap;
        while (ls.avail_out > 0) {
            if (ls.avail_in == 0 && !in_eof) {
                const size_t got = inner->read(inbuf.data(), inbuf.size());
                if (got == 0) in_eof = true;
                ls.next_in = inbuf.data();
                ls.avail_in = got;
            }
            const int rc = api().code(&ls, in_eof ? 3 /* LZMA_FINISH */ : 0 /* LZMA_RUN */);
            if (rc == 1 /* LZMA_STREAM_END */) {
                eof = true;
                break;
            }
            if (rc != 0 /* LZMA_OK */) {
                bad = true;
                break;
            }
        }
        return (size_t)(ls.next_out - dst);
    }
};

// A gzip file whose first member is decoded by several threads (fh_pargz.h); members after the first, and anything the
// parallel pass cannot make sense of, go through the sequential reader.  FINCH_PARGZ=0 turns it off; FINCH_PARGZ_CHUNK sets
// the compressed bytes per chunk (default 1 MiB; a batch is two chunks per thread, and the next batch is decoded while one is handed out).
struct ParGzSource : ByteSource {
    std::unique_ptr<ByteSource> inner;
    unsigned n_thr;
    size_t chunk_bytes;
    std::vector<uint8_t> cb; // compressed bytes [0, c_n) of the current batch (+ 64 bytes of zeros), decoding starts at bit c_bit
    size_t c_n = 0;
    uint64_t c_bit = 0;
    bool in_eof = false, started = false, member_done = false, bad = false;
    std::vector<uint8_t> window; // the last <= 32 KiB of text delivered: what the next batch's first chunk may refer to
    uint32_t crc = 0;
    uint64_t member_len = 0, delivered = 0;
    std::vector<pargz::Chunk> ready; // text of the current batch, in order
    size_t r_chunk = 0, r_off = 0;
    // (the chunks' buffers come from and go back to pargz::BufPool)
    static void recycle(pargz::Chunk &c) {
        pargz::BufPool &pool = pargz::BufPool::global();
        pool.put(pargz::rebind<uint8_t>(std::move(c.sym)));
        pool.put(std::move(c.bytes));
    }
    std::unique_ptr<ByteSource> tail; // the sequential reader, once it has taken over
    uint64_t n_batches = 0, n_chunks = 0, n_false_starts = 0, sym_total = 0;
    static constexpr size_t CHUNKS_PER_THREAD = 2; // (chunks differ in how long they take: more than one per thread evens that out)
    double t_fill = 0, t_find = 0, t_decode = 0, t_resolve = 0, t_deliver = 0; // FH_TRACE
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    ParGzSource(std::unique_ptr<ByteSource> in, unsigned threads) : inner(std::move(in)), n_thr(std::max(2u, threads)) {
        rewindable = inner && inner->can_rewind();
        const char *e = cfg("pargz_chunk");
        chunk_bytes = e ? (size_t)std::max(4096ll, atoll(e)) : ((size_t)1 << 20);
    }
    ~ParGzSource() override {
        drop_prefetch();
        for (auto &c : ready) recycle(c);
        static const bool trace = cfg("trace") != nullptr;
        if (trace && n_batches)
            fprintf(stderr, "[finch] parallel gzip: %llu batches, %llu chunks (%llu false starts), %.1f %% of the text decoded with markers, %u threads%s\n",
                    (unsigned long long)n_batches, (unsigned long long)n_chunks, (unsigned long long)n_false_starts,
                    100.0 * (double)sym_total / (double)std::max<uint64_t>(1, delivered), n_thr, tail ? "; sequential reader took over" : ""),
            fprintf(stderr, "[finch] parallel gzip: read %.1f ms, block search %.1f ms, decode %.1f ms, windows %.1f ms, hand-over (markers, CRC) %.1f ms\n",
                    t_fill * 1e3, t_find * 1e3, t_decode * 1e3, t_resolve * 1e3, t_deliver * 1e3);
    }
    bool failed() const override { return bad || (tail && tail->failed()); }
    // (answered from what the input said when the reader was made: `inner` belongs to the prefetch thread while a batch
    //  is in the making -- it may be handing it to the reader of the members behind the first at this very moment)
    bool rewindable = false;
    bool can_rewind() const override { return rewindable; }
    bool rewind() override {
        drop_prefetch();
        if (!inner) { // the sequential reader owns the input: it starts over at the first byte of the file
            if (!tail || !tail->rewind()) return false;
            for (auto &c : ready) recycle(c);
            ready.clear();
            r_chunk = r_off = 0;
            member_done = true;
            bad = false;
            delivered = 0;
            return true;
        }
        if (!inner->rewind()) return false;
        c_n = 0;
        c_bit = 0;
        in_eof = started = member_done = bad = false;
        window.clear();
        crc = 0;
        member_len = delivered = 0;
        ready.clear();
        r_chunk = r_off = 0;
        crc_due = false;
        tail.reset();
        return true;
    }
    unsigned threads_hint() const override { return n_thr; }

    void fill_to(size_t want) { // cb holds at least `want` bytes (+ padding) if the input has them
        if (cb.size() < want + 64) cb.resize(want + 64);
        while (!in_eof && c_n < want) {
            const size_t got = inner->read(cb.data() + c_n, want - c_n);
            if (got == 0) in_eof = true;
            c_n += got;
        }
        memset(cb.data() + c_n, 0, 64);
    }
    // RFC 1952 header at cb[0, c_n): its length, 0 if more bytes are needed, -1 if it is not one
    long header_len() const {
        const uint8_t *p = cb.data();
        const size_t n = c_n;
        if (n < 10) return 0;
        if (p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 || (p[3] & 0xE0)) return -1;
        const uint8_t flg = p[3];
        size_t off = 10;
        if (flg & 4) {
            if (n < off + 2) return 0;
            const size_t xlen = p[off] | ((size_t)p[off + 1] << 8);
            off += 2 + xlen;
            if (n < off) return 0;
        }
        for (int bit : {8, 16})
            if (flg & bit) {
                const void *z = off < n ? memchr(p + off, 0, n - off) : nullptr;
                if (!z) return n > ((size_t)1 << 20) ? -1 : 0;
                off = (size_t)((const uint8_t *)z - p) + 1;
            }
        if (flg & 2) off += 2;
        return n < off ? 0 : (long)off;
    }
    // Everything from the start of the file again through the sequential reader, minus what has been delivered already.
    bool fall_back() {
        ready.clear();
        r_chunk = r_off = 0;
        if (!inner->can_rewind() || !inner->rewind()) {
            bad = true;
            return false;
        }
        tail = std::make_unique<FastGzSource>(std::move(inner));
        std::vector<uint8_t> skip((size_t)1 << 20);
        uint64_t left = delivered;
        while (left) {
            const size_t g = tail->read(skip.data(), (size_t)std::min<uint64_t>(left, skip.size()));
            if (g == 0) {
                bad = true;
                return false;
            }
            left -= g;
        }
        return true;
    }
    template <class F>
    void parallel(size_t n, F f) { // f(i) for i in [0, n) on up to n_thr threads
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) break;
                f(i);
            }
        };
        fork_join((unsigned)std::min<size_t>(n_thr, n), [&](unsigned) { work(); }); // (a share that got no thread: the others take its items)
    }

    // The next batch, decoded.  Runs on a thread of its own while the batch before it is handed out: it owns the compressed
    // side of the reader (inner, cb, window, crc, member_len) and touches nothing of the hand-over side.
    enum class Prep { BATCH, END, FALLBACK, BAD };
    struct Prepared {
        Prep st = Prep::END;
        std::vector<pargz::Chunk> chunks;
        std::unique_ptr<ByteSource> tail; // the reader of the members behind the first, if there are any
        bool more = false;                // another batch follows
        bool check_crc = false;           // the member ends with this batch: its CRC-32 is due when the text is out
        uint32_t want_crc = 0;
    };
    Prepared prepare() {
        Prepared out;
        if (member_done) return out;
        // (the first batch is a short one: nothing is sketched before it is through)
        const size_t per_thread = n_batches == 0 ? 1 : CHUNKS_PER_THREAD;
        const size_t batch_bytes = chunk_bytes * n_thr * per_thread;
        const double t0 = now_s();
        fill_to(batch_bytes);
        const double t1 = now_s();
        t_fill += t1 - t0;
        if (!started) {
            long h = header_len();
            if (h == 0 && !in_eof) { // (a header longer than a batch: not worth a special case)
                h = -1;
            }
            if (h <= 0) {
                out.st = Prep::BAD;
                return out;
            }
            c_bit = (uint64_t)h * 8u;
            started = true;
        }
        if ((c_bit >> 3) >= c_n) { // no block in sight
            out.st = Prep::BAD;
            return out;
        }
        n_batches++;
        const size_t n_c = std::max<size_t>(1, std::min<size_t>((size_t)n_thr * per_thread, c_n / chunk_bytes));
        std::vector<pargz::Chunk> ch(n_c);
        ch[0].start_bit = c_bit;
        ch[0].known_window = true;
        for (size_t i = 0; i < n_c; ++i) {
            pargz::BufPool &pool = pargz::BufPool::global();
            if (i > 0) ch[i].sym = pargz::rebind<uint16_t>(pool.get());
            else ch[i].bytes = pool.get(); // (the first chunk decodes to bytes from the start)
        }
        const uint8_t *base = cb.data();
        const size_t n = c_n;
        // 1. where the other chunks begin
        parallel(n_c - 1, [&](size_t k) {
            const size_t i = k + 1;
            std::unique_ptr<inf::Decoder> scratch(new inf::Decoder());
            const uint64_t from = std::max<uint64_t>((uint64_t)i * chunk_bytes * 8u, c_bit + 1);
            ch[i].start_bit = pargz::find_block_start(base, n, from, (uint64_t)(i + 1) * chunk_bytes * 8u, *scratch);
        });
        const double t2 = now_s();
        t_find += t2 - t1;
        // 2. decode
        parallel(n_c, [&](size_t i) {
            if (ch[i].start_bit != UINT64_MAX) pargz::decode_chunk(base, n, ch, i, window.data(), window.size());
        });
        const double t3 = now_s();
        t_decode += t3 - t2;
        // 3. the chain of chunks that really follow each other
        std::vector<size_t> live;
        for (size_t i = 0; i < n_c;) {
            live.push_back(i);
            const pargz::Chunk &c = ch[i];
            if (!c.ok || c.member_end || c.out_of_input) break;
            size_t j = i + 1;
            while (j < n_c && ch[j].start_bit != c.end_bit) j++;
            if (j == n_c) { // (decode_chunk stops only where one of these holds)
                out.st = Prep::BAD;
                return out;
            }
            for (size_t k = i + 1; k < j; ++k) n_false_starts += ch[k].start_bit != UINT64_MAX;
            i = j;
        }
        n_chunks += live.size();
        if (cfg("trace_pargz"))
            for (size_t li : live)
                fprintf(stderr, "[pargz] batch %llu chunk %zu: bits %llu..%llu text %zu (+%zu sym) ok %d end %d ooi %d\n", (unsigned long long)n_batches, li,
                        (unsigned long long)ch[li].start_bit, (unsigned long long)ch[li].end_bit, ch[li].n_bytes, ch[li].n_sym, ch[li].ok,
                        ch[li].member_end, ch[li].out_of_input);
        if (!ch[live.back()].ok) return fallback_result();
        // the window in front of every live chunk, in order (a chunk's own tail may still hold markers); the markers
        // themselves are looked up when the text is handed out
        std::vector<uint8_t> win = window;
        bool ok = true;
        for (size_t li = 0; li < live.size(); ++li) {
            pargz::Chunk &c = ch[live[li]];
            c.win_in = win;
            std::vector<uint8_t> nxt;
            ok = pargz::window_behind(c, c.win_in, nxt) && ok;
            win.swap(nxt);
        }
        if (!ok) return fallback_result();
        t_resolve += now_s() - t3;
        window = win;
        for (size_t li : live) {
            member_len += ch[li].text_len();
            sym_total += ch[li].n_sym;
        }
        // 4. where the batch ended
        const pargz::Chunk &last = ch[live.back()];
        if (last.member_end) {
            size_t t = (size_t)((last.end_bit + 7) >> 3);
            if (c_n < t + 8) fill_to(t + 8 + 65536);
            if (c_n < t + 8) return fallback_result(); // (truncated: the sequential reader reports it)
            const uint8_t *p = cb.data() + t;
            const uint32_t want_crc = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            const uint32_t want_len = p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
            if (want_len != (uint32_t)member_len) { // (text has gone out already: there is no quiet way back)
                out.st = Prep::BAD;
                return out;
            }
            out.check_crc = true; // (known once this batch's text has been handed out)
            out.want_crc = want_crc;
            member_done = true;
            t += 8;
            if (c_n > t || !in_eof) { // more members: the sequential reader's
                auto pre = std::make_unique<PrefixedSource>();
                pre->prefix.assign(cb.data() + t, cb.data() + c_n);
                pre->inner = std::move(inner);
                // (a clean end right here must not count as "empty file": look for a byte first)
                uint8_t b;
                if (pre->read(&b, 1) == 1) {
                    auto pre2 = std::make_unique<PrefixedSource>();
                    pre2->prefix.assign(1, b);
                    pre2->inner = std::move(pre);
                    out.tail = std::make_unique<FastGzSource>(std::move(pre2));
                } else {
                    inner = std::move(pre->inner); // nothing behind the member after all: the input stays here (rewind)
                    in_eof = true;
                }
            }
        } else if (last.out_of_input) {
            // not one whole block in a batch's worth of bytes (or in the rest of the file: truncated): the sequential
            // reader's case
            if (live.size() == 1 && last.end_bit == c_bit) return fallback_result();
            const size_t keep_from = (size_t)(last.end_bit >> 3);
            memmove(cb.data(), cb.data() + keep_from, c_n - keep_from);
            c_n -= keep_from;
            c_bit = last.end_bit & 7u;
        } else {
            out.st = Prep::BAD;
            return out;
        }
        out.chunks.reserve(live.size());
        for (size_t li : live) out.chunks.push_back(std::move(ch[li]));
        for (auto &c : ch) recycle(c); // (the chunks that began at false starts; moved-from ones hold nothing)
        out.st = Prep::BATCH;
        out.more = !member_done;
        return out;
    }

    static Prepared fallback_result() {
        Prepared p;
        p.st = Prep::FALLBACK;
        return p;
    }
    std::future<Prepared> fut;
    bool crc_due = false; // the batch in `ready` ends the member
    uint32_t crc_wanted = 0;
    void drop_prefetch() { // (rewind, destruction)
        if (!fut.valid()) return;
        Prepared p = fut.get();
        for (auto &c : p.chunks) recycle(c);
        if (p.tail) tail = std::move(p.tail); // (it owns the input now: rewind() goes through it)
    }
    // the next batch into `ready` (and the one after it into the making); false: nothing more from this reader
    bool next_batch() {
        for (auto &c : ready) recycle(c);
        ready.clear();
        r_chunk = r_off = 0;
        if (bad) return false;
        Prepared p = fut.valid() ? fut.get() : prepare();
        ready = std::move(p.chunks);
        if (p.tail) tail = std::move(p.tail);
        crc_due = p.check_crc;
        crc_wanted = p.want_crc;
        switch (p.st) {
        case Prep::BAD: bad = true; return false;
        case Prep::FALLBACK: return fall_back();
        case Prep::END: return false;
        case Prep::BATCH: break;
        }
        if (p.more && n_thr > 1) fut = std::async(std::launch::async, [this] { return prepare(); });
        return true;
    }

    size_t read(uint8_t *dst, size_t cap) override {
        size_t n = 0;
        while (n < cap && !bad) {
            if (tail && r_chunk >= ready.size()) { // (the last batch of the first member goes out first)
                const size_t g = tail->read(dst + n, cap - n);
                n += g;
                delivered += g;
                if (g == 0) break;
                continue;
            }
            if (r_chunk < ready.size()) {
                // The segments of the ready text that fit the request: bytes are copied, symbols narrowed (markers looked up
                // in the window in front of their chunk) -- by several threads when there is much of it, each of which also
                // checksums the stretch it has just written.
                const double td0 = now_s();
                struct Seg {
                    const pargz::Chunk *c;
                    bool sym;
                    size_t off, len; // within the chunk's symbols / bytes
                };
                std::vector<Seg> segs;
                size_t m = 0;
                while (r_chunk < ready.size() && n + m < cap) {
                    const pargz::Chunk &c = ready[r_chunk];
                    const size_t total = c.text_len();
                    if (r_off >= total) {
                        r_chunk++;
                        r_off = 0;
                        continue;
                    }
                    const bool in_sym = r_off < c.n_sym;
                    const size_t len = std::min(cap - n - m, (in_sym ? c.n_sym : total) - r_off);
                    segs.push_back(Seg{&c, in_sym, in_sym ? r_off : r_off - c.n_sym, len});
                    m += len;
                    r_off += len;
                }
                uint8_t *const base = dst + n;
                std::atomic<bool> ok{true};
                auto stretch = [&](size_t lo, size_t hi) { // text [lo, hi) of this hand-over; returns its CRC-32
                    size_t pos = 0;
                    for (const Seg &sg : segs) {
                        const size_t a = std::max(lo, pos), b = std::min(hi, pos + sg.len);
                        if (a < b) {
                            const size_t o = sg.off + (a - pos);
                            if (sg.sym) {
                                const std::vector<uint8_t> &w = sg.c->win_in;
                                if (!pargz::resolve_span(sg.c->sym.data() + pargz::WINDOW + o, b - a, w.data() + w.size(), w.size(), base + a)) ok = false;
                            } else {
                                memcpy(base + a, sg.c->bytes.data() + o, b - a);
                            }
                        }
                        pos += sg.len;
                    }
                    return inf::crc32_fast(0, base + lo, hi - lo);
                };
                if (m >= ((size_t)8 << 20) && n_thr > 1) {
                    const size_t per = (m + n_thr - 1) / n_thr;
                    std::vector<uint32_t> crcs(n_thr, 0);
                    parallel(n_thr, [&](size_t t) {
                        const size_t lo = std::min(m, t * per), hi = std::min(m, lo + per);
                        if (lo < hi) crcs[t] = stretch(lo, hi);
                    });
                    for (size_t t = 0; t < n_thr; ++t) {
                        const size_t lo = std::min(m, t * per), hi = std::min(m, lo + per);
                        if (lo < hi) crc = (uint32_t)crc32_combine(crc, crcs[t], (z_off_t)(hi - lo));
                    }
                } else if (m) {
                    crc = (uint32_t)crc32_combine(crc, stretch(0, m), (z_off_t)m);
                }
                if (!ok) bad = true; // a marker that points before the start of the stream
                n += m;
                delivered += m;
                t_deliver += now_s() - td0;
                if (r_chunk < ready.size() && r_off >= ready[r_chunk].text_len()) {
                    r_chunk++;
                    r_off = 0;
                }
                if (r_chunk >= ready.size() && crc_due) { // the member's text is out: its checksum
                    crc_due = false;
                    if (crc != crc_wanted) bad = true;
                }
                continue;
            }
            if (!next_batch() && !tail) break;
        }
        return n;
    }
};

static std::unique_ptr<ByteSource> make_gzip_reader(std::unique_ptr<ByteSource> in, unsigned threads) {
    if (use_zlib_inflate()) return std::make_unique<GzSource>(std::move(in));
    const char *e = cfg("pargz");
    if (threads > 1 && !(e && e[0] == '0')) return std::make_unique<ParGzSource>(std::move(in), threads);
    return std::make_unique<FastGzSource>(std::move(in));
}

// needletail parse_fastx_reader: sniff two magic bytes (lib.rs:60)
static int open_source(std::unique_ptr<ByteSource> raw, std::unique_ptr<ByteSource> &out, bool *is_gz = nullptr,
                       int *first_byte = nullptr) {
    auto pre = std::make_unique<PrefixedSource>();
    pre->prefix.resize(2);
    size_t got = 0;
    while (got < 2) {
        const size_t g = raw->read(pre->prefix.data() + got, 2 - got);
        if (g == 0) break;
        got += g;
    }
    pre->prefix.resize(got);
    const bool gz = got == 2 && pre->prefix[0] == 0x1F && pre->prefix[1] == 0x8B;
    const bool bz = got == 2 && pre->prefix[0] == 0x42 && pre->prefix[1] == 0x5A;
    const bool xz = got == 2 && pre->prefix[0] == 0xFD && pre->prefix[1] == 0x37;
    // threads the decompressor may use: what the source was given (finch_sketch_files shares FINCH_READ_THREADS
    // among its workers); FINCH_BGZF_THREADS overrides it (tests, in-memory inputs)
    unsigned dec_threads = raw->threads_hint();
    if (const char *e = cfg("bgzf_threads")) dec_threads = (unsigned)std::max(1, atoi(e));
    pre->inner = std::move(raw);
    if (is_gz) *is_gz = gz;
    if (first_byte) *first_byte = got ? pre->prefix[0] : -1;
    if (bz && !Bz2Source::api().ok) return hfail(FH_ERR_UNSUPPORTED, "bzip2-compressed input: libbz2.so.1 not found");
    if (xz && !XzSource::api().ok) return hfail(FH_ERR_UNSUPPORTED, "xz-compressed input: liblzma.so.5 not found");
    if (is_gz) *is_gz = gz || bz || xz; // "compressed": not eligible for device-side text parsing
    // (FINCH_GZ_FRONT=1, A/B: the reader that knows BGZF and hands compressed bytes to the device stands in front of gzip
    //  input even when the call has no read thread to spare for it -- the workers of a many-file call)
    static const bool front_always = [] {
        const char *e = cfg("gz_front");
        return e && e[0] == '1';
    }();
    if (gz && (dec_threads > 1 || (front_always && !use_zlib_inflate()))) out = std::make_unique<BgzfSource>(std::move(pre), dec_threads); // falls back member by member
    else if (gz && use_zlib_inflate()) out = std::make_unique<GzSource>(std::move(pre));
    else if (gz) out = std::make_unique<FastGzSource>(std::move(pre));
    else if (bz) out = std::make_unique<Bz2Source>(std::move(pre));
    else if (xz) out = std::make_unique<XzSource>(std::move(pre));
    else out = std::move(pre);
    return FH_OK;
}

// ---------------------------------------------------------------------------------------------
// record sink: either the device sketcher or a counter (finch_fastx_scan)
// ---------------------------------------------------------------------------------------------
struct RecordSink {
    virtual ~RecordSink() {}
    // a piece of the current record's sequence() bytes; `raw_len` is what total_bases counts for it
    virtual int piece(const uint8_t *p, size_t n) = 0;
    virtual int end_record() = 0;
};

struct CountSink : RecordSink {
    uint64_t records = 0;
    int piece(const uint8_t *, size_t) override { return FH_OK; }
    int end_record() override {
        records++;
        return FH_OK;
    }
};

// copy n bytes dropping ' ', '\t', '\r', '\n' (what normalize(false) removes, mash.rs:73); 8 bytes at a time when clean
// (normalize(false) drops blanks, mash.rs:73: fh_strip.h; the destination needs 32 bytes of slack behind what is kept)
static inline size_t strip_copy(uint8_t *dst, const uint8_t *src, size_t n) { return fh_strip::strip_scalar(dst, src, n); }

// SketchScheme::process (mash.rs:67-80) over the C ABI: record bytes + one breaker byte, written (whitespace
// already dropped) straight into the sketcher's pinned staging buffer and committed in large blocks; a record
// that does not fit continues in the next block with FH_PUSH_CONTINUE.
struct DeviceSink : RecordSink {
    fh_sketcher *h;
    uint8_t *buf = nullptr;
    uint64_t cap = 0, fill = 0;
    uint64_t limit = 0;      // test knob: commit after this many bytes
    bool continuing = false; // the block starts inside a record that an earlier commit cut
    bool in_record = false;
    explicit DeviceSink(fh_sketcher *h_) : h(h_) {
        const char *e = cfg("block_bytes"); // test knob: force records to span blocks
        limit = e ? strtoull(e, nullptr, 10) : 0;
    }
    int acquire() {
        if (buf) return FH_OK;
        if (int rc = fh_text_buffer(h, &buf, &cap)) return hfail(rc, "%s", fh_last_error());
        if (limit && limit < cap) cap = limit;
        if (cap < 64) return hfail(FH_ERR_INVALID, "staging buffer too small");
        fill = 0;
        return FH_OK;
    }
    int flush() {
        if (!buf || fill == 0) return FH_OK;
        const int rc = fh_push_staged(h, fill, continuing ? FH_PUSH_CONTINUE : 0u);
        if (rc != FH_OK) return hfail(rc, "%s", fh_last_error());
        buf = nullptr;
        fill = 0;
        continuing = in_record;
        return FH_OK;
    }
    int piece(const uint8_t *p, size_t n) override {
        in_record = true;
        while (n) {
            if (int rc = acquire()) return rc;
            const size_t room = (size_t)(cap - fill);
            const size_t m = std::min(room, n);
            fill += strip_copy(buf + fill, p, m);
            p += m;
            n -= m;
            if (fill >= cap)
                if (int rc = flush()) return rc;
        }
        return FH_OK;
    }
    int end_record() override {
        if (int rc = acquire()) return rc;
        buf[fill++] = 0;
        in_record = false;
        if (fill >= cap) return flush();
        return FH_OK;
    }
};

// ---------------------------------------------------------------------------------------------
// FASTX reader restating needletail 0.5.0 (lib.rs:60-68): first byte '>' => FASTA (multi-line),
// '@' => FASTQ (4-line records, equal-length seq/qual, CR trimmed).  sequence() of a FASTA record is the
// raw slice from after the header line to the end of the last sequence line (internal newlines included,
// the final newline and a CR before it excluded) -- that length is what total_bases counts (mash.rs:72).
// ---------------------------------------------------------------------------------------------
struct FastxStats {
    uint64_t n_records = 0, total_bases = 0;
    int format = 0; // 1 FASTA, 2 FASTQ
};

static int parse_fastx(ByteSource &src, RecordSink &sink, FastxStats &st) {
    std::vector<uint8_t> buf(8u << 20);
    size_t lo = 0, hi = 0; // valid bytes [lo, hi)
    bool eof = false;
    auto refill = [&]() -> bool { // returns false if nothing more could be read
        if (eof) return false;
        if (lo > 0) {
            memmove(buf.data(), buf.data() + lo, hi - lo);
            hi -= lo;
            lo = 0;
        }
        if (hi == buf.size()) buf.resize(buf.size() * 2);
        const size_t got = src.read(buf.data() + hi, buf.size() - hi);
        if (got == 0) {
            eof = true;
            return false;
        }
        hi += got;
        return true;
    };
    refill();
    if (src.failed()) return hfail(FH_ERR_INVALID, "corrupt compressed stream");
    if (hi == 0) return hfail(FH_ERR_INVALID, "empty input: not a FASTA/FASTQ file");
    if (buf[0] == '>') {
        st.format = 1;
        // line-oriented state machine over the buffered stream
        enum { HEADER, SEQ } state = HEADER;
        bool at_line_start = true, have_record = false;
        uint64_t raw_len = 0;        // bytes of the current record's sequence region so far
        uint8_t prev = 0, last = 0;  // its last two bytes
        auto close_record = [&]() -> int {
            uint64_t trim = 0;
            if (raw_len >= 1 && last == '\n') {
                trim = 1;
                if (raw_len >= 2 && prev == '\r') trim = 2;
            } else if (raw_len >= 1 && last == '\r') {
                trim = 1;
            }
            st.total_bases += raw_len - trim;
            st.n_records++;
            return sink.end_record();
        };
        for (;;) {
            if (lo == hi && !refill()) break;
            if (at_line_start && buf[lo] == '>') {
                if (have_record)
                    if (int rc = close_record()) return rc;
                have_record = true;
                raw_len = 0;
                prev = last = 0;
                state = HEADER;
            }
            const uint8_t *nl = (const uint8_t *)memchr(buf.data() + lo, '\n', hi - lo);
            const size_t end = nl ? (size_t)(nl - buf.data()) : hi; // [lo,end): (part of) a line, newline excluded
            if (state == SEQ) {
                const size_t n = end - lo;
                if (n) {
                    if (int rc = sink.piece(buf.data() + lo, n)) return rc; // a CR goes along; the device side drops it
                    raw_len += n;
                    prev = n >= 2 ? buf[end - 2] : last;
                    last = buf[end - 1];
                }
                if (nl) {
                    raw_len += 1;
                    prev = last;
                    last = '\n';
                }
            }
            if (nl) {
                lo = end + 1;
                at_line_start = true;
                if (state == HEADER) state = SEQ;
            } else {
                lo = hi;
                at_line_start = false;
            }
        }
        if (src.failed()) return hfail(FH_ERR_INVALID, "corrupt compressed stream");
        if (have_record)
            if (int rc = close_record()) return rc;
        return FH_OK;
    }
    if (buf[0] != '@') return hfail(FH_ERR_INVALID, "not a FASTA/FASTQ file (first byte 0x%02x)", buf[0]);
    st.format = 2;
    for (;;) {
        // skip blank lines between records / at EOF
        for (;;) {
            if (lo == hi && !refill()) break;
            if (lo < hi && (buf[lo] == '\n' || buf[lo] == '\r')) ++lo;
            else break;
        }
        if (lo == hi) break;
        // the four line ends of the record (the last one may be EOF); refill() keeps [lo,hi) and sets lo = 0
        size_t ends[4];
        bool at_eof = false;
        for (;;) {
            int found = 0;
            size_t scan = lo;
            while (found < 4 && scan < hi) {
                const uint8_t *nl = (const uint8_t *)memchr(buf.data() + scan, '\n', hi - scan);
                if (!nl) break;
                ends[found++] = (size_t)(nl - buf.data());
                scan = ends[found - 1] + 1;
            }
            if (found == 4) break;
            if (at_eof) {
                if (found == 3) {
                    ends[3] = hi; // last line without a newline
                    break;
                }
                return hfail(FH_ERR_INVALID, "truncated FASTQ record");
            }
            if (!refill()) at_eof = true; // refill may have moved [lo,hi): rescan either way
        }
        const size_t h0 = lo, s0 = ends[0] + 1, p0 = ends[1] + 1, q0 = ends[2] + 1;
        if (buf[h0] != '@') return hfail(FH_ERR_INVALID, "invalid FASTQ record: expected '@'");
        if (p0 >= hi || buf[p0] != '+') return hfail(FH_ERR_INVALID, "invalid FASTQ record: expected '+'");
        size_t s1 = ends[1], q1 = ends[3];
        if (s1 > s0 && buf[s1 - 1] == '\r') --s1;
        if (q1 > q0 && buf[q1 - 1] == '\r') --q1;
        if (s1 - s0 != q1 - q0) return hfail(FH_ERR_INVALID, "invalid FASTQ record: sequence and quality lengths differ");
        if (s1 > s0)
            if (int rc = sink.piece(buf.data() + s0, s1 - s0)) return rc;
        st.total_bases += s1 - s0;
        st.n_records++;
        if (int rc = sink.end_record()) return rc;
        lo = std::min(hi, ends[3] + 1);
    }
    if (src.failed()) return hfail(FH_ERR_INVALID, "corrupt compressed stream");
    return FH_OK;
}

// ---------------------------------------------------------------------------------------------
// sketch_stream (lib.rs:51-94)
// ---------------------------------------------------------------------------------------------
static uint64_t env_max_launch_value() {
    const char *e = cfg("max_launch");
    return e ? strtoull(e, nullptr, 10) : 0;
}

static fh_params to_fh(const finch_sketch_params &sp, uint64_t max_launch, uint64_t stage_bytes = 0) {
    fh_params p{};
    p.kind = sp.kind;
    p.k = sp.kmer_length;
    p.size = sp.kmers_to_sketch;
    p.seed = sp.hash_seed;
    p.scale = sp.scale;
    p.max_launch = max_launch;
    p.hash_mask = 0;
    p.stage_bytes = stage_bytes;
    return p;
}

// Device-side record splitting for plain 4-line FASTQ (fh_push_fastq_text): the host reads raw file bytes straight
// into the sketcher's pinned staging buffer and only looks for a record boundary near the end of each chunk.
// A line is a header iff it starts with '@' and the line two below starts with '+' (a quality line may start
// with '@', but then the line two below is a sequence line).
static int pump_text_to_device(ByteSource &src, fh_sketcher *h, bool fastq, uint32_t k, struct FastxStats &st);
static int bgzf_fastq_to_device(struct BgzfSource &bz, fh_sketcher *h);
static std::atomic<uint64_t> g_bgzf_on_device{0}, g_bgzf_reread{0};
static int gzip_fastq_to_device(struct BgzfSource &bz, size_t hdr_len, fh_sketcher *h);
static std::atomic<uint64_t> g_gzip_on_device{0}, g_gzip_reread{0};
// finch_debug_kernel_times: the sketch kernel's own time (HIP events on each worker's stream, fh_kernel_time) over the files
// this process sketches while it is on -- what a batch's roofline line is made of (bench.py --workload c5)
static std::atomic<int> g_ktimes_on{0};
// files of finch_sketch_files calls that the batch path (fh_batch_*) took / handed to a sketcher of their own
static std::atomic<uint64_t> g_batch_taken{0}, g_batch_not_taken{0};
// FINCH_FILE_BATCH=0: every file of a batch through a sketcher of its own (A/B, tests)
static bool file_batch_enabled() {
    const char *e = cfg("file_batch");
    return !(e && e[0] == '0');
}
static std::atomic<uint64_t> g_ktimes_us{0}, g_ktimes_launches{0}, g_ktimes_positions{0};
static int fastq_text_to_device(ByteSource &src, fh_sketcher *h, uint32_t k, struct FastxStats *st_out, bool *host_counted);

// Device-side FASTA (fh_push_fasta_text): the host reads raw file bytes into the pinned staging buffer, cuts chunks
// after a newline and does the bookkeeping that needs no per-base work: the record count and total_bases =
// sum of the raw lengths of the records' sequence regions (everything between a header line and the next
// line-start '>', internal newlines included, one trailing line end trimmed -- what parse_fastx counts, mash.rs:72).
// '>' is rare, so one memchr for it per chunk finds the header lines.
struct FastaCounter {
    bool at_line_start = true, in_header = false, have_record = false;
    uint64_t raw_len = 0, total_bases = 0, n_records = 0;
    uint8_t prev = 0, last = 0;
    void close_record() {
        uint64_t trim = 0;
        if (raw_len >= 1 && last == '\n') {
            trim = 1;
            if (raw_len >= 2 && prev == '\r') trim = 2;
        } else if (raw_len >= 1 && last == '\r') {
            trim = 1;
        }
        total_bases += raw_len - trim;
        n_records++;
    }
    void seq_bytes(const uint8_t *p, size_t n) { // n bytes of a sequence region
        if (!n) return;
        raw_len += n;
        prev = n >= 2 ? p[n - 2] : last;
        last = p[n - 1];
        at_line_start = last == '\n';
    }
    // what fh_push_fasta_text has to be told about the point the next chunk starts at
    uint32_t start_state() const { return in_header ? 2u : (at_line_start ? 0u : 1u); }
    // every '>' of p[0, n), in order, found by `threads` threads (a 64 MiB chunk is 5 ms of memchr on one thread -- more than
    // the chunk's copy to the device takes -- and holds a handful of them)
    static void find_gt(const uint8_t *p, size_t n, unsigned threads, std::vector<size_t> &out, size_t min_piece = (size_t)4 << 20) {
        out.clear();
        const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, threads), std::max<size_t>(1, n / min_piece));
        std::vector<std::vector<size_t>> part(nt);
        const size_t per = (n + nt - 1) / nt;
        auto job = [&](unsigned t) {
            const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
            for (size_t i = lo; i < hi;) {
                const uint8_t *g = (const uint8_t *)memchr(p + i, '>', hi - i);
                if (!g) break;
                part[t].push_back((size_t)(g - p));
                i = (size_t)(g - p) + 1;
            }
        };
        fork_join(nt, job);
        for (auto &v : part) out.insert(out.end(), v.begin(), v.end());
    }
    // gts: the positions of every '>' in p[0, n) if the caller has them (find_gt), else they are searched for as the walk goes
    void feed(const uint8_t *p, size_t n, const std::vector<size_t> *gts = nullptr) {
        size_t i = 0, gc = 0;
        while (i < n) {
            if (in_header) {
                const uint8_t *nl = (const uint8_t *)memchr(p + i, '\n', n - i);
                if (!nl) return; // the header line continues in the next chunk
                i = (size_t)(nl - p) + 1;
                in_header = false;
                at_line_start = true;
                continue;
            }
            const uint8_t *g;
            if (gts) {
                while (gc < gts->size() && (*gts)[gc] < i) ++gc;
                g = gc < gts->size() ? p + (*gts)[gc] : nullptr;
            } else {
                g = (const uint8_t *)memchr(p + i, '>', n - i);
            }
            const size_t gi = g ? (size_t)(g - p) : n;
            const bool line_start = g && (gi == i ? at_line_start : p[gi - 1] == '\n');
            if (g && !line_start) { // a '>' inside a line is sequence text
                seq_bytes(p + i, gi + 1 - i);
                i = gi + 1;
                continue;
            }
            seq_bytes(p + i, gi - i);
            i = gi;
            if (g) {
                if (have_record) close_record();
                have_record = true;
                raw_len = 0;
                prev = last = 0;
                in_header = true;
            }
        }
    }
    void finish() {
        if (have_record) close_record();
        have_record = false;
    }
};

static int fasta_text_to_device(ByteSource &src, fh_sketcher *h, FastxStats &st, uint32_t k) { return pump_text_to_device(src, h, false, k, st); }
// FASTQ text that lies in host memory, with read threads to spare: the call's read threads drop headers, '+' lines and
// quality strings on the HOST (fh_fqstrip.h) and only the packed sequence stream -- 1.007 bytes per base instead of the
// text's 2.1 -- crosses the PCIe link (fh_push_staged; each chunk's copy starts the moment it is packed, fh_text_prefetch).
// The device-side splitter stays what a call without threads to spare goes through (the workers of a many-file call have
// one each), and what compressed and file input goes through (their reads already keep the threads busy).
// -> FH_OK (st: records, total_bases), FH_ERR_STATE = does not apply (nothing consumed), FH_ERR_INVALID = not plain 4-line
// FASTQ (the caller rewinds and lets the parser that is the judge of that read it), or an error.
static std::atomic<uint64_t> g_fastq_host_strip{0};

// The threads of a memory-bound pass over a buffer run next to the buffer: the CPUs of the NUMA node the buffer's pages are on,
// as far as the thread is allowed on them (a strip whose threads sit on the other socket runs at half the rate: 18-21 ms
// against 11 for 1.26 GB of FASTQ text on a two-socket box).  Restored when the object goes.  Nothing happens if the node
// cannot be told (no NUMA, no sysfs), or with the option no_numa_pin.
struct NearMemory {
    cpu_set_t want;
    bool have = false;
    explicit NearMemory(const void *addr) {
        if (cfg("no_numa_pin")) return;
        int node = -1;
#ifdef SYS_get_mempolicy
        if (syscall(SYS_get_mempolicy, &node, nullptr, 0UL, addr, 3UL /* MPOL_F_NODE | MPOL_F_ADDR */) != 0) node = -1;
#endif
        if (node < 0) return;
        char path[96], list[4096] = {0};
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        if (FILE *f = fopen(path, "r")) {
            if (!fgets(list, (int)sizeof list, f)) list[0] = 0;
            fclose(f);
        }
        cpu_set_t allowed;
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
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
            if (*e != ',') break;
            p = e + 1;
        }
        have = n >= 8; // (fewer CPUs than the strip has threads: better spread out)
    }
    struct Seat { // one thread's stay
        cpu_set_t old;
        bool moved = false;
        explicit Seat(const NearMemory &m) {
            if (m.have && sched_getaffinity(0, sizeof old, &old) == 0 && sched_setaffinity(0, sizeof m.want, &m.want) == 0) moved = true;
        }
        ~Seat() {
            if (moved) (void)sched_setaffinity(0, sizeof old, &old);
        }
    };
};
static int fastq_host_strip_to_device(ByteSource &src, fh_sketcher *h, FastxStats &st) {
    const char *opt = cfg("fastq_host_strip"); // 0 = never, 1 = whatever the thread count (tests), default: from 8 read threads on
    if (opt && opt[0] == '0') return FH_ERR_STATE;
    const unsigned hint = src.threads_hint();
    const bool forced = opt && opt[0] == '1';
    if (!forced && hint < 8) return FH_ERR_STATE;
    const unsigned T = std::max(2u, std::min(32u, hint));
    const uint8_t *text = nullptr;
    size_t n = 0;
    if (!src.whole_view(&text, &n) || n == 0 || text[0] != '@') return FH_ERR_STATE;
    uint8_t *stage[2] = {nullptr, nullptr};
    uint64_t cap = 0;
    int next = 0;
    if (int rc = fh_text_buffers(h, stage, &cap, &next)) return hfail(rc, "%s", fh_last_error());
    if (cap < (1u << 16)) return FH_ERR_STATE;
    // a chunk of text whose packed stream fits a staging buffer whatever it holds (a record's sequence is less than half of it)
    static const uint64_t chunk_opt = cfg("fastq_strip_chunk") ? strtoull(cfg("fastq_strip_chunk"), nullptr, 10) : 0; // (tests: many chunks)
    const size_t CHUNK = (size_t)std::min<uint64_t>(chunk_opt ? std::max<uint64_t>(chunk_opt, 4096) : (128ull << 20), 2 * (cap - 4096));
    struct Job {
        int slot;
        uint64_t m;
    };
    std::mutex mu;
    std::condition_variable cv;
    bool is_free[2] = {true, true}, producer_done = false;
    std::vector<Job> ready;
    std::atomic<bool> abort{false};
    int prc = FH_OK;
    std::string pmsg;
    uint64_t n_rec_total = 0, bases_total = 0;
    // the team: T - 1 helpers parked between chunks, the producer is member 0
    std::vector<fqstrip::Piece> pieces(T);
    fqstrip::Barrier bar(T);
    struct Work {
        const uint8_t *text = nullptr;
        size_t n = 0;
        uint8_t *out = nullptr;
        unsigned gen = 0;
        bool quit = false;
    } work;
    std::mutex wmu;
    std::condition_variable wcv;
    bool ok = false;
    uint64_t m_out = 0, rec_out = 0, bases_out = 0;
    const NearMemory near_text(text + n / 2);
    auto helper_main = [&](unsigned t) {
        NearMemory::Seat seat(near_text);
        unsigned seen = 0;
        for (;;) {
            Work w;
            {
                std::unique_lock<std::mutex> lk(wmu);
                wcv.wait(lk, [&] { return work.gen != seen || work.quit; });
                if (work.quit) return;
                seen = work.gen;
                w = work;
            }
            fqstrip::strip_chunk(t, T, w.text, w.n, w.out, pieces, bar, &ok, &m_out, &rec_out, &bases_out);
        }
    };
    auto stop_helpers = [&] {
        {
            std::lock_guard<std::mutex> g(wmu);
            work.quit = true;
        }
        wcv.notify_all();
    };
    static const bool trace = cfg("trace") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_s();
    // (trace: how long the cgroup's CPU quota held this process's threads back meanwhile -- a team of as many threads as the
    // quota has CPUs, next to the pushing thread and the runtime's, is throttled as soon as anything else runs)
    auto throttled_us = [] {
        unsigned long long v = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.stat", "r")) {
            char key[64];
            unsigned long long x;
            while (fscanf(f, "%63s %llu", key, &x) == 2)
                if (strcmp(key, "throttled_usec") == 0) v = x;
            fclose(f);
        }
        return v;
    };
    const unsigned long long thr0 = trace ? throttled_us() : 0;
    double t_wait_slot = 0, t_strip = 0, t_push = 0, t_wait_job = 0;
    unsigned n_chunks = 0;
    int rc = FH_OK;
    std::string msg;
    auto pipeline = [&] { // the reader (a thread of its own: member 0 of the team) and, on this thread, the pushes
    std::thread producer([&] {
        NearMemory::Seat seat(near_text);
        int slot = next;
        size_t off = 0;
        while (off < n && !abort) {
            {
                const double w0 = trace ? now_s() : 0;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return is_free[slot] || abort.load(); });
                if (abort) break;
                is_free[slot] = false;
                if (trace) t_wait_slot += now_s() - w0;
            }
            const double s0 = trace ? now_s() : 0;
            size_t len = std::min(CHUNK, n - off);
            if (off + len < n) { // cut behind the last whole record: the last header line whose line two below is a '+' line
                const uint8_t *buf = text + off;
                size_t ls[16];
                int nl = 0;
                size_t pos = len;
                while (nl < 16) {
                    const uint8_t *q = pos > 0 ? (const uint8_t *)memrchr(buf, '\n', pos) : nullptr;
                    const size_t start = q ? (size_t)(q - buf) + 1 : 0;
                    if (start < len) ls[nl++] = start;
                    if (!q) break;
                    pos = (size_t)(q - buf);
                }
                size_t cut = 0;
                for (int i = 2; i < nl && !cut; ++i)
                    if (buf[ls[i]] == '@' && buf[ls[i - 2]] == '+') cut = ls[i];
                if (!cut) {
                    prc = FH_ERR_INVALID;
                    pmsg = "no FASTQ record boundary found in a chunk";
                    break;
                }
                len = cut;
            }
            {
                std::lock_guard<std::mutex> g(wmu);
                work.text = text + off;
                work.n = len;
                work.out = stage[slot];
                ++work.gen;
            }
            wcv.notify_all();
            fqstrip::strip_chunk(0, T, text + off, len, stage[slot], pieces, bar, &ok, &m_out, &rec_out, &bases_out);
            if (!ok) {
                prc = FH_ERR_INVALID;
                pmsg = "not plain 4-line FASTQ";
                break;
            }
            n_rec_total += rec_out;
            bases_total += bases_out;
            if (trace) t_strip += now_s() - s0, n_chunks++;
            (void)fh_text_prefetch(h, slot, m_out); // the chunk's copy starts now, behind the previous chunk's
            {
                std::lock_guard<std::mutex> g(mu);
                ready.push_back(Job{slot, m_out});
            }
            cv.notify_all();
            off += len;
            slot ^= 1;
        }
        std::lock_guard<std::mutex> g(mu);
        producer_done = true;
        cv.notify_all();
    });
    int prev_slot = -1;
    for (;;) {
        Job job;
        const double j0 = trace ? now_s() : 0;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || producer_done; });
            if (ready.empty()) break;
            job = ready.front();
            ready.erase(ready.begin());
        }
        const double j1 = trace ? now_s() : 0;
        t_wait_job += j1 - j0;
        if (rc == FH_OK) {
            rc = fh_push_staged(h, job.m, 0u);
            if (trace) t_push += now_s() - j1;
            if (rc != FH_OK) {
                msg = fh_last_error();
                abort = true;
            }
        }
        // the slot pushed BEFORE this one is free again: this push waited for its sketch launch (which had waited for its copy)
        std::lock_guard<std::mutex> g(mu);
        if (prev_slot >= 0) is_free[prev_slot] = true;
        if (rc != FH_OK) is_free[0] = is_free[1] = true;
        prev_slot = job.slot;
        cv.notify_all();
    }
    producer.join();
    stop_helpers();
    };
    // the helpers are the process's parked team threads (TeamPool) where those are free -- starting fifteen threads per call is
    // a millisecond of a 15 ms call, and threads that have just been created run slower than threads that have run -- else
    // threads of this call's own
    if (!TeamPool::instance().run(T - 1, [&](unsigned i) { helper_main(i + 1); }, pipeline)) {
        std::vector<std::thread> helpers;
        try {
            for (unsigned t = 1; t < T; ++t) helpers.emplace_back(helper_main, t);
        } catch (...) { // not enough threads: this path is not for now (the helpers that exist wait for a chunk that never comes)
            stop_helpers();
            for (auto &x : helpers) x.join();
            return FH_ERR_STATE;
        }
        pipeline();
        for (auto &x : helpers) x.join();
    }
    if (trace)
        fprintf(stderr, "[finch] fastq host strip: %u chunks of <= %.0f MiB of text on %u threads in %.1f ms: strip %.1f ms, producer waited %.1f ms for a buffer, pushes took %.1f ms and waited %.1f ms for chunks; the cgroup throttled its threads for %.1f ms meanwhile\n",
                n_chunks, CHUNK / 1048576.0, T, (now_s() - t_begin) * 1e3, t_strip * 1e3, t_wait_slot * 1e3, t_push * 1e3, t_wait_job * 1e3,
                (throttled_us() - thr0) / 1e3);
    if (rc != FH_OK) return hfail(rc, "%s", msg.c_str());
    if (prc != FH_OK) return hfail(prc, "%s", pmsg.c_str());
    st.total_bases = bases_total;
    st.n_records = n_rec_total;
    g_fastq_host_strip++;
    return FH_OK;
}

// (host_counted: total_bases / n_records of st are the host's count; otherwise the device's, fh_text_bases)
static int fastq_text_to_device(ByteSource &src, fh_sketcher *h, uint32_t k, FastxStats *st_out, bool *host_counted) {
    FastxStats st;
    if (host_counted) *host_counted = false;
    const int rc = fastq_host_strip_to_device(src, h, st);
    if (rc != FH_ERR_STATE) {
        if (rc == FH_OK && st_out && host_counted) {
            st_out->total_bases = st.total_bases;
            st_out->n_records = st.n_records;
            *host_counted = true;
        }
        return rc;
    }
    return pump_text_to_device(src, h, true, k, st);
}

// The sketchers of one worker.  With filtering off, a Mash sketch of `kmers_to_sketch` hashes that is then truncated to
// `final_size` (mod.rs:115-128) is hash for hash, count for count the sketch of `final_size` hashes: the bottom
// final_size of the bottom kmers_to_sketch are the bottom final_size, and the counts are exact occurrence counts either
// way.  The CLI's defaults oversketch 200-fold, and FASTA input defaults to filtering off (lib.rs:70-76), so
// `finch sketch *.fa` only ever needs the small sketcher (a batch of 5 Mb genomes: 1160 -> 4900 files/s); FASTQ input
// (filtering on by default) and explicit filters get the full one.  Both are created on first use.
struct HandleSet {
    fh_params full{};
    uint64_t final_size = 0;
    int device = 0;
    fh_sketcher *h_full = nullptr, *h_small = nullptr;
    fh_sketcher *get(bool small) {
        fh_sketcher *&h = small ? h_small : h_full;
        if (!h) {
            fh_params p = full;
            if (small) p.size = final_size;
            h = fh_new(&p, device);
        }
        if (h && g_ktimes_on.load(std::memory_order_relaxed)) fh_set_profiling(h, 1);
        return h;
    }
    ~HandleSet() {
        if (h_full) fh_free(h_full);
        if (h_small) fh_free(h_small);
    }
};

// to_vec -> filter_counts -> process_post_filter -> Sketch (lib.rs:70-93) from a sketcher that holds the whole input
// (one handle, or the merge of the partial sketches of a sharded input)
// f(lo, hi) over [0, n) on a few threads (the passes over a 2 M-hash oversketch)
template <class F>
static void host_parallel(size_t n, F f) {
    const unsigned t_max = (unsigned)std::min<size_t>(8, std::max<size_t>(1, n >> 16));
    if (t_max <= 1) {
        f(0, 0, n);
        return;
    }
    const size_t per = (n + t_max - 1) / t_max;
    fork_join(t_max, [=](unsigned t) { f(t, std::min(n, (size_t)t * per), std::min(n, ((size_t)t + 1) * per)); });
}

// the same team of threads for several passes in a row: starting and joining a std::thread costs 50-100 us, which for three
// passes over a 2 M-hash oversketch was more than the passes themselves.  f(t, n_threads, barrier) runs on every thread;
// barrier() returns once all of them have called it.
// barrier() returns false once any thread of the team has failed (an allocation, in practice): everybody then leaves at its
// next barrier instead of waiting for a thread that is gone, and host_team reports the failure.
template <class F>
static bool host_team(size_t n, F f) {
    const unsigned t_max = (unsigned)std::min<size_t>(8, std::max<size_t>(1, n >> 16));
    std::atomic<unsigned> arrived{0}, generation{0};
    std::atomic<bool> failed{false};
    auto barrier = [&]() -> bool {
        const unsigned g = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == t_max) {
            arrived.store(0, std::memory_order_relaxed);
            generation.store(g + 1, std::memory_order_release);
        } else {
            while (generation.load(std::memory_order_acquire) == g && !failed.load(std::memory_order_acquire)) std::this_thread::yield();
        }
        return !failed.load(std::memory_order_acquire);
    };
    auto run = [&](unsigned t) {
        try {
            f(t, t_max, barrier);
        } catch (...) {
            failed.store(true, std::memory_order_release);
        }
    };
    // the team's threads are kept between calls (TeamPool): starting and joining seven std::threads was 0.5-1 ms of the 2-3 ms
    // the three filter passes over configs[2]'s 2 M records took.  A second team at the same time (workers of a many-file
    // call that filter large sketches side by side) starts threads of its own as before.
    if (t_max > 1 && TeamPool::instance().run(t_max - 1, [&](unsigned i) { run(i + 1); }, [&] { run(0u); })) return !failed.load();
    std::vector<std::thread> th;
    th.reserve(t_max - 1);
    try {
        for (unsigned t = 1; t < t_max; ++t) th.emplace_back(run, t);
    } catch (...) { // a thread could not be created: the team is short of a member its barriers wait for -- everybody leaves
        failed.store(true, std::memory_order_release);
    }
    if (!failed.load(std::memory_order_acquire)) run(0u);
    for (auto &x : th) x.join();
    return !failed.load();
}

// FilterParams::filter_counts + process_post_filter (filtering.rs:60-87, mod.rs:115-128) for a Mash sketch, without a copy
// of the oversketch: the filters are per-record tests plus one histogram, and only the first final_size survivors (in
// hash order) are ever wanted -- so the count columns are read where the sketcher left them (fh_result_counts), the strand
// test and the histogram run on a few threads, and the scan for survivors stops when it has final_size of them.  configs[2]:
// 2 M hashes -> 10 000.  `fp` is updated like the reference updates it.
static int finish_mash_in_place(fh_sketcher *h, const std::string &name, const finch_sketch_params &sp, finch_filter_params &fp,
                                const FastxStats &st, uint64_t n, uint64_t total_kmers, Sketch &out) {
    const uint32_t *cnt = nullptr, *ext = nullptr;
    uint64_t n_view = 0;
    if (fh_result_counts(h, &cnt, &ext, &n_view) != FH_OK || n_view != n) return FH_ERR_STATE;
    const bool filter_on = fp.filter_on == 1;
    static const bool trace = cfg("trace") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tt0 = trace ? now_ms() : 0;
    double tt1 = 0, tt2 = 0, tt3 = 0;
    // Three passes by ONE team of threads, each thread over its own stretch of the records:
    //   1. the strand filter (filtering.rs:413-432) and, for the error filter, the largest count among what it leaves;
    //   2. the histogram of those counts per thread; thread 0 adds them up and runs guess_filter_threshold (filtering.rs:154-195);
    //   3. the abundance filter (filtering.rs:329-343) and the cut to final_size: every thread lists the survivors of its
    //      stretch (at most final_size -- the first final_size in hash order are all that is wanted); joined in order below.
    //      (With the error filter on, most of a 2 M-hash oversketch of reads are error k-mers below the cutoff: the first
    //      10 000 survivors end around record 1.2 M.)
    std::vector<uint8_t> dropped;
    const bool strand = filter_on && fp.strand_filter > 0.0, errf = filter_on && fp.err_filter > 0.0;
    if (strand) dropped.resize(n); // (every byte is written in pass 1)
    uint8_t *const dw = strand ? dropped.data() : nullptr;
    const uint8_t *const d = dw;
    const double strand_cutoff = fp.strand_filter;
    uint32_t maxes[8] = {0};
    std::vector<uint64_t> hist_part[8];
    std::vector<uint32_t> rows_part[8];
    uint32_t max_count = 0, lo_t = 0, hi_t = UINT32_MAX;
    const size_t want = (size_t)std::min<uint64_t>(n, sp.final_size);
    const bool team_ok = host_team(n, [&](unsigned t, unsigned nt, const auto &barrier) {
        const size_t per = (n + nt - 1) / nt, lo = std::min<size_t>(n, t * per), hi = std::min<size_t>(n, lo + per);
        if (strand || errf) {
            uint32_t m = 0;
            for (size_t i = lo; i < hi; ++i) {
                const uint32_t c = cnt[i];
                bool drop = false;
                if (dw) {
                    if (c >= 16) {
                        const uint32_t e = ext[i];
                        const uint32_t lowest = std::min(e, c - std::min(e, c));
                        drop = !(((double)lowest / (double)c) >= strand_cutoff);
                    }
                    dw[i] = drop;
                }
                if (!drop) m = std::max(m, c);
            }
            maxes[t] = m;
        }
        if (errf) {
            if (!barrier()) return;
            if (t == 0) {
                for (unsigned j = 0; j < nt; ++j) max_count = std::max(max_count, maxes[j]);
            }
            if (!barrier()) return;
            if (max_count <= (1u << 22)) {
                std::vector<uint64_t> &p = hist_part[t];
                p.assign(max_count, 0);
                for (size_t i = lo; i < hi; ++i)
                    if (!(d && d[i]) && cnt[i]) p[cnt[i] - 1] += 1;
            }
            if (!barrier()) return;
            if (t == 0) {
                std::vector<uint64_t> hist_data(max_count, 0);
                if (max_count <= (1u << 22)) {
                    for (unsigned j = 0; j < nt; ++j)
                        for (size_t q = 0; q < hist_part[j].size(); ++q) hist_data[q] += hist_part[j][q];
                } else { // (counts beyond 4 M: one histogram, one thread)
                    for (size_t i = 0; i < n; ++i)
                        if (!(d && d[i]) && cnt[i]) hist_data[cnt[i] - 1] += 1;
                }
                const uint32_t cutoff = guess_filter_threshold_hist(hist_data, fp.err_filter);
                if (fp.has_abun_lo) {
                    if (cutoff > fp.abun_lo) fp.abun_lo = cutoff;
                } else {
                    fp.has_abun_lo = 1;
                    fp.abun_lo = cutoff;
                }
            }
        }
        if (!barrier()) return;
        if (t == 0) {
            const bool abun = filter_on && (fp.has_abun_lo || fp.has_abun_hi);
            lo_t = (abun && fp.has_abun_lo) ? fp.abun_lo : 0u;
            hi_t = (abun && fp.has_abun_hi) ? fp.abun_hi : UINT32_MAX;
        }
        if (!barrier()) return;
        std::vector<uint32_t> &p = rows_part[t];
        for (size_t i = lo; i < hi && p.size() < want; ++i)
            if (!(d && d[i]) && lo_t <= cnt[i] && cnt[i] <= hi_t) p.push_back((uint32_t)i);
    });
    if (!team_ok) return hfail(FH_ERR_CAPACITY, "out of host memory");
    if (trace) tt1 = tt2 = now_ms();
    std::vector<uint32_t> rows;
    rows.reserve(want);
    for (const auto &p : rows_part) // (thread t took the t-th stretch: rows_part[0] is the lowest)
        for (uint32_t r : p) {
            if (rows.size() >= want) break;
            rows.push_back(r);
        }
    if (trace) tt3 = now_ms();
    if (!sp.no_strict && rows.size() < sp.final_size)
        return hfail(FH_ERR_INVALID, "%s had too few kmers (%zu) to sketch", name.c_str(), rows.size());
    const uint32_t k = sp.kmer_length;
    std::unique_ptr<fh_kmer_count[]> recs(new fh_kmer_count[rows.size() + 1]);
    std::unique_ptr<uint8_t[]> km(new uint8_t[rows.size() * (size_t)k + 1]);
    if (int rc = fh_copy_out_rows(h, rows.data(), rows.size(), recs.get(), km.get())) return hfail(rc, "%s", fh_last_error());
    out.name = name;
    out.seq_length = st.total_bases;
    out.num_valid_kmers = total_kmers;
    out.comment = "";
    out.hashes.resize(rows.size());
    for (size_t i = 0; i < rows.size(); ++i)
        out.hashes[i] = KmerCount{recs[i].hash, KmerBytes((const char *)km.get() + i * (size_t)k, (size_t)k), recs[i].count, recs[i].extra_count};
    out.filter_params = fp;
    out.sketch_params = sp;
    if (trace)
        fprintf(stderr, "[finch] filters in place (n=%llu): three passes %.2f ms, joining the survivors %.2f ms, rows+records %.2f ms\n",
                (unsigned long long)n, tt1 - tt0, tt3 - tt2, now_ms() - tt3);
    return FH_OK;
}

static int finish_sketch(fh_sketcher *h, const std::string &name, const finch_sketch_params &sp, const finch_filter_params &filters,
                         const FastxStats &st, Sketch &out) {
    finch_filter_params fp = filters;
    // lib.rs:70-76: filtering defaults to off for FASTA, on for FASTQ
    if (fp.filter_on < 0) fp.filter_on = st.format == 2 ? 1 : 0;
    uint64_t n = 0, total_kmers = 0;
    if (int rc = fh_finish(h, &n, &total_kmers)) return hfail(rc, "%s", fh_last_error());
    if (g_ktimes_on.load(std::memory_order_relaxed)) { // (since the handle's reset: this input's launches)
        double ms = 0;
        uint64_t nl = 0, np = 0;
        if (fh_kernel_time(h, &ms, &nl, &np) == FH_OK) {
            g_ktimes_us += (uint64_t)(ms * 1000.0 + 0.5);
            g_ktimes_launches += nl;
            g_ktimes_positions += np;
        }
    }
    const uint32_t k = sp.kmer_length;
    if (n > UINT32_MAX) return hfail(FH_ERR_UNSUPPORTED, "sketch of %llu hashes", (unsigned long long)n);
    if (sp.kind == 0) { // a Mash sketch: cut to final_size right behind the filters
        const int rc = finish_mash_in_place(h, name, sp, fp, st, n, total_kmers, out);
        if (rc != FH_ERR_STATE) return rc; // (FH_ERR_STATE: the result is not laid out as columns; the general way below)
    }
    // (arrays the library fills completely: allocated without zeroing -- 2 M hashes are 100 MB here)
    std::unique_ptr<fh_kmer_count[]> recs(new fh_kmer_count[n ? n : 1]);
    if (int rc = fh_copy_out_records(h, recs.get(), nullptr, nullptr)) return hfail(rc, "%s", fh_last_error());
    std::vector<KmerRef> hashes(n);
    for (uint64_t i = 0; i < n; ++i) hashes[i] = KmerRef{recs[i].hash, recs[i].count, recs[i].extra_count, (uint32_t)i};
    recs.reset();
    // (a Mash sketch is cut to final_size right after the filters: the abundance pass may stop there)
    const size_t keep = sp.kind == 0 ? (size_t)sp.final_size : SIZE_MAX;
    std::vector<KmerRef> filtered = filter_counts(fp, std::move(hashes), keep); // lib.rs:82
    if (int rc = process_post_filter(sp, filtered, name)) return rc;            // lib.rs:83
    // the k-mer bytes of what is left (10 000 of a 2 M-hash oversketch)
    std::vector<uint32_t> rows(filtered.size());
    for (size_t i = 0; i < filtered.size(); ++i) rows[i] = filtered[i].row;
    std::unique_ptr<uint8_t[]> km(new uint8_t[filtered.size() * (size_t)k + 1]);
    if (int rc = fh_copy_out_kmers(h, rows.data(), rows.size(), km.get())) return hfail(rc, "%s", fh_last_error());
    out.name = name;
    out.seq_length = st.total_bases;
    out.num_valid_kmers = total_kmers;
    out.comment = "";
    out.hashes.resize(filtered.size());
    for (size_t i = 0; i < filtered.size(); ++i) {
        const KmerRef &r = filtered[i];
        out.hashes[i] = KmerCount{r.hash, KmerBytes((const char *)km.get() + i * (size_t)k, (size_t)k), r.count, r.extra_count};
    }
    out.filter_params = fp;
    out.sketch_params = sp;
    return FH_OK;
}

// FASTA text raw[0, n) (raw[0] == '>') -> its packed stream at dst: the records' sequence regions without their blanks
// (fh_strip.h), one breaker byte behind every record; st gets the records and total_bases as parse_fastx counts them (a record
// starts at a line that begins with '>', its sequence region runs to the next such line, internal newlines count --
// mash.rs:72 --, one trailing line end is trimmed).  false: dst[0, cap) does not hold it (nothing useful written).
static bool pack_fasta_text(const uint8_t *raw, size_t n, uint8_t *dst, size_t cap, FastxStats &st, size_t &m_out) {
    size_t pos = 0, m = 0; // pos: at the '>' of a header line
    while (pos < n) {
        const uint8_t *nl = (const uint8_t *)memchr(raw + pos, '\n', n - pos);
        const size_t start = nl ? (size_t)(nl - raw) + 1 : n; // the sequence region begins behind the header line
        // the next header: a '>' at a line start
        size_t next = n;
        for (size_t q = start; q < n;) {
            const uint8_t *g = (const uint8_t *)memchr(raw + q, '>', n - q);
            if (!g) break;
            const size_t at = (size_t)(g - raw);
            if (at == start || raw[at - 1] == '\n') {
                next = at;
                break;
            }
            q = at + 1;
        }
        const size_t len = next - start;
        uint64_t trim = 0;
        if (len >= 1 && raw[next - 1] == '\n') trim = (len >= 2 && raw[next - 2] == '\r') ? 2 : 1;
        else if (len >= 1 && raw[next - 1] == '\r') trim = 1;
        st.total_bases += len - trim;
        st.n_records++;
        if (m + len + 33 > cap) return false;
        m += fh_strip::strip(dst + m, raw + start, len);
        dst[m++] = 0; // the record's breaker
        pos = next;
    }
    m_out = m;
    return true;
}

// The same walk with the sequence leaving in the batch sketcher's two-bit form (fh_pack2.h), and piece by piece: the worker
// reads a file in pieces the core's L2 keeps and hands each on while it is there, so a genome's text is never written to
// memory and read back (16 workers doing that moved four times the bytes the link does).  `region` has room for
// fh_pack2::region_bytes(bytes of the file): a file never has more positions than bytes, a record's breaker stands where
// its header stood.
struct FastaTwoBit {
    fh_pack2::Packer &pk;
    FastxStats &st;
    bool in_header = true;      // the text begins with a header line (the caller has seen its '>')
    bool at_line_start = false; // the byte in front of the next piece was a line end
    uint64_t region_len = 0;    // bytes of the current record's sequence region so far
    uint8_t last1 = 0, last2 = 0; // ... its last and second-to-last byte
    FastaTwoBit(fh_pack2::Packer &p, FastxStats &s, uint8_t *region) : pk(p), st(s) { pk.begin(region); }
    void close_record() {
        uint64_t trim = 0; // one trailing line end is not sequence (parse_fastx: the region ends in front of it)
        if (region_len >= 1 && last1 == '\n') trim = (region_len >= 2 && last2 == '\r') ? 2 : 1;
        else if (region_len >= 1 && last1 == '\r') trim = 1;
        st.total_bases += region_len - trim;
        st.n_records++;
        pk.byte(0); // the record's breaker
        region_len = 0; // (a header line the text ends in is a record of no bases)
    }
    void piece(const uint8_t *p, size_t n) {
        const uint8_t *const e = p + n;
        while (p < e) {
            if (in_header) {
                const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
                if (!nl) return; // (the header goes on in the next piece; at_line_start is not looked at inside one)
                p = nl + 1;
                in_header = false;
                at_line_start = true;
                region_len = 0;
                continue;
            }
            // the sequence region runs to the next '>' at a line start
            const uint8_t *stop = e;
            bool header = false;
            for (const uint8_t *q = p; q < e;) {
                const uint8_t *g = (const uint8_t *)memchr(q, '>', (size_t)(e - q));
                if (!g) break;
                if (g == p ? at_line_start : g[-1] == '\n') {
                    stop = g;
                    header = true;
                    break;
                }
                q = g + 1;
            }
            const size_t len = (size_t)(stop - p);
            if (len) {
                pk.text(p, len);
                last2 = len >= 2 ? stop[-2] : last1;
                last1 = stop[-1];
                region_len += len;
                at_line_start = last1 == '\n';
            }
            p = stop;
            if (header) {
                close_record();
                in_header = true;
            }
        }
    }
    uint64_t finish() { // -> positions
        close_record(); // (a header line without a sequence region behind it is a record of no bases)
        return pk.finish();
    }
};

// A plain FASTA file that fits the staging buffer twice over (a genome of a batch: configs[4]) is packed on the HOST, in
// one pass, while it is staged: the file is read into the upper part of the sketcher's pinned staging buffer and its
// sequence regions are copied to the front without their blanks (fh_strip.h), one breaker byte per record -- the packed
// stream fh_push_staged commits.  The device then runs THREE launches for the file (queue reset, sketch kernel, the fused
// epilogue that also leaves the handle reset) behind one host-to-device copy, and the host waits once.  The device-side
// splitter (fh_push_fasta_text: five more launches and a round trip for the packed length) stays what large inputs go through,
// where the host could not strip at the rate the link moves text.  Same records, same total_bases as parse_fastx:
// a record starts at a line that begins with '>', its sequence region runs to the next such line, internal newlines count
// (mash.rs:72), one trailing line end is trimmed.
// -> FH_OK, or FH_ERR_STATE ("does not apply": nothing consumed that a rewind does not give back), or an error.
static int fasta_small_on_host(ByteSource &src, fh_sketcher *h, FastxStats &st) {
    const uint64_t hint = src.remaining_hint();
    if (hint == UINT64_MAX || !src.can_rewind()) return FH_ERR_STATE;
    uint8_t *buf = nullptr;
    uint64_t cap = 0;
    if (int rc = fh_text_buffer(h, &buf, &cap)) return hfail(rc, "%s", fh_last_error());
    if (hint + 4096 > cap) return FH_ERR_STATE;
    // the raw text goes to the upper part of the staging buffer if the packed stream has room in front of it, else to a heap
    // buffer this worker keeps
    static thread_local std::vector<uint8_t> scratch;
    uint8_t *raw;
    size_t room;
    if (2 * hint + 4096 <= cap) {
        raw = buf + ((cap - hint - 64) & ~(uint64_t)63);
        room = (size_t)(buf + cap - raw);
    } else {
        if (scratch.size() < hint + 64) scratch.resize((size_t)hint + 64 + (hint >> 2));
        raw = scratch.data();
        room = scratch.size();
    }
    // (the source may deliver more than it said -- a file that grew since its size was asked for, FileSource::read asks
    // again --, and everything below is sized by what it said: the packed stream, at most one byte per byte read plus
    // fh_strip's 32 bytes of slack, fits the staging buffer because hint + 4096 <= cap.  One byte more than the hint is "longer
    // than it said".)
    room = std::min<size_t>(room, (size_t)hint + 1);
    size_t n = 0;
    for (;;) {
        const size_t g = src.read(raw + n, room - n);
        if (g == 0) break;
        n += g;
        if (n == room) { // longer than it said: not for this path
            if (!src.rewind()) return hfail(FH_ERR_INVALID, "input grew while it was read");
            return FH_ERR_STATE;
        }
    }
    if (src.failed()) return hfail(FH_ERR_INVALID, "corrupt compressed stream");
    if (n == 0 || raw[0] != '>') {
        if (!src.rewind()) return hfail(FH_ERR_INVALID, "not a FASTA file");
        return FH_ERR_STATE;
    }
    st.format = 1;
    size_t m = 0;
    if (!pack_fasta_text(raw, n, buf, cap, st, m)) { // (cannot happen with n <= hint; a guard in front of the one place that writes)
        if (!src.rewind()) return hfail(FH_ERR_INVALID, "input grew while it was read");
        st = FastxStats();
        return FH_ERR_STATE;
    }
    if (int rc = fh_push_staged(h, m, 0u)) return hfail(rc, "%s", fh_last_error());
    return FH_OK;
}

static int sketch_stream(std::unique_ptr<ByteSource> raw, const std::string &name, const finch_sketch_params &sp,
                         const finch_filter_params &filters, HandleSet &handles, Sketch &out) {
    std::unique_ptr<ByteSource> src;
    bool is_gz = false;
    int first = -1;
    if (int rc = open_source(std::move(raw), src, &is_gz, &first)) return rc;
    FastxStats st;
    const char *dp = cfg("device_parse");
    // FINCH_DEVICE_PARSE: unset = plain FASTA and FASTQ text is split on the device (FASTQ with the host parser as the
    // fallback, see below); 1 = on the device, no fallback; 0 = on the host.  Compressed input is inflated on the host
    // and its text treated the same way.
    const bool dp_on = dp && dp[0] == '1', dp_off = dp && dp[0] == '0';
    // FINCH_DEVICE_INFLATE: unset / 1 = BGZF-compressed FASTQ is inflated on the device (with the host-side inflate as the
    // fallback whenever the device pass refuses the file); 0 = always on the host
    BgzfSource *bgzf_dev = nullptr;
    if (is_gz && !dp_off) {
        const char *di = cfg("device_inflate");
        BgzfSource *bz = dynamic_cast<BgzfSource *>(src.get());
        if (bz && !(di && di[0] == '0') && src->can_rewind() && bz->peek_first_text_byte() == '@') bgzf_dev = bz;
    }
    // ... and so is plain gzip (FINCH_DEVICE_GZIP=0: on the host, by the call's read threads together, fh_pargz.h)
    bool gzip_dev = false;
    size_t gzip_hdr = 0;
    if (is_gz && !dp_off && !bgzf_dev) {
        const char *di = cfg("device_inflate"), *dg = cfg("device_gzip");
        BgzfSource *bz = dynamic_cast<BgzfSource *>(src.get());
        if (bz && !(di && di[0] == '0') && !(dg && dg[0] == '0') && src->can_rewind() && bz->peek_plain_gzip(&gzip_hdr) == '@') gzip_dev = true;
    }
    if (bgzf_dev || gzip_dev) {
        first = '@';
    } else if (is_gz) {
        // compressed: the format shows in the first inflated byte; what follows it reaches the staging buffer straight
        // from the decompressor, so the host only inflates (gzip: one thread, BGZF: the call's read threads)
        uint8_t b = 0;
        const size_t g = src->read(&b, 1);
        auto pre = std::make_unique<PrefixedSource>();
        pre->prefix.assign(&b, &b + g);
        pre->inner = std::move(src);
        src = std::move(pre);
        first = g ? (int)b : -1;
    }
    // which sketcher: needletail takes the format from the first byte, and the format decides the filtering default
    const int filter_on_eff = filters.filter_on < 0 ? (first == '@' ? 1 : 0) : filters.filter_on;
    const bool small = (first == '>' || first == '@') && sp.kind == 0 && filter_on_eff == 0 && sp.final_size >= 1 &&
                       sp.final_size < sp.kmers_to_sketch && cfg("no_small_sketcher") == nullptr;
    fh_sketcher *h = handles.get(small);
    if (!h) return hfail(FH_ERR_NO_DEVICE, "%s", fh_last_error());
    if (int rc = fh_reset(h)) return hfail(rc, "%s", fh_last_error());
    if (bgzf_dev) {
        st.format = 2;
        const int rc = bgzf_fastq_to_device(*bgzf_dev, h);
        if (rc == FH_OK) {
            g_bgzf_on_device++;
            if (int r2 = fh_text_bases(h, &st.total_bases)) return hfail(r2, "%s", fh_last_error());
            return finish_sketch(h, name, sp, filters, st, out);
        }
        if (rc != FH_ERR_INVALID || !src->rewind()) return rc;
        g_bgzf_reread++;
        if (int r2 = fh_reset(h)) return hfail(r2, "%s", fh_last_error());
        // (the text now comes through the host-side inflate; its first byte is known)
    }
    if (gzip_dev) {
        st.format = 2;
        const int rc = gzip_fastq_to_device(*static_cast<BgzfSource *>(src.get()), gzip_hdr, h);
        if (rc == FH_OK) {
            g_gzip_on_device++;
            if (int r2 = fh_text_bases(h, &st.total_bases)) return hfail(r2, "%s", fh_last_error());
            return finish_sketch(h, name, sp, filters, st, out);
        }
        { // FINCH_DEVICE_GZIP=1: the device pass or nothing (its refusals stay loud)
            const char *dg = cfg("device_gzip");
            if (dg && dg[0] == '1') return rc;
        }
        if (rc != FH_ERR_INVALID || !src->rewind()) return rc;
        g_gzip_reread++;
        if (int r2 = fh_reset(h)) return hfail(r2, "%s", fh_last_error());
    }
    bool device_parse = !dp_off && (first == '>' || first == '@');
    bool fastq_host_counted = false; // the FASTQ text was stripped on the host (fastq_host_strip_to_device): st has the totals
    if (device_parse && first == '@' && !dp_on && !src->can_rewind()) device_parse = false; // no second chance: host parser
    if (device_parse && first == '@') {
        // FASTQ on the device has to be strictly 4-line.  Unless the caller insists (FINCH_DEVICE_PARSE=1: errors stay
        // loud), a file the device pass rejects is read again through the host parser, which is the judge of what
        // needletail accepts (blank lines between records, ...); sources that cannot rewind start on the host.
        st.format = 2;
        const int rc = fastq_text_to_device(*src, h, sp.kmer_length, &st, &fastq_host_counted);
        if (rc != FH_OK) {
            if (dp_on || rc != FH_ERR_INVALID || !src->rewind()) return rc;
            if (int r2 = fh_reset(h)) return hfail(r2, "%s", fh_last_error());
            device_parse = false;
            fastq_host_counted = false;
            st = FastxStats{};
            st.format = 2;
        }
    }
    if (device_parse && first == '>') {
        st.format = 1;
        // FINCH_SMALL_FASTA_HOST: unset / 1 = a small plain file is packed while it is staged (fasta_small_on_host), 0 = never
        const char *sh_env = cfg("small_fasta_host");
        const bool small_host = !(sh_env && sh_env[0] == '0');
        int rc = (small_host && !dp_on && !is_gz) ? fasta_small_on_host(*src, h, st) : FH_ERR_STATE;
        if (rc == FH_ERR_STATE) { // does not apply: split on the device
            st = FastxStats{};
            st.format = 1;
            rc = fasta_text_to_device(*src, h, st, sp.kmer_length);
        }
        if (rc) return rc;
    } else if (!device_parse) {
        DeviceSink sink(h);
        if (int rc = parse_fastx(*src, sink, st)) return rc;
        if (int rc = sink.flush()) return rc;
    }
    if (device_parse && st.format == 2 && !fastq_host_counted) {
        if (int rc = fh_text_bases(h, &st.total_bases)) return hfail(rc, "%s", fh_last_error());
    }
    return finish_sketch(h, name, sp, filters, st, out);
}

// ---------------------------------------------------------------------------------------------
// One input across several devices (north_star: "a single large input is partitioned by read blocks across the
// GPUs of one node with a final host-side merge of the tiny partial sketches").  The reference parallelises over
// files only (lib.rs:34-36); this goes beyond it on the strength of SURVEY 8e: a sketch is a function of the multiset
// of k-mers, so the partial sketches of any partition of the input merge exactly.
//
// One reader (the calling thread) cuts the decompressed text into chunks -- FASTQ: after a whole record; FASTA: after
// a newline -- and deals them round-robin to one worker per device handle.  A worker copies its chunk into its
// handle's pinned staging buffer, tells the handle where the chunk sits in the input (fh_set_stream_offset: text
// offsets order the chunks of all handles, which is all "first occurrence" needs) and has the device split and sketch it
// (fh_push_fastq_text / fh_push_fasta_text).  A FASTA record may span chunks: the reader keeps the last k-1 sequence
// bytes before each cut and the worker passes them along as the chunk's halo (fh_set_text_halo), so that the k-mers
// across a cut are formed exactly once, on the device that takes the later chunk.  Then: fh_finish per handle,
// fh_merge into the first, filters, Sketch.
// ---------------------------------------------------------------------------------------------
struct TextBuf { // a chunk buffer of the text readers: heap memory, or one of a sketcher's pinned staging buffers
    uint8_t *p = nullptr;
    size_t cap = 0;
    int id = 0;
    uint8_t *data() const { return p; }
    size_t size() const { return cap; }
};

struct ShardWork {
    TextBuf *buf = nullptr;
    size_t len = 0;
    uint64_t text_off = 0;
    uint32_t start_state = 0;
    uint8_t halo[64];
    uint32_t halo_len = 0;
    bool stop = false;
};

struct ShardQueue { // one per worker, depth <= 2
    std::mutex mu;
    std::condition_variable cv;
    std::vector<ShardWork> q;
};

// Last `want` kept (non-whitespace) sequence bytes of the record that text[0, cut) ends in the middle of; none if
// the text ends in a header line.  `state0` = what the text starts in the middle of (0 line start, 1 sequence line,
// 2 header line); `prev` = the same answer for the text before it (used when the walk reaches the start of this text).
static void fasta_tail(const uint8_t *text, size_t cut, uint32_t state0, uint32_t want, const uint8_t *prev, uint32_t prev_len,
                       uint8_t *out, uint32_t *out_len) {
    uint8_t rev[64];
    uint32_t n = 0;
    size_t pos = cut;
    bool boundary = false;
    while (pos > 0 && n < want) {
        const size_t scan_end = text[pos - 1] == '\n' ? pos - 1 : pos; // the line ending at pos (its newline included)
        const void *nl = scan_end ? memrchr(text, '\n', scan_end) : nullptr;
        const size_t ls = nl ? (size_t)((const uint8_t *)nl - text) + 1 : 0;
        const bool header = ls > 0 ? text[ls] == '>' : (state0 == 2 || (state0 == 0 && text[0] == '>'));
        if (header) {
            boundary = true;
            break;
        }
        for (size_t i = pos; i > ls && n < want; --i) {
            const uint8_t c = text[i - 1];
            if (c == ' ' || c == '\t' || c == '\r' || c == '\n') continue;
            rev[n++] = c;
        }
        pos = ls;
    }
    if (!boundary && n < want && pos == 0) // the record began before this text
        for (uint32_t i = prev_len; i > 0 && n < want; --i) rev[n++] = prev[i - 1];
    for (uint32_t i = 0; i < n; ++i) out[i] = rev[n - 1 - i];
    *out_len = n;
}

// The reader of a sharded input: cuts the text into chunks and hands them to `emit` with everything a sketcher needs to
// take the chunk on its own (offset in the text, FASTA start state and halo).  Also counts what needs no per-base work
// (FASTA: records, total_bases).
template <class Take, class Give, class Emit>
static int shard_reader(ByteSource &src_ref, bool fastq, uint32_t K, Take take_buf, Give give_back, Emit emit,
                        const std::atomic<bool> &abort, FastxStats &st) {
    ByteSource *src = &src_ref;
    int rrc = FH_OK;
    {
        std::vector<uint8_t> left;
        bool eof = false;
        uint64_t text_off = 0; // offset of the next chunk's first byte in the decompressed text
        FastaCounter fc;
        std::vector<size_t> gt_pos;
        uint8_t halo[64];
        uint32_t halo_len = 0;
        while ((!eof || !left.empty()) && !abort) {
            TextBuf *b = take_buf();
            uint8_t *buf = b->data();
            const size_t cap = b->size();
            if (left.size() >= cap && fastq) {
                give_back(b);
                rrc = hfail(FH_ERR_INVALID, "FASTQ record longer than the staging buffer");
                break;
            }
            size_t fill = std::min(left.size(), cap);
            memcpy(buf, left.data(), fill);
            left.erase(left.begin(), left.begin() + (long)fill);
            while (!eof && fill < cap) {
                const size_t got = src->read(buf + fill, cap - fill);
                if (got == 0) eof = true;
                fill += got;
            }
            if (fill == 0) {
                give_back(b);
                break;
            }
            size_t cut = fill;
            if (fastq) {
                if (!eof) { // the last header line whose line two below is a '+' line (see fastq_text_to_device)
                    size_t ls[16];
                    int n = 0;
                    size_t pos = fill;
                    while (n < 16) {
                        const uint8_t *nl = pos > 0 ? (const uint8_t *)memrchr(buf, '\n', pos) : nullptr;
                        const size_t start = nl ? (size_t)(nl - buf) + 1 : 0;
                        if (start < fill) ls[n++] = start;
                        if (!nl) break;
                        pos = (size_t)(nl - buf);
                    }
                    cut = 0;
                    bool found = false;
                    for (int i = 2; i < n && !found; ++i)
                        if (buf[ls[i]] == '@' && buf[ls[i - 2]] == '+') {
                            cut = ls[i];
                            found = true;
                        }
                    if (!found) {
                        give_back(b);
                        rrc = hfail(FH_ERR_INVALID, "no FASTQ record boundary found in a %zu byte chunk", fill);
                        break;
                    }
                    left.assign(buf + cut, buf + fill);
                }
            } else if (!eof || !left.empty()) {
                const uint8_t *nl = (const uint8_t *)memrchr(buf, '\n', fill);
                if (nl) cut = (size_t)(nl - buf) + 1; // else: one line longer than the buffer, cut anywhere
                left.insert(left.begin(), buf + cut, buf + fill);
            }
            if (cut == 0) {
                give_back(b);
                continue;
            }
            ShardWork job;
            job.buf = b;
            job.len = cut;
            job.text_off = text_off;
            if (!fastq) {
                job.start_state = fc.start_state();
                // a chunk that begins inside a record (not in or at a header line) carries the k-1 bases before it
                job.halo_len = 0;
                if (job.start_state != 2 && fc.have_record && halo_len) {
                    memcpy(job.halo, halo, halo_len);
                    job.halo_len = halo_len;
                }
                uint8_t nh[64];
                uint32_t nh_len = 0;
                fasta_tail(buf, cut, job.start_state, K - 1, halo, (job.start_state != 2 && fc.have_record) ? halo_len : 0u, nh, &nh_len);
                if (cut >= ((size_t)8 << 20) && src->threads_hint() > 1) {
                    FastaCounter::find_gt(buf, cut, src->threads_hint(), gt_pos);
                    fc.feed(buf, cut, &gt_pos);
                } else {
                    fc.feed(buf, cut);
                }
                memcpy(halo, nh, nh_len);
                halo_len = nh_len;
            }
            text_off += cut;
            emit(job);
        }
        if (rrc == FH_OK && src->failed()) rrc = hfail(FH_ERR_INVALID, "read error or corrupt compressed stream");
        if (!fastq) {
            fc.finish();
            st.total_bases = fc.total_bases;
            st.n_records = fc.n_records;
        }
    }
    return rrc;
}

// The device-side text paths of ONE sketcher (finch_sketch_files / finch_sketch_buffer): the reader -- file reads, inflate,
// the search for a record boundary, the FASTA bookkeeping -- runs on a thread of its own and fills one of the sketcher's
// two pinned staging buffers while the calling thread has the other one copied to the device, split into records and
// sketched (fh_push_fastq_text / fh_push_fasta_text synchronise with the device once per chunk).  Compressed input gains
// most: the inflate no longer waits for the pushes.  Chunks of a FASTA file continue each other on the same sketcher
// (FH_PUSH_CONTINUE carries the k-1 bases on the device; the halo shard_reader computes is for the sharded path).
static int pump_text_to_device(ByteSource &src, fh_sketcher *h, bool fastq, uint32_t k, FastxStats &st) {
    uint8_t *raw[2] = {nullptr, nullptr};
    uint64_t cap = 0;
    int next = 0;
    if (int rc = fh_text_buffers(h, raw, &cap, &next)) return hfail(rc, "%s", fh_last_error());
    if (!fastq) cap = std::min<uint64_t>(cap, (1ull << 30) - 1);
    TextBuf tb[2] = {TextBuf{raw[0], (size_t)cap, 0}, TextBuf{raw[1], (size_t)cap, 1}};
    std::mutex mu;
    std::condition_variable cv;
    bool is_free[2] = {true, true}, producer_done = false;
    int fill = next; // the slot the next chunk goes to: pushes consume the slots alternately, starting with `next`
    std::vector<ShardWork> ready;
    std::atomic<bool> abort{false};
    int prc = FH_OK;
    std::string pmsg;
    FastxStats pst;
    // FH_TRACE: how long each side waited for the other, and what the pushes took
    static const bool trace = cfg("trace") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_s();
    double t_reader_waits = 0, t_pusher_waits = 0, t_push = 0;
    unsigned n_chunks = 0;
    if (src.remaining_hint() < cap) {
        // The whole input fits one staging buffer (a genome of a batch: configs[4]): read it and push it right here -- no reader
        // thread to start, hand over to and join per file.
        int rc = FH_OK;
        std::string msg;
        const int prc1 = shard_reader(
            src, fastq, k, [&] { return &tb[fill]; }, [&](TextBuf *) {},
            [&](const ShardWork &job) {
                n_chunks++;
                const double tw1 = trace ? now_s() : 0;
                if (rc == FH_OK) {
                    rc = fastq ? fh_push_fastq_text(h, job.len) : fh_push_fasta_text(h, job.len, job.start_state, n_chunks == 1 ? 0u : FH_PUSH_CONTINUE);
                    if (rc != FH_OK) {
                        msg = fh_last_error();
                        abort = true;
                    }
                }
                if (trace) t_push += now_s() - tw1;
                fill ^= 1;
            },
            abort, pst);
        if (trace)
            fprintf(stderr, "[finch] text pump (inline): %u chunk(s) in %.1f ms, pushes took %.1f ms\n", n_chunks, (now_s() - t_begin) * 1e3, t_push * 1e3);
        if (rc != FH_OK) return hfail(rc, "%s", msg.c_str());
        if (prc1 != FH_OK) return prc1;
        st.total_bases = pst.total_bases;
        st.n_records = pst.n_records;
        return FH_OK;
    }
    std::thread producer([&] {
        const int rc = shard_reader(
            src, fastq, k,
            [&] {
                const double t0 = trace ? now_s() : 0;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return is_free[fill] || abort.load(); });
                is_free[fill] = false;
                if (trace) t_reader_waits += now_s() - t0;
                return &tb[fill];
            },
            [&](TextBuf *b) { // taken but not used
                std::lock_guard<std::mutex> g(mu);
                is_free[b->id] = true;
            },
            [&](const ShardWork &job) {
                // the chunk's copy to the device starts now, behind the previous chunk's, while that one's push is still busy
                // with its record-splitting kernel -- the link never idles between pushes
                (void)fh_text_prefetch(h, job.buf->id, job.len);
                std::lock_guard<std::mutex> g(mu);
                ready.push_back(job);
                fill ^= 1;
                cv.notify_all();
            },
            abort, pst);
        std::lock_guard<std::mutex> g(mu);
        prc = rc;
        if (rc != FH_OK) pmsg = g_host_err; // (thread-local: carried over to the caller's thread below)
        producer_done = true;
        cv.notify_all();
    });
    int rc = FH_OK;
    std::string msg;
    bool first = true;
    for (;;) {
        ShardWork job;
        const double tw0 = trace ? now_s() : 0;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || producer_done; });
            if (ready.empty()) break;
            job = ready.front();
            ready.erase(ready.begin());
        }
        const double tw1 = trace ? now_s() : 0;
        t_pusher_waits += tw1 - tw0;
        n_chunks++;
        if (rc == FH_OK) {
            rc = fastq ? fh_push_fastq_text(h, job.len) : fh_push_fasta_text(h, job.len, job.start_state, first ? 0u : FH_PUSH_CONTINUE);
            if (trace) t_push += now_s() - tw1;
            if (rc != FH_OK) {
                msg = fh_last_error();
                abort = true;
            }
            first = false;
        }
        std::lock_guard<std::mutex> g(mu);
        is_free[job.buf->id] = true;
        cv.notify_all();
    }
    producer.join();
    if (trace)
        fprintf(stderr, "[finch] text pump: %u chunks of <= %.0f MiB in %.1f ms: reader waited %.1f ms for a buffer, pushes took %.1f ms and waited %.1f ms for text\n",
                n_chunks, cap / 1048576.0, (now_s() - t_begin) * 1e3, t_reader_waits * 1e3, t_push * 1e3, t_pusher_waits * 1e3);
    if (rc != FH_OK) return hfail(rc, "%s", msg.c_str());
    if (prc != FH_OK) return hfail(prc, "%s", pmsg.c_str());
    st.total_bases = pst.total_bases;
    st.n_records = pst.n_records;
    return FH_OK;
}

// BGZF-compressed FASTQ with the inflate on the device (fh_push_bgzf_fastq): the reader thread only moves whole members
// from the file into the sketcher's pinned buffers -- a table of them in front, their DEFLATE bytes behind -- while the
// calling thread has the previous batch copied over, inflated, CRC-checked, split and sketched.  The compressed bytes
// cross PCIe instead of the text, and no host core inflates anything.  Any failure (a member that is not BGZF, damage,
// text that is not plain 4-line FASTQ, a record longer than the buffers' spare room) is FH_ERR_INVALID: the caller
// reads the file again through the host-side inflate, whose errors are the ones reported.
static int bgzf_fastq_to_device(BgzfSource &bz, fh_sketcher *h) {
    uint8_t *raw[2] = {nullptr, nullptr};
    uint64_t cap = 0;
    int next = 0;
    if (int rc = fh_text_buffers(h, raw, &cap, &next)) return hfail(rc, "%s", fh_last_error());
    uint64_t text_cap = 0;
    if (int rc = fh_bgzf_text_capacity(h, &text_cap)) return hfail(rc, "%s", fh_last_error());
    const uint32_t max_members = (uint32_t)std::min<uint64_t>(16384, std::max<uint64_t>(1, cap / 4096));
    // what follows the last whole record of a batch joins the next batch's text: leave it room
    const uint64_t text_budget = text_cap - std::min<uint64_t>(text_cap / 4, (uint64_t)32 << 20);
    struct Job {
        int slot;
        uint64_t bytes;
        uint32_t n;
        bool last, more; // more: only copied over; inflated together with the jobs that follow (FH_BGZF_MORE)
    };
    std::mutex mu;
    std::condition_variable cv;
    bool is_free[2] = {true, true}, producer_done = false, producer_ok = true;
    constexpr uint32_t MAX_LAUNCH_MEMBERS = 1u << 16; // (fh_push_bgzf_fastq's limit)
    // members are handed over ~1500 at a time (96 MiB of text): each push starts inflating at once, on one of four side
    // streams, while the next ones are still being read, so the device fills up as the file comes in
    constexpr uint64_t PUSH_TEXT = (uint64_t)96 << 20;
    std::vector<Job> ready;
    std::atomic<bool> abort{false};
    static const bool trace = cfg("trace") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_s();
    double t_reader_waits = 0, t_pusher_waits = 0, t_push = 0, t_read = 0;
    uint64_t n_members = 0, n_bytes = 0;
    unsigned n_batches = 0;
    std::thread producer([&] {
        int slot = next;
        uint64_t acc_text = 0; // text of the members handed over since the last launch
        uint32_t acc_n = 0;
        for (;;) {
            {
                const double t0 = trace ? now_s() : 0;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return is_free[slot] || abort.load(); });
                if (trace) t_reader_waits += now_s() - t0;
                if (abort) break;
                is_free[slot] = false;
            }
            Job job{slot, 0, 0, false, false};
            uint64_t text = 0;
            bool budget_hit = false;
            const double t0 = trace ? now_s() : 0;
            const bool ok = bz.raw_batch(raw[slot], cap, std::min(max_members, MAX_LAUNCH_MEMBERS - acc_n),
                                         std::min(text_budget - acc_text, PUSH_TEXT),
                                         (fh_bgzf_member *)raw[slot], &job.n, &job.bytes, &text, &job.last, &budget_hit);
            if (trace) t_read += now_s() - t0;
            // the push is full before the text budget is: the members that follow join the same batch of text
            acc_text += text;
            acc_n += job.n;
            if (budget_hit && acc_text + 65536 <= text_budget) budget_hit = false; // (only this push's share was used up)
            job.more = ok && !job.last && !budget_hit && job.n > 0 && acc_n < MAX_LAUNCH_MEMBERS;
            if (!job.more) acc_text = acc_n = 0;
            std::lock_guard<std::mutex> g(mu);
            if (!ok) {
                producer_ok = false;
                break;
            }
            if (job.n == 0) job.bytes = 0;
            n_members += job.n;
            n_bytes += job.bytes;
            ready.push_back(job);
            cv.notify_all();
            if (job.last) break;
            slot ^= 1;
        }
        std::lock_guard<std::mutex> g(mu);
        producer_done = true;
        cv.notify_all();
    });
    int rc = FH_OK;
    std::string msg;
    for (;;) {
        Job job;
        const double tw0 = trace ? now_s() : 0;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || producer_done; });
            if (ready.empty()) break;
            job = ready.front();
            ready.erase(ready.begin());
        }
        const double tw1 = trace ? now_s() : 0;
        t_pusher_waits += tw1 - tw0;
        n_batches++;
        if (rc == FH_OK) {
            rc = fh_push_bgzf_fastq(h, job.bytes, job.n, job.last ? FH_BGZF_LAST : job.more ? FH_BGZF_MORE : 0u);
            if (trace) t_push += now_s() - tw1;
            if (rc != FH_OK) {
                msg = fh_last_error();
                abort = true;
            }
        }
        std::lock_guard<std::mutex> g(mu);
        is_free[job.slot] = true;
        cv.notify_all();
    }
    producer.join();
    if (trace)
        fprintf(stderr, "[finch] bgzf on the device: %u batches, %llu members, %.1f MB in %.1f ms: reads %.1f ms, reader waited %.1f ms for a buffer, pushes took %.1f ms and waited %.1f ms for members\n",
                n_batches, (unsigned long long)n_members, n_bytes / 1e6, (now_s() - t_begin) * 1e3, t_read * 1e3, t_reader_waits * 1e3,
                t_push * 1e3, t_pusher_waits * 1e3);
    if (rc != FH_OK) return hfail(rc == FH_ERR_INVALID ? FH_ERR_INVALID : rc, "%s", msg.c_str());
    if (!producer_ok) return hfail(FH_ERR_INVALID, "not a plain chain of BGZF members");
    return FH_OK;
}

// gzip-compressed FASTQ -- one DEFLATE stream -- with the inflate on the device (fh_push_gzip_fastq): the reader thread
// moves the file's bytes into the sketcher's pinned buffers, a buffer's worth per push, while the calling thread has the
// previous push decoded, checked, split and sketched.  Anything but one sound member of plain 4-line FASTQ whose blocks
// fit a push (several members, trailing bytes, damage, text more than about 8 x its DEFLATE bytes) is FH_ERR_INVALID: the
// caller reads the file again through the host-side inflate, whose verdict is the one reported.
static int gzip_fastq_to_device(BgzfSource &bz, size_t hdr_len, fh_sketcher *h) {
    uint8_t *raw[2] = {nullptr, nullptr};
    uint64_t buf_cap = 0, cap = 0;
    int next = 0;
    if (int rc = fh_text_buffers(h, raw, &buf_cap, &next)) return hfail(rc, "%s", fh_last_error());
    if (int rc = fh_gzip_batch_capacity(h, &cap)) return hfail(rc, "%s", fh_last_error());
    cap = std::min(cap, buf_cap);
    if (cap < ((uint64_t)1 << 16)) return hfail(FH_ERR_INVALID, "staging buffers too small for batches of gzip blocks");
    { // the member's header is the host's to skip
        std::vector<uint8_t> skip(hdr_len);
        if (bz.raw_read(skip.data(), hdr_len) != hdr_len) return hfail(FH_ERR_INVALID, "gzip header cut short");
    }
    // A batch fills one buffer, in pieces: every piece is handed over as soon as it has been read (FH_GZ_MORE), so that the
    // device decodes the front of the batch while the rest of it is still coming in.  (What counts is when the LAST byte is
    // there -- every chunk of a batch is resident at once, and a live one takes ~10 ms whenever it starts --, so pieces are
    // as large as it takes for the reads to be split over the call's read threads.)
    static const uint64_t PIECE = [] {
        const char *e = cfg("gzip_piece"); // (A/B)
        return e ? std::max<uint64_t>(65536, strtoull(e, nullptr, 10)) : ((uint64_t)16 << 20); // (what FileSource splits over the call's read threads)
    }();
    struct Job {
        int slot;
        uint64_t bytes;
        bool more, last;
    };
    std::mutex mu;
    std::condition_variable cv;
    bool is_free[2] = {true, true}, producer_done = false;
    std::vector<Job> ready;
    std::atomic<bool> abort{false};
    static const bool trace = cfg("trace") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_s();
    double t_read = 0, t_push = 0, t_wait = 0;
    uint64_t n_bytes = 0;
    unsigned n_pushes = 0, n_batches = 0;
    std::thread producer([&] {
        int slot = next;
        uint64_t acc = 0;
        for (;;) {
            if (acc == 0) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return is_free[slot] || abort.load(); });
                if (abort) break;
                is_free[slot] = false;
            } else if (abort) {
                break;
            }
            const double t0 = trace ? now_s() : 0;
            const uint64_t want = std::min<uint64_t>(PIECE, cap - acc);
            const size_t got = bz.raw_read(raw[slot] + acc, (size_t)want);
            if (trace) t_read += now_s() - t0;
            acc += got;
            const bool eof = got < want;
            Job job{slot, got, !eof && cap - acc >= ((uint64_t)1 << 16), eof};
            {
                std::lock_guard<std::mutex> g(mu);
                n_bytes += got;
                ready.push_back(job);
                cv.notify_all();
            }
            if (eof) break;
            if (!job.more) {
                slot ^= 1;
                acc = 0;
            }
        }
        std::lock_guard<std::mutex> g(mu);
        producer_done = true;
        cv.notify_all();
    });
    int rc = FH_OK;
    std::string msg;
    bool first = true, done = false;
    for (;;) {
        Job job;
        const double tw0 = trace ? now_s() : 0;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !ready.empty() || producer_done; });
            if (ready.empty()) break;
            job = ready.front();
            ready.erase(ready.begin());
        }
        const double tw1 = trace ? now_s() : 0;
        t_wait += tw1 - tw0;
        if (rc == FH_OK) {
            if (done) { // bytes behind the member's end
                if (job.bytes) rc = FH_ERR_INVALID, msg = "more than one gzip member";
            } else {
                uint32_t member_done = 0;
                uint64_t trailing = 0;
                n_pushes++;
                n_batches += !job.more;
                rc = fh_push_gzip_fastq(h, job.bytes, (first ? FH_GZ_FIRST : 0u) | (job.more ? FH_GZ_MORE : 0u) | (job.last ? FH_GZ_LAST : 0u), &member_done,
                                        &trailing);
                first = false;
                if (rc != FH_OK) msg = fh_last_error();
                else if (member_done) {
                    done = true;
                    if (trailing) rc = FH_ERR_INVALID, msg = "more than one gzip member";
                }
            }
            if (trace) t_push += now_s() - tw1;
            if (rc != FH_OK) abort = true;
        }
        if (!job.more) {
            std::lock_guard<std::mutex> g(mu);
            is_free[job.slot] = true;
            cv.notify_all();
        } else if (rc != FH_OK) {
            std::lock_guard<std::mutex> g(mu);
            cv.notify_all();
        }
    }
    producer.join();
    if (trace)
        fprintf(stderr, "[finch] gzip on the device: %u batches in %u pushes, %.1f MB in %.1f ms: reads %.1f ms, pushes took %.1f ms and waited %.1f ms for bytes\n",
                n_batches, n_pushes, n_bytes / 1e6, (now_s() - t_begin) * 1e3, t_read * 1e3, t_push * 1e3, t_wait * 1e3);
    if (rc != FH_OK) return hfail(rc, "%s", msg.c_str());
    if (!done) return hfail(FH_ERR_INVALID, "gzip: the stream ends before its final block");
    return FH_OK;
}

// *device_rejected: the input is FASTQ and the device-side splitter refused a chunk of it (not strictly 4-line, a record
// longer than a chunk): the caller sends the file through the single-handle path, whose host parser is the judge of
// what needletail accepts.
static int sketch_stream_sharded(std::unique_ptr<ByteSource> raw, const std::string &name, const finch_sketch_params &sp,
                                 const finch_filter_params &filters, const std::vector<int> &devs, uint64_t chunk_bytes,
                                 Sketch &out, bool *device_rejected) {
    *device_rejected = false;
    std::unique_ptr<ByteSource> src;
    bool compressed = false;
    int first = -1;
    if (int rc = open_source(std::move(raw), src, &compressed, &first)) return rc;
    { // the format shows in the first (decompressed) byte
        uint8_t b = 0;
        size_t g = 0;
        if (compressed || first < 0) g = src->read(&b, 1);
        if (compressed) {
            auto pre = std::make_unique<PrefixedSource>();
            pre->prefix.assign(&b, &b + g);
            pre->inner = std::move(src);
            src = std::move(pre);
            first = g ? (int)b : -1;
        }
    }
    if (src->failed()) return hfail(FH_ERR_INVALID, "corrupt compressed stream");
    if (first < 0) return hfail(FH_ERR_INVALID, "empty input: not a FASTA/FASTQ file");
    if (first != '>' && first != '@') return hfail(FH_ERR_INVALID, "not a FASTA/FASTQ file (first byte 0x%02x)", first);
    const bool fastq = first == '@';
    FastxStats st;
    st.format = fastq ? 2 : 1;
    const int filter_on_eff = filters.filter_on < 0 ? (fastq ? 1 : 0) : filters.filter_on;
    const bool small = sp.kind == 0 && filter_on_eff == 0 && sp.final_size >= 1 && sp.final_size < sp.kmers_to_sketch &&
                       cfg("no_small_sketcher") == nullptr;
    finch_sketch_params sp_dev = sp;
    if (small) sp_dev.kmers_to_sketch = sp.final_size; // (see HandleSet)
    uint64_t stage = chunk_bytes ? std::max<uint64_t>(chunk_bytes, 4096) : (32ull << 20);
    const fh_params fp = to_fh(sp_dev, env_max_launch_value(), stage);
    const uint32_t K = sp.kmer_length;
    if (K < 1 || K > 64) return hfail(FH_ERR_UNSUPPORTED, "kmer_length %u", K);

    const size_t n_w = devs.size();
    // The reader fills the workers' own pinned staging buffers (two per handle, fh_text_buffers) in turn, so a chunk is
    // copied once -- source to pinned memory, by the call's read threads -- and, FASTQ, its host-to-device copy starts the
    // moment it is complete (fh_text_prefetch).  (Until round 3 the chunks went through heap buffers and each worker copied
    // its chunk again: 13-19 GB/s of text against 52 for the single-handle path on the same box.)
    struct Worker {
        fh_sketcher *h = nullptr;
        ShardQueue q;
        std::thread th;
        int rc = FH_OK;
        std::string msg;
        TextBuf tb[2];
        bool is_free[2] = {true, true};
        int fill = 0; // the slot the reader fills next: pushes consume the slots alternately
    };
    std::vector<std::unique_ptr<Worker>> W;
    struct Cleanup {
        std::vector<std::unique_ptr<Worker>> &W;
        ~Cleanup() {
            for (auto &w : W)
                if (w->h) fh_free(w->h);
        }
    } cleanup{W};
    for (size_t d = 0; d < n_w; ++d) {
        W.push_back(std::make_unique<Worker>());
        W[d]->h = fh_new(&fp, devs[d]);
        if (!W[d]->h) return hfail(FH_ERR_NO_DEVICE, "%s", fh_last_error());
        if (int rc = fh_reset(W[d]->h)) return hfail(rc, "%s", fh_last_error());
        // (the handles' staging buffers may be smaller than asked for -- the FH_STAGE_BYTES test knob: cut chunks that fit)
        uint8_t *raw2[2] = {nullptr, nullptr};
        uint64_t cap = 0;
        int next = 0;
        if (int rc = fh_text_buffers(W[d]->h, raw2, &cap, &next)) return hfail(rc, "%s", fh_last_error());
        stage = std::min(stage, cap);
        W[d]->fill = next;
        for (int i = 0; i < 2; ++i) W[d]->tb[i] = TextBuf{raw2[i], 0, (int)(2 * d) + i};
    }
    for (auto &w : W)
        for (auto &t : w->tb) t.cap = (size_t)stage;
    std::mutex free_mu;
    std::condition_variable free_cv;
    std::atomic<bool> abort{false};
    auto give_back = [&](TextBuf *b) { // (a buffer taken by the reader but not sent: free again, the same slot is next)
        std::lock_guard<std::mutex> g(free_mu);
        W[(size_t)b->id / 2]->is_free[b->id & 1] = true;
        free_cv.notify_all();
    };
    auto worker_main = [&](Worker *w) {
        for (;;) {
            ShardWork job;
            {
                std::unique_lock<std::mutex> lk(w->q.mu);
                w->q.cv.wait(lk, [&] { return !w->q.q.empty(); });
                job = w->q.q.front();
                w->q.q.erase(w->q.q.begin());
                w->q.cv.notify_all();
            }
            if (job.stop) return;
            if (w->rc == FH_OK && !abort) {
                int rc = fh_set_stream_offset(w->h, job.text_off);
                if (rc == FH_OK && fastq) rc = fh_push_fastq_text(w->h, job.len);
                if (rc == FH_OK && !fastq) {
                    if (job.halo_len) rc = fh_set_text_halo(w->h, job.halo, job.halo_len);
                    if (rc == FH_OK) rc = fh_push_fasta_text(w->h, job.len, job.start_state, 0u);
                }
                if (rc != FH_OK) {
                    w->rc = rc;
                    w->msg = fh_last_error();
                    abort = true;
                }
            }
            give_back(job.buf); // (the push is done with the host copy of its buffer when it returns)
        }
    };
    auto send_stop = [&](Worker *w) {
        ShardWork stop;
        stop.stop = true;
        std::lock_guard<std::mutex> lk(w->q.mu);
        w->q.q.push_back(stop);
        w->q.cv.notify_all();
    };
    try {
        for (auto &w : W) w->th = std::thread(worker_main, w.get());
    } catch (...) { // a worker thread could not be created: stop the ones that run, nothing joinable may unwind
        for (auto &w : W)
            if (w->th.joinable()) {
                send_stop(w.get());
                w->th.join();
            }
        return hfail(FH_ERR_STATE, "could not start the worker threads of a sharded input");
    }
    auto send =[&](size_t d, const ShardWork &job) {
        Worker *w = W[d].get();
        std::unique_lock<std::mutex> lk(w->q.mu);
        w->q.cv.wait(lk, [&] { return w->q.q.size() < 2; });
        w->q.q.push_back(job);
        w->q.cv.notify_all();
    };
    size_t next_w = 0; // the worker the next chunk goes to
    auto take_buf = [&]() {
        Worker *w = W[next_w].get();
        std::unique_lock<std::mutex> lk(free_mu);
        free_cv.wait(lk, [&] { return w->is_free[w->fill]; });
        w->is_free[w->fill] = false;
        return &w->tb[w->fill];
    };

    // ---- the reader ----
    const int rrc = shard_reader(*src, fastq, K, take_buf, give_back, [&](const ShardWork &job) {
        Worker *w = W[next_w].get();
        if (!abort) {
            (void)fh_text_prefetch(w->h, job.buf->id & 1, job.len);
            w->fill ^= 1;
            send(next_w, job);
        } else {
            give_back(job.buf); // (a worker failed: nothing more is pushed; the same slot stays next)
        }
        next_w = (next_w + 1) % n_w;
    }, abort, st);
    for (size_t d = 0; d < n_w; ++d) {
        ShardWork stop;
        stop.stop = true;
        send(d, stop);
    }
    for (auto &w : W) w->th.join();
    if (rrc != FH_OK) {
        *device_rejected = fastq && rrc == FH_ERR_INVALID;
        return rrc;
    }
    for (auto &w : W)
        if (w->rc != FH_OK) {
            *device_rejected = fastq && w->rc == FH_ERR_INVALID;
            return hfail(w->rc, "%s", w->msg.c_str());
        }
    // ---- partial sketches -> one ----
    uint64_t text_bases = 0;
    for (size_t d = 0; d < n_w; ++d) {
        uint64_t n = 0, tk = 0;
        if (int rc = fh_finish(W[d]->h, &n, &tk)) return hfail(rc, "%s", fh_last_error());
        if (fastq) {
            uint64_t tb = 0;
            if (int rc = fh_text_bases(W[d]->h, &tb)) return hfail(rc, "%s", fh_last_error());
            text_bases += tb;
        }
        if (d > 0)
            if (int rc = fh_merge(W[0]->h, W[d]->h)) return hfail(rc, "%s", fh_last_error());
    }
    if (fastq) st.total_bases = text_bases;
    return finish_sketch(W[0]->h, name, sp, filters, st, out);
}

// ---------------------------------------------------------------------------------------------
// .sk (Mash-JSON) writer: MultiSketch::from_sketches + serde_json (json.rs:64-89,141-158,199-218)
// ---------------------------------------------------------------------------------------------
static void json_escape(std::string &o, const std::string &s) {
    o.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        default:
            if (c < 0x20) {
                char b[8];
                snprintf(b, sizeof b, "\\u%04x", c);
                o += b;
            } else o.push_back((char)c);
        }
    }
    o.push_back('"');
}

// shortest round-trip digits
static void shortest_digits(double v, std::string &digits, int &exp10) {
    char b[64];
    auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::scientific);
    std::string s(b, r.ptr); // d.ddddde[+-]XX
    const size_t e = s.find('e');
    std::string mant = s.substr(0, e);
    exp10 = atoi(s.c_str() + e + 1);
    digits.clear();
    for (char c : mant)
        if (c != '.' && c != '-') digits.push_back(c);
}

// Rust `f64::to_string()` (Display): never scientific, shortest digits (filtering.rs:96-99)
static std::string rust_display_f64(double v) {
    if (v == 0.0) return "0";
    std::string d;
    int e;
    shortest_digits(std::fabs(v), d, e);
    std::string o = v < 0 ? "-" : "";
    const int nd = (int)d.size();
    if (e >= nd - 1) { // integer
        o += d;
        o.append((size_t)(e - (nd - 1)), '0');
    } else if (e >= 0) {
        o += d.substr(0, (size_t)e + 1) + "." + d.substr((size_t)e + 1);
    } else {
        o += "0.";
        o.append((size_t)(-e - 1), '0');
        o += d;
    }
    return o;
}

// serde_json f64 (ryu): fixed for 1e-5 <= |v| < 1e16 (always with a fractional part), else scientific
static std::string json_f64(double v) {
    if (v == 0.0) return "0.0";
    std::string d;
    int e;
    shortest_digits(std::fabs(v), d, e);
    std::string o = v < 0 ? "-" : "";
    const int nd = (int)d.size();
    if (e >= -5 && e < 16) {
        if (e >= nd - 1) {
            o += d;
            o.append((size_t)(e - (nd - 1)), '0');
            o += ".0";
        } else if (e >= 0) {
            o += d.substr(0, (size_t)e + 1) + "." + d.substr((size_t)e + 1);
        } else {
            o += "0.";
            o.append((size_t)(-e - 1), '0');
            o += d;
        }
    } else {
        o += d.substr(0, 1);
        if (nd > 1) o += "." + d.substr(1);
        o += "e" + std::to_string(e);
    }
    return o;
}

// FilterParams::to_serialized (filtering.rs:89-108); the reference's HashMap has no defined key order
static void json_filters(std::string &o, const finch_filter_params &fp) {
    o.push_back('{');
    if (fp.filter_on == 1) {
        bool first = true;
        auto kv = [&](const char *k, const std::string &v) {
            if (!first) o.push_back(',');
            first = false;
            o += "\"";
            o += k;
            o += "\":\"" + v + "\"";
        };
        if (fp.strand_filter > 0.0) kv("strandFilter", rust_display_f64(fp.strand_filter));
        if (fp.err_filter > 0.0) kv("errFilter", rust_display_f64(fp.err_filter));
        if (fp.has_abun_lo) kv("minCopies", std::to_string(fp.abun_lo));
        if (fp.has_abun_hi) kv("maxCopies", std::to_string(fp.abun_hi));
    }
    o.push_back('}');
}

// SketchParams::check_compatibility (mod.rs:182-212) of every sketch with the first (from_sketches, mod.rs:158-180)
int check_compatible(const std::vector<Sketch> &sketches) {
    if (sketches.empty()) return hfail(FH_ERR_INVALID, "no sketches");
    auto hash_type = [](const finch_sketch_params &p) { return p.kind == 2 ? "None" : "MurmurHash3_x64_128"; };
    auto hash_bits = [](const finch_sketch_params &p) { return p.kind == 2 ? 0u : 64u; };
    auto hash_seed = [](const finch_sketch_params &p) { return p.kind == 2 ? 0ull : (unsigned long long)p.hash_seed; };
    const finch_sketch_params &sp = sketches[0].sketch_params;
    for (size_t i = 1; i < sketches.size(); ++i) {
        const finch_sketch_params &q = sketches[i].sketch_params;
        if (q.kmer_length != sp.kmer_length)
            return hfail(FH_ERR_INVALID, "First sketch has k %u, but sketch %zu has k %u", sp.kmer_length, i + 1, q.kmer_length);
        if (strcmp(hash_type(q), hash_type(sp)) != 0)
            return hfail(FH_ERR_INVALID, "First sketch has hash type %s, but sketch %zu has hash type %s", hash_type(sp), i + 1, hash_type(q));
        if (hash_bits(q) != hash_bits(sp))
            return hfail(FH_ERR_INVALID, "First sketch has hash bits %u, but sketch %zu has hash bits %u", hash_bits(sp), i + 1, hash_bits(q));
        if (hash_seed(q) != hash_seed(sp))
            return hfail(FH_ERR_INVALID, "First sketch has hash seed %llu, but sketch %zu has hash seed %llu", hash_seed(sp), i + 1, hash_seed(q));
    }
    return FH_OK;
}

static int to_json(const std::vector<Sketch> &sketches, std::string &o) {
    if (sketches.empty()) return hfail(FH_ERR_INVALID, "no sketches to serialise");
    if (int rc = check_compatible(sketches)) return rc; // SketchParams::from_sketches (mod.rs:158-180)
    const finch_sketch_params &sp = sketches[0].sketch_params;
    // expected_size (mod.rs:148-156); AllCounts: 4^k as the reference's `as u32` leaves it
    const uint64_t expected = sp.kind == 0 ? sp.final_size : sp.kind == 1 ? sp.kmers_to_sketch
                              : (sp.kmer_length < 32 ? (1ull << (2 * sp.kmer_length)) : 0ull);
    o.clear();
    o += "{\"kmer\":" + std::to_string(sp.kmer_length);
    o += ",\"alphabet\":\"ACGT\",\"preserveCase\":false,\"canonical\":true";
    o += ",\"sketchSize\":" + std::to_string((uint32_t)expected);
    if (sp.kind == 2) o += ",\"hashType\":\"None\",\"hashBits\":0,\"hashSeed\":0"; // hash_info of AllCounts (mod.rs:138-146)
    else o += ",\"hashType\":\"MurmurHash3_x64_128\",\"hashBits\":64,\"hashSeed\":" + std::to_string(sp.hash_seed);
    o += ",\"scale\":" + (sp.kind == 1 ? json_f64(sp.scale) : std::string("null"));
    o += ",\"sketches\":[";
    for (size_t i = 0; i < sketches.size(); ++i) {
        const Sketch &s = sketches[i];
        if (i) o.push_back(',');
        o += "{\"name\":";
        json_escape(o, s.name);
        o += ",\"seqLength\":" + std::to_string(s.seq_length);
        o += ",\"numValidKmers\":" + std::to_string(s.num_valid_kmers);
        o += ",\"comment\":";
        json_escape(o, s.comment);
        o += ",\"filters\":";
        json_filters(o, s.filter_params);
        o += ",\"hashes\":[";
        for (size_t j = 0; j < s.hashes.size(); ++j) {
            if (j) o.push_back(',');
            o += "\"" + std::to_string(s.hashes[j].hash) + "\"";
        }
        o += "],\"kmers\":[";
        for (size_t j = 0; j < s.hashes.size(); ++j) {
            if (j) o.push_back(',');
            json_escape(o, s.hashes[j].kmer.str());
        }
        o += "],\"counts\":[";
        for (size_t j = 0; j < s.hashes.size(); ++j) {
            if (j) o.push_back(',');
            o += std::to_string(s.hashes[j].count);
        }
        o += "]}";
    }
    o += "]}";
    return FH_OK;
}

} // namespace finch

// =============================================================================================
// C ABI
// =============================================================================================
using namespace finch;

extern "C" {

int finch_gzip_probe(const uint8_t *data, uint64_t len, uint64_t piece_bytes, uint64_t *hdr_len, int *first_byte, uint64_t *deflate_bytes,
                     uint32_t *crc_of_pieces) try {
    if (!data || !hdr_len || !first_byte || !deflate_bytes || !crc_of_pieces) return finch::hfail(FH_ERR_INVALID, "null argument");
    finch::BgzfSource bz(std::make_unique<finch::MemSource>(data, (size_t)len), 2);
    size_t h = 0;
    *first_byte = bz.peek_plain_gzip(&h);
    *hdr_len = h;
    *deflate_bytes = 0;
    *crc_of_pieces = 0;
    if (*first_byte < 0) return FH_OK;
    std::vector<uint8_t> skip(h), piece((size_t)std::max<uint64_t>(1, piece_bytes));
    if (bz.raw_read(skip.data(), h) != h) return finch::hfail(FH_ERR_INVALID, "gzip header cut short");
    for (;;) {
        const size_t got = bz.raw_read(piece.data(), piece.size());
        *crc_of_pieces = finch::inf::crc32_fast(*crc_of_pieces, piece.data(), got);
        *deflate_bytes += got;
        if (got < piece.size()) break;
    }
    return FH_OK;
} FINCH_CATCH

void finch_debug_device_inflate(uint64_t *files_on_device, uint64_t *files_reread) {
    if (files_on_device) *files_on_device = finch::g_bgzf_on_device.load();
    if (files_reread) *files_reread = finch::g_bgzf_reread.load();
}

void finch_debug_device_gzip(uint64_t *files_on_device, uint64_t *files_reread) {
    if (files_on_device) *files_on_device = finch::g_gzip_on_device.load();
    if (files_reread) *files_reread = finch::g_gzip_reread.load();
}

void finch_debug_kernel_times(int enable, double *kernel_ms, uint64_t *launches, uint64_t *positions) {
    if (kernel_ms) *kernel_ms = (double)finch::g_ktimes_us.load() / 1000.0;
    if (launches) *launches = finch::g_ktimes_launches.load();
    if (positions) *positions = finch::g_ktimes_positions.load();
    if (enable >= 0) {
        finch::g_ktimes_on.store(enable ? 1 : 0);
        if (enable) finch::g_ktimes_us = 0, finch::g_ktimes_launches = 0, finch::g_ktimes_positions = 0;
    }
}

uint64_t finch_debug_fastq_host_strip(void) { return finch::g_fastq_host_strip.load(); }

// test hook: text[0, len) (whole records of plain 4-line FASTQ) through the host-side strip (fh_fqstrip.h) on `threads` threads
int finch_fastq_strip_probe(const uint8_t *text, uint64_t len, uint32_t threads, uint8_t *out, uint64_t cap, uint64_t *packed,
                            uint64_t *n_records, uint64_t *total_bases) try {
    if ((!text && len) || !out || !packed || !n_records || !total_bases || threads < 1 || threads > 64) return hfail(FH_ERR_INVALID, "bad argument");
    if (cap < len / 2 + 64) return hfail(FH_ERR_INVALID, "output needs len / 2 + 64 bytes");
    std::vector<fqstrip::Piece> pieces(threads);
    fqstrip::Barrier bar(threads);
    bool ok = false;
    uint64_t m = 0, nr = 0, nb = 0;
    fork_join(threads, [&](unsigned t) { fqstrip::strip_chunk(t, threads, text, (size_t)len, out, pieces, bar, &ok, &m, &nr, &nb); });
    if (!ok) return hfail(FH_ERR_INVALID, "not plain 4-line FASTQ");
    *packed = m;
    *n_records = nr;
    *total_bases = nb;
    return FH_OK;
} FINCH_CATCH

// test hook: FASTA text through the workers' piecewise walk into the batch sketcher's two-bit form (FastaTwoBit), `piece` bytes at a time
int finch_fasta_two_bit_probe(const uint8_t *text, uint64_t len, uint64_t piece, uint8_t *region, uint64_t cap, uint64_t *positions,
                              uint64_t *n_records, uint64_t *total_bases) try {
    if (!text || len < 1 || text[0] != '>' || piece < 1 || !region || !positions || !n_records || !total_bases) return hfail(FH_ERR_INVALID, "bad argument");
    if (cap < fh_pack2::region_bytes(len)) return hfail(FH_ERR_INVALID, "the region needs fh_batch_packed_bytes(len) bytes");
    std::unique_ptr<fh_pack2::Packer> pk(new fh_pack2::Packer);
    pk->force_form((unsigned)cfg_u64("pack_scalar", 0));
    finch::FastxStats st;
    finch::FastaTwoBit walk(*pk, st, region);
    for (uint64_t o = 0; o < len; o += piece) walk.piece(text + o, (size_t)std::min<uint64_t>(piece, len - o));
    *positions = walk.finish();
    *n_records = st.n_records;
    *total_bases = st.total_bases;
    return FH_OK;
} FINCH_CATCH

void finch_debug_file_batch(uint64_t *taken, uint64_t *not_taken) {
    if (taken) *taken = finch::g_batch_taken.load();
    if (not_taken) *not_taken = finch::g_batch_not_taken.load();
}

const char *finch_last_error(void) { return g_host_err.c_str(); }

void finch_default_sketch_params(finch_sketch_params *out) {
    if (!out) return;
    memset(out, 0, sizeof *out);
    out->kind = 0;
    out->kmers_to_sketch = 1000;
    out->final_size = 1000;
    out->no_strict = 0;
    out->kmer_length = 21;
    out->hash_seed = 0;
    out->scale = 0.001;
}

void finch_default_filter_params(finch_filter_params *out) {
    if (!out) return;
    memset(out, 0, sizeof *out);
    out->filter_on = 0; // Some(false)
}

static uint64_t env_max_launch() {
    const char *e = cfg("max_launch");
    return e ? strtoull(e, nullptr, 10) : 0;
}

int finch_sketch_buffer(const uint8_t *data, uint64_t len, const char *name, const finch_sketch_params *sp,
                        const finch_filter_params *filters, int device, finch_sketches **out) try {
    if ((!data && len) || !sp || !filters || !out) return hfail(FH_ERR_INVALID, "null argument");
    HandleSet handles;
    handles.full = to_fh(*sp, env_max_launch());
    handles.final_size = sp->final_size;
    handles.device = device;
    auto res = std::make_unique<finch_sketches>();
    res->v.resize(1);
    const int rc = sketch_stream(std::make_unique<MemSource>(data, (size_t)len, read_threads_total(cfg("read_threads"))),
                                 name ? name : "", *sp, *filters, handles, res->v[0]);
    if (rc != FH_OK) return rc;
    *out = res.release();
    return FH_OK;
} FINCH_CATCH

int finch_sketch_files(const char *const *filenames, uint32_t n_files, const finch_sketch_params *sp,
                       const finch_filter_params *filters, const int *devices, uint32_t n_devices, uint32_t n_threads,
                       finch_sketches **out) try {
    if (!filenames || !sp || !filters || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::vector<int> devs;
    if (devices && n_devices) devs.assign(devices, devices + n_devices);
    else devs.push_back(0);
    // host workers per GPU: a worker's time is reading its files and packing them (page cache -> pinned memory, line ends out, the
    // two-bit form), the device and the link have room to spare (docs/MEASUREMENTS_r06.md 1b); 16 is what the boxes so far grant,
    // and more workers than cores granted run slower than fewer (24 on a 16-core grant: 5500 files/s, profiles/r04_c5_threads.txt)
    if (n_threads == 0) n_threads = std::min<uint32_t>(16u, std::max<uint32_t>(4u, usable_cpus() / (uint32_t)devs.size())) * (uint32_t)devs.size();
    n_threads = std::max<uint32_t>(1, std::min<uint32_t>(n_threads, std::max<uint32_t>(n_files, 1)));
    static const bool trace_call = cfg("trace") != nullptr;
    const auto call_t0 = std::chrono::steady_clock::now();
    auto res = std::make_unique<finch_sketches>();
    res->v.resize(n_files);
    std::atomic<uint32_t> next{0};
    std::mutex err_mu;
    int first_err_code = FH_OK;
    uint32_t first_err_idx = UINT32_MAX;
    std::string first_err_msg;
    // lib.rs:34-36: par_iter over the files; here one worker = one device sketcher reused across its files
    // many files: one modest sketcher per worker (2 M positions in flight, 16 MiB staging) instead of one
    // sized for a 10 Gbase stream
    const bool batch = n_files > 1;
    const uint64_t ml = env_max_launch() ? env_max_launch() : (batch ? (2ull << 20) : 0ull);
    // a single small file: size the one sketcher for it (pinning 2 x 64 MiB of staging memory alone takes ~40 ms, the
    // whole sketch of a 5 Mb genome well under 10)
    uint64_t single_stage = 0, single_ml = ml;
    if (!batch && n_files == 1 && strcmp(filenames[0], "-") != 0) {
        struct stat sb;
        if (stat(filenames[0], &sb) == 0 && S_ISREG(sb.st_mode) && (uint64_t)sb.st_size < (48ull << 20)) {
            single_stage = std::max<uint64_t>(((uint64_t)sb.st_size + (1ull << 20)) & ~((1ull << 20) - 1), 1ull << 20);
            if (!single_ml) single_ml = 2ull << 20;
        }
    }
    // threads for the large reads of plain files, shared among the workers (a batch of genomes reads with one each)
    const char *rt_env = cfg("read_threads");
    const unsigned read_total = read_threads_total(rt_env);
    const unsigned read_threads = std::max(1u, read_total / n_threads);
    // Many files per launch (fh_batch_*, fh_k2b.hip): a worker stages the packed streams of plain FASTA files side by side in
    // a slot of its batch handle and has them sketched by ONE launch, finished by ONE epilogue launch (a workgroup per file),
    // behind one copy and in front of one synchronisation -- while it stages the next group in the other slot.  Applies to
    // what a batch of genomes is: Mash sketches of <= 3000 hashes (after the cut to final_size the small sketcher makes),
    // k <= 32, no filtering (the default for FASTA, lib.rs:70-76), regular uncompressed files that begin with '>'.
    // Anything else, and every file the batch path reports as not taken, goes through sketch_stream as before.
    const uint64_t group_n = (sp->final_size >= 1 && sp->final_size < sp->kmers_to_sketch) ? sp->final_size : sp->kmers_to_sketch;
    const bool group_ok = batch && sp->kind == 0 && sp->kmer_length >= 1 && sp->kmer_length <= 32 && group_n >= 1 && group_n <= 3000 &&
                          filters->filter_on <= 0 && file_batch_enabled();
    constexpr uint64_t GROUP_STAGE = 32ull << 20;
    constexpr uint32_t GROUP_FILES = 64;
    const size_t READ_PIECE = std::max<uint64_t>(4096, cfg_u64("batch_read_piece", 256u << 10)); // bytes of a file read and packed at a time
    auto worker = [&](uint32_t w) {
        HandleSet handles;
        handles.full = to_fh(*sp, batch ? ml : single_ml, batch ? (16ull << 20) : single_stage);
        handles.final_size = sp->final_size;
        handles.device = devs[w % devs.size()];
        auto record_error = [&](uint32_t i, int rc, const std::string &msg) {
            std::lock_guard<std::mutex> g(err_mu);
            if (i < first_err_idx) {
                first_err_idx = i;
                first_err_code = rc;
                first_err_msg = msg;
            }
        };
        auto through_sketcher = [&](uint32_t i) { // one file through its own sketcher (sketch_stream, lib.rs:51-94)
            int rc = FH_OK;
            std::string msg;
            const std::string fn = filenames[i];
            FILE *f = fn == "-" ? stdin : fopen(fn.c_str(), "rb");
            if (!f) {
                rc = FH_ERR_INVALID;
                msg = fn + ": " + strerror(errno) + " (os error " + std::to_string(errno) + ")";
            } else {
                rc = sketch_stream(std::make_unique<FileSource>(f, f != stdin, read_threads), fn, *sp, *filters, handles, res->v[i]);
                if (rc != FH_OK) msg = g_host_err;
            }
            if (rc != FH_OK) record_error(i, rc, msg);
        };
        // --- the worker's groups ---
        struct Group {
            std::vector<uint32_t> idx;
            std::vector<uint64_t> off, len;
            std::vector<FastxStats> st;
            uint64_t fill = 0;
            bool in_flight = false;
        } grp[2];
        fh_batch *bt = nullptr;
        bool bt_failed = false;
        int cur = 0;
        uint8_t *stage[2] = {nullptr, nullptr};
        uint64_t stage_cap = 0;
        std::vector<uint8_t> raw; // a file's text as read
        const bool two_bit = !(cfg("batch_two_bit") && cfg("batch_two_bit")[0] == '0'); // how a group's files cross the link
        std::unique_ptr<fh_pack2::Packer> packer;
        std::vector<uint8_t> status;
        static const bool trace = cfg("trace") != nullptr;
        auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_read = 0, t_pack = 0, t_submit = 0, t_wait = 0, t_collect = 0;
        const double t_worker0 = now_s();
        auto collect = [&](int slot) { // wait for the group in `slot`, turn its results into Sketches
            Group &g = grp[slot];
            if (!g.in_flight) return;
            g.in_flight = false;
            status.assign(g.idx.size() + 1, 1);
            const double w0 = trace ? now_s() : 0;
            int rc = fh_batch_wait(bt, slot, status.data());
            const double w1 = trace ? now_s() : 0;
            t_wait += w1 - w0;
            for (size_t j = 0; j < g.idx.size(); ++j) {
                const uint32_t i = g.idx[j];
                if (rc != FH_OK || status[j] != 0) { // not taken (or the batch failed as a whole): the long way, which is exact for anything
                    through_sketcher(i);
                    continue;
                }
                uint64_t n = 0, tk = 0;
                int r2 = fh_batch_result(bt, slot, (uint32_t)j, &n, &tk);
                const size_t keep = (size_t)std::min<uint64_t>(n, sp->final_size); // process_post_filter (mod.rs:115-128)
                const uint32_t k = sp->kmer_length;
                std::unique_ptr<fh_kmer_count[]> recs(new fh_kmer_count[n + 1]);
                std::unique_ptr<uint8_t[]> km(new uint8_t[n * (size_t)k + 1]);
                if (r2 == FH_OK) r2 = fh_batch_copy_out_records(bt, slot, (uint32_t)j, recs.get(), km.get());
                if (r2 != FH_OK) {
                    record_error(i, r2, fh_last_error());
                    continue;
                }
                const std::string name = filenames[i];
                if (!sp->no_strict && keep < sp->final_size) {
                    char buf[512];
                    snprintf(buf, sizeof buf, "%s had too few kmers (%zu) to sketch", name.c_str(), keep);
                    record_error(i, FH_ERR_INVALID, buf);
                    continue;
                }
                Sketch &out = res->v[i];
                out.name = name;
                out.seq_length = g.st[j].total_bases;
                out.num_valid_kmers = tk;
                out.comment = "";
                out.hashes.resize(keep);
                for (size_t q = 0; q < keep; ++q)
                    out.hashes[q] = KmerCount{recs[q].hash, KmerBytes((const char *)km.get() + q * (size_t)k, (size_t)k), recs[q].count, recs[q].extra_count};
                out.filter_params = *filters;
                out.filter_params.filter_on = 0; // lib.rs:70-76: FASTA defaults to no filtering (group_ok: not asked for either)
                out.sketch_params = *sp;
            }
            g.idx.clear();
            g.off.clear();
            g.len.clear();
            g.st.clear();
            g.fill = 0;
            if (trace) t_collect += now_s() - w1;
        };
        auto submit_cur = [&]() { // hand the current group to the device, go on in the other slot (its previous group collected)
            Group &g = grp[cur];
            if (g.idx.empty()) return;
            const double s0 = trace ? now_s() : 0;
            const int src = two_bit ? fh_batch_submit_packed(bt, cur, g.off.data(), g.len.data(), (uint32_t)g.idx.size())
                                    : fh_batch_submit(bt, cur, g.off.data(), g.len.data(), (uint32_t)g.idx.size());
            if (trace) t_submit += now_s() - s0;
            if (src != FH_OK) {
                for (uint32_t i : g.idx) through_sketcher(i);
                g.idx.clear(), g.off.clear(), g.len.clear(), g.st.clear();
                g.fill = 0;
                return;
            }
            g.in_flight = true;
            cur ^= 1;
            collect(cur);
        };
        auto try_stage = [&](uint32_t i) -> bool { // true: file i is part of the current group
            if (!group_ok || bt_failed) return false;
            const char *fn = filenames[i];
            if (strcmp(fn, "-") == 0) return false;
            const int fd = open(fn, O_RDONLY | O_CLOEXEC);
            if (fd < 0) return false; // (sketch_stream reports it)
            struct stat sb;
            if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 1 || (uint64_t)sb.st_size + 4096 > GROUP_STAGE) {
                close(fd);
                return false;
            }
            const size_t size = (size_t)sb.st_size;
            if (!bt) {
                fh_params bp = to_fh(*sp, 0);
                bp.size = group_n;
                bt = fh_batch_new(&bp, handles.device, GROUP_FILES, GROUP_STAGE);
                if (bt && (fh_batch_stage(bt, 0, &stage[0], &stage_cap) != FH_OK || fh_batch_stage(bt, 1, &stage[1], &stage_cap) != FH_OK)) {
                    fh_batch_free(bt);
                    bt = nullptr;
                }
                if (!bt) {
                    bt_failed = true; // (no memory for it, say: every file the long way)
                    close(fd);
                    return false;
                }
                if (g_ktimes_on.load(std::memory_order_relaxed)) fh_batch_set_profiling(bt, 1);
            }
            const uint64_t room_needed = two_bit ? fh_pack2::region_bytes(size) + 64 : (uint64_t)size + 64;
            if (grp[cur].idx.size() >= GROUP_FILES || grp[cur].fill + room_needed > stage_cap) submit_cur();
            Group &g = grp[cur];
            FastxStats st;
            st.format = 1;
            size_t m = 0;
            uint64_t occupied;
            const double r0 = trace ? now_s() : 0;
            double r1 = r0;
            if (two_bit) {
                // piece by piece: read, strip, pack while the piece is in the core's cache
                if (!packer) {
                    packer.reset(new fh_pack2::Packer);
                    packer->force_form((unsigned)cfg_u64("pack_scalar", 0));
                }
                if (raw.size() < READ_PIECE + 64) raw.resize(READ_PIECE + 64);
                FastaTwoBit walk(*packer, st, stage[cur] + g.fill);
                size_t got = 0;
                bool bad = false;
                while (got < size) {
                    const double p0 = trace ? now_s() : 0;
                    const ssize_t r = pread(fd, raw.data(), std::min<size_t>(READ_PIECE, size - got), (off_t)got);
                    if (r <= 0 || (got == 0 && raw[0] != '>')) {
                        bad = true;
                        break;
                    }
                    const double p1 = trace ? now_s() : 0;
                    walk.piece(raw.data(), (size_t)r);
                    got += (size_t)r;
                    if (trace) t_read += p1 - p0, t_pack += now_s() - p1;
                }
                // one byte more than fstat said = the file grew: not for this path (its region is abandoned: g.fill stays)
                uint8_t extra;
                const bool grew = !bad && pread(fd, &extra, 1, (off_t)size) > 0;
                close(fd);
                if (bad || grew) return false;
                m = (size_t)walk.finish();
                occupied = fh_pack2::region_bytes(m);
                r1 = trace ? now_s() : 0;
            } else {
                if (raw.size() < size + 64) raw.resize(size + 64 + (size >> 2));
                size_t got = 0;
                while (got < size) {
                    const ssize_t r = pread(fd, raw.data() + got, size - got, (off_t)got);
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                uint8_t extra;
                const bool grew = got == size && pread(fd, &extra, 1, (off_t)size) > 0;
                close(fd);
                if (got != size || grew || raw[0] != '>') return false;
                r1 = trace ? now_s() : 0;
                t_read += r1 - r0;
                if (!pack_fasta_text(raw.data(), size, stage[cur] + g.fill, (size_t)(stage_cap - g.fill), st, m)) return false;
                occupied = m;
            }
            g.idx.push_back(i);
            g.off.push_back(g.fill);
            g.len.push_back(m);
            g.st.push_back(st);
            g.fill = (g.fill + occupied + 63) & ~63ull;
            if (trace) t_pack += now_s() - r1;
            return true;
        };
        for (;;) {
            const uint32_t i = next.fetch_add(1);
            if (i >= n_files) break;
            if (!try_stage(i)) through_sketcher(i);
        }
        if (bt) {
            submit_cur();
            collect(0);
            collect(1);
            if (g_ktimes_on.load(std::memory_order_relaxed)) {
                double ms = 0;
                uint64_t nl = 0, np = 0;
                if (fh_batch_kernel_time(bt, &ms, &nl, &np) == FH_OK) {
                    g_ktimes_us += (uint64_t)(ms * 1000.0 + 0.5);
                    g_ktimes_launches += nl;
                    g_ktimes_positions += np;
                }
            }
            uint64_t tk = 0, nt = 0;
            if (fh_batch_counters(bt, &tk, &nt) == FH_OK) g_batch_taken += tk, g_batch_not_taken += nt;
            if (trace && w == 0)
                fprintf(stderr, "[finch] worker 0 of a batch: %.1f ms in all: reading %.1f, packing %.1f, submits %.1f, waiting for the device %.1f, results %.1f ms (%llu files taken)\n",
                        (now_s() - t_worker0) * 1e3, t_read * 1e3, t_pack * 1e3, t_submit * 1e3, t_wait * 1e3, t_collect * 1e3, (unsigned long long)tk);
            fh_batch_free(bt);
        }
    };
    {
        std::vector<std::thread> th;
        th.reserve(n_threads);
        // (no exception leaves a worker's thread -- that would be std::terminate in the caller's process: a worker that runs out
        // of memory reports it as the error of the call)
        std::atomic<bool> worker_threw{false};
        auto guarded = [&](uint32_t w) {
            try {
                worker(w);
            } catch (...) {
                worker_threw = true;
            }
        };
        try {
            for (uint32_t w = 0; w < n_threads; ++w) th.emplace_back(guarded, w);
        } catch (...) { // fewer workers than asked for: the files are pulled from one queue, those that started take them all
        }
        if (th.empty()) guarded(0u);
        const auto spawned = std::chrono::steady_clock::now();
        for (auto &t : th) t.join();
        if (trace_call && n_files > 1)
            fprintf(stderr, "[finch] sketch_files: %u files on %u workers: %.1f ms to the last worker's start, %.1f ms until all had ended\n", n_files, n_threads,
                    std::chrono::duration<double, std::milli>(spawned - call_t0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - call_t0).count());
        if (worker_threw) return hfail(FH_ERR_CAPACITY, "out of host memory");
    }
    if (first_err_code != FH_OK) return hfail(first_err_code, "%s", first_err_msg.c_str());
    *out = res.release();
    return FH_OK;
} FINCH_CATCH

static int sharded_devices(const int *devices, uint32_t n_devices, std::vector<int> &devs) {
    if (devices && n_devices) devs.assign(devices, devices + n_devices);
    else devs.push_back(0);
    if (devs.size() > 64) return hfail(FH_ERR_INVALID, "more than 64 device handles");
    return FH_OK;
}

int finch_sketch_file_sharded(const char *filename, const finch_sketch_params *sp, const finch_filter_params *filters,
                              const int *devices, uint32_t n_devices, uint64_t chunk_bytes, finch_sketches **out) try {
    if (!filename || !sp || !filters || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::vector<int> devs;
    if (int rc = sharded_devices(devices, n_devices, devs)) return rc;
    const std::string fn = filename;
    FILE *f = fn == "-" ? stdin : fopen(fn.c_str(), "rb");
    if (!f) return hfail(FH_ERR_INVALID, "%s: %s (os error %d)", fn.c_str(), strerror(errno), errno);
    const char *rt_env = cfg("read_threads");
    const unsigned read_threads = read_threads_total(rt_env);
    auto res = std::make_unique<finch_sketches>();
    res->v.resize(1);
    bool rejected = false;
    int rc = sketch_stream_sharded(std::make_unique<FileSource>(f, f != stdin, read_threads), fn, *sp, *filters, devs, chunk_bytes,
                                   res->v[0], &rejected);
    if (rc != FH_OK && rejected && f != stdin) {
        // FASTQ the device-side splitter does not take (blank lines between records, ...): one handle, host parser
        FILE *f2 = fopen(fn.c_str(), "rb");
        if (!f2) return hfail(FH_ERR_INVALID, "%s: %s (os error %d)", fn.c_str(), strerror(errno), errno);
        HandleSet handles;
        handles.full = to_fh(*sp, env_max_launch());
        handles.final_size = sp->final_size;
        handles.device = devs[0];
        res->v[0] = Sketch();
        rc = sketch_stream(std::make_unique<FileSource>(f2, true, read_threads), fn, *sp, *filters, handles, res->v[0]);
    }
    if (rc != FH_OK) return rc;
    *out = res.release();
    return FH_OK;
} FINCH_CATCH

int finch_sketch_buffer_sharded(const uint8_t *data, uint64_t len, const char *name, const finch_sketch_params *sp,
                                const finch_filter_params *filters, const int *devices, uint32_t n_devices, uint64_t chunk_bytes,
                                finch_sketches **out) try {
    if ((!data && len) || !sp || !filters || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::vector<int> devs;
    if (int rc = sharded_devices(devices, n_devices, devs)) return rc;
    auto res = std::make_unique<finch_sketches>();
    res->v.resize(1);
    bool rejected = false;
    const unsigned mem_threads = read_threads_total(cfg("read_threads"));
    int rc = sketch_stream_sharded(std::make_unique<MemSource>(data, (size_t)len, mem_threads), name ? name : "", *sp, *filters, devs, chunk_bytes,
                                   res->v[0], &rejected);
    if (rc != FH_OK && rejected) { // (see finch_sketch_file_sharded)
        HandleSet handles;
        handles.full = to_fh(*sp, env_max_launch());
        handles.final_size = sp->final_size;
        handles.device = devs[0];
        res->v[0] = Sketch();
        rc = sketch_stream(std::make_unique<MemSource>(data, (size_t)len, mem_threads), name ? name : "", *sp, *filters, handles, res->v[0]);
    }
    if (rc != FH_OK) return rc;
    *out = res.release();
    return FH_OK;
} FINCH_CATCH

// Test hook (no device): the chunks the sharded reader would deal out for an input image -- per chunk 4 words
// (text offset, length, FASTA start state, halo length) in `meta` and 64 halo bytes in `halos`.
int finch_shard_probe(const uint8_t *data, uint64_t len, uint32_t k, uint64_t chunk_bytes, uint64_t max_chunks, uint64_t *meta,
                      uint8_t *halos, uint64_t *n_chunks, uint64_t *n_records, uint64_t *total_bases) try {
    if ((!data && len) || !n_chunks || k < 1 || k > 64 || chunk_bytes < 16) return hfail(FH_ERR_INVALID, "bad argument");
    std::unique_ptr<ByteSource> src;
    int first = -1;
    if (int rc = open_source(std::make_unique<MemSource>(data, (size_t)len), src, nullptr, &first)) return rc;
    if (first != '>' && first != '@') return hfail(FH_ERR_INVALID, "not a FASTA/FASTQ file");
    std::vector<uint8_t> mem((size_t)chunk_bytes);
    TextBuf buf{mem.data(), mem.size(), 0};
    std::atomic<bool> abort{false};
    FastxStats st;
    uint64_t n = 0;
    const int rc = shard_reader(*src, first == '@', k, [&] { return &buf; }, [](TextBuf *) {},
                                [&](const ShardWork &job) {
                                    if (n < max_chunks) {
                                        if (meta) {
                                            meta[4 * n] = job.text_off;
                                            meta[4 * n + 1] = job.len;
                                            meta[4 * n + 2] = job.start_state;
                                            meta[4 * n + 3] = job.halo_len;
                                        }
                                        if (halos) memcpy(halos + 64 * n, job.halo, job.halo_len);
                                    }
                                    ++n;
                                }, abort, st);
    if (rc != FH_OK) return rc;
    *n_chunks = n;
    if (n_records) *n_records = st.n_records;
    if (total_bases) *total_bases = st.total_bases;
    return FH_OK;
} FINCH_CATCH

// The tail of sketch_stream (lib.rs:70-93) for a caller that drove the device engine itself (fh_push_* on `h`): to_vec ->
// filter_counts -> process_post_filter -> Sketch.  `format`: 1 FASTA, 2 FASTQ (decides the filtering default when
// filters->filter_on is None, lib.rs:70-76).
int finch_sketch_from_sketcher(fh_sketcher *h, const char *name, uint64_t seq_length, int format, const finch_sketch_params *sp,
                               const finch_filter_params *filters, finch_sketches **out) try {
    if (!h || !sp || !filters || !out) return hfail(FH_ERR_INVALID, "null argument");
    FastxStats st;
    st.total_bases = seq_length;
    st.format = format;
    auto res = std::make_unique<finch_sketches>();
    res->v.resize(1);
    if (int rc = finish_sketch(h, name ? name : "", *sp, *filters, st, res->v[0])) return rc;
    *out = res.release();
    return FH_OK;
} FINCH_CATCH

void finch_sketches_free(finch_sketches *s) { delete s; }
uint32_t finch_sketches_len(const finch_sketches *s) { return s ? (uint32_t)s->v.size() : 0; }
const char *finch_sketch_name(const finch_sketches *s, uint32_t i) { return (s && i < s->v.size()) ? s->v[i].name.c_str() : ""; }
uint64_t finch_sketch_seq_length(const finch_sketches *s, uint32_t i) { return (s && i < s->v.size()) ? s->v[i].seq_length : 0; }
uint64_t finch_sketch_num_valid_kmers(const finch_sketches *s, uint32_t i) {
    return (s && i < s->v.size()) ? s->v[i].num_valid_kmers : 0;
}
uint64_t finch_sketch_n_hashes(const finch_sketches *s, uint32_t i) { return (s && i < s->v.size()) ? s->v[i].hashes.size() : 0; }

int finch_sketch_filter_params(const finch_sketches *s, uint32_t i, finch_filter_params *out) try {
    if (!s || i >= s->v.size() || !out) return hfail(FH_ERR_INVALID, "bad argument");
    *out = s->v[i].filter_params;
    return FH_OK;
} FINCH_CATCH

int finch_sketch_copy(const finch_sketches *s, uint32_t i, uint64_t *hashes, uint32_t *counts, uint32_t *extra_counts,
                      uint8_t *kmers) try {
    if (!s || i >= s->v.size()) return hfail(FH_ERR_INVALID, "bad argument");
    const Sketch &sk = s->v[i];
    const size_t k = sk.sketch_params.kmer_length;
    for (size_t j = 0; j < sk.hashes.size(); ++j) {
        if (hashes) hashes[j] = sk.hashes[j].hash;
        if (counts) counts[j] = sk.hashes[j].count;
        if (extra_counts) extra_counts[j] = sk.hashes[j].extra_count;
        if (kmers) memcpy(kmers + j * k, sk.hashes[j].kmer.data(), std::min(k, sk.hashes[j].kmer.size()));
    }
    return FH_OK;
} FINCH_CATCH

int finch_sketches_to_json(const finch_sketches *s, char **out, uint64_t *len) try {
    if (!s || !out) return hfail(FH_ERR_INVALID, "null argument");
    std::string o;
    if (int rc = to_json(s->v, o)) return rc;
    char *p = (char *)malloc(o.size() + 1);
    if (!p) return hfail(FH_ERR_INVALID, "out of memory");
    memcpy(p, o.data(), o.size());
    p[o.size()] = 0;
    *out = p;
    if (len) *len = o.size();
    return FH_OK;
} FINCH_CATCH

void finch_free_string(char *p) { free(p); }

int finch_sketches_from_arrays(const char *name, uint64_t seq_length, uint64_t num_valid_kmers, uint64_t n,
                               const uint64_t *hashes, const uint32_t *counts, const uint32_t *extra_counts,
                               const uint8_t *kmers, const finch_sketch_params *sp, const finch_filter_params *filters,
                               finch_sketches **out) try {
    if (!sp || !filters || !out || (n && (!hashes || !counts || !extra_counts))) return hfail(FH_ERR_INVALID, "null argument");
    // KmerCount invariants of the reference's sketchers (mash.rs:45-56: count starts at 1, extra_count is bumped with it):
    // the filters index a histogram by count - 1 and subtract extra_count from count
    for (uint64_t i = 0; i < n; ++i) {
        if (counts[i] == 0) return hfail(FH_ERR_INVALID, "record %llu has count 0", (unsigned long long)i);
        if (extra_counts[i] > counts[i])
            return hfail(FH_ERR_INVALID, "record %llu has extra_count %u > count %u", (unsigned long long)i, extra_counts[i], counts[i]);
    }
    auto res = std::make_unique<finch_sketches>();
    res->v.resize(1);
    Sketch &s = res->v[0];
    s.name = name ? name : "";
    s.seq_length = seq_length;
    s.num_valid_kmers = num_valid_kmers;
    s.sketch_params = *sp;
    s.filter_params = *filters;
    const size_t k = sp->kmer_length;
    s.hashes.resize(n);
    for (uint64_t i = 0; i < n; ++i)
        s.hashes[i] = KmerCount{hashes[i], kmers ? KmerBytes((const char *)kmers + i * k, (size_t)k) : KmerBytes(), counts[i],
                                extra_counts[i]};
    *out = res.release();
    return FH_OK;
} FINCH_CATCH

int finch_apply_filters(finch_sketches *s, uint32_t i, finch_filter_params *filters) try {
    if (!s || i >= s->v.size() || !filters) return hfail(FH_ERR_INVALID, "bad argument");
    Sketch &sk = s->v[i];
    std::vector<KmerCount> f = filter_counts(*filters, sk.hashes);
    if (int rc = process_post_filter(sk.sketch_params, f, sk.name)) return rc;
    sk.hashes.swap(f);
    sk.filter_params = *filters;
    return FH_OK;
} FINCH_CATCH

int finch_raw_distance(const uint64_t *query, uint64_t nq, const uint64_t *ref, uint64_t nr, double scale,
                       finch_distance_out *out) try {
    if (!out || (nq && !query) || (nr && !ref)) return hfail(FH_ERR_INVALID, "null argument");
    raw_distance(query, nq, ref, nr, scale, out->containment, out->jaccard, out->common_hashes, out->total_hashes);
    out->mash_distance = 0.;
    return FH_OK;
} FINCH_CATCH

int finch_distance(const finch_sketches *a, uint32_t ia, const finch_sketches *b, uint32_t ib, int old_mode,
                   finch_distance_out *out) try {
    if (!a || !b || !out || ia >= a->v.size() || ib >= b->v.size()) return hfail(FH_ERR_INVALID, "bad argument");
    const Sketch &qs = a->v[ia], &rs = b->v[ib];
    std::vector<uint64_t> q(qs.hashes.size()), r(rs.hashes.size());
    for (size_t i = 0; i < q.size(); ++i) q[i] = qs.hashes[i].hash;
    for (size_t i = 0; i < r.size(); ++i) r[i] = rs.hashes[i].hash;
    if (old_mode) {
        if (int rc = old_distance(q.data(), q.size(), r.data(), r.size(), out->containment, out->jaccard, out->common_hashes,
                                  out->total_hashes))
            return rc;
    } else {
        // distance.rs:16-29: a scale only if both sketches are scaled
        double min_scale = 0.;
        if (qs.sketch_params.kind == 1 && rs.sketch_params.kind == 1) min_scale = std::min(qs.sketch_params.scale, rs.sketch_params.scale);
        raw_distance(q.data(), q.size(), r.data(), r.size(), min_scale, out->containment, out->jaccard, out->common_hashes,
                     out->total_hashes);
    }
    const double k = (double)qs.sketch_params.kmer_length;
    const double md = -1.0 * std::log((2.0 * out->jaccard) / (1.0 + out->jaccard)) / k; // distance.rs:37
    out->mash_distance = std::min(1.0, std::max(0.0, md));
    return FH_OK;
} FINCH_CATCH

// statistics.rs:8-23: the k-minimum-values estimate, in the reference's f32 arithmetic (`as u64` saturates, NaN -> 0)
int finch_sketch_cardinality(const finch_sketches *s, uint32_t i, uint64_t *out) try {
    if (!s || i >= s->v.size() || !out) return hfail(FH_ERR_INVALID, "bad argument");
    const std::vector<KmerCount> &h = s->v[i].hashes;
    if (h.empty()) {
        *out = 0;
        return FH_OK;
    }
    const float ratio = (float)h.back().hash / 18446744073709551616.0f; // usize::MAX as f32 rounds to 2^64
    const float est = (float)(h.size() - 1) / ratio;
    *out = est != est ? 0ull : (est >= 18446744073709551616.0f ? UINT64_MAX : (est <= 0.0f ? 0ull : (uint64_t)est));
    return FH_OK;
} FINCH_CATCH

// statistics.rs:30-47 (hist): out[c - 1] = number of hashes with count c, for c = 1..max count; *n = max count.  Call with
// out = NULL to learn *n.
int finch_sketch_hist(const finch_sketches *s, uint32_t i, uint64_t *out, uint64_t cap, uint64_t *n) try {
    if (!s || i >= s->v.size() || !n) return hfail(FH_ERR_INVALID, "bad argument");
    const std::vector<uint64_t> hd = hist(s->v[i].hashes);
    *n = hd.size();
    if (out) {
        if (cap < hd.size()) return hfail(FH_ERR_INVALID, "histogram of %zu entries does not fit %llu", hd.size(), (unsigned long long)cap);
        memcpy(out, hd.data(), hd.size() * sizeof(uint64_t));
    }
    return FH_OK;
} FINCH_CATCH

uint32_t finch_guess_filter_threshold(const uint32_t *counts, uint64_t n, double filter_level) {
    if (n && !counts) {
        hfail(FH_ERR_INVALID, "null argument");
        return 0; // (a threshold is always >= 1)
    }
    for (uint64_t i = 0; i < n; ++i)
        if (counts[i] == 0) {
            hfail(FH_ERR_INVALID, "count %llu is 0", (unsigned long long)i);
            return 0;
        }
    std::vector<KmerCount> v(n);
    for (uint64_t i = 0; i < n; ++i) v[i] = KmerCount{i, KmerBytes(), counts[i], 0};
    return guess_filter_threshold(v, filter_level);
}

int finch_fastx_scan(const uint8_t *data, uint64_t len, uint64_t *n_records, uint64_t *total_bases, int *format) try {
    if (!data && len) return hfail(FH_ERR_INVALID, "null argument");
    std::unique_ptr<ByteSource> src;
    if (int rc = open_source(std::make_unique<MemSource>(data, (size_t)len), src)) return rc;
    CountSink sink;
    FastxStats st;
    if (int rc = parse_fastx(*src, sink, st)) return rc;
    if (n_records) *n_records = st.n_records;
    if (total_bases) *total_bases = st.total_bases;
    if (format) *format = st.format;
    return FH_OK;
} FINCH_CATCH

// Reads a file the way the text paths do (FileSource::read in `chunk`-byte requests, 2 sniffed bytes first, large
// requests split over `read_threads` threads): lets the host-only tests compare the bytes with the file's.
int finch_read_file_probe(const char *path, uint64_t chunk, uint32_t read_threads, uint8_t *dst, uint64_t cap, uint64_t *got) try {
    if (!path || !dst || !got || chunk == 0) return hfail(FH_ERR_INVALID, "bad argument");
    FILE *f = fopen(path, "rb");
    if (!f) return hfail(FH_ERR_INVALID, "%s: %s", path, strerror(errno));
    FileSource src(f, true, read_threads);
    uint64_t n = 0;
    bool rewound = false;
    for (;;) {
        const uint64_t want = std::min<uint64_t>(n < 2 ? 2 - n : chunk, cap - n); // the sniff of open_source, then chunks
        if (want == 0) break;
        const size_t g = src.read(dst + n, (size_t)want);
        if (g == 0) break;
        n += g;
        if (!rewound && n >= 2 + chunk) { // once: start over, as the FASTQ fallback does
            if (!src.rewind()) return hfail(FH_ERR_INVALID, "rewind failed");
            rewound = true;
            n = 0;
        }
    }
    *got = n;
    return FH_OK;
} FINCH_CATCH

// What the parsers and the device-side text paths read from an input image after magic-byte sniffing and decompression,
// requested `chunk` bytes at a time (large requests take BgzfSource's inflate-into-the-caller's-buffer route).
int finch_bgzf_batch_probe(const uint8_t *data, uint64_t len, uint64_t buf_bytes, uint32_t max_members, uint64_t text_budget,
                           uint8_t *text_out, uint64_t text_cap, uint64_t *text_len, uint64_t *n_batches, int *first_byte) try {
    if ((!data && len) || !text_out || !text_len || !n_batches || !first_byte) return hfail(FH_ERR_INVALID, "bad argument");
    std::unique_ptr<ByteSource> src;
    if (!cfg("bgzf_threads")) fh::cfg_assign("bgzf_threads", "2"); // (a BgzfSource only stands in front of gzip input when it may use threads)
    if (int rc = open_source(std::make_unique<MemSource>(data, (size_t)len), src)) return rc;
    finch::BgzfSource *bz = dynamic_cast<finch::BgzfSource *>(src.get());
    if (!bz) return hfail(FH_ERR_INVALID, "not gzip input");
    *first_byte = bz->peek_first_text_byte();
    std::vector<uint8_t> buf(buf_bytes);
    uint64_t out = 0, batches = 0;
    std::unique_ptr<finch::inf::Decoder> dec(new finch::inf::Decoder());
    for (;;) {
        uint32_t n = 0;
        uint64_t bytes = 0, text = 0;
        bool eof = false, budget_hit = false;
        if (!bz->raw_batch(buf.data(), buf.size(), max_members, text_budget, (fh_bgzf_member *)buf.data(), &n, &bytes, &text, &eof, &budget_hit))
            return hfail(FH_ERR_INVALID, "not a plain chain of BGZF members");
        batches++;
        const fh_bgzf_member *mt = (const fh_bgzf_member *)buf.data();
        for (uint32_t i = 0; i < n; ++i) { // what the device would do with the table: inflate every member where it says
            const fh_bgzf_member &m = mt[i];
            if ((uint64_t)m.in_off + m.in_len + 8 > bytes + 8 || out + m.out_off + m.isize > text_cap) return hfail(FH_ERR_INVALID, "bad table entry");
            std::vector<uint8_t> in(buf.begin() + m.in_off, buf.begin() + m.in_off + m.in_len);
            in.resize(in.size() + 16);
            if (!finch::inf::inflate_exact(*dec, in.data(), m.in_len, text_out + out + m.out_off, m.isize) ||
                finch::inf::crc32_fast(0, text_out + out + m.out_off, m.isize) != m.crc32)
                return hfail(FH_ERR_INVALID, "member %u of batch %llu does not inflate to its trailer", i, (unsigned long long)batches);
        }
        out += text;
        if (eof) break;
        if (n == 0 && !budget_hit) return hfail(FH_ERR_INVALID, "no progress");
    }
    *text_len = out;
    *n_batches = batches;
    return FH_OK;
} FINCH_CATCH

int finch_source_probe(const uint8_t *data, uint64_t len, uint64_t chunk, uint8_t *dst, uint64_t cap, uint64_t *got) try {
    if ((!data && len) || !dst || !got || chunk == 0) return hfail(FH_ERR_INVALID, "bad argument");
    std::unique_ptr<ByteSource> src;
    if (int rc = open_source(std::make_unique<MemSource>(data, (size_t)len), src)) return rc;
    uint64_t n = 0;
    while (n < cap) {
        const size_t g = src->read(dst + n, (size_t)std::min<uint64_t>(chunk, cap - n));
        if (g == 0) break;
        n += g;
    }
    if (src->failed()) return hfail(FH_ERR_INVALID, "corrupt compressed stream");
    *got = n;
    return FH_OK;
} FINCH_CATCH

// The record / total_bases bookkeeping of the device-side FASTA path (FastaCounter), fed in chunks of `chunk` bytes cut
// the way fasta_text_to_device cuts them: lets the host-only tests check it against finch_fastx_scan without a GPU.
int finch_fasta_count_chunked(const uint8_t *data, uint64_t len, uint64_t chunk, uint64_t *n_records, uint64_t *total_bases) try {
    if ((!data && len) || chunk == 0) return hfail(FH_ERR_INVALID, "bad argument");
    FastaCounter fc, fc2;
    std::vector<size_t> gts;
    uint64_t off = 0;
    while (off < len) {
        uint64_t n = std::min<uint64_t>(chunk, len - off);
        if (off + n < len) { // not the last chunk: cut after the last newline, if there is one
            const void *nl = memrchr(data + off, '\n', (size_t)n);
            if (nl) n = (uint64_t)((const uint8_t *)nl - (data + off)) + 1;
        }
        fc.feed(data + off, (size_t)n);
        // (the same chunk the way large chunks are fed: the '>' positions found beforehand, here by three threads on pieces
        //  of >= 64 bytes so that small test inputs split too)
        FastaCounter::find_gt(data + off, (size_t)n, 3, gts, 64);
        fc2.feed(data + off, (size_t)n, &gts);
        off += n;
    }
    fc.finish();
    fc2.finish();
    if (fc2.n_records != fc.n_records || fc2.total_bases != fc.total_bases)
        return hfail(FH_ERR_STATE, "FastaCounter: the two ways of feeding a chunk disagree (%llu / %llu records, %llu / %llu bases)",
                     (unsigned long long)fc.n_records, (unsigned long long)fc2.n_records, (unsigned long long)fc.total_bases,
                     (unsigned long long)fc2.total_bases);
    if (n_records) *n_records = fc.n_records;
    if (total_bases) *total_bases = fc.total_bases;
    return FH_OK;
} FINCH_CATCH

} // extern "C"
