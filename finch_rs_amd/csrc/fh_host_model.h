// fh_host_model.h -- the host layer's data model, shared by fh_host.cpp (sketch_files / sketch_stream / filters / .sk
// writer) and fh_serial.cpp (.sk reader, .bsk / .msh Cap'n Proto writers and readers).
//   KmerCount (sketch_schemes/mod.rs:16-22), FilterParams (filtering.rs:11-16), SketchParams (mod.rs:54-71),
//   Sketch (serialization/mod.rs:46-55) of the reference.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <exception>
#include <memory>
#include <new>
#include <vector>

#include "../../include/finch_hip.h"
#include "../../include/finch_host.h"

namespace finch {

extern thread_local std::string g_host_err;
int hfail(int code, const char *fmt, ...);

// No exception crosses the C ABI: every `int finch_*` entry point is a function-try-block ending in this.  (An allocation a
// crafted input talks the library into, or one the machine cannot serve, is an error return -- the reference's readers
// return Err there too -- not std::terminate in the caller's process.)
#define FINCH_CATCH                                                                                            \
    catch (const std::bad_alloc &) { return finch::hfail(FH_ERR_CAPACITY, "out of host memory"); }              \
    catch (const std::exception &e) { return finch::hfail(FH_ERR_INVALID, "internal error: %s", e.what()); }    \
    catch (...) { return finch::hfail(FH_ERR_INVALID, "internal error"); }

// The bytes of one k-mer.  A batch of 10 000 genomes returns ten million of them: as std::string (15 bytes inline) every k = 21
// k-mer was a heap block of its own to build and to free.  44 bytes inline cover k <= 44 (a KmerCount is 80 bytes); longer ones
// (and whatever a sketch file read from disk holds) go to the heap.
class KmerBytes {
    static constexpr uint32_t INL = 44;
    uint32_t n_ = 0;
    union {
        char inl_[INL];
        char *heap_;
    };
    void set(const char *p, size_t n) {
        n_ = (uint32_t)n;
        char *d = inl_;
        if (n > INL) d = heap_ = (char *)::operator new(n);
        if (n) memcpy(d, p, n);
    }
    void drop() {
        if (n_ > INL) ::operator delete(heap_);
        n_ = 0;
    }

  public:
    KmerBytes() {}
    KmerBytes(const char *p, size_t n) { set(p, n); }
    KmerBytes(const std::string &s) { set(s.data(), s.size()); }
    KmerBytes(const KmerBytes &o) { set(o.data(), o.size()); }
    KmerBytes(KmerBytes &&o) noexcept {
        n_ = o.n_;
        if (n_ > INL) heap_ = o.heap_;
        else memcpy(inl_, o.inl_, n_);
        o.n_ = 0;
    }
    KmerBytes &operator=(const KmerBytes &o) {
        if (this != &o) {
            KmerBytes t(o);
            drop();
            new (this) KmerBytes(std::move(t));
        }
        return *this;
    }
    KmerBytes &operator=(KmerBytes &&o) noexcept {
        if (this != &o) {
            drop();
            new (this) KmerBytes(std::move(o));
        }
        return *this;
    }
    KmerBytes &operator=(const std::string &s) {
        drop();
        set(s.data(), s.size());
        return *this;
    }
    ~KmerBytes() { drop(); }
    const char *data() const { return n_ > INL ? heap_ : inl_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    std::string str() const { return std::string(data(), n_); }
    bool operator==(const KmerBytes &o) const { return n_ == o.n_ && memcmp(data(), o.data(), n_) == 0; }
};

struct KmerCount {
    uint64_t hash;
    KmerBytes kmer;
    uint32_t count, extra_count;
    // label: Option<Vec<u8>>; None (null) for everything the sketchers emit, carried through the readers
    std::shared_ptr<const std::string> label;
};

// the same without the k-mer bytes (they stay in the copy-out array, `row` says where): what the filters work on when a
// 2 M-hash oversketch is about to be cut down to its final 10 000 -- no point in building 2 M strings first
struct KmerRef {
    uint64_t hash;
    uint32_t count, extra_count;
    uint32_t row;
};

struct Sketch {
    std::string name;
    uint64_t seq_length = 0, num_valid_kmers = 0;
    std::string comment;
    std::vector<KmerCount> hashes;
    finch_filter_params filter_params{};
    finch_sketch_params sketch_params{};
};

// SketchParams::check_compatibility over a list (mod.rs:158-226): the error text of the first mismatch with sketch 0
int check_compatible(const std::vector<Sketch> &sketches);

} // namespace finch

// Vec<Sketch> behind the C ABI
struct finch_sketches {
    std::vector<finch::Sketch> v;
};
