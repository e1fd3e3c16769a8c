// fh_host_model.h -- the host layer's data model, shared by fh_host.cpp (sketch_files / sketch_stream / filters / .sk
// writer) and fh_serial.cpp (.sk reader, .bsk / .msh Cap'n Proto writers and readers).
//   KmerCount (sketch_schemes/mod.rs:16-22), FilterParams (filtering.rs:11-16), SketchParams (mod.rs:54-71),
//   Sketch (serialization/mod.rs:46-55) of the reference.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <exception>
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

struct KmerCount {
    uint64_t hash;
    std::string kmer;
    uint32_t count, extra_count;
    bool has_label = false; // label: Option<Vec<u8>>; None for everything the sketchers emit, carried through the readers
    std::string label;
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
