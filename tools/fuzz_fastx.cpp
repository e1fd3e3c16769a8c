// Mutation fuzzer for the host reader stack (fh_host.cpp: format sniffing, gzip / BGZF containers, the serial and the
// multi-threaded inflate sources, the FASTA / FASTQ parser) through its GPU-free entry points, under ASan + UBSan:
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude -Ifinch_rs_amd/csrc \
//       tools/fuzz_fastx.cpp finch_rs_amd/csrc/fh_host.cpp finch_rs_amd/csrc/fh_serial.cpp finch_rs_amd/csrc/fh_options.cpp \
//       -Lfinch_rs_amd -lfinch_hip -Wl,-rpath,$PWD/finch_rs_amd -lz -ldl -lpthread -o /tmp/fuzz_fastx
//   ASAN_OPTIONS=detect_leaks=0 /tmp/fuzz_fastx [iterations [seed]]
// (the device engine's entry points come from libfinch_hip.so and are never reached: finch_fastx_scan only reads and
// counts).  Undamaged inputs must give the record count and base total they were built with, whatever the container and
// the number of inflate threads; damaged ones may fail, not crash.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>

#include "finch_host.h"

static std::vector<uint8_t> fastq(std::mt19937_64 &rng, int n_rec, uint64_t *bases, bool crlf) {
    std::vector<uint8_t> t;
    static const char B[] = "ACGTN";
    *bases = 0;
    const char *nl = crlf ? "\r\n" : "\n";
    for (int r = 0; r < n_rec; ++r) {
        char h[48];
        const int hl = snprintf(h, sizeof h, "@read%d x%s", r, nl);
        t.insert(t.end(), h, h + hl);
        const int L = 1 + (int)(rng() % 300);
        for (int i = 0; i < L; ++i) t.push_back((uint8_t)B[rng() % 5]);
        t.insert(t.end(), nl, nl + strlen(nl));
        t.push_back('+');
        t.insert(t.end(), nl, nl + strlen(nl));
        for (int i = 0; i < L; ++i) t.push_back((uint8_t)(33 + rng() % 60));
        t.insert(t.end(), nl, nl + strlen(nl));
        *bases += (uint64_t)L;
    }
    return t;
}

static std::vector<uint8_t> fasta(std::mt19937_64 &rng, int n_rec, uint64_t *bases) {
    std::vector<uint8_t> t;
    static const char B[] = "ACGTacgtN";
    *bases = 0;
    for (int r = 0; r < n_rec; ++r) {
        char h[48];
        const int hl = snprintf(h, sizeof h, ">seq%d some text\n", r);
        t.insert(t.end(), h, h + hl);
        const int L = (int)(rng() % 5000), W = 40 + (int)(rng() % 60);
        for (int i = 0; i < L; ++i) {
            t.push_back((uint8_t)B[rng() % 9]);
            if ((i + 1) % W == 0 && i + 1 < L) {
                t.push_back('\n');
                *bases += 1; // needletail counts the raw region: line ends inside it included, the last one not
            }
        }
        if (L) t.push_back('\n');
        *bases += (uint64_t)L;
    }
    return t;
}

static std::vector<uint8_t> gzip_of(const std::vector<uint8_t> &text, int level) {
    z_stream z{};
    deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);
    std::vector<uint8_t> out(deflateBound(&z, text.size()) + 64);
    z.next_in = const_cast<Bytef *>(text.data());
    z.avail_in = (uInt)text.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    deflate(&z, Z_FINISH);
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// bgzip's container: members of <= 65280 text bytes, extra field BC with the member's size, the empty end-of-file member
static std::vector<uint8_t> bgzf_of(const std::vector<uint8_t> &text, int level) {
    std::vector<uint8_t> out;
    size_t off = 0;
    for (;;) {
        const size_t n = std::min<size_t>(65280, text.size() - off);
        z_stream z{};
        deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        std::vector<uint8_t> d(deflateBound(&z, n) + 64);
        z.next_in = const_cast<Bytef *>(text.data() + off);
        z.avail_in = (uInt)n;
        z.next_out = d.data();
        z.avail_out = (uInt)d.size();
        deflate(&z, Z_FINISH);
        d.resize(z.total_out);
        deflateEnd(&z);
        const uint32_t bsize = (uint32_t)(d.size() + 25);
        const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)bsize, (uint8_t)(bsize >> 8)};
        out.insert(out.end(), hdr, hdr + 18);
        out.insert(out.end(), d.begin(), d.end());
        const uint32_t crc = (uint32_t)crc32(0, text.data() + off, (uInt)n), isz = (uint32_t)n;
        for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(crc >> (8 * i)));
        for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(isz >> (8 * i)));
        off += n;
        if (n == 0) break; // (the empty member closes the file)
    }
    return out;
}

static void mutate(std::mt19937_64 &rng, std::vector<uint8_t> &d) {
    const int how = (int)(rng() % 7), n = 1 + (int)(rng() % 3);
    for (int i = 0; i < n && !d.empty(); ++i) {
        const size_t at = rng() % d.size();
        switch (how) {
        case 0: d[at] ^= (uint8_t)(1u << (rng() & 7)); break;
        case 1: d[at] = (uint8_t)rng(); break;
        case 2: d.resize(at); break;
        case 3: d.insert(d.begin() + (long)at, (uint8_t)"\n>@+\r \t"[rng() % 7]); break;
        case 4: d.erase(d.begin() + (long)at); break;
        case 5: d[at] = (uint8_t)"\n>@+\r \t"[rng() % 7]; break;
        default: { const size_t m = std::min<size_t>(d.size() - at, 1 + rng() % 64); d.erase(d.begin() + (long)at, d.begin() + (long)(at + m)); } break;
        }
    }
}

struct Case {
    std::vector<uint8_t> image;
    uint64_t records, bases;
    int format;
    const char *what;
};

static uint64_t n_ok = 0, n_err = 0;
static void scan(const std::vector<uint8_t> &img, const Case *expect) {
    // exactly-sized heap copy: a read past the image is an ASan report
    uint8_t *p = (uint8_t *)malloc(img.size() ? img.size() : 1);
    if (!img.empty()) memcpy(p, img.data(), img.size());
    uint64_t nr = 0, tb = 0;
    int fmt = 0;
    const int rc = finch_fastx_scan(p, img.size(), &nr, &tb, &fmt);
    free(p);
    if (rc == 0) n_ok++;
    else n_err++;
    if (expect && (rc != 0 || nr != expect->records || tb != expect->bases || fmt != expect->format)) {
        fprintf(stderr, "%s: undamaged input scanned as rc %d records %llu bases %llu format %d (want %llu %llu %d): %s\n", expect->what, rc,
                (unsigned long long)nr, (unsigned long long)tb, fmt, (unsigned long long)expect->records, (unsigned long long)expect->bases,
                expect->format, rc ? finch_last_error() : "");
        abort();
    }
}

static uint64_t n_two_bit = 0;

int main(int argc, char **argv) {
    const uint64_t iters = argc > 1 ? strtoull(argv[1], nullptr, 10) : 20000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed);
    // (the library's options travel in FH_DEBUG: several chunks per gzip member even in these small files, 1 or 4 inflate threads)
    auto options = [](const char *threads) { setenv("FH_DEBUG", (std::string("pargz_chunk=65536,bgzf_threads=") + threads).c_str(), 1); };
    std::vector<Case> cases;
    for (int rep = 0; rep < 3; ++rep) {
        uint64_t b = 0;
        const int nq = 50 + (int)(rng() % 3000);
        const std::vector<uint8_t> q = fastq(rng, nq, &b, rep == 2);
        cases.push_back({q, (uint64_t)nq, b, 2, "fastq"});
        cases.push_back({gzip_of(q, rep == 0 ? 1 : 6), (uint64_t)nq, b, 2, "fastq.gz"});
        cases.push_back({bgzf_of(q, 6), (uint64_t)nq, b, 2, "fastq.bgzf"});
        const int na = 1 + (int)(rng() % 200);
        const std::vector<uint8_t> a = fasta(rng, na, &b);
        cases.push_back({a, (uint64_t)na, b, 1, "fasta"});
        cases.push_back({gzip_of(a, 6), (uint64_t)na, b, 1, "fasta.gz"});
        cases.push_back({bgzf_of(a, 1), (uint64_t)na, b, 1, "fasta.bgzf"});
    }
    const char *threads[] = {"1", "4"};
    for (const char *t : threads) {
        options(t);
        for (const Case &c : cases) scan(c.image, &c);
    }
    fprintf(stderr, "%zu undamaged inputs scanned right with 1 and 4 inflate threads\n", cases.size());
    for (uint64_t it = 0; it < iters; ++it) {
        options(threads[rng() & 1]);
        const Case &c = cases[rng() % cases.size()];
        std::vector<uint8_t> d = c.image;
        mutate(rng, d);
        scan(d, nullptr);
        // the batch path's FASTA walk into the two-bit form (FastaTwoBit, fh_pack2.h) on the same bytes, in pieces of a random size, in
        // each of the packer's forms: the region it is given is EXACTLY as large as the header says it must be (ASan watches the end)
        if (!d.empty() && d[0] == '>') {
            const uint64_t cap = ((d.size() + 2047) / 2048 + 1) * 768;
            std::vector<uint8_t> region(cap);
            static const uint64_t pieces[] = {1, 7, 31, 32, 33, 64, 1000, 4096, 16384, 1 << 20};
            static const char *forms[] = {"0", "1", "2"};
            setenv("FH_DEBUG", (std::string("pack_scalar=") + forms[rng() % 3]).c_str(), 1);
            uint64_t pos = 0, nr = 0, nb = 0;
            if (finch_fasta_two_bit_probe(d.data(), d.size(), pieces[rng() % 10], region.data(), cap, &pos, &nr, &nb) != 0 || pos > d.size()) {
                fprintf(stderr, "two-bit walk failed on a text of %zu bytes (positions %llu)\n", d.size(), (unsigned long long)pos);
                return 1;
            }
            ++n_two_bit;
        }
        if ((it + 1) % 2000 == 0) fprintf(stderr, "%llu mutated inputs\n", (unsigned long long)(it + 1));
    }
    printf("done: %llu mutated inputs, %llu scanned, %llu refused; %llu plain FASTA texts also walked into the two-bit form\n", (unsigned long long)iters,
           (unsigned long long)n_ok, (unsigned long long)n_err, (unsigned long long)n_two_bit);
    return 0;
}
