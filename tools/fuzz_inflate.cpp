// Mutation fuzzer for the host-side DEFLATE code (fh_inflate.h: one-shot and streaming decode; fh_pargz.h: block finder,
// marker decode from arbitrary bit offsets, window resolution), meant to run under ASan + UBSan:
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Ifinch_rs_amd/csrc tools/fuzz_inflate.cpp -lz -o /tmp/fuzz_inflate
//   /tmp/fuzz_inflate [iterations [seed]]
// Every input buffer is allocated at exactly the size the decoders are promised (payload + their padding), every output
// buffer at exactly its capacity, so a read or write beyond either is an ASan report.  Undamaged streams must reproduce
// the text (that is checked); damaged ones may fail in any way but a crash.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>

#include "fh_pargz.h"

using namespace finch;

static std::vector<uint8_t> make_text(std::mt19937_64 &rng, size_t n, int kind) {
    std::vector<uint8_t> t;
    t.reserve(n + 512);
    static const char B[] = "ACGT";
    while (t.size() < n) {
        if (kind == 2) { // bytes of any value (stored blocks, long codes)
            for (int i = 0; i < 300; ++i) t.push_back((uint8_t)rng());
            continue;
        }
        char hdr[64];
        const int hl = snprintf(hdr, sizeof hdr, "@r%llu\n", (unsigned long long)(rng() % 1000000));
        t.insert(t.end(), hdr, hdr + hl);
        const int L = 100 + (int)(rng() % 100);
        for (int i = 0; i < L; ++i) t.push_back((uint8_t)B[rng() & 3]);
        t.push_back('\n');
        t.push_back('+');
        t.push_back('\n');
        for (int i = 0; i < L; ++i) t.push_back(kind == 0 ? (uint8_t)'I' : (uint8_t)(33 + rng() % 41));
        t.push_back('\n');
    }
    t.resize(n);
    return t;
}

static std::vector<uint8_t> raw_deflate(const std::vector<uint8_t> &text, int level, int strategy) {
    z_stream z{};
    deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy);
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

static void mutate(std::mt19937_64 &rng, std::vector<uint8_t> &d) {
    if (d.empty()) return;
    const int how = (int)(rng() % 6);
    const int n = 1 + (int)(rng() % 4);
    for (int i = 0; i < n; ++i) {
        const size_t at = rng() % d.size();
        switch (how) {
        case 0: d[at] ^= (uint8_t)(1u << (rng() & 7)); break;
        case 1: d[at] = (uint8_t)rng(); break;
        case 2: d.resize(std::max<size_t>(1, at)); break;                       // truncation
        case 3: d.insert(d.begin() + (long)at, (uint8_t)rng()); break;
        case 4: d.erase(d.begin() + (long)at); break;
        default: for (size_t j = at; j < d.size() && j < at + 16; ++j) d[j] = (uint8_t)rng(); break;
        }
        if (d.empty()) d.push_back(0);
    }
}

static uint64_t n_exact_ok = 0, n_stream_ok = 0, n_par_ok = 0, n_par_tried = 0;

// one-shot decode of a stream of known size (what a BGZF member goes through)
static void run_exact(const std::vector<uint8_t> &comp, const std::vector<uint8_t> &text, bool pristine) {
    std::unique_ptr<uint8_t[]> in(new uint8_t[comp.size() + 8]);
    memcpy(in.get(), comp.data(), comp.size());
    memset(in.get() + comp.size(), 0, 8);
    std::unique_ptr<uint8_t[]> out(new uint8_t[std::max<size_t>(1, text.size())]);
    std::unique_ptr<inf::Decoder> dec(new inf::Decoder());
    const bool ok = inf::inflate_exact(*dec, in.get(), comp.size(), out.get(), text.size());
    if (ok) n_exact_ok++;
    if (pristine && (!ok || (text.size() && memcmp(out.get(), text.data(), text.size()) != 0))) {
        fprintf(stderr, "inflate_exact: undamaged stream not reproduced\n");
        abort();
    }
    // (a damaged stream may still be a valid one of the same length: nothing more to check)
}

// streaming decode into successive small buffers, history handed over the way a byte source does it
static void run_stream(std::mt19937_64 &rng, const std::vector<uint8_t> &comp, const std::vector<uint8_t> &text, bool pristine) {
    std::unique_ptr<uint8_t[]> in(new uint8_t[comp.size() + 8]);
    memcpy(in.get(), comp.data(), comp.size());
    memset(in.get() + comp.size(), 0, 8);
    std::unique_ptr<inf::Decoder> dec(new inf::Decoder());
    dec->reset();
    std::vector<uint8_t> hist; // last 32 KiB delivered
    std::vector<uint8_t> all;
    const uint8_t *ip = in.get(), *const in_end = in.get() + comp.size();
    const size_t cap = 1 + rng() % 70000;
    std::unique_ptr<uint8_t[]> buf(new uint8_t[cap]);
    size_t guard = 0;
    for (;;) {
        dec->ext_end = hist.data() + hist.size();
        dec->ext_len = hist.size();
        uint8_t *op = buf.get();
        const inf::Status s = dec->run(ip, in_end, op, buf.get() + cap, buf.get());
        const size_t got = (size_t)(op - buf.get());
        if (got) {
            all.insert(all.end(), buf.get(), buf.get() + got);
            hist.insert(hist.end(), buf.get(), buf.get() + got);
        }
        if (hist.size() > 32768) hist.erase(hist.begin(), hist.end() - 32768);
        if (s == inf::NEED_OUTPUT && all.size() < text.size() * 4 + (1u << 20) && ++guard < 100000) continue;
        if (s == inf::STREAM_END) {
            n_stream_ok++;
            if (pristine && all != text) {
                fprintf(stderr, "streaming decode: undamaged stream not reproduced\n");
                abort();
            }
        } else if (pristine) {
            fprintf(stderr, "streaming decode: undamaged stream ended with status %d\n", (int)s);
            abort();
        }
        break;
    }
}

// the parallel scheme: block starts searched from a few offsets, every chunk decoded on its own, windows chained
static void run_parallel(std::mt19937_64 &rng, const std::vector<uint8_t> &comp, const std::vector<uint8_t> &text, bool pristine) {
    // (a gzip member's DEFLATE stream is followed by its 8-byte trailer: the batch a reader hands over never ends with the stream)
    const size_t n = comp.size() + 8;
    std::unique_ptr<uint8_t[]> base(new uint8_t[n + 64]);
    memcpy(base.get(), comp.data(), comp.size());
    for (size_t i = comp.size(); i < n; ++i) base[i] = (uint8_t)rng();
    memset(base.get() + n, 0, 64);
    std::unique_ptr<inf::Decoder> scratch(new inf::Decoder());
    const size_t want = 2 + rng() % 6;
    std::vector<pargz::Chunk> chunks(want);
    chunks[0].start_bit = 0;
    chunks[0].known_window = true;
    for (size_t i = 1; i < want; ++i) {
        const uint64_t from = (uint64_t)(n * i / want) * 8u + rng() % 8;
        const uint64_t to = std::min<uint64_t>((uint64_t)n * 8u, from + (uint64_t)(n / want) * 8u);
        chunks[i].start_bit = pargz::find_block_start(base.get(), n, from, to, *scratch);
    }
    for (size_t i = 1; i < want; ++i) // starts must ascend (a later search that found an earlier block found nothing new)
        if (chunks[i].start_bit != UINT64_MAX)
            for (size_t j = 1; j < i; ++j)
                if (chunks[j].start_bit != UINT64_MAX && chunks[j].start_bit >= chunks[i].start_bit) chunks[i].start_bit = UINT64_MAX;
    n_par_tried++;
    // any order: the chunks do not depend on one another
    std::vector<size_t> order;
    for (size_t i = 0; i < want; ++i)
        if (chunks[i].start_bit != UINT64_MAX) order.push_back(i);
    for (size_t i = order.size(); i > 1; --i) std::swap(order[i - 1], order[rng() % i]);
    for (size_t ci : order) pargz::decode_chunk(base.get(), n, chunks, ci, nullptr, 0);
    // follow the chain from chunk 0: a chunk continues where the one before it ended
    std::vector<uint8_t> all, win;
    size_t ci = 0;
    bool ok = true, ended = false;
    for (;;) {
        pargz::Chunk &c = chunks[ci];
        if (!c.ok) { ok = false; break; }
        std::vector<uint8_t> piece(c.text_len());
        if (c.n_sym && !pargz::resolve_span(c.sym.data() + pargz::WINDOW, c.n_sym, win.data() + win.size(), win.size(), piece.data())) ok = false;
        if (c.n_bytes) memcpy(piece.data() + c.n_sym, c.bytes.data(), c.n_bytes);
        std::vector<uint8_t> wout;
        if (!pargz::window_behind(c, win, wout)) ok = false;
        all.insert(all.end(), piece.begin(), piece.end());
        win.swap(wout);
        if (!ok) break;
        if (c.member_end) { ended = true; break; }
        if (c.out_of_input) break;
        size_t nx = ci + 1;
        while (nx < want && chunks[nx].start_bit != c.end_bit) nx++;
        if (nx >= want) { ok = false; break; }
        ci = nx;
    }
    if (ok && ended) n_par_ok++;
    if (pristine) {
        if (!ok || !ended || all != text) {
            fprintf(stderr, "parallel decode: undamaged stream not reproduced (ok %d ended %d, %zu of %zu bytes)\n", (int)ok, (int)ended, all.size(), text.size());
            abort();
        }
    }
}

int main(int argc, char **argv) {
    const uint64_t iters = argc > 1 ? strtoull(argv[1], nullptr, 10) : 20000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed);
    struct Case { std::vector<uint8_t> text, comp; };
    std::vector<Case> cases;
    const int levels[] = {1, 6, 9};
    for (int kind = 0; kind < 3; ++kind)
        for (int lv : levels)
            for (int strategy : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY}) {
                if (strategy != Z_DEFAULT_STRATEGY && lv != 6) continue;
                Case c;
                c.text = make_text(rng, 60000 + rng() % 400000, kind);
                c.comp = raw_deflate(c.text, lv, strategy);
                cases.push_back(std::move(c));
            }
    { // level 0: stored blocks only; and the empty stream
        Case c;
        c.text = make_text(rng, 150000, 1);
        c.comp = raw_deflate(c.text, 0, Z_DEFAULT_STRATEGY);
        cases.push_back(c);
        Case e;
        e.comp = raw_deflate(e.text, 6, Z_DEFAULT_STRATEGY);
        cases.push_back(e);
    }
    for (const Case &c : cases) { // the undamaged streams first
        run_exact(c.comp, c.text, true);
        run_stream(rng, c.comp, c.text, true);
        if (c.comp.size() > 4096) run_parallel(rng, c.comp, c.text, true);
    }
    fprintf(stderr, "%zu undamaged streams reproduced by all three decoders\n", cases.size());
    for (uint64_t it = 0; it < iters; ++it) {
        const Case &c = cases[rng() % cases.size()];
        std::vector<uint8_t> d = c.comp;
        mutate(rng, d);
        switch (rng() % 3) {
        case 0: run_exact(d, c.text, false); break;
        case 1: run_stream(rng, d, c.text, false); break;
        default: if (d.size() > 4096) run_parallel(rng, d, c.text, false); break;
        }
        if ((it + 1) % 2000 == 0) fprintf(stderr, "%llu mutated streams\n", (unsigned long long)(it + 1));
    }
    printf("done: %llu mutated streams; still valid: %llu one-shot, %llu streaming, %llu of %llu parallel\n", (unsigned long long)iters,
           (unsigned long long)n_exact_ok, (unsigned long long)n_stream_ok, (unsigned long long)n_par_ok, (unsigned long long)n_par_tried);
    return 0;
}
