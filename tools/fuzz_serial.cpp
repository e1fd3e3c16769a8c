// ASan/UBSan fuzz driver of the sketch-file readers (fh_serial.cpp): 600 000 mutated .bsk / .msh / .sk inputs, none may
// trip a sanitizer.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all tools/fuzz_serial.cpp -o /tmp/fz && /tmp/fz
#include "../finch_rs_amd/csrc/fh_serial.cpp"
#include <random>
namespace finch {
thread_local std::string g_host_err;
int hfail(int code, const char *fmt, ...) { char b[512]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); g_host_err = b; return code; }
int check_compatible(const std::vector<Sketch> &) { return 0; }
}
extern "C" int finch_sketches_to_json(const finch_sketches *, char **, uint64_t *) { return -1; }
extern "C" void finch_free_string(char *) {}
int main() {
    std::vector<Sketch> v(2);
    for (int i = 0; i < 2; ++i) { v[i].name = "nm"; v[i].comment = "c"; v[i].sketch_params.kmer_length = 5;
        for (int j = 0; j < 50; ++j) v[i].hashes.push_back(KmerCount{(uint64_t)j * 77, "ACGTA", (uint32_t)j + 1, 0}); }
    std::string a, b;
    write_bsk(v, a); write_msh(v, b);
    std::mt19937_64 rng(1);
    size_t ok = 0, bad = 0;
    for (int it = 0; it < 300000; ++it) {
        std::string s = (it & 1) ? a : b;
        int nm = 1 + rng() % 4;
        for (int m = 0; m < nm; ++m) { size_t i = rng() % s.size(); if (rng() & 1) s[i] = (char)rng(); else s[i] ^= (char)(1u << (rng() % 8)); }
        if (rng() % 8 == 0) s.resize(rng() % (s.size() + 1));
        std::vector<Sketch> o;
        int r1 = read_bsk((const uint8_t *)s.data(), s.size(), o);
        int r2 = read_msh((const uint8_t *)s.data(), s.size(), o);
        (r1 == 0 || r2 == 0) ? ++ok : ++bad;
    }
    // JSON
    std::string js = "{\"kmer\":21,\"alphabet\":\"ACGT\",\"preserveCase\":false,\"canonical\":true,\"sketchSize\":3,\"hashType\":\"MurmurHash3_x64_128\",\"hashBits\":64,\"hashSeed\":42,\"scale\":null,\"sketches\":[{\"name\":\"a\\u00e9\",\"seqLength\":10,\"filters\":{\"minCopies\":\"2\"},\"hashes\":[\"1\",\"7\"],\"kmers\":[\"AC\",\"GT\"],\"counts\":[3,1]}]}";
    for (int it = 0; it < 300000; ++it) {
        std::string s = js;
        int nm = 1 + rng() % 3;
        for (int m = 0; m < nm; ++m) { size_t i = rng() % s.size(); s[i] = "{}[]\",:\\u0e-1tfn "[rng() % 18]; }
        if (rng() % 8 == 0) s.resize(rng() % (s.size() + 1));
        std::vector<Sketch> o;
        read_sk((const uint8_t *)s.data(), s.size(), o) == 0 ? ++ok : ++bad;
    }
    printf("ok %zu bad %zu\n", ok, bad);
}
