"""fh_strip.h (the blank-dropping copy of the staging paths: normalize(false)'s ' ', '\\t', '\\r', '\\n', mash.rs:73) compiled for
the host: the AVX2 loop against the byte-by-byte definition on random text with every density of blanks, packing towards the
front of the same buffer, and count_kept.  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "fh_strip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
static size_t ref(uint8_t *d, const uint8_t *s, size_t n) { size_t m = 0; for (size_t i = 0; i < n; ++i) if (s[i] != ' ' && s[i] != '\t' && s[i] != '\r' && s[i] != '\n') d[m++] = s[i]; return m; }
int main() {
    srand(5);
    const char *alpha = "ACGTNacgt>@\xff*-";
    for (int it = 0; it < 30000; ++it) {
        const size_t n = (size_t)(rand() % 900);
        std::vector<uint8_t> a(n + 64), b(n + 64), c(n + 64), d(n + 64);
        const int pb = rand() % 101;
        for (size_t i = 0; i < n; ++i) a[i] = (rand() % 100 < pb) ? (uint8_t)" \t\r\n"[rand() % 4] : (uint8_t)alpha[rand() % 16];
        const size_t m0 = ref(b.data(), a.data(), n), m1 = fh_strip::strip(c.data(), a.data(), n), m2 = fh_strip::strip_scalar(d.data(), a.data(), n);
        if (m0 != m1 || m0 != m2 || memcmp(b.data(), c.data(), m0) || memcmp(b.data(), d.data(), m0)) { printf("MISMATCH n=%zu\n", n); return 1; }
        if (fh_strip::count_kept(a.data(), n) != m0 || fh_strip::count_kept_scalar(a.data(), n) != m0) { printf("COUNT n=%zu\n", n); return 1; }
    }
    for (int it = 0; it < 3000; ++it) { // towards the front of the same buffer, source at least 32 bytes behind the destination
        const size_t n = 1000 + (size_t)(rand() % 5000);
        std::vector<uint8_t> a(2 * n + 160), r(n + 64);
        uint8_t *src = a.data() + 32 + rand() % 64;
        for (size_t i = 0; i < n; ++i) src[i] = (i % 71 == 70 || rand() % 50 == 0) ? '\n' : (uint8_t)"ACGT"[rand() % 4];
        const size_t m0 = ref(r.data(), src, n), m1 = fh_strip::strip(a.data(), src, n);
        if (m0 != m1 || memcmp(r.data(), a.data(), m0)) { printf("INPLACE MISMATCH\n"); return 1; }
    }
    for (size_t n = 0; n < 200; ++n) // the overlapping last vector: clean records of every length, then one blank at every place of the tail
        for (size_t blank = 0; blank <= n; ++blank) {
            std::vector<uint8_t> a(n + 64), b(n + 64), c(n + 96, 0xEE);
            for (size_t i = 0; i < n; ++i) a[i] = (uint8_t)"ACGT"[(i * 7 + n) & 3];
            if (blank < n) a[blank] = '\n';
            const size_t m0 = ref(b.data(), a.data(), n), m1 = fh_strip::strip(c.data(), a.data(), n);
            if (m0 != m1 || memcmp(b.data(), c.data(), m0)) { printf("TAIL MISMATCH n=%zu blank=%zu\n", n, blank); return 1; }
            for (size_t i = m0 + 32; i < n + 96; ++i) if (c[i] != 0xEE) { printf("TAIL WROTE BEYOND ITS SLACK n=%zu\n", n); return 1; }
        }
    puts("strip ok");
    return 0;
}
'''


def test_strip_matches_the_definition(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src, exe = tmp_path / "t.cpp", tmp_path / "t"
    src.write_text(SRC)
    b = subprocess.run([gxx, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "finch_rs_amd", "csrc"), str(src), "-o", str(exe)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert b.returncode == 0, b.stdout[-2000:]
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "strip ok" in r.stdout, r.stdout[-500:]
