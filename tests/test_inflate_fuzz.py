"""The host-side DEFLATE code (fh_inflate.h, fh_pargz.h) reads files users hand it: damaged input must fail cleanly.
tools/fuzz_inflate.cpp mutates valid streams and runs the one-shot, the streaming and the parallel decoder over them with
every buffer allocated at exactly the promised size; this builds it with ASan + UBSan and runs a short campaign (the long
ones -- 160 000 streams -- are run by hand)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mutated_deflate_streams_under_sanitizers(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_inflate")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "finch_rs_amd", "csrc"), os.path.join(ROOT, "tools", "fuzz_inflate.cpp"), "-lz", "-lpthread", "-o", exe]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if b.returncode != 0 and "asan" in b.stdout.lower():
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stdout[-2000:]
    r = subprocess.run([exe, "800", "7"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "undamaged streams reproduced by all three decoders" in r.stdout
    assert "done: 800 mutated streams" in r.stdout
