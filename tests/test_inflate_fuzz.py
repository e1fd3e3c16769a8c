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


def test_mutated_fastx_images_under_sanitizers(tmp_path):
    """tools/fuzz_fastx.cpp: the whole reader stack of fh_host.cpp (sniffing, gzip / BGZF containers, the serial and the
    multi-threaded inflate sources, the FASTA / FASTQ parser) through finch_fastx_scan, and the batch path's FASTA walk into the
    two-bit form (finch_fasta_two_bit_probe) on the mutated plain FASTA texts, rebuilt with ASan + UBSan; the
    device engine comes from libfinch_hip.so and is never called."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    from finch_rs_amd import _lib
    _lib.load()  # (builds the library if it is missing)
    so_dir = os.path.dirname(_lib.SO_PATH)
    exe = str(tmp_path / "fuzz_fastx")
    csrc = os.path.join(ROOT, "finch_rs_amd", "csrc")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(ROOT, "tools", "fuzz_fastx.cpp"),
           os.path.join(csrc, "fh_host.cpp"), os.path.join(csrc, "fh_serial.cpp"), "-L" + so_dir, "-lfinch_hip",
           "-Wl,-rpath," + so_dir, "-lz", "-ldl", "-lpthread", "-o", exe]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if b.returncode != 0 and "asan" in b.stdout.lower():
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stdout[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe, "400", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "undamaged inputs scanned right with 1 and 4 inflate threads" in r.stdout
    assert "done: 400 mutated inputs" in r.stdout
    assert "walked into the two-bit form" in r.stdout  # (the batch path's FASTA walk and packer, exact-size regions)
