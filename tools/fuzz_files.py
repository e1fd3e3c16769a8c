#!/usr/bin/env python3
"""Fuzz of finch_sketch_files' worker groups (the many-files-per-launch path of configs[4]: fh_host.cpp workers, FastaTwoBit,
fh_pack2.h, fh_batch_*, fh_k2b.hip) against the oracle's sketch_stream: random FASTA texts -- line widths from 1 to no breaks
at all, LF / CR LF, no final line end, blank lines, header-only records, many contigs, lower case, N runs, IUPAC letters,
'>' inside lines, blanks and bytes >= 0x80 inside sequences --, random k, n, seed, thread counts and the options that change
how a file reaches the device (bytes or the two-bit form on the link, the read piece, the packer's form).  Every file's sketch,
seq_length and num_valid_kmers must be the oracle's, whichever way it went.
usage: python tools/fuzz_files.py [files=1500] [seed=1]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F  # noqa: E402
from finch_rs_amd import host as H  # noqa: E402
from finch_rs_amd.sketch_schemes import SketchParams  # noqa: E402
from oracle import oracle as O  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
ODD = np.frombuffer(b"NnUuRYKMSW>- \t\xc1\xff\x00\x0b", dtype=np.uint8)


def make_text(rng):
    eol = b"\r\n" if rng.random() < 0.25 else b"\n"
    n_rec = int(rng.choice([1, 1, 1, 2, 5, 40]))
    out = []
    for r in range(n_rec):
        kind = rng.integers(0, 8)
        L = int(rng.integers(0, 40)) if kind == 0 else int(rng.integers(40, 400_000 // n_rec + 41))
        if kind == 1:
            unit = rng.choice(ACGT, size=int(rng.integers(1, 300)))
            seq = np.tile(unit, L // len(unit) + 1)[:L]
        else:
            seq = rng.choice(ACGT, size=L)
        m = rng.random(L)
        p_odd = float(rng.choice([0.0, 0.0, 0.0005, 0.02]))
        odd = m < p_odd
        seq[odd] = rng.choice(ODD, size=int(odd.sum()))
        if rng.random() < 0.3:
            low = m > 1 - float(rng.choice([0.02, 0.5]))
            seq[low] = seq[low] | 0x20
        if rng.random() < 0.2 and L > 100:  # an N run
            a = int(rng.integers(0, L - 50))
            seq[a:a + int(rng.integers(1, 50))] = ord("N")
        seq = bytes(seq)
        width = int(rng.choice([1, 7, 60, 70, 80, 1000, 1 << 30]))
        body = eol.join(seq[j:j + width] for j in range(0, len(seq), width))
        if rng.random() < 0.1:
            body = body.replace(eol, eol + eol, 1)  # a blank line
        out.append(b">r%d some description > with a bracket" % r + (eol + body if (body or rng.random() < 0.5) else b""))
    return eol.join(out) + (eol if rng.random() < 0.8 else b"")


def main():
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    done = calls = 0
    t0 = time.time()
    taken0, not0 = H.debug_file_batch()
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        while done < want:
            k = int(rng.integers(4, 33))
            n = int(rng.choice([10, 100, 500, 1000, 1000, 1000, 2000, 3000]))
            seed = int(rng.choice([0, 0, 42, 2**63 + 5]))
            nf = int(rng.integers(1, 60))
            datas = [make_text(rng) for _ in range(nf)]
            paths = []
            for i, d in enumerate(datas):
                p = os.path.join(td, "f%03d.fa" % i)
                with open(p, "wb") as f:
                    f.write(d)
                paths.append(p)
            opts = dict(batch_two_bit=rng.choice([None, None, "0"]), batch_read_piece=rng.choice([None, "4096", "5003", "70000"]),
                        pack_scalar=rng.choice([None, None, "1", "2"]))
            F.debug_set(**opts)
            res = H.sketch_files(paths, SketchParams.mash(n, n, True, k, seed), H.FilterParams(None), n_threads=int(rng.integers(1, 9)))
            calls += 1
            for i, d in enumerate(datas):
                o = O.OracleSketcher(O.MASH, n, k, seed)
                o.sketch_stream(d)
                okc, okm = o.to_vec()
                sk = res.sketch(i)
                if not (np.array_equal(sk.arrays[0], okc) and np.array_equal(sk.arrays[1], okm)
                        and (sk.seq_length, sk.num_valid_kmers) == o.total_bases_and_kmers()):
                    with open("/tmp/fuzz_files_fail.fa", "wb") as f:
                        f.write(d)
                    print("MISMATCH k=%d n=%d seed=%d file %d of %d (%d bytes) options %s: %d vs %d hashes, (%d, %d) vs %s -> /tmp/fuzz_files_fail.fa"
                          % (k, n, seed, i, nf, len(d), opts, len(sk.arrays[0]), len(okc), sk.seq_length, sk.num_valid_kmers, o.total_bases_and_kmers()))
                    sys.exit(1)
            done += nf
    t1, n1 = H.debug_file_batch()
    print("fuzz_files: %d files in %d calls, %d sketched many-per-launch, %d of those not taken and sketched one by one: all equal to the oracle (%.0f s)"
          % (done, calls, t1 - taken0, n1 - not0, time.time() - t0))


if __name__ == "__main__":
    main()
