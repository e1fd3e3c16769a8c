"""End-to-end rate of one large FASTQ / FASTA file (page cache -> pinned staging -> device-side record splitting ->
sketch) by the option read_threads.  Each setting runs in a child process (the knob is read once).
usage (GPU box): python tools/e2e_read_threads.py"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
from finch_rs_amd import host as H
from finch_rs_amd.sketch_schemes import SketchParams
p = SketchParams.mash(1000, 1000, True, 21, 0)
for path, bases in (("/tmp/e2e.fastq", 600e6), ("/tmp/e2e.fa", 200e6)):
    size = os.path.getsize(path)
    best = 1e9
    for rep in range(4):
        t = time.time(); res = H.sketch_files([path], p, H.FilterParams(False)); best = min(best, time.time() - t)
    a = res.sketch(0).arrays[0]
    print("  %%s: %%.3f s  %%.1f GB/s text  %%.2f Gbases/s  (xor %%x)" %% (os.path.basename(path), best, size / best / 1e9, bases / best / 1e9, int(__import__("numpy").bitwise_xor.reduce(a["hash"]))), flush=True)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.exists("/tmp/e2e.fastq"):
    from finch_rs_amd import sketch_schemes as S
    n_reads, rl = 4_000_000, 150
    g = S.synth_genome_host(5_000_000, 1)
    reads = S.synth_reads_host(g, 0, n_reads, rl, 1, 10000, 500).reshape(n_reads, rl + 1)[:, :rl]
    with open("/tmp/e2e.fastq", "wb") as f:
        for i in range(n_reads):
            f.write(b"@r%d\n" % i); f.write(reads[i].tobytes()); f.write(b"\n+\n"); f.write(b"I" * rl); f.write(b"\n")
    seq = S.synth_genome_host(200_000_000, 7).tobytes()
    with open("/tmp/e2e.fa", "wb") as f:
        f.write(b">chr\n")
        for i in range(0, len(seq), 70 * 100000):
            blk = seq[i:i + 70 * 100000]
            f.write(b"\n".join(blk[j:j + 70] for j in range(0, len(blk), 70))); f.write(b"\n")
for nt in (1, 2, 4, 8):
    print("read_threads=%d" % nt, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FH_DEBUG="read_threads=%d" % nt), check=True)
