"""Error behaviour of the C ABI on a real device: loud, specific failures instead of fallbacks."""
import ctypes as C

import numpy as np
import pytest

import finch_rs_amd as F
from finch_rs_amd import _lib
from finch_rs_amd import sketch_schemes as S

pytestmark = pytest.mark.gpu


def test_unsupported_and_invalid_parameters():
    with pytest.raises(F.FinchHipError, match="outside the device range"):
        F.SketchParams.mash(kmer_length=65).create_sketcher()  # 33..64 run the two-word kernels; beyond that: refused
    with pytest.raises(F.FinchHipError, match="outside the device range"):
        F.SketchParams.mash(kmer_length=0).create_sketcher()
    with pytest.raises(F.FinchHipError, match="scale"):
        F.SketchParams.scaled(10, 21, 0.0).create_sketcher()
    with pytest.raises(F.FinchHipError, match="no usable HIP device"):
        F.SketchParams.default().create_sketcher(device=99)
    with pytest.raises(F.FinchError):
        F.SketchParams("none").create_sketcher()


def test_state_machine():
    L = _lib.load()
    sk = F.SketchParams.mash(10, 10, True, 5, 0).create_sketcher()
    rc = L.fh_copy_out(sk._h, None, None, None, None, None)
    assert rc == _lib.FH_ERR_STATE and b"before fh_finish" in L.fh_last_error()
    sk.process(b"ACGTACGTAC")
    assert sk.finish()[0] > 0
    blk = np.frombuffer(b"ACGTAC\x00", dtype=np.uint8)
    rc = L.fh_push_block(sk._h, blk.ctypes.data_as(C.c_void_p), len(blk))
    assert rc == _lib.FH_ERR_STATE and b"already finished" in L.fh_last_error()
    sk.reset()
    sk.process(b"ACGTACGTAC")
    assert sk.finish()[0] > 0
    # device blocks must be 16-byte aligned
    buf = F.DeviceBuffer(256)
    sk.reset()
    with pytest.raises(F.FinchHipError, match="16-byte aligned"):
        sk.push_device(buf.ptr + 1, 64)
    # merging incompatible sketches is refused
    a = F.SketchParams.mash(10, 10, True, 5, 0).create_sketcher()
    b = F.SketchParams.mash(10, 10, True, 7, 0).create_sketcher()
    a.process(b"ACGTACGTACGT"); b.process(b"ACGTACGTACGT")
    a.finish(); b.finish()
    with pytest.raises(F.FinchHipError, match="incompatible"):
        a.merge(b)


def test_strict_mode_error_text_matches_reference():
    from finch_rs_amd import host as H
    # mod.rs:123-125: "{name} had too few kmers ({n}) to sketch"
    with pytest.raises(S.FinchError, match=r"^tiny had too few kmers \(2\) to sketch$"):
        H.sketch_stream(b">x\nACGTACGTACGTACGTACGTACGT\n", "tiny", F.SketchParams.default(), H.FilterParams(False))
