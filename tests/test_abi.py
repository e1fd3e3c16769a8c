"""CPU-side checks of the drop-in boundary: libfinch_hip.so builds for gfx950, loads, exports every
symbol include/finch_hip.h declares, and fails loudly (no CPU fallback) when no device is present."""
import os
import re

import pytest

import finch_rs_amd as F
from finch_rs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as G
    G.build()
    return _lib.load()


def test_header_symbols_all_exported(built):
    hdr = open(os.path.join(ROOT, "include", "finch_hip.h")).read()
    declared = set(re.findall(r"\b(fh_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fh_sketcher", "fh_params", "fh_batch"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(built, name), name


def test_host_header_symbols_all_exported(built):
    from finch_rs_amd import host as H
    hdr = open(os.path.join(ROOT, "include", "finch_host.h")).read()
    declared = set(re.findall(r"\b(finch_[a-z0-9_]+)\s*\(", hdr))
    assert declared and declared == set(H._SYMS), (declared ^ set(H._SYMS))
    L = H.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_abi_version(built):
    # the library reports the version its header declares (bumped on any change of the header's functions)
    hdr = open(os.path.join(ROOT, "include", "finch_hip.h")).read()
    want = int(re.search(r"#define\s+FH_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert want >= 5
    assert built.fh_abi_version() == want


def test_no_silent_cpu_fallback(built):
    if built.fh_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(F.FinchHipError) as ei:
        F.SketchParams.default().create_sketcher()
    assert "no usable HIP device" in str(ei.value)


def test_param_validation(built):
    if built.fh_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(F.FinchHipError):
        F.SketchParams.mash(kmer_length=33).create_sketcher()


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure: nothing under finch_rs_amd/ may reference it
    pkg = os.path.join(ROOT, "finch_rs_amd")
    for dp, _, fns in os.walk(pkg):
        if "obj" in dp:
            continue
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "finch_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_one_configuration_surface(built):
    """the library reads ONE environment variable, in ONE place (csrc/fh_options.cpp: FH_DEBUG); everything else is an option
    of fh_set_option's table"""
    csrc = os.path.join(ROOT, "finch_rs_amd", "csrc")
    sites = []
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".cpp")):
            for i, line in enumerate(open(os.path.join(csrc, fn), errors="replace"), 1):
                if re.search(r"\bgetenv\s*\(", line) and not line.lstrip().startswith("//"):
                    sites.append("%s:%d" % (fn, i))
    assert len(sites) == 1 and sites[0].startswith("fh_options.cpp:"), sites
    names = [n for n, _ in F.option_list()]
    assert len(names) == len(set(names)) >= 40 and "trace" in names and "file_batch" in names
    # set / get / unset, and the one environment variable
    assert F.get_option("no_seg") is None
    F.set_option("no_seg", 1)
    assert F.get_option("no_seg") == "1"
    F.set_option("no_seg", None)
    assert F.get_option("no_seg") is None
    with pytest.raises(F.FinchHipError):
        F.set_option("no_such_option", 1)
    old = os.environ.get("FH_DEBUG")
    try:
        F.debug_set(seg_stride=151, trace=None, no_fast=1)
        assert F.get_option("seg_stride") == "151" and F.get_option("no_fast") == "1" and F.get_option("trace") is None
        F.set_option("seg_stride", 77)  # an explicit fh_set_option wins over FH_DEBUG
        assert F.get_option("seg_stride") == "77"
        F.set_option("seg_stride", None)
        assert F.get_option("seg_stride") == "151"
        os.environ["FH_DEBUG"] = "trace gz_chunk=4096;unknown_thing=3"
        assert F.get_option("trace") == "1" and F.get_option("gz_chunk") == "4096" and F.get_option("seg_stride") is None
    finally:
        if old is None:
            os.environ.pop("FH_DEBUG", None)
        else:
            os.environ["FH_DEBUG"] = old


def test_readme_lists_the_options_the_library_has(built):
    readme = open(os.path.join(ROOT, "README.md")).read()
    table = readme[readme.index("| option | effect |"):]
    listed = re.findall(r"^\| `([a-z0-9_]+)` \|", table, re.M)
    assert listed == [n for n, _ in F.option_list()]
