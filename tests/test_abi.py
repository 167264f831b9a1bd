"""CPU: the C-ABI shared library loads and exports every symbol include/segk.h declares; the ctypes table
(image_segmentation_amd/_lib.py) mirrors the header one to one (same names, same argument counts).
No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_protos():
    txt = open(os.path.join(ROOT, "include", "segk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(segk_\w+)\s*\(([^)]*)\)\s*;", txt):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",")])
        protos[m.group(1)] = n
    return protos


@pytest.fixture(scope="module")
def lib():
    from image_segmentation_amd import build, _lib
    stamp = os.path.join(os.path.dirname(_lib.LIB_PATH), ".build_id")
    built = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if not os.path.exists(_lib.LIB_PATH) or built != build.source_hash():
        build.build(verbose=False)          # incremental: only what changed (before the library is first loaded)
    return _lib


def test_header_is_nonempty():
    p = header_protos()
    assert len(p) >= 25 and "segk_conv3x3" in p and "segk_wgrad" in p and "segk_loss_fwd" in p


def test_library_exports_every_declared_symbol(lib):
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in header_protos():
        assert hasattr(so, name), f"{name} declared in include/segk.h but not exported by libsegk.so"


def test_ctypes_table_matches_header(lib):
    protos = header_protos()
    assert set(protos) == set(lib.SIGNATURES), set(protos) ^ set(lib.SIGNATURES)
    for name, nargs in protos.items():
        assert len(lib.SIGNATURES[name][1]) == nargs, name


def test_header_macros_match_declarations_and_binding(lib):
    """SEGK_ENTRY_COUNT / SEGK_ABI_VERSION are what _lib.load() compares a library against: they must follow the header's
    own declarations and the ctypes table."""
    txt = open(os.path.join(ROOT, "include", "segk.h")).read()
    count = int(re.search(r"#define\s+SEGK_ENTRY_COUNT\s+(\d+)", txt).group(1))
    version = int(re.search(r"#define\s+SEGK_ABI_VERSION\s+(\d+)", txt).group(1))
    assert count == len(header_protos()) == len(lib.SIGNATURES)
    assert version == lib.ABI_VERSION


def test_load_refuses_a_library_with_another_abi(lib, monkeypatch):
    """A diagnostic build made from another header (SEGK_LIB / kbench --lib) is refused instead of being driven with this
    binding's buffer-size queries (the round-2 GPU fault: DESIGN.md 4.2)."""
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "ABI_VERSION", lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="refusing to drive"):
        lib.load()
    monkeypatch.setattr(lib, "ABI_VERSION", lib.ABI_VERSION - 1)
    sig = dict(lib.SIGNATURES); sig.pop("segk_clock_probe")
    monkeypatch.setattr(lib, "SIGNATURES", sig)
    with pytest.raises(RuntimeError, match="refusing to drive"):
        lib.load()


def test_load_and_version(lib):
    so = lib.load()
    assert so.segk_version() == lib.ABI_VERSION and so.segk_entry_count() == len(lib.SIGNATURES)
    assert so.segk_last_error() is not None
    # pure size queries work without a GPU
    assert lib.query("segk_conv_tiles", 2, 32, 32, 256, 64, 1) == 2 * 2 * 1    # 16x32 tiles: bf16 64-ch output, long K
    assert lib.query("segk_conv_tiles", 2, 32, 32, 256, 64, 0) == 2 * 4 * 2    # 8x16 tiles: the same layer in fp32
    assert lib.query("segk_conv_tiles", 2, 32, 32, 64, 64, 1) == 8 * 1 * 4     # register-stationary kernel: one row of
    #                                                   partials per XCD slot (8), workgroup per slot (1) and wave slab (4)
    assert lib.query("segk_conv_tiles", 2, 32, 16, 64, 64, 1) == 2 * 1 * 2     # 16x16 tiles: weight-stationary kernel
    assert lib.query("segk_conv_tiles", 2, 32, 32, 128, 128, 1) == 2 * 4 * 1   # 8x32 tiles for 128 channels
    assert lib.query("segk_bn_stats_floats", 4, 64) == 4 * 64 * 2 + 32 * 64 * 4
    assert lib.query("segk_loss_state_floats") >= 4 + 3 * 8


def test_library_was_built_from_the_sources_in_tree(lib):
    """segk_build_id() is the source hash compiled into api.o: a stale libsegk.so (built from other sources than the
    ones shipped beside it) fails here instead of silently running old kernels."""
    from image_segmentation_amd import build
    assert lib.build_id() == build.source_hash()


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libsegk.so")
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        lib.load()
