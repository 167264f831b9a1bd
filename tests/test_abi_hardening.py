"""CPU: host-side hardening of the C ABI (include/segk.h).

  * every entry point, called with NULL pointers / zero / negative / misaligned sizes / unknown dtypes, returns the
    validation error code and a message -- driven in a child process against an AddressSanitizer + UBSan build of the
    HOST half of the library (the device code cannot be sanitized on this pool), so an out-of-bounds read, an integer
    overflow or a division by zero in the argument handling fails the test instead of passing silently;
  * property tests (hypothesis) over the pure size queries the Python host sizes its buffers with.
No kernel is launched (this box has no GPU)."""
import json
import os
import subprocess
import sys

import pytest
from hypothesis import given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_every_entry_rejects_invalid_arguments_under_asan_ubsan():
    from image_segmentation_amd import build
    lib = build.build_sanitized()
    env = dict(os.environ, LD_PRELOAD=build.asan_runtime(),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_fuzz_worker.py"), lib], env=env,
                       capture_output=True, text=True, timeout=600)
    tail = "\n".join(r.stderr.splitlines()[-25:])
    assert r.returncode == 0, f"worker died (sanitizer report or crash), last lines:\n{tail}"
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["calls"] > 250
    assert out["bad"] == [], out["bad"][:5]


@pytest.fixture(scope="module")
def lib():
    from image_segmentation_amd import _lib
    _lib.load()
    return _lib


dims = st.integers(min_value=1, max_value=300)
chans = st.sampled_from([32, 64, 96, 128, 256, 512, 1024])


@settings(max_examples=200, deadline=None)
@given(B=st.integers(1, 9), H=dims, W=dims, cin=chans, cout=chans, dtype=st.integers(0, 1))
def test_conv_tiles_properties(lib, B, H, W, cin, cout, dtype):
    t = lib.query("segk_conv_tiles", B, H, W, cin, cout, dtype)
    assert t >= 1
    # never more rows than the smallest tile (8 x 16 pixels) would give, or -- register-stationary kernel -- four rows
    # per workgroup of at most 256
    assert t <= max(B * ((H + 7) // 8) * ((W + 15) // 16), 1024)
    # the statistics buffer the host allocates from it holds every row (sum, sum of squares per channel) plus the
    # finalize scratch
    floats = lib.query("segk_bn_stats_floats", t, cout)
    assert floats >= t * cout * 2
    # deterministic, and monotone in the batch
    assert lib.query("segk_conv_tiles", B, H, W, cin, cout, dtype) == t
    assert lib.query("segk_conv_tiles", B + 1, H, W, cin, cout, dtype) >= t


@settings(max_examples=200, deadline=None)
@given(B=st.integers(1, 9), H=dims, W=dims, geo=st.integers(0, 2), dtype=st.integers(0, 1))
def test_wgrad_tiles_properties(lib, B, H, W, geo, dtype):
    R = (4 if dtype else 2) if geo == 2 else (8 if dtype else 4)
    assert lib.query("segk_wgrad_tiles", B, H, W, geo, dtype) == B * ((H + R - 1) // R) * ((W + 15) // 16)


@settings(max_examples=300, deadline=None)
@given(P=st.integers(-5, 1 << 24), C=st.integers(-64, 2048), dtype=st.integers(0, 1))
def test_bn_block_queries_properties(lib, P, C, dtype):
    nb = lib.query("segk_bn_bwd_blocks", P, C, dtype)
    if P <= 0 or C <= 0 or C % 32:
        assert nb == 0
    else:
        assert 1 <= nb <= 512
        assert lib.query("segk_bn_bwd_blocks", P + 1000, C, dtype) >= nb


@settings(max_examples=200, deadline=None)
@given(B=st.integers(-1, 9), H=st.integers(-1, 300), W=st.integers(-1, 300), C=st.integers(-32, 1100), dtype=st.integers(0, 1))
def test_maxpool_stat_blocks_properties(lib, B, H, W, C, dtype):
    nb = lib.query("segk_maxpool_bwd_stat_blocks", B, H, W, C, dtype)
    assert 0 <= nb <= 1024
    if B <= 0 or H < 2 or W < 2 or C <= 0 or C % 32:
        assert nb == 0


@settings(max_examples=200, deadline=None)
@given(P=st.integers(1, 1 << 24), Cp=chans)
def test_head_and_loss_scratch_properties(lib, P, Cp):
    hb = lib.query("segk_head_bwd_blocks", P)
    assert hb >= 1
    assert lib.query("segk_head_part_floats", P, Cp) == hb * lib.MAX_CLASSES * (Cp + 1)
    lp = lib.query("segk_loss_part_floats", P)
    assert lp >= 1 and lib.query("segk_loss_part_floats", P + 4096) >= lp
    assert lib.query("segk_loss_state_floats") >= 4 + 3 * lib.MAX_CLASSES
