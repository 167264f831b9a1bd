"""GPU (-m gpu): each HIP kernel family through the C ABI against the CPU oracle (stock torch fp32 ops on
the same seeded inputs).  fp32 mode must agree to ~1e-5 (exact-fp32 MFMA vs oneDNN summation order);
bf16 mode is compared with a tolerance scaled to bf16 rounding of inputs/outputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.fill import fill, labels as fill_labels

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from image_segmentation_amd import ops as o
    return o


def tol(dtype, k):
    """abs tolerance for a length-k dot product of O(1) operands"""
    return 2e-5 * max(1.0, np.sqrt(k) / 8) if dtype == torch.float32 else 2.5e-2 * max(1.0, np.sqrt(k) / 8)


def dev(t):
    return t.cuda()


def back(t):
    return t.detach().float().cpu()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (1, 32, 4, 4), (2, 70, 9, 3)])
def test_layout_roundtrip(ops, dtype, shape):
    x = fill(shape, 3, -1, 1)
    a = ops.to_act(dev(x), dtype)
    assert ops.act_info(a, dtype) is not None and a.shape == x.shape
    ref = x.to(dtype).float()
    assert torch.equal(back(a), ref)
    assert torch.equal(ops.act_to_nchw(a).cpu(), ref)
    # padding channels are zero
    ptr, Cp = ops.act_info(a, dtype)
    assert Cp % 32 == 0 and Cp >= shape[1]


def _conv_direct(ops, x, w, dtype, xb=None, scale=None, shift=None, stats=False, mode=0):
    """run segk_conv3x3 on act tensors; returns (out[, out2]) as NCHW fp32 cpu and stats"""
    from image_segmentation_amd import _lib
    B, CA, H, W = x.shape
    CB = 0 if xb is None else xb.shape[1]
    Cout = w.shape[0]
    xa = ops.to_act(dev(x), dtype); pa, CAp = ops.act_info(xa, dtype)
    pb, CBp = (0, 0)
    if xb is not None:
        xbt = ops.to_act(dev(xb), dtype); pb, CBp = ops.act_info(xbt, dtype)
    wp = ops.pack_conv(dev(w), CA, CB, dtype, mode)
    Coutp = ops.pad32(Cout)
    out = torch.empty((B, H, W, Coutp), dtype=dtype, device="cuda")
    tiles = _lib.query("segk_conv_tiles", B, H, W, CAp + CBp, Coutp, 1 if dtype == torch.bfloat16 else 0)
    st = torch.zeros((_lib.query("segk_bn_stats_floats", tiles, Coutp),), dtype=torch.float32, device="cuda") if stats else None
    ops.conv3x3(xa, pa, CAp, pb, CBp, wp, out.data_ptr(), Coutp, 0, 0, B, H, W, dtype,
                scale=None if scale is None else dev(scale), shift=None if shift is None else dev(shift), stats=st)
    torch.cuda.synchronize()
    y = back(ops.act_view(out, Cout))
    return (y, st[:tiles * Coutp * 2].view(tiles, Coutp, 2).cpu()) if stats else y


CONV_CASES = [  # B, Cin, Cout, H, W
    (2, 32, 32, 16, 16), (1, 64, 64, 32, 32), (2, 3, 64, 24, 40), (1, 128, 128, 8, 8),
    (3, 96, 160, 14, 14), (1, 32, 64, 17, 33), (2, 256, 64, 16, 48),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_forward(ops, dtype, case):
    B, Cin, Cout, H, W = case
    x = fill((B, Cin, H, W), 1, -1, 1)
    w = fill((Cout, Cin, 3, 3), 2, -1, 1) / np.sqrt(9 * Cin)
    xr, wr = x.to(dtype).float(), w.to(dtype).float()
    ref = F.conv2d(xr, wr, padding=1)
    y, st = _conv_direct(ops, x, w, dtype, stats=True)
    assert (y - ref).abs().max().item() < tol(dtype, 1) * 1.5
    # BN statistics partials: sum and sum of squares per channel (taken from the fp32 accumulators)
    s = st.sum(0)[:Cout]
    close = 1e-3 if dtype == torch.float32 else 2e-2
    assert torch.allclose(s[:, 0], ref.sum(dim=(0, 2, 3)), rtol=close, atol=close * B * H * W * 0.05)
    assert torch.allclose(s[:, 1], (ref * ref).sum(dim=(0, 2, 3)), rtol=close, atol=close)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_prologue_and_dual(ops, dtype):
    B, H, W = 2, 20, 36
    # fused BN+ReLU prologue
    z = fill((B, 64, H, W), 1, -2, 2); w = fill((32, 64, 3, 3), 2, -1, 1) / 24
    sc = fill((64,), 3, -1.5, 1.5); sh = fill((64,), 4, -0.5, 0.5)
    zr = z.to(dtype).float()
    a = torch.relu(zr * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dtype).float()
    ref = F.conv2d(a, w.to(dtype).float(), padding=1)
    y = _conv_direct(ops, z, w, dtype, scale=sc, shift=sh)
    assert (y - ref).abs().max().item() < tol(dtype, 1) * 2
    # two sources == conv of the concatenation (40 + 72 channels: both padded, 40 -> 64, 72 -> 96)
    xa = fill((B, 40, H, W), 5, -1, 1); xb = fill((B, 72, H, W), 6, -1, 1)
    w2 = fill((64, 112, 3, 3), 7, -1, 1) / 32
    ref = F.conv2d(torch.cat([xa, xb], 1).to(dtype).float(), w2.to(dtype).float(), padding=1)
    y = _conv_direct(ops, xa, w2, dtype, xb=xb)
    assert (y - ref).abs().max().item() < tol(dtype, 1) * 2


def test_conv3x3_bf16_persistent_units(ops):
    """producer/consumer kernel with several work units per workgroup (the weight ring and the patch pipeline run
    across unit boundaries) and the 512-pixel x 64-channel variant, against fp32 torch on the same bf16 inputs"""
    dtype = torch.bfloat16
    for B, Cin, Cout, H, W in [(4, 128, 256, 128, 128), (3, 128, 64, 96, 160)]:
        x = fill((B, Cin, H, W), 11, -1, 1)
        w = fill((Cout, Cin, 3, 3), 12, -1, 1) / np.sqrt(9 * Cin)
        ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), padding=1)
        y, st = _conv_direct(ops, x, w, dtype, stats=True)
        assert (y - ref).abs().max().item() < tol(dtype, 1) * 1.5
        s = st.sum(0)[:Cout]
        assert torch.allclose(s[:, 0], ref.sum(dim=(0, 2, 3)), rtol=2e-2, atol=2e-2 * B * H * W * 0.05)
        assert torch.allclose(s[:, 1], (ref * ref).sum(dim=(0, 2, 3)), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("case", [(3, 128, 128, 37, 53), (2, 256, 192, 20, 24), (2, 128, 128, 13, 11), (1, 512, 256, 16, 16),
                                  (5, 64 + 64, 64, 40, 72)], ids=["ragged_8x32", "bn64_16x32", "ragged_16x16", "one_tile", "two_sources_bn64"])
def test_conv3x3_bf16_lds_dma_form(ops, case):
    """The LDS-DMA form of the producer/consumer kernel (round 4: no prologue, an even number of 32-channel chunks): images that
    are not whole tiles in either direction (zero-page halos AND partial tiles: pixels past the edge are neither stored nor
    counted), both tile shapes, the 512-pixel x 64-channel variant, one unit per workgroup and many, two sources; with the
    BatchNorm statistics row per unit (the consumer waves' LDS exchange) and, separately, a bias -- against fp32 torch on the
    same bf16 inputs.  The canary pixels behind the output must stay untouched (results are stored from registers)."""
    from image_segmentation_amd import _lib
    dtype = torch.bfloat16
    B, Cin, Cout, H, W = case
    x = fill((B, Cin, H, W), 21, -1, 1)
    w = fill((Cout, Cin, 3, 3), 22, -1, 1) / np.sqrt(9 * Cin)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), padding=1)
    if Cin == 128 and Cout == 64:
        y, st = _conv_direct(ops, x[:, :64], w, dtype, xb=x[:, 64:], stats=True)
    else:
        y, st = _conv_direct(ops, x, w, dtype, stats=True)
    assert (y - ref).abs().max().item() < tol(dtype, 1) * 1.5
    s = st.sum(0)[:Cout]
    assert torch.allclose(s[:, 0], ref.sum(dim=(0, 2, 3)), rtol=2e-2, atol=2e-2 * B * H * W * 0.05)
    assert torch.allclose(s[:, 1], (ref * ref).sum(dim=(0, 2, 3)), rtol=2e-2, atol=2e-2)
    # through the raw entry with canary rows behind the output, no statistics: without a bias (the LDS-DMA form, results stored
    # from registers) and with one (a biased conv takes the staged form: the register epilogue carries no bias)
    if Cin != 128 or Cout != 64:
        bias = fill((Cout,), 23, -1, 1)
        xa = ops.to_act(dev(x), dtype); pa, CAp = ops.act_info(xa, dtype)
        wp = ops.pack_conv(dev(w), Cin, 0, dtype, 0)
        for bt in (None, dev(bias)):
            buf = torch.full((B * H * W + 64, Cout), 7.0, dtype=dtype, device="cuda")
            _lib.call("segk_conv3x3", pa, 0, wp.data_ptr(), bt.data_ptr() if bt is not None else 0, 0, 0, buf.data_ptr(), 0, 0,
                      B, H, W, CAp, 0, Cout, 0, 1, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = buf[:B * H * W].view(B, H, W, Cout).permute(0, 3, 1, 2).float().cpu()
            want = ref + bias.view(1, -1, 1, 1) if bt is not None else ref
            assert (got - want).abs().max().item() < tol(dtype, 1) * 1.5
            assert bool((buf[B * H * W:].float() == 7.0).all())


def test_conv3x3_bf16_register_stationary(ops):
    """conv_rs_kernel (Cin <= 64, 64-channel tiles, W > 16): many tiles per workgroup (the three-slot DMA ring wraps
    several times), partial tiles on both edges, one and two channel tiles, two destinations (the concat data gradient
    of an Up block), two 32-channel sources, statistics rows per wave -- against fp32 torch on the same bf16 inputs."""
    from image_segmentation_amd import _lib
    dtype = torch.bfloat16
    cases = [(8, 64, 64, 256, 256, 0), (3, 64, 128, 72, 104, 0), (2, 32, 64, 50, 70, 0), (2, 64, 128, 40, 96, 64),
             (1, 64, 64, 8, 32, 0)]
    for B, Cin, Cout, H, W, split in cases:
        x = fill((B, Cin, H, W), 31, -1, 1)
        w = fill((Cout, Cin, 3, 3), 32, -1, 1) / np.sqrt(9 * Cin)
        ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), padding=1)
        xa = ops.to_act(dev(x), dtype); pa, CAp = ops.act_info(xa, dtype)
        wp = ops.pack_conv(dev(w), Cin, 0, dtype, 0)
        tiles = _lib.query("segk_conv_tiles", B, H, W, CAp, Cout, 1)
        st = torch.full((_lib.query("segk_bn_stats_floats", tiles, Cout),), 7.0, dtype=torch.float32, device="cuda")
        if split:
            o1 = torch.empty((B, H, W, split), dtype=dtype, device="cuda")
            o2 = torch.empty((B, H, W, Cout - split), dtype=dtype, device="cuda")
            ops.conv3x3(xa, pa, CAp, 0, 0, wp, o1.data_ptr(), split, o2.data_ptr(), Cout - split, B, H, W, dtype, stats=st)
            y = torch.cat([back(ops.act_view(o1, split)), back(ops.act_view(o2, Cout - split))], 1)
        else:
            out = torch.empty((B, H, W, Cout), dtype=dtype, device="cuda")
            ops.conv3x3(xa, pa, CAp, 0, 0, wp, out.data_ptr(), Cout, 0, 0, B, H, W, dtype, stats=st)
            y = back(ops.act_view(out, Cout))
        torch.cuda.synchronize()
        assert (y - ref).abs().max().item() < tol(dtype, 1) * 1.5, (B, Cin, Cout, H, W)
        sums = st[:tiles * Cout * 2].view(tiles, Cout, 2).cpu().sum(0)
        refq = y                                   # statistics are taken from the fp32 accumulators: compare loosely
        assert torch.allclose(sums[:, 0], ref.sum(dim=(0, 2, 3)), rtol=2e-2, atol=2e-2 * B * H * W * 0.05), (B, Cin, Cout, H, W)
        assert torch.allclose(sums[:, 1], (ref * ref).sum(dim=(0, 2, 3)), rtol=2e-2, atol=2e-2), (B, Cin, Cout, H, W)
    # two 32-channel sources == conv of the concatenation
    B, H, W = 2, 24, 64
    xa_ = fill((B, 32, H, W), 41, -1, 1); xb_ = fill((B, 32, H, W), 42, -1, 1)
    w2 = fill((64, 64, 3, 3), 43, -1, 1) / 24
    ref = F.conv2d(torch.cat([xa_, xb_], 1).to(dtype).float(), w2.to(dtype).float(), padding=1)
    y = _conv_direct(ops, xa_, w2, dtype, xb=xb_)
    assert (y - ref).abs().max().item() < tol(dtype, 1) * 1.5


@pytest.mark.parametrize("C", [64, 128])
def test_conv3x3_bf16_prologue_side_output(ops, C):
    """segk_conv3x3_act (second conv of a block): conv(relu(z*scale+shift)) plus the hidden activation itself as a
    side output, on an image with partial tiles; C = 64 runs the weight-stationary, 128 the producer/consumer kernel"""
    from image_segmentation_amd import _lib
    dtype = torch.bfloat16
    B, H, W = 2, 40, 72
    z = fill((B, C, H, W), 21, -2, 2); w = fill((C, C, 3, 3), 22, -1, 1) / np.sqrt(9 * C)
    sc = fill((C,), 23, -1.5, 1.5); sh = fill((C,), 24, -0.5, 0.5)
    assert _lib.query("segk_conv_writes_act_q", C, C, 1) == 1
    a_ref = torch.relu(z.to(dtype).float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dtype).float()
    ref = F.conv2d(a_ref, w.to(dtype).float(), padding=1)
    za = ops.to_act(dev(z), dtype); pz, Cp = ops.act_info(za, dtype)
    wp = ops.pack_conv(dev(w), C, 0, dtype, 0)
    out = torch.empty((B, H, W, Cp), dtype=dtype, device="cuda")
    act = torch.full((B, H, W, Cp), 7.0, dtype=dtype, device="cuda")
    scd, shd = dev(sc), dev(sh)
    _lib.call("segk_conv3x3_act", pz, wp.data_ptr(), scd.data_ptr(), shd.data_ptr(), out.data_ptr(), act.data_ptr(), 0,
              B, H, W, Cp, Cp, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (back(ops.act_view(out, C)) - ref).abs().max().item() < tol(dtype, 1) * 2
    # every pixel and channel written exactly once with the bf16-rounded activation (fma vs mul+add: 1 bf16 ulp)
    got = back(ops.act_view(act, C))
    assert (got - a_ref).abs().max().item() <= 2 ** -7 * max(1.0, a_ref.abs().max().item())
    assert ((got - a_ref).abs() > 0).float().mean().item() < 0.02


def test_conv3x3_bf16_concat_and_split_gradient(ops):
    """Up-block shapes on the producer/consumer kernel: two NHWC sources (skip | upsampled) forward, and the data
    gradient written to two destinations (out | out2)"""
    dtype = torch.bfloat16
    B, H, W, CA, CB, Cout = 2, 24, 40, 64, 64, 128
    xa = fill((B, CA, H, W), 31, -1, 1); xb = fill((B, CB, H, W), 32, -1, 1)
    w = fill((Cout, CA + CB, 3, 3), 33, -1, 1) / np.sqrt(9 * (CA + CB))
    ref = F.conv2d(torch.cat([xa, xb], 1).to(dtype).float(), w.to(dtype).float(), padding=1)
    y = _conv_direct(ops, xa, w, dtype, xb=xb)
    assert (y - ref).abs().max().item() < tol(dtype, 1) * 2
    g = fill((B, Cout, H, W), 34, -1, 1)
    refg = F.conv_transpose2d(g.to(dtype).float(), w.to(dtype).float(), padding=1)
    ga = ops.to_act(dev(g), dtype); pg, Gp = ops.act_info(ga, dtype)
    wd = ops.pack_conv(dev(w), CA, CB, dtype, 1)
    o1 = torch.empty((B, H, W, CA), dtype=dtype, device="cuda"); o2 = torch.empty((B, H, W, CB), dtype=dtype, device="cuda")
    ops.conv3x3(ga, pg, Gp, 0, 0, wd, o1.data_ptr(), CA, o2.data_ptr(), CB, B, H, W, dtype)
    torch.cuda.synchronize()
    got = torch.cat([back(ops.act_view(o1, CA)), back(ops.act_view(o2, CB))], 1)
    assert (got - refg).abs().max().item() < tol(dtype, 1) * 2


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_dgrad_weights(ops, dtype):
    """mode-1 packing turns the same kernel into the data gradient"""
    B, Cin, Cout, H, W = 2, 64, 96, 12, 20
    w = fill((Cout, Cin, 3, 3), 2, -1, 1) / np.sqrt(9 * Cout)
    g = fill((B, Cout, H, W), 3, -1, 1)
    ref = F.conv_transpose2d(g.to(dtype).float(), w.to(dtype).float(), padding=1)
    from image_segmentation_amd import _lib
    ga = ops.to_act(dev(g), dtype); pg, Gp = ops.act_info(ga, dtype)
    wd = ops.pack_conv(dev(w), Cin, 0, dtype, 1)
    out = torch.empty((B, H, W, ops.pad32(Cin)), dtype=dtype, device="cuda")
    ops.conv3x3(ga, pg, Gp, 0, 0, wd, out.data_ptr(), ops.pad32(Cin), 0, 0, B, H, W, dtype)
    assert (back(ops.act_view(out, Cin)) - ref).abs().max().item() < tol(dtype, 1) * 2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 32, 32, 16, 16, 0), (1, 64, 128, 24, 40, 0), (2, 3, 64, 20, 20, 0),
                                  (3, 96, 64, 14, 14, 0), (2, 40, 32, 9, 21, 72),
                                  # images that are not whole 8 x 16 tiles on every LDS-DMA configuration (RAG instances): smaller
                                  # than one tile, the 4 x 2-wave workgroup (CD % 128 == 0), and the CLIP decoder's 28 / 56 levels
                                  (2, 32, 32, 5, 7, 0), (1, 128, 128, 28, 28, 0), (1, 64, 128, 56, 56, 64), (2, 64, 64, 8, 17, 0)])
def test_wgrad_conv3x3(ops, dtype, case):
    B, Cin, Cout, H, W, CB = case
    x = fill((B, Cin, H, W), 1, -1, 1); g = fill((B, Cout, H, W), 2, -1, 1)
    xb = fill((B, CB, H, W), 3, -1, 1) if CB else None
    xin = x if xb is None else torch.cat([x, xb], 1)
    xr = xin.to(dtype).float().requires_grad_(False)
    w = torch.zeros((Cout, Cin + CB, 3, 3), requires_grad=True)
    F.conv2d(xr, w, padding=1).backward(g.to(dtype).float())
    ga = ops.to_act(dev(g), dtype); pg, Gp = ops.act_info(ga, dtype)
    xa = ops.to_act(dev(x), dtype); pa, Ap = ops.act_info(xa, dtype)
    pb, Bp = 0, 0
    if xb is not None:
        xbt = ops.to_act(dev(xb), dtype); pb, Bp = ops.act_info(xbt, dtype)
    slabs, S = ops.wgrad(pg, Gp, pa, Ap, pb, Bp, B, H, W, 0, dtype, "cuda")
    dw = ops.wgrad_to_param(slabs, S, w.shape, Cout, Cin, CB, 9, "cuda").cpu()
    scale = np.sqrt(B * H * W)
    assert (dw - w.grad).abs().max().item() < tol(dtype, 1) * scale * 0.5


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_prologue(ops, dtype):
    B, C, H, W = 2, 64, 16, 24
    z = fill((B, C, H, W), 1, -2, 2); g = fill((B, 32, H, W), 2, -1, 1)
    sc = fill((C,), 3, -1.5, 1.5); sh = fill((C,), 4, -0.5, 0.5)
    a = torch.relu(z.to(dtype).float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dtype).float()
    w = torch.zeros((32, C, 3, 3), requires_grad=True)
    F.conv2d(a, w, padding=1).backward(g.to(dtype).float())
    ga = ops.to_act(dev(g), dtype); pg, Gp = ops.act_info(ga, dtype)
    za = ops.to_act(dev(z), dtype); pz, Zp = ops.act_info(za, dtype)
    slabs, S = ops.wgrad(pg, Gp, pz, Zp, 0, 0, B, H, W, 0, dtype, "cuda", scale=dev(sc), shift=dev(sh))
    dw = ops.wgrad_to_param(slabs, S, w.shape, 32, C, 0, 9, "cuda").cpu()
    assert (dw - w.grad).abs().max().item() < tol(dtype, 1) * np.sqrt(B * H * W) * 0.5


@pytest.mark.parametrize("case", [(7, 64), (512, 96), (1024, 32), (1025, 64), (4096, 64), (3000, 1024)])
def test_bn_finalize_from_partial_rows(ops, case):
    """segk_bn_finalize in training mode, both launch forms (one block per channel group up to 1024 partial rows; chunked
    blocks whose last arriver finishes the group above that): scale / shift / mean / rstd and the running statistics from
    per-tile (sum, sum of squares) rows, against float64 arithmetic on the same rows (reference semantics:
    torch.nn.BatchNorm2d at unet/unet.py:17,20 -- biased variance to normalise, unbiased into running_var)."""
    import types
    MT, C = case
    Cr = C - 5 if C > 32 else C
    g = torch.Generator().manual_seed(MT * 131 + C)
    rows = torch.rand((MT, C, 2), generator=g, dtype=torch.float32)
    rows[:, :, 0] = rows[:, :, 0] * 2 - 1                       # sums of either sign
    rows[:, :, 1] = rows[:, :, 1] * 4 + 3.0                     # sums of squares: keeps the variance positive
    count = float(MT * 4)
    gamma, beta, cb = fill((Cr,), 1, 0.5, 1.5), fill((Cr,), 2, -0.5, 0.5), fill((Cr,), 3, -0.2, 0.2)
    rm0, rv0 = fill((Cr,), 4, -0.1, 0.1), fill((Cr,), 5, 0.5, 1.5)
    bn = types.SimpleNamespace(running_mean=dev(rm0.clone()), running_var=dev(rv0.clone()))
    from image_segmentation_amd import _lib
    st = torch.empty((_lib.query("segk_bn_stats_floats", MT, C),), dtype=torch.float32, device="cuda")
    st[:MT * C * 2] = dev(rows).reshape(-1)
    for rep in range(3):                                        # repeated launches re-use the ticket counters
        bn.running_mean.copy_(dev(rm0)); bn.running_var.copy_(dev(rv0))
        sc, sh, mu, rs = ops.bn_finalize(st, MT, Cr, count, dev(cb), dev(gamma), dev(beta), bn.running_mean, bn.running_var,
                                         0.1, 1e-5, True, "cuda")
        torch.cuda.synchronize()
        s = rows.double().sum(0)[:Cr]
        mean = s[:, 0] / count
        var = (s[:, 1] / count - mean * mean).clamp(min=0)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        assert (back(mu)[:Cr].double() - mean).abs().max() < 1e-6
        assert ((back(rs)[:Cr].double() - rstd).abs() / rstd).max() < 1e-6
        assert (back(sc)[:Cr].double() - gamma.double() * rstd).abs().max() < 1e-5
        assert (back(sh)[:Cr].double() - (beta.double() - mean * gamma.double() * rstd)).abs().max() < 1e-5
        assert back(sc)[Cr:].abs().max().item() == 0 if C > Cr else True
        want_rm = 0.9 * rm0.double() + 0.1 * (mean + cb.double())
        want_rv = 0.9 * rv0.double() + 0.1 * var * count / (count - 1)
        assert (back(bn.running_mean).double() - want_rm).abs().max() < 1e-6
        assert (back(bn.running_var).double() - want_rv).abs().max() < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_relu_bwd(ops, dtype):
    B, C, H, W = 3, 40, 10, 14
    z0 = fill((B, C, H, W), 1, -2, 2).to(dtype).float()
    g0 = fill((B, C, H, W), 2, -1, 1).to(dtype).float()
    gamma = fill((C,), 3, 0.5, 1.5); beta = fill((C,), 4, -0.3, 0.3)
    z = z0.clone().requires_grad_(True)
    gam = gamma.clone().requires_grad_(True); bet = beta.clone().requires_grad_(True)
    y = torch.relu(F.batch_norm(z, None, None, gam, bet, True, 0.1, 1e-5))
    y.backward(g0)
    mean = z0.mean(dim=(0, 2, 3)); var = z0.var(dim=(0, 2, 3), unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    Cp = ops.pad32(C)
    pad = lambda v: dev(torch.cat([v, torch.zeros(Cp - C)]))
    scale = pad(gamma * rstd); shift = pad(beta - mean * gamma * rstd)
    za = ops.to_act(dev(z0), dtype); ga = ops.to_act(dev(g0), dtype)
    out = torch.empty((B, H, W, Cp), dtype=dtype, device="cuda")
    dg, db = ops.bn_relu_bwd(ops.act_info(ga, dtype)[0], ops.act_info(za, dtype)[0], out.data_ptr(), scale, shift,
                             pad(mean), pad(rstd), B * H * W, C, dtype, "cuda")
    t = 1e-4 if dtype == torch.float32 else 3e-2
    assert torch.allclose(dg.cpu(), gam.grad, rtol=t, atol=t * 5)
    assert torch.allclose(db.cpu(), bet.grad, rtol=t, atol=t * 5)
    assert (back(ops.act_view(out, C)) - z.grad).abs().max().item() < t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 32, 8, 8), (1, 70, 6, 10), (2, 64, 7, 9)])
def test_maxpool(ops, dtype, shape):
    x0 = fill(shape, 1, -1, 1).to(dtype).float()
    x0[0, :, 0, 0] = x0[0, :, 0, 1]            # a tie: gradient must go to the FIRST maximum
    x = x0.clone().requires_grad_(True)
    y = F.max_pool2d(x, 2, 2)
    g0 = fill(tuple(y.shape), 2, -1, 1).to(dtype).float()
    y.backward(g0)
    xa = ops.to_act(dev(x0), dtype).requires_grad_(True)
    ya = ops.MaxPoolFn.apply(xa, dtype)
    assert torch.equal(back(ya), y.detach())
    ya.backward(ops.to_act(dev(g0), dtype))
    assert torch.equal(back(xa.grad), x.grad)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [64, 280, 512])
def test_output_head_any_width(ops, dtype, C):
    """Output Conv2d(C, classes, 1) forward + backward (reference unet/unet.py:91, clip/clipunet.py:181: nn.Conv2d accepts any
    width; UNetDecoder(decoder_channels=[..., C]) feeds it C channels).  The backward kernel owns 4 channels per thread
    and 64 channel vectors per block row: above 256 padded channels it runs channel slices (blockIdx.y) -- refused with -2
    until round 4."""
    import types
    B, H, W, ncls = 2, 24, 40, 3
    x = fill((B, C, H, W), 5, -1, 1); w = fill((ncls, C, 1, 1), 6, -0.2, 0.2); b = fill((ncls,), 7, -0.1, 0.1)
    gl = fill((B, ncls, H, W), 8, -1, 1)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    if dtype == torch.bfloat16:
        xq = xr.to(dtype).float()
    else:
        xq = xr
    ref = torch.nn.functional.conv2d(xq, wr, br)
    ref.backward(gl)
    xd = dev(x).to(memory_format=torch.channels_last).requires_grad_(True)
    wd = dev(w).requires_grad_(True); bd = dev(b).requires_grad_(True)
    lg = ops.HeadFn.apply(types.SimpleNamespace(compute_dtype=dtype), xd, wd, bd)
    lg.backward(dev(gl))
    torch.cuda.synchronize()
    t = 2e-5 if dtype == torch.float32 else 3e-2
    assert (back(lg) - ref.detach()).abs().max().item() < t * (C ** 0.5) * 0.2
    assert (back(xd.grad).float() - xr.grad).abs().max().item() < (1e-5 if dtype == torch.float32 else 8e-3)
    assert (back(wd.grad) - wr.grad).abs().max().item() < t * (B * H * W) ** 0.5 * 0.2
    assert (back(bd.grad) - br.grad).abs().max().item() < 1e-3


@pytest.mark.parametrize("pattern", [0x00000005, 0xffffffff, 0x00000001, 0x7fffffff],
                         ids=["count5", "all_ones", "count1", "large"])
def test_ticket_ring_survives_poisoned_slots(ops, pattern):
    """The in-launch finishers (segk_bn_finalize above 1024 partial rows, segk_loss_fwd) elect the last-arriving block through a
    ticket word of a process-wide ring (csrc/ticket.hpp).  Rounds 2-3 relied on every word being zero when a launch drew
    it -- a launch that aborted left its counters behind and a later launch on that slot never elected a finisher
    (scale / shift stayed uninitialised, silently).  The host now zeroes the counters a launch uses on its stream right before
    the launch.  Here EVERY counter of the ring is overwritten (segk_debug_poison_tickets) with what an aborted launch or a
    stray store would leave, and both kernels must still produce the reference values, repeatedly."""
    from image_segmentation_amd import _lib, losses
    MT, C = 3000, 128
    g = torch.Generator().manual_seed(77)
    rows = torch.rand((MT, C, 2), generator=g, dtype=torch.float32)
    rows[:, :, 1] = rows[:, :, 1] * 4 + 3.0
    count = float(MT * 4)
    gamma, beta = fill((C,), 1, 0.5, 1.5), fill((C,), 2, -0.5, 0.5)
    st = torch.empty((_lib.query("segk_bn_stats_floats", MT, C),), dtype=torch.float32, device="cuda")
    s = rows.double().sum(0)
    mean = s[:, 0] / count
    rstd = 1.0 / torch.sqrt((s[:, 1] / count - mean * mean).clamp(min=0) + 1e-5)
    lg0 = fill((4, 3, 64, 96), 41, -3, 3); Y = fill_labels((4, 64, 96), 42, 3)
    want_loss = torch.nn.functional.cross_entropy(lg0, Y).item()
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(3):
        _lib.call("segk_debug_poison_tickets", (pattern + rep) & 0xffffffff, stream)
        st[:MT * C * 2] = dev(rows).reshape(-1)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        sc, sh, mu, rs = ops.bn_finalize(st, MT, C, count, None, dev(gamma), dev(beta), rm, rv, 0.1, 1e-5, True, "cuda")
        torch.cuda.synchronize()
        assert (back(mu).double() - mean).abs().max() < 1e-6
        assert (back(sc).double() - gamma.double() * rstd).abs().max() < 1e-5
        assert (back(sh).double() - (beta.double() - mean * gamma.double() * rstd)).abs().max() < 1e-5
        for _ in range(3):                                       # several draws: slots straight after the poisoning and later ones
            v = losses.CrossEntropyLoss()(dev(lg0), dev(Y))
            assert abs(v.item() - want_loss) < 5e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_loss_kernels(ops, dtype, golden):
    if dtype != torch.float32:
        pytest.skip("loss kernels are fp32 only")
    from image_segmentation_amd import losses
    g = golden("losses_small")
    lg0 = fill((2, 4, 12, 20), 41, -3, 3); Y = fill_labels((2, 12, 20), 42, 4)
    w4 = torch.tensor([0.3, 1.1, 0.9, 1.7])
    cases = {
        "ce": (losses.CrossEntropyLoss(), Y),
        "ce_w": (losses.CrossEntropyLoss(weight=w4), Y),
        "ce_w_ign3": (losses.CrossEntropyLoss(weight=w4, ignore_index=3), Y),
        "dice": (losses.WeightedMemoryEfficientDiceLoss(smooth=1e-5), Y.unsqueeze(1)),
        "dice_w_ign3": (losses.WeightedMemoryEfficientDiceLoss(smooth=1.0, class_weights=w4, ignore_index=3), Y.unsqueeze(1)),
        "dicece": (losses.WeightedDiceCELoss(), Y),
        "dicece_w_ign3": (losses.WeightedDiceCELoss(dice_weight=0.7, ce_weight=1.3, ignore_index=3, class_weights=w4,
                                                    smooth_dice=1.0), Y.unsqueeze(1)),
    }
    for k, (fn, y) in cases.items():
        l = dev(lg0).requires_grad_(True)
        v = fn(l, dev(y)); v.backward()
        assert abs(v.item() - float(g[k])) < 5e-6, k
        np.testing.assert_allclose(l.grad.cpu().numpy(), g[k + "_grad"], rtol=2e-4, atol=2e-8, err_msg=k)
    with pytest.raises(ValueError):
        losses.WeightedMemoryEfficientDiceLoss()(dev(lg0), dev(Y))


def test_confusion(ops):
    lg = fill((3, 4, 9, 11), 5, -1, 1); lg[0, 1, 0, 0] = lg[0, 2, 0, 0] = 5.0   # tie -> lowest index
    Y = fill_labels((3, 9, 11), 6, 4)
    M = ops.confusion_matrix(dev(lg), dev(Y), 4).cpu()
    hard = lg.argmax(1)
    ref = torch.zeros(4, 4, dtype=torch.int64)
    for p in range(4):
        for l in range(4):
            ref[p, l] = ((hard == p) & (Y == l)).sum()
    assert torch.equal(M, ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 128, 64, 16, 16), (1, 64, 32, 8, 8), (2, 256, 128, 12, 20), (1, 96, 64, 16, 16),
                                  (3, 64, 128, 10, 14), (1, 256, 128, 8, 32), (2, 128, 64, 5, 48), (1, 128, 32, 4, 16),
                                  (1, 256, 64, 3, 16), (3, 128, 128, 7, 32)])
def test_convt2x2_forward_and_data_gradient(ops, dtype, case):
    """ConvTranspose2d(k=2,s=2) as GEMM + pixel-shuffle store and its data gradient as an un-shuffle gather GEMM
    (reference unet/unet.py:59): the long-K shapes run the producer/consumer GEMM (gemm.hip), bf16 with Cin 128 / 256 and
    whole 16-pixel blocks per row the register-stationary streaming kernel (convt_stream.hip: all of its wave layouts are
    among the cases), the rest the generic kernel; all against F.conv_transpose2d / its autograd."""
    from image_segmentation_amd import _lib
    B, Cin, Cout, H, W = case
    x0 = fill((B, Cin, H, W), 1, -1, 1).to(dtype).float()
    w0 = (fill((Cin, Cout, 2, 2), 2, -1, 1) / Cin ** 0.5).to(dtype).float()
    b0 = fill((Cout,), 3, -0.5, 0.5)
    g0 = fill((B, Cout, 2 * H, 2 * W), 4, -1, 1).to(dtype).float()
    x = x0.clone().requires_grad_(True)
    y = F.conv_transpose2d(x, w0, b0, stride=2)
    y.backward(g0)
    s = torch.cuda.current_stream().cuda_stream
    xa = ops.to_act(dev(x0), dtype)
    px, _ = ops.act_info(xa, dtype)
    wf, wd = ops.pack_convt(dev(w0), dtype, 0), ops.pack_convt(dev(w0), dtype, 1)
    b4 = dev(b0).repeat(4).contiguous()
    out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=dtype, device="cuda")
    _lib.call("segk_convt2x2_fwd", px, wf.data_ptr(), b4.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, ops._DT[dtype], s)
    got = back(out.permute(0, 3, 1, 2))
    assert (got - y.detach()).abs().max() < tol(dtype, Cin)
    ga = ops.to_act(dev(g0), dtype)
    pg, _ = ops.act_info(ga, dtype)
    dx = torch.empty((B, H, W, Cin), dtype=dtype, device="cuda")
    _lib.call("segk_convt2x2_dgrad", pg, wd.data_ptr(), dx.data_ptr(), B, H, W, Cin, Cout, ops._DT[dtype], s)
    assert (back(dx.permute(0, 3, 1, 2)) - x.grad).abs().max() < tol(dtype, 4 * Cout)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 128, 64, 16, 16), (1, 256, 128, 6, 20), (3, 128, 64, 9, 7), (2, 64, 32, 8, 8),
                                  (1, 96, 64, 5, 18), (2, 64, 128, 3, 33)])
def test_convt2x2_weight_gradient(ops, dtype, case):
    """ConvTranspose2d(k=2,s=2) weight gradient (wgrad geometry 2, reference unet/unet.py:59) against autograd: the
    128 x 64 workgroup tiles with 8-row pixel tiles (bf16, Cin % 128 == 0, Cout % 64 == 0) and the smaller configurations,
    heights and widths that are not multiples of the tile (rows past the image contribute zeros)."""
    B, Cin, Cout, H, W = case
    x0 = fill((B, Cin, H, W), 1, -1, 1).to(dtype).float()
    w0 = (fill((Cin, Cout, 2, 2), 2, -1, 1) / Cin ** 0.5)
    g0 = fill((B, Cout, 2 * H, 2 * W), 4, -1, 1).to(dtype).float()
    w = w0.clone().requires_grad_(True)
    F.conv_transpose2d(x0, w, None, stride=2).backward(g0)
    xa = ops.to_act(dev(x0), dtype); px, Cinp = ops.act_info(xa, dtype)
    ga = ops.to_act(dev(g0), dtype); pg, Coutp = ops.act_info(ga, dtype)
    slabs, S = ops.wgrad(px, Cinp, pg, Coutp, 0, 0, B, H, W, 2, dtype, "cuda")
    dw = ops.wgrad_to_param(slabs, S, w0.shape, Cin, Cout, 0, 4, "cuda").cpu()
    # the operands are exactly representable in `dtype` and accumulate in fp32: only the summation order differs
    assert (dw - w.grad).abs().max() <= 1e-4 * np.sqrt(B * H * W), case


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(64, 3, 0), (8, 3, 0), (64, 64, 0), (128, 64, 64), (70, 40, 0), (32, 20, 50)])
def test_pack_both_layouts_matches_single_packs(ops, dtype, case):
    """The one-pass forward + data-gradient pack (segk_pack_conv3x3_both) against the per-mode pack, bit for bit."""
    Cout, CA, CB = case
    w = dev(fill((Cout, CA + CB, 3, 3), 1, -1, 1))
    f, d = ops.pack_conv_both(w, CA, CB, dtype)
    assert torch.equal(f, ops.pack_conv(w, CA, CB, dtype, 0))
    assert torch.equal(d, ops.pack_conv(w, CA, CB, dtype, 1))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 64, 16, 24, 1), (1, 128, 9, 14, 1), (2, 32, 8, 8, 0), (1, 1024, 4, 6, 1)])
@pytest.mark.parametrize("degenerate", [False, True])
def test_maxpool_bwd_with_bn_reductions(ops, dtype, case, degenerate):
    """segk_maxpool2x2_bwd_bnstat: same dx as the plain pooling backward, and partials whose finalize + apply
    (segk_bn_relu_bwd_from_part) reproduce the three-kernel BatchNorm backward on that dx.  degenerate: channels with
    gamma == 0 and with |gamma| = 1e-3 |beta|, where xhat cannot be recovered from the block output: the kernel takes it
    from z for the threads that own such a channel."""
    from image_segmentation_amd import _lib
    B, C, H, W, acc = case
    s = torch.cuda.current_stream().cuda_stream
    z = fill((B, H, W, C), 1, -2, 2)
    gamma, beta = fill((C,), 2, 0.5, 1.5), fill((C,), 3, -0.5, 0.5)
    if degenerate:
        gamma[1] = 0.0; beta[1] = 0.4                    # y = relu(beta) everywhere
        gamma[C - 3] = 0.0; beta[C - 3] = -0.2            # y = 0 everywhere
        gamma[10] = 1e-3 * beta[10].abs().clamp(min=0.1); beta[10] = beta[10].abs().clamp(min=0.1)
    mu, var = z.reshape(-1, C).mean(0), z.reshape(-1, C).var(0, unbiased=False)
    rs = 1.0 / torch.sqrt(var + 1e-5)
    sc, sh = gamma * rs, beta - mu * gamma * rs
    zq = z.to(dtype)
    y = torch.relu(zq.float() * sc + sh).to(dtype)
    dyp = fill((B, H // 2, W // 2, C), 4, -1, 1).to(dtype)
    dskip = fill((B, H, W, C), 5, -1, 1).to(dtype)
    dt = ops._DT[dtype]
    yd, dypd, zd = dev(y), dev(dyp), dev(zq)
    scd, shd, mud, rsd = dev(sc), dev(sh), dev(mu), dev(rs)
    dx_ref = dev(dskip.clone()) if acc else torch.empty((B, H, W, C), dtype=dtype, device="cuda")
    dx_new = dx_ref.clone()
    _lib.call("segk_maxpool2x2_bwd", yd.data_ptr(), dypd.data_ptr(), dx_ref.data_ptr(), B, H, W, C, acc, dt, s)
    nb = _lib.query("segk_maxpool_bwd_stat_blocks", B, H, W, C, dt)
    assert nb > 0
    part = torch.empty(nb * C * 2, device="cuda")
    _lib.call("segk_maxpool2x2_bwd_bnstat", yd.data_ptr(), dypd.data_ptr(), dx_new.data_ptr(), B, H, W, C, acc, scd.data_ptr(),
              shd.data_ptr(), mud.data_ptr(), rsd.data_ptr(), part.data_ptr(), zd.data_ptr(), dt, s)
    assert torch.equal(dx_new, dx_ref)
    P = B * H * W
    dz_a, dz_b = torch.empty_like(dx_ref), torch.empty_like(dx_ref)
    dg_a, db_a = ops.bn_relu_bwd(dx_ref.data_ptr(), zd.data_ptr(), dz_a.data_ptr(), scd, shd, mud, rsd, P, C, dtype, "cuda")
    dg_b, db_b = ops.bn_relu_bwd(dx_ref.data_ptr(), zd.data_ptr(), dz_b.data_ptr(), scd, shd, mud, rsd, P, C, dtype, "cuda",
                                 ready=(part, nb))
    t = 1e-4 if dtype == torch.float32 else 3e-2
    scale = max(1.0, float(db_a.abs().max()), float(dg_a.abs().max()))
    assert (back(db_a) - back(db_b)).abs().max() < t * scale
    assert (back(dg_a) - back(dg_b)).abs().max() < t * scale
    assert (back(dz_a) - back(dz_b)).abs().max() < (1e-4 if dtype == torch.float32 else 2e-2)
    assert _lib.query("segk_maxpool_bwd_stat_blocks", 1, 8, 8, 96, dt) == 0       # 12 or 24 channel vectors: not served


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 64, 16, 24), (1, 96, 9, 14), (2, 32, 7, 5)])
def test_bn_relu_apply_with_pool_is_bit_identical(ops, dtype, case):
    from image_segmentation_amd import _lib
    B, C, H, W = case
    s = torch.cuda.current_stream().cuda_stream
    z = dev(fill((B, H, W, C), 1, -2, 2).to(dtype))
    sc, sh = dev(fill((C,), 2, 0.5, 1.5)), dev(fill((C,), 3, -0.5, 0.5))
    dt = ops._DT[dtype]
    y0 = torch.empty_like(z); p0 = torch.empty((B, H // 2, W // 2, C), dtype=dtype, device="cuda")
    _lib.call("segk_bn_relu_apply", z.data_ptr(), y0.data_ptr(), sc.data_ptr(), sh.data_ptr(), B * H * W, C, dt, s)
    _lib.call("segk_maxpool2x2_fwd", y0.data_ptr(), p0.data_ptr(), B, H, W, C, dt, s)
    y1 = torch.empty_like(z); p1 = torch.empty_like(p0)
    _lib.call("segk_bn_relu_apply_pool", z.data_ptr(), y1.data_ptr(), p1.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, H, W, C, dt, s)
    assert torch.equal(y0, y1) and torch.equal(p0, p1)


@pytest.mark.parametrize("case", [(2, 3, 16, 32), (1, 3, 9, 16), (3, 1, 5, 48), (2, 2, 33, 16), (1, 3, 64, 64)])
def test_stem_conv_on_nchw_input(ops, case):
    """segk_stem3x3 (bf16): Conv2d(Cin <= 3, 64, 3, padding=1) straight from the NCHW fp32 batch (reference unet/unet.py:16
    on the tensor of utils/training.py:45) against F.conv2d on the bf16-rounded operands; the BatchNorm partial rows sum to
    the statistics of the fp32 result; the padded NHWC side output equals segk_nchw_to_nhwc's bit for bit."""
    from image_segmentation_amd import _lib
    dtype = torch.bfloat16
    B, Cin, H, W = case
    x = fill((B, Cin, H, W), 1, -1, 1); w = fill((64, Cin, 3, 3), 2, -1, 1) / np.sqrt(9 * Cin)
    rows = _lib.query("segk_stem3x3_rows", B, H, W, Cin, 64, ops._DT[dtype])
    assert rows > 0
    assert _lib.query("segk_stem3x3_rows", B, H, W + 1, Cin, 64, ops._DT[dtype]) == 0
    assert _lib.query("segk_stem3x3_rows", B, H, W, 4, 64, ops._DT[dtype]) == 0
    xd, wd = dev(x), dev(w)
    z = torch.empty((B, H, W, 64), dtype=dtype, device="cuda")
    xn = torch.full((B, H, W, 32), 7.0, dtype=dtype, device="cuda")
    st = torch.full((_lib.query("segk_bn_stats_floats", rows, 64),), float("nan"), dtype=torch.float32, device="cuda")
    _lib.call("segk_stem3x3", xd.data_ptr(), wd.data_ptr(), z.data_ptr(), xn.data_ptr(), st.data_ptr(), B, H, W, Cin, 64,
              ops._DT[dtype], torch.cuda.current_stream().cuda_stream)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), padding=1)
    got = back(z.permute(0, 3, 1, 2))
    assert (got - ref).abs().max().item() < tol(dtype, 1) * 2
    want = ops.to_act(xd, dtype)                                   # segk_nchw_to_nhwc: logical [B,Cin,H,W] view of NHWC
    assert ops.act_info(want, dtype)[1] == 32
    assert torch.equal(xn[..., :Cin], want.permute(0, 2, 3, 1))
    assert xn[..., Cin:].abs().max().item() == 0
    s = st[:rows * 64 * 2].view(rows, 64, 2).double().sum(0).cpu()
    assert torch.allclose(s[:, 0], ref.double().sum(dim=(0, 2, 3)), rtol=2e-4, atol=2e-3 * np.sqrt(B * H * W))
    assert torch.allclose(s[:, 1], (ref.double() ** 2).sum(dim=(0, 2, 3)), rtol=2e-4, atol=1e-3)


@pytest.mark.parametrize("case", [(2, 3, 16, 32), (1, 3, 9, 16), (3, 1, 5, 48), (2, 2, 33, 16), (1, 3, 64, 64)])
def test_stem_weight_gradient_from_nchw_input(ops, case):
    """segk_stem3x3_wgrad (bf16): the stem's weight gradient from the NCHW fp32 batch and dz (no padded NHWC copy of the
    input): slabs of [64][32] in OIHW column order, summed by segk_wgrad_reduce(taps = 1, CA = 9 Cin), against autograd on
    the bf16-rounded operands (exact products, fp32 accumulation: only the summation order differs)."""
    from image_segmentation_amd import _lib
    dtype = torch.bfloat16
    B, Cin, H, W = case
    x = fill((B, Cin, H, W), 1, -1, 1); g = fill((B, 64, H, W), 2, -1, 1).to(dtype).float()
    w = torch.zeros((64, Cin, 3, 3), requires_grad=True)
    F.conv2d(x.to(dtype).float(), w, padding=1).backward(g)
    S = _lib.query("segk_stem3x3_wgrad_slabs", B, H, W, Cin, 64, ops._DT[dtype])
    assert S > 0
    ga = ops.to_act(dev(g), dtype); pg, Gp = ops.act_info(ga, dtype)
    assert Gp == 64
    xd = dev(x)
    slabs = torch.full((S * 64 * 32,), float("nan"), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("segk_stem3x3_wgrad", xd.data_ptr(), pg, slabs.data_ptr(), B, H, W, Cin, 64, ops._DT[dtype], s)
    grad = torch.empty((64, Cin, 3, 3), dtype=torch.float32, device="cuda")
    _lib.call("segk_wgrad_reduce", slabs.data_ptr(), S, grad.data_ptr(), 64, 9 * Cin, 0, 64, 32, 0, 1, s)
    assert (back(grad) - w.grad).abs().max() <= 1e-4 * np.sqrt(B * H * W), case
    assert torch.isfinite(slabs).all()


def test_biased_conv_on_rs_shape_falls_through_to_ws_and_refuses_statistics(ops):
    """The round-2 GPU memory fault: on the shapes the register-stationary kernel serves (bf16, Cin 32/64,
    N % 64 == 0, W > 16) a BIASED conv falls through to the weight-stationary kernel, whose BatchNorm partial rows are per
    tile while segk_conv_tiles() sizes the statistics buffer for conv_rs's per-wave rows.  (i) the fall-through itself
    (bias, no statistics: reference autoencoder/autoencoder.py:188-191 is such a layer) computes conv + bias;
    (ii) bias together with statistics is refused with -2 instead of overrunning the buffer."""
    from image_segmentation_amd import _lib
    dtype = torch.bfloat16
    B, C, H, W = 2, 64, 24, 48
    x = fill((B, C, H, W), 1, -1, 1); w = fill((C, C, 3, 3), 2, -1, 1) / 24; b = fill((C,), 3, -1, 1)
    per_tile_rows = B * ((H + 7) // 8) * ((W + 31) // 32)                         # what the weight-stationary kernel would write
    assert _lib.query("segk_conv_tiles", B, H, W, C, C, 1) % 32 == 0 and per_tile_rows == 12   # sized for conv_rs (8 x GW x 4 rows)
    xa = ops.to_act(dev(x), dtype); pa, Cp = ops.act_info(xa, dtype)
    wp = ops.pack_conv(dev(w), C, 0, dtype, 0)
    bp = dev(b)
    out = torch.empty((B, H, W, Cp), dtype=dtype, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("segk_conv3x3", pa, 0, wp.data_ptr(), bp.data_ptr(), 0, 0, out.data_ptr(), 0, 0, B, H, W, Cp, 0, Cp, 0, 1, s)
    torch.cuda.synchronize()
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), b, padding=1)
    assert (back(ops.act_view(out, C)) - ref).abs().max().item() < tol(dtype, 1) * 1.5
    tiles = _lib.query("segk_conv_tiles", B, H, W, Cp, Cp, 1)
    st = torch.zeros((_lib.query("segk_bn_stats_floats", tiles, Cp),), dtype=torch.float32, device="cuda")
    rc = _lib.load().segk_conv3x3(pa, 0, wp.data_ptr(), bp.data_ptr(), 0, 0, out.data_ptr(), 0, st.data_ptr(), B, H, W,
                                  Cp, 0, Cp, 0, 1, s)
    assert rc == -2 and b"bias together with BatchNorm statistics" in _lib.load().segk_last_error()
    torch.cuda.synchronize()
    assert float(st.abs().sum()) == 0.0                                           # nothing was launched




def test_wgrad_reduce_multi_equals_single_launches(ops):
    """segk_wgrad_reduce_multi: the slab reductions of a block's backward (row form S <= 16, slab-parallel form S > 16, a
    two-source 3x3 weight, a 4-tap ConvTranspose weight) and the ConvTranspose bias gradient as column sums of BatchNorm-style
    partial rows, four jobs in ONE launch: bit-identical to the single-job launches; the column sums equal torch's."""
    from image_segmentation_amd import _lib
    s = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)

    def weight_job(S, N, CA, CB, taps):
        Np, CAp, CBp = ops.pad32(N), ops.pad32(CA), (ops.pad32(CB) if CB else 0)
        slabs = torch.randn((S, Np, taps, CAp + CBp), device="cuda", generator=gen)
        single = torch.empty((N, CA + CB, taps), device="cuda")
        _lib.call("segk_wgrad_reduce", slabs.data_ptr(), S, single.data_ptr(), N, CA, CB, Np, CAp, CBp, taps, s)
        multi = torch.full_like(single, 7.0)
        return slabs, single, multi, _lib.ReduceJob(slabs.data_ptr(), multi.data_ptr(), 0, S, N, CA, CB, Np, CAp, CBp, taps, 0)
    jobs = [weight_job(8, 128, 128, 0, 9), weight_job(64, 64, 40, 72, 9), weight_job(256, 64, 64, 0, 4)]
    rows, ntot, col0, ncols = 300, 192, 64, 100
    part = torch.randn((rows, ntot, 2), device="cuda", generator=gen)
    colsum = torch.empty((ncols,), device="cuda")
    arr = (_lib.ReduceJob * 4)(jobs[0][3], jobs[1][3], jobs[2][3],
                               _lib.ReduceJob(part.data_ptr(), colsum.data_ptr(), 1, rows, ntot, col0, ncols, 0, 0, 0, 0, 0))
    for _ in range(2):
        _lib.call("segk_wgrad_reduce_multi", arr, 4, s)
    torch.cuda.synchronize()
    for slabs, single, multi, _ in jobs:
        assert torch.equal(single, multi)
    want = part[:, col0:col0 + ncols, 0].double().sum(0)
    assert (colsum.double() - want).abs().max().item() < 1e-4
    with pytest.raises(RuntimeError, match="1..4 jobs"):
        _lib.call("segk_wgrad_reduce_multi", arr, 5, s)
