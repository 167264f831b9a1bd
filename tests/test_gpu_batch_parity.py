"""GPU (-m gpu): parity AT THE BATCH the bench numbers are taken on (BASELINE configs 2, 4 and 5): forward + loss in both
compute modes, and (round 4) the fp32 BACKWARD of configs 2 and 5 against the oracle's backward on the host cores
(reference utils/training.py:45-53), per parameter:

  config 2  U-Net 3-class, B=32, 3x256x256, CrossEntropy       (reference unet/unet.py:93-105, utils/training.py:47)
  config 5  U-Net 3-class, B=8, 3x512x512, weighted Dice + CE   (utils/weighted_loss.py:140-166)
  config 4  CLIP-UNet decoder + head, B=16, 224x224, fed the ORACLE's encoder features (clip/clipunet.py:139-144,184-188)

These geometries reach what the B=1 / B=2 / B=4 tests cannot: persistent-unit ranges per XCD over thousands of units,
BatchNorm partial rows beyond 1024 inside a model, 32-bit offsets into 537 MB tensors.
fp32 mode: logits within 1e-3 of the oracle and argmax masks bit-exact (the north-star gate); bf16 mode: the gates of
tests/test_gpu_fullsize.py.  Plus a bf16 GRADIENT test at a conditioned batch (B=8, 128x128), where batch statistics rest
on >= 512 values per channel at every level and the per-tensor gates can be tight."""
import numpy as np
import pytest
import torch

from oracle.fill import fill, labels, fill_module
from oracle import unet_ref, losses_ref, clipunet_ref, clip_vit_ref

pytestmark = pytest.mark.gpu
CW3 = [0.2046795970925636, 1.0271954434416883, 1.2293222812780409]


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import image_segmentation_amd as s
    yield s
    s.set_compute_dtype(torch.bfloat16)


_ORACLE = {}


def unet_oracle(B, S, loss_name):
    """CPU oracle: training-mode forward (batch statistics) + loss, no backward (cached for both compute modes)."""
    key = (B, S, loss_name)
    if key not in _ORACLE:
        ref = unet_ref.unet(3, 3); fill_module(ref, 1000); ref.train()
        X = fill((B, 3, S, S), 1, 0, 1); Y = labels((B, S, S), 2, 3)
        with torch.no_grad():
            lr = ref(X)
            if loss_name == "ce":
                loss = losses_ref.cross_entropy(lr, Y)
            else:
                loss = losses_ref.dice_ce(lr, Y, class_weights=torch.tensor(CW3), smooth_dice=1.0)
        _ORACLE[key] = (X, Y, lr, float(loss))
    return _ORACLE[key]


def unet_hip(seg, dtype, B, S, loss_name):
    X, Y, _, _ = unet_oracle(B, S, loss_name)
    seg.set_compute_dtype(dtype)
    m = seg.unet(3, 3); fill_module(m, 1000); m.cuda().train()
    # autograd stays ON (no backward is run): the training forward then takes exactly the kernels of the bench step
    # (both weight layouts, the hidden-activation side output of the second conv of every block)
    lg = m(X.cuda())
    if loss_name == "ce":
        loss = seg.CrossEntropyLoss()(lg, Y.cuda())
    else:
        loss = seg.WeightedDiceCELoss(smooth_dice=1.0, class_weights=torch.tensor(CW3))(lg, Y.cuda())
    torch.cuda.synchronize()
    return lg.detach().float().cpu(), float(loss.detach())


def check_fp32(lg, loss, lr, loss_ref):
    dmax = (lg - lr).abs().max().item()
    assert dmax < 1e-3                                         # north-star gate: logits within 1e-3
    # argmax masks bit-exact -- wherever that is defined: with millions of pixels a few have their two best REFERENCE
    # logits closer together than the fp32 summation-order noise between any two implementations (the oracle itself
    # moves by that much between thread counts), and only there may the masks differ.  Every disagreeing pixel must be
    # such a tie (gap below twice the measured logit difference, itself < 1e-3), and there may be at most 1 in 20 000.
    same = lg.argmax(1) == lr.argmax(1)
    top2 = lr.topk(2, dim=1).values
    tie = (top2[:, 0] - top2[:, 1]) <= 2.0 * dmax
    n_bad = int((~same).sum())
    print(f"fp32 mode: max |dlogit| {dmax:.2e}, {n_bad} of {same.numel()} argmax pixels differ, all inside reference ties: "
          f"{bool((same | tie).all())} ({int(tie.sum())} tie pixels)")
    assert bool((same | tie).all())
    # the STRICT count is a tracked number (0 of 2 097 152 at both configurations when this bound was set): at most 4
    # pixels, far inside the 1-in-20 000 that the tie argument alone would allow
    assert n_bad <= 4, n_bad
    assert abs(loss - loss_ref) < 2e-5


def check_bf16(lg, loss, lr, loss_ref):
    d = (lg - lr).abs()
    assert d.max().item() < 0.2 and d.mean().item() < 0.02, (d.max().item(), d.mean().item())
    assert (lg.argmax(1) == lr.argmax(1)).float().mean().item() > 0.975
    assert abs(loss - loss_ref) < 2e-2


@pytest.mark.timeout(900)
@pytest.mark.parametrize("cfg", [(32, 256, "ce"), (8, 512, "dicece")], ids=["config2_B32_256", "config5_B8_512_dicece"])
def test_unet_forward_at_bench_batch_fp32(seg, cfg):
    B, S, loss_name = cfg
    _, _, lr, loss_ref = unet_oracle(B, S, loss_name)
    lg, loss = unet_hip(seg, torch.float32, B, S, loss_name)
    check_fp32(lg, loss, lr, loss_ref)


_ORACLE_BWD = {}


def oracle_backward(B, S, loss_name):
    """CPU oracle forward + loss + backward at the bench batch, in float64 (the yardstick: the same graph in double precision)
    and float32 (the oracle proper: what the reference computes); cached for the fp32 and bf16 legs."""
    key = (B, S, loss_name)
    if key not in _ORACLE_BWD:
        X = fill((B, 3, S, S), 1, 0, 1); Y = labels((B, S, S), 2, 3)

        def run(dt):
            ref = unet_ref.unet(3, 3); fill_module(ref, 1000); ref.train(); ref.to(dt)
            lr = ref(X.to(dt))
            if loss_name == "ce":
                loss = losses_ref.cross_entropy(lr, Y)
            else:
                loss = losses_ref.dice_ce(lr, Y, class_weights=torch.tensor(CW3, dtype=dt), smooth_dice=1.0)
            loss.backward()
            return {n: p.grad.detach().double() for n, p in ref.named_parameters()}, lr.detach(), float(loss.detach())
        g64, _, _ = run(torch.float64)
        g32, lr, loss_ref = run(torch.float32)
        _ORACLE_BWD[key] = (X, Y, g64, g32, lr, loss_ref)
    return _ORACLE_BWD[key]


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("cfg", [(32, 256, "ce"), (8, 512, "dicece")], ids=["config2_B32_256", "config5_B8_512_dicece"])
def test_unet_backward_at_bench_batch_bf16(seg, cfg):
    """The bf16 GRADIENTS of the step the headline number is taken on (config 2: B=32, 256x256; config 5: B=8, 512x512), per
    parameter against the float64 evaluation of the oracle graph.  With 2 M pixels behind every reduction the gates can be
    much tighter than at one image (tests/test_gpu_fullsize.py allows cosine 0.75 / norms within 25 % there): a sign flip, a
    permutation, or a 10 % scale error of ANY tensor -- BatchNorm vectors included -- fails."""
    B, S, loss_name = cfg
    X, Y, g64, g32, lr, loss_ref = oracle_backward(B, S, loss_name)
    seg.set_compute_dtype(torch.bfloat16)
    m = seg.unet(3, 3); fill_module(m, 1000); m.cuda().train()
    lg = m(X.cuda())
    if loss_name == "ce":
        loss = seg.CrossEntropyLoss()(lg, Y.cuda())
    else:
        loss = seg.WeightedDiceCELoss(smooth_dice=1.0, class_weights=torch.tensor(CW3))(lg, Y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    check_bf16(lg.detach().float().cpu(), float(loss.detach()), lr, loss_ref)
    stats = {}
    for n, p in m.named_parameters():
        g = p.grad.detach().double().cpu().flatten()
        if n.endswith(".bias") and ("doubleConvReLU.0" in n or "doubleConvReLU.3" in n):
            assert g.abs().max().item() == 0.0, n
            continue
        r = g64[n].flatten()
        stats[n] = (float(g @ r / (g.norm() * r.norm() + 1e-30)), float(g.norm() / (r.norm() + 1e-30)), p.dim())
    worst = sorted(stats.items(), key=lambda kv: kv[1][0])[:6]
    wr = sorted(stats.items(), key=lambda kv: -abs(kv[1][1] - 1.0))[:6]
    allg = torch.cat([p.grad.detach().double().cpu().flatten() for n, p in m.named_parameters() if n in stats])
    allr = torch.cat([g64[n].flatten() for n, p in m.named_parameters() if n in stats])
    gcos = float(allg @ allr / (allg.norm() * allr.norm()))
    med = sorted(v[0] for v in stats.values())[len(stats) // 2]
    print(f"bf16 backward B={B} {S}x{S} {loss_name}: whole-gradient cosine {gcos:.4f}, median per-parameter cosine {med:.4f}; "
          f"lowest (cos, norm ratio, dim)", worst, "largest norm deviations", wr)
    # Measured on MI355X when the gates were set (round 4): whole-gradient cosine 0.9985 (config 2) / 0.9997 (config 5); per
    # parameter cosine >= 0.868 / 0.857 (median 0.94 / 0.93: direction noise of bf16-stored activations through 23 layers of
    # nearly cancelling sums, cf. the autocast yardstick of the B=8 test below); norm ratio of every conv / ConvTranspose WEIGHT
    # within 0.2 % of 1; of the vectors (BatchNorm weight / bias, ConvTranspose and head bias) within 5.9 % (config 2) / 15 %
    # (config 5: up4.upsample.bias 1.150, everything else within 8.7 %).
    assert gcos >= 0.995, gcos
    vec_tol = 0.09 if S == 256 else 0.18
    for n, (cos, ratio, dim) in stats.items():
        assert cos >= 0.83, (n, cos, worst)
        assert abs(ratio - 1.0) <= (0.02 if dim > 1 else vec_tol), (n, ratio, wr)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("cfg", [(32, 256, "ce"), (8, 512, "dicece")], ids=["config2_B32_256", "config5_B8_512_dicece"])
def test_unet_backward_at_bench_batch_fp32(seg, cfg):
    """fp32 GRADIENT parity at the batch the number is taken on (reference utils/training.py:45-53: forward, loss,
    backward).  Every kernel of the backward runs at the bench geometry here -- split-K slab counts and persistent-unit
    ranges of the weight gradient, the data-gradient convs, the BatchNorm-backward reductions over 2 M pixels, the
    pooling backward with 32-bit offsets into 537 MB tensors -- and every parameter gradient is compared with the
    oracle's: relative L2 error <= 2e-3 per tensor (batch statistics rest on >= 2048 values per channel at every level,
    so the B=1 tests' 2e-2 allowance does not apply) or, where the oracle's own fp32 noise is larger than that, within three
    times that noise of the graph evaluated in float64; conv biases ahead of a batch-statistics BatchNorm exactly zero."""
    B, S, loss_name = cfg
    X, Y, g64, g32, lr, loss_ref = oracle_backward(B, S, loss_name)
    seg.set_compute_dtype(torch.float32)
    m = seg.unet(3, 3); fill_module(m, 1000); m.cuda().train()
    lg = m(X.cuda())
    if loss_name == "ce":
        loss = seg.CrossEntropyLoss()(lg, Y.cuda())
    else:
        loss = seg.WeightedDiceCELoss(smooth_dice=1.0, class_weights=torch.tensor(CW3))(lg, Y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    check_fp32(lg.detach().float().cpu(), float(loss.detach()), lr, loss_ref)

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    errs = {}
    for n, p in m.named_parameters():
        g = p.grad.detach().double().cpu()
        if n.endswith(".bias") and ("doubleConvReLU.0" in n or "doubleConvReLU.3" in n):
            assert g.abs().max().item() == 0.0, n                  # cancels in the batch-statistics BatchNorm
            assert g32[n].norm().item() <= 1e-5 * max(1.0, g32[n.replace(".bias", ".weight")].norm().item()), n
            continue
        errs[n] = (rel(g, g32[n]), rel(g, g64[n]), rel(g32[n], g64[n]))   # vs oracle, vs fp64, the oracle's own fp32 noise
    worst = sorted(errs.items(), key=lambda kv: -kv[1][0])[:6]
    worst64 = sorted(errs.items(), key=lambda kv: -kv[1][1] / max(kv[1][2], 1e-12))[:4]
    print(f"fp32 backward B={B} {S}x{S} {loss_name}: {len(errs)} parameter gradients; (vs oracle fp32, vs fp64, oracle fp32 vs fp64) "
          f"worst vs oracle", worst, "worst relative to the oracle's own fp32 noise", worst64)
    assert len(errs) == 82 - 18, len(errs)                        # 82 parameters, 18 cancelled conv biases
    for n, (e32, e64, eo) in errs.items():
        # The gate asked for is 2e-3 against the oracle, and most tensors meet it.  But the ORACLE'S OWN fp32 rounding noise --
        # its distance from the same graph evaluated in double precision -- is 3.7e-3 at the deep encoder levels (measured on
        # the MI355X box's host, config 2: down4 / down5 weights and BatchNorm vectors; their gradient has come through 20
        # layers of nearly cancelling sums), so no fp32 implementation can sit within 2e-3 of it there: two correct ones differ
        # by about the root sum of squares of their noises.  Those tensors are held to the fp64 yardstick instead: no further
        # from it than three times the oracle is (measured: 1.1 .. 2.4 times; 5.9e-3 where the oracle has 3.7e-3), and never
        # beyond 1e-2 from the oracle itself.
        assert e32 <= 2e-3 or e64 <= 3.0 * eo, (n, e32, e64, eo)
        assert e32 <= 1e-2, (n, e32)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("cfg", [(32, 256, "ce"), (8, 512, "dicece")], ids=["config2_B32_256", "config5_B8_512_dicece"])
def test_unet_forward_at_bench_batch_bf16(seg, cfg):
    B, S, loss_name = cfg
    _, _, lr, loss_ref = unet_oracle(B, S, loss_name)
    lg, loss = unet_hip(seg, torch.bfloat16, B, S, loss_name)
    check_bf16(lg, loss, lr, loss_ref)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_clip_decoder_at_config4_batch(seg, dtype):
    """Config 4 (B=16, 224x224): the decoder + head on the HIP kernels against the oracle decoder, both fed the ORACLE's
    ViT-B/16 features (local random-weight config, portable-fill weights: oracle/clip_vit_ref.py)."""
    pytest.importorskip("transformers")
    from transformers import CLIPVisionConfig, CLIPVisionModel
    B = 16
    vit = CLIPVisionModel(CLIPVisionConfig(patch_size=16)); fill_module(vit, 8100); vit.eval()
    X = fill((B, 3, 224, 224), 9, -1, 1)
    bott, skips = clip_vit_ref.encoder_features(vit, X, [3, 5, 7, 9])
    ref = clipunet_ref.UNetDecoder(768, [1024, 512, 256, 128, 64]); fill_module(ref, 5000); ref.train()
    rh = torch.nn.Conv2d(64, 4, 1); fill_module(rh, 6000)
    Y = labels((B, 224, 224), 3, 4)
    with torch.no_grad():
        lr = rh(ref(bott, skips))
        loss_ref = float(losses_ref.cross_entropy(lr, Y))
    seg.set_compute_dtype(dtype)
    from image_segmentation_amd import ops
    dec = seg.UNetDecoder(768, [1024, 512, 256, 128, 64]); fill_module(dec, 5000)
    head = torch.nn.Conv2d(64, 4, 1); fill_module(head, 6000)
    dec.cuda().train(); head.cuda()
    d = dec(bott.cuda(), [s.cuda() for s in skips])               # autograd on: the training-forward kernels
    lg = ops.HeadFn.apply(dec, d, head.weight, head.bias)
    loss = float(seg.CrossEntropyLoss()(lg, Y.cuda()))
    lg = lg.detach().float().cpu()
    if dtype == torch.float32:
        assert (lg - lr).abs().max().item() < 2e-3                  # four decoder blocks deep on O(1) features
        same = lg.argmax(1) == lr.argmax(1)
        top2 = lr.topk(2, dim=1).values
        assert bool((same | ((top2[:, 0] - top2[:, 1]) < 1e-4)).all())    # disagreements only inside reference near-ties
        assert abs(loss - loss_ref) < 1e-4
    else:
        check_bf16(lg, loss, lr, loss_ref)


@pytest.mark.timeout(900)
def test_unet_bf16_gradients_at_a_conditioned_batch(seg):
    """B=8, 3x128x128, CrossEntropy: every BatchNorm of the U-Net sees >= 512 values per channel (down5: 8 x 8 x 8).
    bf16 gradients against the fp32 oracle, per tensor, with a YARDSTICK: stock torch's own bf16 autocast of the same
    oracle model on the CPU.  At random initialisation with random labels the gradient is an almost completely cancelling
    sum, and bf16 rounding of the stored activations / gradients (2^-9 per element, 23 layers deep) does not cancel: ANY
    bf16 pipeline then sits at per-tensor cosines of 0.85 .. 0.99 against fp32 (torch autocast: median 0.92 measured here)
    -- the 0.99 asked for is not reachable by rounding alone, so the gates are: every tensor cosine >= 0.8, norm within
    8 %, and the median cosine no worse than torch autocast's (- 0.02).  A sign flip (-1), a permutation (~0) or a 10 %
    scale error of any tensor fails; fp32 mode pins the same kernels' arithmetic to 1e-2 relative (test_gpu_fullsize)."""
    B, S = 8, 128
    X = fill((B, 3, S, S), 1, 0, 1); Y = labels((B, S, S), 2, 3)

    def oracle_grads(autocast):
        ref = unet_ref.unet(3, 3); fill_module(ref, 1000); ref.train()
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = ref(X)
            losses_ref.cross_entropy(out.float(), Y).backward()
        else:
            losses_ref.cross_entropy(ref(X), Y).backward()
        return {n: p.grad.detach().clone() for n, p in ref.named_parameters()}
    gref, gauto = oracle_grads(False), oracle_grads(True)
    seg.set_compute_dtype(torch.bfloat16)
    m = seg.unet(3, 3); fill_module(m, 1000); m.cuda().train()
    seg.CrossEntropyLoss()(m(X.cuda()), Y.cuda()).backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}

    def cos_ratio(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return float(a @ b / (a.norm() * b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30))
    stats, yard = {}, {}
    for n, r in gref.items():
        if n.endswith(".bias") and ("doubleConvReLU.0" in n or "doubleConvReLU.3" in n):
            assert g[n].abs().max().item() == 0.0, n               # cancels in the batch-statistics BatchNorm
            continue
        stats[n] = cos_ratio(g[n], r) + (r.dim(),)
        yard[n] = cos_ratio(gauto[n], r)[0]
    worst = sorted(stats.items(), key=lambda kv: kv[1][0])[:6]
    wr = sorted(stats.items(), key=lambda kv: -abs(kv[1][1] - 1.0))[:6]
    med = sorted(v[0] for v in stats.values())[len(stats) // 2]
    med_yard = sorted(yard.values())[len(yard) // 2]
    print(f"bf16 B=8 128x128: median cosine {med:.4f} (torch CPU autocast bf16: {med_yard:.4f}, min {min(yard.values()):.4f}); "
          f"lowest", worst, "largest norm deviations", wr)
    assert med >= med_yard - 0.02, (med, med_yard)
    for n, (cos, ratio, dim) in stats.items():
        assert cos >= 0.8, (n, cos, worst)
        assert abs(ratio - 1.0) <= 0.08, (n, ratio, wr)
