"""CPU: host-side logic of the drop-in surface -- train_loop / eval_loop protocol (reference
utils/training.py:18-121) driven with the CPU oracle model, MetricsHistory formulas, loss-module argument
contracts, state_dict layout, act-tensor bookkeeping and the guarantee that the product path refuses CPU
tensors instead of silently falling back."""
import numpy as np
import pytest
import torch

from oracle.fill import fill, labels, fill_module
from oracle import unet_ref, losses_ref
import image_segmentation_amd as seg
from image_segmentation_amd import training, ops
from image_segmentation_amd.metrics import MetricsHistory

training.VERBOSE = False


class OracleCE(torch.nn.Module):
    def forward(self, pred, y):
        return losses_ref.cross_entropy(pred, y)


def test_train_loop_matches_reference_protocol(golden):
    """same data / model / AdamW as tools/gen_golden.py:gen_trainloop, run through OUR train_loop"""
    g = golden("trainloop_unet_32")
    for acc in (1, 2):
        m = unet_ref.unet(3, 3); fill_module(m, 1000)
        opt = torch.optim.AdamW(m.parameters(), weight_decay=0.01)
        data = [(fill((2, 3, 32, 32), 10 + i, 0, 1), labels((2, 1, 32, 32), 20 + i, 3)) for i in range(3)]
        avg = training.train_loop(data, m, OracleCE(), opt, acc, torch.device("cpu"))
        assert abs(avg - float(g[f"acc{acc}_avg"])) < 1e-5
        np.testing.assert_allclose(m.output.weight.detach().numpy(), g[f"acc{acc}_out_w"], atol=2e-5)
        np.testing.assert_allclose(m.down1.doubleConvReLU[1].running_mean.numpy(), g[f"acc{acc}_rm"], atol=1e-6)


def test_train_loop_scheduler_and_accumulation_order():
    calls = []

    class Opt:
        param_groups = [{"lr": 0.1}]
        def zero_grad(self): calls.append("zero")
        def step(self): calls.append("step")

    class Sched:
        def step(self): calls.append("sched")

    class GS:
        def arm(self): calls.append("arm")
        def sync(self): calls.append("sync")

    w = torch.nn.Parameter(torch.ones(1))
    model = torch.nn.Module(); model.forward = lambda X: X * w
    data = [(torch.ones(1, 1), torch.zeros(1, 1, dtype=torch.long)) for _ in range(3)]
    loss_fn = lambda p, y: (p.sum() - 3.0) ** 2
    avg = training.train_loop(data, model, loss_fn, Opt(), 2, "cpu", scheduler=Sched(), grad_sync=GS())
    # zero first; step after micro-batch 2 and after the last (3rd) batch; scheduler right after optimizer
    assert calls == ["zero", "arm", "sync", "step", "sched", "zero", "arm", "sync", "step", "sched", "zero"]
    assert avg == pytest.approx(4.0)


class OracleAgg(MetricsHistory):
    """CPU stand-in for the device confusion kernel (host formulas unchanged)."""
    def accumulate(self, pred, label):
        c = losses_ref.confusion_counts(pred, label, self.num_classes)
        self.total_tp += c[0]; self.total_fp += c[1]; self.total_fn += c[2]; self.total_tn += c[3]


def test_eval_loop_and_metrics_known_answers(golden):
    g = golden("unet_3_3_b4_128")
    logits = torch.from_numpy(g["logits"])
    Y = labels((4, 1, 128, 128), 2, 3)

    class Fixed(torch.nn.Module):       # a "model" that returns the golden logits for each batch of one image
        def __init__(self): super().__init__(); self.i = 0
        def forward(self, X):
            out = logits[self.i:self.i + X.shape[0]]; self.i += X.shape[0]; return out
    data = [([fill((3, 128, 128), 50 + i, 0, 1)], [Y[i, 0]]) for i in range(4)]
    agg = OracleAgg(3)
    loss, dice, iou = training.eval_loop(data, Fixed(), OracleCE(), "cpu", 128, agg)
    counts = np.stack([agg.total_tp.numpy(), agg.total_fp.numpy(), agg.total_fn.numpy(), agg.total_tn.numpy()])
    assert (counts == g["counts"]).all()
    np.testing.assert_allclose([dice, iou], g["metrics"][:2], rtol=1e-12)
    assert abs(loss - float(g["ce"])) < 1e-4        # mean of per-image CE == batch CE (equal image sizes)
    assert len(agg.get_mean_iou_history()) == 1 and agg.get_last_per_class_iou().shape == (3,)


def test_metrics_from_confusion_matrix(golden):
    g = golden("unet_3_3_b4_128")
    hard = g["argmax"].astype(np.int64); lab = labels((4, 128, 128), 2, 3).numpy()
    M = np.zeros((3, 3), dtype=np.int64)
    for p in range(3):
        for l in range(3):
            M[p, l] = ((hard == p) & (lab == l)).sum()
    tp, fp, fn, tn = MetricsHistory.counts_from_confusion(torch.from_numpy(M), hard.size)
    assert (np.stack([tp, fp, fn, tn]) == g["counts"]).all()
    agg = MetricsHistory(4, ignore_index=3)
    assert agg.mask.tolist() == [True, True, True, False] and agg.get_ignore_index() == 3


def test_state_dict_layout_matches_reference_names():
    a, b = seg.unet(3, 4), unet_ref.unet(3, 4)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    assert "down1.doubleConvReLU.0.weight" in a.state_dict()
    assert "down2.maxpool_doubleConv.1.doubleConvReLU.4.num_batches_tracked" in a.state_dict()
    assert a.state_dict()["up1.upsample.weight"].shape == (1024, 512, 2, 2)
    torch.manual_seed(0); x = seg.unet(3, 3)
    torch.manual_seed(0); y = unet_ref.unet(3, 3)
    assert all(torch.equal(p, q) for p, q in zip(x.parameters(), y.parameters()))   # same default init stream


def test_product_path_refuses_cpu_tensors():
    with pytest.raises(RuntimeError, match="no CPU path"):
        seg.unet(3, 3)(torch.zeros(1, 3, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU path"):
        seg.CrossEntropyLoss()(torch.zeros(1, 3, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))


def test_loss_argument_contracts():
    x = torch.zeros(2, 3, 4, 4)
    with pytest.raises(ValueError):      # bare Dice rejects [N,H,W] like the reference (weighted_loss.py:42-46)
        seg.WeightedMemoryEfficientDiceLoss()(x, torch.zeros(2, 4, 4, dtype=torch.long))
    with pytest.raises(ValueError):      # multi-channel targets (weighted_loss.py:146-150)
        seg.WeightedDiceCELoss()(x, torch.zeros(2, 3, 4, 4, dtype=torch.long))
    with pytest.raises(ValueError):
        seg.WeightedDiceCELoss()(x, torch.zeros(2, 4, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        seg.CrossEntropyLoss(reduction="sum")


def test_act_tensor_bookkeeping():
    assert ops.pad32(3) == 32 and ops.pad32(64) == 64 and ops.pad32(65) == 96
    buf = torch.zeros(2, 5, 7, 32)
    v = ops.act_view(buf, 3)
    assert v.shape == (2, 3, 5, 7) and v.stride() == (5 * 7 * 32, 1, 7 * 32, 32)
    assert ops.act_info(v, torch.float32) is None            # CPU tensors never qualify
    ops.set_compute_dtype(torch.float32); assert ops.get_compute_dtype() == torch.float32
    ops.set_compute_dtype(torch.bfloat16)
    with pytest.raises(ValueError):
        ops.set_compute_dtype(torch.float16)


def test_pack_cache_follows_fused_optimizer_steps():
    """torch.optim.AdamW(fused=True) updates a parameter without moving its autograd version counter: the packed-weight
    cache must still notice the step (global optimizer hook); frozen parameters are packed once."""
    p = torch.nn.Parameter(torch.randn(4, 4))
    cache, n = ops.PackCache(), [0]

    def build():
        n[0] += 1
        return n[0]
    opt = torch.optim.AdamW([p], fused=True)
    assert cache.get("k", p, build) == 1 and cache.get("k", p, build) == 1
    v = p._version
    p.grad = torch.randn(4, 4); opt.step()
    if p._version == v:          # the behaviour that made the hook necessary (torch 2.10)
        assert cache.get("k", p, build) == 2 and cache.get("k", p, build) == 2
    a, = [cache.get_pair("a", "b", p, lambda: (10 + n[0], 20 + n[0]))]
    assert cache.get("b", p, build) == a + 10            # served by the pair, no rebuild
    ops.invalidate_packed_weights()
    assert cache.get("b", p, build) != a + 10
    q = torch.nn.Parameter(torch.randn(2, 2), requires_grad=False)
    first = cache.get("q", q, build)
    opt.step()
    assert cache.get("q", q, build) == first             # frozen: optimizer steps do not invalidate it
    with torch.no_grad():
        q.add_(1.0)                                      # a real in-place write does
    assert cache.get("q", q, build) != first


def test_nll_nonlin_recognition():
    from image_segmentation_amd.losses import _log_eps
    assert _log_eps(None) == (0, 0.0)
    mode, eps = _log_eps(lambda t: torch.log(t + 1e-9))
    assert mode == 1 and abs(eps - 1e-9) < 1e-15
    assert _log_eps(torch.log) == (1, 0.0)
    for bad in (torch.sqrt, lambda t: 2 * torch.log(t + 1e-9), lambda t: torch.log(t + 0.5)):
        with pytest.raises(NotImplementedError):
            _log_eps(bad)


def test_resize_geometry_and_host_path_match_oracle():
    from oracle import resize_ref
    from image_segmentation_amd.utils import process_batch_forward, process_batch_reverse, NEAREST
    imgs = [fill(s, 10 + i, 0, 1) for i, s in enumerate([(3, 37, 53), (4, 64, 64), (3, 20, 30), (3, 500, 375)])]
    out, meta = process_batch_forward(imgs, target_size=64)
    ref, rmeta = resize_ref.process_batch_forward(imgs, 64)
    assert meta == rmeta and torch.equal(out, ref)
    # torchvision < 0.17 did not anti-alias tensor resizes: the switch reaches the host path too
    out2, _ = process_batch_forward(imgs, target_size=64, antialias=False)
    ref2, _ = resize_ref.process_batch_forward(imgs, 64, antialias=False)
    assert torch.equal(out2, ref2) and not torch.equal(out2, out)
    labs = [labels((1, 37, 53), 1, 4), labels((1, 20, 30), 2, 4)]
    lo, _ = process_batch_forward(labs, target_size=48, interpolation=NEAREST)
    lr, _ = resize_ref.process_batch_forward(labs, 48, nearest=True)
    assert torch.equal(lo, lr)
    logits = fill((4, 3, 64, 64), 3, -2, 2)
    back = process_batch_reverse(logits, meta)
    for b, l, m in zip(back, logits, meta):
        assert torch.equal(b, resize_ref.reverse_resize_and_padding(l, m))


def test_clip_encoder_loads_both_transformers_key_layouts():
    """Reference CLIP-UNet checkpoints were written with transformers 4.x, whose CLIPVisionModel nests the transformer
    under `.vision_model` (keys encoder.clip_vit.vision_model.*; reference clip/clipunet.py:25-26,
    prompt_based/segmentation_webapp/app.py:65-79); 5.x has no such level.  ClipViTEncoder maps the keys to whichever
    layout the installed module has -- checked in both directions with synthetic state dicts, strict loading, through
    the ClipUNet parent (prefix `encoder.`)."""
    pytest.importorskip("transformers")
    from transformers import CLIPVisionConfig
    cfg = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                           image_size=32, patch_size=16)
    src = seg.ClipUNet(num_classes=4, decoder_channels=[64, 32], encoder=seg.ClipViTEncoder.from_config(cfg, skip_indices=[1]))
    fill_module(src, 4000)
    sd = src.state_dict()
    own = "encoder.clip_vit."
    nested_here = hasattr(src.encoder.clip_vit, "vision_model")
    flat = {(own + k[len(own) + len("vision_model."):] if k.startswith(own + "vision_model.") else k): v for k, v in sd.items()}
    nested = {(own + "vision_model." + k[len(own):] if k.startswith(own) else k): v for k, v in flat.items()}
    assert any(k.startswith(own + "vision_model.") for k in nested) and not any("vision_model" in k for k in flat)
    for name, ckpt in (("4.x layout", nested), ("5.x layout", flat)):
        dst = seg.ClipUNet(num_classes=4, decoder_channels=[64, 32], encoder=seg.ClipViTEncoder.from_config(cfg, skip_indices=[1]))
        res = dst.load_state_dict(dict(ckpt), strict=True)
        assert not res.missing_keys and not res.unexpected_keys, name
        for k, v in dst.state_dict().items():
            assert torch.equal(v, sd[k]), (name, k)
    # the opposite module layout (a transformer nested under `.vision_model`, as transformers 4.x builds it)
    if not nested_here:
        class Nested(torch.nn.Module):
            def __init__(self, inner):
                super().__init__()
                self.vision_model, self.config = inner, inner.config
        dst = seg.ClipUNet(num_classes=4, decoder_channels=[64, 32], encoder=seg.ClipViTEncoder.from_config(cfg, skip_indices=[1]))
        dst.encoder.clip_vit = Nested(dst.encoder.clip_vit)
        for name, ckpt in (("4.x layout", nested), ("5.x layout", flat)):
            res = dst.load_state_dict(dict(ckpt), strict=True)
            assert not res.missing_keys and not res.unexpected_keys, name
        got = dst.state_dict()
        assert all(torch.equal(got[k], v) for k, v in nested.items()) and set(got) == set(nested)
    # a persistent position_ids buffer of an old release is dropped instead of failing the strict load
    old = dict(nested)
    old[own + "vision_model.embeddings.position_ids"] = torch.arange(5).unsqueeze(0)
    dst = seg.ClipUNet(num_classes=4, decoder_channels=[64, 32], encoder=seg.ClipViTEncoder.from_config(cfg, skip_indices=[1]))
    if own[:-1] + ".embeddings.position_ids" not in dst.state_dict() and own + "embeddings.position_ids" not in dst.state_dict():
        dst.load_state_dict(old, strict=True)


def test_pack_caches_reach_the_fused_drivers_outside_the_module_tree():
    """ops._pack_caches (what repack_stale refreshes in one launch): the CLIP decoder keeps its fused drivers (`_dc`, `_skip`,
    `_init`) out of the module tree so that the state_dict keeps the reference's names; their caches must be found all the
    same (round 3: they were not, and every decoder weight was re-packed by a launch of its own)."""
    import image_segmentation_amd as seg
    from image_segmentation_amd import ops
    dec = seg.UNetDecoder(64, [64, 32, 32, 32, 32])
    caches = ops._pack_caches(dec)
    assert len(caches) == len({id(c) for c in caches})
    want = {id(dec.cache), id(dec._init.cache)}
    for blk in dec.decoder_blocks:
        want |= {id(blk.cache), id(blk._dc.cache), id(blk._skip.cache)}
    assert want <= {id(c) for c in caches}, (len(want), len(caches))
    # a plain U-Net: every registered DoubleConv / Up / Down module, nothing twice
    m = seg.unet(3, 3)
    cu = ops._pack_caches(m)
    assert len(cu) == len({id(c) for c in cu}) == sum(1 for x in m.modules() if isinstance(getattr(x, "cache", None), ops.PackCache))
