"""GPU (-m gpu): the autoencoder family (reference autoencoder/autoencoder.py) on the HIP kernels against
the goldens captured from the reference and the CPU oracle.  fp32 mode: logits / reconstruction within 1e-3, argmax
bit-exact, gradient norms within 2e-3 relative; bf16 mode gated on the loss."""
import io
import contextlib

import numpy as np
import pytest
import torch

from oracle.fill import fill, labels, fill_module

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import image_segmentation_amd as s
    return s


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _check_grads(model, g, rel=2e-3, abs_=2e-6, skip_frozen=False):
    norms = dict(zip([str(n) for n in g["gnames"]], g["gnorms"]))
    heads = dict(zip([str(n) for n in g["gnames"]], g["gheads"]))
    for n, p in model.named_parameters():
        if p.grad is None:
            assert skip_frozen and norms[n] == 0.0, n
            continue
        gr = p.grad.detach().float().cpu()
        assert abs(gr.double().norm().item() - norms[n]) <= rel * norms[n] + abs_, (n, gr.norm().item(), norms[n])
        h = gr.flatten()[:8].numpy()
        np.testing.assert_allclose(h, heads[n][:h.size], rtol=5e-3, atol=5e-5, err_msg=n)


def test_state_dict_keys_match_oracle(seg):
    from image_segmentation_amd import autoencoder as ae
    from oracle import autoencoder_ref as ref
    a = _quiet(ae.SegmentationAutoencoder, 3, 32, 3, None, False)
    b = ref.SegmentationAutoencoder(3, 32, 3, None, False)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    assert [tuple(v.shape) for v in a.state_dict().values()] == [tuple(v.shape) for v in b.state_dict().values()]
    r1, r2 = ae.ReconstructionAutoencoder(3, 3, 32), ref.ReconstructionAutoencoder(3, 3, 32)
    assert list(r1.state_dict().keys()) == list(r2.state_dict().keys())


def test_segmentation_autoencoder_fp32(seg, golden):
    from image_segmentation_amd import autoencoder as ae
    seg.set_compute_dtype(torch.float32)
    x = fill((2, 3, 32, 32), 1, 0, 1).cuda()
    y = labels((2, 32, 32), 2, 3).cuda()
    g = golden("autoencoder_seg_b2_32")
    m = _quiet(ae.SegmentationAutoencoder, 3, base_channels=32, num_classes=3, freeze_encoder=False)
    fill_module(m, 3000); m.cuda().train()
    logits = m(x)
    loss = seg.CrossEntropyLoss()(logits, y)
    loss.backward()
    lg = logits.detach().float().cpu()
    assert (lg - torch.from_numpy(g["logits"])).abs().max().item() < 1e-3
    assert torch.equal(lg.argmax(1), torch.from_numpy(g["logits"]).argmax(1))
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    _check_grads(m, g)
    rm = m.encoder.encoder.encoderPart1.bn1.running_mean.cpu().numpy()
    np.testing.assert_allclose(rm, g["buf.encoder.encoder.encoderPart1.bn1.running_mean"], rtol=1e-5, atol=1e-6)
    assert int(m.decoder.decoderBlock3.convs[4].num_batches_tracked) == 1
    # frozen encoder: no encoder gradients, decoder/head gradients as the reference
    gf = golden("autoencoder_seg_frozen_b2_32")
    mf = _quiet(ae.SegmentationAutoencoder, 3, base_channels=32, num_classes=3, freeze_encoder=True)
    fill_module(mf, 3000); mf.cuda().train()
    lf = seg.CrossEntropyLoss()(mf(x), y)
    lf.backward()
    assert abs(lf.item() - float(gf["loss"])) < 2e-5
    assert all(p.grad is None for p in mf.encoder.parameters())
    _check_grads(mf, gf, skip_frozen=True)


def test_reconstruction_autoencoder_fp32(seg, golden):
    from image_segmentation_amd import autoencoder as ae
    seg.set_compute_dtype(torch.float32)
    x = fill((2, 3, 32, 32), 1, 0, 1).cuda()
    g = golden("autoencoder_rec_b2_32")
    r = ae.ReconstructionAutoencoder(3, 3, base_channels=32); fill_module(r, 4000); r.cuda().train()
    rec = r(x)
    assert rec.dtype == torch.float32 and rec.is_contiguous() and rec.shape == (2, 3, 32, 32)
    loss = torch.nn.functional.mse_loss(rec, x)
    loss.backward()
    assert (rec.detach().cpu() - torch.from_numpy(g["rec"])).abs().max().item() < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    _check_grads(r, g, abs_=2e-7)


def test_autoencoders_bf16_against_oracle(seg):
    """bf16 performance mode has no reference counterpart: gate on the loss against the fp32 CPU oracle at a size that
    runs the MFMA-bound kernels (base 64, 64x64)"""
    from image_segmentation_amd import autoencoder as ae
    from oracle import autoencoder_ref as ref
    seg.set_compute_dtype(torch.bfloat16)
    x = fill((2, 3, 64, 64), 1, 0, 1)
    y = labels((2, 64, 64), 2, 3)
    o = ref.SegmentationAutoencoder(3, 64, 3, None, False); fill_module(o, 5000); o.train()
    want = torch.nn.functional.cross_entropy(o(x), y).item()
    m = _quiet(ae.SegmentationAutoencoder, 3, 64, 3, None, False); fill_module(m, 5000); m.cuda().train()
    loss = seg.CrossEntropyLoss()(m(x.cuda()), y.cuda())
    loss.backward()
    assert abs(loss.item() - want) < 2e-2
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    seg.set_compute_dtype(torch.float32)
