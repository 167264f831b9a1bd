#!/usr/bin/env python3
"""Launch census of one U-Net training step (B=32, 3x256x256, bf16): every kernel launch by name, and for the stock torch
kernels (fills, copies, reductions: launches the HIP library did not make) the Python frames that issued them.
    python tools/launch_census.py [--steps 3] [--batch 32] [--size 256]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--model", default="unet", choices=["unet", "clipunet"])
    args = ap.parse_args()
    import torch
    from torch.profiler import profile, ProfilerActivity
    import image_segmentation_amd as seg
    import bench
    dev = torch.device("cuda", 0)
    seg.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(1234)
    ncls = 3
    if args.model == "clipunet":
        ncls = 4
        model = seg.ClipUNet(num_classes=4, encoder=seg.ClipViTEncoder.from_config()).to(dev).train()
    else:
        model = seg.unet(3, 3).to(dev).train()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], weight_decay=0.01, fused=True)
    loss_fn = seg.CrossEntropyLoss()
    X = bench.fill((args.batch, 3, args.size, args.size), 1, 0, 1).to(dev)
    Y = bench.labels((args.batch, args.size, args.size), 2, ncls).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(X), Y)
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    kern = collections.Counter()
    for e in prof.events():
        if e.device_type is not None and str(e.device_type).endswith("CUDA"):
            kern[e.name[:100]] += 1
    total = sum(kern.values())
    print(f"{total / args.steps:.1f} device launches per step ({total} over {args.steps} steps)")
    for name, c in kern.most_common():
        print(f"  {c / args.steps:7.2f}  {name}")
    # who issues the stock kernels: CPU-side aten ops with a stack inside this repository
    print("\nstock aten ops issued from this repository's Python (per step):")
    ops_ = collections.Counter()
    for e in prof.events():
        if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::sum", "aten::clone", "aten::add_", "aten::mul",
                      "aten::_foreach_add_", "aten::zeros", "aten::ones_like", "aten::_to_copy") and e.stack:
            fr = [f for f in e.stack if ROOT in f and "launch_census" not in f]
            if fr:
                ops_[(e.name, fr[0].strip()[-110:])] += 1
    for (name, fr), c in sorted(ops_.items(), key=lambda kv: -kv[1]):
        print(f"  {c / args.steps:6.2f}  {name:22s} {fr}")


if __name__ == "__main__":
    main()
