#!/usr/bin/env python3
"""Micro-benchmark of single kernels at U-Net layer shapes (B=32, 256x256 input): conv3x3 forward /
weight-gradient through the C ABI.  Usage: python tools/kbench.py [conv|wgrad|all] [--iters N]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from image_segmentation_amd import _lib
if "--lib" in sys.argv:                      # diagnostic builds (ablation / stamp variants of libsegk.so)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from image_segmentation_amd import ops

LAYERS = [  # name, Cin, Cout, HW   (B = 32)
    ("64->64@256", 64, 64, 256), ("128->64@256", 128, 64, 256), ("64->128@256(dgrad up4)", 64, 128, 256),
    ("128->128@128", 128, 128, 128), ("256->256@64", 256, 256, 64), ("512->512@32", 512, 512, 32),
    ("1024->1024@16", 1024, 1024, 16), ("3->64@256", 32, 64, 256),
    # the Cin != Cout layers of the U-Net (first conv of each block and the data gradients of those)
    ("64->128@128", 64, 128, 128), ("128->256@64", 128, 256, 64), ("256->512@32", 256, 512, 32), ("512->1024@16", 512, 1024, 16),
    ("1024->512@32", 1024, 512, 32), ("512->256@64", 512, 256, 64), ("256->128@128", 256, 128, 128),
    ("128->64@128(dgrad)", 128, 64, 128), ("256->128@64(dgrad)", 256, 128, 64), ("512->256@32(dgrad)", 512, 256, 32),
    ("1024->512@16(dgrad)", 1024, 512, 16), ("512->1024@32(dgrad)", 512, 1024, 32), ("256->512@64(dgrad)", 256, 512, 64),
    ("128->256@128(dgrad)", 128, 256, 128),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--pro", action="store_true")
    ap.add_argument("--no-stats", action="store_true", help="conv without the BatchNorm statistics epilogue")
    ap.add_argument("--lib", type=str, default="")
    ap.add_argument("--list", type=str, default="", help="write the layers that ran (name|cin|cout|hw per line) to this file")
    ap.add_argument("--stamps", action="store_true", help="with the stamp build (tools/stamp_build.sh): print the conv_rs "
                    "kernel's per-phase cycle shares (read from the statistics buffer the diagnostic build overwrites)")
    args = ap.parse_args()
    dt = torch.bfloat16
    B = 32
    ran = []
    for name, cin, cout, hw in LAYERS:
        if args.only and args.only not in name:
            continue
        ran.append(f"{name}|{cin}|{cout}|{hw}")
        if args.list:
            open(args.list, "w").write("\n".join(ran) + "\n")
        x = torch.randn((B, hw, hw, cin), device="cuda").to(dt)
        g = torch.randn((B, hw, hw, cout), device="cuda").to(dt)
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
        wp = ops.pack_conv(w, cin, 0, dt, 0)
        out = torch.empty((B, hw, hw, cout), dtype=dt, device="cuda")
        sc = torch.rand(cin, device="cuda") + 0.5 if args.pro else None
        sh = torch.rand(cin, device="cuda") - 0.5 if args.pro else None
        fl = 2.0 * B * hw * hw * 9 * cin * cout
        by = B * hw * hw * (cin + cout) * 2
        if args.what in ("conv", "all"):
            tiles = _lib.query("segk_conv_tiles", B, hw, hw, cin, cout, 1)
            st = torch.empty((_lib.query("segk_bn_stats_floats", tiles, cout),), dtype=torch.float32, device="cuda")
            us = timeit(lambda: ops.conv3x3(x, x.data_ptr(), cin, 0, 0, wp, out.data_ptr(), cout, 0, 0, B, hw, hw, dt,
                                            scale=sc, shift=sh, stats=None if args.no_stats else st), args.iters)
            print(f"conv  {name:26s} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {by/us/1e3:7.1f} GB/s(alg)")
            if args.stamps:
                torch.cuda.synchronize()
                n = min(st.numel() // 2, 256 * 8 * 8)
                t = st[:2 * n].view(torch.int64).view(-1, 8, 8).double().cpu()       # [workgroup][wave][8]
                tot = t[:, :, 6].clamp(min=1)
                names = ["barrier wait", "MFMA loop", "DMA issue", "wait next tile", "epilogue", "loop tail/transform"]
                if cin >= 128:
                    # conv3x3_pipe_kernel (waves 0-3 consumers, 4-7 producers)
                    names = ["step work (MFMA | staging)", "step barrier wait", "stage tile | E1 wait", "store tile",
                             "E2..E3 (refill)", "prologue"]
                    if not args.pro and os.environ.get("SEGK_PIPE_DMA", "1") != "0":     # LDS-DMA form (no prologue)
                        names = ["step work (MFMA | DMA issue)", "step barrier wait", "direct epilogue | vmcnt wait", "-",
                                 "zero + first reads", "prologue"]
                print(f"      stamps over {t.shape[0]} workgroups: kernel {tot.mean():.0f} cycles per wave (min {tot.min():.0f} max {tot.max():.0f})")
                for i, nm in enumerate(names):
                    sh = (t[:, :, i] / tot)
                    print(f"        {nm:22s} {100*sh.mean():5.1f} %   waves0-3 {100*sh[:, :4].mean():5.1f} %  waves4-7 {100*sh[:, 4:].mean():5.1f} %   "
                          f"({t[:, :, i].mean():.0f} cycles)")
        if args.what in ("wgrad", "all"):
            def f():
                ops.wgrad(g.data_ptr(), cout, x.data_ptr(), cin, 0, 0, B, hw, hw, 0, dt, "cuda", scale=sc, shift=sh)
            us = timeit(f, args.iters)
            print(f"wgrad {name:26s} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {by/us/1e3:7.1f} GB/s(alg)")
            if args.stamps:      # stamp build: wgrad_dma_kernel wrote per-wave phase cycle sums over the head of the slab buffer
                slabs, _ = ops.wgrad(g.data_ptr(), cout, x.data_ptr(), cin, 0, 0, B, hw, hw, 0, dt, "cuda", scale=sc, shift=sh)
                torch.cuda.synchronize()
                t = slabs[:256 * 8 * 8 * 2].view(torch.int64).view(-1, 8).double().cpu()
                t = t[(t[:, 4] > 0) & (t[:, 4] < 1e9)]
                tot = t[:, 4]
                for i, nm in enumerate(["DMA issue + MFMA rows", "wait for the next tile's DMA", "barrier", "slab stores"]):
                    print(f"        {nm:30s} {100 * (t[:, i] / tot).mean():5.1f} %   ({t[:, i].mean():.0f} cycles of {tot.mean():.0f})")


def convt():
    """ConvTranspose2d(k=2,s=2) layers of the U-Net up path: forward, data gradient, weight gradient."""
    dt = torch.bfloat16
    B = 32
    for cin, cout, hw in [(1024, 512, 16), (512, 256, 32), (256, 128, 64), (128, 64, 128)]:
        x = torch.randn((B, hw, hw, cin), device="cuda").to(dt)
        dy = torch.randn((B, 2 * hw, 2 * hw, cout), device="cuda").to(dt)
        w = torch.randn((cin, cout, 2, 2), device="cuda") / cin ** 0.5
        wf, wd = ops.pack_convt(w, dt, 0), ops.pack_convt(w, dt, 1)
        out = torch.empty_like(dy)
        dx = torch.empty_like(x)
        by = B * hw * hw * (cin + 4 * cout) * 2
        fl = 2.0 * B * hw * hw * cin * 4 * cout
        s = torch.cuda.current_stream().cuda_stream
        t1 = timeit(lambda: _lib.call("segk_convt2x2_fwd", x.data_ptr(), wf.data_ptr(), 0, out.data_ptr(), B, hw, hw, cin,
                                      cout, 1, s), 20)
        t2 = timeit(lambda: _lib.call("segk_convt2x2_dgrad", dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), B, hw, hw, cin,
                                      cout, 1, s), 20)
        t3 = timeit(lambda: ops.wgrad(x.data_ptr(), cin, dy.data_ptr(), cout, 0, 0, B, hw, hw, 2, dt, "cuda"), 20)
        for nm, t in (("fwd", t1), ("dgrad", t2), ("wgrad", t3)):
            print(f"convt {nm:5s} {cin}->{cout}@{hw}  {t:8.1f} us  {fl/t/1e6:7.1f} TF/s  {by/t/1e3:7.1f} GB/s(alg)")


def bn():
    """BatchNorm+ReLU backward (reduce + finalize + apply) and forward apply at the U-Net activation shapes."""
    dt = torch.bfloat16
    B = 32
    for C, hw in [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)]:
        P = B * hw * hw
        z = torch.randn((P, C), device="cuda").to(dt)
        dy = torch.randn((P, C), device="cuda").to(dt)
        dz = torch.empty_like(z)
        sc = torch.rand(C, device="cuda") + 0.5; sh = torch.rand(C, device="cuda") - 0.5
        mu = torch.zeros(C, device="cuda"); rs = torch.ones(C, device="cuda")
        t = timeit(lambda: ops.bn_relu_bwd(dy.data_ptr(), z.data_ptr(), dz.data_ptr(), sc, sh, mu, rs, P, C, dt, "cuda"), 20)
        by = 5.0 * P * C * 2
        print(f"bn_bwd C={C:5d}@{hw:3d}  {t:8.1f} us  {by/t/1e3:7.1f} GB/s (5 passes)")


def loss():
    """Loss forward / backward (CrossEntropy and Dice+CE) through the C ABI at the bench batch: 32 x 3 x 256 x 256 logits."""
    N, C, H, W = 32, 3, 256, 256
    lg = torch.randn((N, C, H, W), device="cuda")
    tg = torch.randint(0, C, (N, H, W), device="cuda")
    part = torch.empty(_lib.query("segk_loss_part_floats", N * H * W), device="cuda")
    state = torch.empty(_lib.query("segk_loss_state_floats"), device="cuda")
    out = torch.empty(1, device="cuda"); go = torch.ones(1, device="cuda"); dl = torch.empty_like(lg)
    st = ops._stream()
    for name, dw, cw in (("ce", 0.0, 1.0), ("dice+ce", 1.0, 1.0)):
        f = lambda: _lib.call("segk_loss_fwd", lg.data_ptr(), tg.data_ptr(), None, N, C, H * W, -1, 1e-5, dw, cw,
                              part.data_ptr(), state.data_ptr(), out.data_ptr(), st)
        b = lambda: _lib.call("segk_loss_bwd", lg.data_ptr(), tg.data_ptr(), None, state.data_ptr(), go.data_ptr(), N, C,
                              H * W, -1, dw, cw, dl.data_ptr(), st)
        tf = timeit(f, 50); tb = timeit(b, 50)
        print(f"loss {name:8s} fwd {tf:7.1f} us ({N*H*W*(4*C+8)/tf/1e3:7.1f} GB/s)   bwd {tb:7.1f} us ({N*H*W*(8*C+8)/tb/1e3:7.1f} GB/s)"
              f"   value {float(out):.6f}")


def head():
    """Output head on the pre-activation of the last block (forward: BN+ReLU + 1x1 conv to 3 classes; backward incl. the
    BatchNorm reductions) through the C ABI at the bench shape: B = 32, 64 channels, 256 x 256."""
    B, C, H, W, ncls = 32, 64, 256, 256, 3
    dt = torch.bfloat16
    P = B * H * W
    z = torch.randn((P, C), device="cuda").to(dt)
    w = torch.randn((ncls, C), device="cuda") / 8; b = torch.zeros(ncls, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.rand(C, device="cuda") - 0.5
    mu = torch.zeros(C, device="cuda"); rs = torch.ones(C, device="cuda")
    logits = torch.empty((B, ncls, H, W), device="cuda"); dl = torch.randn_like(logits)
    dy = torch.empty_like(z)
    part = torch.empty(_lib.query("segk_head_part_floats", P, C), device="cuda")
    nb = _lib.query("segk_head_bwd_blocks", P)
    bnpart = torch.empty(nb * C * 2, device="cuda"); dw = torch.empty((ncls, C), device="cuda"); db = torch.empty(ncls, device="cuda")
    st = ops._stream()
    f = lambda: _lib.call("segk_head_fwd_bn", z.data_ptr(), sc.data_ptr(), sh.data_ptr(), w.data_ptr(), b.data_ptr(),
                          logits.data_ptr(), B, H, W, C, C, ncls, 1, st)
    g = lambda: _lib.call("segk_head_bwd_bn", dl.data_ptr(), z.data_ptr(), w.data_ptr(), dy.data_ptr(), part.data_ptr(),
                          dw.data_ptr(), db.data_ptr(), B, H, W, C, C, ncls, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
                          rs.data_ptr(), bnpart.data_ptr(), 1, st)
    tf = timeit(f, 30); tb = timeit(g, 30)
    print(f"head fwd {tf:7.1f} us ({P*(2*C+4*ncls)/tf/1e3:7.1f} GB/s)   bwd {tb:7.1f} us ({P*(4*C+4*ncls)/tb/1e3:7.1f} GB/s)")


def vit():
    """The four GEMMs of a ViT-B/16 encoder layer at BASELINE config 4 (B = 16, 197 tokens: M = 3152 rows): QKV, out_proj
    (K split three ways), fc1 (+ quick_gelu), fc2 (K split three ways), through the C ABI."""
    dt = torch.bfloat16
    M, D, I = 16 * 197, 768, 3072
    Mp = (M + 15) // 16 * 16
    st = ops._stream()
    def packed(n, k):
        w = torch.randn((n, k), device="cuda") / k ** 0.5
        return ops.pack_conv(w.reshape(n, k, 1, 1), k, 0, dt, 0, taps=1)
    nosplit = os.environ.get("KB_NOSPLIT") is not None      # with SEGK_GEMM_PIPE_MIN_CHUNKS=1000: the generic kernel, no split-K
    for name, K, N, act, S in (("qkv", D, 3 * D, 0, 1), ("out_proj", D, D, 0, 3), ("fc1", D, I, 1, 1), ("fc2", I, D, 0, 3)):
        S = 1 if nosplit else S
        a = torch.randn((Mp, K), device="cuda").to(dt)
        w = packed(N, K)
        bias = torch.zeros(N, device="cuda")
        o = torch.empty((S * Mp, N), dtype=dt, device="cuda")
        if S > 1:
            f = lambda: _lib.call("segk_linear_splitk", a.data_ptr(), w.data_ptr(), bias.data_ptr(), o.data_ptr(), Mp, K, N, S, 1, st)
        else:
            f = lambda: _lib.call("segk_linear", a.data_ptr(), w.data_ptr(), bias.data_ptr(), o.data_ptr(), Mp, K, N, act, 1, st)
        t = timeit(f, 50)
        fl = 2.0 * M * K * N
        print(f"vit {name:9s} M={M} K={K:5d} N={N:5d} S={S}  {t:7.1f} us  {fl/t/1e6:7.1f} TF/s")


def stem():
    """The U-Net stem (3 -> 64 @ 256 x 256 from the NCHW fp32 batch, B = 32): forward with statistics, and weight gradient."""
    B, Cin, Cout, H, W = 32, 3, 64, 256, 256
    x = torch.rand((B, Cin, H, W), device="cuda")
    w = torch.randn((Cout, Cin, 3, 3), device="cuda") / 5
    z = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device="cuda")
    dz = torch.randn((B, H, W, Cout), device="cuda").to(torch.bfloat16)
    rows = _lib.query("segk_stem3x3_rows", B, H, W, Cin, Cout, 1)
    stats = torch.empty(_lib.query("segk_bn_stats_floats", rows, Cout), device="cuda")
    S = _lib.query("segk_stem3x3_wgrad_slabs", B, H, W, Cin, Cout, 1)
    slabs = torch.empty(S * 64 * 32, device="cuda")
    st = ops._stream()
    tf = timeit(lambda: _lib.call("segk_stem3x3", x.data_ptr(), w.data_ptr(), z.data_ptr(), 0, stats.data_ptr(), B, H, W, Cin,
                                  Cout, 1, st), 30)
    tw = timeit(lambda: _lib.call("segk_stem3x3_wgrad", x.data_ptr(), dz.data_ptr(), slabs.data_ptr(), B, H, W, Cin, Cout, 1, st), 30)
    P = B * H * W
    print(f"stem fwd {tf:7.1f} us ({P*(Cin*4+Cout*2)/tf/1e3:7.1f} GB/s)   wgrad {tw:7.1f} us ({P*(Cin*4+Cout*2)/tw/1e3:7.1f} GB/s)")


def pack():
    """Weight re-layout after an optimizer step: one-pass forward + data-gradient pack against the two per-mode packs,
    all 18 Conv3x3 weights of the U-Net."""
    dt = torch.bfloat16
    shapes = [(64, 3, 0), (64, 64, 0), (128, 64, 0), (128, 128, 0), (256, 128, 0), (256, 256, 0), (512, 256, 0), (512, 512, 0),
              (1024, 512, 0), (1024, 1024, 0), (512, 512, 512), (512, 512, 0), (256, 256, 256), (256, 256, 0),
              (128, 128, 128), (128, 128, 0), (64, 64, 64), (64, 64, 0)]
    ws = [torch.randn((co, ca + cb, 3, 3), device="cuda") for co, ca, cb in shapes]
    t_both = timeit(lambda: [ops.pack_conv_both(w, ca, cb, dt) for w, (co, ca, cb) in zip(ws, shapes)], 20)
    t_two = timeit(lambda: [(ops.pack_conv(w, ca, cb, dt, 0), ops.pack_conv(w, ca, cb, dt, 1)) for w, (co, ca, cb) in zip(ws, shapes)], 20)
    nbytes = sum(w.numel() for w in ws) * (4 + 2 + 2)
    print(f"pack all 18 weights: one-pass {t_both:8.1f} us ({nbytes/t_both/1e3:7.1f} GB/s)   per-mode {t_two:8.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pack":
        pack()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stem":
        stem()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vit":
        vit()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "head":
        head()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "loss":
        loss()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bn":
        bn()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "convt":
        convt()
        sys.exit(0)
    main()
