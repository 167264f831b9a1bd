#!/usr/bin/env python3
"""Post-processing of the rocprofv3 --pmc passes the shell tools collect (kept out of the shell scripts: a heredoc hides
syntax errors until the GPU passes have already been paid for; tests/test_tools.py compiles this file on the CPU box).

    pmc_post.py traffic  <out_dir> <repo_root> [round_tag]   FETCH_SIZE / WRITE_SIZE passes  -> <out_dir>/pmc_traffic.json
    pmc_post.py mfma_lds <out_dir> <repo_root>               MFMA busy / LDS conflict passes -> <out_dir>/pmc_mfma_lds.json
    pmc_post.py summary  <out_dir>                           mean counter value per kernel   -> <out_dir>/summary.txt
    pmc_post.py stats    <kernel_stats.csv> <steps>          per-step table of a rocprofv3 --stats summary
    pmc_post.py layers   <out_dir> <launches_per_layer>      FETCH / WRITE passes of `kbench.py conv` -> <out_dir>/layers.txt
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    """kernel name without its argument list; template arguments kept (they tell the variants apart)"""
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    d, k = 0, len(n)
    for i, ch in enumerate(n):          # cut at the '(' that opens the argument list (depth 0 outside <...>)
        if ch == "<":
            d += 1
        elif ch == ">":
            d -= 1
        elif ch == "(" and d == 0:
            k = i
            break
    return n[:k].strip()


def family(name):
    if any(t in name for t in ("conv3x3_pipe", "conv_ws_kernel", "conv_rs_kernel", "stem_stream_kernel")):
        return "conv3x3"
    if "convt_stream_kernel" in name:
        return "conv_other"
    if "conv_igemm_kernel" in name:
        tail = name.split("conv_igemm_kernel")[1]
        return "conv3x3" if ("Li0ELi" in tail[:12] or "E, 0," in name or ", 0, " in tail[:24]) else "conv_other"
    if "wgrad_dma_kernel" in name or "stem_wgrad_kernel" in name:
        return "wgrad3x3"
    if "wgrad_kernel" in name:
        return "wgrad_other"
    return None


def build_id(repo):
    sys.path.insert(0, repo)
    from image_segmentation_amd import _lib
    return _lib.build_id()


def traffic(out, repo):
    per = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for ctr, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        for r in csv.DictReader(open(f"{out}/{d}/r_counter_collection.csv")):
            if r["Counter_Name"] != ctr:
                continue
            v = float(r["Counter_Value"])
            per[short(r["Kernel_Name"])][ctr].append(v)
            f = family(r["Kernel_Name"])
            if f:
                agg[f][ctr].append(v)

    def row(v):
        fe = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        wr = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        return {"launches_sampled": len(v["FETCH_SIZE"]), "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
                "hbm_bytes_per_launch": int((2 * fe + wr) * 1024)}
    res = {"build_id": build_id(repo),
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 "
                      "--no-cpu-baseline --profile-steps 0 --no-extras (separate passes)",
           "unit": "KB per launch (rocprofv3 raw); hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE "
                   "doubled on gfx950 (MI355X_MICROARCH.md HBM section)",
           "families": {f: row(v) for f, v in agg.items() if v["FETCH_SIZE"] and v["WRITE_SIZE"]},
           "kernels": {k: row(v) for k, v in sorted(per.items()) if v["FETCH_SIZE"] and v["WRITE_SIZE"]}}
    json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(res["families"], indent=1))


def mfma_lds(out, repo):
    ctrs = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ctrs:
        for r in csv.DictReader(open(f"{out}/pmc_{c}/r_counter_collection.csv")):
            if r["Counter_Name"] == c:
                per[short(r["Kernel_Name"])][c].append(float(r["Counter_Value"]))
    res = {"build_id": build_id(repo),
           "command": "rocprofv3 --pmc <one counter> --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                      "--profile-steps 0 --no-extras (one pass per counter)",
           "note": "averages per launch, summed over the chip as rocprofv3 reports them; GRBM_GUI_ACTIVE comes summed over "
                   "the 8 XCDs (16.8 'GHz' against wall time), so mfma_duty = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE "
                   "/ 8 x 1024 SIMDs) is the matrix-pipe duty cycle; lds_conflict_share = SQ_LDS_BANK_CONFLICT / "
                   "SQ_LDS_IDX_ACTIVE",
           "kernels": {}}
    for k, v in sorted(per.items()):
        if not all(c in v and v[c] for c in ctrs):
            continue
        a = {c: sum(v[c]) / len(v[c]) for c in ctrs}
        if a["GRBM_GUI_ACTIVE"] < 20000:          # tiny kernels
            continue
        res["kernels"][k] = {"launches_sampled": len(v[ctrs[0]]), **{c: round(a[c], 1) for c in ctrs},
                             "mfma_duty": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 1024), 4),
                             "lds_conflict_share": round(a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"], 4)
                             if a["SQ_LDS_IDX_ACTIVE"] else None}
    json.dump(res, open(f"{out}/pmc_mfma_lds.json", "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{k[:60]:60s} mfma {v['mfma_duty']:.3f}  lds_conflict {v['lds_conflict_share']}")


def summary(out):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{out}/p*/r_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(f"{out}/summary.txt", "w") as o:
        for k, v in agg.items():
            o.write(k + "\n")
            for c, vals in v.items():
                o.write(f"    {c:44s} {sum(vals)/len(vals):16.1f}  (n={len(vals)})\n")
    print(open(f"{out}/summary.txt").read()[:6000])


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps
    print(f"kernel time per step {tot / 1e6:.3f} ms, launches per step {sum(int(r['Calls']) for r in rows) / steps:.1f}")
    for r in rows[:50]:
        print(f"{float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step {int(r['Calls']) / steps:6.2f} x "
              f"{float(r['AverageNs']) / 1e3:8.1f} us  {short(r['Name'])[:100]}")


def layers(out, per_layer):
    """Per-layer HBM traffic of `tools/kbench.py conv [--pro]` (tools/pmc_layers.sh): the conv dispatches arrive in LAYERS order,
    `per_layer` launches each (3 warm-up + --iters); mean over a layer's launches, beside the layer's algorithmic bytes."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    conv = ("conv3x3_pipe", "conv_rs_kernel", "conv_ws_kernel", "conv_igemm_kernel", "stem_stream")
    wg = ("wgrad_dma_kernel", "wgrad_kernel", "stem_wgrad_kernel")
    lines = []
    for tag in ("plain", "pro", "wgrad"):
        if tag == "wgrad":
            conv = wg
        seq = {}
        for ctr, d in (("FETCH_SIZE", f"{tag}_fetch"), ("WRITE_SIZE", f"{tag}_write")):
            path = f"{out}/{d}/r_counter_collection.csv"
            if not os.path.exists(path):
                continue
            rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == ctr and any(t in r["Kernel_Name"] for t in conv)]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            seq[ctr] = [(short(r["Kernel_Name"]), float(r["Counter_Value"])) for r in rows]
        if len(seq) < 2:
            continue
        layer_list = [ln.rstrip("\n").split("|") for ln in open(f"{out}/{tag}_layers.txt")]
        n = min(len(seq["FETCH_SIZE"]), len(seq["WRITE_SIZE"])) // per_layer
        what = "wgrad" if tag == "wgrad" else "conv" + (" --pro" if tag == "pro" else "")
        lines.append(f"== kbench {what}: MB per launch, (2 x FETCH_SIZE + WRITE_SIZE) against (Cin + Cout) x B x H x W x 2 B + "
                     + ("the fp32 split-K slabs actually written (alg out: ONE fp32 weight-gradient tensor)" if tag == "wgrad" else "weights"))
        lines.append(f"{'layer':28s} {'kernel':44s} {'alg in':>8s} {'alg out':>8s} {'fetch':>8s} {'write':>8s} {'ratio':>6s}")
        for i in range(min(n, len(layer_list))):
            name, cin, cout, hw = layer_list[i][0], int(layer_list[i][1]), int(layer_list[i][2]), int(layer_list[i][3])
            fe = [v for _, v in seq["FETCH_SIZE"][i * per_layer:(i + 1) * per_layer]]
            wr = [v for _, v in seq["WRITE_SIZE"][i * per_layer:(i + 1) * per_layer]]
            kern = seq["FETCH_SIZE"][i * per_layer][0]
            P = 32 * hw * hw
            a_in = (P * cin * 2 + 9 * cin * cout * 2) / 1e6
            a_out = P * cout * 2 / 1e6
            if tag == "wgrad":
                a_in = P * (cin + cout) * 2 / 1e6
                a_out = 9 * cin * cout * 4 / 1e6
            f_mb = 2 * sum(fe) / len(fe) * 1024 / 1e6
            w_mb = sum(wr) / len(wr) * 1024 / 1e6
            lines.append(f"{name:28s} {kern[:44]:44s} {a_in:8.1f} {a_out:8.1f} {f_mb:8.1f} {w_mb:8.1f} {(f_mb + w_mb) / (a_in + a_out):6.2f}")
    open(f"{out}/layers.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "layers":
        layers(sys.argv[2], int(sys.argv[3]))
    elif cmd == "traffic":
        traffic(sys.argv[2], sys.argv[3])
    elif cmd == "mfma_lds":
        mfma_lds(sys.argv[2], sys.argv[3])
    elif cmd == "summary":
        summary(sys.argv[2])
    elif cmd == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    else:
        sys.exit(f"unknown command {cmd}")
