#!/usr/bin/env python3
"""Data-parallel equivalence check on the one-GPU box: 2 ranks (one process each) sharing cuda:0 over gloo, real
unet(3,3) on the HIP kernels, fp32 parity mode and bf16 (tests/ddp_worker.py does the work; tests/test_gpu_ddp.py is
the same check under pytest).  Writes profiles/<name>.json.   python tools/ddp_check.py [out.json]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_ddp import run_check          # noqa: E402

if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ddp_check.json")
    report = {"what": "2 ranks x unet(3,3) B=2 3x64x64 per rank, gloo, both ranks on cuda:0; synced .grad vs mean of the "
                      "single-process gradients, accumulation 1 and 2 (rel-L2, worst parameter)", "runs": {}}
    ok = True
    with tempfile.TemporaryDirectory() as d:
        for dt in ("f32", "bf16"):
            rcs, res = run_check(dt, os.path.join(d, dt))
            report["runs"][dt] = {"exit_codes": rcs, "ranks": res}
            ok = ok and rcs == [0, 0] and all(r.get("ok") for r in res)
    report["ok"] = ok
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(report, open(out, "w"), indent=1)
    print(json.dumps({"ok": ok, "worst": {k: [r["worst_rel_l2"] for r in v["ranks"] if "worst_rel_l2" in r]
                                          for k, v in report["runs"].items()}}))
    sys.exit(0 if ok else 1)
