"""mean WRITE_SIZE (MB) per group of five consecutive conv launches (kbench: 3 warm-up + 2 timed per layer)"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "WRITE_SIZE"
        and any(t in r["Kernel_Name"] for t in ("conv3x3_pipe", "conv_rs", "conv_igemm_kernel", "stem_stream"))]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals = [float(r["Counter_Value"]) * 1024 / 1e6 for r in rows]
print(sys.argv[2], [round(sum(vals[i:i + 5]) / 5, 1) for i in range(0, len(vals), 5)])
