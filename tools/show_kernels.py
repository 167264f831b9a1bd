import json,sys
d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"]):
    print("%-22s %5s %.4f ms  %.0f TF  %.0f GB/s" % (k, v["launches_per_step"], v["ms_per_step"], v["tflops"], v["alg_GBps"]))
