R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4/kil_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base kil; do
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$v -o r -- python $R/tools/kbench.py conv --iters 2 --lib $R/tools/ubench/bin/libsegk_$v.so > $OUT/$v.log 2>&1
  python - <<PY
import csv,collections
rows=[r for r in csv.DictReader(open("$OUT/$v/r_counter_collection.csv")) if r["Counter_Name"]=="WRITE_SIZE" and ("conv3x3_pipe" in r["Kernel_Name"] or "conv_rs" in r["Kernel_Name"])]
rows.sort(key=lambda r:int(r["Dispatch_Id"]))
vals=[float(r["Counter_Value"])*1024/1e6 for r in rows]
print("$v", [round(sum(vals[i:i+5])/5,1) for i in range(0,len(vals),5)])
PY
done
