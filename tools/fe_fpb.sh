for f in 1 2 4 8; do echo -n "FPB=$f "; LELE_HIP_LAB=1 LELE_HIP_FE_FPB=$f bash tools/kstats.sh fe$f python tools/microbench.py --only frontend > /dev/null 2>&1; python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/kstats_fe$f/k_kernel_trace.csv")))
v=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "generic_fused" in r["Kernel_Name"])
print(v[0], v[8], v[9], v[-1]) if len(v)>=18 else print(v)
PY
done
