#!/bin/bash
# GPU box: HBM traffic of the two DP kernels for one bench step at 1 M reads (two PMC passes, see MI355X_MICROARCH.md "HBM")
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_hbm_$C; rm -rf $OUT; mkdir -p $OUT
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-step > $OUT/run.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmc_hbm_%s/**/*counter_collection.csv"%C,recursive=True)[0]
    agg=collections.defaultdict(float); disp=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k="k_sg_align" if "k_sg_align" in r["Kernel_Name"] else ("k_poa_tile" if "k_poa_tile" in r["Kernel_Name"] else None)
        if k and r["Counter_Name"]==C: agg[k]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in agg: res[k][C]=agg[k]; res[k]["launches"]=len(disp[k])
print(json.dumps(res))
PY
