#!/bin/bash
# dev helper (GPU box): SQ counters of k_poa_tile on the POA micro workload; prints per-dispatch sums
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_poa
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --pmc ${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM} --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/tools/micro/time_poa.py ${1:-40000} > $OUT/run.log 2>&1
tail -3 $OUT/run.log
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_poa/**/*counter_collection.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "k_poa_tile" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
for d in sorted(agg,key=int)[:3]:
    print(d, {k:int(v) for k,v in agg[d].items()})
PY
