#!/bin/bash
# GPU box, dev tool: extra SQ / SQC counter passes for the POA micro workload (instruction cache, issue stalls) -> gpurun_out/r2/pmcx_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
i=0
for SET in "$@"; do
  i=$((i+1)); rm -rf $O/pmcx_$i
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmcx_$i -o pmc -- python $R/tools/micro/time_poa.py 200000 0 > $O/pmcx_$i.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections
for d in sorted(glob.glob("gpurun_out/r2/pmcx_*/")):
    f=glob.glob(d+"**/*counter_collection.csv",recursive=True)
    if not f: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if "k_poa_tile" in r["Kernel_Name"]: agg[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
    if not agg: print(d,"no rows"); continue
    big=max(agg.values(), key=lambda x: sum(x.values()))
    print(d,{k:int(v) for k,v in big.items()})
PY
