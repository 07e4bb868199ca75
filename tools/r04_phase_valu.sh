#!/bin/bash
# GPU box: VALU / SALU / VMEM instructions of k_poa_tile per phase: SQ counter pass with the shipped build and the timing builds that run one phase twice
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O
for V in base "$@"; do
  LIB=$R/build_alt/libngsid_hip_$V.so; [ $V = base ] && LIB=$R/ngspeciesid_amd/libngsid_hip.so
  rm -rf $O/pmc_$V; timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pmc_$V -o pmc -- python $R/tools/micro/bench_with_lib.py $LIB --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-extra-step > $O/pmc_$V.log 2>&1
done
cd $R
python - "$@" <<PY
import csv,glob,collections,sys
rows=3873389396; res={}
for v in ["base"]+sys.argv[1:]:
    fs=glob.glob("gpurun_out/r4v/pmc_%s/**/*counter_collection.csv"%v,recursive=True); sq=collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if "k_poa_tile1" in r["Kernel_Name"]: sq[r["Counter_Name"]]+=float(r["Counter_Value"])
    kt=glob.glob("gpurun_out/r4v/pmc_%s/**/*kernel_trace.csv"%v,recursive=True); dur=0
    for r in csv.DictReader(open(kt[0])):
        if "k_poa_tile1" in r["Kernel_Name"]: dur+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    res[v]={k:sq[k]/rows for k in sq}; res[v]["ms"]=dur/1e6
    print(v, {k:round(x,2) for k,x in res[v].items()})
for v in sys.argv[1:]:
    print("delta",v,{k:round(res[v][k]-res["base"][k],2) for k in res[v]})
PY
