#!/bin/bash
# GPU box: cost of each phase of k_poa_tile in the real, contended setting: bench step (C3) with the shipped build and with the four timing builds that run
# ONE phase twice (tools/micro/build_repeat.sh); a time run and an SQ-counter run each.  -> gpurun_out/r4/phase_cost.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step"
for V in base rep1 rep2 rep3 rep4; do
  LIB=$R/build_alt/libngsid_hip_$V.so; [ $V = base ] && LIB=$R/ngspeciesid_amd/libngsid_hip.so
  timeout 600 python $R/tools/micro/bench_with_lib.py $LIB $BARGS > $O/pc_$V.json 2> $O/pc_$V.err
  rm -rf $O/pc_pmc_$V; timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pc_pmc_$V -o pmc -- python $R/tools/micro/bench_with_lib.py $LIB --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-extra-step > $O/pc_pmc_$V.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections,json
out={}
for v in ("base","rep1","rep2","rep3","rep4"):
    d={}
    try:
        j=json.loads(open("gpurun_out/r4/pc_%s.json"%v).read().strip().splitlines()[-1])
        d["ms_per_step"]=j["ms_per_step"]; d["kernel_ms"]=j["config"]["kernel_ms_per_step"]; d["check"]=j["config"].get("check")
    except Exception as e: d["err"]=repr(e)
    fs=glob.glob("gpurun_out/r4/pc_pmc_%s/**/*counter_collection.csv"%v,recursive=True)
    sq=collections.defaultdict(float)
    if fs:
        for r in csv.DictReader(open(fs[0])):
            if "k_poa_tile1" in r["Kernel_Name"]: sq[r["Counter_Name"]]+=float(r["Counter_Value"])
    d["sq"]={k:int(x) for k,x in sq.items()}
    out[v]=d
json.dump(out,open("gpurun_out/r4/phase_cost.json","w"),indent=1)
print(json.dumps(out))
PY
