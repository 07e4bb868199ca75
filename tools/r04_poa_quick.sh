#!/bin/bash
# GPU box: quick look at k_poa_tile after a kernel change: one bench step with phase cycle counters (NGSID_POA_PHASES=1: host-driven levels) and one SQ-counter pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O
BARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-extra-step"
NGSID_POA_PHASES=${PHASES:-1} timeout 600 python $R/bench.py $BARGS > $O/phases.json 2> $O/phases.txt
rm -rf $O/pmc; timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc -o pmc -- python $R/bench.py $BARGS > $O/pmc.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,re
ph=collections.defaultdict(float); rows=0
for line in open("gpurun_out/r4q/phases.txt"):
    m=re.search(r"prepass ([\d.]+) forward ([\d.]+) traceback ([\d.]+) update ([\d.]+) emit ([\d.]+) \| rows (\d+)",line)
    if m:
        for k,v in zip(("prepass","forward","traceback","update","emit"),m.groups()[:5]): ph[k]+=float(v)
        rows+=int(m.group(6))
    m=re.search(r"update: A ([\d.]+) S\+D ([\d.]+) N ([\d.]+)",line)
    if m:
        for k,v in zip(("uA","uSD","uN"),m.groups()): ph[k]+=float(v)
    m=re.search(r"row kinds: tight (\d+) in (\d+) runs, chain (\d+), near (\d+), generic (\d+)",line)
    if m:
        for k,v in zip(("tight","runs","chain","near","generic"),m.groups()): ph["k_"+k]+=int(v)
tot=sum(ph[k] for k in ("prepass","forward","traceback","update","emit"))
print("update split (share of all phases):",{k:round(ph[k]/tot,3) for k in ("uA","uSD","uN")})
print("rows",rows,{k:round(ph[k]/tot,3) for k in ("prepass","forward","traceback","update","emit")}, {k:int(v) for k,v in ph.items() if k.startswith("k_")})
fs=glob.glob("gpurun_out/r4q/pmc/**/*counter_collection.csv",recursive=True)
sq=collections.defaultdict(float)
for r in csv.DictReader(open(fs[0])):
    if "k_poa_tile1" in r["Kernel_Name"]: sq[r["Counter_Name"]]+=float(r["Counter_Value"])
kt=glob.glob("gpurun_out/r4q/pmc/**/*kernel_trace.csv",recursive=True); dur=0
for r in csv.DictReader(open(kt[0])):
    if "k_poa_tile1" in r["Kernel_Name"]: dur+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
if rows: print("per row: VALU %.2f SALU %.2f LDS %.2f VMEM %.2f | kernel ms %.1f | waiting %.3f"%(sq["SQ_INSTS_VALU"]/rows,sq["SQ_INSTS_SALU"]/rows,sq["SQ_INSTS_LDS"]/rows,sq["SQ_INSTS_VMEM"]/rows,dur/1e6,sq["SQ_WAIT_ANY"]/max(sq["SQ_WAVE_CYCLES"],1)))
PY
