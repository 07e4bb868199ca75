#!/bin/bash
# GPU box: the round-2 evidence set -> gpurun_out/r2/ (copied into profiles/ afterwards).  Each rocprofv3 pass is its own run (counters never together with traces).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
# 1. the bench line (defaults)
python $R/bench.py > $O/r02_bench_1m.json 2> $O/r02_bench_1m.err
# 2. kernel trace + stats of the same command (no CPU / CLI legs: they add no kernels of interest)
rm -rf $O/prof_stats; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o r02 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > $O/prof_stats.log 2>&1
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/r02_rocprofv3_kernel_stats_1m.csv 2>/dev/null
# 3. HBM traffic: separate FETCH_SIZE / WRITE_SIZE passes, one bench step
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C; timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-step --no-cli > $O/pmc_$C.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/r2/pmc_%s/**/*counter_collection.csv"%C,recursive=True)[0]
    agg=collections.defaultdict(float); disp=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        nm=r["Kernel_Name"]
        k="k_sg_align" if "k_sg_align" in nm else ("k_poa_tile" if "k_poa_tile" in nm else ("k_ed_align" if "k_ed_align" in nm else ("k_hpc_minimizers" if "k_hpc_minimizers" in nm else None)))
        if k and r["Counter_Name"]==C: agg[k]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in agg: res[k][C]=agg[k]; res[k]["launches"]=len(disp[k])
json.dump(res,open("gpurun_out/r2/r02_hbm_raw.json","w"),indent=1)
print(json.dumps(res))
PY
# 4. SQ counters of the POA kernel on the micro workload (200 k reads, one group)
cd /tmp
rm -rf $O/pmc_poa; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pmc_poa -o pmc -- python $R/tools/micro/time_poa.py 200000 0 > $O/pmc_poa.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,json
f=glob.glob("gpurun_out/r2/pmc_poa/**/*counter_collection.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "k_poa_tile" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
big=max(agg.values(), key=lambda d: d.get("SQ_WAVE_CYCLES",0))
json.dump({k:int(v) for k,v in big.items()},open("gpurun_out/r2/r02_pmc_poa_raw.json","w"),indent=1); print({k:int(v) for k,v in big.items()})
PY
NGSID_POA_PHASES=1 python $R/tools/micro/time_poa.py 200000 0 2> $O/r02_poa_phases.txt | tail -1
grep "jobs 2[0-9][0-9][0-9][0-9] " $O/r02_poa_phases.txt | tail -1
