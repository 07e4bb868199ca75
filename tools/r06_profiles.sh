#!/bin/bash
# GPU box: the round-6 evidence set -> gpurun_out/r6/ (the summaries are copied into profiles/ afterwards).  Every rocprofv3 pass is its own run
# (counters never together with traces other than --kernel-trace).  Usage: NGSID_COMMIT=<git rev> bash tools/r06_profiles.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
BARGS="--no-cpu-baseline --no-cli"
# 1. the bench line (defaults: C3, 1 GPU)
timeout 900 python $R/bench.py > $O/r06_bench_1m.json 2> $O/r06_bench_1m.err
# 2. kernel trace + stats of the same command
rm -rf $O/prof_stats; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o r04 -- python $R/bench.py --steps 2 --warmup 1 $BARGS > $O/prof_stats.log 2>&1
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/r06_rocprofv3_kernel_stats_1m.csv 2>/dev/null
# 2b. the same in ONE context (NGSID_LANES=1): with the default two lanes the launches of the two contexts overlap on the device and a launch's duration includes what it waited for
#     the other lane; this pass gives every kernel's time alone (bench.py: roofline.one_lane).  The counter passes below run in one context as well.
rm -rf $O/prof_stats1; NGSID_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats1 -o r04 -- python $R/bench.py --steps 2 --warmup 1 $BARGS > $O/prof_stats1.log 2>&1
cp $(find $O/prof_stats1 -name "*kernel_stats.csv" | head -1) $O/r06_rocprofv3_kernel_stats_1m_one_lane.csv 2>/dev/null
export NGSID_LANES=1
# 3. HBM traffic: separate FETCH_SIZE / WRITE_SIZE passes, one bench step
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C; timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 $BARGS --no-extra-step > $O/pmc_$C.log 2>&1
done
# 4. SQ counters of the POA kernel ON THE BENCH WORKLOAD at the default tile depth (one step)
rm -rf $O/pmc_sq; timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pmc_sq -o pmc -- python $R/bench.py --steps 1 --warmup 0 $BARGS --no-extra-step > $O/pmc_sq.log 2>&1
# 4b. VALU pipe utilisation from counters alone: cycles in which the VALU executes an instruction / cycles the SQs are busy, + the FLAT (scratch / global) instruction split
rm -rf $O/pmc_sq2; timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_sq2 -o pmc -- python $R/bench.py --steps 1 --warmup 0 $BARGS --no-extra-step > $O/pmc_sq2.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,json,os,re
commit=os.environ.get("NGSID_COMMIT","unknown")
def cls(nm):
    for k in ("k_sg_align","k_poa_tile","k_ed_align","k_hpc_minimizers"):
        if k in nm: return k
    return None
res=collections.defaultdict(dict)
for C in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob("gpurun_out/r6/pmc_%s/**/*counter_collection.csv"%C,recursive=True)
    if not fs: continue
    agg=collections.defaultdict(float); disp=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k=cls(r["Kernel_Name"])
        if k and r["Counter_Name"]==C: agg[k]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in agg: res[k][C]=agg[k]; res[k]["launches_"+C]=len(disp[k])
out={"_how":"tools/r06_profiles.sh on the GPU box: rocprofv3 --pmc FETCH_SIZE --kernel-trace and (separate pass) --pmc WRITE_SIZE -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-step --no-cli (C3: 1 M reads, POA tile depth 4, device-driven hierarchy levels)",
     "_units":"counter values are KiB summed over all dispatches of the kernel in ONE bench step; FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section): hbm_bytes_per_step = 2 x fetch + write",
     "workload_reads":1000000,"config":"c3","commit":commit}
for k,v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out[k]={"launches_per_step":v.get("launches_FETCH_SIZE"),"fetch_kib":v["FETCH_SIZE"],"write_kib":v["WRITE_SIZE"],"hbm_bytes_per_step":int((2*v["FETCH_SIZE"]+v["WRITE_SIZE"])*1024)}
json.dump(out,open("gpurun_out/r6/r06_hbm_traffic.json","w"),indent=1)
print(json.dumps({k:v.get("hbm_bytes_per_step") for k,v in out.items() if isinstance(v,dict)}))
# second pass: VALU-active cycles against busy cycles, per kernel family
pb={}; raw2={}
fs2=glob.glob("gpurun_out/r6/pmc_sq2/**/*counter_collection.csv",recursive=True)
if fs2:
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs2[0])):
        k="k_poa_tile1" if "k_poa_tile1" in r["Kernel_Name"] else ("k_sg_align" if "k_sg_align" in r["Kernel_Name"] else None)
        if k: acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items():
        raw2[k]={a:int(b) for a,b in v.items()}
        if v.get("SQ_BUSY_CU_CYCLES"): pb[k]=round(v["SQ_ACTIVE_INST_VALU"]/v["SQ_BUSY_CU_CYCLES"],4)      # both in quad-cycles per SIMD (see _units of profiles/r06_pmc_pipe_busy.json)
    json.dump({"_how":"rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-extra-step; counters summed over all dispatches of the kernel family in the step","commit":commit,"active_valu_over_busy":pb,"raw":raw2},open("gpurun_out/r6/r06_pmc_pipe_busy.json","w"),indent=1)
    print(json.dumps({"active_valu_over_busy":pb}))
# SQ counters of k_poa_tile1 over the step
fs=glob.glob("gpurun_out/r6/pmc_sq/**/*counter_collection.csv",recursive=True)
sq=collections.defaultdict(float)
if fs:
    for r in csv.DictReader(open(fs[0])):
        if "k_poa_tile1" in r["Kernel_Name"]: sq[r["Counter_Name"]]+=float(r["Counter_Value"])
rows=0
# DP rows of the SAME run as the counters: the kernels count them (poa_dp_rows of ngsid_profile_read -> roofline.dp_kernels of the bench line the counter pass printed);
# the phase-cycle instrumentation that used to provide them is compiled out of the product kernel since round 5 (-DPOA_PHASES=1 dev builds only)
for line in open("gpurun_out/r6/pmc_sq.log", errors="replace"):
    if line.startswith("{") and '"roofline"' in line:
        try: rows=int(json.loads(line)["roofline"]["dp_kernels"]["k_poa_tile"]["dp_rows"])
        except Exception: pass
kt=glob.glob("gpurun_out/r6/pmc_sq/**/*kernel_trace.csv",recursive=True)
dur=0
if kt:
    for r in csv.DictReader(open(kt[0])):
        if "k_poa_tile1" in r["Kernel_Name"]: dur+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
if sq and rows:
    j={"workload_reads":1000000,"config":"c3","commit":commit,"rows":rows,"valu_per_row":round(sq["SQ_INSTS_VALU"]/rows,2),"salu_per_row":round(sq["SQ_INSTS_SALU"]/rows,2),
       "lds_per_row":round(sq["SQ_INSTS_LDS"]/rows,2),"vmem_per_row":round(sq["SQ_INSTS_VMEM"]/rows,2),
       "wave_cycles_waiting_frac":round(sq["SQ_WAIT_ANY"]/max(sq["SQ_WAVE_CYCLES"],1),3),"wave_cycles_issuing_frac":round(sq["SQ_ACTIVE_INST_ANY"]/max(sq["SQ_WAVE_CYCLES"],1),3),
       "kernel_ns_under_the_counter_pass":dur,
       "pipe_busy_assuming_4_cycles_per_instruction":round(sq["SQ_INSTS_VALU"]*4/(1024*(sq["SQ_BUSY_CYCLES"]/32.0)),3) if sq.get("SQ_BUSY_CYCLES") else None,
       "pipe_busy":pb.get("k_poa_tile1"),"pipe_busy_how":"second counter pass (pmc_sq2): SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES summed over the kernel's dispatches (both in quad-cycles per SIMD: no cycle count per instruction is assumed); see raw_second_pass and profiles/r06_pmc_pipe_busy.json",
       "raw_second_pass":raw2.get("k_poa_tile1"),
       "raw":{k:int(v) for k,v in sq.items()},
       "_how":"rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --kernel-trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-extra-step, summed over every k_poa_tile1 dispatch of the step; rows = DP rows the kernels counted in the same run (poa_dp_rows); pipe_busy = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES of a second counter pass"}
    json.dump(j,open("gpurun_out/r6/r06_pmc_poa_tile.json","w"),indent=1); print(json.dumps({k:v for k,v in j.items() if k not in ("raw","_how")}))
PY
python - <<PY
# the clustering aligner from the same SQ pass: VALU instructions of every k_sg_align* dispatch of the step / the DP cells the kernels counted in the bench line (per step)
import csv,glob,collections,json,os
commit=os.environ.get("NGSID_COMMIT","unknown")
fs=glob.glob("gpurun_out/r6/pmc_sq/**/*counter_collection.csv",recursive=True)
sq=collections.defaultdict(float)
if fs:
    for r in csv.DictReader(open(fs[0])):
        if "k_sg_align" in r["Kernel_Name"]: sq[r["Counter_Name"]]+=float(r["Counter_Value"])
try: pb2=json.load(open("gpurun_out/r6/r06_pmc_pipe_busy.json"))
except Exception: pb2={}
try:
    d=json.loads(open("gpurun_out/r6/r06_bench_1m.json").read().strip().splitlines()[-1]); v=d["roofline"]["dp_kernels"]["k_sg_align"]; cells=v["dp_cells"]/d["steps"]
except Exception: cells=0
kt=glob.glob("gpurun_out/r6/pmc_sq/**/*kernel_trace.csv",recursive=True); dur=0
if kt:
    for r in csv.DictReader(open(kt[0])):
        if "k_sg_align" in r["Kernel_Name"]: dur+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
if sq and cells:
    j={"workload_reads":1000000,"config":"c3","commit":commit,"cells":int(cells),"valu_per_cell":round(sq["SQ_INSTS_VALU"]*64/cells,2),"salu_per_cell":round(sq["SQ_INSTS_SALU"]*64/cells,2),
       "wave_cycles_waiting_frac":round(sq["SQ_WAIT_ANY"]/max(sq["SQ_WAVE_CYCLES"],1),3),"kernel_ns_under_the_counter_pass":dur,
       "pipe_busy_assuming_4_cycles_per_instruction":round(sq["SQ_INSTS_VALU"]*4/(1024*(sq["SQ_BUSY_CYCLES"]/32.0)),3) if sq.get("SQ_BUSY_CYCLES") else None,
       "pipe_busy":pb2.get("active_valu_over_busy",{}).get("k_sg_align"),"raw_second_pass":pb2.get("raw",{}).get("k_sg_align"),"raw":{k:int(v) for k,v in sq.items()},
       "_how":"same counter pass as r06_pmc_poa_tile.json, summed over every k_sg_align* dispatch (clustering + the reverse-complement merge) of one bench step; cells = DP cells (query x target bases) the kernels counted per step in the bench line; valu_per_cell = wave VALU instructions x 64 lanes / cells"}
    json.dump(j,open("gpurun_out/r6/r06_pmc_sg_align.json","w"),indent=1); print(json.dumps({k:v for k,v in j.items() if k not in ("raw","_how")}))
PY
python -c "
import json; d=json.loads(open('gpurun_out/r6/r06_bench_1m.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['stage_s_per_step'], d['config']['kernel_ms_per_step'], d['roofline'], d['config'].get('cli',{}).get('reads_per_s'), d['cpu_baseline']['value'])"
