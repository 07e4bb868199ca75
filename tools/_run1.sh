cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4/ed; mkdir -p $O
for V in base notb; do
LIB=$R/ngspeciesid_amd/libngsid_hip.so; [ $V = notb ] && LIB=$R/build_alt/libngsid_hip_notb.so
rm -rf $O/q$V; timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/q$V -o pmc -- python $R/tools/micro/time_ed2.py $LIB 400000 > $O/q$V.log 2>&1
python - <<PY
import csv,glob,collections
fs=glob.glob("$O/q$V/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(float); 
for r in csv.DictReader(open(fs[0])):
    if "k_ed_align<8, true>" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
kt=glob.glob("$O/q$V/**/*kernel_trace.csv",recursive=True)
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in csv.DictReader(open(kt[0])) if "k_ed_align<8, true>" in r["Kernel_Name"]]
nb=3*2*6250   # 3 calls x (main + retry launches share the name) ~ bundles of the main launches
print("$V", {k:int(v) for k,v in agg.items()}, "launch ms", [round(x,2) for x in d])
print("$V per bundle-column (6250 bundles x 750 columns x 3 calls):", {k: round(v/(3*6250*750),1) for k,v in agg.items() if k.startswith("SQ_INSTS")})
PY
done
