#!/bin/bash
# GPU box: memory-system counters of the bench step (C3, one step), per kernel class -> gpurun_out/r3m/.  One rocprofv3 run per counter set
# (counters never together with traces other than --kernel-trace).  Counter names are taken from what the box lists.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || rocprofv3 -L > $O/avail.txt 2>&1
BARGS="--no-cpu-baseline --no-cli --no-extra-step"
have() { grep -q -w "$1" $O/avail.txt; }
pass() {   # name, counters...
  n=$1; shift; sel=""
  for c in "$@"; do if have $c; then sel="$sel $c"; fi; done
  [ -z "$sel" ] && { echo "pass $n: no counter available" >> $O/passes.log; return; }
  echo "pass $n:$sel" >> $O/passes.log
  rm -rf $O/pmc_$n; timeout 600 rocprofv3 --pmc $sel --kernel-trace --output-format csv -d $O/pmc_$n -o pmc -- python $R/bench.py --steps 1 --warmup 0 $BARGS > $O/pmc_$n.log 2>&1
}
: > $O/passes.log
pass l2a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass l2b TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum
pass l2c TCC_WRITE_sum TCC_ATOMIC_sum TCC_EA_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass l1a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass l1b TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
pass gr GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES SQ_WAVES
cd $R
python - <<'PY'
import csv,glob,collections,json
def cls(nm):
    for k in ("k_sg_align","k_poa_tile","k_ed_align","k_hpc_minimizers"):
        if k in nm: return k
    return None
out=collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/r3m/pmc_*")):
    if not d.split("/")[-1].startswith("pmc_") or d.endswith(".log"): continue
    fs=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k=cls(r["Kernel_Name"])
        if k: out[k][r["Counter_Name"]]=out[k].get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
    kt=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)
    if kt:
        dur=collections.defaultdict(int)
        for r in csv.DictReader(open(kt[0])):
            k=cls(r["Kernel_Name"])
            if k: dur[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
        for k,v in dur.items(): out[k]["ns_"+d.split("pmc_")[-1]]=v
json.dump(out,open("gpurun_out/r3m/r03_memsys.json","w"),indent=1)
for k,v in out.items(): print(k, {a:(int(b) if b>100 else round(b,3)) for a,b in sorted(v.items())})
PY
