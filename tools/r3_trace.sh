#!/bin/bash
# kernel trace of one bench step (gaps between kernels): gpurun_out/r3_trace/
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-extra-step > $GRAFT_REPO_ROOT/gpurun_out/r3_trace.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/r3_trace | head
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3_trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step = after the last k_score? take last 45% of timeline: find last k_count_hits start sequence; simpler: print gaps > 150us in the last 1.1 s
tend=max(int(r['End_Timestamp']) for r in rows)
sel=[r for r in rows if int(r['Start_Timestamp'])>tend-1.05e9]
prev=None; tot=0
out=[]
for r in sel:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if prev is not None and s-prev>100000:
        out.append(((s-prev)/1e6, pn, r['Kernel_Name'][:40], (s-(tend-1.05e9))/1e6)); tot+=s-prev
    if prev is None or e>prev: prev=e; pn=r['Kernel_Name'][:40]
print("gaps >0.1ms total %.1f ms"%(tot/1e6))
for g in out: print("gap %.2f ms after %-40s before %-40s at %.1f"%g)
PY
