mkdir -p gpurun_out; rm -f gpurun_out/*.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
NUTT=2 rocprofv3 --kernel-trace -d $R/gpurun_out/one -o p --output-format csv -- python $R/tools/_one.py > $R/gpurun_out/one.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/one/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'syn_overlap' in r['Kernel_Name']]
a,b=idx[-3]+1,idx[-1]+1
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    if e-s>60000: print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} q{r['Queue_Id']} {r['Kernel_Name'][:60]:60} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
PY
