import csv
rows=list(csv.DictReader(open("gpurun_out/one/p_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "syn_overlap" in r["Kernel_Name"]]
a=idx[-2]+1; b=idx[-1]+1
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%8.1f %7.1f q%s %-52s grid %s" % ((s-t0)/1e3,(e-s)/1e3,r["Queue_Id"],r["Kernel_Name"].replace("void ","").replace("wc::","")[:52],r["Grid_Size_X"]))
