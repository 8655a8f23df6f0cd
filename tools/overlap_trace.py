import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id"), r.get("Stream_Id"), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", ""), r.get("LDS_Block_Size", "")) for r in rows]
ks.sort()
c1 = [k for k in ks if "c1_wgrad" in k[2]]
print("c1_wgrad launches:", len(c1))
for n, k in enumerate(c1[:3]):
    print("=== c1_wgrad", n, "start", k[0], "dur us", (k[1] - k[0]) / 1e3, "queue", k[3], "stream", k[4])
    for o in ks:
        if o is k: continue
        if o[1] > k[0] - 30000 and o[0] < k[1] + 5000:
            print("   %-70s q=%s s=%s  start %+9.1f us  end %+9.1f us  grid %s lds %s" % (o[2], o[3], o[4], (o[0] - k[0]) / 1e3, (o[1] - k[0]) / 1e3, o[5], o[7]))
