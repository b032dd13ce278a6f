import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
# take the last step: group by kernel name + grid
agg=collections.OrderedDict()
for r in rows:
    name=r["Kernel_Name"].split("(")[0].replace("void vae::","")
    key=(name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r.get("Workgroup_Size_X",""))
    d=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    a=agg.setdefault(key,[0,0]); a[0]+=1; a[1]+=d
tot=sum(a[1] for a in agg.values())
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 40]:
    print(f"{a[1]/a[0]/1e3:9.1f} us x{a[0]:4d}  {100*a[1]/tot:5.1f}%  grid=({k[1]},{k[2]},{k[3]}) wg={k[4]}  {k[0][:90]}")
