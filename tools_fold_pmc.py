import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    name=r["Kernel_Name"].split("(")[0].replace("void vae::","")[:60]
    agg[name][r["Counter_Name"]]+=float(r["Counter_Value"]); agg[name]["_n"]+=1
ctrs=sorted({c for a in agg.values() for c in a if c!="_n"})
print("kernel".ljust(60), " ".join(c[-18:].rjust(18) for c in ctrs))
for k,a in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))[:int(sys.argv[2]) if len(sys.argv)>2 else 14]:
    print(k.ljust(60), " ".join(f"{a.get(c,0):18.3e}" for c in ctrs))
