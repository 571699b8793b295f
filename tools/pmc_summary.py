#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter per dispatch."""
import csv, glob, re, sys, collections
def main(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                m = re.search(r"(k_\w+)<([^>]*(?:<[^>]*>[^>]*)*)>", name)
                short = (m.group(1) + "<" + m.group(2)[:60] + ">") if m else name.split("(")[0][-60:]
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[short]["_vgpr"].append(float(r.get("VGPR_Count", 0) or 0))
                agg[short]["_lds"].append(float(r.get("LDS_Block_Size", 0) or 0))
                agg[short]["_scratch"].append(float(r.get("Scratch_Size", 0) or 0))
    for k, cs in agg.items():
        if not any(s in k for s in sys.argv[1].split(",")) and sys.argv[1] != "all":
            continue
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
if __name__ == "__main__":
    main(sys.argv[2:])
