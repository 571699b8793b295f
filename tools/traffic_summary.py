#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from rocprofv3 FETCH_SIZE / WRITE_SIZE passes (tools/traffic.sh).
gfx950 correction (MI355X_MICROARCH.md, HBM): counters are in KiB; FETCH_SIZE tallies 128-B requests at 64 B,
so reads are doubled; WRITE_SIZE is taken as is.  Prints a table and, with --json KEY PREFIX, the mean bytes per launch of the kernels named PREFIX*."""
import collections
import csv
import glob
import json
import re
import sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"(k_\w+)<(.*)>\(", name)
            short = (m.group(1) + "<" + re.sub(r"qi::native::\(anonymous namespace\)::", "", m.group(2)) + ">") if m else name[:60]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    d = sys.argv[1]
    agg = load(d)
    rows = {}
    for k, cs in sorted(agg.items()):
        rd = 2.0 * 1024.0 * sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [0])), 1)
        wr = 1024.0 * sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [0])), 1)
        rows[k] = (rd, wr, len(cs.get("WRITE_SIZE", [])))
        print(f"{k[:100]:100s} read {rd/1e6:10.2f} MB  write {wr/1e6:10.2f} MB  launches {rows[k][2]}")
    if "--json" in sys.argv:  # --json KEY PREFIX: mean HBM bytes per launch over the kernels whose name starts with PREFIX
        key, prefix = sys.argv[sys.argv.index("--json") + 1], sys.argv[sys.argv.index("--json") + 2]
        sel = [v for k, v in rows.items() if k.startswith(prefix)]
        tot = sum((rd + wr) * n for rd, wr, n in sel) / max(sum(n for _, _, n in sel), 1)
        print(json.dumps({key: int(tot)}))


if __name__ == "__main__":
    main()
