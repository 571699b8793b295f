#!/usr/bin/env python3
"""profiles/traffic.json from one round's counter passes (tools/profile_round.sh): per bench leg and profiling stage the HBM
bytes (TCC: 2 x FETCH_SIZE + WRITE_SIZE, KiB units -- MI355X guide, HBM section) and the vector wave instructions
(SQ_INSTS_VALU) of the stage's kernels per STAGE SPAN (= what bench.py calls a launch of the stage: one per run of a table, or
one per joint tile), and per streamed item for configs[4].
usage: tools/make_traffic_json.py <gpurun_out/prof_rNN> <round tag, e.g. r05> > profiles/traffic.json"""
import collections
import csv
import glob
import json
import os
import re
import sys

STAGES = {"zoom": ("k_zoom2", "k_zoom<", "k_z64_fine", "k_z64_interp", "k_z64_mfma"), "block": ("k_block",)}
TAILS = ("k_tail2", "k_tail<")  # one dispatch per stage span (a joint tile's tail, or a table run's)
LEGS = {"cfg1": ("f32", 20, 3, 1), "cfg2": ("f32", 20, 12, 64), "f64": ("f64", 20, 12, 4)}


def counters(d, ctr):
    per = collections.defaultdict(float)
    count = collections.Counter()
    files = sorted(glob.glob(os.path.join(d, ctr, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        return None, None
    for r in csv.DictReader(open(files[-1])):  # (the newest pass)
        if r["Counter_Name"] != ctr:
            continue
        m = re.search(r"(k_\w+<?)", r["Kernel_Name"])
        key = m.group(1) if m else r["Kernel_Name"][:40]
        per[key] += float(r["Counter_Value"])
        count[key] += 1
    return per, count


def main(root, tag):
    out = {"note": "per stage span (bench.py's launch of a stage): HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE in KiB units, vector wave "
                   "instructions = SQ_INSTS_VALU; separate rocprofv3 --pmc passes of the bench's own command per leg "
                   "(tools/profile_round.sh, tools/make_traffic_json.py)"}
    srcs = []
    for leg, (dt, log2n, order, ch) in LEGS.items():
        d = os.path.join(root, "pmc_" + leg)
        fetch, cnt = counters(d, "FETCH_SIZE")
        write, _ = counters(d, "WRITE_SIZE")
        valu, vcnt = counters(d, "SQ_INSTS_VALU")
        if fetch is None or write is None:
            continue
        spans = sum(c for k, c in cnt.items() if k.startswith(TAILS))
        for stage, prefixes in STAGES.items():
            keys = [k for k in fetch if k.startswith(prefixes)]
            if not keys or not spans:
                continue
            b = sum(2.0 * 1024.0 * fetch[k] + 1024.0 * write.get(k, 0.0) for k in keys) / spans
            out[f"{stage}:{dt}:n{log2n}:o{order}:c{ch}"] = int(b)
            if valu is not None:
                vspans = sum(c for k, c in vcnt.items() if k.startswith(TAILS))
                out[f"valu:{stage}:{dt}:n{log2n}:o{order}:c{ch}"] = int(sum(valu[k] for k in valu if k.startswith(prefixes)) / max(vspans, 1))
        srcs.append(f"profiles/{tag}_{leg}_traffic_summary.txt")
    out["source"] = ", ".join(srcs) + f" (round {tag[1:]}, final build: FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes of every leg)"
    out["source_valu"] = out["source"]
    sj = os.path.join(root, "cfg4_pmc", "summary.json")
    if os.path.exists(sj):
        s = json.load(open(sj))
        out["valu_wave_insts_per_item:f64:n20:o12:c16:stream"] = int(s["valu_wave_insts_per_item"])
        out["item:f64:n20:o12:c16:stream"] = int(s["hbm_bytes_per_item"])
        out["source_f64"] = f"profiles/{tag}_cfg4_stream_pmc_summary.json"
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
