#!/bin/bash
# configs[4] streaming sample: vector instructions and HBM bytes per streamed item (16 records x 2^20, order 12, float64),
# summed over every kernel of the run and divided by the items (separate --pmc passes; FETCH_SIZE x 2, KiB units)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-cfg4_pmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
A="python3 $GRAFT_REPO_ROOT/bench.py --config 4 --cpu-seconds 0 --stream-chunks 9 --warmup 2"
for ctr in SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$ctr -- $A > $out/$ctr.log 2>&1
  echo "$ctr rc=$?"
done
cd $GRAFT_REPO_ROOT
python - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
tot = {}
names = {}
for ctr in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{ctr}/**/*_counter_collection.csv", recursive=True)[-1]
    s = 0.0
    per = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("k_band_support", "k_bank_rows", "k_copy_window", "k_block_taps", "k_stx_window", "twiddle_gen", "k_bank_convert", "k_block_rotate")):
            continue  # plan construction, not the streamed items
        v = float(r["Counter_Value"])
        s += v
        key = n.split("(")[0][-48:]
        per[key] = per.get(key, 0.0) + v
    tot[ctr] = s
    names[ctr] = per
line = json.loads([l for l in open(f"{out}/SQ_INSTS_VALU.log") if l.startswith("{")][-1])
items = 9  # every item of the run ran under the counters (2 warm-up + 7 timed); plan-time kernels excluded above
res = {"items": items, "valu_wave_insts_per_item": tot["SQ_INSTS_VALU"] / items,
       "hbm_bytes_per_item": (2 * 1024 * tot["FETCH_SIZE"] + 1024 * tot["WRITE_SIZE"]) / items,
       "read_bytes_per_item": 2 * 1024 * tot["FETCH_SIZE"] / items, "write_bytes_per_item": 1024 * tot["WRITE_SIZE"] / items}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res))
top = sorted(names["SQ_INSTS_VALU"].items(), key=lambda kv: -kv[1])[:12]
with open(f"{out}/valu_by_kernel.txt", "w") as fh:
    for k, v in top:
        fh.write(f"{k:50s} {v / items / 1e6:10.2f} M wave instructions per item\n")
        print(f"{k:50s} {v / items / 1e6:10.2f} M wave instructions per item")
PY
rm -rf $out/SQ_INSTS_VALU $out/FETCH_SIZE $out/WRITE_SIZE
