#!/usr/bin/env python3
"""Print the headline figures of bench.py JSON lines (tools/show_bench.py file.json ...)."""
import json
import sys


def show(x, ind=0):
    pad = " " * ind
    r = x["roofline"]
    print(f"{pad}{x['config']['workload'][:60]}")
    print(f"{pad}  value {x['value']:.0f}  ms/step {x['ms_per_step']}  step_frac {x['step_roofline'].get('frac')}")
    vi = r.get("valu_issue")
    print(f"{pad}  dominant {r['kernel']} frac {r['frac']} launch_ms {r.get('launch_ms')}" + (f" valu_issue {vi['frac']}" if vi else ""))
    if "stage_ms_per_step" not in x["step_roofline"]:
        return
    for k, v in x.get("stage_rooflines", {}).items():
        vi = v.get("valu_issue")
        print(f"{pad}  stage {k} frac {v['frac']} launch_ms {v['launch_ms']} x{v['launches_per_step']}" + (f" valu_issue {vi['frac']}" if vi else ""))
    print(f"{pad}  stages {x['step_roofline']['stage_ms_per_step']}")
    if "stft" in x:
        print(f"{pad}  stft {x['stft']['ms_per_step']} ms frac {x['stft']['frac']}")
    if "two_streams" in x:
        print(f"{pad}  two_streams {x['two_streams']['value']}")


for f in sys.argv[1:]:
    for ln in open(f):
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        print("==", f)
        show(d)
        for k in ("configs2", "f64", "configs4", "configs1_per_gpu"):
            if k in d:
                print(" ", k)
                show(d[k], 4)
