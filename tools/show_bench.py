#!/usr/bin/env python3
"""Development tool: the headline fields of a bench.py output file. usage: tools/show_bench.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f  ms/step %.3f  whole_pass %s" % (d["value"], d["ms_per_step"], d.get("whole_pass", {}).get("value")))
for k in ("reference_text_paths", "lu_plus_solve", "roofline", "hbm_roofline"):
    if k in d:
        print(k, json.dumps(d[k])[:700])
if "kernel_classes_rank0" in d:
    for k, v in d["kernel_classes_rank0"].items():
        print("  %-12s %s" % (k, json.dumps(v)))
