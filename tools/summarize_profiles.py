#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of a bench run (kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes) into the per-kernel
summary committed under profiles/. HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB of 64-B requests; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced
streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B-per-lane stores.
An optional SQ-counter pass (SQ_* of the same command) is folded in per kernel: busy fractions of the vector ALU and the LDS,
wait fractions, so that "what bounds this kernel" can be read from the committed file.
usage: summarize_profiles.py <stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> <bench.json> <out.json> [sq_counter_collection.csv ...]"""
import collections, csv, json, sys

stats, fetch, write, bench, out = sys.argv[1:6]
sq_files = sys.argv[6:]
short = lambda n: n.split("(")[0].replace("void ", "").replace("idahip::", "")
k = {}
for r in csv.DictReader(open(stats)):
    k[short(r["Name"])] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_us": float(r["AverageNs"]) / 1e3,
                           "pct": float(r["Percentage"])}
def pmc(path, name):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
            cnt[short(r["Kernel_Name"])] += 1
    return acc, cnt
f, fc = pmc(fetch, "FETCH_SIZE")
w, wc = pmc(write, "WRITE_SIZE")
for name, d in k.items():
    if name in f:
        d["hbm_read_GB_total"] = round(2.0 * f[name] * 1024 / 1e9, 3)   # gfx950 correction: x2
        d["hbm_write_GB_total"] = round(w.get(name, 0.0) * 1024 / 1e9, 3)
        d["hbm_bytes_per_launch"] = int((2.0 * f[name] + w.get(name, 0.0)) * 1024 / max(1, fc[name]))
        d["hbm_GBps"] = round((2.0 * f[name] + w.get(name, 0.0)) * 1024 / 1e9 / (d["total_ms"] / 1e3), 1)
sq = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sq_files:
    for r in csv.DictReader(open(path)):
        sq[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
for name, c in sq.items():
    if name not in k:
        continue
    d = {kk: vv for kk, vv in c.items()}
    wc_, bc_ = c.get("SQ_WAVE_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0)
    der = {}
    # SQ_ACTIVE_INST_* advance by one per four cycles in which a SIMD issues to the unit; SQ_BUSY_CYCLES sums the busy cycles of
    # the 32 shader engines: VALU busy per SIMD = ratio / 8, LDS busy per CU = ratio / 2 (profiles/README.md, round 3)
    if bc_ > 0:
        for cn in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY"):
            if cn in c:
                der[cn + "/SQ_BUSY_CYCLES"] = round(c[cn] / bc_, 4)
    if wc_ > 0:
        for cn in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
            if cn in c:
                der[cn + "/SQ_WAVE_CYCLES"] = round(c[cn] / wc_, 4)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        der["SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    k[name]["sq_counters"] = {kk: float("%.6g" % vv) for kk, vv in sorted(d.items())}
    k[name]["sq_derived"] = der
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib.util
_spec = importlib.util.spec_from_file_location("bench_for_sha", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
_bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bench)
b = json.loads(open(bench).read().strip().splitlines()[-1])
lu_mats = b["kernel_classes_rank0"]["lu"]["systems"]
# traffic against the algorithmic bytes, per kernel, COMPUTED here (not transcribed into a document): the byte counts are
# bench.py's, the systems / matrices served are the ones the bench line of the profiled run reports. The trace
# also holds the launches of the stagger calibration (128 systems of the first group), which the bench line's class counters do
# not: the ratio of a kernel is therefore high by the share of those launches (about 1 % in the default run).
_n = b.get("config", {}).get("n", 512)
_ab = _bench.algorithmic_bytes(_n, "linear_dense")
_cls = b["kernel_classes_rank0"]
_alg = {}
for name in k:
    if name.startswith("newton_iter_kernel"):
        _alg[name] = _ab["newton_iter"] * _cls["newton_iter"]["systems"]
    elif name.startswith("linear_sys_kernel") and "true" in name:
        _alg[name] = _ab["sys_jac"] * _cls["sys_jac"]["systems"]
    elif name.startswith("linear_sys_kernel"):
        _alg[name] = _ab["sys"] * _cls["sys"]["systems"]
    elif name.startswith("lu_trail64w_kernel"):
        _alg[name] = _bench.trailing_work(_n)[1] * lu_mats
    elif name.startswith("lu_finalize_kernel"):
        _alg[name] = 16.0 * _n * _n * lu_mats  # the matrix read and written once (U rows right of their panel are already in place: a little less)
_wp = [name for name in k if name.startswith("lu_wavepanel_kernel") and "hbm_read_GB_total" in k[name]]
for name, d in k.items():
    if name in _alg and "hbm_read_GB_total" in d and _alg[name] > 0:
        d["algorithmic_GB_total"] = round(_alg[name] / 1e9, 3)
        d["traffic_over_algorithmic"] = round((d["hbm_read_GB_total"] + d["hbm_write_GB_total"]) * 1e9 / _alg[name], 3)
_panels = None
if _wp:
    _palg = sum(16.0 * (_n - k0) * min(64, _n - k0) for k0 in range(0, _n, 64)) * lu_mats  # every super-panel read and written once
    _pt = sum(k[name]["hbm_read_GB_total"] + k[name]["hbm_write_GB_total"] for name in _wp) * 1e9
    _panels = {"algorithmic_GB_total": round(_palg / 1e9, 3), "traffic_over_algorithmic": round(_pt / _palg, 3),
               "total_ms": round(sum(k[name]["total_ms"] for name in _wp), 3)}
res = {"commit": os.environ.get("GIT_COMMIT", "unknown"), "kernel_sources_sha": _bench.kernel_sources_sha(),
       "bench": {**{kk: b[kk] for kk in ("value", "steps", "warmup", "ms_per_step", "kernel_classes_rank0") if kk in b},
                 "lu_kernels_rank0": b.get("lu_kernels_rank0"), "lu_matrices": lu_mats},
       "kernels": k,
       "lu_wavepanel_kernels_together": _panels,
       "lu_hbm_bytes_per_matrix": {name: int((d["hbm_read_GB_total"] + d["hbm_write_GB_total"]) * 1e9 / max(1, lu_mats))
                                   for name, d in k.items() if name.startswith("lu_") and "hbm_read_GB_total" in d},
       "note": "kernel stats and PMC passes are separate runs of the same command (python3 bench.py --steps 40 --warmup 0 "
               "--no-cpu-baseline --no-extras under IDAHIP_BENCH_TIME_ALL=1: the per-kernel LU timers cover every launch of the process)"}
res["lu_hbm_bytes_per_matrix"]["all LU kernels"] = sum(res["lu_hbm_bytes_per_matrix"].values())
json.dump(res, open(out, "w"), indent=1)
for name, d in sorted(k.items(), key=lambda x: -x[1]["total_ms"])[:10]:
    print("%-28s calls %5d total %9.2f ms avg %9.1f us  %s" % (name[:28], d["calls"], d["total_ms"], d["avg_us"],
          ("read %.1f GB write %.1f GB -> %.0f GB/s" % (d["hbm_read_GB_total"], d["hbm_write_GB_total"], d["hbm_GBps"])) if "hbm_GBps" in d else ""))
print(b["kernel_classes_rank0"])
