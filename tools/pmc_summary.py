"""Merge rocprofv3 counter passes (one directory per pass: --pmc FETCH_SIZE / WRITE_SIZE / SQ_*) of
tools/steps.py into profiles/<round>_pmc_summary.json: per-kernel averages per launch.
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE (KB) is doubled on gfx950 (128-B requests are
tallied at 64 B), WRITE_SIZE (KB) is taken as is.
   python tools/pmc_summary.py out.json blocks_per_launch dir1 dir2 ..."""
import csv, glob, collections, json, os, re, sys

out, nblk, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = k.split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in agg.items():
    if "hint_" not in k:
        continue
    e = {"launches": max(len(x) for x in v.values()), "blocks_per_launch": nblk if any(t in k for t in ("hint_apply", "hint_bwd", "hint_wl_", "hint_wgrad", "hint_wreduce")) else None}
    sq = {}
    for c, x in sorted(v.items()):
        m = sum(x) / len(x)
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            e[c + "_KB"] = m
        else:
            sq[c] = m
    if "FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e:
        e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024
        e["note"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)"
    if sq:
        e["sq"] = sq
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "SQ_BUSY_CU_CYCLES" in sq and sq["SQ_BUSY_CU_CYCLES"] > 0:
            e["mfma_busy_frac_of_cu_busy"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * sq["SQ_BUSY_CU_CYCLES"])
    res[k] = e
# which binary the counters belong to: the library's own build string (its source stamp is a hash of hint_amd/csrc + include, and the
# HINT_* knobs set in the profiled process ride behind it) and the commit the tree was at (GIT_REV: .git does not travel to the GPU box)
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from hint_amd import _lib
    info = _lib.load().hint_build_info().decode()
except Exception as e:      # noqa: BLE001
    info = f"unavailable ({type(e).__name__})"
m = re.search(r"src ([0-9a-f]+)", info)
res["_build"] = {"hint_build_info": info, "src_stamp": m.group(1) if m else None, "git_head": os.environ.get("GIT_REV", "unknown")}
json.dump(res, open(out, "w"), indent=1)
for k, e in res.items():
    if k == "_build":
        continue
    print(k, {a: b for a, b in e.items() if a not in ("sq", "note")})
