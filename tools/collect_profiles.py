"""Copies the summaries of tools/profile_round.sh from gpurun_out/<tag>/ into profiles/<tag>/ (tracked):
    python tools/collect_profiles.py <tag> <commit the box ran> [scene ...]
Only the scenes named (all three without names) get their counter summaries rewritten and stamped with the commit."""
import sys, os, glob, shutil, csv, collections, json
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
commit = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("SOLR_COMMIT", "unknown")   # the tree the box ran
SCENES = tuple(sys.argv[3:]) or ("cornell", "height_field", "molecule")
src, dst = "gpurun_out/" + tag, "profiles/" + tag
os.makedirs(dst, exist_ok=True)
for scene in SCENES:
    b = os.path.join(src, "bench_%s.json" % scene)
    if os.path.exists(b):
        shutil.copy(b, os.path.join(dst, "bench_%s.json" % scene))
    st = os.path.join(src, "trace_%s" % scene, "trace_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(dst, "kernel_stats_%s.csv" % scene))
    out = []
    for p in ("sq", "sq2", "fetch", "write"):
        agg = collections.defaultdict(list)
        meta = None
        for f in glob.glob(os.path.join(src, "pmc_%s_%s" % (p, scene), "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if ("k_standardRenderer<0," in r["Kernel_Name"] or "k_standardRenderer<false" in r["Kernel_Name"]):
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"])); meta = r
        for k in sorted(agg):
            out.append("%-22s %16.0f   (mean of %d launches)" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
        if meta and p == "sq":
            out.append("kernel %s  grid %s  workgroup %s  scratch %s B/lane  LDS %s B/workgroup" % (
                meta["Kernel_Name"][:60], meta["Grid_Size"], meta["Workgroup_Size"], meta["Scratch_Size"], meta["LDS_Block_Size"]))
    if out:
        hdr = ("# commit %s\n" % commit +
               "# rocprofv3 --pmc, separate passes (two sets of SQ counters; FETCH_SIZE; WRITE_SIZE) of\n"
               "#   SOLR_BENCH_REGIONS=3 python3 bench.py %s --warmup 12 --no-cpu-baseline --no-walk-bound --frames-in-flight 1\n"
               "# per launch of the renderer kernel.  FETCH_SIZE / WRITE_SIZE are in KB (L2 <-> fabric requests x 64 B;\n"
               "# MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads by 2x on gfx950, WRITE_SIZE is exact).\n"
               % ("--config cfg4 --steps 74" if scene == "cfg4" else "--scene %s --steps 24" % scene))
        open(os.path.join(dst, "pmc_%s.txt" % scene), "w").write(hdr + "\n".join(out) + "\n")
traffic = {}
if os.path.exists(os.path.join(dst, "hbm_traffic.json")):
    traffic = json.load(open(os.path.join(dst, "hbm_traffic.json")))   # the scenes not named keep their entries
for scene in SCENES:
    vals = {}
    for p, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        v = []
        for f in glob.glob(os.path.join(src, "pmc_%s_%s" % (p, scene), "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if ("k_standardRenderer<0," in r["Kernel_Name"] or "k_standardRenderer<false" in r["Kernel_Name"]) and r["Counter_Name"] == name:
                    v.append(float(r["Counter_Value"]))
        if v:
            vals[name] = sum(v) / len(v)
    if len(vals) == 2:
        traffic[scene] = {"workload": "%s 1920x1080" % scene if scene != "cfg4" else "cfg4: Cornell 3840x2160, mean over the launches of passes 0...73",
                          "commit": commit, "FETCH_SIZE_KB": round(vals["FETCH_SIZE"], 1),
                          "WRITE_SIZE_KB": round(vals["WRITE_SIZE"], 1),
                          "bytes_per_launch": int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
                          "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KB -> bytes; "
                                  "FETCH_SIZE taken at face value (the gfx950 2x under-count is calibrated for 16 B/lane "
                                  "streaming reads only; the reads here are dword scratch reloads and scalar loads)"}
if traffic:
    json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "tile_timeline_*.txt")) + glob.glob(os.path.join(src, "strip_times_*.txt")) + \
        glob.glob(os.path.join(src, "reference_opencl_speed.txt")) + glob.glob(os.path.join(src, "valu_issue_bench.txt")) + \
        glob.glob(os.path.join(src, "wave_time_split_*.txt")) + \
        glob.glob(os.path.join(src, "upload_time.txt")) + glob.glob(os.path.join(src, "group_sweep.txt")):
    shutil.copy(f, dst)
print(sorted(os.listdir(dst)))
