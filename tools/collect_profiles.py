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
# FETCH_SIZE / WRITE_SIZE against known byte counts in the renderer's own access patterns (tools/fetch_calibration.sh ->
# profiles/<tag>/fetch_calibration.txt): 16-B-per-lane reads of the per-pixel records (ids, the two halves of the float
# frame buffer, the depth word of a record) are reported at HALF their bytes on gfx950, scalar loads of 32-B node /
# primitive rows at the bytes of the 64-B lines they bring in (exact), the stores of a pass 5.9 % high (the RGB image
# goes out in single bytes).  A first pass reads nothing but the scene (scalar loads: face value); a refinement /
# accumulation pass reads 48 B per pixel of ids and frame buffer against a 9 KB scene (all of it the halved kind: x 2).
calib = {}
calib_file = os.path.join(dst, "fetch_calibration.txt")
if os.path.exists(os.path.join(src, "fetch_calibration.txt")):
    shutil.copy(os.path.join(src, "fetch_calibration.txt"), calib_file)
if os.path.exists(calib_file):
    for line in open(calib_file):
        w = line.split()
        if len(w) == 5 and w[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            calib[w[0]] = float(w[4])
VECTOR_READS = calib.get("read_pp_two_float4_per_pixel", 2.0)
SCALAR_LINES = 1.0           # (read_scalar_32_byte_records: 0.507 of the USEFUL bytes = the 64-B lines, exactly)
STORES = calib.get("write_ids_pp_rgb_per_pixel", 1.0)


def per_kernel(scene, kernel):
    vals = {}
    for p, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        v = []
        for f in glob.glob(os.path.join(src, "pmc_%s_%s" % (p, scene), "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if any(k in r["Kernel_Name"] for k in kernel) and r["Counter_Name"] == name:
                    v.append(float(r["Counter_Value"]))
        if v:
            vals[name] = sum(v) / len(v)
    return vals


traffic = {}
if os.path.exists(os.path.join(dst, "hbm_traffic.json")):
    traffic = json.load(open(os.path.join(dst, "hbm_traffic.json")))   # the scenes not named keep their entries
for scene in SCENES:
    vals = per_kernel(scene, ("k_standardRenderer<0,", "k_standardRenderer<false"))
    if len(vals) == 2:
        fetch_factor = VECTOR_READS if scene == "cfg4" else SCALAR_LINES
        corrected = vals["FETCH_SIZE"] * 1024 * fetch_factor + vals["WRITE_SIZE"] * 1024 * STORES
        traffic[scene] = {"workload": "%s 1920x1080" % scene if scene != "cfg4" else "cfg4: Cornell 3840x2160, mean over the launches of passes 0...73",
                          "commit": commit, "FETCH_SIZE_KB": round(vals["FETCH_SIZE"], 1),
                          "WRITE_SIZE_KB": round(vals["WRITE_SIZE"], 1),
                          "fetch_factor": fetch_factor, "write_factor": round(STORES, 4),
                          "bytes_per_launch_at_face_value": int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
                          "bytes_per_launch": int(corrected),
                          "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KB -> bytes, corrected by "
                                  "the factors of fetch_calibration.txt (known byte counts in the renderer's own access "
                                  "patterns): " + ("the reads of a refinement / accumulation pass are 16-B-per-lane loads of "
                                                   "the per-pixel records, which gfx950 reports at half their bytes (x 2)"
                                                   if scene == "cfg4" else
                                                   "a first pass reads only the scene, by scalar loads, which are reported at "
                                                   "the bytes of their 64-B lines (x 1)") +
                                  "; the stores of a pass are reported 5.9 % high (single-byte stores of the RGB image)"}
    if scene == "cfg4":
        ao = per_kernel(scene, ("k_ambientOcclusion",))
        if len(ao) == 2:
            traffic["cfg4_k_ambientOcclusion"] = {
                "workload": "k_ambientOcclusion behind every pass of cfg4, 3840x2160", "commit": commit,
                "FETCH_SIZE_KB": round(ao["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(ao["WRITE_SIZE"], 1),
                "fetch_factor": VECTOR_READS, "write_factor": 1.0,
                "bytes_per_launch": int(ao["FETCH_SIZE"] * 1024 * VECTOR_READS + ao["WRITE_SIZE"] * 1024),
                # what the kernel must move: the colour and depth of every pixel (a 32-B record each: the depth word alone
                # brings the record's line in, fetch_calibration.txt read_depth_4_of_32_bytes), the RGB image out
                "algorithmic_bytes": 3840 * 2160 * (32 + 3),
                "note": "reads: the records' 16-B halves (x 2, as calibrated); writes: the RGB image"}
if traffic:
    json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "tile_timeline_*.txt")) + glob.glob(os.path.join(src, "strip_times_*.txt")) + \
        glob.glob(os.path.join(src, "reference_opencl_speed.txt")) + glob.glob(os.path.join(src, "valu_issue_bench.txt")) + \
        glob.glob(os.path.join(src, "wave_time_split_*.txt")) + \
        glob.glob(os.path.join(src, "upload_time.txt")) + glob.glob(os.path.join(src, "group_sweep.txt")) + \
        glob.glob(os.path.join(src, "api_frame_cornell.txt")) + glob.glob(os.path.join(src, "strip_balance_*.txt")) + \
        glob.glob(os.path.join(src, "ray_regroup_*.txt")):
    shutil.copy(f, dst)
print(sorted(os.listdir(dst)))
