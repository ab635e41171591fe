#!/bin/bash
# Development aid: rocprofv3's FETCH_SIZE / WRITE_SIZE against known byte counts in the renderer's own access patterns
# (tools/fetch_calibration.hip), separate --pmc passes as MI355X_MICROARCH.md prescribes.
#   bash tools/fetch_calibration.sh TAG        (GPU box, from the repo root)  -> gpurun_out/TAG/fetch_calibration.txt
TAG=${1:-r6}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calibration $ROOT/tools/fetch_calibration.hip || exit 1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/calib_$C
  rocprofv3 --pmc $C --output-format csv -d $OUT/calib_$C -o pmc -- /tmp/fetch_calibration > $OUT/calib_$C.log 2>&1
done
cd $ROOT
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
known = {"read_ids_int4_per_pixel": 3840 * 2160 * 16, "read_pp_two_float4_per_pixel": 3840 * 2160 * 32,
         "read_depth_4_of_32_bytes": 3840 * 2160 * 32, "read_linear_16_bytes_per_lane": 3840 * 2160 * 16,
         "read_scalar_32_byte_records": 32768 * 64 * 32, "write_ids_pp_rgb_per_pixel": 3840 * 2160 * 51}
agg = collections.defaultdict(list)
for f in glob.glob(out + "/calib_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
lines = ["# tools/fetch_calibration.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB) per launch of kernels that move a KNOWN byte count",
         "# the way the renderer does (one wave per 8 x 8 tile of a 3840 x 2160 frame; 768 MB written between launches so that",
         "# nothing is left in the Infinity Cache); bytes the kernel moved / bytes the counter reports = the correction factor",
         "%-34s %-11s %14s %14s %8s" % ("kernel", "counter", "known bytes", "reported", "factor")]
for (name, counter), v in sorted(agg.items()):
    if name not in known:
        continue
    if (counter == "FETCH_SIZE") != name.startswith("read"):
        continue
    rep = sum(v[1:]) / max(len(v) - 1, 1) * 1024.0      # (the first launch pays for cold page tables)
    lines.append("%-34s %-11s %14d %14.0f %8.3f" % (name, counter, known[name], rep, known[name] / rep if rep else 0))
open(out + "/fetch_calibration.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
