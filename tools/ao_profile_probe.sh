export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ao_trace2 -o t -- python3 $R/tools/ao_paths.py > /dev/null 2>&1
cd $R; python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ao_trace2/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "ambient" in row["Name"]:
        print("synthetic 4K buffer:", row["Calls"], "calls, mean", float(row["AverageNs"]) / 1e3, "us")
f = glob.glob("gpurun_out/ao_trace/**/*kernel_trace.csv", recursive=True)
PY
