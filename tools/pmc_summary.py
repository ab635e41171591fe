"""Summarise rocprofv3 --pmc counter_collection.csv files for one kernel (development aid)."""
import csv, collections, glob, sys
pattern = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_standardRenderer<0,"
agg = collections.defaultdict(list)
meta = None
for f in glob.glob(pattern, recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value'])); meta = r
for k in sorted(agg):
    print("%-28s %16.0f  (n=%d)" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
if meta:
    print({a: meta[a] for a in ('Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count')})
