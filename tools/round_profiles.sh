#!/bin/bash
# Everything a round commits under profiles/<tag>/ (GPU box, from the repo root):  bash tools/round_profiles.sh r4
# then here:  python tools/collect_profiles.py r4 <commit> cornell height_field molecule cfg4 ; python tools/design_tables.py r4
TAG=${1:-r4}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for s in cornell height_field molecule cfg4; do bash tools/profile_round.sh $TAG $s; done
python tools/api_frame.py > $OUT/api_frame_cornell.txt 2>/dev/null
python tools/upload_time.py > $OUT/upload_time.txt 2>/dev/null
for s in cornell height_field molecule; do python tools/strip_balance.py $s 8 > $OUT/strip_balance_$s.txt 2>/dev/null; done
ls $OUT | head -60
