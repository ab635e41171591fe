#!/bin/bash
bash tools/ab_bench.sh "height_field cornell" ab/libsolr_hip_A.so ab/libsolr_hip_prio3.so ab/libsolr_hip_prio3_first.so
for l in A prio3 prio3_first; do cp ab/libsolr_hip_$l.so sol-r_amd/csrc/libsolr_hip.so; echo "== $l"; python tools/strip_balance.py height_field 8 2>/dev/null | grep -E "slowest"; done
cp ab/libsolr_hip_A.so sol-r_amd/csrc/libsolr_hip.so
