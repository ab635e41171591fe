#!/bin/bash
# sweep of the grouping knobs of groupSiblings (solr_scene.hip) over the bench scenes: ms per 1080p frame
for scene in cornell height_field molecule irt_model pdb_molecule swc_morphology; do
  for flat in 2 4 8; do
    for levels in 1 2 3; do
      ms=$(SOLR_HIP_GROUP_FLAT=$flat SOLR_HIP_GROUP_LEVELS=$levels python bench.py --scene $scene --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "$scene flat=$flat levels=$levels $ms"
    done
  done
done
