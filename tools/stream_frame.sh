#!/bin/bash
# The record behind the defaults of ImageStreaming (csrc/renderer.h; solr_image_ring.hip imageStreamingCuts):
#   bash tools/stream_frame.sh > gpurun_out/r6/stream_frame.txt     (GPU box, from the repo root)
run() { python tools/stream_frame.py "$@" 2>&1 | grep -v "^$"; }
echo "# one frame at a time at 1920x1080: read back behind the kernel (SOLR_HIP_NO_IMAGE_STREAMING=1) / in bands while it renders"
for s in cornell molecule height_field; do
    echo "## $s, as shipped (5 equal bands, the heaviest eighth of the tiles first)"; run $s 200
    echo "## $s, SOLR_HIP_NO_IMAGE_STREAMING=1"; SOLR_HIP_NO_IMAGE_STREAMING=1 run $s 200 | grep -v "left in bands"
done
for s in cornell molecule; do
    for b in 1 2 3 4 6 8; do echo "## $s, SOLR_HIP_STREAM_BANDS=$b"; SOLR_HIP_STREAM_BANDS=$b run $s 100 | grep SolRx_Render; done
    echo "## $s, SOLR_HIP_STREAM_EQUAL=0 (bands of 3 : 2 : 1)"; SOLR_HIP_STREAM_EQUAL=0 run $s 100 | grep SolRx_Render
    for h in 2 4 16 100000; do echo "## $s, SOLR_HIP_STREAM_HEAVY=$h (the heaviest 1 / $h of the tiles first)"; SOLR_HIP_STREAM_HEAVY=$h run $s 100 | grep SolRx_Render; done
    echo "## $s, tiles in launch order (solr_hip_set_tile_scheduling(0))"; TILE_SCHEDULING=0 run $s 100 | grep -v "left in bands"
done
