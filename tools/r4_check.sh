#!/bin/bash
# round-4 fuzz: random scenes against the oracle AS PINNED, and through in-process devices
mkdir -p gpurun_out
{
echo "# tools/fuzz_parity.py at $(cat gpurun_out/.commit 2>/dev/null), oracle as pinned (libm binary32 transcendentals), order-free lists built with the first frame"
echo "# seeds 110001..110300 plain | 111001..111300 CONTAINED TWINS | 112001..112200 CONTAINED DEVICES | 113001..113200 MISC DEVICES | 114001..114150 CONTAINED MISC ROTATE | 115001..115150 TEXTURES MISC | 116001..116100 CONTAINED MODES STRIPS | 117001..117060 CONTAINED FLIGHTS DEVICES"
python tools/fuzz_parity.py 110001 300 2>/dev/null | tail -4
FUZZ_CONTAINED=1 FUZZ_TWINS=1 python tools/fuzz_parity.py 111001 300 2>/dev/null | tail -4
FUZZ_CONTAINED=1 FUZZ_DEVICES=1 python tools/fuzz_parity.py 112001 200 2>/dev/null | tail -4
FUZZ_MISC=1 FUZZ_DEVICES=1 python tools/fuzz_parity.py 113001 200 2>/dev/null | tail -4
FUZZ_CONTAINED=1 FUZZ_MISC=1 FUZZ_ROTATE=1 python tools/fuzz_parity.py 114001 150 2>/dev/null | tail -4
FUZZ_TEXTURES=1 FUZZ_MISC=1 python tools/fuzz_parity.py 115001 150 2>/dev/null | tail -4
FUZZ_CONTAINED=1 FUZZ_MODES=1 FUZZ_STRIPS=1 python tools/fuzz_parity.py 116001 100 2>/dev/null | tail -4
FUZZ_CONTAINED=1 FUZZ_FLIGHTS=1 FUZZ_DEVICES=1 python tools/fuzz_parity.py 117001 60 2>/dev/null | tail -4
} > gpurun_out/fuzz_r4.txt 2>&1
cat gpurun_out/fuzz_r4.txt
