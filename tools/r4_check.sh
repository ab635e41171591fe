#!/bin/bash
export SOLR_ORACLE_ROUNDED_TRANSCENDENTALS=1
FUZZ_CONTAINED=1 FUZZ_MISC=1 FUZZ_ROTATE=1 python tools/fuzz_parity.py 114110 1 2>/dev/null | tail -2
FUZZ_TEXTURES=1 FUZZ_MISC=1 python tools/fuzz_parity.py 115078 1 2>/dev/null | tail -2
FUZZ_TEXTURES=1 FUZZ_MISC=1 python tools/fuzz_parity.py 115085 1 2>/dev/null | tail -2
