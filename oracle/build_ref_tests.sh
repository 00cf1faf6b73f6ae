#!/bin/bash
# The one program of the reference that builds in this image: test/test-structs.c needs nothing but the reference's
# own llsm.h / buffer.h (every other test includes <ciglet/ciglet.h> or <libpyin/pyin.h>, which are absent -- those stay
# unbuilt; no stand-ins are written).  It is compiled WHERE IT LIES, with the REFERENCE's headers (its #include
# "../llsm.h" resolves inside /root/reference), and linked against the product: if its assertions hold, the product's
# data model (containers, frames, chunks; struct layouts as the reference's headers declare them) is a binary drop-in
# for container.c / frame.c.  Output only into oracle/_ref/ (git-ignored, travels to the GPU box); test infrastructure.
#
#   bash oracle/build_ref_tests.sh [/path/to/reference]
set -euo pipefail
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
LIB=$HERE/../libllsm2_amd
[ -f "$REF/test/test-structs.c" ] || { echo "no reference tree at $REF: nothing built"; exit 0; }
[ -f "$LIB/libllsm2_amd.so" ] || { echo "build the product first (python -m libllsm2_amd.build)"; exit 1; }
mkdir -p "$HERE/_ref"
# FP_TYPE=float as the reference's makefile passes it; rpath relative to the binary so that it runs from any checkout
gcc -std=gnu99 -O1 -DFP_TYPE=float -o "$HERE/_ref/ref_test_structs" "$REF/test/test-structs.c" \
  -L"$LIB" -lllsm2_amd -lm -Wl,-rpath,'$ORIGIN/../../libllsm2_amd'
echo "built $HERE/_ref/ref_test_structs"
