#!/bin/bash
# Builds the REAL reference (Sleepwalking/libllsm2 under /root/reference, untouched) against a real ciglet
# checkout into oracle/_ref/libllsm2_ref.so -- the one thing that can pin the oracle to the reference's actual
# numbers.  ciglet is neither vendored nor pinned by the reference (README.md:43-50) and is absent from this
# image (no network), so this recipe has NOT been run here: "parity unpinned" stands until it has.
#
#   bash oracle/build_ref.sh /path/to/ciglet [/path/to/reference]
#
# /path/to/ciglet: a checkout of github.com/Sleepwalking/ciglet in which `make single-file` has been run
# (the reference's README asks for exactly that), or any directory holding ciglet.h + ciglet.c.
# Nothing is copied into the repository: sources are compiled where they lie, outputs go to oracle/_ref/
# (git-ignored, but shipped to the GPU box with the tree).  No stand-ins are written for anything missing.
set -euo pipefail
CIGLET=${1:?usage: build_ref.sh /path/to/ciglet [/path/to/reference]}
REF=${2:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
H=$(find "$CIGLET" -name ciglet.h | head -1)
C=$(find "$CIGLET" -name ciglet.c | head -1)
[ -n "$H" ] && [ -n "$C" ] || { echo "ciglet.h / ciglet.c not found under $CIGLET (run 'make single-file' there)"; exit 1; }
mkdir -p "$OUT/inc"
ln -sfn "$(dirname "$H")" "$OUT/inc/ciglet"            # the reference includes <ciglet/ciglet.h>
# FP_TYPE=float and -O2 without -ffast-math / FMA contraction: the evaluation order SURVEY Appendix B fixes
gcc -std=c99 -O2 -fPIC -shared -ffp-contract=off -DFP_TYPE=float -DUSE_PTHREAD -I"$OUT/inc" -I"$REF" \
  -o "$OUT/libllsm2_ref.so" \
  "$REF"/container.c "$REF"/frame.c "$REF"/dsputils.c "$REF"/llsmutils.c "$REF"/layer0.c "$REF"/layer1.c \
  "$REF"/coder.c "$REF"/llsmrt.c "$C" -lm -lpthread
echo "built $OUT/libllsm2_ref.so; now: python oracle/make_golden_from_ref.py"
