#!/bin/bash
# materialise.sh <variant> -> experiments/build_<variant>/<conv_fwd|conv_wgrad>.hip
# A variant is stored as a patch against the revision of the product kernel it was branched from (patches/INDEX.json).
set -e
cd "$(dirname "$0")/.."
V=$1
REV=$(python3 -c "import json;print(json.load(open('experiments/patches/INDEX.json'))['$V']['base_rev'])")
P=$(python3 -c "import json;print(json.load(open('experiments/patches/INDEX.json'))['$V']['base_path'])")
OUT=experiments/build_$V; mkdir -p $OUT
git show $REV:$P > $OUT/$(basename $P)
[ -s experiments/patches/$V.patch ] && patch -s -p1 -d $OUT < experiments/patches/$V.patch
echo $OUT/$(basename $P)
