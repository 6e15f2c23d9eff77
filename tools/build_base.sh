#!/bin/bash
# A/B helper: builds the env / learner library of a commit (default HEAD) as apex_amd/lib/libapx_base.so (sources checked out under build/ab, which does not travel to the GPU box)
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf $ROOT/build/ab && mkdir -p $ROOT/build/ab
git -C $ROOT --work-tree=$ROOT/build/ab checkout $REV -- apex_amd/csrc include && git -C $ROOT reset -q
make -C $ROOT/build/ab/apex_amd/csrc -j3 VARIANT=base > /dev/null 2>&1
cp $ROOT/build/ab/apex_amd/lib/libapx_base.so $ROOT/apex_amd/lib/libapx_base.so
echo built $ROOT/apex_amd/lib/libapx_base.so from $REV
