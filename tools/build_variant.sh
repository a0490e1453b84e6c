#!/bin/bash
# Build the csrc of a git revision into ab/libosud_<name>.so for same-box A/B runs (OSUD_LIB=ab/libosud_<name>.so).
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/osud_variant_$name
rm -rf $tmp && mkdir -p $tmp/osu_diffusion_amd $tmp/include
git -C $root archive $rev osu_diffusion_amd/csrc include | tar -x -C $tmp
make -C $tmp/osu_diffusion_amd/csrc -j8 > $tmp/build.log 2>&1 || { tail $tmp/build.log; exit 1; }
mkdir -p $root/ab && cp $tmp/osu_diffusion_amd/libosud.so $root/ab/libosud_$name.so
echo built $root/ab/libosud_$name.so
