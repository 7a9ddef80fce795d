#!/bin/bash
# Builds a variant of libgf2hip.so into scratch_ab/<name>.so for same-box A/B runs (scratch_ab/ is git-ignored but travels to the GPU box):
#   bash profiles/build_variant.sh <name> [<git rev whose gf2_slabs.hip to take> | -] [extra hipcc flags, e.g. -DGAT_EXP_NOLOOKUP]
# The what-if variants of the gather kernel (GAT_EXP_*: wrong results, timing only) are NOT in the product source: they are
# profiles/gather_what_if_variants.patch, applied to the build copy here whenever a -DGAT_EXP_* flag is given.
set -e
name=$1; rev=${2:--}; shift; shift || true
root=$(cd $(dirname $0)/.. && pwd)
work=/tmp/variant_$name
rm -rf $work && mkdir -p $work/quantum_css_codes_amd $work/include
cp -r $root/quantum_css_codes_amd/csrc $work/quantum_css_codes_amd/ && rm -rf $work/quantum_css_codes_amd/csrc/build
cp $root/include/gf2hip.h $work/include/
if [ "$rev" != "-" ]; then      # (check_isa.py knows the kernels by name: it goes with the source it checks)
  git -C $root show $rev:quantum_css_codes_amd/csrc/gf2_slabs.hip > $work/quantum_css_codes_amd/csrc/gf2_slabs.hip
  git -C $root show $rev:quantum_css_codes_amd/csrc/check_isa.py > $work/quantum_css_codes_amd/csrc/check_isa.py
fi
case "$*" in *GAT_EXP_*) patch -d $work -p0 < $root/profiles/gather_what_if_variants.patch ;; esac
# ... and so is "every record takes 32 bytes of HBM" (REC_EXP_SHORT_ONLY: the bound on what smaller records could gain)
case "$*" in *REC_EXP_*) patch -d $work -p1 < $root/profiles/record_bytes_what_if.patch ;; esac
# ... and "the two halves of a row in arrays of their own" (WHATIF_HALF_PITCH: gf2_syndrome_sparse_dev takes a pitch of half a row)
case "$*" in *WHATIF_HALF_PITCH*) patch -d $work -p1 < $root/profiles/half_pitch_what_if.patch ;; esac
make -C $work/quantum_css_codes_amd/csrc ROOT=$work FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -I$work/include -Wall -Wno-unused-function $*" 2>&1 | grep -E "rror|check_isa:" || true
mkdir -p $root/scratch_ab && cp $work/quantum_css_codes_amd/libgf2hip.so $root/scratch_ab/$name.so && ls -la $root/scratch_ab/$name.so
