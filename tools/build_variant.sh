#!/bin/bash
# An A/B build of libsylph_hip.so: tools/build_variant.sh NAME "FLAGS" file.hip [file.hip ...] recompiles the named sources with the extra
# flags (the others are taken from the default build) and links sylph_amd/libsylph_hip.so.NAME, which travels to the GPU box;
# `SYLPH_HIP_LIBRARY=sylph_amd/libsylph_hip.so.NAME` makes sylph_amd/binding.py load it (tools/gpu_call.sh ab-env alternates such settings).
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../sylph_amd/csrc" || exit 1
make -j8 > /dev/null || exit 1
SRC="$PWD"; tmp=/tmp/sylph_variant_$name; mkdir -p $tmp
all="capi prims seeds sketch replay_lds a10 contain hits shard genomes reads pipeline fastq inflate"
objs=""
for f in $all; do
  if [[ " $* " == *" $f.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $flags -I"$SRC" -c "$SRC/$f.hip" -o $tmp/$f.o || { echo "FAILED $f"; exit 1; }
    objs="$objs $tmp/$f.o"
  else objs="$objs $SRC/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$SRC/../libsylph_hip.so.$name" $objs -ldl -lz && echo "built sylph_amd/libsylph_hip.so.$name ($flags: $*)"
