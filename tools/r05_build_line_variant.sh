#!/bin/bash
# A/B build of libsylph_hip.so with 32-byte bucket lines (SYLPH_LINE_SLOTS=4, lambda 2) beside the default 64-byte lines (round 5).
# Run here (hipcc cross-compiles); the variant travels to the GPU box as sylph_amd/libsylph_hip.so.line4.
cd "$(dirname "$0")/../sylph_amd/csrc" || exit 1
make -j8 > /dev/null || exit 1
mkdir -p /tmp/line4 && cd /tmp/line4 || exit 1
SRC="$OLDPWD"
for f in capi prims seeds sketch replay_lds a10 contain hits shard genomes reads pipeline; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -DSYLPH_LINE_SLOTS=4 -I"$SRC" -c "$SRC/$f.hip" -o $f.o || echo "FAILED $f" ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$SRC/../libsylph_hip.so.line4" capi.o prims.o seeds.o sketch.o replay_lds.o a10.o contain.o hits.o shard.o genomes.o reads.o pipeline.o -ldl && echo "built libsylph_hip.so.line4"
