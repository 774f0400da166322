#!/bin/bash
# A/B builds of libsylph_hip.so for the replay kernel (round 5): the marker-tag scan (SYLPH_REPLAY_TAGS) x one or two wavefronts per
# bucket (SYLPH_REPLAY_TPB).  Run here (hipcc cross-compiles); the variants travel to the GPU box as sylph_amd/libsylph_hip.so.<name>.
cd "$(dirname "$0")/../sylph_amd/csrc" || exit 1
make -j8 > /dev/null || exit 1
cp ../libsylph_hip.so /tmp/libsylph_hip.keep
OBJS="capi.o prims.o seeds.o sketch.o a10.o contain.o hits.o shard.o genomes.o reads.o pipeline.o"
for v in "base 0 128" "tags 1 128" "b64 0 64" "tags64 1 64"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -DSYLPH_REPLAY_TAGS=$2 -DSYLPH_REPLAY_TPB=$3 -c replay_lds.hip -o /tmp/replay_lds_$1.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libsylph_hip.so.$1 $OBJS /tmp/replay_lds_$1.o -ldl || exit 1
  echo "built libsylph_hip.so.$1 (TAGS=$2 TPB=$3)"
done
