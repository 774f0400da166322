#!/bin/bash
# A/B builds of libsylph_hip.so for the read kernel's bookkeeping trims (round 6): only reads.hip differs.
#   v0  round 5's bookkeeping (-DSYLPH_NO_DPP_SCAN -DSYLPH_READS_ALWAYS_DEAL -DSYLPH_READS_COOP_PLAIN_HASH)
#   v1  + (b) DPP wave scans            v2  + (a) dealing decided before the histogram            v3  + (d) gfx950 hash in the cooperative pass (= the default build)
# Run here (hipcc cross-compiles); the variants travel to the GPU box as sylph_amd/libsylph_hip.so.v0 .. v3.
cd "$(dirname "$0")/../sylph_amd/csrc" || exit 1
make -j8 > /dev/null || exit 1
SRC="$PWD"
mkdir -p /tmp/reads_ab && cd /tmp/reads_ab || exit 1
build() {   # name, flags
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $2 -I"$SRC" -c "$SRC/reads.hip" -o reads_$1.o || { echo "FAILED $1"; return; }
  objs=""; for f in capi prims seeds sketch replay_lds a10 contain hits shard genomes pipeline fastq inflate; do objs="$objs $SRC/$f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$SRC/../libsylph_hip.so.$1" $objs reads_$1.o -ldl -lz && echo "built libsylph_hip.so.$1"
}
build v0 "-DSYLPH_NO_DPP_SCAN -DSYLPH_READS_ALWAYS_DEAL -DSYLPH_READS_COOP_PLAIN_HASH" &
build v1 "-DSYLPH_READS_ALWAYS_DEAL -DSYLPH_READS_COOP_PLAIN_HASH" &
build v2 "-DSYLPH_READS_COOP_PLAIN_HASH" &
build v3 "" &
wait
