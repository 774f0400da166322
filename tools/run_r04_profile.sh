#!/bin/bash
# Starts tools/r04_profile.sh on a GPU box and records which commit the box ran: the snapshot gpurun sends is the working tree,
# so the tree must be clean (then it IS `git rev-parse HEAD`).  Afterwards: python tools/make_r04_profile_md.py
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04_final
if [ -n "$(git status --porcelain -- sylph_amd bench.py synth.py include tools oracle)" ]; then echo "working tree not clean: commit first" >&2; exit 1; fi
git rev-parse HEAD > gpurun_out/r04_final/head_local.txt
git log -1 --format=%s >> gpurun_out/r04_final/head_local.txt
exec /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_profile.sh'
