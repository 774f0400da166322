#!/bin/bash
# Starts tools/r02_profile.sh on a GPU box and records which commit the box ran: the snapshot gpurun sends is the working tree,
# so the tree must be clean (then it IS `git rev-parse HEAD`).  Afterwards: python tools/make_profile_md.py
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r02_final
if [ -n "$(git status --porcelain -- sylph_amd bench.py include tools oracle)" ]; then echo "working tree not clean: commit first" >&2; exit 1; fi
git rev-parse HEAD > gpurun_out/r02_final/head_local.txt
git log -1 --format=%s >> gpurun_out/r02_final/head_local.txt
exec /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r02_profile.sh'
