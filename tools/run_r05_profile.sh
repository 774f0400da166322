#!/bin/bash
# Starts tools/r05_profile.sh on a GPU box and records which commit the box ran: the snapshot gpurun sends is the working tree,
# so the tree must be clean (then it IS `git rev-parse HEAD`).  Afterwards: python tools/make_r05_profile_md.py
# `bash tools/run_r05_profile.sh r05_pmc_refresh.sh`: only the counter passes + seeds_traffic.json + the default line, after a late change under csrc/
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r05_final
if [ -n "$(git status --porcelain -- sylph_amd bench.py synth.py include tools oracle)" ]; then echo "working tree not clean: commit first" >&2; exit 1; fi
git rev-parse HEAD > profiles/.head_local          # (untracked, travels with the snapshot: the box's seeds_traffic.json names the commit)
git log -1 --format=%s >> profiles/.head_local
exec /usr/local/graft/bin/gpurun --timeout 2400 -- "bash tools/${1:-r05_profile.sh}"
