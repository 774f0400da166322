#!/bin/bash
# Builds the REAL reference (bluenote-1577/sylph, pure Rust) from the sources where they lie, into oracle/_ref/ only.
# Test infrastructure: the day a box has `cargo` (this image and the GPU boxes of rounds 1-4 do not: `which cargo rustc` is empty)
# this turns the oracle's "parity unpinned" into a pin against the reference binary:
#     bash oracle/ref_build.sh && python tests/golden/regen_from_ref.py && python -m pytest tests/test_ref_binary.py
# Exit status 3 = cannot be built here (no toolchain / no sources / crates not available offline); nothing is written then.
set -u
REF=${SYLPH_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
command -v cargo >/dev/null 2>&1 || { echo "ref_build: no cargo on PATH — the reference cannot be built on this box" >&2; exit 3; }
[ -f "$REF/Cargo.toml" ] || { echo "ref_build: no reference sources at $REF" >&2; exit 3; }
mkdir -p "$HERE/_ref"
# --locked: the reference's own Cargo.lock (scalable_cuckoo_filter 0.2.4, statrs 0.16.1, fastrand 2.1.1, needletail 0.5.1, ...);
# the target directory is ours, nothing is written under $REF.  Without network the crates must be in ~/.cargo already.
if ! cargo build --release --locked --manifest-path "$REF/Cargo.toml" --target-dir "$HERE/_ref/target" ${SYLPH_REF_CARGO_FLAGS:-}; then
    echo "ref_build: cargo build failed (offline box without the crates?)" >&2
    exit 3
fi
cp "$HERE/_ref/target/release/sylph" "$HERE/_ref/sylph"
"$HERE/_ref/sylph" --version
