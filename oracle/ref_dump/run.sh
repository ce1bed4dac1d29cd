#!/bin/bash
# Produces tests/golden/ref_stage_vectors/*.vec from the reference itself (see README.md).  Needs cargo.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF="${1:-/root/reference}"
command -v cargo >/dev/null || { echo "cargo not found: the reference cannot be built here (parity stays unpinned)"; exit 3; }
WORK="$ROOT/oracle/_ref/jxl-rs"
rm -rf "$WORK" && mkdir -p "$WORK"
cp -r "$REF"/. "$WORK"/
# functions that need a decoded frame around them are instrumented in the scratch copy (anchors verified first)
python3 "$HERE/instrument.py" --check "$WORK"
python3 "$HERE/instrument.py" "$WORK"
# the .vec writer becomes a crate-level test module ...
cat "$HERE/vec_io.rs" >> "$WORK/jxl/src/lib.rs"
# ... and each dump module is appended to the file named in its first line
for f in "$HERE"/*_dump.rs; do
  target=$(head -1 "$f" | sed -n 's#^// APPEND-TO: ##p')
  [ -n "$target" ] || { echo "$f: no APPEND-TO line"; exit 1; }
  tail -n +2 "$f" >> "$WORK/$target"
done
export JXL_REF_DUMP_DIR="$ROOT/tests/golden/ref_stage_vectors"
mkdir -p "$JXL_REF_DUMP_DIR"
(cd "$WORK" && cargo test -p jxl --release ref_dump -- --nocapture)
ls -l "$JXL_REF_DUMP_DIR"
