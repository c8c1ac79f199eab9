#!/bin/bash
# Build libstito_hip.so of another revision for same-box A/B runs: tools/build_rev_lib.sh <git rev> <tag>
# -> st-ito_amd/st_ito/_lib/ab/libstito_hip_<tag>.so (use with STITO_LIB_PATH; ABI-compatible revisions only)
set -e
rev=$1; tag=$2
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d)
git -C "$root" archive "$rev" st-ito_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/st-ito_amd/csrc" -j8 OUT="$tmp/lib.so" > /dev/null
mkdir -p "$root/st-ito_amd/st_ito/_lib/ab"
cp "$tmp/lib.so" "$root/st-ito_amd/st_ito/_lib/ab/libstito_hip_$tag.so"
rm -rf "$tmp"
echo "built st-ito_amd/st_ito/_lib/ab/libstito_hip_$tag.so from $rev"
