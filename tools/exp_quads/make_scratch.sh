#!/usr/bin/env bash
# A scratch copy of the kernel sources with the experiment applied: <dir>/umr_amd/csrc (+ <dir>/include -> the repo's).
# usage: tools/exp_quads/make_scratch.sh <dir>
set -euo pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"
D="$1"; mkdir -p "$D/umr_amd"
rm -rf "$D/umr_amd/csrc"; cp -r "$R/umr_amd/csrc" "$D/umr_amd/csrc"
ln -sfn "$R/include" "$D/include"
(cd "$D" && patch -p1 -s < "$R/tools/exp_quads/quads_ag.patch")
echo "$D/umr_amd/csrc"
