#!/usr/bin/env bash
# GPU box: rebuild libumr_hip.so with compile-time variants of the face-major backward and time it at the bench's
# sizes.  Usage: tools/sweep_fm.sh "<flags of variant 1>" "<flags of variant 2>" ...   (flags space-separated)
set -euo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"
for v in "$@"; do
  python -c "import sys; from umr_amd.build import build; build(force=True, verbose=False, extra_flags=sys.argv[1].split())" "$v"
  python tools/sweep_fm.py "$v"
done
python -c "from umr_amd.build import build; build(force=True, verbose=False)"
