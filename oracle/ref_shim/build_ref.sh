#!/usr/bin/env bash
# Builds oracle/_ref/libsoftras_ref.so from the reference's own kernel source where it
# lies under /root/reference (this container only; /root/reference does not exist on the
# GPU box, which uses the prebuilt .so that travels with the snapshot).
# No reference text is written inside the repository: the extracted device-code span goes
# to a scratch dir under ${TMPDIR:-/tmp}.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC=/root/reference/external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu
OUT="$HERE/../_ref"
if [ ! -f "$SRC" ]; then
  echo "[ref_shim] $SRC not present; keeping any prebuilt $OUT/libsoftras_ref.so" >&2
  exit 0
fi
SCRATCH="$(mktemp -d "${TMPDIR:-/tmp}/umr_ref_shim.XXXXXX")"
trap 'rm -rf "$SCRATCH"' EXIT
# device code only: the anonymous namespace (helpers + 3 kernels); the host launchers
# below it use <<<>>> and ATen and are replaced by the loops in host_exec_shim.cpp
sed -n '22,659p' "$SRC" > "$SCRATCH/kernels_body.inc"
# texture-atlas bake (functional/save_obj.py's device step): anonymous namespace of
# create_texture_image_cuda_kernel.cu, lines 8-72 (the launcher below it is replaced by a loop)
sed -n '8,72p' "$(dirname "$SRC")/create_texture_image_cuda_kernel.cu" > "$SCRATCH/atlas_body.inc"
mkdir -p "$OUT"
CXX=/opt/rocm/lib/llvm/bin/clang++
"$CXX" -O2 -fPIC -shared -std=c++17 -ffp-contract=off -ftrivial-auto-var-init=zero \
  -Wno-unused-function -I"$SCRATCH" "$HERE/host_exec_shim.cpp" -o "$OUT/libsoftras_ref.so"
echo "[ref_shim] built $OUT/libsoftras_ref.so"
