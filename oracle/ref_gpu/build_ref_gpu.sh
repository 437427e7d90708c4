#!/usr/bin/env bash
# Builds oracle/_ref/libsoftras_ref_gfx950.so: the reference's rasterizer device code compiled NATIVELY by hipcc for
# gfx950 from the source where it lies under /root/reference (this container only; the GPU box uses the prebuilt .so
# that travels with the snapshot).  No reference text is written inside the repository.
# Flags: -ffp-contract=off (nvcc contracts a*b+c into FMA by default; the oracle and the host shim do not, and the
# committed goldens come from the non-contracted build, so this build matches THEM -- the contracted variant is built
# next to it as ..._fma.so to measure what nvcc's default would change); -ftrivial-auto-var-init=zero for the same
# reason as oracle/ref_shim (backward_sample_texture returns an uninitialised local, :199-218).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC=/root/reference/external/SoftRas/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu
OUT="$HERE/../_ref"
if [ ! -f "$SRC" ]; then
  echo "[ref_gpu] $SRC not present; keeping any prebuilt $OUT/libsoftras_ref_gfx950.so" >&2
  exit 0
fi
SCRATCH="$(mktemp -d "${TMPDIR:-/tmp}/umr_ref_gpu.XXXXXX")"
trap 'rm -rf "$SCRATCH"' EXIT
sed -n '22,659p' "$SRC" > "$SCRATCH/kernels_body.inc"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON="-O2 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ftrivial-auto-var-init=zero -munsafe-fp-atomics -Wno-unused-function -Wno-unused-value -I$SCRATCH"
"$HIPCC" $COMMON -ffp-contract=off "$HERE/device_exec.hip" -o "$OUT/libsoftras_ref_gfx950.so"
"$HIPCC" $COMMON -ffp-contract=fast "$HERE/device_exec.hip" -o "$OUT/libsoftras_ref_gfx950_fma.so"
echo "[ref_gpu] built $OUT/libsoftras_ref_gfx950.so (+ _fma variant)"
