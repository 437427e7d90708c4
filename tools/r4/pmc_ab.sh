#!/usr/bin/env bash
# GPU box: SQ instruction counters (passes sq1, sq2) for the product library and for each experimental library given
set -uo pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$R"
TAG="$1"; shift
OUT="$R/gpurun_out/r4_pmc_$TAG"; mkdir -p "$OUT"
PMC_GROUPS=sq1,sq2 python tools/r4/pmc_passes.py "$OUT/product" 3 0.6 0.9 > /dev/null 2>&1
for lib in "$@"; do
  b=$(basename "$lib" .so)
  UMR_LIB_FILE="$lib" PMC_GROUPS=sq1,sq2 python tools/r4/pmc_passes.py "$OUT/$b" 3 0.6 0.9 > /dev/null 2>&1
done
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+'/*/pmc_summary.json')):
    r=json.load(open(f))
    print('==',os.path.basename(os.path.dirname(f)))
    for k,c in r['kernels'].items():
        if 'k_raster' not in k: continue
        dur=c.get('GRBM_GUI_ACTIVE',0)/8
        print('  %-52s valu %.4gM salu %.4gM smem %.3gM cycles %.0fk  active_valu/insts %.3f  valu_busy %.3f  lane_use %.3f wait %.3f stall %.3f'%(k[:52],c.get('SQ_INSTS_VALU',0)/1e6,c.get('SQ_INSTS_SALU',0)/1e6,c.get('SQ_INSTS_SMEM',0)/1e6,dur/1e3,
           c.get('SQ_ACTIVE_INST_VALU',0)/max(c.get('SQ_INSTS_VALU',1),1), c.get('SQ_ACTIVE_INST_VALU',0)*4/max(dur*1024,1), c.get('SQ_THREAD_CYCLES_VALU',0)/max(c.get('SQ_ACTIVE_INST_VALU',1)*64,1), c.get('SQ_WAIT_ANY/WAVE_CYCLES',0), c.get('SQ_WAIT_INST_ANY/WAVE_CYCLES',0)))
PY
