#!/bin/bash
# GPU box: the GPU suite on the current library, then raster kernel timings (library-owned HIP events, tools/sweep_fm.py) for
# several thin-face thresholds of k_face_setup (umr_debug_set("thin_face_h_1e6")): 0 = only faces with an ill-conditioned
# edge take the reference's inside route (round 2's behaviour) ... 1e9 = every face does.
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/thin; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for t in 0 16000 32000 64000 1000000000; do
  UMR_THIN=$t UMR_CFG4=0 timeout 400 python tools/sweep_fm.py thin 2>&1 | grep '^{' >> $O/sweep.log
done
for t in 0 16000 1000000000; do
  UMR_THIN=$t timeout 400 python tools/sweep_fm.py thin-cfg4 2>&1 | grep '^{' >> $O/sweep_cfg4.log
done
cat $O/sweep.log $O/sweep_cfg4.log
