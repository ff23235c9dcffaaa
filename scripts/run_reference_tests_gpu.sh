#!/bin/bash
# Gate G4: the reference's own tests, unchanged, on the MI355X with this backend plugged in.
#   in the build container:   bash scripts/run_reference_tests_gpu.sh --stage     (scratch copy of /root/reference -> .refcopy/, git-ignored)
#   on the GPU box:           gpurun -- 'bash scripts/run_reference_tests_gpu.sh'  (writes gpurun_out/g4/)
#   afterwards:               bash scripts/run_reference_tests_gpu.sh --unstage
# The copy is never committed (the reference's sources do not belong in this repository); it only travels with the gpurun snapshot.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; REPO=$PWD
if [ "$1" = "--stage" ]; then
  rm -rf .refcopy && mkdir .refcopy
  (cd /root/reference && tar cf - --exclude=.git --exclude=bench --exclude=examples --exclude='*.png' --exclude=external .) | tar xf - -C .refcopy
  du -sh .refcopy; exit 0
fi
if [ "$1" = "--unstage" ]; then rm -rf .refcopy; exit 0; fi
[ -d .refcopy/optimum/quanto ] || { echo "no .refcopy (run --stage in the build container first)"; exit 2; }
OUT=$REPO/gpurun_out/g4; mkdir -p $OUT
export PYTHONPATH=$REPO/.refcopy:$REPO:$REPO/scripts/g4 TMPDIR=/tmp
cd .refcopy
SUITES="${G4_SUITES:-tests/library tests/tensor/weights tests/tensor/ops tests/nn/test_qlinear.py tests/tensor/test_packed_tensor.py}"
timeout ${G4_TIMEOUT:-1500} python -m pytest -p quanto_amd_plugin $SUITES -q --tb=short -p no:cacheprovider 2>&1 | tail -n 400 > $OUT/reference_tests_plugin_mode.log
tail -n 15 $OUT/reference_tests_plugin_mode.log
