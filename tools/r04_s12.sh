#!/bin/bash
# round 4, session 12: lookup tables / state guards through the dropin tests; occupancy experiments (COO, f32 CSR)
TAG=${1:-r04s12}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== dropin"
timeout 900 python -m pytest tests/test_dropin_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $OUT/tests.txt
(cd oracle/_ref/dropin && LD_LIBRARY_PATH=.:../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib timeout 600 ./dropin_test 2>&1 | grep -i "lookup\|FAIL" | head -5)
echo "== formats, default"
timeout 600 python tools/format_bench.py 256 2>&1 | tail -12 | tee $OUT/formats_default.txt
echo "== formats, COO with 5 waves per SIMD (GKOC_TUNE_9=1)"
GKOC_TUNE_9=1 timeout 600 python tools/format_bench.py 256 2>&1 | grep -i "coo\|hybrid" | tee $OUT/formats_coo5.txt
echo "== dtypes, default"
timeout 600 python tools/dtype_bench.py 256 2>&1 | tail -8 | tee $OUT/dtypes_default.txt
echo "== dtypes, f32 CSR with 5 waves per SIMD (GKOC_TUNE_9=2)"
GKOC_TUNE_9=2 timeout 600 python tools/dtype_bench.py 256 2>&1 | tail -8 | tee $OUT/dtypes_f32_5.txt
echo done
