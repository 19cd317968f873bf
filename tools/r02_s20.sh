cd $GRAFT_REPO_ROOT
GKOC_TUNE_4=2 timeout 600 python -m pytest tests/test_coo_hybrid_gpu.py -q -x 2>&1 | tail -2
for v in 1 2 1 2; do GKOC_TUNE_4=$v timeout 600 python tools/format_bench.py 256 2>&1 | grep -E " coo " | sed "s/^/fused mode $v: /"; done
