cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do timeout 100 python -m pytest tests/test_gpu_raster.py -q -k "very_deep" 2>&1 | grep -E "passed|failed|AssertionError|where=|rotations|scales|means" | cut -c1-700 | head -8; done
