# round 5, GPU call 12: mesh_bwd with the face's splat loop unrolled (loads of all splats issued together)
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05l
bash tools/ab.sh $T "-" "-" "-"
timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_training.py -m gpu -q --maxfail=6 2>&1 | tail -5 > gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
