# round 5, GPU call 1: parity of the resident-unit compositing kernels, A/B against the row-queue kernels, phase stamps of both
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05a
timeout 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_deterministic.py tests/test_gpu_negative_controls.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest_ru.log; tail -3 gpurun_out/${T}_pytest_ru.log
bash tools/ab.sh $T "-" "GMS_MICRO_RU=0" "GMS_SEG_LEN=128" "GMS_MICRO_RU=0 GMS_SEG_LEN=128" "GMS_TRIP_BWD=4" "GMS_TRIP_BWD=1" "GMS_TRIP=2" "-" "GMS_MICRO_RU=0"
X=$GRAFT_REPO_ROOT/gaussian-mesh-splatting_amd/lib_exp
for RU in 0 1; do
  LD_LIBRARY_PATH=$X:$LD_LIBRARY_PATH GMSPLAT_LIB=$X/libgmsplat.so GMS_DBG=1024 GMS_MICRO_RU=$RU timeout 300 python tools/micro_phases.py > gpurun_out/${T}_phases_ru$RU.txt 2>&1; tail -16 gpurun_out/${T}_phases_ru$RU.txt | cut -c1-400
done
timeout 400 python tools/fuzz_parity.py 80 51000 > gpurun_out/${T}_fuzz_80cases.log 2>&1; tail -3 gpurun_out/${T}_fuzz_80cases.log | cut -c1-300
