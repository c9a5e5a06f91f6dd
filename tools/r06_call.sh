# Round-6 GPU calls (one parametrised script; `bash tools/r06_call.sh <step> [args]`), outputs under gpurun_out/.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
S=$1; shift
case $S in
valu)      # instruction issue costs (tools/valu_bench.hip) + the headline on this box
  tools/valu_bench.bin > gpurun_out/r06_valu_bench.txt 2>&1; cat gpurun_out/r06_valu_bench.txt
  bash tools/ab.sh r06_base "-" "-" ;;
ab)        # A/B of environments: bash tools/r06_call.sh ab TAG "ENV_A" "ENV_B" ...
  bash tools/ab.sh "$@" ;;
test)      # parity subset + headline: bash tools/r06_call.sh test TAG [pytest args]
  T=$1; shift
  python -m pytest "$@" -x -q 2>&1 | tail -15 | tee gpurun_out/${T}_pytest.log
  bash tools/ab.sh $T "-" "-" ;;
phases)    # per-wave phase stamps of micro_bwd on the EXPERIMENTS build (tools/build_experiments.sh): bash tools/r06_call.sh phases TAG [ENV...]
  T=$1; shift
  X=$PWD/gaussian-mesh-splatting_amd/lib_exp
  env "$@" GMS_PHASES_DUMP=gpurun_out/${T}_blocks_bwd.npz LD_LIBRARY_PATH=$X:$LD_LIBRARY_PATH GMSPLAT_LIB=$X/libgmsplat.so GMS_DBG=1024 timeout 300 python tools/micro_phases.py > gpurun_out/${T}_micro_bwd_phases.txt 2>&1
  tail -14 gpurun_out/${T}_micro_bwd_phases.txt | cut -c1-300 ;;
fwdphases) # per-wave stamps of micro_head (GMS_DBG 2048) and micro_fwd (4096) on the EXPERIMENTS build: bash tools/r06_call.sh fwdphases TAG
  T=$1; shift
  X=$PWD/gaussian-mesh-splatting_amd/lib_exp
  for B in 2048 4096; do
    env "$@" GMS_PHASES_DUMP=gpurun_out/${T}_blocks_$B.npz LD_LIBRARY_PATH=$X:$LD_LIBRARY_PATH GMSPLAT_LIB=$X/libgmsplat.so GMS_DBG=$B timeout 300 python tools/micro_fwd_phases.py > gpurun_out/${T}_micro_fwd_phases_$B.txt 2>&1
    tail -12 gpurun_out/${T}_micro_fwd_phases_$B.txt | cut -c1-330
  done ;;
*) echo "unknown step $S"; exit 2 ;;
esac
