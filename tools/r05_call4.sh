# round 5, GPU call 4: what an integer LDS atomic would buy the micro-tile backward (EXPERIMENTS build, wrong results, timing only),
# at 6 and at 5 blocks per CU
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05d
X=$GRAFT_REPO_ROOT/gaussian-mesh-splatting_amd/lib_exp
export LD_LIBRARY_PATH=$X:$LD_LIBRARY_PATH GMSPLAT_LIB=$X/libgmsplat.so
bash tools/ab.sh $T "-" "GMS_DBG=2048" "GMS_LDS_PAD=6000" "GMS_DBG=2048 GMS_LDS_PAD=6000" "GMS_DBG=2048 GMS_LDS_PAD=14000" "-" "GMS_DBG=2048"
