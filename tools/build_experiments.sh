# EXPERIMENTS build of libgmsplat.so (GMS_DBG switches, wave timelines / phase stamps) into gaussian-mesh-splatting_amd/lib_exp/,
# beside the product build.  Use it with:  LD_LIBRARY_PATH=$PWD/gaussian-mesh-splatting_amd/lib_exp GMSPLAT_LIB=.../lib_exp/libgmsplat.so
cd "$(dirname "$0")/../gaussian-mesh-splatting_amd/csrc" && make OUT=../lib_exp EXPERIMENTS=1 ../lib_exp/libgmsplat.so 2>&1 | grep -v "argument unused" | tail -3
