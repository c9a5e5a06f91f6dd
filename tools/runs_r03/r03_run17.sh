cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/host_profile_ddp.py factor > gpurun_out/r03_host_ddp_factor.txt 2>&1
timeout 300 python tools/host_profile_ddp.py dense > gpurun_out/r03_host_ddp_dense.txt 2>&1
head -50 gpurun_out/r03_host_ddp_factor.txt | cut -c1-150
