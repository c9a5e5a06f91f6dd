cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --profile-steps 0"
rm -rf /tmp/p5 /tmp/p5a /tmp/pfi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5/trace -o c5 -- $B --steps 30 --warmup 10 --workload c5_flame_like_1m > /tmp/p5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5a/trace -o c5a -- $B --steps 40 --warmup 10 --workload c5_flame_like_1m --mode animate > /tmp/p5a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfi/trace -o fi -- $B --steps 60 --warmup 10 --loss l1_ssim --optimizer fused_adam > /tmp/pfi.log 2>&1
python $R/tools/prof_summary.py /tmp/p5 $R/gpurun_out/r03_rocprofv3_c5_summary.txt > /dev/null
python $R/tools/prof_summary.py /tmp/p5a $R/gpurun_out/r03_rocprofv3_c5_animate_summary.txt > /dev/null
python $R/tools/prof_summary.py /tmp/pfi $R/gpurun_out/r03_rocprofv3_full_iteration_summary.txt > /dev/null
head -16 $R/gpurun_out/r03_rocprofv3_c5_summary.txt | cut -c1-130
head -14 $R/gpurun_out/r03_rocprofv3_full_iteration_summary.txt | cut -c1-130
