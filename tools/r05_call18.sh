# round 5, GPU call 18: where the GPU idles inside a step (kernel trace of the headline bench -> tools/gpu_gaps.py), with the
# untraced bench line of the same box beside it (the tracer itself slows the host's launches: read the gaps as locations, not sizes)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; T=r05r
timeout 60 python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/${T}_bench_untraced.json.log 2>/dev/null
rm -rf /tmp/prof_gaps
timeout 90 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gaps -o t -- python $R/bench.py --no-cpu-baseline --profile-steps 0 --steps 100 --warmup 20 > $R/gpurun_out/${T}_bench_traced.json.log 2>/dev/null
F=$(find /tmp/prof_gaps -name "*kernel_trace.csv" | head -1)
(echo "# untraced: $(python -c "import json,sys; j=json.loads(open('$R/gpurun_out/${T}_bench_untraced.json.log').read().strip().splitlines()[-1]); print(j['value'], 'it/s', j['ms_per_step'], 'ms')")"
 echo "# traced:   $(python -c "import json,sys; j=json.loads(open('$R/gpurun_out/${T}_bench_traced.json.log').read().strip().splitlines()[-1]); print(j['value'], 'it/s', j['ms_per_step'], 'ms')")"
 python $R/tools/gpu_gaps.py $F 0.4) > $R/gpurun_out/${T}_gpu_gaps.txt 2>&1
cat $R/gpurun_out/${T}_gpu_gaps.txt | cut -c1-120 | head -20
