# round 5, GPU call 17: the in-tree binaries as they stand at the end of the round -- smoke() and the default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05q
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -n 1 gpurun_out/${T}_smoke.log | cut -c1-300
timeout 90 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${T}_bench.json.log 2>gpurun_out/${T}_bench.err; cut -c1-400 gpurun_out/${T}_bench.json.log
