# round 5, GPU call 16: a short fuzz run whose summary separates the conditioning cap's ratios above the RARE floor from those below it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05p
timeout 150 python tools/fuzz_parity.py 120 74000 > gpurun_out/${T}_fuzz_120cases.log 2>&1; tail -n 5 gpurun_out/${T}_fuzz_120cases.log | cut -c1-300
timeout 60 python tools/fuzz_parity.py 40 75000 det > gpurun_out/${T}_fuzz_40cases_deterministic_strict.log 2>&1; tail -n 5 gpurun_out/${T}_fuzz_40cases_deterministic_strict.log | cut -c1-300
