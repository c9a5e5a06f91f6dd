# round 5, GPU call 15: more random scenes on the final build (default mode, and deterministic mode under the strict criterion)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05o
timeout 600 python tools/fuzz_parity.py 600 70000 > gpurun_out/${T}_fuzz_600cases_final_build.log 2>&1; tail -3 gpurun_out/${T}_fuzz_600cases_final_build.log | cut -c1-300
timeout 300 python tools/fuzz_parity.py 200 72000 det > gpurun_out/${T}_fuzz_200cases_deterministic_strict_final_build.log 2>&1; tail -3 gpurun_out/${T}_fuzz_200cases_deterministic_strict_final_build.log | cut -c1-300
