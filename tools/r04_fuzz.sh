# parity sweeps on the final sources: bash tools/r04_fuzz.sh N_DEFAULT N_DET
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python tools/fuzz_parity.py ${1:-600} 31000 > gpurun_out/r04_fuzz_${1:-600}cases_final_sources.log 2>&1; tail -4 gpurun_out/r04_fuzz_${1:-600}cases_final_sources.log | cut -c1-250
timeout 300 python tools/fuzz_parity.py ${2:-100} 33000 det > gpurun_out/r04_fuzz_${2:-100}cases_deterministic_strict_final_sources.log 2>&1; tail -4 gpurun_out/r04_fuzz_${2:-100}cases_deterministic_strict_final_sources.log | cut -c1-250
