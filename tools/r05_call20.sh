# round 5, GPU call 20: the host side of the headline step on this box (tools/host_chain.py), with the bench line beside it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=r05t
(timeout 60 python tools/host_chain.py 400; timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py:', j['value'], 'it/s', j['ms_per_step'], 'ms')"; grep -m1 "model name" /proc/cpuinfo; nproc) > gpurun_out/${T}_host_chain.txt 2>&1
cat gpurun_out/${T}_host_chain.txt | cut -c1-250
