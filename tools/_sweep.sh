run() { env "$@" python bench.py --steps 300 --warmup 30 $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '$EXTRA', d['value'], d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if 'blend' in k})"; }
run GMS_BWD_PF=0
run GMS_BWD_PF=2
run GMS_BWD_PF=1 GMS_TRIP_BWD=2
run GMS_BWD_PF=0 GMS_TRIP_BWD=2
run GMS_BWD_PF=2
