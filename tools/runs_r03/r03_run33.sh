cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload c5_flame_like_1m --profile-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"'; }
echo default; run
echo "BWD_WPB=4"; GMS_BWD_WPB=4 run
echo "FWD_WPB=4 BWD_WPB=4"; GMS_FWD_WPB=4 GMS_BWD_WPB=4 run
echo "TRIP_BWD=2"; GMS_TRIP_BWD=2 run
echo "TRIP=2"; GMS_TRIP=2 run
echo "UNIT_RUN=1"; GMS_UNIT_RUN=1 run
echo "UNIT_RUN=16"; GMS_UNIT_RUN=16 run
echo "MICRO=1 (forced micro, L=256)"; GMS_MICRO=1 run
