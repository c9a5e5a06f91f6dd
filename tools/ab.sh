# A/B of bench.py under environments, one line per arm: value, ms/step and the compositing / binning kernels' HIP-event durations.
#   bash tools/ab.sh TAG "ENV_A" "ENV_B" ... [-- bench args]      (an arm of "-" is the plain environment)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
ARMS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARMS+=("$1"); shift; done; [ "$1" = "--" ] && shift
mkdir -p $R/gpurun_out
for A in "${ARMS[@]}"; do
  E=$A; [ "$A" = "-" ] && E=""
  env $E python $R/bench.py --no-cpu-baseline --steps 150 --warmup 20 --profile-steps 20 "$@" 2>/dev/null | tail -1 > /tmp/ab_line.json
  python - "$A" <<'PY' | tee -a $R/gpurun_out/${TAG}_ab.txt
import json, sys
try:
    d = json.load(open("/tmp/ab_line.json"))
except Exception as e:
    print(f"{sys.argv[1]:40s} FAILED {e}"); sys.exit(0)
k = d.get("kernels", {})
def us(n):
    v = k.get(n)
    return round(v["avg_us"] * v.get("launches_per_step", 1), 1) if isinstance(v, dict) else None
names = ["blend_head", "blend_fwd", "blend_finalize", "blend_bwd", "emit_instances", "tile_sort", "preprocess_fwd", "preprocess_bwd", "mesh_fwd", "mesh_bwd_face"]
print(f"{sys.argv[1]:40s} {d['value']:8.1f} it/s {d['ms_per_step']:.4f} ms  " + " ".join(f"{n.replace('blend_','b_').replace('preprocess','pre')}={us(n)}" for n in names if us(n) is not None))
PY
done
