cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
GMS_MICRO=0 timeout 100 python - > gpurun_out/r03_deep_buggy_micro0.log 2>&1 <<'P'
import sys, os, torch
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import conftest
import test_gpu_raster as T
from games_hip import synthetic as syn
import _util as U
import diff_gaussian_rasterization as dgr
sc, cam = syn.random_scene(6000, seed=33, extent=0.5, scale_lo=0.08, scale_hi=0.4, opacity_lo=0.02, opacity_hi=0.3), syn.orbit_camera(3, width=48, height=48, radius=2.0)
for call in range(3):
    try:
        h, o, rep, g = T._check(T._inputs(sc), U.settings_kwargs(cam, torch.tensor([0.3, 0.1, 0.2])), 48, 48)
        print("call", call, "ok", rep, dgr.last_stats())
    except Exception as e:
        print("call", call, "FAILED", repr(e)[:1500]); print(dgr.last_stats())
P
tail -c 3000 gpurun_out/r03_deep_buggy_micro0.log
