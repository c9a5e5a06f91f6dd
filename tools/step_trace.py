"""List every GPU kernel of one headline step in launch order with the torch op that launched it (GPU box)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
dev = torch.device("cuda", 0)
scene = syn.mesh_scene("c2_hotdog_like", state="trained")
size = scene.meta["image"]
model = HipGaussianMeshModel.from_scene(scene, dev)
cam = syn.orbit_camera(0, width=size, height=size).to(dev)
bg = torch.ones(3, device=dev); pipe = PipelineParams(); params = model.parameters()
inv = 1.0 / (3.0 * size * size); c = torch.tensor(-0.5 * inv, device=dev)
def step():
    model.update_alpha(); model.prepare_scaling_rot()
    image = render(cam, model, pipe, bg)["render"]
    with torch.no_grad():
        grad = torch.add(c, image, alpha=inv)
    image.backward(grad)
    for p in params: p.grad = None
for _ in range(10): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
for e in evs:
    print(f"{e.time_range.start - t0:9.1f} us  +{e.time_range.elapsed_us():7.1f}  {e.name[:90]}")
