"""cProfile of the host side of the headline step (GPU box)."""
import cProfile, pstats, os, sys, io
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import torch
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
dev = torch.device("cuda", 0)
scene = syn.mesh_scene("c2_hotdog_like", state="trained")
size = scene.meta["image"]
model = HipGaussianMeshModel.from_scene(scene, dev)
cam = syn.orbit_camera(0, width=size, height=size).to(dev)
bg = torch.ones(3, device=dev); pipe = PipelineParams(); params = model.parameters()
inv = 1.0 / (3.0 * size * size)
def step():
    model.update_alpha(); model.prepare_scaling_rot()
    image = render(cam, model, pipe, bg)["render"]
    with torch.no_grad():
        grad = (image - 0.5) * inv
    image.backward(grad)
    for p in params: p.grad = None
for _ in range(30): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(28); print(s.getvalue()[:6000])
