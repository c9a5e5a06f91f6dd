"""cProfile of the host side of a step (GPU box).  usage: host_profile.py [full]   (full = + fused loss + FusedAdam)"""
import cProfile, pstats, os, sys, io
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import torch
from games_hip import synthetic as syn
from games_hip.model import HipGaussianMeshModel
from games_hip.render import PipelineParams, render
from games_hip.loss import l1_ssim_loss
full = sys.argv[1:] == ["full"]
dev = torch.device("cuda", 0)
scene = syn.mesh_scene("c2_hotdog_like", state="trained")
size = scene.meta["image"]
model = HipGaussianMeshModel.from_scene(scene, dev)
cam = syn.orbit_camera(0, width=size, height=size).to(dev)
bg = torch.ones(3, device=dev); pipe = PipelineParams(); params = model.parameters()
inv = 1.0 / (3.0 * size * size); c = torch.tensor(-0.5 * inv, device=dev)
gt = torch.rand(3, size, size, device=dev)
if full:
    model.training_setup(1e-12, 1e-12, 1e-12, 1e-12, 1e-12, fused=True)
def step():
    model.update_alpha(); model.prepare_scaling_rot()
    image = render(cam, model, pipe, bg)["render"]
    if full:
        l1_ssim_loss(image, gt, 0.2).backward()
        model.optimizer.step(); model.optimizer.zero_grad(set_to_none=True)
    else:
        with torch.no_grad():
            grad = torch.add(c, image, alpha=inv)
        image.backward(grad)
        for p in params: p.grad = None
for _ in range(30): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("cumtime"); ps.print_stats(45); print(s.getvalue()[:9000])
