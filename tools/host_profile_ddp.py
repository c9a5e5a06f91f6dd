"""cProfile of the host side of a view-parallel step inside a ONE-RANK RCCL group (GPU box): where the 0.14 ms go that the
exchange costs when it moves nothing.  usage: host_profile_ddp.py [dense|factor]"""
import cProfile, pstats, os, sys, io, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-mesh-splatting_amd"))
import torch
import torch.distributed as dist
from games_hip import synthetic as syn
from games_hip.ddp import OverlappedGradAllReduce, ShFactorExchange
from games_hip.model import HipGaussianMultiMeshModel
from games_hip.render import PipelineParams, render
mode = sys.argv[1] if len(sys.argv) > 1 else "factor"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
scenes = syn.multi_mesh_scenes("c4_ficus_like", state="trained")
size = scenes[0].meta["image"]
model = HipGaussianMultiMeshModel.from_scenes(scenes, dev)
cam = syn.orbit_camera(0, width=size, height=size).to(dev)
bg = torch.ones(3, device=dev); pipe = PipelineParams(); params = model.parameters()
inv = 1.0 / (3.0 * size * size); c = torch.tensor(-0.5 * inv, device=dev)
reducer = OverlappedGradAllReduce(params, 1, average=False, force=True, algorithm="ring")
exchange = ShFactorExchange(model._features_dc, model._features_rest, 1, force=True) if mode == "factor" else None
def step():
    if exchange is not None: exchange.enable()
    model.update_alpha(); model.prepare_scaling_rot()
    image = render(cam, model, pipe, bg)["render"]
    if exchange is not None: exchange.watch(model.get_xyz)
    with torch.no_grad():
        grad = torch.add(c, image, alpha=inv)
    image.backward(grad)
    reducer.finish()
    if exchange is not None:
        exchange.finish(model.get_xyz, model.active_sh_degree); exchange.disable()
    for p in params: p.grad = None
for _ in range(300): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print(f"{mode}: {1e3 * (time.perf_counter() - t0) / 300:.4f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(28); print(s.getvalue()[:7000])
dist.destroy_process_group()
