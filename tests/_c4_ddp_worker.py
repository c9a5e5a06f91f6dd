"""Worker of tests/test_gpu_c4.py (one process per rank, every rank on cuda:0 over gloo -- the 1-GPU test box; on the
8-GPU node the same code runs over RCCL with one GPU per rank).  BASELINE config 4: gs_multi_mesh, one camera view per rank
per step, gradient all-reduce.  For every exchange variant rank 0 saves the parameter gradients it holds after the exchange."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    workload, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    from games_hip import synthetic as syn
    from games_hip.ddp import OverlappedGradAllReduce, PackedGradExchange, ShFactorExchange
    from games_hip.model import HipGaussianMultiMeshModel
    from games_hip.render import PipelineParams, render
    scenes = syn.multi_mesh_scenes(workload, state="trained")
    size = scenes[0].meta["image"]
    model = HipGaussianMultiMeshModel.from_scenes(scenes, dev)
    params = model.parameters()
    cam = syn.orbit_camera(rank, width=size, height=size).to(dev)
    bg = torch.ones(3, device=dev)
    gcs = torch.load(os.path.join(os.path.dirname(out_path), "upstream.pt"))       # [world][3,H,W]: the serial run uses the same
    results = {}
    for variant in ("ring", "direct", "direct_ag", "factor", "packed"):
        sh_factor = variant == "factor"
        algo = "ring" if variant in ("factor", "packed") else variant
        packed = PackedGradExchange(params, model._features_dc, model._features_rest, world, average=False) if variant == "packed" else None
        reducer = OverlappedGradAllReduce(params, world if packed is None else 1, average=False, algorithm=algo, big_numel=1 << 16)
        exchange = ShFactorExchange(model._features_dc, model._features_rest, world, average=False) if sh_factor else None
        try:
            for p in params:
                p.grad = None
            if packed is not None:
                packed.enable()
            if exchange is not None:
                exchange.enable()
            model.update_alpha(); model.prepare_scaling_rot()
            img = render(cam, model, PipelineParams(), bg)["render"]
            (img * gcs[rank].to(dev)).sum().backward()
            if exchange is not None:
                exchange.start()
            reducer.finish()
            if exchange is not None:
                exchange.finish(model.get_xyz, model.active_sh_degree)
                exchange.disable()
            if packed is not None:
                packed.finish(model.get_xyz, model.active_sh_degree)
                packed.disable()
            torch.cuda.synchronize()
            results[variant] = [p.grad.detach().cpu().clone() for p in params]
        except Exception as e:  # noqa: BLE001 -- a collective this backend cannot run on CUDA tensors is reported, not hidden
            results[variant] = f"failed: {e!r}"
            if exchange is not None:
                exchange.disable()
            if packed is not None:
                packed.disable()
        finally:
            reducer.remove()
        ok = torch.tensor([0.0 if isinstance(results[variant], str) else 1.0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok) == 0.0 and not isinstance(results[variant], str):
            results[variant] = "failed on another rank"
    if rank == 0:
        torch.save(results, out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
