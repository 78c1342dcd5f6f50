import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from iefvad_amd import harness
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(1000, device="cuda", dtype=torch.float32) / 7
y = harness.gather_scores(x)
assert torch.equal(x, y), "gather mismatch"
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
print("RCCL world=1 gather OK", y.shape, float(t))
dist.destroy_process_group()
