"""GPU box: the torch.distributed (RCCL) calls bench.py makes for N > 1 -- init_process_group("nccl", device_id=...),
barrier, all_reduce(MAX) of a float64 tensor, all_gather -- on a world of ONE rank, the most a 1-GPU box can run: checks
that RCCL loads and those calls work on this image (the N = 8 run itself is the driver's).   python tools/rccl_selfcheck.py"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"; os.environ["LOCAL_RANK"]="0"
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
dist.barrier()
t=torch.tensor([1.25],dtype=torch.float64,device="cuda"); dist.all_reduce(t,op=dist.ReduceOp.MAX)
g=[torch.zeros_like(t) for _ in range(1)]; dist.all_gather(g,t)
print("rccl ok", float(t.item()), float(g[0].item()), dist.get_world_size())
dist.barrier(); dist.destroy_process_group()
