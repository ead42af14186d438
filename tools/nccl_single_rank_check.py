import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from nerf_loc_amd.sharding import gather_ray_outputs
out={"rgb":torch.rand(7,3,device="cuda"),"depth":torch.rand(7,device="cuda"),"mask":torch.rand(7,device="cuda")>0.5,"feat":torch.rand(7,192,device="cuda")}
g=gather_ray_outputs(out, dist)
for k in out: assert torch.equal(g[k], out[k]), k
dist.barrier(); torch.cuda.synchronize()
t=torch.tensor([1.0,2.0],device="cuda",dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl world=1 gather ok", t.tolist())
dist.destroy_process_group()
