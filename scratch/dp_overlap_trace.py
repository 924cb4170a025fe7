"""DSFVT train steps in a ONE-rank RCCL group with the gradient reducers active (LVT_DP_SINGLE_RANK): the launches, the side
stream and the joins of the data-parallel path on a single-GPU box.  Run under `rocprofv3 --kernel-trace`; summarise with
scratch/dp_overlap_summary.py.   python scratch/dp_overlap_trace.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LVT_DP_SINGLE_RANK"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
import torch
import torch.distributed as dist
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
leg = bench.DsfvtLeg("cuda:0", 1, 0, 0, 64, 4, dp=True)
r = leg.model._reducers[0]
print("buckets", len(r.buckets), "bytes", r.bytes_per_backward, "active", r.world > 1, "avg", r._avg)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for i in range(3 + steps):
    leg.step(i)
torch.cuda.synchronize()
dist.destroy_process_group()
print("done")
