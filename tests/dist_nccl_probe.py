"""Manual probe (not collected by pytest): the exact distributed calls bench.py makes for N > 1 -- NCCL (=RCCL)
process group, barrier, MAX all-reduce of the elapsed time on the device -- on however many GPUs are visible.
usage: python -m torch.distributed.run --nproc-per-node N tests/dist_nccl_probe.py"""
import os
import sys

import torch
import torch.distributed as dist

lr = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(lr)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
dist.init_process_group('nccl', device_id=torch.device('cuda', lr))
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.0 + dist.get_rank()], dtype=torch.float64, device=f'cuda:{lr}')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == float(dist.get_world_size())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpyc_amd import multigpu  # noqa: E402,F401  (the module bench.py / the mirror use for sharding)
print('nccl probe ok: world', dist.get_world_size(), 'backend', dist.get_backend())
dist.destroy_process_group()
