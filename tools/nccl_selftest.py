"""The RCCL (torch.distributed backend "nccl") code paths of the trainers at world_size 1, on a one-GPU box: process-group
init with device_id, the parameter broadcast, all_reduce_gradients on a CUDA bucket, and the fused trainer's flat-gradient
all-reduce between ewn_a2c_grad and ewn_a2c_apply.  Every N>1 rehearsal elsewhere uses gloo; this makes sure the first RCCL
call of an 8-GPU job is not the first time that code runs.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import ewn_gym_amd as ea  # noqa: E402
from ewn_gym_amd.a2c import ActorCritic, FusedA2CTrainer, all_reduce_gradients  # noqa: E402

out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
m = ActorCritic(5, 6).cuda()
for p in m.parameters():
    p.grad = torch.full_like(p, 2.0)
all_reduce_gradients(list(m.parameters()), force=True)          # one flattened bucket through RCCL
out["bucket_ok"] = bool(all(bool((p.grad == 2.0).all()) for p in m.parameters()))
N = 4096
env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="philox", shaped=True, reward=10.0, autoreset=True, shaped_refresh_on_reset=True)
env.reset(seeds=torch.arange(N, dtype=torch.int32))
tr = FusedA2CTrainer(env, n_steps=5, seed=0)                      # _sync_parameters: a no-op broadcast at world 1
tr.force_collective = True
a = tr.params.clone()
for _ in range(3):
    tr.collect_and_update()                                       # rollout, grad, all_reduce (RCCL), apply
torch.cuda.synchronize()
out["fused_updated"] = bool((tr.params != a).any()) and bool(torch.isfinite(tr.params).all())
out["grad_norm"] = float(tr.grad_norm)
one = torch.ones(1, dtype=torch.int32, device="cuda")
dist.all_reduce(one)
out["rccl_ranks"] = int(one.item())
dist.barrier()
dist.destroy_process_group()
print(json.dumps(out), flush=True)
