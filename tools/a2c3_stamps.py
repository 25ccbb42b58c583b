"""Region timing of k_a2c_grad3 (diagnostic): build with EWN_HIPCC_FLAGS=-DA2C3_STAMPS, then
   python tools/a2c3_stamps.py   -> shader-clock ticks (100 MHz s_memtime) per region of the step loop, block 0 wave 0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ewn_gym_amd as ea  # noqa: E402
from ewn_gym_amd.a2c import FusedA2CTrainer  # noqa: E402

N = 65536
env = ea.VecEWN(N, opponent_policy="minimax", max_depth=3, rng="philox", shaped=True, reward=10.0, illegal_move_reward=-1.0,
                illegal_move_tolerance=10, autoreset=True, shaped_refresh_on_reset=True, philox_key=1)
env.reset(seeds=torch.arange(N, dtype=torch.int32))
tr = FusedA2CTrainer(env, n_steps=5, learning_rate=3e-4, seed=0)
for _ in range(5):
    tr.collect_and_update()
torch.cuda.synchronize()
f = tr.scratch.view(torch.float32)
tail = f[-64:].cpu().tolist()
names = ["decode", "L1+tanh+split0", "L2 loop", "tanh h2+split0 / value head", "head loop", "logit gather", "loss+dop", "dh2/dU+h2k", "g2+dk",
         "split g2 0", "dh1 loop", "g1U+db2", "ka split", "dW2 loop", "dW1", "-"]
for net, label in ((1, "value pass (6 steps x 2 tiles)"), (0, "policy pass (5 steps x 2 tiles)")):
    v = tail[net * 16:net * 16 + 16]
    tot = sum(v)
    print(label, "total ticks", tot)
    for n, x in zip(names, v):
        print("  %-28s %8.0f  %5.1f %%" % (n, x, 100.0 * x / max(tot, 1.0)))
