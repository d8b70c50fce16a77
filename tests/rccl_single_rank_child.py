"""Child process of test_e2e_gpu.py::test_rccl_collectives_single_rank: the collectives of the N > 1 paths on a ONE-rank
RCCL communicator (backend "nccl" on ROCm), on the device buffers those paths really use.  One GPU cannot show more
ranks over RCCL (it refuses two ranks on one device); the multi-rank logic is covered over gloo."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))

from rela_amd.learner import HipApexLearner, broadcast_weights, global_is_weights
from rela_amd.pyrela.apex import ApexAgent
from rela_amd.pyrela.net import AtariFFNet

A, B = 6, 32
torch.manual_seed(3)
agent = ApexAgent(lambda: AtariFFNet(A), 3, 0.99).to("cuda:0")
learner = HipApexLearner.from_agent(agent, B)
rng = np.random.default_rng(0)
from types import SimpleNamespace

obs = lambda: {"s": torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda"),
               "eps": torch.zeros(B, 1, device="cuda"), "legal_move": torch.ones(B, A, device="cuda")}
batch = SimpleNamespace(obs=obs(), next_obs=obs(), action={"a": torch.randint(0, A, (B,), device="cuda")},
                        reward=torch.randn(B, device="cuda"), terminal=torch.zeros(B, dtype=torch.bool, device="cuda"),
                        bootstrap=torch.ones(B, device="cuda"))
w = torch.rand(B, device="cuda") + 0.5
loss, prio = learner.backward(batch, w)
g = learner.flat()[1]
before = g.clone()
dist.all_reduce(g, op=dist.ReduceOp.SUM)  # the flat gradient bucket of replicated learners
g.div_(1)
same_grad = bool(torch.equal(g, before))
learner.apply()
raw = torch.rand(B, device="cuda") + 0.1
isw = global_is_weights(raw, float(raw.sum()) * 3, 1000, 0.4)  # SUM of (sum, size), MAX of the maximum
ref = (1000.0 * (raw / (float(raw.sum()) * 3))).pow(-0.4)
ref = ref / ref.max()
flat = learner.flat()[0].clone()
broadcast_weights(flat, src=0)  # the weight publish to actor-only ranks
torch.cuda.synchronize()
print(json.dumps({"same_grad": same_grad, "isw_err": float((isw - ref).abs().max()),
                  "bcast_same": bool(torch.equal(flat, learner.flat()[0])), "backend": dist.get_backend()}))
dist.destroy_process_group()
learner.close()
