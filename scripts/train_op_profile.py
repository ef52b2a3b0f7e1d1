"""Which framework operators launch kernels in one eager training step (SchNet / PaiNN), and from where: torch profiler grouped by
operator name and python stack (forward side; autograd-engine work has no python frame)."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from schnetpack_amd import model as M, synthetic as S
from schnetpack_amd.train import GraphedTrainStep
kind = sys.argv[1] if len(sys.argv) > 1 else "schnet"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = M.build_model(kind).to(dev)
b = S.molecule_batch("aspirin", 8, seed=0)
N, E = int(b["Z"].shape[0]), int(b["idx_i"].shape[0])
ts = GraphedTrainStep(m, N, 8, E + 64, 5.0, lr=1e-3, use_graph=False)
Et, Ft = torch.zeros(8, device=dev), torch.zeros(N, 3, device=dev)
ts.load(b, Et, Ft)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as pr:
    ts.step()
    torch.cuda.synchronize()
evs = pr.events()
kern = [e for e in evs if str(getattr(e, "device_type", "")).endswith("CUDA") and not e.name.lower().startswith(("memcpy", "memset"))]
print("kernels in one step:", len(kern))
cnt = collections.Counter(e.name[:70] for e in kern)
for k, v in cnt.most_common(40):
    print("  %4d  %s" % (v, k))
# CPU-side operators that launched something: leaf ops with cuda children
ops = collections.Counter()
src = collections.defaultdict(collections.Counter)
for e in evs:
    if str(getattr(e, "device_type", "")).endswith("CPU") and e.name.startswith(("aten::", "spk_hip::")):
        nk = sum(1 for k in e.kernels) if hasattr(e, "kernels") else 0
        if nk:
            ops[e.name] += nk
            st = [f for f in (e.stack or []) if "schnetpack_amd" in f]
            src[e.name][st[0].split("schnetpack_amd/")[-1][:60] if st else "(autograd engine / no python frame)"] += nk
print("operators (kernels launched):")
for k, v in ops.most_common(30):
    print("  %4d  %-40s %s" % (v, k, dict(src[k].most_common(4))))
