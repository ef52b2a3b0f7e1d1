#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider --tb=short -k "tabulated" 2>&1 | grep -v "amdgpu\|Warning\|warn" | tail -8
python - <<'PY'
import time, torch, json
from schnetpack_amd import _lib, model as M, synthetic as S, tabulate
dev = torch.device("cuda:0")
torch.manual_seed(0)
b = S.water_box(n_side=22, seed=0)
inp = M.batch_to_inputs(b, dev)
E = int(b["idx_i"].shape[0])
for kind in ("painn", "schnet"):
    torch.manual_seed(0)
    m = M.build_model(kind).to(dev).eval()
    def call():
        return m(dict(inp))["forces"].detach()
    def t(reps=8):
        for _ in range(2): call()
        torch.cuda.synchronize(); c0 = time.perf_counter()
        for _ in range(reps): call()
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - c0) / reps
    f0 = call().clone(); ms0 = t()
    tabulate.tabulate_filters(m.representation, 512)
    _lib.profile_enable(True); _lib.profile_report()
    f1 = call().clone()
    tags = _lib.profile_report(); _lib.profile_enable(False)
    ms1 = t()
    tabulate.clear_filter_tables()
    print(kind, "water box: contract %.3f ms, tabulated %.3f ms (%.2fx), forces rel diff %.2e" % (ms0, ms1, ms0 / ms1, float((f1 - f0).abs().max() / f0.abs().max())),
          {k: round(1e3 * v[1] / v[0], 1) for k, v in tags.items() if "tab" in k})
PY
