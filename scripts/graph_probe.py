"""Which part of a force call survives HIP-graph capture?  Each stage runs in its own process."""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGES = ["rep_nograd:256", "force_call:32", "force_call:64", "force_call:128", "force_call:256", "force_call_simple:256", "force_call_painn:256", "scatter_big:256", "dense_big:256"]


def run(stage):
    faulthandler.enable()
    stage, frames = stage.split(":")
    frames = int(frames)
    import torch
    from oracle import spk_oracle as O
    from schnetpack_amd import _lib, model as M, ops, synthetic as S
    dev = torch.device("cuda:0")
    b = S.molecule_batch("aspirin", frames, seed=0)
    kind = "painn" if "painn" in stage else "schnet"
    m = M.build_model(kind)
    M.load_reference_params(m, O.init_schnet_params() if kind == "schnet" else O.init_painn_params(), O.init_atomwise_params(128, seed=1))
    if "simple" in stage:
        _lib.set_variant(_lib.VARIANT_SIMPLE)
    m = m.to(dev).eval()
    inp = M.batch_to_inputs(b, dev)
    x = torch.randn(256, 128, device=dev)
    w = torch.randn(128, 128, device=dev)
    idx = inp["_idx_i"]

    def body():
        if stage == "torch_only":
            return (x @ w).sum()
        if stage == "dense":
            return ops.dense_raw(x, w, None, 1)[0]
        if stage == "scatter_big":
            xs = torch.ones(idx.shape[0], 128, device=dev)
            n = int(inp["_atomic_numbers"].shape[0])
            return ops._scatter_raw(xs, idx, n, 0, ops.segment_rowptr(idx, n))
        if stage == "dense_big":
            xb = torch.randn(5376, 128, device=dev)
            return ops.dense_raw(xb, w, None, 1)[0]
        if stage == "rep_nograd":
            with torch.no_grad():
                d = dict(inp)
                d["_Rij"] = inp["_positions"][inp["_idx_j"]] - inp["_positions"][inp["_idx_i"]]
                return m.representation(d)["scalar_representation"]
        out = m(dict(inp))
        return out["forces"]

    for _ in range(3):
        ref = body()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print(stage, "warm ok", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    print(stage, "captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(stage, "replayed; maxdiff", float((out - ref).abs().max()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for st in STAGES:
            r = subprocess.run([sys.executable, __file__, st], capture_output=True, text=True, timeout=300)
            print("=== %s rc=%d" % (st, r.returncode))
            print(r.stdout[-600:])
            print(r.stderr[-1500:])
