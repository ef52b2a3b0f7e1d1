"""Build libspk_hip.so in-tree with hipcc for gfx950 (no torch headers involved).

    python -m schnetpack_amd.csrc.build [--force]

The shared library travels to the GPU box with the repo snapshot (it is git-ignored, not
gpurun-ignored).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["spk_util.hip", "spk_dense.hip", "spk_chain.hip", "spk_cfconv.hip", "spk_schnet.hip", "spk_schnet_mol.hip", "spk_painn.hip", "spk_painn_tile.hip", "spk_painn_blk.hip", "spk_painn_mol.hip", "spk_tabfilter.hip", "spk_nbl.hip", "spk_md.hip", "spk_potential.hip", "spk_train.hip", "spk_fm.hip"]
HEADERS = ["spk_common.h", "spk_painn_msg.h", "spk_painn_mol.h", "spk_painn_blk.h", "spk_pack.h", "spk_gemm_tn.h", "spk_fm_engine.h", "spk_fm_kernels.h", "spk_fm_chain.h", "spk_split.h", "spk_filter_split.h", os.path.join("..", "..", "include", "spk_hip.h")]
LIB = os.path.join(HERE, "libspk_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-mcode-object-version=5", "-Wall", "-Wno-unused-function"] + os.environ.get("SPK_EXTRA_FLAGS", "").split()
# Per-file flags.  spk_cfconv.hip is built WITHOUT the SLP vectoriser: with it the split-precision pair backward (k_cfconv_pair_t_sp) came out
# with its two per-pair sums packed into v_pk_* / v_pk_mov_b32 instructions between the f16 matrix instructions, and on the device the LOW half
# of those pairs was wrong in lanes 48..63 of about one tile iteration in 10^4, differently in every run (profiles/r06_box_split_glitch.md: the
# operand registers compare equal to a reload, fences / late loads / waits do not help, the scalar code is clean in every run; same speed).
FILE_FLAGS = {"spk_cfconv.hip": ["-fno-slp-vectorize"]}


def flags_for(src):
    return FLAGS + FILE_FLAGS.get(os.path.basename(src), [])


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objs = []
    procs = []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [_hipcc()] + flags_for(src) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            print("FAILED:", s)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_native_example(force, verbose)
    build_torch_ops(force, verbose)
    build_jit_client(force, verbose)
    return LIB


TORCH_SRC = os.path.join(HERE, "spk_torch.cpp")
TORCH_LIB = os.path.join(HERE, "libspk_torch.so")


def build_torch_ops(force=False, verbose=True):
    """libspk_torch.so: the TORCH_LIBRARY(spk_hip) operator registrations (spk_torch.cpp; host C++ only, every operator
    calls into libspk_hip.so), compiled against the installed PyTorch-ROCm headers and linked next to libspk_hip.so."""
    if not (force or _stale(TORCH_LIB, [TORCH_SRC, os.path.join(HERE, "spk_torch_train.h"), os.path.join(HERE, "spk_torch_fm.h"), LIB, os.path.join(HERE, HEADERS[-1])])):
        return TORCH_LIB
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in ce.include_paths()] + ["-I/opt/rocm/include", TORCH_SRC, "-o", TORCH_LIB,
            "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip", "-L" + HERE, "-lspk_hip",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_LIB


RUN_SRC = os.path.join(HERE, "..", "..", "examples", "native", "spk_run.c")
RUN_BIN = os.path.join(HERE, "spk_run")


def build_native_example(force=False, verbose=True):
    """Plain-C program on the deployment runtime (examples/native/spk_run.c): gcc, links libspk_hip.so only."""
    if not os.path.exists(RUN_SRC):
        return None
    if force or _stale(RUN_BIN, [RUN_SRC, LIB, os.path.join(HERE, HEADERS[-1])]):
        cmd = [os.environ.get("CC", "gcc"), "-O2", "-Wall", "-std=c11", "-D_POSIX_C_SOURCE=199309L", RUN_SRC, "-o", RUN_BIN,
               "-L" + HERE, "-lspk_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return RUN_BIN


JIT_SRC = os.path.join(HERE, "..", "..", "examples", "native", "spk_jit_client.cpp")
JIT_BIN = os.path.join(HERE, "spk_jit_client")


def build_jit_client(force=False, verbose=True):
    """C++ libtorch client of a deployed TorchScript archive (examples/native/spk_jit_client.cpp): what LAMMPS' pair style does
    (interfaces/lammps/pair_schnetpack.cpp:125-131, :285-328) -- torch::jit::load + forward, no Python -- after dlopen of the two
    operator libraries.  Links libtorch only; the spk_hip libraries are given at run time."""
    if not os.path.exists(JIT_SRC):
        return None
    if not (force or _stale(JIT_BIN, [JIT_SRC])):
        return JIT_BIN
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O1", "-std=c++17", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in ce.include_paths()] + ["-I/opt/rocm/include", JIT_SRC, "-o", JIT_BIN, "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10",
            "-ltorch_hip", "-lc10_hip", "-ldl", "-Wl,--no-as-needed", "-Wl,-rpath," + tlib, "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return JIT_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
