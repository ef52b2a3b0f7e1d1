"""Register / LDS / scratch / occupancy table of every gfx950 kernel in libspk_hip.so, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles, no GPU needed).

    python scripts/kernel_resources.py > profiles/r01_kernel_resources.md
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from schnetpack_amd.csrc import build as B  # noqa: E402

rows = []
for src in B.SOURCES:
    path = os.path.join(B.HERE, src)
    cmd = [B._hipcc()] + B.flags_for(path) + ["-c", path, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line)
        if m:
            if cur:
                rows.append(cur)
            cur = {"file": src, "name": m.group(2)}
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                cur[key] = int(m.group(1))
    if cur:
        rows.append(cur)


def demangle(names):
    try:
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
            try:
                p = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
                if p.returncode == 0 and p.stdout:
                    return p.stdout.splitlines()
            except OSError:
                continue
        return names
    except Exception:
        return names


names = demangle([r["name"] for r in rows])
print("# Kernel resources (gfx950, hipcc -O3; static LDS only -- the MFMA kernels add dynamic LDS at launch)\n")
print("| file | kernel | VGPR | AGPR | SGPR | scratch B/lane | waves/SIMD | static LDS B |")
print("|---|---|---|---|---|---|---|---|")
pairs = [(r, n) for r, n in zip(rows, names) if "rocprim" not in n]      # library kernels instantiated from rocPRIM are not listed
for r, n in pairs:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    print("| %s | `%s` | %s | %s | %s | %s | %s | %s |" % (r["file"], n[:110], r.get("vgpr", ""), r.get("agpr", ""), r.get("sgpr", ""), r.get("scratch", ""),
                                                       r.get("occ", ""), r.get("lds", "")))
spilling = [(r, n) for r, n in pairs if r.get("scratch", 0) > 0]
print("\nKernels with scratch (spills): %d of %d" % (len(spilling), len(pairs)))
for r, n in spilling:
    print("* `%s`: %d B/lane" % (re.sub(r"\(.*$", "", re.sub(r"^void ", "", n))[:140], r["scratch"]))
