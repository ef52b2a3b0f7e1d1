"""Register / scratch budget of the hot kernels, from hipcc's kernel-resource remarks (cross-compiles: no GPU needed).

Round 2 measured what scratch costs on this part: a few dozen spilled values, parked in the prologue of a kernel, added 12-14 k
cycles to every workgroup of the molecule-resident SchNet launches (HISTORY.md section 4.1a), and 240 B/lane of scratch in
`k_gemm_pair<true>` cost the training step 6 % (section 7).  A refactoring that pushes one of these kernels back over its register
budget compiles, passes every numerical test and is slower -- so the budget is pinned here.
"""
import os
import re
import subprocess

import pytest

from schnetpack_amd.csrc import build as B

# mangled-name fragment -> (max scratch bytes per lane, min waves per SIMD)
BUDGET = {
    # (round 4: the pair Dense kernels of the training step and the double-buffered weight-gradient loop -- at 32 row pairs per batch the
    #  latter went 940 B/lane into scratch with its 64-bit lane addresses; it stays at 16)
    "spk_dense.hip": {"k_gemm_pairILb1E": (0, 2), "k_gemm_pairILb0E": (0, 2), "k_dense_mfmaILi0ELb1ELi0E": (0, 2), "k_dense_mfmaILi0ELb0ELi0E": (0, 2),
                      "k_dense_dual_sk": (0, 2), "k_dense_dual_tiles": (0, 2), "k_dense_mfma_skILi0ELb0ELi0ELi4ELi4E": (0, 4)},
    "spk_fm.hip": {"k_gemm_tn_batched11GemmTnBatch": (0, 2), "k_fm_painn_msg_T_dualIfE": (0, 2), "k_fm_painn_msg_tIfE": (0, 2), "k_fm_cfconv_T_dualIfE": (0, 4)},
    "spk_painn.hip": {"k_painn_mixing_fwd8ILi128E": (0, 2), "k_painn_mixing_bwd8ILi128E": (0, 2), "k_painn_mixing_fwd8sILi128E": (0, 2), "k_painn_mixing_bwd8sILi128E": (0, 2)},
    # row-tile kernels of the PaiNN message (round 6; backward <F, KPB, geometry, sums, mu == 0, batch, skin>, forward <F, KPB, mu == 0, batch, skin>):
    # the geometry pass sits at the 256-register limit -- in ONE sweep it spilled 140-324 B per lane and every change of a few registers moved it by
    # 20-40 % (profiles/r06_painn_box.md); the two-sweep form keeps a few dozen bytes of prologue spills
    "spk_painn_tile.hip": {"k_painn_msg_rowtile_bwdILi128ELi3ELb1ELb0ELb0ELi8ELb0E": (64, 2), "k_painn_msg_rowtile_bwdILi128ELi3ELb1ELb0ELb1ELi8ELb0E": (32, 2),
                           "k_painn_msg_rowtile_bwdILi128ELi3ELb0ELb1ELb0ELi8ELb0E": (0, 2),
                           "k_painn_msg_rowtile_fwdILi128ELi3ELb0ELi8ELb0E": (0, 2), "k_painn_msg_rowtile_fwdILi128ELi3ELb1ELi8ELb0E": (0, 2)},
    "spk_chain.hip": {"k_dense_chain_sp": (0, 2)},
    # row-tile forward of the SchNet convolution (round 6): sixteen waves per workgroup = 128 registers; a few prologue spills
    "spk_cfconv.hip": {"k_cfconv_rowtile_fwdILi3ELi16E": (96, 4)},
    # the two molecule-resident launches sit AT the 256-register limit of two waves per SIMD; the metadata reports a small
    # private segment although no scratch instruction is on a hot path
    # (round 6: second template flag = the split-precision matrix path, the default; it carries the split operands of a tile on top)
    "spk_schnet_mol.hip": {"k_schnet_mol_fwdILi3ELb0E": (128, 2), "k_schnet_mol_fwdILi3ELb1E": (192, 2), "k_schnet_mol_bwdILi3E": (128, 2)},
    # molecule-resident PaiNN (round 3): both launches at the 256-register limit; what is left in scratch are values parked in the
    # prologue and a handful of reloads in the message loops -- when whole prefetched weight tiles were being spilled behind their
    # loads the figures were 2 176 / 652 B per lane and every Dense phase waited for a chain of L2 round trips (HISTORY.md 4.3a)
    # instances <n_rbf, tiled, potential> / <n_rbf, potential>: the row form (default) with and without the two-launch potential; the
    # tile-form experiment (SPK_PM_TILED=1) is not a budgeted path
    # (round 6: last template flag = split Dense phases, the default)
    "spk_painn_mol.hip": {"k_painn_mol_fwdILi20ELb0ELb0ELb0E": (256, 2), "k_painn_mol_fwdILi20ELb0ELb1ELb0E": (320, 2),
                          "k_painn_mol_fwdILi20ELb0ELb0ELb1E": (288, 2), "k_painn_mol_fwdILi20ELb0ELb1ELb1E": (352, 2),
                          "k_painn_mol_bwdILi20ELb0E": (448, 2), "k_painn_mol_bwdILi20ELb1E": (448, 2)},
}


def _resources(src):
    path = os.path.join(B.HERE, src)
    cmd = [B._hipcc()] + B.flags_for(path) + ["-c", path, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900).stderr
    rows, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark: [^ ]+ +(?:Function Name|Name): (\S+)", line)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        for key, pat in (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("vgpr", r" VGPRs: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


@pytest.mark.parametrize("src", sorted(BUDGET))
def test_hot_kernels_keep_their_register_budget(src):
    try:
        B._hipcc()
    except Exception as exc:  # pragma: no cover
        pytest.skip("no hipcc: %s" % exc)
    rows = _resources(src)
    assert rows, "hipcc printed no kernel-resource remarks for %s" % src
    for frag, (max_scratch, min_occ) in BUDGET[src].items():
        hits = {n: r for n, r in rows.items() if frag in n}
        assert hits, "kernel %s not found in %s (renamed? update BUDGET)" % (frag, src)
        for name, r in hits.items():
            assert r.get("scratch", 0) <= max_scratch, "%s: %d B/lane of scratch (budget %d)" % (name, r.get("scratch", 0), max_scratch)
            assert r.get("occ", 0) >= min_occ, "%s: %d waves/SIMD (budget >= %d)" % (name, r.get("occ", 0), min_occ)
