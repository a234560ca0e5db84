"""Markdown rows for the SQ-counter table of profiles/rNN_pmc_*.md from the two passes of tools/pmc_run.sh.

usage: python tools/pmc_table.py gpurun_out/pmc_mlp [gpurun_out/pmc_cin ...]
"""
import sys


def parse(path):
    d, cur = {}, None
    for ln in open(path):
        if ln.strip() and not ln.startswith(" "):
            cur = ln.split("  (")[0].strip()
            d.setdefault(cur, {})
        elif cur and ln.strip():
            k, v = ln.split()
            d[cur][k] = float(v)
    return d


print("| kernel | matrix pipe busy | VALU busy | VALU instr / MFMA | LDS instr / MFMA | LDS unit busy | bank-conflict cycles | wait_any | wait_inst |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
for out in sys.argv[1:]:
    a, b = parse(out + "/pmc_a.txt"), parse(out + "/pmc_b.txt")
    for k in a:
        A, B = a[k], b.get(k)
        if not B or not B.get("SQ_INSTS_MFMA"):
            continue
        cu_cycles = B["GRBM_GUI_ACTIVE"] / 8            # GRBM_GUI_ACTIVE sums the 8 XCDs
        print(f"| `{k}` | {A['SQ_VALU_MFMA_BUSY_CYCLES'] / (cu_cycles * 1024) * 100:.1f} % | "
              f"{A.get('SQ_ACTIVE_INST_VALU', 0.0) / (cu_cycles * 1024) * 100:.1f} % | "
              f"{B['SQ_INSTS_VALU'] / B['SQ_INSTS_MFMA']:.2f} | {B['SQ_INSTS_LDS'] / B['SQ_INSTS_MFMA']:.2f} | "
              f"{B['SQ_LDS_IDX_ACTIVE'] / (cu_cycles * 256) * 100:.0f} % | {A['SQ_LDS_BANK_CONFLICT']:.2e} | "
              f"{A['SQ_WAIT_ANY'] / A['SQ_WAVE_CYCLES']:.2f} | {A['SQ_WAIT_INST_ANY'] / A['SQ_WAVE_CYCLES']:.2f} |")
