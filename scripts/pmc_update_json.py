"""CSV counter files of scripts/pmc_update.sh -> the JSON bench.pmc_traffic() reads (copied to
profiles/rN_pmc_counters.json): per bench region the mean counters of its kernel(s) and
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the gfx950 correction of
MI355X_MICROARCH.md's HBM section (FETCH_SIZE reports half of wide coalesced reads) -- beside the
algorithmic bytes of the same launch (SURVEY 8(d) per-unit figures x M = 8192 images / rows)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

M = 8192
GEMM_ALG = 4 * (8192 * 3456 + 512 * 3456 + 8192 * 512)
# region -> ([kernel-name substrings, all summed per launch], algorithmic bytes per launch)
REGIONS = {
    "conv1_fwd": (["conv1_fwd_kernel"], M * (33280 + 4 * 7600)),
    "conv2_fwd": (["conv2_fwd_x6_kernel"], M * (4 * (7600 + 3456) + 512)),
    "convs_fwd": (["convs_fwd_fused_kernel"], M * (33280 + 4 * (7600 + 3456) + 512)),
    "conv2_bwd": (["conv2_bwd_x6_kernel"], M * (4 * (3456 + 2 * 7600) + 512)),
    "conv1_wgrad": (["conv1_wgrad_kernel"], M * (33280 + 4 * 7600)),
    "gemm_nt": (["gemm_nt_x6_kernel<128>"], GEMM_ALG),
    # (round 6: the input gradient is two launches -- 256-row tiles + a tail of 128-row tiles, told from
    #  the forward GEMM's 256 workgroups by its grid size)
    "gemm_nt_dgrad": (["gemm_nt_x6_kernel<256>", "gemm_nt_x6_kernel<128>#tail"], GEMM_ALG),
    "gemm_tn": (["gemm_tn_x6_kernel", "gemm_reduce_slots_kernel"], GEMM_ALG),
}
RENAME = {"SQ_VALU_MFMA_BUSY_CYCLES": "mfma_busy_cycles", "SQ_BUSY_CYCLES": "sq_busy_cycles",
          "GRBM_GUI_ACTIVE": "gui_active", "SQ_WAVE_CYCLES": "wave_cycles_quad",
          "SQ_WAIT_INST_ANY": "wait_inst_any", "SQ_INSTS_VALU": "insts_valu",
          "SQ_LDS_BANK_CONFLICT": "lds_bank_conflict", "SQ_LDS_IDX_ACTIVE": "lds_idx_active"}


def main(d):
    acc = defaultdict(lambda: defaultdict(list))       # kernel substring -> counter -> values
    subs = sorted({s for v in REGIONS.values() for s in v[0]}, key=len, reverse=True)
    for path in sorted(glob.glob(os.path.join(d, "*.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name") or ""
                key = next((s for s in subs if s in name), None)
                # "conv2_bwd_kernel" must not swallow conv2_bwd_x6_kernel (longest substring first
                # takes care of it), gemm_nt_x6_kernel<128> / <256> are told apart by the template
                if key is None:
                    continue
                if key == "gemm_nt_x6_kernel<128>" and int(float(row.get("Grid_Size") or 0)) not in (0, 256 * 512):
                    key += "#tail"
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"],
                                         cwd=os.path.dirname(os.path.abspath(__file__)),
                                         stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001  (the GPU box has no .git: stamped when copied to profiles/)
        commit = None
    out = {"note": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*; one pass each, no other "
                   "trace domain) over scripts/conv_bench.py 8192 --no-model and scripts/gemm_bench.py "
                   "(scripts/pmc_update.sh), means over the launches of each kernel; "
                   "hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md "
                   "(gfx950 FETCH_SIZE reports half of wide coalesced reads); a region of two kernels "
                   "(gemm_tn: partial tiles + their fixed-order sum) is the sum of both",
           "commit": commit, "kernels": {}}
    for region, (names, alg) in REGIONS.items():
        if not all(n in acc and "FETCH_SIZE" in acc[n] for n in names):
            continue
        e = defaultdict(float)
        for n in names:
            for c, v in acc[n].items():
                e[c] += sum(v) / len(v)
        hbm = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        r = {"kernels": names, "dispatches": max(len(v) for v in acc[names[0]].values()),
             "FETCH_SIZE_KB": round(e["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(e["WRITE_SIZE"], 1),
             "hbm_bytes_corrected": int(hbm), "alg_bytes": alg, "traffic_over_alg": round(hbm / alg, 3)}
        for src, dst in RENAME.items():
            if src in e:
                r[dst] = round(e[src], 1)
        out["kernels"][region] = r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
