"""summary.json of scripts/pmc_conv.sh -> profiles/rN_conv_pmc_counters.json (the file
bench.pmc_traffic() reads): HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950
correction of MI355X_MICROARCH.md's HBM section, beside the algorithmic bytes of the same launch."""
import json
import sys

M = 8192
ALG = {  # bench name -> (kernel name in the trace, algorithmic bytes per launch at M images)
    "conv1_fwd": ("conv1_fwd_kernel", M * (33280 + 4 * 7600)),
    "conv2_fwd": ("conv2_fwd_x6_kernel", M * 4 * (7600 + 3456)),
    "conv2_dgrad": ("conv2_dgrad_kernel", M * 4 * (2 * 3456 + 2 * 7600)),
    "conv2_wgrad": ("conv2_wgrad_kernel", M * 4 * (2 * 3456 + 7600)),
    "conv2_bwd": ("conv2_bwd_kernel", M * 4 * (2 * 3456 + 2 * 7600)),
    "conv1_wgrad": ("conv1_wgrad_kernel", M * (33280 + 4 * 7600)),
}


def main(path):
    with open(path) as f:
        s = json.load(f)
    out = {"note": "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*; one pass each, no other "
                   "trace domain) over scripts/conv_bench.py 8192 --no-model, averages over the "
                   "launches of each kernel; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                   "per MI355X_MICROARCH.md HBM section (gfx950 FETCH_SIZE reports half of wide "
                   "coalesced reads); M=8192 images per launch",
           "kernels": {}}
    for name, (kern, alg) in ALG.items():
        k = next((v for kk, v in s.items() if kk.startswith(kern)), None)
        if k is None or "FETCH_SIZE" not in k:
            continue
        hbm = (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024
        e = {"FETCH_SIZE_KB": k["FETCH_SIZE"], "WRITE_SIZE_KB": k["WRITE_SIZE"],
             "hbm_bytes_corrected": int(hbm), "alg_bytes": alg,
             "traffic_over_alg": round(hbm / alg, 3)}
        for src, dst in (("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles"),
                         ("SQ_BUSY_CYCLES", "sq_busy_cycles"), ("GRBM_GUI_ACTIVE", "gui_active"),
                         ("SQ_WAVE_CYCLES", "wave_cycles_quad"), ("SQ_WAIT_ANY", "wait_any"),
                         ("SQ_WAIT_INST_ANY", "wait_inst_any"),
                         ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                         ("SQ_INSTS_VALU", "insts_valu"), ("SQ_INSTS_LDS", "insts_lds"),
                         ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict"),
                         ("SQ_LDS_IDX_ACTIVE", "lds_idx_active")):
            if src in k:
                e[dst] = k[src]
        out["kernels"][name] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
