"""Fold three rocprofv3 --pmc passes of tools/run_kernels.py (SQ_* | FETCH_SIZE | WRITE_SIZE+GRBM_GUI_ACTIVE)
into profiles/<round>_pmc_traffic.json: per hand-written kernel, mean over its launches.
usage: python tools/pmc_summary.py pass1.csv pass2.csv pass3.csv out.json"""
import collections
import csv
import json
import re
import sys

KEYS = (("scan_op_fwd", "scan_short_fwd"), ("scan_op_bwd", "scan_short_bwd"), ("scan_op_fwd_wave", "scan_bdl_fwd"), ("scan_op_bwd_wave", "scan_bdl_bwd"),
        ("scan_op_reduce", "reduce_leading"),
        ("conv_pool_bwd_dgrad_addnorm_bwd", "conv_pool_bwd_dgrad_kernel"), ("conv_pool_fwd", "conv_pool_fwd"), ("scan_fwd", "scan_cl_fwd"), ("combine_out_proj_addnorm_fwd", "combine_out_proj_addnorm_kernel"),
        ("combine_fwd", "combine_fwd"),
        ("combine_bwd", "combine_bwd"), ("scan_bwd_xproj", "14, true, true>"), ("scan_bwd", "scan_cl_bwd"),
        ("conv_pool_bwd_two_addends", "conv_pool_bwd_row_kernel<__hip_bfloat16, 14, true>"), ("conv_pool_bwd", "conv_pool_bwd"),
        ("chunk_rows_bf16", "chunk_rows_bf16"),
        ("xproj_bwd", "xproj_bwd"), ("add_norm_fwd", "add_norm_fwd"), ("add_norm_bwd", "add_norm_bwd"),
        ("gemm_out_proj_addnorm_fwd", "gemm_addnorm_kernel"), ("gemm_in_proj_dgrad_addnorm_bwd", "gemm_dgrad_addnorm_bwd_kernel"),
        ("gemm", "gemm_bf16_kernel"))


def kernel_key(name):
    for key, pat in KEYS:
        if pat in name:
            if key == "gemm":
                m = re.search(r"gemm_bf16_kernel<([^>]*)>", name)
                return "gemm<" + (m.group(1).replace(" ", "") if m else "?") + ">"
            return key
    return None


def fold(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = kernel_key(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main(p1, p2, p3, out):
    a, b, c = fold(p1), fold(p2), fold(p3)
    res = {}
    for k in sorted(set(a) | set(b) | set(c)):
        e = {}
        f, w = b.get(k, {}).get("FETCH_SIZE"), c.get(k, {}).get("WRITE_SIZE")
        if f is not None:
            e["FETCH_SIZE_KiB"] = round(f, 1)
        if w is not None:
            e["WRITE_SIZE_KiB"] = round(w, 1)
        if f is not None and w is not None:
            e["traffic_bytes"] = int((2 * f + w) * 1024)        # gfx950: FETCH_SIZE counts 64 B per 128-B request
        s = a.get(k, {})
        if s.get("SQ_WAVES"):
            wv = s["SQ_WAVES"]
            e["SQ_WAVES"] = int(wv)
            e["SQ_INSTS_VALU_per_wave"] = round(s.get("SQ_INSTS_VALU", 0) / wv, 1)
            e["SQ_INSTS_SALU_per_wave"] = round(s.get("SQ_INSTS_SALU", 0) / wv, 1)
            if s.get("SQ_WAVE_CYCLES"):
                e["wave_cycles_per_wave"] = int(s["SQ_WAVE_CYCLES"] / wv)
                e["SQ_WAIT_ANY_frac"] = round(s.get("SQ_WAIT_ANY", 0) / s["SQ_WAVE_CYCLES"], 3)
                e["SQ_ACTIVE_INST_VALU_frac"] = round(s.get("SQ_ACTIVE_INST_VALU", 0) / s["SQ_WAVE_CYCLES"], 3)
        res[k] = e
    note = ("rocprofv3 --pmc passes (separate runs: SQ_*, FETCH_SIZE, WRITE_SIZE+GRBM_GUI_ACTIVE) of tools/run_kernels.py "
            "at the FastVim-T bs=128 shape, mean over the launches of each kernel. FETCH_SIZE/WRITE_SIZE are in KiB; on "
            "gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): "
            "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. The row walkers read 4-12 B per lane, a width the guide "
            "marks uncalibrated, so treat traffic as an upper estimate.")
    json.dump({"_note": note, "kernels": res}, open(out, "w"), indent=1)
    for k, e in res.items():
        print(k, e.get("traffic_bytes"), e.get("SQ_INSTS_VALU_per_wave"), e.get("SQ_WAIT_ANY_frac"))


if __name__ == "__main__":
    main(*sys.argv[1:5])
