"""Turn two rocprofv3 counter passes over `bench.py --kernels-only` into HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --kernels-only
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --kernels-only
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_traffic.json

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced
read at 64 bytes, so the read side is doubled (all kernels here read 8 or 16 bytes per lane,
coalesced); WRITE_SIZE is exact for wide streaming stores.  `--kernels-only` launches every label
exactly 3 times in a fixed order, which is how dispatches are matched to labels: by kernel name
(every label has its own instantiation), the two pointwise and rasteriser variants by order.
"""
import csv
import glob
import json
import os
import sys

CALLS = 3


def read_pass(directory, counter):
    rows = []
    for path in sorted(glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def per_label(rows):
    """{label: mean counter value per call}"""
    def pick(key):
        # rocprofv3 prints some names demangled ("k<32, 16, 2, 0, true>") and some mangled ("kILi32ELi16E...")
        base, _, targs = key.partition("<")
        forms = [key.replace(" ", "")]
        if targs:
            forms.append(base + "I" + "".join("Li%sE" % a for a in targs.rstrip(">").replace(" ", "").split(",") if a))
        return [v for _, name, v in rows if any(f in name.replace(" ", "") for f in forms)]
    out = {}

    def put(label, values, per_call=1):
        if values:
            assert len(values) == CALLS * per_call, (label, len(values))
            out[label] = sum(values) / CALLS

    put("bwd_fused[layer0 C=33]", pick("convlstm_bwd_fused_kernel<32, 2, 8"))
    put("bwd_fused[layer1 C=16]", pick("convlstm_bwd_fused_kernel<16, 1, 8"))
    put("cell_fwd[layer0 C=33]", pick("conv3x3_pp_kernel<32, 16, 2, 0"))
    put("cell_fwd[layer1 C=16]", pick("conv3x3_pp_kernel<16, 16, 2, 0"))
    pw = pick("lstm_pointwise_bwd_kernel")
    put("pointwise_bwd[layer0 C=33]", pw[:CALLS])
    put("pointwise_bwd[layer1 C=16]", pw[CALLS:])
    put("dgrad[layer0 C=33]", pick("conv3x3_pp_kernel<64, 0, 2, 1"))  # both row tiles in one launch (4-row pixel tiles)
    put("dgrad[layer1 C=16]", pick("conv3x3_pp_kernel<64, 0, 1, 1"))
    put("wgrad[layer0 C=33]", pick("conv_wgrad_pp_kernel<32, 2"))
    put("wgrad[layer1 C=16]", pick("conv_wgrad_pp_kernel<16, 1"))
    ra = pick("raster_iou_kernel")
    # building the synthetic batch rasterises too; the six launches of interest come last
    put("raster_iou[masks]", ra[-2 * CALLS:-CALLS])
    put("raster_iou[counts]", ra[-CALLS:])
    return out


def main():
    fetch = per_label(read_pass(sys.argv[1], "FETCH_SIZE"))
    write = per_label(read_pass(sys.argv[2], "WRITE_SIZE"))
    kernels = {}
    for label in fetch:
        rd, wr = 2.0 * fetch[label] * 1024.0, write.get(label, 0.0) * 1024.0
        kernels[label] = {"fetch_size_kib": fetch[label], "write_size_kib": write.get(label),
                          "read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench  # for the fingerprint of the kernel sources these numbers belong to
    json.dump({"csrc_hash": bench.csrc_hash(), "note": "per launch; read_bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction), "
                       "write_bytes = WRITE_SIZE x 1024; from bench.py --kernels-only at the default workload",
               "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
