"""HBM traffic per launch of every conv kernel of the inner step, from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in separate runs: the TCC block cannot hold both, MI355X_MICROARCH.md "HBM").  Writes profiles/pmc_traffic.json,
which bench.py reads for roofline.traffic (labelled with this file, the round tag and the commit in "traffic_source").

usage (repo root, after tools/measure_round.sh on the GPU box merged gpurun_out/pmc_{FETCH,WRITE}_SIZE back):
    python tools/pmc_traffic.py <round-tag> <git-sha>
Counter unit KiB.  gfx950 corrections as the guide prescribes: FETCH_SIZE counts 128-byte requests of wide coalesced streams at
64 bytes -> doubled; WRITE_SIZE is exact."""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

SLOTS = {   # bench.py roofline slot -> substring of the kernel name
    "conv2_fwd": "conv3x3_resw_kernel<16, 16",
    "conv3_fwd": "conv3x3_stream_kernel<64, 128, 16, 8, 0, false>",
    "conv4_fwd": "conv3x3_stream_kernel<128, 128, 32, 8, 0, false>",
    "conv2_dgrad": "conv3x3_resw_w1x_kernel",
    "conv3_dgrad": "conv3x3_stream_kernel<128, 64, 32, 8, 0, false>",
    "conv4_dgrad": "conv3x3_stream_kernel<128, 128, 32, 8, 2, true>",
    "conv2_wgrad": "conv3x3_wgrad2_kernel<64, 64,",
    "conv3_wgrad": "conv3x3_wgrad2_kernel<64, 128,",
    "conv4_wgrad": "conv3x3_wgrad2_kernel<128, 128,",
    "wgrad_enc": "gemm_wgrad_grouped16_kernel",
}


def per_launch(counter):
    import os
    files = sorted(glob.glob(f"gpurun_out/pmc_{counter}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime, reverse=True)
    assert files, f"no counter_collection.csv under gpurun_out/pmc_{counter}"
    acc = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for slot, pat in SLOTS.items():
        vals = [v for k, vs in acc.items() if pat in k for v in vs]
        if vals:
            out[slot] = sum(vals) / len(vals)                # KiB per launch (mean over the traced steps)
    return out, files[0]


def main():
    tag, sha = sys.argv[1], sys.argv[2]
    fetch, f0 = per_launch("FETCH_SIZE")
    write, f1 = per_launch("WRITE_SIZE")
    kernels = {k: int((2.0 * fetch[k] + write.get(k, 0.0)) * 1024) for k in fetch}
    js = {"_comment": "HBM bytes per launch = 2 x FETCH_SIZE (gfx950 correction for wide coalesced reads) + WRITE_SIZE, KiB -> bytes; "
                      "means over the dispatches of the traced steps",
          "workload": {"batch": 16, "frames": 1000, "idim": 80},
          "collected": f"{tag} @ {sha}",
          "command": "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e --tasks-per-gpu 1",
          "kernels": kernels,
          "raw_KiB": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}}
    Path("profiles/pmc_traffic.json").write_text(json.dumps(js, indent=1) + "\n")
    for c, f in (("fetch_size", f0), ("write_size", f1)):
        rows = [r for r in csv.DictReader(open(f)) if "conv" in r["Kernel_Name"]]
        with open(f"profiles/{tag}_pmc_{c}.csv", "w") as out:
            w = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
