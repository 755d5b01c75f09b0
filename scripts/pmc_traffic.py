"""Sum the rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) of the
trace kernel into HBM bytes per launch, with the gfx950 corrections from
MI355X_MICROARCH.md (HBM section): both counters are in KiB; FETCH_SIZE
under-reports a wide coalesced streaming read by 2x on this rocprofv3.
Writes <out>/traffic.json; copy it to profiles/traffic.json to have bench.py
report it."""
import csv
import glob
import json
import os
import sys


def kernel_counter(outdir, counter):
    vals = []
    for path in glob.glob(os.path.join(outdir, "prof_pmc_" + counter, "**",
                                       "*counter_collection.csv"),
                          recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "rt_trace_kernel" in row.get("Kernel_Name", "") and \
                        row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
    return vals


def main():
    out = sys.argv[1]
    fetch = kernel_counter(out, "FETCH_SIZE")
    write = kernel_counter(out, "WRITE_SIZE")
    if not fetch or not write:
        print("pmc_traffic: no counter rows found", len(fetch), len(write))
        return
    f_kib = sum(fetch)/len(fetch)
    w_kib = sum(write)/len(write)
    res = {
        "kernel": "rt_trace_kernel",
        "launches": [len(fetch), len(write)],
        "FETCH_SIZE_KiB_raw": f_kib,
        "WRITE_SIZE_KiB_raw": w_kib,
        "fetch_bytes_corrected_x2": f_kib*1024*2,
        "write_bytes": w_kib*1024,
        "hbm_bytes_per_launch": f_kib*1024*2 + w_kib*1024,
        "rays": 10_000_000, "clip": True, "alias_i": 1,
        "profile": "profiles/%s" % (sys.argv[2] if len(sys.argv) > 2
                                    else os.path.basename(out.rstrip("/"))),
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                "rocprofv3 tallies 128-B requests at 64 B); WRITE_SIZE "
                "uncalibrated by the guide, taken at face value",
    }
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
