"""Builds a profiles/r0N_pmc_*ffn1.json from three rocprofv3 --pmc passes over `tools/one_gemm.py ffn1`:
python tools/pmc_ffn_collect.py <fetch_dir> <write_dir> <sq_dir> [out.json] [kernel-name substring]
(round 2: the fp32 MFMA kernel, gemm_f32_fast_kernel; round 3: gemm_x3w_kernel, the bf16-pieces kernel)."""
import collections, csv, glob, json, sys

KERNEL = sys.argv[5] if len(sys.argv) > 5 else "gemm_f32_fast_kernel"


def mean_of(directory, counter):
    v = []
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and KERNEL in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    return sum(v[1:]) / max(len(v) - 1, 1)


fetch, write = mean_of(sys.argv[1], "FETCH_SIZE"), mean_of(sys.argv[2], "WRITE_SIZE")
gui, mfma = mean_of(sys.argv[3], "GRBM_GUI_ACTIVE"), mean_of(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES")
T, L, I = 279 * 128, 512, 2048
x3w = "x3w" in KERNEL
doc = {
    "kernel": KERNEL + " (FFN linear1: [35712,512] x [2048,512]^T + bias + ReLU, S = 279, B = 128)",
    "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 tools/one_gemm.py ffn1   (one pass per counter "
               "group: FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; 5 launches, first dropped)",
    "corrections": "MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane reads "
                   "-> doubled; WRITE_SIZE taken as is",
    "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
    "fetch_bytes": int(2 * fetch * 1024), "write_bytes": int(write * 1024),
    "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
    "algorithmic_bytes_per_launch": (4 * T * L + 6 * I * L + 4 * I + 4 * T * I) if x3w else 4 * (T * L + I * L + I + T * I),
    "mfma_busy_fraction": round((mfma / 1024.0) / (gui / 8.0), 3) if gui > 0 else None,
    "effective_clock_note": "GRBM_GUI_ACTIVE / 8 = shader cycles of the launch (sum over the 8 XCDs)",
    "GRBM_GUI_ACTIVE": round(gui, 1), "SQ_VALU_MFMA_BUSY_CYCLES": round(mfma, 1),
}
json.dump(doc, open(sys.argv[4] if len(sys.argv) > 4 else "profiles/r02_pmc_ffn1.json", "w"), indent=1)
print(doc["traffic_bytes_per_launch"], doc["algorithmic_bytes_per_launch"], doc["mfma_busy_fraction"])
