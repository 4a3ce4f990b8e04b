#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from the rocprofv3 --pmc passes of tools/prof_bench.sh / tools/prof_solvers.sh.
usage: python tools/pmc_to_json.py gpurun_out/<name> profiles/<out>.json [--commit SHA] [--build-id ID]
`build_id` = hipk_build_id() of the library the counters were taken on (default: the library in the tree, which is the one the
profiled command loaded when this runs in the same gpurun call); bench.py quotes `traffic` only when it equals its own.
Launches that returned at once (iterations past the stop word, unwanted second CGS passes) are dropped: only launches whose
counter is at least half of the kernel's maximum enter the mean -- except for the GMRES kernels, whose traffic grows with the
Arnoldi step (every launch above 1 % of the maximum counts there; the mean is then the mean over a restart cycle).
traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; until call 29 of round 2 this script multiplied by 1000 and read 2.3 % low): gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact.  `commit` = the code the counters were taken on (bench.py prints it as
`traffic_from_commit` and drops the traffic when the kernel that ran is another one)."""
import collections, csv, glob, json, os, subprocess, sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
src, dst = args[0], args[1]
commit = None
if "--commit" in sys.argv:
    commit = sys.argv[sys.argv.index("--commit") + 1]
else:
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        pass
build_id = None
if "--build-id" in sys.argv:
    build_id = sys.argv[sys.argv.index("--build-id") + 1]
else:
    try:
        import ctypes
        _L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd/pytorch_sparse_solver/_lib/libhipk.so"))
        _L.hipk_build_id.restype = ctypes.c_char_p
        build_id = _L.hipk_build_id().decode()
    except Exception:
        pass
N = 4_000_000
NB = 64_000_000
MB = 1e6
# key -> (kernel-name pattern, algorithmic bytes per launch at N = 4 M fp64 (None: varies), what they are, cycle-mean kernels)
KEYS = {
    "spmv": ("hipk_spmv_sell_wide_kernel<5, 1, 0>", None, "coded SpMV: bytes its format streams (hipk_csr_format_bytes)", False),
    "spmv_plain": ("hipk_spmv_kernel<double", 319_904_004, "SURVEY 8d: nnz*12 + (n+1)*4 + 2n*8", False),
    "cg_update": ("hipk_cg_update_kernel<double, false, false>", 24 * N, "read Ap, r; write r", False),
    "cg_direction": ("hipk_cg_direction_kernel<double, false, false, false>", 40 * N, "read r, p, x; write p, x", False),
    "gm_multidot": ("hipk_gm_multidot_stream_kernel<double", int(17.5 * 8 * N), "mean over k = 0..29 of 8n(k+2): w + k+1 columns", True),
    "gm_update": ("hipk_gm_update_stream_kernel<double", int(18.5 * 8 * N), "mean over k = 0..29 of 8n(k+3): w in/out + k+1 columns", True),
    "gm_normalize": ("hipk_gm_normalize_kernel<double", 16 * N, "read w, write v", False),
    "gm_xupdate": ("hipk_gm_xupdate_kernel<double", None, "8n(k+2): x in/out + k columns", True),
    "bi_direction": ("hipk_bi_direction_kernel<double", 32 * N, "read r, p, q; write p", False),
    "bi_supdate": ("hipk_bi_supdate_kernel<double", 24 * N, "read r, q; write s", False),
    "bi_xupdate": ("hipk_bi_xupdate_kernel<double", 56 * N, "read x, p, s, t, rhat; write x, r", False),
}
# the same CG kernels on the N = 64 M system of bench.py's HBM-resident leg (other instantiations: streaming policy, grouped walk)
KEYS_N64M = {
    "spmv": ("hipk_spmv_sell_wide_kernel<5, 1, 1>", None, "coded SpMV, grouped walk: bytes its format streams", False),
    "cg_update": ("hipk_cg_update_kernel<double, false, true>", 24 * NB, "read Ap, r; write r", False),
    "cg_direction": ("hipk_cg_direction_flat_kernel<double>", 40 * NB, "read r, p, x; write p, x", False),
}


def means(counter_dir, counter, KEYS=KEYS):
    out = {}
    for f in glob.glob(os.path.join(src, "**", counter_dir, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for name, v in acc.items():
            for key, (pat, _, _, cyc) in KEYS.items():
                if pat in name:
                    top = max(v)
                    real = [a for a in v if a >= (0.01 if cyc else 0.5) * top]
                    if key in out and out[key]["launches"] >= len(real):
                        continue  # several instantiations match: the one launched most often (the solver loop's) stands for the key
                    out[key] = {"kernel": name.split("(")[0].replace("void ", ""), "mean_KB": sum(real) / len(real),
                                "launches": len(real), "dropped_noop_launches": len(v) - len(real)}
    return out


def section(KEYS):
  fetch, write = means("pmc_fetch", "FETCH_SIZE", KEYS), means("pmc_write", "WRITE_SIZE", KEYS)
  kern = {}
  for key in fetch:
    w = write.get(key, {"mean_KB": 0.0})
    _, alg, what, _ = KEYS[key]
    kern[key] = {"kernel": fetch[key]["kernel"], "FETCH_SIZE_KB_mean": fetch[key]["mean_KB"],
                 "WRITE_SIZE_KB_mean": w["mean_KB"], "launches": fetch[key]["launches"],
                 "dropped_noop_launches": fetch[key]["dropped_noop_launches"],
                 "traffic_bytes_per_launch": int(round((2.0 * fetch[key]["mean_KB"] + w["mean_KB"]) * 1024.0)),
                 "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_are": what}
  return kern


kern, kern_big = section(KEYS), section(KEYS_N64M)
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes ({src})", "commit": commit, "build_id": build_id,
           "fetch_correction": 2.0,
           "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE; counters are KiB (1024 B: WRITE_SIZE of a 32,000,000-byte vector reads 31250.0); "
                   "FETCH_SIZE includes Infinity-Cache hits",
           "kernels": kern, "kernels_n64m": kern_big}, open(dst, "w"), indent=1)
print(json.dumps(kern, indent=1))
