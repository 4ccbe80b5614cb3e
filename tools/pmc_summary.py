"""Summarise rocprofv3 --pmc passes into the profiles/*pmc_traffic.json layout bench.py reads.

    python tools/pmc_summary.py WORKLOAD FETCH_DIR WRITE_DIR TCC_DIR [BASE_JSON] > profiles/rNN_pmc_traffic.json

Each DIR is the -d directory of one pass (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc TCC_HIT_sum TCC_MISS_sum`).
hbm_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of MI355X_MICROARCH.md's HBM section.
BASE_JSON: an earlier summary whose other workloads are carried over unchanged.
"""
import collections, csv, glob, json, re, sys


def per_kernel(directory):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    workload, fetch, write, tcc = sys.argv[1:5]
    base = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else {"workloads": {}}
    f, w, t = per_kernel(fetch), per_kernel(write), per_kernel(tcc)
    out = {}
    for k in sorted(f):
        if not k.startswith(("ftm_", "ftb_", "ft_", "gram_", "sgd_apply", "l1_", "tail_train", "conv_binarize", "ste_conv")):
            continue
        fs, ws = f[k].get("FETCH_SIZE", 0.0), w.get(k, {}).get("WRITE_SIZE", 0.0)
        hit, miss = t.get(k, {}).get("TCC_HIT_sum", 0.0), t.get(k, {}).get("TCC_MISS_sum", 0.0)
        out[k] = {"FETCH_SIZE_KiB": round(fs, 1), "WRITE_SIZE_KiB": round(ws, 1), "hbm_bytes_corrected": int((2 * fs + ws) * 1024),
                  "TCC_HIT_sum": int(hit), "TCC_MISS_sum": int(miss), "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
    base.setdefault("workloads", {})[workload] = out
    base["_about"] = ("rocprofv3 --pmc passes (one counter group per run: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum; no trace "
                      "domains) of the shipped training step launched eagerly (tools/probe_step.py <workload>, the kernels the bench "
                      "line's graph replays). FETCH_SIZE/WRITE_SIZE in KiB as reported; per MI355X_MICROARCH.md "
                      "(HBM section) FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950: hbm_bytes_corrected = (2*FETCH + "
                      "WRITE)*1024. Averages over the launches of the run; summarised by tools/pmc_summary.py.")
    json.dump(base, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
