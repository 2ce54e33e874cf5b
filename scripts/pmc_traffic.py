"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into profiles/rNN_traffic.json.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv|-> <algorithmic bytes per launch> <out.json> [note]
Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is reported in KiB and counts
128-byte requests at 64 B -> bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE (KiB) is taken 1:1 (calibrated in round 1 on
k_fill, which writes exactly its buffer)."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    for key, tag in (("k_sha256_pair<pbsk::RecordSource>", "k_sha256_pair<RecordSource>"), ("k_scan3<34, 4>", "k_scan3<34,4>"),
                     ("k_sha256_pair<pbsk::DescSource>", "k_sha256_pair<DescSource>"), ("k_xxh3", "k_xxh3"),
                     ("k_resolve", "k_resolve"), ("k_compact", "k_compact")):
        if key in name:
            return tag
    return None


def main():
    fetch, write, alg, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    fk = per_kernel(fetch, "FETCH_SIZE")
    wk = per_kernel(write, "WRITE_SIZE") if write != "-" else {}
    kernels = {}
    for name, vals in fk.items():
        tag = short(name)
        if not tag:
            continue
        mean = sum(vals) / len(vals)
        k = kernels.setdefault(tag, {})
        k.update({"launches": len(vals), "fetch_size_kib_per_launch": round(mean, 1), "hbm_read_bytes_per_launch": int(mean * 2048),
                  "read_ratio_vs_algorithmic": round(mean * 2048 / alg, 3)})
    for name, vals in wk.items():
        tag = short(name)
        if tag and tag in kernels:
            kernels[tag]["write_size_kib_per_launch"] = round(sum(vals) / len(vals), 1)
    json.dump({"source": note, "corrections": "bytes = FETCH_SIZE[KiB] * 1024 * 2 (gfx950 counts 128-byte requests at 64 B); "
               "WRITE_SIZE[KiB] 1:1", "algorithmic_bytes_per_launch": int(alg), "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
