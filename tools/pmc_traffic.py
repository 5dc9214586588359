#!/usr/bin/env python3
"""HBM-side traffic per launch and kernel from two rocprofv3 PMC passes of the bench command (FETCH_SIZE and WRITE_SIZE
cannot share a pass on gfx950).  Units and corrections as MI355X_MICROARCH.md 'HBM' prescribes: both counters are in
KiB per dispatch; FETCH_SIZE counts 128-byte fabric requests at 64 bytes on gfx950 -> doubled; WRITE_SIZE is exact.
usage: pmc_traffic.py <fetch run_results.db> <write run_results.db> <config name> <out.json>"""
import collections, json, re, sqlite3, sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for name, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if cname != counter:
            continue
        k = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*$", "", k).strip()  # drop the argument list: what remains is the tag bench.py reports
        a = agg[k]
        a[0] += float(val)
        a[1] += 1
    return agg


def main():
    fdb, wdb, cfg, out = sys.argv[1:5]
    steps_in_pass = int(sys.argv[5]) if len(sys.argv) > 5 else 5  # (tools/profile_round.sh: --steps 3 --warmup 2, no replay)
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    import importlib, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = {"config": cfg, "steps_in_pass": steps_in_pass, "source_hash": importlib.import_module("__graft_entry__").source_hash(), "source": f"profiles/{out.split('/')[-1].replace('_traffic.json', '')}_fetch_size.txt, ..._write_size.txt "
           "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; KiB per dispatch; FETCH_SIZE x2 on gfx950)",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fb = 2.0 * 1024.0 * f[k][0] / max(f[k][1], 1) if k in f else 0.0
        wb = 1024.0 * w[k][0] / max(w[k][1], 1) if k in w else 0.0
        res["kernels"][k] = {"launches_fetch_pass": f[k][1] if k in f else 0, "launches_write_pass": w[k][1] if k in w else 0,
                             "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_fetch_pass"])[:12]:
        print(f"{k[:70]:70s} launches {v['launches_fetch_pass']:6d} fetch {v['fetch_bytes_per_launch'] / 1e6:8.2f} MB write "
              f"{v['write_bytes_per_launch'] / 1e6:8.2f} MB per launch")


if __name__ == "__main__":
    main()
