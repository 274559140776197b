"""Summaries of rocprofv3 output directories (rocpd .db files, the default output format of ROCm 7).
  python tools/pmc_summary.py pmc OUT.json NAME=DIR [NAME=DIR ...]   one --pmc pass per DIR, NAME = counter
  python tools/pmc_summary.py stats OUT.csv DIR                      --kernel-trace --stats run
"""
import csv, glob, json, os, sqlite3, sys

def short(name):
    return name.split("(")[0].split(" ")[-1]

def dbs(d):
    return glob.glob(os.path.join(d, "**", "*.db"), recursive=True)

def pmc(out_path, args):
    out = {}
    for arg in args:
        cname, d = arg.split("=", 1)
        per = {}
        for path in dbs(d):
            c = sqlite3.connect(path)
            q = "select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name=? group by kernel_name"
            for name, n, total in c.execute(q, (cname,)):
                k = short(name)
                if k.startswith("mrz_"):
                    e = per.setdefault(k, {"dispatches": 0, "sum": 0.0})
                    e["dispatches"] += n
                    e["sum"] += total
        out[cname] = per
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))

def stats(out_path, d):
    rows = []
    for path in dbs(d):
        c = sqlite3.connect(path)
        q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
             "group by name order by sum(duration) desc")
        rows += list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / total, 4), r[4], r[5]])
    for r in rows[:8]:
        print(short(r[0]), r[1], r[2], round(r[3], 1))

if __name__ == "__main__":
    if sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3:])
    else:
        stats(sys.argv[2], sys.argv[3])
