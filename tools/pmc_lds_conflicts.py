"""Per-kernel LDS bank-conflict share of a step from ONE rocprofv3 --pmc pass:
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pl -o l --output-format csv -- python3 bench.py --eager ...
    python tools/pmc_lds_conflicts.py [gpurun_out/pl]
(the data gradient's border fragments were found this way: 0 conflict cycles in the forward, 4.5 % of the LDS-active cycles in the fold)"""
import collections, csv, glob, re, sys
base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pl"
g = glob.glob(f"{base}/**/*counter_collection.csv", recursive=True)
if not g:
    raise SystemExit(f"missing {base}")
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(g[0])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:80]
    d[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
        n[name] += 1
print(f"{'kernel':80s} {'calls':>6s} {'lds_active':>12s} {'conflict':>12s} {'share':>6s}")
for k in sorted(d, key=lambda k: -d[k]["SQ_LDS_BANK_CONFLICT"]):
    a, c = d[k]["SQ_LDS_IDX_ACTIVE"], d[k]["SQ_LDS_BANK_CONFLICT"]
    if a > 0:
        print(f"{k:80s} {n[k]:6d} {a:12.0f} {c:12.0f} {c / a:6.3f}")
