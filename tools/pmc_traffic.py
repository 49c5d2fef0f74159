"""Per-kernel fabric traffic (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, MI355X_MICROARCH.md) from two rocprofv3 --pmc
passes written by:   rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d gpurun_out/pt_f -o f --output-format csv -- python3 <bench>
                     rocprofv3 --kernel-trace --pmc WRITE_SIZE  -d gpurun_out/pt_w -o w --output-format csv -- python3 <bench>
    python tools/pmc_traffic.py [gpurun_out/pt]      -> MB fetched / written per launch, per kernel (and grid size)
"""
import collections, csv, glob, re, sys
base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pt"


def agg(n, counter):
    g = glob.glob(f"{base}_{n}/**/*counter_collection.csv", recursive=True)
    if not g:
        raise SystemExit(f"missing {base}_{n}")
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(g[0])):
        if r["Counter_Name"] == counter:
            name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
            d[(name, r.get("Grid_Size", r.get("Grid_Size_X", "")))].append(float(r["Counter_Value"]))
    return d


f, w = agg("f", "FETCH_SIZE"), agg("w", "WRITE_SIZE")
print(f"{'kernel':70s} {'grid':>9s} {'calls':>6s} {'fetch_MB':>9s} {'write_MB':>9s}")
for k in sorted(f, key=lambda k: -sum(f[k])):
    fv, wv = f[k], w.get(k, [0.0])
    print(f"{k[0]:70s} {k[1]:>9s} {len(fv):6d} {sum(fv) / len(fv) * 2 * 1024 / 1e6:9.1f} {sum(wv) / len(wv) * 1024 / 1e6:9.1f}")
