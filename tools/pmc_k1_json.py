"""Combine the three PMC passes of tools/pmc_k1.sh into profiles/<round>_k1_<mode>_pmc.json (bench.py reads the fwd one):
    python tools/pmc_k1_json.py fwd|dgrad|wgrad out.json"""
import collections, csv, glob, json, os, sys
MODE = sys.argv[1] if len(sys.argv) > 1 else "fwd"
KEYS = {"fwd": "igemm_pipe_patch_kernel<true, false, true>", "dgrad": "igemm_pipe_patch_kernel<true, true, true>",
        "wgrad": "wgrad_pipe_kernel<true"}
KEY = KEYS[MODE]
DESC = {"fwd": "igemm_pipe_patch_kernel<bf16, nofold, taps9> forward 3x3 s1 256->256 on 64x64, N=16 (77.3 GFLOP per launch)",
        "dgrad": "igemm_pipe_patch_kernel<bf16, fold, taps9> data gradient of the same layer in one launch (reflection fold inside "
                 "the pixel operand), N=16 (77.3 GFLOP per launch)",
        "wgrad": "wgrad_pipe_kernel<bf16> weight gradient of the same layer (9 tiles x 28 pixel splits; the slab sum is a separate "
                 "kernel), N=16 (77.3 GFLOP per launch)"}


def find(pat):
    g = glob.glob(pat, recursive=True)
    if not g:
        raise SystemExit(f"missing {pat}")
    return g[0]


def agg(n):
    rows = list(csv.DictReader(open(find(f"gpurun_out/pmc_{MODE}_{n}/**/*counter_collection.csv"))))
    d = collections.defaultdict(list)
    for r in rows:
        if KEY in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}, max((len(v) for v in d.values()), default=0)


f, nl = agg("f")
w, _ = agg("w")
s, _ = agg("s")
kt = list(csv.DictReader(open(find(f"gpurun_out/pmc_{MODE}_s/**/*kernel_trace.csv"))))
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt if KEY in r["Kernel_Name"]]
waves = (256 if MODE != "wgrad" else 252) * 8
fetch = f["FETCH_SIZE"] * 1024 * 2      # KB units; gfx950: 16-B/lane streaming reads are counted at half their bytes
write = w["WRITE_SIZE"] * 1024
wave_cycles = s["SQ_WAVE_CYCLES"] * 4 / waves   # quad-cycles -> cycles, per wave
out = {
    "kernel": DESC[MODE],
    "git_commit": os.environ.get("MT_GIT_COMMIT", "unknown"),
    "command": f"MT_GIT_COMMIT=<hash> bash tools/pmc_k1.sh  (rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/bench_k1.py {MODE}; "
               "separate passes for FETCH_SIZE, WRITE_SIZE, SQ_*)",
    "launches_averaged": nl,
    "avg_duration_us_under_pmc": sum(durs) / len(durs) / 1e3,
    "FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"],
    "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
    "traffic_bytes_per_launch": fetch + write,
    # fwd / dgrad: read one activation, write one, read the bf16 weights; wgrad: read two activations, write 28 fp32 slabs
    "algorithmic_bytes_per_launch": (16 * 64 * 64 * 256 * 2 * 2 + 256 * 256 * 9 * 2) if MODE != "wgrad" else
                                    (16 * 64 * 64 * 256 * 2 * 2 + 28 * 256 * 256 * 9 * 4),
    "note_fetch": "gfx950 FETCH_SIZE counts 16-B/lane streaming reads at half their bytes (MI355X_MICROARCH.md, HBM section) -> "
                  "doubled; Infinity-Cache hits are included in this fabric-side counter, so the 9-tap re-reads that miss the "
                  "XCD L2 show up here although x (33.5 MB) stays resident in the 256 MiB Infinity Cache",
    "SQ": s,
    "mfma_busy_fraction": s["SQ_VALU_MFMA_BUSY_CYCLES"] / (s["SQ_BUSY_CYCLES"] * 4 * 4 / 8) if False else None,
    "wave_cycles_per_wave": wave_cycles,
    "wait_any_fraction": s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"],
    "issue_stall_fraction": s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"],
    "active_fraction": s["SQ_ACTIVE_INST_ANY"] / s["SQ_WAVE_CYCLES"],
    "lds_bank_conflict_cycles": s["SQ_LDS_BANK_CONFLICT"],
}
# MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; the kernel occupies 1024 SIMDs for
# wave_cycles_per_wave cycles (two waves per SIMD run concurrently, so the SIMD's time is one wave's lifetime)
out["mfma_busy_fraction"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * wave_cycles)
name = sys.argv[2] if len(sys.argv) > 2 else f"profiles/round3_k1_{MODE}_pmc.json"
json.dump(out, open(name, "w"), indent=1)
print(json.dumps(out, indent=1))
