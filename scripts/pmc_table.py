"""Per-kernel-family roofline table from rocprofv3 counter dumps of ONE step of a bench command:
    python scripts/pmc_table.py <dir with the passes' *counter_collection.csv files> [kernel_stats.csv]
Passes expected (separate runs, as MI355X_MICROARCH.md prescribes): FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE.
HBM-side bytes: FETCH_SIZE x 2 (gfx950 wide-read correction) + WRITE_SIZE, units of 1024 B (FETCH counts L2 misses: Infinity Cache
hits included).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), both summed over the family's dispatches
(rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs)."""
import collections, csv, os, sys

def family(name):
    k = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("rvb::", "")
    return k.split("(")[0].split("<")[0][:40]

tot = collections.defaultdict(collections.Counter)
n = collections.Counter()
for d, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(d, f))):
                k = family(r["Kernel_Name"])
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "FETCH_SIZE":
                    n[k] += 1
dur = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        k = family(r["Name"])
        dur[k] = dur.get(k, 0.0) + float(r["TotalDurationNs"]) / max(int(r["Calls"]), 1) * 1.0      # per-call average, summed per family below
calls = collections.Counter()
avg = collections.Counter()
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        k = family(r["Name"]); calls[k] += int(r["Calls"]); avg[k] += float(r["TotalDurationNs"])
print(f"{'kernel family':40s} {'launches':>8s} {'read GB':>9s} {'write GB':>9s} {'ms (trace avg x launches)':>26s} {'TB/s':>6s} {'MFMA busy':>9s}")
rows = []
for k, c in tot.items():
    rd, wr = 2.0 * c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9
    ms = avg[k] / calls[k] * n[k] / 1e6 if calls[k] else float("nan")
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if c["GRBM_GUI_ACTIVE"] else float("nan")
    rows.append((rd + wr, k, n[k], rd, wr, ms, (rd + wr) / ms if ms == ms and ms > 0 else float("nan"), busy))
for _, k, nn, rd, wr, ms, tbs, busy in sorted(rows, reverse=True)[:24]:
    print(f"{k:40s} {nn:8d} {rd:9.2f} {wr:9.2f} {ms:26.2f} {tbs:6.2f} {busy:9.1%}")
