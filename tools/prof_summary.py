"""Turns rocprofv3 CSV output (kernel trace / --pmc passes) into the markdown summaries committed under profiles/.

    python tools/prof_summary.py stats  <kt_kernel_trace.csv>                      > profiles/rNN_rocprofv3_kernel_stats.md
    python tools/prof_summary.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_pmc_traffic.md

`stats` restricts itself to the step region (from the first attention dispatch on) so that the one-off weight generation
does not drown the hot path. `pmc` applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide
coalesced read): traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB.
"""
import csv
import collections
import sys


def short(name):
    return name if len(name) <= 110 else name[:107] + "..."


def stats(path, whole=False):
    rows = list(csv.DictReader(open(path)))
    if whole:    # VAE runs: no warm-up region to cut; drop torch's own fill / copy / RNG kernels (input generation)
        rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"] and "rocprim" not in r["Kernel_Name"]]
        t0 = min(int(r["Start_Timestamp"]) for r in rows)
    else:
        t0 = min(int(r["Start_Timestamp"]) for r in rows if "flash_attn" in r["Kernel_Name"])
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values())
    wall = max(int(r["End_Timestamp"]) for r in rows) - t0
    print(f"kernel time in the step region: {total / 1e6:.1f} ms over {wall / 1e6:.1f} ms of wall time ({len(rows)} dispatches)\n")
    print("| kernel | grid x wg | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|---|")
    for (name, g, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / total < 0.0005:
            continue
        print(f"| `{short(name)}` | {g} x {w} | {len(v)} | {sum(v) / 1e6:.2f} | {sum(v) / len(v) / 1e3:.1f} | {min(v) / 1e3:.1f} | "
              f"{max(v) / 1e3:.1f} | {100 * sum(v) / total:.2f} |")
    # the attention kernel serves self-attention (Lk = L) and cross-attention (Lk = 512) under one name: split by duration
    att = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "flash_attn" in r["Kernel_Name"]]
    big, small = [x for x in att if x > 1e6], [x for x in att if x <= 1e6]
    if big and small:
        print(f"\nattention dispatches split by use: self-attention {len(big)} launches, avg {sum(big) / len(big) / 1e3:.1f} us "
              f"(min {min(big) / 1e3:.1f}, max {max(big) / 1e3:.1f}); cross-attention {len(small)} launches, avg {sum(small) / len(small) / 1e3:.1f} us")


def pmc(fetch_path, write_path):
    def load(path, counter):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                d[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
        return d
    f, w = load(fetch_path, "FETCH_SIZE"), load(write_path, "WRITE_SIZE")
    print("| kernel | grid | dispatches | FETCH_SIZE KB (avg / max) | WRITE_SIZE KB (avg / max) | traffic per launch (max dispatch), MB |")
    print("|---|---|---|---|---|---|")
    keys = [k for k in f if "at::native" not in k[0] and "rocclr" not in k[0] and "rocprim" not in k[0]]
    for k in sorted(keys, key=lambda k: -(2 * max(f[k]) + max(w.get(k, [0])))):
        fv, wv = f[k], w.get(k, [0.0])
        tr = (2 * max(fv) + max(wv)) * 1024 / 1e6
        if tr < 1:
            continue
        print(f"| `{short(k[0])}` | {k[1]} | {len(fv)} | {sum(fv) / len(fv):.0f} / {max(fv):.0f} | {sum(wv) / len(wv):.0f} / {max(wv):.0f} | {tr:.0f} |")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "stats_all":
        stats(sys.argv[2], whole=True)
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3])


def clocks(path):
    """rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES per dispatch -> per-kernel effective
    clock (GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X_MICROARCH.md 'DVFS give-back') and matrix-pipe occupancy
    (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM cycles per XCD))."""
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = {}
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"], r["Grid_Size"], r["Dispatch_Id"])
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(list)
    for (name, grid, _), c in per.items():
        if "at::native" in name or "rocclr" in name or "rocprim" in name or dur[(name, grid, _)] < 20000:
            continue
        gui = sum(c.get("GRBM_GUI_ACTIVE", [0]))
        mfma = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
        ns = dur[(name, grid, _)]
        clk = gui / 8 / ns                       # GHz: cycles per XCD / ns
        busy = mfma / (gui / 8 * 256 * 4) if gui else 0.0
        agg[(name, grid)].append((ns, clk, busy))
    print("| kernel | grid | dispatches | avg us | effective clock GHz | matrix pipe busy | busy x clock / 2.4 GHz |")
    print("|---|---|---|---|---|---|---|")
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        n = len(v)
        us, clk, busy = sum(x[0] for x in v) / n / 1e3, sum(x[1] for x in v) / n, sum(x[2] for x in v) / n
        print(f"| `{short(name)}` | {grid} | {n} | {us:.1f} | {clk:.2f} | {100 * busy:.1f} % | {busy * clk / 2.4:.3f} |")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "clocks":
    clocks(sys.argv[2])
