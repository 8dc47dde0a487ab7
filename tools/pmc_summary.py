"""Summarises rocprofv3 --pmc passes (gpurun_out/pmc_*/p_counter_collection.csv) per kernel name + grid size."""
import csv, sys, collections, glob
def load(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(float)
    for r in rows:
        k = (r["Kernel_Name"].split("(")[0][-60:], int(r["Grid_Size"]))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[(k, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    d2 = collections.defaultdict(float); n2 = collections.Counter()
    for (k, d), v in dur.items():
        d2[k] += v; n2[k] += 1
    return agg, d2, n2
if __name__ == "__main__":
    base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    f, fd, fn = load(f"{base}/pmc_FETCH_SIZE/p_counter_collection.csv")
    w, wd, wn = load(f"{base}/pmc_WRITE_SIZE/p_counter_collection.csv")
    s, sd, sn = load(f"{base}/pmc_SQ_WAVES/p_counter_collection.csv")
    keys = sorted(fd, key=lambda k: -fd[k])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]
    print(f"{'kernel':62s} {'grid':>9s} {'n':>3s} {'us':>8s} {'fetchMB*2':>9s} {'writeMB':>8s} {'GB/s':>7s} {'mfma%':>6s} {'wait%':>6s} {'instw%':>6s} {'act%':>5s}")
    for k in keys:
        n = fn[k]; us = fd[k] / n / 1e3
        fe = f[k]["FETCH_SIZE"] / n * 1024 * 2 / 1e6     # KB units; gfx950: FETCH_SIZE reads half of a wide coalesced stream
        wr = w[k]["WRITE_SIZE"] / max(wn[k], 1) * 1024 / 1e6
        sq = s[k]; wc = sq["SQ_WAVE_CYCLES"] or 1
        busy = sq["SQ_BUSY_CYCLES"] or 1
        print(f"{k[0]:62s} {k[1]:9d} {n:3d} {us:8.1f} {fe:9.1f} {wr:8.1f} {(fe+wr)/us*1e3/1e3:7.0f} {100*sq['SQ_VALU_MFMA_BUSY_CYCLES']/ (sd[k]/sn[k]*2.4*256*4/1e0 if False else 1) if False else 0:6.1f} {100*sq['SQ_WAIT_ANY']/wc:6.1f} {100*sq['SQ_WAIT_INST_ANY']/wc:6.1f} {100*sq['SQ_ACTIVE_INST_ANY']/wc:5.1f}")
