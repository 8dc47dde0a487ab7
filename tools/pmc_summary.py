"""Summarises the rocprofv3 --pmc passes of one profiling session (tools/profile_round.sh) per kernel name + grid size:
HBM-side traffic (FETCH_SIZE x 2 -- the gfx950 correction of MI355X_MICROARCH.md -- and WRITE_SIZE, separate passes) and
the matrix pipe (pmc_MFMA pass): counter-based MFMA utilisation
    util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE)      (rocprofv3's MfmaUtil expression)
and the MFMA rate from SQ_INSTS_VALU_MFMA_MOPS_{BF16,F32} x 512 flop / kernel duration, as a fraction of the dense peak
(2500 TFLOP/s bf16, 157.3 TFLOP/s f32).  pmc_MFMA_peak (tools/mfma_peak under the same counters) calibrates both.
usage: python tools/pmc_summary.py [gpurun_out] [rows] [suffix]      suffix: "_c2" for the passes of a --config 2 session (tools/profile_round.sh <tag> 2)"""
import csv, sys, collections, os

CUS, SIMDS = 256, 4
PEAK_BF16, PEAK_F32 = 2500.0, 157.3


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, n = collections.defaultdict(float), collections.Counter()
    if not os.path.exists(path):
        return agg, dur, n
    seen = set()
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-64:], int(r["Grid_Size"]))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        d = (k, r["Dispatch_Id"])
        if d not in seen:
            seen.add(d)
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            n[k] += 1
    return agg, dur, n


def mfma_rows(base, sub):
    m, md, mn = load(f"{base}/{sub}/p_counter_collection.csv")
    rows = {}
    for k in md:
        c = m[k]
        gui = c["GRBM_GUI_ACTIVE"] or 1.0
        us = md[k] / mn[k] / 1e3
        tf_bf16 = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / mn[k] / us / 1e6
        tf_f32 = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / mn[k] / us / 1e6
        rows[k] = dict(n=mn[k], us=us, util=100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * CUS * gui), tf_bf16=tf_bf16, tf_f32=tf_f32,
                       frac_peak=100.0 * (tf_bf16 / PEAK_BF16 + tf_f32 / PEAK_F32), wait=100.0 * c["SQ_WAIT_INST_ANY"] / (c["SQ_WAVE_CYCLES"] or 1.0),
                       active=100.0 * c["SQ_ACTIVE_INST_ANY"] / (c["SQ_WAVE_CYCLES"] or 1.0))
    return rows


if __name__ == "__main__":
    base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    sfx = sys.argv[3] if len(sys.argv) > 3 else ""
    f, fd, fn = load(f"{base}/pmc_FETCH_SIZE{sfx}/p_counter_collection.csv")
    w, wd, wn = load(f"{base}/pmc_WRITE_SIZE{sfx}/p_counter_collection.csv")
    mf = mfma_rows(base, f"pmc_MFMA{sfx}")
    keys = sorted(fd, key=lambda k: -fd[k])[:top]
    print(f"{'kernel':66s} {'grid':>8s} {'n':>3s} {'us':>7s} {'fetchMB*2':>9s} {'writeMB':>8s} {'GB/s':>6s} | {'mfma busy%':>10s} {'bf16 TF':>8s} {'f32 TF':>7s} {'%peak':>6s} {'issue-stall%':>12s}")
    for k in keys:
        n = fn[k]; us = fd[k] / n / 1e3
        fe = f[k]["FETCH_SIZE"] / n * 1024 * 2 / 1e6
        wr = w[k]["WRITE_SIZE"] / max(wn[k], 1) * 1024 / 1e6
        m = mf.get(k)
        tail = f"{m['util']:10.1f} {m['tf_bf16']:8.1f} {m['tf_f32']:7.1f} {m['frac_peak']:6.1f} {m['wait']:12.1f}" if m else ""
        print(f"{k[0]:66s} {k[1]:8d} {n:3d} {us:7.1f} {fe:9.1f} {wr:8.1f} {(fe + wr) / us * 1e3:6.0f} | {tail}")
    cal = mfma_rows(base, "pmc_MFMA_peak")
    if cal:
        print("\ncalibration: tools/mfma_peak under the same counters (a kernel at the MFMA issue peak)")
        for k, m in sorted(cal.items(), key=lambda kv: -kv[1]["us"]):
            print(f"{k[0]:66s} {k[1]:8d} {m['n']:3d} {m['us']:7.1f}  mfma busy% {m['util']:6.1f}  bf16 {m['tf_bf16']:7.1f} TF  f32 {m['tf_f32']:6.1f} TF  %peak {m['frac_peak']:5.1f}")
