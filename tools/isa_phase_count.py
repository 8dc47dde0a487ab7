#!/usr/bin/env python3
"""Counts the instructions of a line range of a `hipcc -S` listing by issue class (tools/isa_phase_count.py file.s first last [label]).
MFMA / VALU (packed f32 ones apart) / LDS (reads, writes) / VMEM (loads incl. LDS-DMA, stores) / SALU / waits / branches."""
import re, sys

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_fma_f32") or op.startswith("v_pk_mul_f32") or op.startswith("v_pk_add_f32"): return "valu_pk_f32"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_read"
    if op.startswith("ds_"): return "lds_write"
    if op.startswith("buffer_load") or op.startswith("global_load"): return "vmem_load"
    if op.startswith("buffer_store") or op.startswith("global_store"): return "vmem_store"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_nop") or op.startswith("s_sleep") or op.startswith("s_setprio") or op.startswith("s_barrier"): return "s_misc"
    if op.startswith("s_"): return "salu"
    return None

def count(lines):
    c = {}
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        m = re.match(r"([a-z_0-9]+)", t)
        if not m: continue
        k = classify(m.group(1))
        if k: c[k] = c.get(k, 0) + 1
    return c

if __name__ == "__main__":
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lab = sys.argv[4] if len(sys.argv) > 4 else f"{a}-{b}"
    c = count(open(f).read().splitlines()[a - 1:b])
    print(lab, " ".join(f"{k}={v}" for k, v in sorted(c.items())), "total=%d" % sum(c.values()))
