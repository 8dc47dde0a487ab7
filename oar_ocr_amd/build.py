"""Builds libOarMi355x.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

No torch / pybind dependency: the library is plain C ABI (include/oar_mi355x.h) and is loaded with ctypes.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIBDIR = HERE / "lib"
LIB = LIBDIR / "libOarMi355x.so"
SOURCES = ["common.cc", "onnx_parse.cc", "engine.cc", "db_host.cc", "poly_host.cc", "pipeline.cc", "c_api.cc", "ctc_host.cc", "image_decode.cc", "image_misc_decode.cc", "jpeg_decode.cc", "layout.cc", "kernels.hip", "igemm.hip", "igemm_ws_1x1.hip", "igemm_ws_gen.hip", "igemm_ws_x6.hip", "igemm_os_x6.hip", "igemm_ws3.hip", "igemm_rs3_x6.hip", "prepost.hip", "contours.hip", "jpeg.hip", "layout.hip", "dsblock.hip", "dsblock_k3s1.hip", "dsblock_k3s2.hip", "dsblock_k5s1.hip", "dsblock_k5s2.hip", "dsblock_wa_a.hip", "dsblock_wa_b.hip", "dsblock_rs_k3s11.hip", "dsblock_rs_k3s21.hip", "dsblock_rs_k3s12.hip", "dsblock_rs_k3s22.hip", "dsblock_rs_dbg.hip", "dsblock_cs.hip", "dsblock_pc.hip", "dsblock_rs2.hip", "ctc_head_x6.hip", "chain.hip", "attention_x6.hip", "igemm_lk_x6.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wno-unused-result", "-Wno-pass-failed"]
if os.environ.get("OAR_DSB_ABLATIONS"):   # timing-ablation instantiations (wrong results) of the dsblock kernels: a build option, not product code
    FLAGS.append("-DOAR_DSB_ABLATIONS")


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + [HERE.parent / "include" / "oar_mi355x.h"]
    jobs = []
    for s in SOURCES:
        src = CSRC / s
        obj = objdir / (s.replace(".", "_") + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr[-4000:]}")
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [objdir / (s.replace(".", "_") + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs] + ["-lpthread", "-lz"]   # zlib: the PNG decoder's inflate / crc32 (image_decode.cc)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


def csrc_fingerprint() -> str:
    """sha256 over the kernel / host sources (csrc/*, sorted by name): recorded next to every committed counter summary
    (profiles/r*/pmc_traffic.json, mfma_util.json) so that bench.py can tell when the code it is timing is no longer the code that
    was profiled and refuse to echo stale counters."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(CSRC.iterdir()):
        if f.suffix in (".hip", ".inc", ".cc", ".h"):
            h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    p = build_lib(force="--force" in sys.argv, verbose=True)
    print(p, p.stat().st_size)
