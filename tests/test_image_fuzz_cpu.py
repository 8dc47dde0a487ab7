"""The host image decoders parse untrusted bytes (oar_image_decode): a short mutation-fuzz batch under AddressSanitizer +
UndefinedBehaviorSanitizer must end with every input either decoded or rejected with an oar::Error -- no sanitizer report, no
bad_alloc, no output whose size disagrees with its header.  tools/fuzz/fuzz_image_decoders.cc is the driver (run it longer by hand:
millions of iterations were clean when this test was written)."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "oar_ocr_amd" / "csrc"


def test_mutated_images_are_decoded_or_rejected(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "fuzz"
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           f"-I{CSRC}", f"-I{ROOT / 'include'}", str(ROOT / "tools" / "fuzz" / "fuzz_image_decoders.cc"), str(CSRC / "image_decode.cc"), str(CSRC / "jpeg_decode.cc"),
           str(CSRC / "image_misc_decode.cc"), "-lz", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "ubsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("sanitizer runtimes not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    seeds = tmp_path / "seeds"
    subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz" / "make_seeds.py"), str(seeds)], check=True)
    files = sorted(str(p) for p in seeds.iterdir())
    assert len(files) >= 35
    for seed in (1, 2):
        r = subprocess.run([str(exe), "30000", str(seed)] + files, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
        decoded, rejected = int(r.stdout.split()[1]), int(r.stdout.split()[3])
        assert decoded > 2000 and rejected > 2000, r.stdout      # both outcomes are exercised
