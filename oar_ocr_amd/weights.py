"""Operator-supplied model files: the "real-weights" mode of SURVEY 8d.

The reference fetches its `.onnx` graphs and dictionaries by name and refuses a file whose size or SHA-256 differs from its registry
(oar-ocr-core/src/core/download/registry.rs; verification in core/download/mod.rs).  Nothing can be downloaded here, so an operator drops the
files into a directory (`models/` by convention) and this module applies the same check before a byte reaches the engine: a file that is not the
one the registry names is refused -- never silently used -- because the only point of the mode is network-level parity with the reference's
ONNX-Runtime path on exactly those weights (core/inference/ort_infer_execution.rs:178,281).

REGISTRY below restates the registry rows of the files this path uses (name -> sha256, size), each with the line it was read from."""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Tuple


@dataclass(frozen=True)
class Entry:
    sha256: str
    size: int
    line: int     # line of oar-ocr-core/src/core/download/registry.rs


REGISTRY: Dict[str, Entry] = {
    "pp-ocrv6_tiny_det.onnx": Entry("193bab7a04fca699a6c82e6abb5b81bdb28177f0abd4062552b04908dafb19f8", 1780590, 83),
    "pp-ocrv6_tiny_rec.onnx": Entry("9ef676d6ed3c88256a2d92c640c44f25b0c40947e111b14b8be8f594091563e6", 4462639, 84),
    "ppocrv6_tiny_dict.txt": Entry("c5cbe34ef40c29c4df07ed012bf96569cb69a2d2a01a07027e9f13cb832bd9cd", 27156, 100),
    "pp-ocrv5_server_det.onnx": Entry("9a910baffbefb807ff2f7bfaa72910e3e470bd17014d798386d87bb46f442839", 88116836, 77),
    "ch_svtrv2_rec.onnx": Entry("3fadaeecebd49d4df4f96155875be393e66161befc26258d3e62ee9968efd648", 84196641, 25),
    "ppocr_keys_v1.txt": Entry("a1c84d9bdb9ab29043c58896224d32941783eb821629618416dcb08f12886492", 26250, 85),
    "pp-ocrv5_server_rec.onnx": Entry("4bfffad2c62eb1340250455856978fb9fb19cb4776b264ae3c2f91c35fbb40b4", 84502992, 78),
    "ppocrv5_dict.txt": Entry("d1979e9f794c464c0d2e0b70a7fe14dd978e9dc644c0e71f14158cdf8342af1b", 74012, 90),
}

# BASELINE.json configs -> (detector, recognizer, dictionary) file names
CONFIG_FILES: Dict[int, Tuple[str, str, str]] = {
    1: ("pp-ocrv6_tiny_det.onnx", "pp-ocrv6_tiny_rec.onnx", "ppocrv6_tiny_dict.txt"),
    2: ("pp-ocrv5_server_det.onnx", "ch_svtrv2_rec.onnx", "ppocr_keys_v1.txt"),
    3: ("pp-ocrv6_tiny_det.onnx", "pp-ocrv6_tiny_rec.onnx", "ppocrv6_tiny_dict.txt"),
    4: ("pp-ocrv6_tiny_det.onnx", "pp-ocrv6_tiny_rec.onnx", "ppocrv6_tiny_dict.txt"),
}


class WeightsError(RuntimeError):
    pass


def verify_file(path, registry: Optional[Dict[str, Entry]] = None) -> bytes:
    """Reads `path` and returns its bytes if -- and only if -- the registry has a row for its file name and both the size and the SHA-256 match."""
    registry = REGISTRY if registry is None else registry
    p = Path(path)
    e = registry.get(p.name)
    if e is None:
        raise WeightsError(f"{p.name}: no registry row for this file name (known: {', '.join(sorted(registry))})")
    if not p.is_file():
        raise WeightsError(f"{p}: missing")
    data = p.read_bytes()
    if len(data) != e.size:
        raise WeightsError(f"{p.name}: {len(data)} bytes, the registry row (registry.rs:{e.line}) says {e.size}")
    got = hashlib.sha256(data).hexdigest()
    if got != e.sha256:
        raise WeightsError(f"{p.name}: sha256 {got} differs from the registry row (registry.rs:{e.line}) {e.sha256}")
    return data


def load_config(models_dir, config: int, registry: Optional[Dict[str, Entry]] = None, files: Optional[Tuple[str, str, str]] = None):
    """(detector bytes, recognizer bytes, dictionary lines, report) for a BASELINE config, every file verified.  `report` lists name / size / sha256."""
    det_n, rec_n, dict_n = files or CONFIG_FILES[config]
    d = Path(models_dir)
    det, rec, dic = (verify_file(d / n, registry) for n in (det_n, rec_n, dict_n))
    lines = read_dict_bytes(dic)
    report = [{"file": n, "bytes": len(b), "sha256": hashlib.sha256(b).hexdigest()} for n, b in ((det_n, det), (rec_n, rec), (dict_n, dic))]
    return det, rec, lines, report


def read_dict_bytes(data: bytes) -> List[str]:
    """Dictionary entries as api.read_dict yields them from the file's text (utils/dict.rs:35-43 reads `content.lines()`; src/oarocr/ocr.rs:277-291 and
    decode.rs:120 keep the first character of each non-empty line)."""
    from . import api
    return api.read_dict(data.decode("utf-8"))


def present(models_dir, config: int) -> List[str]:
    """Which of the config's three files exist in `models_dir` (unverified): for messages."""
    d = Path(models_dir)
    return [n for n in CONFIG_FILES[config] if (d / n).is_file()]
