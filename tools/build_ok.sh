#!/bin/bash
# build the library and fail loudly (a failed build must never be followed by a GPU run of the stale .so)
cd "$(dirname "$0")/.." && python -c "
from oar_ocr_amd import build
build.build_lib(verbose=False)" > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; echo BUILD FAILED; exit 1; }
grep -i "error" /tmp/build.log && { echo BUILD FAILED; exit 1; }
echo build ok
