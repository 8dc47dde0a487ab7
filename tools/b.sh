#!/bin/bash
# build from anywhere: tools/b.sh  (prints the last line of the build log and the library's time stamp)
cd "$(dirname "$0")/.." && python -m oar_ocr_amd.build 2>&1 | tail -${1:-1}; ls -la --time-style=+%H:%M:%S oar_ocr_amd/lib/libOarMi355x.so | awk '{print $6, $7}'; date +%H:%M:%S
