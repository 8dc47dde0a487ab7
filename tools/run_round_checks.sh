#!/bin/bash
python tools/probes/host_register_probe.py
python tools/host_entry_packed.py
