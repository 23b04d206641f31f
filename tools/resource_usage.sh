#!/bin/bash
# Register / LDS / spill summary of every kernel in cm_api.hip (device-only compile, no GPU needed).
cd "$(dirname "$0")/../color_modem_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -c --cuda-device-only -o /dev/null \
    -Rpass-analysis=kernel-resource-usage "$@" cm_api.hip 2>&1 |
python3 -c '
import re, sys
cur = {}
for ln in sys.stdin:
    m = re.search(r"remark: (.*)", ln)
    if not m: continue
    t = m.group(1).replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    for key in ("VGPRs", "AGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize", "Occupancy", "LDS Size"):
        if t.startswith(key + ":"):
            cur[key] = t.split(":", 1)[1].strip()
    if t.startswith("LDS Size"):
        print("%-150s vgpr %s agpr %s sspill %s vspill %s scratch %s occ %s lds %s" % (
            cur["name"][:150], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("SGPRs Spill"), cur.get("VGPRs Spill"),
            cur.get("ScratchSize"), cur.get("Occupancy"), cur.get("LDS Size")))
'
