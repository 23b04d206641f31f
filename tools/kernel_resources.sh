#!/bin/bash
# tools/kernel_resources.sh BUILD_LOG [PATTERN]: one line per kernel out of a hipcc -Rpass-analysis=kernel-resource-usage log
grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Occupancy|LDS Size|ScratchSize" "$1" | sed 's/remark: [^ ]*:[0-9]*:[0-9]*: *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - - - \
  | sed 's/Function Name: //; s/TotalSGPRs: /S /; s/VGPRs: /V /; s/AGPRs: /A /; s/ScratchSize \[bytes\/lane\]: /scratch /; s/Occupancy \[waves\/SIMD\]: /occ /; s/LDS Size \[bytes\/block\]: /lds /' \
  | while IFS=$'\t' read -r name rest; do echo "$(echo "$name" | c++filt | cut -c1-90) | $rest"; done | grep -E "${2:-.}"
