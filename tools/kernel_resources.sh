#!/bin/bash
# Register / LDS / scratch use and occupancy of every kernel in a translation unit (the compiler's own report):
#   tools/kernel_resources.sh [file.hip] [name filter (regex on the demangled name)]
cd "$(dirname "$0")/../locus_amd/csrc"
f=${1:-lh_kernels.hip}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep "remark:" | sed 's/.*remark: *//; s/ \[-Rpass.*//' |
  awk '
    /^Function Name:/ { name=$3 }
    /^TotalSGPRs:/ { s=$2 }
    /^VGPRs:/ { v=$2 }
    /^AGPRs:/ { a=$2 }
    /^ScratchSize/ { sc=$NF }
    /^Occupancy/ { o=$NF }
    /^VGPRs Spill/ { sp=$NF }
    /^LDS Size/ { printf "%s vgpr %s agpr %s sgpr %s scratch %s spill %s occ %s lds %s\n", name, v, a, s, sc, sp, o, $NF }' |
  while read -r name rest; do echo "$(echo "$name" | c++filt | sed 's/(.*//') $rest"; done | grep -E "${2:-.}"
