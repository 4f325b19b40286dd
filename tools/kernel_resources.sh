#!/bin/bash
# Registers, scratch (spills) and occupancy of every kernel of a source file (gfx950, the library's flags):
#   bash tools/kernel_resources.sh sba-gan_amd/csrc/encoder.hip [more.hip ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/sba-gan_amd/csrc -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
  awk -v F="$(basename $f)" '
    /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name) }
    /    VGPRs: /   { v=$0; sub(/.*VGPRs: /,"",v); sub(/ .*/,"",v) }
    /    AGPRs: /   { a=$0; sub(/.*AGPRs: /,"",a); sub(/ .*/,"",a) }
    /ScratchSize/   { s=$0; sub(/.*: /,"",s); sub(/ .*/,"",s) }
    /Occupancy/     { o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o) }
    /LDS Size/      { l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); printf "%-14s vgpr %3s agpr %3s scratch %4s occ %s lds %6s  %s\n", F, v, a, s, o, l, name }'
done
