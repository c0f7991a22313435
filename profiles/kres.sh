#!/bin/bash
# kernel resource usage of one HIP source (cross-compiles, no GPU): profiles/kres.sh vita_amd/csrc/vh_gemm_sp.hip [filter]
src=$1; filt=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $OLDPWD/$src -o /tmp/kres.o -I $OLDPWD/vita_amd/csrc -I $OLDPWD/include \
  -Wno-unused-result -Rpass-analysis=kernel-resource-usage $EXTRA 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | \
  sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | awk '/Function Name/{if (l) print l; l=$0; next} {l=l" | "$0} END{print l}' | c++filt | grep -E "$filt" | sed -E 's/\(anonymous namespace\):://g; s/Function Name: //'
