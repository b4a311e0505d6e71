#!/bin/bash
# Compile one csrc/*.hip with the product flags and print VGPR / spill numbers per kernel (greps for $2 if given); keeps the ISA in /tmp.
f=${1:-gemm}
cd /root/repo/safevla_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -I ../../include -c $f.hip -o /tmp/${f}_chk.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|ScratchSize" | grep -E "error|Name|VGPRs:|Scratch|Spill" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - - - - | grep -E "${2:-.}"
