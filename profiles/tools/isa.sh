#!/bin/bash
# isa.sh TAG [EXTRA flags...] : bhray_kernels.hip -> /tmp/isa/TAG.s (the product's compile flags), then the census of the dense RK and Euler step loops
TAG=$1; shift
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -S --cuda-device-only -o /tmp/isa/$TAG.s /root/repo/bhusie_amd/csrc/bhray_kernels.hip 2>&1 | grep -v "hip-link" 
for k in 'trace_kernel<1, false, false, true, 0>' 'trace_kernel<0, false, false, true, 0>'; do echo "== $k"; python /root/repo/profiles/tools/loop_isa.py /tmp/isa/$TAG.s "$k"; done
