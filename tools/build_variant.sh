#!/bin/bash
# tools/build_variant.sh NAME "-DXC_STAMPS" (or any -D flag of xc_hist.hip)  -> xcontour_amd/libxc_NAME.so (diagnostic builds only)
set -e
cd /root/repo/xcontour_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value"
/opt/rocm/bin/hipcc $F $2 -c xc_hist.hip -o /tmp/h_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libxc_$1.so xc_capi.o /tmp/h_$1.o xc_misc.o xc_lwa.o xc_sort.o xc_cross.o xc_comm.o -ldl
echo built libxc_$1.so
