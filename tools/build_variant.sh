#!/bin/bash
# tools/build_variant.sh NAME "-DXC_STAMPS" [file]  -> xcontour_amd/libxc_NAME.so (diagnostic builds only): rebuilds ONE
# translation unit (default xc_hist.hip) with extra -D flags and links it with the regular objects
set -e
cd "$(dirname "$0")/../xcontour_amd/csrc"
SRC=${3:-xc_hist.hip}
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value"
/opt/rocm/bin/hipcc $F $2 -c $SRC -o /tmp/v_$1.o
OBJS=""
for o in xc_capi.o xc_hist.o xc_hist_det.o xc_misc.o xc_lwa.o xc_sort.o xc_cross.o xc_comm.o; do
  if [ "$o" = "${SRC%.hip}.o" ]; then OBJS="$OBJS /tmp/v_$1.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libxc_$1.so $OBJS -ldl
echo built libxc_$1.so
