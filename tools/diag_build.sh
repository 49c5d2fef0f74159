#!/bin/bash
# Diagnostic build of the library with extra -D flags for conv_pipe_kernel.hip (stamps / experiments):
#   tools/diag_build.sh NAME "-DMT_STAMPS -DFOO"   -> _diag/libmt_NAME.so   (never the product library)
set -e
cd "$(dirname "$0")/../masterthesis_amd/csrc"
make -s >/dev/null
mkdir -p ../../_diag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -c conv_pipe_kernel.hip -o /tmp/pipe_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 conv_kernels.o /tmp/pipe_$1.o pointwise_kernels.o conv_api.o \
  norm_kernels.o elementwise_kernels.o loss_kernels.o misc_kernels.o -o ../../_diag/libmt_$1.so
echo built _diag/libmt_$1.so
