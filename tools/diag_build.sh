#!/bin/bash
# Diagnostic build of the library with extra -D flags for ONE kernel file (stamps / experiments):
#   tools/diag_build.sh NAME "-DMT_STAMPS -DFOO" [file.hip]   -> _exp/libmt_NAME.so   (never the product library; _exp/ travels to the GPU box, _diag/ does not)
set -e
cd "$(dirname "$0")/../masterthesis_amd/csrc"
make -s >/dev/null
mkdir -p ../../_exp
F=${3:-conv_pipe_kernel.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -c $F -o /tmp/diag_$1.o
OBJS=""
for o in conv_kernels conv_pipe_kernel conv_persist_kernel conv_patch_kernel conv_pipe_patch_kernel conv_wsreg_kernel stem_kernel wgrad_pipe_kernel wgrad_rows_kernel weight_pack_kernels conv_aux_kernels pointwise_kernels conv_api norm_kernels elementwise_kernels loss_kernels spectral_norm_kernels misc_kernels comm_api; do
  if [ "$o.hip" = "$F" ]; then OBJS="$OBJS /tmp/diag_$1.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -o ../../_exp/libmt_$1.so
echo built _exp/libmt_$1.so
