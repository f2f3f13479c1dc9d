#!/bin/bash
# variant build of libposepipe_hip.so for kernel ablations: bash tools/build_variant.sh <tag> "<extra hipcc flags>"
# -> posepipeline_amd/libposepipe_hip_<tag>.so (select with POSEPIPE_LIB=...); only conv_split.hip is recompiled
set -e
R=$(cd $(dirname $0)/.. && pwd)
TAG=$1; FLAGS=$2
make -C $R/posepipeline_amd/csrc > /dev/null
mkdir -p $R/build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$R/posepipeline_amd/csrc -ffp-contract=off -Wno-unused-function $FLAGS \
    -x hip -c $R/posepipeline_amd/csrc/conv_split.hip -o $R/build/variants/conv_split_$TAG.o
OBJS=$(ls $R/build/csrc/*.o | grep -v conv_split.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/build/variants/conv_split_$TAG.o -o $R/posepipeline_amd/libposepipe_hip_$TAG.so
echo built $R/posepipeline_amd/libposepipe_hip_$TAG.so
