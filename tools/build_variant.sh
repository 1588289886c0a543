#!/bin/bash
# Build an experimental variant of the library WITHOUT touching the shipped puresound_amd/libpuresound_hip.so:
#   tools/build_variant.sh NAME "<-D flags>" [SOURCE.hip]  ->  tools/_variants/NAME.so   (select it with PURESOUND_HIP_LIB=...)
# Only SOURCE (default conv1x1_bf16.hip) is rebuilt with the flags; the other objects are the production ones.
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2; src=${3:-conv1x1_bf16.hip}; base=${src%.hip}
csrc=puresound_amd/csrc
make -C $csrc -j4 > /dev/null
mkdir -p tools/_variants
/opt/rocm/bin/hipcc $extra -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc \
  -c $csrc/$src -o tools/_variants/$name.o
objs=$(ls $csrc/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_variants/$name.so $objs tools/_variants/$name.o
rm -f tools/_variants/$name.o
echo "built tools/_variants/$name.so ($extra)"
