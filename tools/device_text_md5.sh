#!/bin/bash
# md5 of the gfx950 machine code (.text of the device code object) inside a built librvio_hip.so: two builds of the same kernels agree here even
# when the shared objects differ in their notes / hashes (hipcc embeds the command line and the source text's hash).
# usage: tools/device_text_md5.sh [r-vio_amd/librvio_hip.so]
set -e
LIB=${1:-r-vio_amd/librvio_hip.so}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.text $T/dev.co $T/dev.text
md5sum $T/dev.text | awk '{print $1}'
rm -rf $T
