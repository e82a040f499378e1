#!/bin/bash
# Lab build of the whole library with extra flags on the decode translation unit only:
#   build_lib.sh <name> [-DQOA_K2U_... ...]   -> tools/k2lab/bin/lib_<name>.so   (use with QOAUDIO_AMD_LIB=...)
# For generated-row variants export QOA_DEC_GEN="<env assignments for tools/gen_decode_asm.py>" (a private .inc is generated).
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p tools/k2lab/bin/obj_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iqoaudio_amd/csrc"
if [ -n "$QOA_DEC_GEN" ]; then
  env $QOA_DEC_GEN QOA_GEN_OUT=$PWD/tools/k2lab/bin/obj_$name/qoa_decode_asm_lab.inc python tools/gen_decode_asm.py > /dev/null
  F="$F -DQOA_DEC_ASM_INC=\"$PWD/tools/k2lab/bin/obj_$name/qoa_decode_asm_lab.inc\""
fi
hipcc $F "$@" -c qoaudio_amd/csrc/qoa_decode.hip -o tools/k2lab/bin/obj_$name/d.o
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/k2lab/bin/lib_$name.so tools/k2lab/bin/obj_$name/d.o qoaudio_amd/csrc/qoa_kernels.o qoaudio_amd/csrc/qoa_kernels_rec.o qoaudio_amd/csrc/qoa_capi.o
echo built tools/k2lab/bin/lib_$name.so
