#!/bin/bash
# Lab build of the library from a generator variant: build_lab.sh <name> <QOA_GEN_LAB list> [extra hipcc flags]
#   -> tools/k1lab/bin/lib_<name>.so (use with QOAUDIO_AMD_LIB=...).  Never touches the product's .inc / .so.
# (round 5: the recording instantiations -- qoa_kernels_rec.hip -- are built from the variant's texts too, so that
# EncodePlan.wave_stats() measures the variant's cycles per slice, wavefront by wavefront)
set -e
cd "$(dirname "$0")/../.."
name=$1; lab=$2; shift 2
mkdir -p tools/k1lab/bin/obj_$name
QOA_GEN_LAB=$lab QOA_GEN_OUT=$PWD/tools/k1lab/bin/obj_$name/qoa_slice_asm_lab.inc python tools/gen_slice_asm.py > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iqoaudio_amd/csrc -DQOA_SLICE_ASM_INC=\"$PWD/tools/k1lab/bin/obj_$name/qoa_slice_asm_lab.inc\" $*"
hipcc $F -c qoaudio_amd/csrc/qoa_kernels.hip -o tools/k1lab/bin/obj_$name/k.o &
hipcc $F -c qoaudio_amd/csrc/qoa_kernels_rec.hip -o tools/k1lab/bin/obj_$name/r.o &
hipcc $F -c qoaudio_amd/csrc/qoa_capi.cpp -o tools/k1lab/bin/obj_$name/c.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/k1lab/bin/lib_$name.so tools/k1lab/bin/obj_$name/k.o tools/k1lab/bin/obj_$name/r.o tools/k1lab/bin/obj_$name/c.o qoaudio_amd/csrc/qoa_decode.o
echo built tools/k1lab/bin/lib_$name.so
