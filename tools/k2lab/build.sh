#!/bin/bash
# usage: build.sh <label> <variant.hip> <nch> <R> <WPB> [extra -D flags...]
cd "$(dirname "$0")"
label=$1; var=$2; nch=$3; R=$4; W=$5; shift 5
mkdir -p bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -DVARIANT_FILE="\"$var\"" -DLABEL="\"$label\"" -DLAB_NCH=$nch -DLAB_R=$R -DLAB_WPB=$W -DLAB_EXTRA_LDS=0 -DLAB_LOADMODE=0 -Dqoa=qoa_lab "$@" lab_main.hip -o bin/$label -L../../qoaudio_amd -lqoaudio_amd -Wl,-rpath,\$ORIGIN/../../../qoaudio_amd 2>&1 | grep -A3 'error' | head -8
