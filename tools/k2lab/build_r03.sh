#!/bin/bash
# round-3 K2u variants (run in the build container): see run_r03.sh
cd "$(dirname "$0")"
set -e
./build.sh prod k2prod.hip 2 4 4 -DLAB_UNIFORM
./build.sh old bin/k2old.hip 2 4 4 -DLAB_UNIFORM -I. 
./build.sh pairdma k2prod.hip 2 4 4 -DLAB_UNIFORM "-DQOA_DEC_ASM_INC=\"$PWD/bin/inc_pair2.inc\""
./build.sh quad_w8 k2prod.hip 2 4 4 -DLAB_UNIFORM -DQOA_K2U_WPB_OVERRIDE=8
./build.sh quad_w4 k2prod.hip 2 4 4 -DLAB_UNIFORM -DQOA_K2U_WPB_OVERRIDE=4
ls -la bin
