#!/bin/bash
cd "$(dirname "$0")"
V=k2_variant.hip
./build.sh base $V 2 4 4 &
./build.sh sink $V 2 4 4 -DLAB_SINK &
./build.sh nocompute $V 2 4 4 -DLAB_NOCOMPUTE &
./build.sh plain $V 2 4 4 -DLAB_PLAIN &
./build.sh occ8 $V 2 4 4 -DLAB_EXTRA_LDS=14336 &
./build.sh occ8_sink $V 2 4 4 -DLAB_EXTRA_LDS=14336 -DLAB_SINK &
./build.sh occ4_sink $V 2 4 4 -DLAB_EXTRA_LDS=40960 -DLAB_SINK &
./build.sh occ4 $V 2 4 4 -DLAB_EXTRA_LDS=40960 &
wait
./build.sh r8w1 $V 2 8 1 &
./build.sh r8w1_plain $V 2 8 1 -DLAB_PLAIN &
./build.sh r8w1_sink $V 2 8 1 -DLAB_SINK &
./build.sh r8w1_nocompute $V 2 8 1 -DLAB_NOCOMPUTE &
./build.sh r8w2 $V 2 8 2 &
./build.sh r16w1 $V 2 16 1 &
./build.sh r6w2 $V 2 6 2 &
./build.sh r5w4 $V 2 5 4 &
wait
./build.sh m_base $V 1 4 4 &
./build.sh m_nt $V 1 4 4 -DLAB_NT &
./build.sh m_r8w2 $V 1 8 2 &
./build.sh m_r8w2_nt $V 1 8 2 -DLAB_NT &
./build.sh m_r16w1 $V 1 16 1 &
./build.sh m_sink $V 1 4 4 -DLAB_SINK &
./build.sh m_nocompute $V 1 4 4 -DLAB_NOCOMPUTE &
wait
ls bin | wc -l
