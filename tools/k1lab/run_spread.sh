# per-wavefront cycles per slice at config 3's geometry: how far is the slowest wavefront from the typical one?
export QOA_HWID=1 QOAUDIO_AMD_LIB=$PWD/tools/k1lab/bin/lib_hwid.so QOA_SHAPES=1024x2
for f in 16 64; do python tools/k1lab/small_launch.py $f 2>&1 | grep -E "^---|default|no helper"; done
QOA_SAME=1 python tools/k1lab/small_launch.py 16 2>&1 | grep -E "^---|default|no helper"
