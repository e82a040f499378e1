export QOAUDIO_AMD_LIB=$PWD/tools/k1lab/bin/lib_stamps.so
for fl in 8208 8256 4112 4160; do echo "== flags $fl (8192 force helper, 4096 no helper, 16 cpw1, 64 cpw4), 4 clips"; QOA_EXTRA_FLAGS=$fl python tools/k1lab/stamps.py 16 4 2>&1 | grep -v amdgpu.ids; done
