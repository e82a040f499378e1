# cache-policy bits of K2u's off-grid flush stores: stereo, 63 full frames per clip off the 128-byte grid (+ the short frame)
cd "$GRAFT_REPO_ROOT"
for l in "" pol_sc0 pol_sc1 pol_sc01 pol_ntsc1 pol_ntsc0; do
  echo "== ${l:-product (plain write-back)}"
  QOAUDIO_AMD_LIB=${l:+$PWD/tools/k2lab/bin/lib_$l.so} QOA_ONLY=0,3 python tools/k2lab/short_frames.py 2>&1 | grep -v amdgpu.ids
done
