# channels.py under several builds of the library (separate processes: compare effects above ~5 % only)
export QOA_CHANNELS=${QOA_CHANNELS:-5,6,7}
for rep in 1 2; do for l in product "$@"; do
  echo "== $l"
  if [ $l = product ]; then python tools/k2lab/channels.py 2>&1 | grep K2u; else QOAUDIO_AMD_LIB=$PWD/tools/k2lab/bin/lib_$l.so python tools/k2lab/channels.py 2>&1 | grep K2u; fi
done; done
