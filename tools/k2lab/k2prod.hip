// lab variant: the PRODUCT decode translation unit as it ships (K2u through launch_decode_uniform)
#include "../../qoaudio_amd/csrc/qoa_decode.hip"
