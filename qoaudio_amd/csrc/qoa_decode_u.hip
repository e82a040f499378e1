// qoa_decode_u.hip -- K2u (qoa_decode_uniform_kernel) for 1 / 2 / 4 / 8 channels, both block shapes, and its absolute-line
// variant, with their launchers, as a translation unit of their own: the same source as qoa_decode.hip, of which the Makefile
// builds part 1 (the general kernels) into qoa_decode.o, part 2 here and part 3 (3 / 5 / 6 / 7 channels) in qoa_decode_v.hip,
// side by side.
#define QOA_DECODE_PART 2
#include "qoa_decode.hip"
