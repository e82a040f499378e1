// qoa_decode_w.hip -- the general decode kernel's PARTIAL variants (frames whose PCM is no whole number of 16-byte pieces): part
// 4 of qoa_decode.hip (see qoa_decode_u.hip).
#define QOA_DECODE_PART 4
#include "qoa_decode.hip"
