// qoa_decode_v.hip -- K2u for the channel counts that do not divide a wavefront (3 / 5 / 6 / 7): part 3 of qoa_decode.hip
// (see qoa_decode_u.hip).
#define QOA_DECODE_PART 3
#include "qoa_decode.hip"
