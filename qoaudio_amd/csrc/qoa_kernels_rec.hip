// qoa_kernels_rec.hip -- the recording instantiations of K1 (qoa_encode_kernel<4, NCH, HELPER, REC = true>: per-wavefront
// form counts and cycles for plans created with QOA_PLAN_WAVE_STATS) as a translation unit of their own: the same source as
// qoa_kernels.hip, of which only the kernel template and launch_encode_rec() are compiled here.  They cost hipcc as long again
// as the rest of that file; as two objects make builds them side by side.
#define QOA_REC_TU 1
#include "qoa_kernels.hip"
