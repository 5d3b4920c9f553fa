// Exact-fp32 instantiations of the row-stationary convolution (k_rconv.hip): launch_rconv_f32.  A translation unit of its own
// so that the two product types compile in parallel.
#define MDT_TF_F32 1
#include "k_rconv.hip"
