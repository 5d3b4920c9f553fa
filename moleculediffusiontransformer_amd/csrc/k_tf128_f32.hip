// Exact-fp32 instantiations of the whole-transformer kernel of the 128-channel level (k_tf128.hip, template parameter F32):
// launch_tf128_f32.  A translation unit of its own so that the two product types compile in parallel.
#define MDT_TF_F32 1
#include "k_tf128.hip"
