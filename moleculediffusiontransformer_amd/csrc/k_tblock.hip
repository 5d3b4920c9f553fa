// MDT_OP_TBLOCK, variant 0 (64-row workgroups, C = 128): dispatch to the loader-wave kernel (k_tblock_lw.hip).
//
// The first-generation kernels of this file (no loader waves: every wave issued its own LDS-DMA; 64-row workgroups for C = 128 and
// 256, cross-attention with up to 64 keys per 16 rows read straight from global memory) and the 16-row feature-split form of
// k_tblock16.hip (variant 1) were removed in round 3: no configuration of BASELINE.json, of the reference's notebooks or of the test
// suite reached them any more (k_tblock_lw / k_tblock32 / k_tf128 / k_tf256 cover every shape the compiler fuses; layers outside
// their envelope run layer by layer).  git history keeps them (round 2: csrc/k_tblock.hip, csrc/k_tblock16.hip).
#include "mdt_kernels.h"

namespace mdt {

hipError_t launch_tblock(const TBlockArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (!tblock_lw_supported(a)) return hipErrorInvalidValue;
  return launch_tblock_lw(a, s);
}

}  // namespace mdt
