// Internal launch interface between the C ABI / program executor (mdt_api.cpp)
// and the gfx950 kernels (*.hip).  Not part of the public ABI (include/mdt_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Tuning builds.  The switches below exist to TIME a section of a ring kernel with it compiled out or altered; every one of them
// produces WRONG (or unsynchronised) results.  They compile only together with -DMDT_TUNING, and a library built that way
// identifies itself: mdt_abi_version() carries MDT_ABI_TUNING_BIT, which runtime.load_library() refuses unless the caller
// (a tool under tools/) sets MDT_ALLOW_TUNING=1 -- a timing-only library cannot be loaded for real sampling by accident.
#if (defined(MDT_ABL_BAR) || defined(MDT_ABL_VMWAIT) || defined(MDT_ABL_LGKM) || defined(MDT_ABL_LDSBC) || defined(MDT_ABL_GELU) ||     \
     defined(MDT_ABL_ATTN) || defined(MDT_RING_NODRAIN) || defined(MDT_STAGGER) || defined(MDT_XH_D2) ||                               \
     (defined(MDT_XH_ADDR) && MDT_XH_ADDR == 0) || (defined(MDT_XH_MODE) && MDT_XH_MODE != 0)) &&                                       \
    !defined(MDT_TUNING)
#error "MDT_ABL_* / MDT_RING_NODRAIN / MDT_STAGGER / MDT_XH_D2 / MDT_XH_ADDR=0 / MDT_XH_MODE!=0 give wrong results: timing builds only, add -DMDT_TUNING"
#endif
#define MDT_ABI_TUNING_BIT 0x40000000

#include <cstdlib>
// Environment switches of the TUNING tools (MDT_TILE / MDT_TILE1 / MDT_TILE16: force a GEMM tile; MDT_NO_AS / MDT_NO_PREFETCH:
// disable a kernel form; MDT_DBG: ablation bits of k_gemm_as): read by -DMDT_TUNING builds only.  The shipped library ignores
// them -- a configuration nobody tests cannot be reached by setting a variable.
static inline const char* mdt_tuning_env(const char* name) {
#ifdef MDT_TUNING
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// priority of the loader waves of the ring kernels (s_setprio): 3 = their few instructions issue ahead of the MFMA waves
#ifndef MDT_LOADER_PRIO
#define MDT_LOADER_PRIO 3
#endif

namespace mdt {

struct GemmArgs {
  const float* A;
  const float* W;     // fp32 [N][K], or the bf16 hi plane when W_lo != nullptr
  const float* W_lo;  // bf16 lo plane [N][K] (split-bf16 GEMM) or nullptr (exact fp32 MFMA)
  const float* bias;
  float* out;
  const float* res;
  const float* p0;  // gain
  const float* p1;  // bias of the norm
  const float* p2;  // GroupNorm stats [batch][G][2] = (mean, rstd)
  const float* p3;  // FiLM [scale(cin) | shift(cin)]
  int M, r_out, r_in, lda, cin, taps, t_stride, t_dj, t_off;
  int N, ldc, o_rows, o_stride, o_off, ldr;
  int pro, groups, gsize, pro_silu, act, a_col, o_col;
  int wfmt;    // 1: W is ONE bf16 plane (plain bf16 products), else fp32 / hi + lo planes
  int phases;  // > 1: ConvTranspose1d(k = 2 f, stride f, padding f / 2) as f output phases of 2 taps each in ONE launch:
               // phase ph = blockIdx.z uses weights [ph][N][K], t_off = (ph < f/2), o_off = f t_off + ph - f/2
  float eps;
};
#ifdef __HIPCC__
// per-phase view of a multi-phase GEMM (first statement of every GEMM kernel)
__device__ __forceinline__ void gemm_select_phase(GemmArgs& g) {
  if (g.phases > 1) {
    const int ph = blockIdx.z, f = g.phases;
    const int shift = ph < f / 2 ? 1 : 0;
    const long wo = (long)ph * g.N * (g.taps * g.cin) / ((g.W_lo || g.wfmt == 1) ? 2 : 1);   // floats: bf16 planes hold 2 per float
    g.W += wo;
    if (g.W_lo) g.W_lo += wo;
    g.t_off = shift;
    g.o_off = f * shift + ph - f / 2;
  }
}
#endif
hipError_t launch_gemm(const GemmArgs& g, hipStream_t s);          // exact fp32 MFMA (k_gemm.hip)
hipError_t launch_gemm_bf16x3(const GemmArgs& g, hipStream_t s);
hipError_t launch_gemm_bf16(const GemmArgs& g, hipStream_t s);     // plain bf16 products: W = one bf16 plane   // split-bf16 MFMA (k_gemm_bf16x3.hip)

// plain-bf16 GEMM with both operands in bf16 (k_gemm_b16.hip) and the pass that prepares its A operand
struct Gemm16Args {
  const unsigned short* A;   // bf16 [batches * rows][lda]
  const unsigned short* W;   // bf16 [N][taps * cin]
  const float* bias;
  const float* res;
  float* out;
  int M, N, cin, taps, rows;  // rows per sample (taps stay inside a sample; source rows outside read as zero)
  int lda, a_col;             // bf16 elements
  int t_dj, t_off, ldc, ldr, o_col, act;
  int out16;                  // 1: out is bf16 (ldc / o_col in bf16 elements)
  int res16;                  // 1: res is bf16 (ldr in bf16 elements): the bf16 residual stream of the plain-bf16 mode (round 6)
  unsigned short* copy16;     // fp32 output: also a bf16 copy [M][N] of it (the next GEMM's A operand), or nullptr
  // LayerNorm folded into the GEMM (round 6, all-bf16 epilogue only): A is the RAW bf16 residual stream (taps == 1); the waves gather
  // each row's (sum, sum of squares) from the A fragments they feed the MFMAs anyway, and the epilogue applies
  const float* csum;          //   out = rstd[m] (acc - mean[m] csum[n]) + bias[n], csum[n] = sum_k W[n][k] (the bf16 values); or nullptr
  float eps;
};
bool gemm_b16_supported(int cin, int taps, int lda, int a_col);
bool gemm_b16_epilogue_ok(const Gemm16Args& g);   // N / ldc / o_col / ldr % 4 == 0 and 16-byte aligned tensors (float4 epilogue)
hipError_t launch_gemm_b16(const Gemm16Args& g, hipStream_t s);
void set_tile16(int v);          // tuning hook behind mdt_set_tuning("tile16", v)
void set_w16(int v);             // mdt_set_tuning("w16", 0 / 1): the all-bf16 epilogue of k_gemm_b16 off / on
struct Prep16Args {
  const float* a;
  unsigned short* out;        // bf16 [total_rows][cin]
  const float* p0; const float* p1; const float* p2; const float* p3;
  int total_rows, rows, lda, a_col, cin, pro, groups, gsize, pro_silu;
  float eps;
  int in16;                   // 1: a is bf16 (lda / a_col in bf16 elements): LayerNorm of the bf16 residual stream (round 6; PRO 0 / 1)
};
hipError_t launch_prep16(const Prep16Args& g, hipStream_t s);
bool gemm_as_eligible(const GemmArgs& g);                           // wide-N / small-K layers
hipError_t launch_gemm_as(const GemmArgs& g, hipStream_t s);       // A-stationary split-bf16 (k_gemm_as.hip)

// "Done once" state that belongs to the DEVICE, not the process: hipFuncSetAttribute(MaxDynamicSharedMemorySize) and values read
// from hipDeviceGetAttribute are per (function, device), so a process that drives a second GPU has to repeat them there (ADVICE
// r4).  One bit per device ordinal; first() is true the first time it is called with that device current.
struct DevOnce {
  unsigned long long mask = 0;
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;      // unknown: repeat the (cheap, idempotent) call
    if ((mask >> d) & 1ull) return false;
    mask |= 1ull << d;
    return true;
  }
};

#ifdef __HIPCC__
// Loader waves of a ring kernel, after their last tile: pull the NEXT launch's weight stream into this XCD's L2 -- one dword per
// 128-byte line, the lines shared out over the workgroups of the XCD (linear workgroup id % 8).  Between two uses of a layer's
// weights an evaluation streams ~85 other layers, so every launch otherwise starts its stream from HBM / the Infinity Cache:
// 27.0-27.4 us against 25.8 us with the weights L2-resident for the most frequent launch (tools/cold_weights_probe.py).
__device__ __forceinline__ void prefetch_next_weights(const void* p, int lines, int lane256) {
  if (!p) return;
  const int nwg = gridDim.x * gridDim.y, id = blockIdx.x + blockIdx.y * gridDim.x;
  const int nx = (nwg + 7) >> 3, xw = id >> 3;
  const int per = (lines + nx - 1) / nx;
  const int lo = xw * per, hi = lo + per < lines ? lo + per : lines;
  const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
  // volatile loads through the compiler: it tracks the landing registers and places the waits itself (an inline-asm load with a
  // bare "=v" output may have its register reused while the load is still in flight); the values are folded into a sink the
  // optimiser cannot drop
  unsigned acc = 0;
  for (int l = lo + lane256; l < hi; l += 256) acc |= *reinterpret_cast<const volatile unsigned*>(b + (int64_t)l * 128);
  asm volatile("" :: "v"(acc));
}
#endif

struct GnStatsArgs {
  const float* x;
  float* stats;  // [batch][G][2]
  int batch, rows, ld, groups, gsize;
  float eps;
};
hipError_t launch_gn_stats(const GnStatsArgs& g, hipStream_t s);

struct GnActArgs {
  const float* x;
  float* y;
  const float* gamma;
  const float* beta;
  const float* film;   // [scale(ld) | shift(ld)] or nullptr
  int batch, rows, ld, groups, gsize, silu;
  float eps;
  int out16;   // 1: y is bf16 [rows][ld] (the A operand of a bf16 x bf16 GEMM)
  // round 6: the input as cat([x (ca channels), scale2 * x2 (ld - ca channels)]) without the concatenated tensor (the up path's
  // ResnetBlock1d, modules.py:828-829), and a raw bf16 copy of that input (the A operand of the block's to_out convolution)
  const float* x2;          // or nullptr
  int ca;
  float scale2;
  unsigned short* raw16;    // bf16 [rows][ld], or nullptr
};
bool gn_act_eligible(int rows, int ld, int groups, int gsize);
hipError_t launch_gn_act(const GnActArgs& a, hipStream_t s);

// MDT_OP_RCONV (k_rconv.hip): GroupNorm + FiLM + SiLU + Conv1d(k = 1 | 3) with C input and C output channels
struct RConvArgs {
  const float* x;      // [M][lda]
  const float* x2;     // second input source [M][lda2] or nullptr (its tiles / gain / bias follow the first source's)
  const float* w;      // weight tiles [64 features][128 k], order (source, tap, K half, feature chunk)
  const float* bias;   // [C] or nullptr
  const float* res;    // [M][ldr] added to the result, or nullptr (may alias out)
  float* out;          // [M][ldc]
  const float* gamma;  // GroupNorm gain / bias of the C input channels (gsize > 0)
  const float* beta;
  const float* film;   // [scale | shift] rows film_ld apart, or nullptr
  const float* dbgbuf; // diagnostic stamps (tuning builds), normally nullptr
  int M, T, C, lda, lda2, ldc, ldr, taps, gsize, silu, film_ld;
  float eps, in_scale, in_scale2;
  const void* pf_ptr;  // weight stream of the NEXT launch (pulled into the L2s by the idle loader waves), or nullptr
  int pf_lines;        // ... its size in 128-byte lines
  int wf32;            // 1: fp32 fragment tiles, exact fp32 MFMA products (MDT_R_WF32)
  int ksrc;            // > 1: x is [M][ksrc C]: ksrc blocks of C channels accumulate into the C outputs (MDT_R_KSRC; 2, or ksrc C == 1024)
  int half_out;        // 1: only output channels 0 .. C / 2 - 1 exist (C = 256; ldc >= 128; MDT_R_HALF_OUT)
  int nb;              // > 1: NB x C output channels -- NB convolutions of the same input, blocks of weights / bias / out consecutive (MDT_R_NB)
};
// MDT_OP_GEMM with MDT_G_WFMT = 16 (k_proj.hip): row-stationary projection, weights as ring tiles
struct ProjArgs {
  const float* x;      // [M][lda] (A_COL applied)
  const float* w;      // 32 KB tiles [64 features][128 k] (bf16 hi plane | lo plane), order (64-feature chunk, K half)
  const float* bias;   // [N] or nullptr
  const float* res;    // [M][ldr] added to the result, or nullptr (may alias out)
  const float* gamma;  // LayerNorm gain / bias over the K input channels (ln = 1)
  const float* beta;
  float* out;          // [M][ldc] (O_COL applied)
  int M, N, K, lda, ldc, ldr, ln;
  float eps;
  int nch;             // (set by the launcher) 64-feature chunks per workgroup
  int wf32;            // 1: fp32 fragment tiles, exact fp32 MFMA products (MDT_G_WFMT = 17)
};
bool proj_supported(int K, int N, int lda, int ldc, int ldr);
hipError_t launch_proj(const ProjArgs& a, hipStream_t s);

bool rconv_supported(int C, int T, int taps, int gsize);
hipError_t launch_rconv(const RConvArgs& a, hipStream_t s);
hipError_t launch_rconv_f32(const RConvArgs& a, hipStream_t s);   // (k_rconv_f32.hip); reached through launch_rconv

// MDT_OP_RESBLOCK (k_resblock.hip): a whole ResnetBlock1d of the 64-token level (one GroupNorm group) in one launch
struct ResBlockArgs {
  const float* x;      // [B][64][cin]
  float* out;          // [B][64][cout]
  const float* w;      // bf16 hi/lo MFMA fragments [step][row tile][hi | lo][64 lanes][8], steps: conv1 | conv2 | to_out
  const float* vec;    // gamma1[cin] | beta1[cin] | b1[cout] | gamma2[cout] | beta2[cout] | b2 + to_out bias [cout]
  const float* film;   // [scale | shift] rows film_ld apart, or nullptr
  int B, T, cin, cout, film_ld;
  float eps;
  int wf32;            // 1: w = fp32 MFMA fragments [step][row tile][half][64 lanes][4], exact fp32 products (MDT_K_WF32)
  int cin_real, cout_real;   // channels the GroupNorm statistics run over (<= cin / cout: the rest is zero padding)
  int patch_in, patch_out;   // > 1: the input is still patched / the output leaves patched ([T / p][C p]; MDT_K_PATCH_IN / _OUT)
};
bool resblock_supported(int T, int cin, int cout);
hipError_t launch_resblock(const ResBlockArgs& a, hipStream_t s);

struct AttnArgs {
  const float* q;
  const float* k;  // v = k + heads*64
  float* out;
  int batch, T, Tk, heads, ldq, ldkv, ldo, kv_bstride;
  float scale;
  int out16;   // MDT_OP_ATTN: 1 = out is bf16 (ldo in bf16 elements)
  int split_scores;   // MDT_OP_ATTN_CTX: 1 = scores as split-bf16 products (MDT_A_SPLIT), 0 = exact fp32 MFMA
  int in16;    // MDT_OP_ATTN: bit 0 = q is bf16, bit 1 = k | v are bf16 (ldq / ldkv in bf16 elements; q / k point at bf16 data)
};
hipError_t launch_attn(const AttnArgs& a, hipStream_t s);
// MDT_OP_ATTN_CTX: rows q [batch][T * heads][128] against the normalised context k [batch | 1][Tk][ldkv >= 128] (K = V)
hipError_t launch_attn_ctx(const AttnArgs& a, hipStream_t s);

struct TBlockArgs {
  float* x;            // [M][ldx] fp32, updated in place
  const float* w;      // weight tile stream (bf16 hi/lo planes, 256*C bytes per tile)
  const float* bias;   // SELF [bq|bk|bv|bo], CROSS [bq|bo], FF [b1|b2]
  const float* kv;     // CROSS: hoisted K|V rows [sample][Tk][ldkv]
  const float* kv2;    // dual batch: batch-invariant K / V rows of the second half of the samples (stride 0), or nullptr
  const float* dbgbuf; // diagnostic stamps (MDT_DBG & 8), normally nullptr
  float* part;         // k_tblock32 with nsplit > 1: partial outputs [nsplit][M][C] (no bias / residual), summed by k_tb_reduce
  int nsplit;          // workgroups sharing a row block, each taking nchunk / nsplit heads or hidden chunks
  // chained form (k_tblock32, no reduce launch): the block input is x + pin, head group 0 writes xout = input + its
  // partial + bias, head group 1 writes its bare partial to pout; xout / pout never alias x / pin
  float* xout;         // nullptr: in place on x (variants 2, 3)
  const float* pin;    // partial of the previous block's second head group, or nullptr
  float* pout;         // where this block's second head group leaves its partial (nsplit == 2)
  // feed-forward with the transformer's closing 1x1 convolution folded in (post > 0, MODE_FF only):
  //   xout = Wout (x + FF(x)) + bout = (Wout W2) gelu(W1 x + b1) + Wout x + (Wout b2 + bout)
  // the host stores Wout W2 as the W2 tiles, Wout as `post` extra output tiles (natural k order) and the fused bias
  int post;            // number of extra [C][64] output tiles (C / 64), 0 = plain residual feed-forward
  int wf32;            // 1 (variant 0 only): fp32 fragment tiles, exact fp32 MFMA products (MDT_B_WF32)
  int mode, C, M, T, nchunk, nbias, ldx, Tk, kv_bstride, ldkv, nheads, nsamples;
  float eps, scale;
  const void* pf_ptr;  // weight stream of the NEXT launch (ring kernels: pulled into the L2s by the loader waves), or nullptr
  int pf_lines;
};
bool tblock_lw_supported(const TBlockArgs& a);                  // k_tblock_lw.hip: C = 128 self-attention / feed-forward
hipError_t launch_tblock_lw(const TBlockArgs& a, hipStream_t s);
hipError_t launch_tblock32(const TBlockArgs& a, hipStream_t s);   // 32-row workgroups, C = 256, sub-tile stream (k_tblock32.hip)

// MDT_OP_TF128 (k_tf128.hip): a whole Transformer1d of a C = 128 level in one launch
struct TFArgs {
  const float* x;        // [M][128] transformer input
  float* out;            // [M][128] transformer output (may alias x: rows are read before the first store of the same wave)
  const float* w;        // weight tile stream of every segment (32 KB tiles: bf16 hi plane + lo plane), consumption order
  const float* vec;      // vectors, nvec floats, staged into LDS: [to_in bias 128] then per block [bq 64 heads | bo 128] (self),
                         // the same (cross), [b1 64 nff | b2 128] (feed-forward)
  const unsigned* tiles; // per tile: kind (0 projection, 1 output, 2 K rows, 3 V rows) | aux << 2; aux = index into the
                         // weight stream, or (cross layer << 4 | head) for K / V tiles
  const float* kv;       // hoisted K | V rows of this transformer's first cross layer [sample][Tk][ldkv]; layer l at + l * kv_lstride
  const float* kv2;      // dual batch: batch-invariant K | V rows of the second half of the samples (layer stride kv2_lstride)
  const float* dbgbuf;
  int64_t kv_lstride, kv2_lstride;
  int M, T, NT, nvec, Tk, kv_bstride, ldkv, nheads, nsamples;
  int has_in;            // 1: the stream starts with GroupNorm(32) + Conv1d(k = 1) (Transformer1d.to_in, 2 projection tiles)
  int nblocks;           // TransformerBlocks: self-attention, cross-attention (iff kv), feed-forward each
  int nff;               // hidden chunks of 64 of the feed-forward blocks
  int npost;             // 2: Transformer1d.to_out folded into the LAST feed-forward block (two extra output tiles), 0: none
  float eps_ln, eps_gn, scale;
  // MDT_OP_TF128 only: ResnetBlock1d blocks in front of the transformer (k_tf128.hip)
  int res_kind;          // 0 none; 1 single source, every block's output also stored to skip + rb * skip_stride;
                         // 2 input cat([x, skip_scale * skip[rb]]) with skip[rb] = skip + rb * skip_stride
  int n_res, res_pair1, res_pair2;   // blocks; GroupNorm groups of 32 (1) or 16 (0) channels in block1 / block2
  int nfilm;             // FiLM floats staged from `film` (2 C per block, a multiple of 256)
  const float* film;     // (scale | shift) rows of the blocks, contiguous (shared time-mapping row)
  float* skip;
  int64_t skip_stride;
  float skip_scale, eps_res;
  const void* pf_ptr;    // weight stream of the NEXT launch (pulled into the L2s by the loader waves), or nullptr
  int pf_lines;
  // MDT_OP_TF256 with the row blocks' heads split over workgroup PAIRS (k_tf256.hip, NSPLIT = 2)
  int nsplit;            // 1 | 2; 2: `tiles` holds two descriptor tables of NT entries (half 0, half 1)
  int pair_stride;       // workgroup ids of a pair are this far apart (8: same XCD as observed; 1: neighbours)
  float* xbuf;           // hand-off blocks [2 parities][row blocks][2 halves][32 x 256] fp32
  unsigned* xflags;      // [0..63] diagnostics (bit 0 of word 0: a poll timed out); from word 64: one 128-byte line per (row block, half)
  int wf32;              // 1: `w` holds fp32 FRAGMENT tiles, every projection / convolution product is an exact fp32 MFMA (MDT_F_WF32)
  int rb_base;           // pair-split launches: first row block of THIS launch (launch_tf256 chunks a batch that does not fit the device)
};
bool tf128_supported(int T, int Tk, int nvec, bool cross);
hipError_t launch_tf128(const TFArgs& a, hipStream_t s);
hipError_t launch_tf128_f32(const TFArgs& a, hipStream_t s);    // the exact-fp32 instantiations (k_tf128_f32.hip); reached through launch_tf128
// MDT_OP_TF256 (k_tf256.hip): the same for a C = 256 level, 32-row workgroups.  Differences in the streams: tile
// descriptors are kind (3 bits: 0 projection sub-tile, 1 output sub-tile, 2 K rows, 3 V rows, 4 scratch, 5 scratch + the next
// sub-block's vectors) | aux << 3; every sub-block is followed by two scratch tiles; vectors are 768 floats per sub-block
// ([bq 512 | bo 256], [b1 512 | b2 256], to_in: [bias 256]); npost = 8 sub-tiles.
bool tf256_supported(int T, int Tk, int nheads, int nff, bool cross);
hipError_t launch_tf256(const TFArgs& a, hipStream_t s);
hipError_t launch_tf256_f32(const TFArgs& a, hipStream_t s);
// MDT_OP_RES256 (k_res256.hip): a chain of ResnetBlock1d blocks of a 256-channel level in one launch; TFArgs fields as the ResNet
// part of MDT_OP_TF128 (res_kind, n_res, skip, skip_stride, skip_scale, film, eps_res) with npost = taps of the block convolutions
bool res256_supported(int T, int kind, int n_res, int taps);
hipError_t launch_res256(const TFArgs& a, hipStream_t s);
int tf256_pair_capacity();              // workgroups of a pair-split launch resident at once on the current device
extern int g_pair_capacity_override;    // tests: > 0 replaces the device's capacity    // (k_tf256_f32.hip); reached through launch_tf256

hipError_t launch_concat(const float* a, const float* b, float* out, int64_t rows, int ca, int cb, float scale_b,
                         hipStream_t s);
hipError_t launch_patch(const float* in, float* out, int batch, int rows_in, int c_in, int ld_in, int ld_out,
                        int patch, int inverse, hipStream_t s);
hipError_t launch_time_embed(const float* cn, const float* w, float* out, int rows, int half, int ld,
                             hipStream_t s);


#if defined(__HIPCC__)
// Coalesced epilogue shared by the GEMM kernels.  The MFMA accumulator layout gives every lane ONE column
// of 16 rows, i.e. 4-byte stores scattered over rows; measured on MI355X those stores cost more than the
// whole rest of the kernel (M=16384 N=1024 K=128: 68.9 us with them, 28.8 us without).  So the accumulator
// tile is first parked in LDS as fp32 [BM][BN+4] and then written by the whole workgroup as 16-byte stores,
// 16 consecutive lanes per 256-byte row segment; bias, exact GELU and the residual are applied on the way
// (bias/residual become float4 loads too).
// The residual may alias the output (in-place accumulation), so a load after a store stays after it: U row segments
// are handled per pass with all their bias / residual loads requested before the first store (one round trip per pass
// instead of one per segment), and the stores are streaming (nothing left dirty in L2 for the kernel-end write-back).
template <int BM, int BN>
__device__ __forceinline__ void store_tile_coalesced(const float* Cs, const GemmArgs& g, int m0, int n0) {
  constexpr int C4 = BN / 4, LDC = BN + 4, U = 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (int base = threadIdx.x; base < BM * C4; base += 256 * U) {
    float4 v[U], b[U], r[U];
    int64_t oidx[U], ridx[U];
    int ncol[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + 256 * u;
      const int row = idx / C4, c4 = idx - row * C4;
      const int m = m0 + row, n = n0 + c4 * 4;
      ok[u] = idx < BM * C4 && m < g.M && n < g.N;
      const int mm = ok[u] ? m : m0, nn = ok[u] ? n : n0;                      // clamped: loads stay in bounds
      const int bb = mm / g.r_out;
      const int64_t orow = (int64_t)bb * g.o_rows + (int64_t)(mm - bb * g.r_out) * g.o_stride + g.o_off;
      oidx[u] = orow * g.ldc + g.o_col + nn;
      ridx[u] = orow * g.ldr + nn;
      ncol[u] = nn;
      v[u] = ok[u] ? *reinterpret_cast<const float4*>(Cs + row * LDC + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (g.bias) {                                    // one block per kind of load: one round trip each
#pragma unroll
      for (int u = 0; u < U; ++u) b[u] = *reinterpret_cast<const float4*>(g.bias + ncol[u]);
    }
    if (g.res) {
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const float4*>(g.res + ridx[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 w = v[u];
      if (g.bias) { w.x += b[u].x; w.y += b[u].y; w.z += b[u].z; w.w += b[u].w; }
      if (g.act == 1) {
        w.x = 0.5f * w.x * (1.0f + erff(w.x * 0.70710678118654752440f));
        w.y = 0.5f * w.y * (1.0f + erff(w.y * 0.70710678118654752440f));
        w.z = 0.5f * w.z * (1.0f + erff(w.z * 0.70710678118654752440f));
        w.w = 0.5f * w.w * (1.0f + erff(w.w * 0.70710678118654752440f));
      }
      if (g.res) { w.x += r[u].x; w.y += r[u].y; w.z += r[u].z; w.w += r[u].w; }
      if (ok[u]) __builtin_nontemporal_store(f4{w.x, w.y, w.z, w.w}, reinterpret_cast<f4*>(g.out + oidx[u]));
    }
  }
}
#endif

}  // namespace mdt
