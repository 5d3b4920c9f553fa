/*
 * mdt_hip.h -- C ABI of libmdt_hip.so, the MI355X (gfx950) kernels behind
 * QMDiffusion.sample / QMDiffusionForward.sample.
 *
 * The reference (lamm-mit/MoleculeDiffusionTransformer) is pure Python/PyTorch and has
 * no FFI of its own; the drop-in surface is the Python class pair QMDiffusion /
 * QMDiffusionForward (generative.py:718-914, :31-225).  This header is the native
 * boundary underneath that surface: plain device pointers, sizes and a hipStream_t
 * (passed as void*), no torch types.  Every entry point names the reference code
 * it replaces.  All pointers are DEVICE pointers to fp32 unless stated otherwise;
 * every call only enqueues work on `stream` (no allocation, no synchronisation), so
 * calls may be captured into a HIP graph.  Return value: 0 on success, non-zero on
 * error (message via mdt_last_error()).
 *
 * Layouts
 *   sampler state / noise / result : (B, C, L)   channel-major, as the reference
 *   U-Net activations              : (B, L, Cp)  token-major, Cp = C padded to 16
 *   context embedding              : (B, n, F)   as the reference
 */
#ifndef MDT_HIP_H
#define MDT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDT_ABI_VERSION 5

/* MDT_ABI_VERSION; a library built with -DMDT_TUNING (timing-only switches that may give WRONG results, csrc/mdt_kernels.h) sets
 * bit 30 on top -- the Python binding refuses such a library for sampling. */
int mdt_abi_version(void);
const char *mdt_last_error(void);
/* Process-wide tuning / test hooks (NOT part of the stable ABI; not needed for correct results): "pair_stride" = v > 0 places the two workgroups of
 * every pair-split MDT_OP_TF256 v workgroup ids apart whatever the op says (1: neighbours, i.e. different XCDs; 0: back to
 * the ops' own value); "tile16" = 0 | 1 | 2 forces a tile of the bf16 x bf16 GEMM (-1: automatic); "w16" = 0 | 1 turns that GEMM's all-bf16
 * epilogue (bf16 output, bf16 residual or none: 8 columns per lane, the residual requested under the main loop) off | on; "pair_capacity": see
 * mdt_pair_capacity.  Returns 0, or 1 for an unknown key. */
int mdt_set_tuning(const char *key, int32_t value);
/* Workgroups of a pair-split MDT_OP_TF256 launch that the current device keeps resident at the same time: compute units x
 * workgroups per compute unit (hipDeviceGetAttribute / hipOccupancyMaxActiveBlocksPerMultiprocessor; 256 on an MI355X), 0 if
 * unknown.  The two workgroups of a pair wait for each other inside the launch, so mdt_program_run never puts more than this
 * many into one launch: a larger batch runs as several launches over consecutive row-block ranges (same results).  The host
 * reads it to choose between the pair-split and the whole-workgroup form by batch size (generative.py::_wide).
 * mdt_set_tuning("pair_capacity", v > 0) replaces it (tests: force chunking on a small batch; 0 = back to the device's); the
 * override is a TEST HOOK: it is refused unless the process has MDT_TEST_HOOKS=1 in its environment. */
int32_t mdt_pair_capacity(void);
/* Test hook: enqueue a kernel of n_workgroups workgroups that only HOLD compute units -- each allocates lds_bytes of LDS (<= 160
 * KiB: nothing else fits next to it) and spins for `ticks` of the 100 MHz real-time counter.  Used to break the co-residency
 * of pair-split launches on purpose (another stream holding most of the device) and check that the failure is reported.
 * NOT part of the stable ABI: refused unless the process has MDT_TEST_HOOKS=1 in its environment; at most 4096 workgroups and
 * 3 s (3e8 ticks). */
int mdt_test_occupy(int32_t n_workgroups, int32_t lds_bytes, uint64_t ticks, void *stream);

/* ------------------------------------------------------------------ */
/* U-Net evaluation as an op program                                   */
/* ------------------------------------------------------------------ */
/* The network (UNet1d.forward, modules.py:1144-1180) is lowered by the
 * Python host (moleculediffusiontransformer_amd/compiler.py) into a flat
 * list of fused ops over packed weights; mdt_program_run enqueues them.  */

enum mdt_space {
  MDT_SP_NONE = 0,
  MDT_SP_WEIGHT = 1, /* off = absolute float offset into the packed weight buffer            */
  MDT_SP_ACT = 2,    /* off = float offset PER SAMPLE; address = act + off * B               */
  MDT_SP_SHR = 3,    /* off = absolute float offset into the batch-invariant (shared) arena   */
  MDT_SP_EXT0 = 4    /* MDT_SP_EXT0 + i : off floats into bindings.ext[i]                      */
};
#define MDT_N_EXT 8

typedef struct mdt_ref {
  int32_t space;
  int32_t reserved;
  int64_t off;
} mdt_ref;

enum mdt_op_kind {
  MDT_OP_GEMM = 1,     /* nn.Linear / nn.Conv1d / nn.ConvTranspose1d phase as implicit GEMM with
                          fused norm-apply/FiLM/SiLU prologue and bias/GELU/residual epilogue
                          (modules.py:114-122, :135, :188, :317-319, :386-391, :486-516, :40-81) */
  MDT_OP_GN_STATS = 2, /* nn.GroupNorm statistics (modules.py:99-103, :485)                     */
  MDT_OP_ATTN = 3,     /* AttentionBase.forward core: softmax(q k^T * scale) v (modules.py:350-363) */
  MDT_OP_CONCAT = 4,   /* UpsampleBlock1d.add_skip: cat([x, skip * s], channel) (modules.py:828-829) */
  MDT_OP_PATCH = 5,    /* Patcher / Unpatcher rearrange (modules.py:230, :255)                   */
  MDT_OP_TIME_EMBED = 6, /* LearnedPositionalEmbedding.forward (modules.py:554-559)              */
  MDT_OP_GN_ACT = 8,   /* nn.GroupNorm + FiLM + SiLU applied in one pass: out = silu(gn(a) * (scale + 1) + shift)
                          (ConvBlock1d.forward before its convolution, modules.py:117-121); a -> out, p0 = gain,
                          p1 = bias, p3 = [scale | shift] or none; ints as MDT_OP_GN_STATS plus MDT_N_SILU        */
  MDT_OP_RCONV = 9,    /* row-stationary Conv1d (k = 1 | 3, C -> C channels, C in {128, 256}) with the ConvBlock1d prologue
                          computed in the kernel: out = bias + conv(silu(gn(s * a) * (scale + 1) + shift)) (+ res);
                          with a2 the input is cat([s * a, s2 * a2]) (2C channels, GroupNorm groups inside a half; the second
                          half's gain / bias / weight tiles follow the first's);
                          a, w = weight tiles, bias | none, out, res | none, p0 = gain, p1 = bias of the GroupNorm | none,
                          p3 = [scale | shift] | none (modules.py:117-121, :193-205)                             */
  MDT_OP_RESBLOCK = 10, /* a whole ResnetBlock1d with one GroupNorm group on a 64-token level (the U-Net's Patcher / Unpatcher,
                          modules.py:145-205, 208-257) in one launch, padded (cin, cout) = (16, 64) | (64, 16) | (16, 16), see MDT_K_CIN_REAL:
                          out = conv3(silu(gn(conv3(silu(gn(a))) + b1) (scale + 1) + shift)) + b2 + to_out(a);
                          a [B][64][cin], w = bf16 hi/lo MFMA fragments (conv1 | conv2 | to_out k-steps),
                          bias = gamma1 | beta1 | b1 | gamma2 | beta2 | (b2 + to_out bias), out [B][64][cout],
                          p3 = [scale | shift] | none                                                              */
  MDT_OP_TF128 = 11,   /* a whole Transformer1d (modules.py:469-524) of a 128-channel level in ONE launch: to_in (GroupNorm(32) +
                          Conv1d k=1), every TransformerBlock (self-attention, [cross-attention], feed-forward, :456-461) and
                          to_out (folded into the last feed-forward block); the residual stream never leaves the registers.
                          a = input [B][T][128], out = output, w = weight tile stream of all sub-blocks in consumption order
                          (projection tiles with their K columns permuted to the accumulator layout, see k_tf128.hip),
                          bias = every sub-block's vectors, p0 = tile descriptors (uint32 per tile: kind | aux << 2),
                          a2 = hoisted K|V rows of the FIRST cross-attention layer (layer l at + l * KV_LSTRIDE per-sample
                          floats), p1 = batch-invariant K|V rows for the second half of a dual batch; ints: enum mdt_tf128_i.
                          Optionally (MDT_F_RES_KIND) the level's ResnetBlock1d blocks (modules.py:145-205) run IN FRONT of the
                          transformer in the same launch (or alone, NBLOCKS = 0): res = the blocks' skip tensors (stored by kind 1,
                          read by kind 2), p3 = their FiLM rows                                                            */
  MDT_OP_TF256 = 12,   /* the same for a 256-channel level (32-row workgroups whose wave pairs split every chunk's features; the pair's
                          partial sums meet in scratch tiles of the ring).  Stream differences: 32 KB SUB-tiles (a [64][256]
                          projection tile = its two K halves, a [256][64] output tile = its two row halves), descriptors
                          kind (3 bits: 0 projection, 1 output, 2 K rows, 3 V rows, 4 scratch, 5 scratch + vectors of the next
                          sub-block) | aux << 3, two scratch descriptors after every sub-block; vectors: 768 floats per
                          sub-block ([bias 256] / [bq 512 | bo 256] / [b1 512 | b2 256]); NPOST = 8 sub-tiles; ints and
                          floats as MDT_OP_TF128 with C = 256.
                          MDT_F_NSPLIT = 2 (batches whose 32-row blocks do not fill the chip): every row block is served by a
                          PAIR of workgroups, half hh = 0 | 1 taking heads / hidden chunks / to_in output chunks / folded
                          to_out k chunks [hh n/2, (hh + 1) n/2); p0 then holds TWO descriptor tables of NT entries (half 0,
                          half 1; each sub-block's two scratch descriptors are followed by two more, kind 4, for the hand-off),
                          and the two workgroups hand each other their 32 x 256 partial sums inside the launch after every
                          sub-block: p3 = hand-off blocks, 2 * ceil(B T / 32) * 2 * 8192 floats; p2 = flag words (uint32):
                          64 diagnostic words (bit 0 of word 0 is raised if a poll timed out), then one 128-byte line per
                          (row block, half), counting hand-offs monotonically across launches -- zero them once, never between
                          launches.  The result does not depend on where the two workgroups run (k_tf256.hip)          */
  MDT_OP_PREP16 = 14,  /* A operand of a bf16 x bf16 GEMM: out (bf16 [B][R_IN][CIN], CIN / 2 floats per row) = bf16(prologue(a[.., A_COL +
                          c])); ints R_IN, LDA, CIN, A_COL, PRO, GROUPS, GSIZE, PRO_SILU and p0..p3 / eps as MDT_OP_GEMM;
                          WFMT = 2 (round 6): a is bf16 itself (LDA / A_COL in bf16 elements) -- LayerNorm (PRO 1) of the bf16
                          residual stream, statistics in fp32 on the exactly widened values; CIN <= 1024                     */
  MDT_OP_ATTN_CTX = 13, /* cross-attention core against the NORMALISED CONTEXT itself (K = V = c, shared by all heads and layers;
                          the per-layer key / value projections are folded into the query / output projections by the host):
                          a = q' [B][T * heads][128] (rows (token, head)), a2 = c [B | 1][Tk <= 64][LDKV], out [B][T * heads][128]
                          = softmax(q' c^T * scale) c; ints as MDT_OP_ATTN (LDQ / LDO unused)                              */
  MDT_OP_RES256 = 15,  /* a CHAIN of ResnetBlock1d blocks (modules.py:145-205) of a 256-channel level in ONE launch, 32-row workgroups that
                          keep their rows in registers from the first block to the last (csrc/k_res256.hip): kind 1: x = Block(x)
                          N_RES times, every block's output also stored as a skip tensor (DownsampleBlock1d / BottleneckBlock1d,
                          modules.py:680-721, :865-928); kind 2: x = Block(cat([x, s * skip[rb]])) (UpsampleBlock1d, :828-862).
                          a = x, out, res = skip tensors (addressed as MDT_OP_TF128's), w = weight sub-tile stream, bias = the blocks'
                          vectors, p0 = tile descriptors, p3 = the blocks' FiLM rows; ints: enum mdt_tf128_i with C = 256, the ResNet
                          fields, NT, NVEC = N_RES x (6 C | 9 C), and NPOST = taps of the block convolutions (3, or 1 when T = 1: only
                          the centre tap of a k = 3 convolution sees data); RES_PAIR1 / RES_PAIR2 are implied (GroupNorm groups of
                          64 channels on a 2C-channel input, 32 otherwise).  Stream: sub-tiles [64 features][128 k] in (tap, K half,
                          chunk) order per convolution; rows 0..31 of chunk c are output channels 32 c .., rows 32..63 channels
                          128 + 32 c ..; K columns in accumulator order (k-slot 32 st + 8 g + e = channel 16 (2 st + (e >> 2)) + 4 g +
                          (e & 3)).  Descriptors (HEADS of them, NT tiles in all) are SEGMENTS: kind (2 bits: 0 = a RUN of aux >= 1 consecutive sub-tiles
                          of the weight stream, which is stored in consumption order (aux = 0 is read as 1; the runs and single tiles sum to NT), 1 skip rows of block aux, 2 scratch, 3 scratch + the
                          vectors of block aux) | aux << 2; tile sequence: [3 (block 0)], then per block, kind 1: X nt X nt, kind 2: X nt X
                          8 S X 8 S X nt X nt (block1 on x, to_out on x, to_out on the skip, block1 on the skip, block2; X = 2, or 3 with aux = next block at a block's first X; nt = 8 taps; S = 1).  Vectors
                          per block: kind 1 [g1 | b1 | bias1 | g2 | b2 | bias2], kind 2 [g1 (2C) | b1 (2C) | bias1 | bias_to_out | g2 |
                          b2 | bias2].  WF32 as MDT_OP_TF256 (fp32 fragment sub-tiles).
                          NSPLIT = 2 (round 6; batches whose 32-row blocks do not fill the chip): a PAIR of workgroups per row block as
                          MDT_OP_TF256's pair split; a2 = the hand-off flags, p1 = the hand-off blocks (the SAME buffers as the
                          MDT_OP_TF256 ops of the level: shared flag lines, 32 KB blocks of which this op uses 16 KB), PAIR_STRIDE
                          as there, NFF = weight sub-tiles of ONE half.  Half hh streams output chunks 2 hh, 2 hh + 1 of every
                          convolution: sub-tiles in (tap, K half, chunk of the half) order, half 1's NFF sub-tiles behind half
                          0's; ONE descriptor table (the runs of both halves are equal, the tile sequence is the unsplit one with
                          half the weight runs); the pair's hand-offs use words 4..7 of the flag lines (one per compute wave; word 0
                          is MDT_OP_TF256's).  The result does not depend on where the two workgroups run.                     */
  MDT_OP_TBLOCK = 7    /* fused transformer sub-block, in place on x (TransformerBlock.forward, modules.py:456-461):
                          x += Attention(x) | x += Attention(x, context) | x += FeedForward(x); LayerNorm affine
                          folded into the projection weights, q/k/v/probabilities/hidden never leave registers */
};

/* prologue applied to the A operand of MDT_OP_GEMM while it is staged into LDS */
enum mdt_prologue {
  MDT_PRO_NONE = 0,
  MDT_PRO_LAYERNORM = 1, /* nn.LayerNorm over the K features of each row; p0 = gain, p1 = bias   */
  MDT_PRO_GROUPNORM = 2, /* normalise with stats from MDT_OP_GN_STATS (p2), p0 = gain, p1 = bias;
                            optional FiLM x*(scale+1)+shift (p3 = [scale(C) | shift(C)]); optional SiLU */
  MDT_PRO_SILU = 3       /* plain SiLU (MappingToScaleShift, modules.py:133-136)                 */
};

/* integer / float parameter slots of mdt_op, per kind */
/* common to the ring kernels' ops (MDT_OP_TBLOCK / RCONV / TF128 / TF256): size of the op's weight stream (w) in KB, 0 = unknown.
   The launch BEFORE such an op pulls that stream into the L2s while it finishes (its loader waves are idle by then). */
enum { MDT_W_KB = 23 };
enum mdt_gemm_i {
  MDT_G_R_OUT = 0,   /* M rows per sample of this GEMM                                           */
  MDT_G_R_IN = 1,    /* rows per sample of the A tensor                                           */
  MDT_G_LDA = 2,     /* floats per A row                                                          */
  MDT_G_CIN = 3,     /* channels consumed per tap (multiple of 16); K = taps * cin                */
  MDT_G_TAPS = 4,
  MDT_G_T_STRIDE = 5, /* source row of tap j for output row r: r*stride + j*dj + off; rows outside */
  MDT_G_T_DJ = 6,     /* [0, R_IN) read as zero AFTER the prologue (conv zero padding)             */
  MDT_G_T_OFF = 7,
  MDT_G_N = 8,       /* output features (multiple of 16)                                          */
  MDT_G_LDC = 9,     /* floats per output row                                                     */
  MDT_G_O_ROWS = 10, /* rows per sample of the output tensor                                      */
  MDT_G_O_STRIDE = 11, /* output row of GEMM row r: r*o_stride + o_off                             */
  MDT_G_O_OFF = 12,
  MDT_G_LDR = 13,    /* floats per residual row (residual uses the output row mapping)            */
  MDT_G_PRO = 14,    /* enum mdt_prologue                                                         */
  MDT_G_GROUPS = 15, /* GroupNorm groups                                                          */
  MDT_G_GSIZE = 16,  /* channels per group                                                        */
  MDT_G_PRO_SILU = 17, /* GROUPNORM prologue: apply SiLU after norm/FiLM                           */
  MDT_G_ACT = 18,    /* epilogue: 0 none, 1 exact-erf GELU                                         */
  MDT_G_M_MODE = 19, /* 0: M = B * R_OUT; 1: M = n_shared_rows * R_OUT; 2: M = R_OUT               */
  MDT_G_A_COL = 20,  /* first channel of the A row to consume                                     */
  MDT_G_O_COL = 21,  /* first column of the output row to write                                   */
  MDT_G_PHASES = 22, /* f > 1: ConvTranspose1d(k = 2f, stride f, padding f/2) as f output phases of 2 taps in ONE
                        launch (modules.py:74-81): phase ph uses weights [ph][N][K], T_OFF = (ph < f/2),
                        O_OFF = f * T_OFF + ph - f/2, O_STRIDE = f; 0 / 1: a plain GEMM                 */
  MDT_G_WFMT = 23    /* weight format: 0 = fp32 [N][K] (exact fp32 MFMA), or with a2 bound the hi (w) / lo (a2) bf16 planes of the
                        split product; 1 = ONE bf16 plane [N][K] in w: plain bf16 products (A rounded to bf16 after the prologue,
                        fp32 accumulation) -- the reduced-precision mode of the deep-UNet configuration;
                        2 = as 1 with the A operand ALREADY bf16 (written by MDT_OP_PREP16; LDA / A_COL in bf16 elements): both
                        operands stream through LDS-DMA; no prologue, stride, phases or output row mapping, cin % 64 == 0;
                        6 = as 2 and the OUTPUT is bf16 too (LDC / O_COL in bf16 elements): a tensor whose only reader is the next
                        bf16 x bf16 GEMM (feed-forward hidden layer); 10 = as 2 and ALSO a bf16 copy [rows][N] of the fp32 output
                        into p0 (the residual stream as the A operand of the next GEMM, written where it is produced);
                        38 (round 6) = as 6 and the RESIDUAL is bf16 too (LDR in bf16 elements; res may alias out): the bf16
                        residual stream of the plain-bf16 mode's transformer blocks -- one bf16 tensor is residual, output and the
                        next GEMM's A operand.
                        134 (round 6) = 6 with the LayerNorm of its A rows FOLDED into the GEMM: A is a raw bf16 row (the residual
                        stream), out = rstd (A W^T - mean colsum) + bias with the rows' mean / rstd over CIN gathered by the kernel
                        from the A fragments it multiplies, p0 = colsum [N] = sum_k W[n][k] of the bf16 weights (the host folds
                        the LayerNorm's gain into W and W bias_ln into bias), eps = f[0]; one tap, no residual, N / LDC / O_COL
                        multiples of 8.
                        Formats 2 / 6 / 10 end in a float4 epilogue: N, LDC, O_COL and LDR must be multiples of 4 and bias /
                        residual / out 16-byte aligned (bf16 out / copy: 8), otherwise the op is rejected;
                        16 = RING TILES (k_proj.hip, round 5): w holds N / 64 * CIN / 128 tiles of 32 KB, tile (chunk c, K half h)
                        at index c * (CIN / 128) + h = W[64 c .. 64 c + 64)[128 h .. 128 h + 128) as a bf16 hi plane [64][128]
                        followed by the lo plane (split-bf16 products); CIN in {128, 256}, N % 64 == 0, N <= 2048, prologue none or
                        LayerNorm (p0 / p1 both unbound = LayerNorm WITHOUT affine: the caller folded gain into the weights and
                        bias into the bias), one tap, bias / residual optional (the residual may alias out), no activation / row mapping;
                        17 = as 16 with fp32 FRAGMENT tiles (the layout of MDT_F_WF32) and exact fp32 MFMA products */
};
enum mdt_gemm_f { MDT_GF_EPS = 0 };

enum mdt_gn_i { MDT_N_ROWS = 0, MDT_N_LD = 1, MDT_N_GROUPS = 2, MDT_N_GSIZE = 3, MDT_N_SILU = 4,
                MDT_N_OUT16 = 5, /* MDT_OP_GN_ACT: 1 = out is bf16 [rows][ld] (A operand of a bf16 x bf16 GEMM) */
                MDT_N_CA = 6     /* MDT_OP_GN_ACT with a2 bound (round 6): the input is cat([a (CA channels, pitch CA), SCALE2 * a2 (LD - CA
                                    channels, pitch LD - CA)]) without the concatenated tensor; CA a multiple of GSIZE.  p2 (optional,
                                    with or without a2): also a raw bf16 copy [rows][LD] of that input */ };
enum mdt_gn_f { MDT_NF_EPS = 0, MDT_NF_SCALE2 = 1 };

enum mdt_rconv_i { MDT_R_T = 0, MDT_R_C = 1, MDT_R_LDA = 2, MDT_R_LDC = 3, MDT_R_LDR = 4, MDT_R_TAPS = 5,
                   MDT_R_GSIZE = 6 /* channels per GroupNorm group, 0 = no normalisation */, MDT_R_SILU = 7,
                   MDT_R_FILM_LD = 8 /* floats between the scale and the shift row */,
                   MDT_R_LDA2 = 9 /* floats per row of the second source (a2), if any */,
                   MDT_R_WF32 = 10 /* 1: w = fp32 fragment tiles, exact fp32 MFMA products (see MDT_F_WF32) */,
                   MDT_R_KSRC = 11 /* > 1 (round 5): a is [rows][KSRC * C] and its KSRC blocks of C channels accumulate into the C outputs,
                                      i.e. out = a W^T + bias (+ res) with K = KSRC * C == 1024 (tiles in the order source block /
                                      K half / feature chunk); one tap, no GroupNorm / FiLM / second source.  KSRC = 2 (C = 256): the two
                                      256-channel blocks of one tensor with 1 or 3 taps -- a strided Conv1d(k = 2 f + 1, stride f) in
                                      PATCH form: f consecutive tokens are one row of f * channels values and the convolution is k = 3
                                      over those rows */,
                   MDT_R_HALF_OUT = 12 /* 1 (C = 256, one source, no prologue): only output channels 0 .. 127 exist (LDC >= 128); the
                                      tile stream keeps the layout of all four 64-feature chunks, chunks 2 / 3 are never read */,
                   MDT_R_NB = 13 /* > 1 (one source, no prologue): NB * C output channels = NB convolutions of the same rows in one launch
                                      (tile streams, bias, residual and output columns of the blocks follow each other): a
                                      ConvTranspose1d(k = 2 f, stride f) in patch form writes f tokens x channels per input token */ };
enum mdt_rconv_f { MDT_RF_EPS = 0, MDT_RF_IN_SCALE = 1 /* factor on the rows of a -- with KSRC > 1 on EVERY K block of a */,
                   MDT_RF_IN_SCALE2 = 2 /* factor on the rows of a2 */ };
/* The KSRC > 1, HALF_OUT and NB > 1 forms move rows in 16-byte pieces: LDA, LDC and LDR (when res is bound) must be multiples of 4
   floats; HALF_OUT needs LDC (and LDR) >= 128, NB needs LDC (and LDR) >= NB * C.  mdt_program_create refuses anything else.      */
enum mdt_resblock_i { MDT_K_T = 0, MDT_K_CIN = 1, MDT_K_COUT = 2, MDT_K_FILM_LD = 3,
                      MDT_K_WF32 = 4 /* 1: w = fp32 MFMA fragments [step][row tile][half lo][64 lanes][4] (lane (i, g) float r =
                                        W[16 rt + i][the step's pair 8 g + 4 lo + r]), exact fp32 MFMA products */,
                      MDT_K_CIN_REAL = 5, MDT_K_COUT_REAL = 6 /* CIN / COUT are channel counts padded to 16; the GroupNorm
                                        statistics run over the first CIN_REAL / COUT_REAL channels (0 = all of them); padded
                                        gains, biases, weights -- hence outputs -- are zero */,
                      MDT_K_PATCH_IN = 7, MDT_K_PATCH_OUT = 8 /* p > 1 (round 5): the Patcher / Unpatcher rearrange folded into the
                                        block -- PATCH_IN: a is still patched, [T / p][CIN p] with a[l][c p + q] = x[l p + q][c];
                                        PATCH_OUT: out is written patched, [T / p][COUT p]; 0 / 1 = plain [T][C] */ };
enum mdt_resblock_f { MDT_KF_EPS = 0 };

enum mdt_attn_i {
  MDT_A_T = 0, MDT_A_TK = 1, MDT_A_HEADS = 2, MDT_A_LDQ = 3, MDT_A_LDKV = 4, MDT_A_LDO = 5,
  MDT_A_KV_BSTRIDE = 6, /* rows between consecutive samples' K/V (TK, or 0 for a batch-invariant context) */
  MDT_A_OUT16 = 7,      /* MDT_OP_ATTN: 1 = out is bf16 (LDO in bf16 elements), read by a bf16 x bf16 GEMM */
  MDT_A_QCOL = 8,       /* MDT_OP_ATTN: first float of q inside its rows (q | k | v projected by ONE GEMM into one tensor) */
  MDT_A_KCOL = 9,       /* ... and of k inside the a2 rows (v follows heads * 64 floats later)                          */
  MDT_A_SPLIT = 10,     /* MDT_OP_ATTN_CTX: 1 = the scores q' c^T as split-bf16 products (the default mode's arithmetic), 0 = exact fp32 */
  MDT_A_IN16 = 11       /* MDT_OP_ATTN: mask, 1 = q is bf16 (LDQ / QCOL in bf16 elements, multiples of 8), 2 = k | v are bf16 (LDKV / KCOL
                           likewise): the plain-bf16 mode's q | k | v GEMM writes bf16; widened exactly to fp32 in registers */
};
enum mdt_attn_f { MDT_AF_SCALE = 0 };

enum mdt_concat_i { MDT_C_ROWS = 0, MDT_C_CA = 1, MDT_C_CB = 2 };
enum mdt_concat_f { MDT_CF_SCALE_B = 0 };

enum mdt_patch_i { MDT_P_ROWS_IN = 0, MDT_P_C_IN = 1, MDT_P_LD_IN = 2, MDT_P_LD_OUT = 3, MDT_P_PATCH = 4,
                   MDT_P_INVERSE = 5 };

enum mdt_time_i { MDT_T_HALF = 0, MDT_T_LD = 1 };

/* MDT_OP_TBLOCK: a = x (in place), w = weight tile stream, bias = packed biases, a2 = hoisted K|V (cross) */
enum mdt_tblock_mode { MDT_TB_SELF = 0, MDT_TB_CROSS = 1, MDT_TB_FF = 2 };
enum mdt_tblock_i {
  MDT_B_MODE = 0, MDT_B_C = 1,      /* features (128 or 256)                                             */
  MDT_B_T = 2,                      /* tokens per sample (must divide 16)                                */
  MDT_B_NCHUNK = 3,                 /* heads (attention) or hidden/64 (feed-forward)                     */
  MDT_B_NBIAS = 4, MDT_B_TK = 5, MDT_B_KV_BSTRIDE = 6, MDT_B_LDKV = 7, MDT_B_HEADS = 8,
  MDT_B_KV2 = 11,                   /* cross blocks, 1: dual batch (both passes of classifier-free guidance in one launch,
                                       UNetCFG1d.forward, modules.py:1248-1253): the second half of the samples reads the
                                       batch-invariant K / V rows p1 (FixedEmbedding) instead of a2; the first half must be
                                       whole workgroups: (B / 2) % (64 / T) == 0 (C = 128), % (32 / T) (variants >= 2)  */
  MDT_B_POST = 10,                  /* feed-forward only, > 0: Transformer1d's closing Conv1d(k=1) folded in (modules.py:524):
                                       out = Wout (x + FF(x)) + bout; the W2 tiles hold Wout W2, POST = C/64 extra output
                                       tiles hold Wout (natural k order), the output bias holds Wout b2 + bout; x is left
                                       untouched                                                            */
  MDT_B_VARIANT = 9,                /* 0: 64-row workgroups, C = 128 (cross: <= 16 keys per 16 rows); 1: removed (round 3);
                                       2: 32-row workgroups, C = 256 (cross: <= 48 keys per 16 rows), weight
                                       stream packed as 128-wide sub-tiles (K halves / output-row halves);
                                       3: as 2, with the heads / hidden chunks of a row block split over two
                                       workgroups: out = scratch [2][B T][C] for their partial sums;
                                       4: chained form of 3 without the reduce launch: block input = a + res (res = the
                                       previous block's second partial | none), out = block output written by head group 0
                                       (never aliasing a), p2 = second head group's partial | none (one workgroup);  */
  MDT_B_WF32 = 12                   /* 1: w = fp32 FRAGMENT tiles (variant 0, round 5) / fp32 fragment SUB-tiles in the layout of MDT_OP_TF256
                                       (variants 2..4, round 6), exact fp32 MFMA products (see MDT_F_WF32)                      */
};
enum mdt_tblock_f { MDT_BF_EPS = 0, MDT_BF_SCALE = 1 };

/* MDT_OP_TF128 */
enum mdt_tf128_i {
  MDT_F_C = 0,           /* 128                                                                              */
  MDT_F_T = 1,           /* tokens per sample (divides 16)                                                    */
  MDT_F_NT = 2,          /* tiles in the stream (weight tiles + K / V tiles)                                  */
  MDT_F_NVEC = 3,        /* floats of vectors (multiple of 256, <= 8192): [to_in bias 128] then per block
                            [bq 64 heads | bo 128] (self), the same (cross), [b1 64 nff | b2 128] (feed-forward)  */
  MDT_F_TK = 4, MDT_F_KV_BSTRIDE = 5, MDT_F_LDKV = 6, MDT_F_HEADS = 7,
  MDT_F_HAS_IN = 8,      /* 1: starts with to_in (two projection tiles, GroupNorm gain / bias folded into them)  */
  MDT_F_NBLOCKS = 9 /* may be 0 (then HAS_IN = 0 too) when ResNet blocks are present */, MDT_F_NFF = 10 /* hidden / 64 */,
  MDT_F_NPOST = 11,      /* 2: to_out folded into the last feed-forward block (two extra output tiles), 0: none   */
  MDT_F_KV2 = 12,        /* 1: dual batch, as MDT_B_KV2 (first half = whole 64-row workgroups)                    */
  MDT_F_CROSS = 13,      /* 1: the blocks have a cross-attention sub-block                                        */
  MDT_F_KV_LSTRIDE = 14, /* per-sample floats between consecutive cross layers' K|V rows                         */
  /* MDT_OP_TF128 only: ResnetBlock1d blocks (modules.py:145-205) of the level IN FRONT of the transformer, same launch */
  MDT_F_RES_KIND = 15,   /* 0 none; 1: x = Block(x), N_RES times, every block's output ALSO stored as a skip tensor: block rb to
                            res + rb * (B * T * C floats) (the up path reads them back); 2: x = Block(cat([x, s * skip[rb]])) with
                            skip[rb] = res - rb * (B * T * C floats) (consumed in reverse order of production), s = f[SKIP_SCALE].
                            Stream per block, kind 1: 6 + 6 projection tiles (convolution taps x output halves, K columns in
                            accumulator order); kind 2: 2 to_out(x), [skip rows: descriptor 0 | (1 << 20 | rb) << 2], 2 to_out(skip),
                            6 block1(x), [skip rows], 6 block1(skip), 6 block2.  Vectors per block ahead of the transformer's: kind 1
                            [g1 | b1 | bias1 | g2 | b2 | bias2] (6 C), kind 2 [g1 (2C) | b1 (2C) | bias1 | bias_to_out | g2 | b2 | bias2]
                            (9 C).  p3 = the blocks' FiLM rows [scale C | shift C] each, contiguous (NFILM floats)             */
  MDT_F_N_RES = 16,
  MDT_F_RES_PAIR1 = 17,  /* GroupNorm groups of block1: 1 = 32 channels, 0 = 16 channels (block2: RES_PAIR2)                */
  MDT_F_RES_PAIR2 = 18,
  MDT_F_NFILM = 19,      /* FiLM floats staged behind the vectors (2 C per block, padded to a multiple of 256; NVEC + NFILM <= 8192) */
  /* MDT_OP_TF256 only */
  MDT_F_NSPLIT = 20,     /* 0 / 1: one workgroup per 32-row block; 2: a pair of workgroups per block (see MDT_OP_TF256)      */
  MDT_F_PAIR_STRIDE = 21,/* NSPLIT = 2: workgroup ids of a pair are this far apart (0 = 8: one XCD under the observed round-robin
                            placement; 1 = neighbours on different XCDs) -- speed only                                       */
  /* MDT_OP_TF128 and MDT_OP_TF256 */
  MDT_F_WF32 = 22        /* product type of every projection / convolution of the launch.  0: split-bf16 -- each 32 KB weight tile is a
                            bf16 hi plane then a lo plane, products hi*hi + hi*lo + lo*hi on bf16 MFMAs.  1: EXACT fp32 -- the same
                            tile holds the same weights as fp32 in MFMA-fragment order: for a tile of R x K weights (64 x 128
                            projection, 128 x 64 output) fragment (row tile rt, k-step st, half lo) is 1 KB at
                            ((rt * (K / 32) + st) * 2 + lo) * 1024 and float r of lane i + 16 g in it is
                            W[16 rt + i][k-slot 32 st + 8 g + 4 lo + r]; every product is a v_mfma_f32_16x16x4_f32 with fp32
                            accumulation, i.e. the reference's fp32 arithmetic (modules.py:314-320, :350-364, :386-391, :105-112) */
};
enum mdt_tf128_f { MDT_FF_EPS_LN = 0, MDT_FF_SCALE = 1, MDT_FF_EPS_GN = 2, MDT_FF_EPS_RES = 3, MDT_FF_SKIP_SCALE = 4 };

typedef struct mdt_op {
  int32_t kind;
  int32_t reserved;
  mdt_ref a;    /* GEMM A / GN input / ATTN q / CONCAT a / PATCH in / TIME c_noise values        */
  mdt_ref a2;   /* ATTN k (v = k + heads*64 floats) / CONCAT b / GEMM: bf16 lo plane of split weights:
                   when set, w is the bf16 hi plane [N][K] and the product is formed as hi*hi + hi*lo + lo*hi
                   on bf16 MFMAs with fp32 accumulation (needs cin % 32 == 0); unset = exact fp32 MFMA */
  mdt_ref w;    /* GEMM weights [N][K], K contiguous / TIME fourier weights                       */
  mdt_ref bias; /* GEMM bias [N]                                                                  */
  mdt_ref out;
  mdt_ref res;  /* GEMM residual                                                                  */
  mdt_ref p0, p1, p2, p3;
  int32_t i[24];
  float f[8];
} mdt_op;

typedef struct mdt_bindings {
  const float *weights;
  float *act; /* per-sample arena, act_floats_per_sample * B floats                               */
  float *shr; /* batch-invariant arena                                                            */
  float *ext[MDT_N_EXT];
} mdt_bindings;

typedef struct mdt_program mdt_program;

/* Validates and copies `n_ops` ops.  Returns NULL on error. */
mdt_program *mdt_program_create(const mdt_op *ops, int32_t n_ops);
void mdt_program_destroy(mdt_program *p);
int32_t mdt_program_num_ops(const mdt_program *p);
/* Enqueue ops [first, first+count) for batch size B (count < 0: to the end). */
int mdt_program_run(const mdt_program *p, const mdt_bindings *b, int32_t B, int32_t n_shared_rows,
                    int32_t first, int32_t count, void *stream);

/* ------------------------------------------------------------------ */
/* conditioning prelude                                                */
/* ------------------------------------------------------------------ */
/* generative.py:838-850 (QMDiffusion.sample) / :149-161 (QMDiffusionForward.sample):
 *   e[b,i,0:D1]      = gelu(fc1_w[d] * seq[b,i] + fc1_b[d])
 *   e[b,i,D1:D1+D2]  = PositionalEncoding1D: [sin(i*f_j) | cos(i*f_j)], f = inv_freq (D2/2 values)
 * (transformer.py:3456-3470).  out is (B, n, D1+D2). */
int mdt_cond_embed(const float *seq, const float *fc1_w, const float *fc1_b, const float *inv_freq,
                   float *out, int32_t B, int32_t n, int32_t D1, int32_t D2, void *stream);
/* The pos_emb_fourier_add form of the same prelude (generative.py:844-846, graphmodel.py:338-339): the positional
 * encoding is ADDED, e[b,i,d] = gelu(fc1_w[d] * seq[b,i] + fc1_b[d]) + PositionalEncoding1D(D2)[i,d] for d < D1 <= D2: the
 * reference's encoding returns its first `orig_ch` = D1 columns of [sin (D2/2 frequencies) | cos (D2/2)] (transformer.py:3456-3470),
 * so text_embed_dim may be smaller than embed_dim_position; out is (B, n, D1), inv_freq holds D2/2 frequencies. */
int mdt_cond_embed_add(const float *seq, const float *fc1_w, const float *fc1_b, const float *inv_freq,
                       float *out, int32_t B, int32_t n, int32_t D1, int32_t D2, void *stream);

/* ------------------------------------------------------------------ */
/* k-diffusion preconditioning + ADPM2 sampler update                   */
/* ------------------------------------------------------------------ */
/* KDiffusion_mod.denoise_fn input scaling (diffusion.py:810): xin[b,l,c] = c_in * x[b,c,l],
 * written token-major with channels padded to Cp (pad = 0). */
int mdt_precond_in(const float *x, float *xin, float c_in, int32_t B, int32_t C, int32_t L, int32_t Cp,
                   void *stream);
/* KDiffusion_mod.denoise_fn output (diffusion.py:811-814, clip :75-88):
 *   D = clip(c_skip * x + c_out * pred);   pred is token-major (B, L, Cp).
 * clip = clamp to [-1, 1] (dyn_scale == NULL: dynamic_threshold = 0.0, what every class of the reference passes), or dynamic
 * thresholding: dyn_scale[b] = max(quantile(|c_skip x + c_out pred|, q), 1) per sample (mdt_dyn_scale), D = clamp(., -s, s) / s.
 * The same optional argument on mdt_adpm2_mid / mdt_adpm2_next, whose denoise stage is this one.
 * The (L x Cp) tile of a sample is staged in LDS: L * (Cp + 1) * 4 <= 160 KiB (max_length = 1024 at Cp = 16: 68 KiB). */
int mdt_precond_out(const float *x, const float *pred, float *D, float c_skip, float c_out, int32_t B,
                    int32_t C, int32_t L, int32_t Cp, const float *dyn_scale, void *stream);
/* clip()'s dynamic threshold (diffusion.py:78-85): scale[b] = max(torch.quantile(|c_skip x[b] + c_out pred[b]|.flatten(), q), 1),
 * 0 < q <= 1, linear interpolation between the neighbouring order statistics as torch.quantile; C * L <= 32768. */
int mdt_dyn_scale(const float *x, const float *pred, float *scale, float c_skip, float c_out, float q, int32_t B,
                  int32_t C, int32_t L, int32_t Cp, void *stream);
/* UNetCFG1d.forward guidance mix (modules.py:1253): out = um + (cond - um) * scale, token-major. */
int mdt_cfg_mix(const float *cond, const float *uncond, float *out, float scale, int64_t n, void *stream);
/* First half of ADPM2Sampler.step (diffusion.py:506-508) fused with the denoise output:
 *   D = clamp(c_skip*x + c_out*pred, -1, 1); d = (x - D) / sigma; x_mid = x + d * dt_mid
 * and, for the next U-Net call, xin_mid = c_in_mid * x_mid (token-major, padded). */
int mdt_adpm2_mid(const float *x, const float *pred, float *x_mid, float *xin_mid, float c_skip,
                  float c_out, float sigma, float dt_mid, float c_in_mid, int32_t B, int32_t C, int32_t L,
                  int32_t Cp, const float *dyn_scale, void *stream);
/* Second half (diffusion.py:510-515):
 *   D = clamp(c_skip*x_mid + c_out*pred, -1, 1); d_mid = (x_mid - D) / sigma_mid;
 *   x = x + d_mid * dt_down;  x = x + noise * sigma_up   (in place on x)
 * and xin_next = c_in_next * x.  noise == NULL selects the counter-based generator:
 * Philox4x32-10 keyed by (seed, step), counter = global element index
 * (sample0 + b) * C * L + c * L + l, Box-Muller -- independent of how the batch is sharded.
 * tokens != NULL (only with xin_next == NULL, i.e. on the last update of a call): the decode step after the
 * path fused in, tokens[b,l] = argmax_c x[b,c,l] of the final x as int32 (generative.py:1212-1213, :1690-1691:
 * permute(0,2,1) -> argmax(dim=2); first maximum as torch.argmax). */
int mdt_adpm2_next(float *x, const float *x_mid, const float *pred, const float *noise, float *xin_next,
                   float c_skip, float c_out, float sigma_mid, float dt_down, float sigma_up,
                   float c_in_next, uint64_t seed, uint32_t step, int64_t sample0, int32_t B, int32_t C,
                   int32_t L, int32_t Cp, int32_t *tokens, const float *dyn_scale, void *stream);
/* One Euler move of ADPM2Sampler.step when the denoised tensor comes from a caller-supplied fn
 * (Sampler.forward(noise, fn, sigmas, num_steps) seam, diffusion.py:352, :502-515); all tensors (B, C, L):
 *   out = x_base + ((x_from - denoised) / sigma) * dt   [+ noise * sigma_up]
 * noise_mode 0: no noise term; 1: explicit `noise` tensor; 2: counter-based generator (seed, step, sample0). */
int mdt_adpm2_euler(const float *x_base, const float *x_from, const float *denoised, const float *noise,
                    float *out, float sigma, float dt, float sigma_up, int32_t noise_mode, uint64_t seed,
                    uint32_t step, int64_t sample0, int32_t B, int32_t C, int32_t L, void *stream);
/* x = sigma0 * noise (diffusion.py:520); noise == NULL: counter-based generator as above. */
int mdt_init_noise(float *x, const float *noise, float sigma0, uint64_t seed, uint32_t step,
                   int64_t sample0, int32_t B, int32_t C, int32_t L, void *stream);
/* dst[0..n) = src[0..n) on the caller's stream (non-overlapping device buffers): selects the (scale | shift) rows of ONE evaluation out
 * of the table the time program fills for every evaluation of a call at once -- MappingToScaleShift of all 18 ResnetBlock1d blocks,
 * modules.py:125-142, evaluated per U-Net call in the reference (:1166-1170), once per sampling call here (rows are identical across the
 * batch: sigma is a scalar, diffusion.py:91-102).  Round 6, an addition inside ABI version 5. */
int mdt_copy_f32(float *dst, const float *src, int64_t n, void *stream);
/* DiffusionSampler final clamp (diffusion.py:590). */
int mdt_clamp(float *x, float lo, float hi, int64_t n, void *stream);
/* Inpainting merge (diffusion.py:539-542, :549): out = where(mask, src + sigma*noise, x);
 * mask is uint8 (B,C,L); noise may be NULL when sigma == 0. */
int mdt_inpaint_merge(float *x, const float *src, const uint8_t *mask, const float *noise, float sigma,
                      uint64_t seed, uint32_t step, int64_t sample0, int32_t B, int32_t C, int32_t L,
                      void *stream);
/* x += s * noise (inpaint re-noise, diffusion.py:546-547). */
int mdt_add_noise(float *x, const float *noise, float s, uint64_t seed, uint32_t step, int64_t sample0,
                  int32_t B, int32_t C, int32_t L, void *stream);
/* Decode step after the path (generative.py:1212-1213): tokens[b,l] = argmax_c x[b,c,l] (int32). */
int mdt_argmax_tokens(const float *x, int32_t *tokens, int32_t B, int32_t C, int32_t L, void *stream);

/* ------------------------------------------------------------------ */
/* measurement helpers (HIP events on the caller's stream)             */
/* ------------------------------------------------------------------ */
typedef struct mdt_timer mdt_timer;
mdt_timer *mdt_timer_create(int32_t max_intervals);
void mdt_timer_destroy(mdt_timer *t);
int mdt_timer_start(mdt_timer *t, void *stream); /* records the start event of the next interval */
int mdt_timer_stop(mdt_timer *t, void *stream);  /* records its stop event                         */
/* Synchronises the stop events and returns the number of intervals; ms[i] = duration of interval i. */
int32_t mdt_timer_collect(mdt_timer *t, float *ms, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MDT_HIP_H */
