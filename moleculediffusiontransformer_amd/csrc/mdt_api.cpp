// C ABI of libmdt_hip.so: op-program executor, error reporting, HIP-event timers.
// Entry points are declared (with the reference code each one replaces) in include/mdt_hip.h.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mdt_hip.h"
#include "mdt_kernels.h"

namespace {
thread_local std::string g_err;

int fail(const std::string& m) {
  g_err = m;
  return 1;
}

int hip_fail(const char* what, hipError_t e) {
  return fail(std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace

// mdt_test_occupy: a kernel that does nothing but HOLD compute units -- every workgroup allocates `lds` bytes of LDS (so that
// nothing else fits next to it) and spins on the 100 MHz real-time counter for `ticks`.  Test infrastructure for the pair
// hand-off's failure path (tests/test_gpu_parity.py): with most of the device held by another stream, the partners of a
// pair-split launch are not resident together, their polls time out, and sample() must raise instead of returning garbage.
__global__ void k_test_occupy(unsigned long long ticks) {
  extern __shared__ unsigned char hog_lds[];
  if (threadIdx.x == 0) hog_lds[0] = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

struct mdt_program {
  std::vector<mdt_op> ops;
};

struct mdt_timer {
  std::vector<hipEvent_t> start, stop;
  int n = 0;
  bool open = false;
};

extern "C" {

static int g_pair_stride = 0;    // mdt_set_tuning("pair_stride", v): overrides MDT_F_PAIR_STRIDE of every pair-split op (tests)

// (library-internal: k_elem.hip reports its launch errors through it; hidden, so it is not part of the exported C ABI)
__attribute__((visibility("hidden"))) void mdt_set_error(const char* msg) { g_err = msg ? msg : ""; }
// Test hooks (mdt_test_occupy, the "pair_capacity" override) can stall the device or make pair-split launches refuse: they act
// only in a process that has MDT_TEST_HOOKS=1 in its environment (tests/conftest.py sets it), never for an ordinary ABI caller.
static bool test_hooks_enabled() {
  const char* e = getenv("MDT_TEST_HOOKS");
  return e && e[0] == '1';
}
int mdt_set_tuning(const char* key, int32_t value) {
  const std::string k = key ? key : "";
  if (k == "pair_stride") { g_pair_stride = value; return 0; }
  if (k == "tile16") { mdt::set_tile16(value); return 0; }
  if (k == "w16") { mdt::set_w16(value); return 0; }
  if (k == "pair_capacity") {
    if (value > 0 && !test_hooks_enabled()) return fail("mdt_set_tuning(pair_capacity): a test hook, needs MDT_TEST_HOOKS=1");
    mdt::g_pair_capacity_override = value > 0 ? value : 0;
    return 0;
  }
  g_err = "mdt_set_tuning: unknown key '" + k + "'";
  return 1;
}
const char* mdt_last_error(void) { return g_err.c_str(); }
int32_t mdt_pair_capacity(void) { return mdt::tf256_pair_capacity(); }
int mdt_test_occupy(int32_t n_workgroups, int32_t lds_bytes, uint64_t ticks, void* stream) {
  if (!test_hooks_enabled()) return fail("mdt_test_occupy: a test hook, needs MDT_TEST_HOOKS=1 in the environment");
  if (n_workgroups <= 0 || n_workgroups > 4096 || lds_bytes < 0 || lds_bytes > 160 * 1024) return fail("mdt_test_occupy: bad arguments");
  if (ticks > 300000000ull) return fail("mdt_test_occupy: at most 3 s (3e8 ticks of the 100 MHz counter)");
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_test_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k_test_occupy, dim3((unsigned)n_workgroups), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, (unsigned long long)ticks);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : hip_fail("mdt_test_occupy", e);
}
#ifdef MDT_TUNING
int mdt_abi_version(void) { return MDT_ABI_VERSION | MDT_ABI_TUNING_BIT; }   // a timing-only build (mdt_kernels.h)
#else
int mdt_abi_version(void) { return MDT_ABI_VERSION; }
#endif

// ------------------------------------------------------------------------------------------------
static const char* validate(const mdt_op& o, int idx, char* buf, size_t nbuf) {
  auto bad = [&](const char* why) {
    snprintf(buf, nbuf, "op %d (kind %d): %s", idx, o.kind, why);
    return buf;
  };
  auto space_ok = [](const mdt_ref& r) { return r.space >= 0 && r.space < MDT_SP_EXT0 + MDT_N_EXT; };
  for (const mdt_ref* r : {&o.a, &o.a2, &o.w, &o.bias, &o.out, &o.res, &o.p0, &o.p1, &o.p2, &o.p3})
    if (!space_ok(*r) || r->off < 0) return bad("bad operand reference");
  switch (o.kind) {
    case MDT_OP_GEMM: {
      const int32_t* i = o.i;
      if (i[MDT_G_CIN] <= 0 || i[MDT_G_CIN] % 16) return bad("cin must be a positive multiple of 16");
      if (i[MDT_G_N] <= 0 || i[MDT_G_TAPS] <= 0 || i[MDT_G_R_OUT] <= 0 || i[MDT_G_R_IN] <= 0) return bad("bad dims");
      if (i[MDT_G_LDA] % 4 || i[MDT_G_A_COL] % 4) return bad("A rows must be 16-byte aligned");
      if (i[MDT_G_LDA] < i[MDT_G_A_COL] + i[MDT_G_CIN]) return bad("lda < a_col + cin");
      if (!o.a.space || !o.w.space || !o.out.space) return bad("missing A / W / out");
      if (i[MDT_G_PRO] < 0 || i[MDT_G_PRO] > 3) return bad("bad prologue");
      if (i[MDT_G_PRO] == MDT_PRO_LAYERNORM) {
        if (i[MDT_G_TAPS] != 1) return bad("LayerNorm prologue needs taps == 1");
        if (i[MDT_G_CIN] > 2048) return bad("LayerNorm prologue supports <= 2048 features");
        if ((i[MDT_G_WFMT] == 16 || i[MDT_G_WFMT] == 17) ? (!o.p0.space != !o.p1.space) : (!o.p0.space || !o.p1.space))
          return bad("LayerNorm prologue needs gain and bias (ring-tile projection: both or neither -- neither = no affine)");
      }
      if (i[MDT_G_PRO] == MDT_PRO_GROUPNORM) {
        if (!o.p0.space || !o.p1.space || !o.p2.space) return bad("GroupNorm prologue needs gain, bias, stats");
        if (i[MDT_G_GROUPS] <= 0 || i[MDT_G_GSIZE] <= 0) return bad("bad groups");
      }
      if (i[MDT_G_M_MODE] < 0 || i[MDT_G_M_MODE] > 2) return bad("bad m_mode");
      if (o.res.space && i[MDT_G_LDR] <= 0) return bad("residual without ldr");
      if (o.a2.space && i[MDT_G_CIN] % 32) return bad("split-bf16 weights need cin % 32 == 0");
      if (i[MDT_G_WFMT] < 0 || (i[MDT_G_WFMT] > 2 && i[MDT_G_WFMT] != 6 && i[MDT_G_WFMT] != 10 && i[MDT_G_WFMT] != 16 && i[MDT_G_WFMT] != 17 &&
                                i[MDT_G_WFMT] != 38 && i[MDT_G_WFMT] != 134))
        return bad("bad weight format");
      if (i[MDT_G_WFMT] == 38 && !o.res.space) return bad("WFMT 38 (bf16 residual stream) needs the residual");
      if (i[MDT_G_WFMT] == 134) {      // LayerNorm of the raw bf16 A rows folded into the GEMM
        if (!o.p0.space || o.res.space || i[MDT_G_TAPS] != 1 || i[MDT_G_T_OFF] || i[MDT_G_N] % 8 || i[MDT_G_LDC] % 8 || i[MDT_G_O_COL] % 8)
          return bad("WFMT 134 (folded LayerNorm): p0 = column sums of W, one tap, no residual, N / ldc / o_col % 8 == 0");
      }
      if (i[MDT_G_WFMT] == 16 || i[MDT_G_WFMT] == 17) {
        if (!mdt::proj_supported(i[MDT_G_CIN], i[MDT_G_N], i[MDT_G_LDA], i[MDT_G_LDC], o.res.space ? i[MDT_G_LDR] : 0) || o.a2.space)
          return bad("ring-tile projection needs cin in {128, 256}, N % 64 == 0, 16-byte aligned rows and no lo plane (the tiles hold both)");
        if (i[MDT_G_PRO] > 1 || i[MDT_G_TAPS] != 1 || i[MDT_G_T_STRIDE] != 1 || i[MDT_G_T_OFF] || i[MDT_G_PHASES] > 1 || i[MDT_G_O_STRIDE] != 1 ||
            i[MDT_G_O_OFF] || i[MDT_G_R_OUT] != i[MDT_G_R_IN] || i[MDT_G_O_ROWS] != i[MDT_G_R_OUT] || i[MDT_G_ACT] || i[MDT_G_O_COL] % 4 ||
            i[MDT_G_A_COL] % 4)
          return bad("ring-tile projection: LayerNorm or no prologue, one tap, no stride / phases / output row mapping / activation, "
                     "A_COL and O_COL multiples of 4 (16-byte row pieces)");
      }
      if (i[MDT_G_WFMT] == 10 && (!o.p0.space || i[MDT_G_O_COL] || i[MDT_G_N] % 2)) return bad("bf16 copy needs its tensor (p0) and whole rows");
      if (i[MDT_G_WFMT] & 2) {
        if (!mdt::gemm_b16_supported(i[MDT_G_CIN], i[MDT_G_TAPS], i[MDT_G_LDA], i[MDT_G_A_COL]) || o.a2.space)
          return bad("bf16 x bf16 GEMM needs cin % 64 == 0 and 16-byte aligned bf16 rows");
        if (i[MDT_G_N] % 4 || i[MDT_G_LDC] % 4 || i[MDT_G_O_COL] % 4 || (o.res.space && i[MDT_G_LDR] % 4))
          return bad("bf16 x bf16 GEMM: N, ldc, o_col and ldr must be multiples of 4 (float4 epilogue)");
        if (i[MDT_G_PRO] || i[MDT_G_T_STRIDE] != 1 || i[MDT_G_PHASES] > 1 || i[MDT_G_O_STRIDE] != 1 || i[MDT_G_O_OFF] ||
            i[MDT_G_R_OUT] != i[MDT_G_R_IN] || i[MDT_G_O_ROWS] != i[MDT_G_R_OUT] || i[MDT_G_M_MODE])
          return bad("bf16 x bf16 GEMM: no prologue / stride / phases / output row mapping (use MDT_OP_PREP16 + WFMT 1 forms)");
      }
      if (i[MDT_G_WFMT] == 1 && (i[MDT_G_CIN] % 32 || o.a2.space)) return bad("bf16 weights need cin % 32 == 0 and no lo plane");
      break;
    }
    case MDT_OP_PREP16: {
      const int32_t* i = o.i;
      if (i[MDT_G_CIN] <= 0 || i[MDT_G_CIN] % 8 || i[MDT_G_R_IN] <= 0) return bad("bad dims");
      if (i[MDT_G_LDA] % 4 || i[MDT_G_A_COL] % 4 || i[MDT_G_LDA] < i[MDT_G_A_COL] + i[MDT_G_CIN]) return bad("bad A rows");
      if (i[MDT_G_PRO] < 0 || i[MDT_G_PRO] > 3) return bad("bad prologue");
      if (i[MDT_G_WFMT] != 0 && (i[MDT_G_WFMT] != 2 || i[MDT_G_PRO] != MDT_PRO_LAYERNORM || i[MDT_G_CIN] > 1024 || i[MDT_G_LDA] % 8 || i[MDT_G_A_COL] % 8))
        return bad("PREP16 with a bf16 input (WFMT 2): LayerNorm of rows of at most 1024 channels, LDA / A_COL multiples of 8 elements");
      if (!o.a.space || !o.out.space) return bad("missing operand");
      if (i[MDT_G_PRO] == MDT_PRO_LAYERNORM && (!o.p0.space || !o.p1.space)) return bad("LayerNorm needs gain and bias");
      if (i[MDT_G_PRO] == MDT_PRO_GROUPNORM &&
          (!o.p0.space || !o.p1.space || !o.p2.space || !o.p3.space || i[MDT_G_GROUPS] <= 0 || i[MDT_G_GSIZE] <= 0))
        return bad("GroupNorm needs gain, bias, stats, FiLM vector and groups");
      break;
    }
    case MDT_OP_GN_STATS:
      if (o.i[MDT_N_ROWS] <= 0 || o.i[MDT_N_GROUPS] <= 0 || o.i[MDT_N_GSIZE] <= 0) return bad("bad dims");
      if (!o.a.space || !o.out.space) return bad("missing operand");
      break;
    case MDT_OP_GN_ACT:
      if (!mdt::gn_act_eligible(o.i[MDT_N_ROWS], o.i[MDT_N_LD], o.i[MDT_N_GROUPS], o.i[MDT_N_GSIZE]))
        return bad("shape not supported by the fused GroupNorm-apply kernel");
      if (!o.a.space || !o.out.space || !o.p0.space || !o.p1.space) return bad("missing operand");
      if (o.a2.space && (o.i[MDT_N_CA] <= 0 || o.i[MDT_N_CA] >= o.i[MDT_N_LD] || o.i[MDT_N_CA] % o.i[MDT_N_GSIZE] || o.i[MDT_N_CA] % 4 ||
                         (o.i[MDT_N_LD] - o.i[MDT_N_CA]) % 4))
        return bad("two-source GroupNorm-apply: 0 < CA < LD, CA a multiple of the group size, both parts multiples of 4 channels");
      if (!o.a2.space && o.i[MDT_N_CA]) return bad("CA without a second source");
      break;
    case MDT_OP_RCONV:
      if (!mdt::rconv_supported(o.i[MDT_R_C], o.i[MDT_R_T], o.i[MDT_R_TAPS], o.i[MDT_R_GSIZE]))
        return bad("shape not supported by the row-stationary convolution");
      if (!o.a.space || !o.w.space || !o.out.space) return bad("missing operand");
      if (o.i[MDT_R_GSIZE] > 0 && (!o.p0.space || !o.p1.space)) return bad("GroupNorm prologue needs gain and bias");
      if (o.i[MDT_R_WF32] != 0 && o.i[MDT_R_WF32] != 1) return bad("WF32 must be 0 (split-bf16 tiles) or 1 (fp32 fragment tiles)");
      if (o.i[MDT_R_KSRC] == 2) {
        if (o.i[MDT_R_C] != 256 || o.i[MDT_R_GSIZE] || o.a2.space || o.p3.space || o.i[MDT_R_LDA] < 512)
          return bad("KSRC = 2: two 256-channel blocks of one tensor, no GroupNorm / FiLM / second source");
      } else if (o.i[MDT_R_KSRC] < 0 || (o.i[MDT_R_KSRC] > 1 && (o.i[MDT_R_KSRC] * o.i[MDT_R_C] != 1024 || o.i[MDT_R_TAPS] != 1 || o.i[MDT_R_GSIZE] ||
                                                               o.a2.space || o.p3.space || o.i[MDT_R_LDA] < 1024)))
        return bad("KSRC > 2 is a plain K = 1024 projection: KSRC * C == 1024, one tap, no GroupNorm / FiLM / second source");
      if (o.i[MDT_R_HALF_OUT] != 0 && (o.i[MDT_R_HALF_OUT] != 1 || o.i[MDT_R_C] != 256 || o.i[MDT_R_GSIZE] || o.a2.space || o.p3.space ||
                                       o.i[MDT_R_KSRC] > 1 || o.i[MDT_R_LDC] < 128 || (o.res.space && o.i[MDT_R_LDR] < 128)))
        return bad("HALF_OUT: a 256 -> 128 channel convolution without prologue (C = 256, one source), LDC (and LDR) >= 128");
      // the K-block / half-output / output-block forms read and write rows in 16-byte pieces at out + m * LDC (res + m * LDR)
      if ((o.i[MDT_R_KSRC] > 1 || o.i[MDT_R_HALF_OUT] || o.i[MDT_R_NB] > 1) &&
          (o.i[MDT_R_LDC] % 4 || o.i[MDT_R_LDA] % 4 || (o.res.space && o.i[MDT_R_LDR] % 4)))
        return bad("KSRC / HALF_OUT / NB forms: LDA, LDC and LDR must be multiples of 4 floats (16-byte row pieces)");
      if (o.i[MDT_R_NB] < 0 || o.i[MDT_R_NB] > 8 || (o.i[MDT_R_NB] > 1 && (o.i[MDT_R_GSIZE] || o.a2.space || o.p3.space || o.i[MDT_R_KSRC] > 1 ||
                                                                         o.i[MDT_R_HALF_OUT] || o.i[MDT_R_LDC] < o.i[MDT_R_NB] * o.i[MDT_R_C] ||
                                                                         (o.res.space && o.i[MDT_R_LDR] < o.i[MDT_R_NB] * o.i[MDT_R_C]))))
        return bad("NB > 1: NB x C output channels of one source without prologue, LDC (and LDR) >= NB * C");
      break;
    case MDT_OP_RESBLOCK:
      if (!mdt::resblock_supported(o.i[MDT_K_T], o.i[MDT_K_CIN], o.i[MDT_K_COUT]))
        return bad("shape not supported by the fused ResNet block");
      if (!o.a.space || !o.w.space || !o.bias.space || !o.out.space) return bad("missing operand");
      if (o.i[MDT_K_WF32] != 0 && o.i[MDT_K_WF32] != 1) return bad("WF32 must be 0 (split-bf16 fragments) or 1 (fp32 fragments)");
      if (o.i[MDT_K_CIN_REAL] < 0 || o.i[MDT_K_CIN_REAL] > o.i[MDT_K_CIN] || o.i[MDT_K_COUT_REAL] < 0 || o.i[MDT_K_COUT_REAL] > o.i[MDT_K_COUT])
        return bad("CIN_REAL / COUT_REAL must be 0 (= CIN / COUT) or a channel count inside the padded one");
      for (int k : {MDT_K_PATCH_IN, MDT_K_PATCH_OUT})
        if (o.i[k] < 0 || (o.i[k] > 1 && o.i[MDT_K_T] % o.i[k])) return bad("PATCH_IN / PATCH_OUT must be 0 / 1 (none) or a divisor of T");
      break;
    case MDT_OP_ATTN:
      if (o.i[MDT_A_T] <= 0 || o.i[MDT_A_T] > 8192 || o.i[MDT_A_TK] <= 0 || o.i[MDT_A_TK] > 8192)
        return bad("attention supports 1..8192 queries and keys per sample (more than 64 of either: the online-softmax kernel)");
      if (o.i[MDT_A_QCOL] < 0 || o.i[MDT_A_KCOL] < 0 || o.i[MDT_A_QCOL] % 4 || o.i[MDT_A_KCOL] % 4) return bad("bad q / k column offset");
      if (o.i[MDT_A_IN16] < 0 || o.i[MDT_A_IN16] > 3) return bad("IN16 is a mask: 1 = q is bf16, 2 = k | v are bf16");
      if (((o.i[MDT_A_IN16] & 1) && (o.i[MDT_A_QCOL] % 8 || o.i[MDT_A_LDQ] % 8)) ||
          ((o.i[MDT_A_IN16] & 2) && (o.i[MDT_A_KCOL] % 8 || o.i[MDT_A_LDKV] % 8)))
        return bad("bf16 q / k | v rows are read in 16-byte pieces: pitch and column offset must be multiples of 8 elements");
      if (o.i[MDT_A_OUT16] && o.i[MDT_A_LDO] % 8) return bad("a bf16 attention output is written in 16-byte pieces: LDO % 8");
      if (!o.a.space || !o.a2.space || !o.out.space) return bad("missing operand");
      break;
    case MDT_OP_ATTN_CTX:
      if (o.i[MDT_A_T] <= 0 || o.i[MDT_A_HEADS] <= 0 || o.i[MDT_A_TK] <= 0 || o.i[MDT_A_TK] > 64 || o.i[MDT_A_LDKV] < 128 || o.i[MDT_A_LDKV] % 4)
        return bad("context attention supports 1..64 keys of 128 features");
      if (!o.a.space || !o.a2.space || !o.out.space) return bad("missing operand");
      if (o.i[MDT_A_SPLIT] != 0 && o.i[MDT_A_SPLIT] != 1) return bad("SPLIT must be 0 (exact fp32 scores) or 1 (split-bf16 scores)");
      break;
    case MDT_OP_CONCAT:
      if (o.i[MDT_C_CA] % 4 || o.i[MDT_C_CB] % 4 || o.i[MDT_C_ROWS] <= 0) return bad("bad dims");
      break;
    case MDT_OP_PATCH:
      if (o.i[MDT_P_PATCH] <= 0 || o.i[MDT_P_ROWS_IN] % o.i[MDT_P_PATCH]) return bad("bad patch");
      break;
    case MDT_OP_TIME_EMBED:
      if (o.i[MDT_T_LD] < 2 * o.i[MDT_T_HALF] + 1) return bad("ld too small");
      break;
    case MDT_OP_TBLOCK: {
      const int32_t* i = o.i;
      if (i[MDT_B_C] != 128 && i[MDT_B_C] != 256) return bad("fused block needs C in {128, 256}");
      if (i[MDT_B_T] <= 0 || 16 % i[MDT_B_T]) return bad("tokens per sample must divide 16");
      if (i[MDT_B_MODE] < 0 || i[MDT_B_MODE] > 2 || i[MDT_B_NCHUNK] <= 0) return bad("bad mode / chunks");
      if (i[MDT_B_MODE] == MDT_TB_CROSS && (!o.a2.space || (16 / i[MDT_B_T]) * i[MDT_B_TK] > 64 || i[MDT_B_TK] <= 0))
        return bad("cross block needs K/V and at most 64 keys per 16 rows");
      if (!o.a.space || !o.w.space || !o.bias.space) return bad("missing operand");
      if (i[MDT_B_VARIANT] < 0 || i[MDT_B_VARIANT] > 4) return bad("unknown fused-block variant");
      if (i[MDT_B_WF32] != 0 && i[MDT_B_WF32] != 1) return bad("WF32 must be 0 (split-bf16 tiles) or 1 (fp32 fragment tiles)");
      if (i[MDT_B_VARIANT] == 1) return bad("variant 1 (16-row feature-split workgroups) was removed; C = 256 blocks are variants 2..4");
      if (i[MDT_B_VARIANT] == 0 && (i[MDT_B_C] != 128 || (i[MDT_B_MODE] == MDT_TB_CROSS && (16 / i[MDT_B_T]) * i[MDT_B_TK] > 16)))
        return bad("variant 0 serves C = 128, cross blocks with at most 16 keys per 16 rows (loader-wave kernel)");
      if (i[MDT_B_POST] && (i[MDT_B_MODE] != MDT_TB_FF || !o.out.space || i[MDT_B_VARIANT] == 1 || i[MDT_B_VARIANT] == 3 ||
                            (i[MDT_B_VARIANT] == 4 && o.p2.space)))
        return bad("a folded closing convolution needs an unsplit feed-forward block of variant 0, 2 or 4 and an output tensor");
      if (i[MDT_B_VARIANT] == 4 && (!o.out.space || (o.p2.space && i[MDT_B_NCHUNK] % 2)))
        return bad("variant 4 needs an output tensor (and an even chunk count when split)");
      if (i[MDT_B_VARIANT] == 3 && (!o.out.space || i[MDT_B_NCHUNK] % 2)) return bad("variant 3 needs a partial-sum buffer and an even chunk count");
      if (i[MDT_B_VARIANT] >= 2 && (i[MDT_B_C] != 256 || (i[MDT_B_MODE] == MDT_TB_CROSS && (16 / i[MDT_B_T]) * i[MDT_B_TK] > 48)))
        return bad("variant 2 (32-row workgroups) serves C = 256, cross blocks with at most 48 keys per 16 rows");
      break;
    }
    case MDT_OP_TF128: {
      const int32_t* i = o.i;
      if (i[MDT_F_C] != 128) return bad("fused transformer needs C = 128");
      if (!mdt::tf128_supported(i[MDT_F_T], i[MDT_F_TK], i[MDT_F_NVEC], i[MDT_F_CROSS] != 0))
        return bad("shape not supported by the fused transformer (tokens per sample must divide 16, <= 16 context rows per 16 tokens, <= 8192 vector floats)");
      if (i[MDT_F_NBLOCKS] < 0 || i[MDT_F_NT] <= 0 || i[MDT_F_HEADS] <= 0 || i[MDT_F_HEADS] > 16 || i[MDT_F_NFF] <= 0)
        return bad("bad block / tile / head counts");
      if (i[MDT_F_NPOST] != 0 && i[MDT_F_NPOST] != 2) return bad("npost must be 0 or 2");
      if (i[MDT_F_WF32] != 0 && i[MDT_F_WF32] != 1) return bad("WF32 must be 0 (split-bf16 tiles) or 1 (fp32 fragment tiles)");
      if (!o.a.space || !o.out.space || !o.w.space || !o.bias.space || !o.p0.space) return bad("missing operand");
      if (i[MDT_F_CROSS] && !o.a2.space) return bad("cross-attention blocks need the hoisted K/V rows");
      if (i[MDT_F_RES_KIND] < 0 || i[MDT_F_RES_KIND] > 2 || (i[MDT_F_RES_KIND] == 0) != (i[MDT_F_N_RES] == 0) || i[MDT_F_N_RES] < 0 ||
          i[MDT_F_N_RES] > 255)
        return bad("bad ResNet block kind / count");
      if (i[MDT_F_N_RES] == 0 && (i[MDT_F_NBLOCKS] == 0 || i[MDT_F_NFILM])) return bad("a transformer launch needs blocks");
      if (i[MDT_F_NBLOCKS] == 0 && (i[MDT_F_HAS_IN] || i[MDT_F_NPOST])) return bad("to_in / to_out without transformer blocks");
      if (i[MDT_F_N_RES] > 0 && (!o.res.space || !o.p3.space || i[MDT_F_NFILM] % 256 || i[MDT_F_NFILM] < 256 * i[MDT_F_N_RES] ||
                                 i[MDT_F_NVEC] + i[MDT_F_NFILM] > 8192))
        return bad("ResNet blocks need the skip tensors (res), the FiLM rows (p3) and NVEC + NFILM <= 8192");
      break;
    }
    case MDT_OP_TF256: {
      const int32_t* i = o.i;
      if (i[MDT_F_C] != 256) return bad("fused transformer (32-row form) needs C = 256");
      if (!mdt::tf256_supported(i[MDT_F_T], i[MDT_F_TK], i[MDT_F_HEADS], i[MDT_F_NFF], i[MDT_F_CROSS] != 0))
        return bad("shape not supported by the fused transformer (tokens per sample must divide 16, 8 heads, hidden 512, <= 48 context rows per 16 tokens)");
      if (i[MDT_F_NBLOCKS] <= 0 || i[MDT_F_NT] <= 0 || i[MDT_F_NVEC] <= 0 || i[MDT_F_NVEC] % 768) return bad("bad block / tile / vector counts");
      if (i[MDT_F_NPOST] != 0 && i[MDT_F_NPOST] != 8) return bad("npost must be 0 or 8");
      if (i[MDT_F_WF32] != 0 && i[MDT_F_WF32] != 1) return bad("WF32 must be 0 (split-bf16 tiles) or 1 (fp32 fragment tiles)");
      if (!o.a.space || !o.out.space || !o.w.space || !o.bias.space || !o.p0.space) return bad("missing operand");
      if (i[MDT_F_CROSS] && !o.a2.space) return bad("cross-attention blocks need the hoisted K/V rows");
      if (i[MDT_F_NSPLIT] != 0 && i[MDT_F_NSPLIT] != 1 && i[MDT_F_NSPLIT] != 2) return bad("NSPLIT must be 1 or 2");
      if (i[MDT_F_NSPLIT] == 2 && (!o.p2.space || !o.p3.space || i[MDT_F_PAIR_STRIDE] < 0 || i[MDT_F_PAIR_STRIDE] > 64))
        return bad("the pair-split form needs the hand-off flags (p2), the hand-off blocks (p3) and a pair stride of 1..64");
      break;
    }
    case MDT_OP_RES256: {
      const int32_t* i = o.i;
      if (i[MDT_F_C] != 256) return bad("the ResNet chain needs C = 256");
      if (!mdt::res256_supported(i[MDT_F_T], i[MDT_F_RES_KIND], i[MDT_F_N_RES], i[MDT_F_NPOST]))
        return bad("ResNet chain: tokens per sample must divide 16, kind 1 | 2, 1..255 blocks, taps 3 (or 1 with one token per sample)");
      if (i[MDT_F_NT] <= 0 || i[MDT_F_HEADS] <= 0 || i[MDT_F_HEADS] > i[MDT_F_NT] || i[MDT_F_NBLOCKS] || i[MDT_F_HAS_IN] || i[MDT_F_CROSS])
        return bad("ResNet chain: bad tile / segment count or stray transformer fields");
      if (i[MDT_F_NSPLIT] != 0 && i[MDT_F_NSPLIT] != 1 && i[MDT_F_NSPLIT] != 2) return bad("NSPLIT must be 1 or 2");
      if (i[MDT_F_NSPLIT] == 2 && (!o.a2.space || !o.p1.space || i[MDT_F_PAIR_STRIDE] < 0 || i[MDT_F_PAIR_STRIDE] > 64 || i[MDT_F_NFF] <= 0))
        return bad("the pair-split chain needs the hand-off flags (a2), the hand-off blocks (p1), a pair stride of 1..64 and NFF = the "
                   "weight sub-tiles of one half");
      if (i[MDT_F_NSPLIT] != 2 && i[MDT_F_NFF]) return bad("ResNet chain: stray transformer fields");
      if (i[MDT_F_WF32] != 0 && i[MDT_F_WF32] != 1) return bad("WF32 must be 0 (split-bf16 tiles) or 1 (fp32 fragment tiles)");
      if (i[MDT_F_NVEC] != i[MDT_F_N_RES] * (i[MDT_F_RES_KIND] == 1 ? 6 : 9) * 256 || i[MDT_F_NFILM] < 512 * i[MDT_F_N_RES])
        return bad("ResNet chain: NVEC must be N_RES x (6 | 9) x 256 floats and NFILM >= 512 N_RES");
      if (!o.a.space || !o.out.space || !o.w.space || !o.bias.space || !o.p0.space || !o.res.space || !o.p3.space) return bad("missing operand");
      break;
    }
    default:
      return bad("unknown op kind");
  }
  return nullptr;
}

mdt_program* mdt_program_create(const mdt_op* ops, int32_t n_ops) {
  if (!ops || n_ops < 0) {
    fail("mdt_program_create: bad arguments");
    return nullptr;
  }
  char buf[256];
  for (int i = 0; i < n_ops; ++i) {
    if (const char* why = validate(ops[i], i, buf, sizeof buf)) {
      fail(std::string("mdt_program_create: ") + why);
      return nullptr;
    }
  }
  mdt_program* p = new mdt_program();
  p->ops.assign(ops, ops + n_ops);
  return p;
}

void mdt_program_destroy(mdt_program* p) { delete p; }
int32_t mdt_program_num_ops(const mdt_program* p) { return p ? (int32_t)p->ops.size() : 0; }

int mdt_program_run(const mdt_program* p, const mdt_bindings* bd, int32_t B, int32_t n_shared_rows, int32_t first,
                    int32_t count, void* stream_) {
  if (!p || !bd) return fail("mdt_program_run: null program or bindings");
  if (B <= 0) return fail("mdt_program_run: B must be positive");
  hipStream_t stream = (hipStream_t)stream_;
  const int n = (int)p->ops.size();
  if (first < 0 || first > n) return fail("mdt_program_run: bad op range");
  const int last = count < 0 ? n : first + count;
  if (last > n) return fail("mdt_program_run: bad op range");

  bool missing = false;
  auto ptr = [&](const mdt_ref& r) -> float* {
    switch (r.space) {
      case MDT_SP_NONE: return nullptr;
      case MDT_SP_WEIGHT:
        if (!bd->weights) missing = true;
        return const_cast<float*>(bd->weights) + r.off;
      case MDT_SP_ACT:
        if (!bd->act) missing = true;
        return bd->act + r.off * (int64_t)B;
      case MDT_SP_SHR:
        if (!bd->shr) missing = true;
        return bd->shr + r.off;
      default: {
        float* e = bd->ext[r.space - MDT_SP_EXT0];
        if (!e) missing = true;
        return e + r.off;
      }
    }
  };

  static const bool no_prefetch = mdt_tuning_env("MDT_NO_PREFETCH") != nullptr;     // tuning aid: no next-launch weight prefetch
  for (int idx = first; idx < last; ++idx) {
    const mdt_op& o = p->ops[idx];
    hipError_t e = hipSuccess;
    // the weight stream of the NEXT op, if it is a ring kernel's (MDT_W_KB: its size, set by the host compiler): the loader
    // waves of this launch pull it into the L2s when they are done (mdt_kernels.h: prefetch_next_weights)
    const void* pf_ptr = nullptr;
    int pf_lines = 0;
    if (!no_prefetch && idx + 1 < last) {
      const mdt_op& nx = p->ops[idx + 1];
      if ((nx.kind == MDT_OP_TBLOCK || nx.kind == MDT_OP_RCONV || nx.kind == MDT_OP_TF128 || nx.kind == MDT_OP_TF256 || nx.kind == MDT_OP_RES256) &&
          nx.i[MDT_W_KB] > 0 && nx.w.space == MDT_SP_WEIGHT && bd->weights) {
        pf_ptr = bd->weights + nx.w.off;
        // at most the first 2 MB: an XCD's L2 holds 4 MB, the long streams of the whole-transformer launches are pulled in from
        // inside the launch as they go (k_tf256.hip, MDT_STREAM_PF), and every line here is fetched by all eight XCDs
        pf_lines = (nx.i[MDT_W_KB] < 2048 ? nx.i[MDT_W_KB] : 2048) * 8;
      }
    }
    switch (o.kind) {
      case MDT_OP_GEMM: {
        const int32_t* i = o.i;
        mdt::GemmArgs g;
        g.A = ptr(o.a); g.W = ptr(o.w); g.W_lo = ptr(o.a2); g.bias = ptr(o.bias); g.out = ptr(o.out); g.res = ptr(o.res);
        g.p0 = ptr(o.p0); g.p1 = ptr(o.p1); g.p2 = ptr(o.p2); g.p3 = ptr(o.p3);
        const int batches = i[MDT_G_M_MODE] == 0 ? B : (i[MDT_G_M_MODE] == 1 ? n_shared_rows : 1);
        g.M = batches * i[MDT_G_R_OUT];
        g.r_out = i[MDT_G_R_OUT]; g.r_in = i[MDT_G_R_IN]; g.lda = i[MDT_G_LDA]; g.cin = i[MDT_G_CIN];
        g.taps = i[MDT_G_TAPS]; g.t_stride = i[MDT_G_T_STRIDE]; g.t_dj = i[MDT_G_T_DJ]; g.t_off = i[MDT_G_T_OFF];
        g.N = i[MDT_G_N]; g.ldc = i[MDT_G_LDC]; g.o_rows = i[MDT_G_O_ROWS]; g.o_stride = i[MDT_G_O_STRIDE];
        g.o_off = i[MDT_G_O_OFF]; g.ldr = i[MDT_G_LDR]; g.pro = i[MDT_G_PRO]; g.groups = i[MDT_G_GROUPS];
        g.gsize = i[MDT_G_GSIZE]; g.pro_silu = i[MDT_G_PRO_SILU]; g.act = i[MDT_G_ACT]; g.a_col = i[MDT_G_A_COL];
        g.o_col = i[MDT_G_O_COL]; g.eps = o.f[MDT_GF_EPS]; g.phases = i[MDT_G_PHASES]; g.wfmt = i[MDT_G_WFMT];
        if (!missing && (i[MDT_G_WFMT] == 16 || i[MDT_G_WFMT] == 17)) {
          mdt::ProjArgs h;
          h.x = g.A + g.a_col; h.w = g.W; h.bias = g.bias; h.res = g.res; h.gamma = g.p0; h.beta = g.p1; h.out = g.out + g.o_col;
          h.M = g.M; h.N = g.N; h.K = g.cin; h.lda = g.lda; h.ldc = g.ldc; h.ldr = g.res ? g.ldr : 0; h.ln = g.pro == MDT_PRO_LAYERNORM;
          h.eps = g.eps; h.nch = 0; h.wf32 = i[MDT_G_WFMT] == 17;
          e = mdt::launch_proj(h, stream);
        } else if (!missing && (i[MDT_G_WFMT] & 2)) {
          mdt::Gemm16Args h;
          h.A = reinterpret_cast<const unsigned short*>(g.A); h.W = reinterpret_cast<const unsigned short*>(g.W);
          h.bias = g.bias; h.res = g.res; h.out = g.out; h.M = g.M; h.N = g.N; h.cin = g.cin; h.taps = g.taps; h.rows = g.r_in;
          h.lda = g.lda; h.a_col = g.a_col; h.t_dj = g.t_dj; h.t_off = g.t_off; h.ldc = g.ldc; h.ldr = g.ldr; h.o_col = g.o_col;
          h.act = g.act; h.out16 = (i[MDT_G_WFMT] & 4) ? 1 : 0; h.res16 = (i[MDT_G_WFMT] & 32) ? 1 : 0;
          h.copy16 = (i[MDT_G_WFMT] & 8) ? reinterpret_cast<unsigned short*>(ptr(o.p0)) : nullptr;
          h.csum = (i[MDT_G_WFMT] & 128) ? ptr(o.p0) : nullptr;
          h.eps = g.eps;
          e = mdt::launch_gemm_b16(h, stream);
        } else if (!missing) {
          static const bool no_as = mdt_tuning_env("MDT_NO_AS") != nullptr;   // tuning aid: disable the A-stationary kernel
          if (i[MDT_G_WFMT] == 1) e = mdt::launch_gemm_bf16(g, stream);
          else if (!g.W_lo) e = mdt::launch_gemm(g, stream);
          else if (!no_as && mdt::gemm_as_eligible(g)) e = mdt::launch_gemm_as(g, stream);
          else e = mdt::launch_gemm_bf16x3(g, stream);
        }
        break;
      }
      case MDT_OP_PREP16: {
        const int32_t* i = o.i;
        mdt::Prep16Args g;
        g.a = ptr(o.a); g.out = reinterpret_cast<unsigned short*>(ptr(o.out)); g.p0 = ptr(o.p0); g.p1 = ptr(o.p1);
        g.p2 = ptr(o.p2); g.p3 = ptr(o.p3); g.rows = i[MDT_G_R_IN]; g.total_rows = B * g.rows; g.lda = i[MDT_G_LDA];
        g.a_col = i[MDT_G_A_COL]; g.cin = i[MDT_G_CIN]; g.pro = i[MDT_G_PRO]; g.groups = i[MDT_G_GROUPS];
        g.gsize = i[MDT_G_GSIZE]; g.pro_silu = i[MDT_G_PRO_SILU]; g.eps = o.f[MDT_GF_EPS]; g.in16 = i[MDT_G_WFMT] == 2 ? 1 : 0;
        if (!missing) e = mdt::launch_prep16(g, stream);
        break;
      }
      case MDT_OP_GN_STATS: {
        mdt::GnStatsArgs g;
        g.x = ptr(o.a); g.stats = ptr(o.out); g.batch = B; g.rows = o.i[MDT_N_ROWS]; g.ld = o.i[MDT_N_LD];
        g.groups = o.i[MDT_N_GROUPS]; g.gsize = o.i[MDT_N_GSIZE]; g.eps = o.f[MDT_NF_EPS];
        if (!missing) e = mdt::launch_gn_stats(g, stream);
        break;
      }
      case MDT_OP_GN_ACT: {
        mdt::GnActArgs a;
        a.x = ptr(o.a); a.y = ptr(o.out); a.gamma = ptr(o.p0); a.beta = ptr(o.p1); a.film = ptr(o.p3);
        a.batch = B; a.rows = o.i[MDT_N_ROWS]; a.ld = o.i[MDT_N_LD]; a.groups = o.i[MDT_N_GROUPS];
        a.gsize = o.i[MDT_N_GSIZE]; a.silu = o.i[MDT_N_SILU]; a.eps = o.f[MDT_NF_EPS]; a.out16 = o.i[MDT_N_OUT16];
        a.x2 = ptr(o.a2); a.ca = o.i[MDT_N_CA]; a.scale2 = o.f[MDT_NF_SCALE2]; a.raw16 = reinterpret_cast<unsigned short*>(ptr(o.p2));
        if (!missing) e = mdt::launch_gn_act(a, stream);
        break;
      }
      case MDT_OP_RCONV: {
        mdt::RConvArgs a;
        a.x = ptr(o.a); a.x2 = ptr(o.a2); a.w = ptr(o.w); a.bias = ptr(o.bias); a.res = ptr(o.res); a.out = ptr(o.out);
        a.gamma = ptr(o.p0); a.beta = ptr(o.p1); a.film = ptr(o.p3); a.dbgbuf = ptr(o.p2);
        a.T = o.i[MDT_R_T]; a.M = B * a.T; a.C = o.i[MDT_R_C]; a.lda = o.i[MDT_R_LDA]; a.ldc = o.i[MDT_R_LDC];
        a.ldr = o.i[MDT_R_LDR]; a.taps = o.i[MDT_R_TAPS]; a.gsize = o.i[MDT_R_GSIZE]; a.silu = o.i[MDT_R_SILU];
        a.film_ld = o.i[MDT_R_FILM_LD]; a.eps = o.f[MDT_RF_EPS]; a.in_scale = o.f[MDT_RF_IN_SCALE];
        a.lda2 = o.i[MDT_R_LDA2]; a.in_scale2 = o.f[MDT_RF_IN_SCALE2];
        a.pf_ptr = pf_ptr; a.pf_lines = pf_lines; a.wf32 = o.i[MDT_R_WF32]; a.ksrc = o.i[MDT_R_KSRC]; a.half_out = o.i[MDT_R_HALF_OUT]; a.nb = o.i[MDT_R_NB];
        if (!missing) e = mdt::launch_rconv(a, stream);
        break;
      }
      case MDT_OP_RESBLOCK: {
        mdt::ResBlockArgs a;
        a.x = ptr(o.a); a.out = ptr(o.out); a.w = ptr(o.w); a.vec = ptr(o.bias); a.film = ptr(o.p3);
        a.B = B; a.T = o.i[MDT_K_T]; a.cin = o.i[MDT_K_CIN]; a.cout = o.i[MDT_K_COUT]; a.film_ld = o.i[MDT_K_FILM_LD];
        a.eps = o.f[MDT_KF_EPS]; a.wf32 = o.i[MDT_K_WF32];
        a.cin_real = o.i[MDT_K_CIN_REAL] > 0 ? o.i[MDT_K_CIN_REAL] : a.cin;
        a.cout_real = o.i[MDT_K_COUT_REAL] > 0 ? o.i[MDT_K_COUT_REAL] : a.cout;
        a.patch_in = o.i[MDT_K_PATCH_IN]; a.patch_out = o.i[MDT_K_PATCH_OUT];
        if (!missing) e = mdt::launch_resblock(a, stream);
        break;
      }
      case MDT_OP_ATTN: {
        mdt::AttnArgs a;
        a.q = ptr(o.a); a.k = ptr(o.a2); a.out = ptr(o.out); a.batch = B; a.T = o.i[MDT_A_T]; a.Tk = o.i[MDT_A_TK];
        a.in16 = o.i[MDT_A_IN16]; a.split_scores = 0;
        if (a.q) a.q = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.q) + (size_t)o.i[MDT_A_QCOL] * ((a.in16 & 1) ? 2 : 4));
        if (a.k) a.k = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.k) + (size_t)o.i[MDT_A_KCOL] * ((a.in16 & 2) ? 2 : 4));
        a.heads = o.i[MDT_A_HEADS]; a.ldq = o.i[MDT_A_LDQ]; a.ldkv = o.i[MDT_A_LDKV]; a.ldo = o.i[MDT_A_LDO];
        a.kv_bstride = o.i[MDT_A_KV_BSTRIDE]; a.scale = o.f[MDT_AF_SCALE]; a.out16 = o.i[MDT_A_OUT16];
        if (!missing) e = mdt::launch_attn(a, stream);
        break;
      }
      case MDT_OP_ATTN_CTX: {
        mdt::AttnArgs a;
        a.q = ptr(o.a); a.k = ptr(o.a2); a.out = ptr(o.out); a.batch = B; a.T = o.i[MDT_A_T]; a.Tk = o.i[MDT_A_TK];
        a.heads = o.i[MDT_A_HEADS]; a.ldq = 128; a.ldkv = o.i[MDT_A_LDKV]; a.ldo = 128;
        a.kv_bstride = o.i[MDT_A_KV_BSTRIDE]; a.scale = o.f[MDT_AF_SCALE]; a.out16 = 0; a.split_scores = o.i[MDT_A_SPLIT]; a.in16 = 0;
        if (!missing) e = mdt::launch_attn_ctx(a, stream);
        break;
      }
      case MDT_OP_CONCAT: {
        float *a = ptr(o.a), *b2 = ptr(o.a2), *out = ptr(o.out);
        if (!missing)
          e = mdt::launch_concat(a, b2, out, (int64_t)B * o.i[MDT_C_ROWS], o.i[MDT_C_CA], o.i[MDT_C_CB],
                                 o.f[MDT_CF_SCALE_B], stream);
        break;
      }
      case MDT_OP_PATCH: {
        float *in = ptr(o.a), *out = ptr(o.out);
        if (!missing)
          e = mdt::launch_patch(in, out, B, o.i[MDT_P_ROWS_IN], o.i[MDT_P_C_IN], o.i[MDT_P_LD_IN], o.i[MDT_P_LD_OUT],
                                o.i[MDT_P_PATCH], o.i[MDT_P_INVERSE], stream);
        break;
      }
      case MDT_OP_TBLOCK: {
        mdt::TBlockArgs a;
        a.x = ptr(o.a); a.w = ptr(o.w); a.bias = ptr(o.bias); a.kv = ptr(o.a2); a.dbgbuf = ptr(o.p0);
        a.mode = o.i[MDT_B_MODE]; a.C = o.i[MDT_B_C]; a.T = o.i[MDT_B_T]; a.M = B * a.T;
        a.nchunk = o.i[MDT_B_NCHUNK]; a.nbias = o.i[MDT_B_NBIAS]; a.ldx = a.C; a.Tk = o.i[MDT_B_TK];
        a.kv_bstride = o.i[MDT_B_KV_BSTRIDE]; a.ldkv = o.i[MDT_B_LDKV]; a.nheads = o.i[MDT_B_HEADS]; a.nsamples = B;
        a.eps = o.f[MDT_BF_EPS]; a.scale = o.f[MDT_BF_SCALE];
        a.part = nullptr; a.nsplit = 1; a.xout = nullptr; a.pin = nullptr; a.pout = nullptr;
        a.pf_ptr = pf_ptr; a.pf_lines = pf_lines;
        a.kv2 = o.i[MDT_B_KV2] ? ptr(o.p1) : nullptr;
        if (o.i[MDT_B_KV2]) {
          // the kernels pick conditional vs shared K/V per WORKGROUP (64 rows at C = 128, 32 rows in the C = 256 kernels):
          // the first half of the samples must be whole workgroups
          const int per_wg = (o.i[MDT_B_VARIANT] >= 2 ? 32 : 64) / a.T;
          if (!o.p1.space || B % 2 || (B / 2) % (per_wg > 0 ? per_wg : 1))
            return fail("mdt_program_run: a dual-batch cross block needs the shared K/V rows and B = 2 x (a multiple of the samples per workgroup)");
        }
        a.post = o.i[MDT_B_POST]; a.wf32 = o.i[MDT_B_WF32];
        if (a.post) a.xout = ptr(o.out);
        if (o.i[MDT_B_VARIANT] == 3) { a.part = ptr(o.out); a.nsplit = 2; }
        if (o.i[MDT_B_VARIANT] == 4) {
          a.xout = ptr(o.out); a.pin = ptr(o.res); a.pout = ptr(o.p2);
          a.nsplit = o.p2.space ? 2 : 1;
        }
        if (!missing)
          e = o.i[MDT_B_VARIANT] >= 2 ? mdt::launch_tblock32(a, stream) : mdt::launch_tblock_lw(a, stream);
        break;
      }
      case MDT_OP_TF128:
      case MDT_OP_TF256: {
        const int32_t* i = o.i;
        const bool wide = o.kind == MDT_OP_TF256;
        mdt::TFArgs a;
        a.x = ptr(o.a); a.out = ptr(o.out); a.w = ptr(o.w); a.vec = ptr(o.bias);
        a.tiles = reinterpret_cast<const unsigned*>(ptr(o.p0));
        a.kv = i[MDT_F_CROSS] ? ptr(o.a2) : nullptr;
        a.kv2 = i[MDT_F_KV2] ? ptr(o.p1) : nullptr;
        a.dbgbuf = nullptr;
        a.T = i[MDT_F_T]; a.M = B * a.T; a.NT = i[MDT_F_NT]; a.nvec = i[MDT_F_NVEC]; a.Tk = i[MDT_F_TK];
        a.kv_bstride = i[MDT_F_KV_BSTRIDE]; a.ldkv = i[MDT_F_LDKV]; a.nheads = i[MDT_F_HEADS]; a.nsamples = B;
        a.has_in = i[MDT_F_HAS_IN]; a.nblocks = i[MDT_F_NBLOCKS]; a.nff = i[MDT_F_NFF]; a.npost = i[MDT_F_NPOST];
        // K|V rows of consecutive cross layers: per-sample arena -> scaled by B; shared arena (fixed embedding) -> as is
        a.kv_lstride = (int64_t)i[MDT_F_KV_LSTRIDE] * (o.a2.space == MDT_SP_ACT ? B : 1);
        a.kv2_lstride = (int64_t)i[MDT_F_KV_LSTRIDE] * (o.p1.space == MDT_SP_ACT ? B : 1);
        a.eps_ln = o.f[MDT_FF_EPS_LN]; a.scale = o.f[MDT_FF_SCALE]; a.eps_gn = o.f[MDT_FF_EPS_GN];
        a.res_kind = wide ? 0 : i[MDT_F_RES_KIND]; a.n_res = wide ? 0 : i[MDT_F_N_RES];
        a.res_pair1 = i[MDT_F_RES_PAIR1]; a.res_pair2 = i[MDT_F_RES_PAIR2]; a.nfilm = wide ? 0 : i[MDT_F_NFILM];
        a.film = a.n_res ? ptr(o.p3) : nullptr; a.skip = a.n_res ? ptr(o.res) : nullptr;
        // skip tensors of consecutive blocks: whole tensors apart (ascending where they are produced, descending where consumed)
        a.skip_stride = (int64_t)B * a.T * i[MDT_F_C] * (a.res_kind == 2 ? -1 : 1);
        a.skip_scale = o.f[MDT_FF_SKIP_SCALE]; a.eps_res = o.f[MDT_FF_EPS_RES];
        a.pf_ptr = pf_ptr; a.pf_lines = pf_lines;
        a.nsplit = wide && i[MDT_F_NSPLIT] == 2 ? 2 : 1;
        a.pair_stride = g_pair_stride > 0 ? g_pair_stride : (i[MDT_F_PAIR_STRIDE] > 0 ? i[MDT_F_PAIR_STRIDE] : 8);
        a.xflags = a.nsplit == 2 ? reinterpret_cast<unsigned*>(const_cast<float*>(ptr(o.p2))) : nullptr;
        a.xbuf = a.nsplit == 2 ? const_cast<float*>(ptr(o.p3)) : nullptr;
        a.wf32 = i[MDT_F_WF32]; a.rb_base = 0;
        if (i[MDT_F_KV2]) {
          const int per_wg = (wide ? 32 : 64) / a.T;
          if (!o.p1.space || B % 2 || (B / 2) % (per_wg > 0 ? per_wg : 1))
            return fail("mdt_program_run: a dual-batch fused transformer needs the shared K/V rows and B = 2 x (a multiple of the samples per workgroup)");
        }
        if (!missing) e = wide ? mdt::launch_tf256(a, stream) : mdt::launch_tf128(a, stream);
        break;
      }
      case MDT_OP_RES256: {
        const int32_t* i = o.i;
        mdt::TFArgs a = {};
        a.x = ptr(o.a); a.out = ptr(o.out); a.w = ptr(o.w); a.vec = ptr(o.bias);
        a.tiles = reinterpret_cast<const unsigned*>(ptr(o.p0));
        a.T = i[MDT_F_T]; a.M = B * a.T; a.NT = i[MDT_F_NT]; a.nvec = i[MDT_F_NVEC]; a.nsamples = B;
        a.nheads = i[MDT_F_HEADS];                       // descriptors (segments) behind p0
        a.npost = i[MDT_F_NPOST];                        // taps of the block convolutions
        a.res_kind = i[MDT_F_RES_KIND]; a.n_res = i[MDT_F_N_RES]; a.nfilm = i[MDT_F_NFILM];
        a.film = ptr(o.p3); a.skip = ptr(o.res);
        a.skip_stride = (int64_t)B * a.T * 256 * (a.res_kind == 2 ? -1 : 1);
        a.skip_scale = o.f[MDT_FF_SKIP_SCALE]; a.eps_res = o.f[MDT_FF_EPS_RES];
        a.pf_ptr = pf_ptr; a.pf_lines = pf_lines;
        a.nsplit = i[MDT_F_NSPLIT] == 2 ? 2 : 1; a.wf32 = i[MDT_F_WF32];
        // pair-split chain (round 6): hand-off flags (a2) / blocks (p1) and pair stride as MDT_OP_TF256's, NFF = sub-tiles of ONE half
        a.pair_stride = g_pair_stride > 0 ? g_pair_stride : (i[MDT_F_PAIR_STRIDE] > 0 ? i[MDT_F_PAIR_STRIDE] : 8);
        a.xflags = a.nsplit == 2 ? reinterpret_cast<unsigned*>(const_cast<float*>(ptr(o.a2))) : nullptr;
        a.xbuf = a.nsplit == 2 ? const_cast<float*>(ptr(o.p1)) : nullptr;
        a.nff = i[MDT_F_NFF]; a.rb_base = 0;
        a.dbgbuf = ptr(o.p2);                            // (tuning builds with -DMDT_STAMPS: clock stamps of workgroup 0; else unused)
        if (!missing) e = mdt::launch_res256(a, stream);
        break;
      }
      case MDT_OP_TIME_EMBED: {
        float *cn = ptr(o.a), *w = ptr(o.w), *out = ptr(o.out);
        if (!missing) e = mdt::launch_time_embed(cn, w, out, n_shared_rows, o.i[MDT_T_HALF], o.i[MDT_T_LD], stream);
        break;
      }
      default:
        return fail("mdt_program_run: unknown op kind");
    }
    if (missing) {
      char buf[128];
      snprintf(buf, sizeof buf, "mdt_program_run: op %d references an unbound buffer", idx);
      return fail(buf);
    }
    if (e != hipSuccess) {
      char buf[128];
      snprintf(buf, sizeof buf, "mdt_program_run: op %d (kind %d) launch failed", idx, o.kind);
      return hip_fail(buf, e);
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
mdt_timer* mdt_timer_create(int32_t max_intervals) {
  if (max_intervals <= 0) {
    fail("mdt_timer_create: max_intervals must be positive");
    return nullptr;
  }
  mdt_timer* t = new mdt_timer();
  t->start.resize(max_intervals);
  t->stop.resize(max_intervals);
  for (int i = 0; i < max_intervals; ++i) {
    if (hipEventCreate(&t->start[i]) != hipSuccess || hipEventCreate(&t->stop[i]) != hipSuccess) {
      fail("mdt_timer_create: hipEventCreate failed");
      delete t;
      return nullptr;
    }
  }
  return t;
}

void mdt_timer_destroy(mdt_timer* t) {
  if (!t) return;
  for (auto e : t->start) (void)hipEventDestroy(e);
  for (auto e : t->stop) (void)hipEventDestroy(e);
  delete t;
}

int mdt_timer_start(mdt_timer* t, void* stream) {
  if (!t || t->open || t->n >= (int)t->start.size()) return fail("mdt_timer_start: timer full or interval open");
  hipError_t e = hipEventRecord(t->start[t->n], (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail("mdt_timer_start", e);
  t->open = true;
  return 0;
}

int mdt_timer_stop(mdt_timer* t, void* stream) {
  if (!t || !t->open) return fail("mdt_timer_stop: no open interval");
  hipError_t e = hipEventRecord(t->stop[t->n], (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail("mdt_timer_stop", e);
  t->open = false;
  t->n++;
  return 0;
}

int32_t mdt_timer_collect(mdt_timer* t, float* ms, int32_t cap) {
  if (!t || t->open) {
    fail("mdt_timer_collect: interval still open");
    return -1;
  }
  const int n = t->n < cap ? t->n : cap;
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipEventSynchronize(t->stop[i]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms[i], t->start[i], t->stop[i]);
    if (e != hipSuccess) {
      hip_fail("mdt_timer_collect", e);
      return -1;
    }
  }
  t->n = 0;
  return n;
}

}  // extern "C"
