/* C ABI of libs2st_hip.so -- the MI355X-native (gfx950) kernels and training engine behind
 * the s2st_transformer hot path.
 *
 * The reference has no FFI on this path: its operators are ATen calls made from Python
 * (SURVEY.md section 8(b)).  Each entry point below therefore cites the reference call
 * site(s) whose arithmetic it replaces.  Conventions: plain pointers + sizes, caller-owned
 * DEVICE buffers, stream-ordered (`stream` is a hipStream_t passed as void*; NULL = the
 * default stream), re-entrant, return 0 on success or a negative code (no exceptions cross
 * the boundary).  All floating-point buffers are fp32; token ids are int64, lengths int32.
 */
#ifndef S2ST_HIP_H
#define S2ST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define S2ST_OK 0
#define S2ST_ERR_LAUNCH (-1)
#define S2ST_ERR_SHAPE (-2)
#define S2ST_ERR_WORKSPACE (-3)
#define S2ST_ERR_ARG (-4)

/* offset of "slow index" i:  per <= 0 ? i*ld : (i / per) * bs + (i % per) * ld          */
typedef struct { int64_t ld; int64_t bs; int32_t per; int32_t _pad; } s2st_split;

typedef struct {
  const float* p;
  int32_t kmajor; /* 1: X(r,k) at p + split(r) + k ; 0: X(r,k) at p + split(k) + r */
  int32_t _pad;
  s2st_split sp;
  int64_t zo, zi; /* batch strides: z -> (z / zdiv) * zo + (z % zdiv) * zi */
} s2st_gemm_operand;

typedef struct {
  float* p;
  s2st_split sp; /* C(m,n) at p + split(m) + n */
  int64_t zo, zi;
} s2st_gemm_out;

typedef struct {
  float alpha;
  int32_t act;        /* 0 none, 1 relu */
  const float* bias;  /* [N] or NULL */
  float drop_p;       /* dropout after activation, 0 = off */
  int32_t accumulate; /* C += value */
  uint64_t seed;
  const float* resid; /* added last; addressed like C; or NULL */
} s2st_gemm_epilogue;

typedef struct {
  s2st_gemm_operand A, B;
  s2st_gemm_out C;
  s2st_gemm_epilogue ep;
  int32_t M, N, K;
  int32_t batch, zdiv;
  int32_t precise; /* 0: bf16 MFMA; 1: bf16x3 split (~fp32 accuracy, parity tests) */
  int32_t splitk, kchunk, avec, bvec; /* filled by the launcher */
} s2st_gemm_args;

/* C(m,n) = epi(alpha * sum_k A(m,k) B(n,k)).  Replaces F.linear / F.conv1d / torch.bmm at
 * fairseq/modules/multihead_attention.py:170-192,332,367, transformer_layer.py:158-162,
 * examples/s2s_trans/models/s2st_transformer.py:135-139,452-455, tacotron2.py:95-126. */
int s2st_gemm_f32(const s2st_gemm_args* args, void* stream);

int s2st_version(void);
/* number of HIP devices visible (0 = none: every compute entry point then fails) */
int s2st_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* S2ST_HIP_H */
