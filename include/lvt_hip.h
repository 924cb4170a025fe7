/*
 * lvt_hip.h -- C ABI of liblvt_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the Latent Video
 * Transformer hot path (VQ-VAE conv encoder/decoder + product vector-quantiser, DSFVT transformer).
 *
 * The reference (rakhimovv/lvt, python package `vidgen`) has NO native code and no FFI: its hot path
 * dispatches torch ops.  Each entry point below replaces the torch
 * op(s) named in its comment; `lvt_amd/hip/` binds them with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success or a negative LVT_E* code; it never throws.  A message is
 *     available from lvt_last_error().
 *   - all pointers are DEVICE pointers to caller-owned, contiguous fp32 / int64 / int32 buffers;
 *     nothing is allocated inside; scratch is a caller-provided workspace.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*).
 *   - activations are CHANNELS-LAST: (N, T, H, W, C) with C fastest; a frame batch is T == 1.
 *   - arithmetic is fp32 in / fp32 out with fp32 accumulation on the matrix cores: by default every product is formed
 *     exactly from a 3-way bf16 split (six v_mfma_f32_32x32x16_bf16 per block); from a 2-way fp16 split after an exact
 *     power-of-two scale (three v_mfma_f32_32x32x16_f16) when a call carries LVT_MATH_F16X2; or on v_mfma_f32_32x32x2_f32
 *     when it carries LVT_MATH_F32 (see below).
 */
#ifndef LVT_HIP_H
#define LVT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LVT_OK            0
#define LVT_EINVAL      (-1)   /* bad argument (shape, alignment, null pointer)        */
#define LVT_EWORKSPACE  (-2)   /* workspace too small                                  */
#define LVT_ELAUNCH     (-3)   /* hipLaunchKernel / runtime error                      */
#define LVT_ENODEVICE   (-4)   /* no gfx950 device visible                             */

const char *lvt_last_error(void);
int lvt_version(void);          /* 610 = round 6 (ABI changes are listed in INTEGRATION.md) */
/* Device probe: name, CU count, clock (kHz), HBM bytes.  Returns LVT_ENODEVICE without a GPU. */
int lvt_device_info(char *name, int name_len, int *cus, int *clock_khz, long long *hbm_bytes);

/* Arithmetic of the GEMM / conv engine.  Operands, results and accumulators are fp32 in both modes; the mode is chosen
 * PER CALL by LVT_MATH_F32 in the `flags` of an entry point -- the library holds no mutable state (thread-safe by
 * construction: the thread that runs a backward pass is not the one that ran the forward).
 *   flag clear (default) "bf16x3": every fp32 operand is split exactly into three bf16 terms (24 mantissa bits) while
 *       it is staged in LDS, and each product block is six v_mfma_f32_32x32x16_bf16 (a1b1 a1b2 a2b1 a1b3 a3b1 a2b2, each
 *       bf16 x bf16 product exact in the fp32 accumulator; dropped terms are below 2^-24 |a||b|).  Measured error
 *       against fp64 is not larger than the fp32 instruction's (tests/test_gpu_engine.py::test_math_modes_accuracy).
 *   LVT_MATH_F32 "f32": plain v_mfma_f32_32x32x2_f32.                                                          */
#define LVT_MATH_F32   (1 << 16)
/*   LVT_MATH_F16X2 "f16x2": every fp32 operand element a is scaled by an exact power of two s taken from the operand's
 *       max |a| (max |a| s in [2^14, 2^15): no overflow, no rounding) and split while it is staged in LDS into two fp16
 *       terms, a s = hi + 2^-11 lo (hi = RN16(a s); the residual is exact in fp32; lo = RN16(2^11 residual): 22 bits + sign,
 *       full for every element within 2^-27 of the max, degrading gradually below).  A block is three
 *       v_mfma_f32_32x32x16_f16: hi hi into one fp32 accumulator, hi lo + lo hi into a second one that enters with weight
 *       2^-11; the dropped lo lo term is <= 2^-22 |a||b| (2^-24.6 rms), the size of one fp32 rounding; every fp16 x fp16
 *       product is exact in fp32.  Half the matrix instructions of bf16x3.  The caller supplies the operands' max |.| as
 *       DEVICE scalars (any upper bound is safe; lvt_amax computes one, the engine's outputs can report theirs through
 *       c_amax / lvt_amax_io.c so that no extra pass is needed along a chain of launches).  Entry points without an
 *       f16x2 path (decode-time kernels) ignore the flag and use bf16x3; the VQ search and the flash attention kernels have
 *       f16x2 forms of their own (one scale per row).
 *       GUARANTEED ENVELOPE (tests/test_gpu_engine.py::test_f16x2_envelope_below_2_pow_minus_27, down to 2^-40 of the max): an
 *       element a of an operand whose bound is A (max |.| <= A < 2 max |.| after the power-of-two rounding) enters every product
 *       with an absolute error <= max(2^-22 |a|, 2^-50 A).  Full fp32-class precision for everything within 2^-28 of the
 *       operand's max; below that the relative error of an element grows as 2^-50 A / |a| (2^-10 at |a| = 2^-40 A) and an
 *       element below 2^-50 A is lost.  Kernels that keep the low term UNSCALED (the frame-resident weight gradient,
 *       csrc/conv_wgrad.hip; the flash attention kernels, per row): absolute error <= max(2^-22 |a|, 2^-39 A).            */
#define LVT_MATH_F16X2 (1 << 18)
/* max |.| of the operands / result of one engine launch (device pointers; see LVT_MATH_F16X2).  a / b: inputs, required in
 * f16x2 mode, ignored otherwise.  c: optional in EVERY mode -- the launch folds max |C| into *c with an integer atomic max
 * on the bit pattern (exact and order-independent for non-negative floats), so *c must be zeroed (or hold a bound to keep)
 * before the launch; not produced by split-K launches (their consumers are optimizers, not GEMMs).                       */
typedef struct { const float *a; const float *b; float *c; } lvt_amax_io;
/* *out = max(*out, max_i |x[i]|) over the FINITE entries of x: the stand-alone form for tensors that no engine launch
 * produced.  Every max |.| of this interface skips inf / nan entries: such an element makes the products it takes part in
 * non-finite through its own fp16 high term, exactly where the reference's result is non-finite, and leaves the scale -- and
 * with it every other row and column of the result -- untouched.                                                          */
int lvt_amax(const float *x, long long n, float *out, void *stream);
/* The same for many tensors in one launch per 64 (the weights of a model, once per pass); `entries` is a HOST array.        */
typedef struct { const float *x; long long n; float *out; } lvt_amax_entry;
int lvt_amax_multi(const lvt_amax_entry *entries, int n, void *stream);
/* *out = max(*out, *a, *b) (b may be NULL): the bound of an operand that spans two tensors (batched launches whose batch
 * strides are address differences, lvt_gemm_desc).                                                                        */
int lvt_amax_merge(const float *a, const float *b, float *out, void *stream);
/* lvt_vq_nearest only: coarse-then-exact search (one bf16 MFMA pass + exact re-evaluation of the codes inside the error band;
 * same exact argmin).  Faster on well-separated codebooks, slower on degenerate ones: opt-in, see csrc/vq.hip.            */
#define LVT_VQ_COARSE  (1 << 17)

/* ---- epilogue flags shared by GEMM / conv ------------------------------------------------------ */
#define LVT_EPI_BIAS        1   /* + bias[n]                                                  */
#define LVT_EPI_RESIDUAL    2   /* + res[m][n]                                                 */
#define LVT_EPI_RELU        4   /* max(.,0)                                                    */
#define LVT_EPI_TANH        8   /* tanhf(.)                                                    */
#define LVT_EPI_MASK       16   /* * (mask[m][n] > 0)   (ReLU backward with the saved output)  */
#define LVT_EPI_ACCUM      32   /* C += result                                                 */
#define LVT_EPI_PLANES     64   /* lvt_gemm_f32 only: C is a bf16 image and receives the result as its EXACT 3-way bf16 split
                                 * (v = p1 + p2 + p3, round-to-nearest-even at every level): plane j of element (m, n) at
                                 * ((uint16_t *)C)[j*c_plane + z-offset + m*ldc + n]; ldc, sC_*, c_plane count bf16 elements.
                                 * The operand format of lvt_attn_fwd_planes / lvt_attn_bwd_planes.                          */
/* causal structure of the batched attention products of a masked layer (M, N, K token positions of one block):          */
#define LVT_CAUSAL_KMAX   (1 << 8)    /* A(m,k) == 0 for k > m: a tile reduces over k < m0 + 128 only  (dQ = dS K)        */
#define LVT_CAUSAL_KMIN   (1 << 9)    /* A(m,k) == 0 for k < m: a tile starts its reduction at k = m0  (dV = P^T dO, dK)  */
#define LVT_CAUSAL_TILE   (1 << 10)   /* C(m,n) is only consumed for n <= m: tiles above the diagonal are written as 0   */

/* ---- general batched fp32 GEMM  (torch addmm / bmm / matmul / linear: K14,K18,K20,K22-K25) -------
 * C[z][m][n] = epilogue( alpha * sum_k A[z](m,k) * B[z](k,n) ),  z = zo*batch_inner + zi.
 *   ta == 0: A(m,k) at A[m*lda + (k/a_kb)*a_skb + k%a_kb]   (k contiguous, optional 2-level k)
 *   ta == 1: A(m,k) at A[k*lda + m]                         (m contiguous; M % 4 == 0)
 *   tb == 0: B(k,n) at B[n*ldb + (k/b_kb)*b_skb + k%b_kb]   (k contiguous)
 *   tb == 1: B(k,n) at B[k*ldb + n]                         (n contiguous; N % 4 == 0)
 * K % 4 == 0, all leading dimensions % 4 == 0, pointers 16-byte aligned.
 * splits > 1 partitions K; partial sums go to `workspace` and are reduced deterministically
 * (epilogue flags other than ACCUM are then not allowed).                                         */
typedef struct {
    int M, N, K;
    int ta, tb;
    const float *A; long long lda; int a_kb; long long a_skb;
    const float *B; long long ldb; int b_kb; long long b_skb;
    float *C;       long long ldc;
    int batch_outer, batch_inner;
    long long sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
    float alpha;
    int flags;
    const float *bias;
    const float *res;  long long ldr;   /* batch strides of res / mask follow C's */
    const float *mask; long long ldm;
    int splits;
    /* ta == 1 with splits > 1 only (weight gradient dW = dY^T X): when non-NULL receives the M column sums of A
     * (= sum over the K rows of dY: the bias gradient), accumulated from the A tiles the kernel streams anyway;
     * batched launches write (batch, M).  The batch strides sA / sB / sC are plain element offsets: two unrelated
     * weight gradients of the same shape run as ONE launch with sA_i = A1 - A0 etc. (half the k ranges, twice as long). */
    float *a_colsum;
    long long c_plane;                  /* LVT_EPI_PLANES: distance between the bf16 planes of C (elements)                 */
    const float *a_amax, *b_amax;       /* LVT_MATH_F16X2: device scalars >= max |A|, max |B| (required in that mode)        */
    const float *a_amax2, *b_amax2;     /* optional second bounds: the operand scale comes from max(*x_amax, *x_amax2) -- for
                                         * operands that span two tensors (batch strides = address differences)           */
    float *c_amax;                      /* optional, any mode: max |C| is folded into *c_amax (see lvt_amax_io)              */
} lvt_gemm_desc;
size_t lvt_gemm_workspace_bytes(const lvt_gemm_desc *d);
int lvt_gemm_f32(const lvt_gemm_desc *d, void *workspace, size_t workspace_bytes, void *stream);

/* ---- nearest code of ONE wide codebook (ABI 600; CODEBOOK.NUM == 1, vidgen/modeling/meta_arch/vqvae.py:25-27, vq_utils.py:13-20) -
 * scores[r][k] = x_r . e_k (rows x K, row pitch ld; made by lvt_gemm_f32 from the (rows, D) inputs and the (K, D) codebook):
 * idx[r] = argmax_k(scores[r][k] - |e_k|^2 / 2) = argmin_k |x_r - e_k|^2, lowest k on exact ties (torch.min).  K <= 2048.
 * The product quantiser (num codebooks of 64 dims) has its own fused search: lvt_vq_nearest.                              */
int lvt_vq_argmax_scores(const float *scores, long long rows, int K, long long ld, const float *codebook, int D,
                         long long *idx, void *stream);

/* ---- plane-fed GEMM of the f16x2 arithmetic (ABI 600, csrc/gemm_p2.hip; replaces the same torch linear / bmm products as
 * lvt_gemm_f32: vt_attention.py:120-128,138 and their autograd backward) ------------------------------------------------------
 * A "P2 image" of a matrix X[rows][K] (K % 32 == 0) is what the f16x2 arithmetic stages in LDS, made ONCE and kept in HBM:
 *     X s = hi + 2^-11 lo,   hi = RN16(X s),  lo = RN16(2^11 (X s - hi)),   s = the power of two of LVT_MATH_F16X2 taken
 * from a device scalar >= max |X| (the SAME scalar must be handed to every consumer as a_amax / b_amax).  Layout: row-major
 * with the fp32 matrix's footprint (row pitch ld floats = 4 ld bytes); each group of 32 consecutive k of a row is one 128-byte
 * line [fp16 hi of the 32 elements | fp16 lo of the 32 elements].  ld % 32 == 0, base 128-byte aligned.
 * lvt_gemm_p2_f32:  C[z][m][n] = epi(alpha sum_k A[z](m,k) B[z](n,k))   -- both operands k-contiguous ("NT").
 *   B is a P2 image (rows = n); A is a P2 image when a_planes != 0, else plain fp32 addressed as lvt_gemm_f32's ta == 0 operand
 *   (two-level k through a_kb / a_skb allowed).  P2 operands go global memory -> LDS by LDS-DMA (no registers, no vector ALU);
 *   an fp32 A is split in the kernel as in lvt_gemm_f32.  Results are bit-identical to lvt_gemm_f32 in LVT_MATH_F16X2 mode on
 *   the fp32 matrices the images were made from.  Epilogue flags: BIAS, RESIDUAL, RELU, MASK.  Cp (nullable): the result is
 *   ALSO written as a P2 image (row pitch ldcp, same batch offsets as C) under the scale of *cp_amax, a device scalar that must
 *   bound max |C| a priori (an overflowing element becomes inf, like an fp16 cast).  c_amax as in lvt_gemm_f32.                 */
typedef struct {
    int M, N, K;
    const void *A; long long lda; int a_planes; int a_kb; long long a_skb;
    const void *B; long long ldb;
    float *C; long long ldc;
    int batch_outer, batch_inner;
    long long sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
    float alpha;
    int flags;
    const float *bias;
    const float *res;  long long ldr;
    const float *mask; long long ldm;
    const float *a_amax, *b_amax;
    float *c_amax;
    void *Cp; long long ldcp; const float *cp_amax;
} lvt_gemm_p2_desc;
int lvt_gemm_p2_f32(const lvt_gemm_p2_desc *d, void *stream);
/* P2 images of up to any number of fp32 matrices, 32 entries per launch (`entries` is a HOST array): src [rows][cols] with row pitch
 * ld_src; transpose == 0: image rows = src rows, k = src columns; transpose == 1: image rows = src columns, k = src rows (the
 * image of src^T).  The image's k extent must be a multiple of 32; ld_dst % 32 == 0; scale from *amax (device scalar).        */
typedef struct { const float *src; void *dst; int rows, cols; long long ld_src, ld_dst; int transpose; const float *amax;
                 int batch; long long bs_src, bs_dst; } lvt_p2_pack_entry;
/* batch > 1: the entry stands for `batch` matrices of the same shape, src + z bs_src floats -> dst + z bs_dst floats (one scale):
 * e.g. the packed (3 na, d, da) q/k/v weights -> one (3 na da, d) image with transpose = 1, bs_src = d da, bs_dst = da d.        */
int lvt_p2_pack_multi(const lvt_p2_pack_entry *entries, int n, void *stream);

/* Small-M variant (a few rows, e.g. one token per sample in incremental decoding): same addressing as
 * lvt_gemm_f32 with ta == 0, a single batch level (A shared, B += z*sB, C += z*sC), flags BIAS|RESIDUAL|RELU.
 * One workgroup per 64 rows x 32 columns; M > 64 needs the k-contiguous layout (tb == 0, K % 8 == 0).        */
int lvt_gemm_smallm_f32(int M, int N, int K, int tb, const float *A, long long lda, const float *B,
                        long long ldb, float *C, long long ldc, int batch, long long sB, long long sC,
                        float alpha, int flags, const float *bias, const float *res, long long ldr,
                        const int *pos, long long c_pos, long long r_pos, void *stream);
/* `pos` (NULL, or a device int) is a row cursor read by the kernel: C += pos[0]*c_pos and res += pos[0]*r_pos.  With it
 * the launch arguments of a decode step do not depend on the position being decoded, so ONE captured hipGraph is
 * replayed for every position (autoregressive/incremental.py); the same three arguments on the split-K form below. */
/* Split-K form of the small-M product for long reductions (K >= 1024: the FFN down-projection, the attention output
 * projection and the wide predictor layers of a decode step).  K is cut into `splits` equal ranges (each a multiple
 * of 8) that run as independent workgroups; the partial tiles go through `workspace` and are added in split order by
 * a second launch that also applies the epilogue (deterministic).  tb == 0 layout only, N % 4 == 0.            */
size_t lvt_gemm_smallm_splitk_workspace_bytes(int M, int N, int splits);
int lvt_gemm_smallm_splitk_f32(int M, int N, int K, int splits, const float *A, long long lda, const float *B,
                               long long ldb, float *C, long long ldc, float alpha, int flags, const float *bias,
                               const float *res, long long ldr, const int *pos, long long c_pos, long long r_pos,
                               void *workspace, size_t workspace_bytes, void *stream);
/* The same partial products without the reduction launch (workspace = [splits][M][N] raw partial tiles), and the
 * LayerNorm that consumes them:  x = sum_s partials[s] (+ bias) (+ res);  x_out = x;  y = LN(x) * w + b.
 * In a decoder layer both products that end in a residual feed a LayerNorm (vt_attention.py:121,138), so the split-K
 * reduction rides on a launch that exists anyway.  One wave per row, d % 4 == 0, d <= 1024, splits summed in order. */
int lvt_gemm_smallm_partial_f32(int M, int N, int K, int splits, const float *A, long long lda, const float *B,
                                long long ldb, void *workspace, size_t workspace_bytes, void *stream);
int lvt_splitsum_layernorm_fwd(const float *partials, int splits, int rows, int d, const float *bias, const float *res,
                               long long ldr, float *x_out, float eps, const float *w, const float *b, float *y,
                               void *stream);

/* ---- 3-D convolution family, channels-last  (torch conv2d / conv3d / conv_transpose2d: K1-K6,K16) --
 * Geometry of the *forward* convolution  y[n,to,ho,wo,co] = sum x[n,to*st-pt+kt, ...,ci] w[co,ci,kt,kh,kw].
 * Ci / Co are the channel counts of the device buffers (multiples of 4; a 3-channel image is carried
 * zero-padded to 4).  Packed weights wp are [Kt*Kh*Kw][Ci][Co] (lvt_conv3d_pack_weight).
 * A ConvTranspose(in=a,out=b) layer is the bwd_data of the conv with Ci=b, Co=a (same weight tensor). */
typedef struct {
    int N, Ti, Hi, Wi, Ci;
    int To, Ho, Wo, Co;
    int Kt, Kh, Kw;
    int st, sh, sw;
    int pt, ph, pw;
} lvt_conv_geom;
/* w: [Co_real][Ci_real][Kt][Kh][Kw] (torch layout)  ->  wp: [taps][Ci][Co], zero padded.            */
int lvt_conv3d_pack_weight(const lvt_conv_geom *g, const float *w, int Ci_real, int Co_real,
                           float *wp, void *stream);
/* Many packs in one launch (`entries` is a HOST array; 64 per launch): kind 0 = lvt_conv3d_pack_weight, 1 = _pack_weight_t, 2 =
 * _pack_weight_phases, 3 = _pack_weight_parity of the (Co_real, Ci_real, taps) weight `w` into `dst` (same layouts, same bits);
 * taps = Kt*Kh*Kw (16 for kinds 2, 3), Ci / Co the padded channel counts of the geometry.                                  */
typedef struct { const float *w; float *dst; int kind, taps, Ci, Co, Ci_real, Co_real; } lvt_pack_entry;
int lvt_conv3d_pack_weights_multi(const lvt_pack_entry *entries, int n, void *stream);
/* Weight tiles as ready LDS images (ABI 610; f16x2 arithmetic, frame-resident kernels only).  A packed weight wp[rows][cols] of any
 * of the packs above (rows % 32 == 0, cols % 128 == 0) gets, RIGHT BEHIND its rows * cols floats in the same buffer,
 * lvt_conv3d_weight_image_bytes(rows, cols) bytes holding every 32 x 128 tile as the two fp16 planes of LVT_MATH_F16X2 under the
 * weight's own scale (`amax`: the device scalar the consuming launch also gets as lvt_amax_io.b), in the order the kernel keeps
 * them in LDS.  lvt_conv3d_fwd / lvt_conv3d_fwd_parity / lvt_conv3d_bwd_data_phases called with LVT_CONV_WEIGHT_IMAGE in `flags`
 * then stage the weight tiles by LDS-DMA (global memory -> LDS, no registers, no split arithmetic) instead of splitting the fp32
 * tile in every workgroup: same bits in LDS, bit-identical results.  The flag is ignored by launches that do not run on the
 * frame-resident kernels or not in f16x2 mode (the fp32 pack in front of the image stays valid for them).
 * Replaces nothing of the reference on its own: it is a staging format of the conv2d weights (vidgen/modeling/encoder/resencoder.py:
 * 25-76, generator/resdecoder.py:25-75).  `entries` is a HOST array; 64 per launch.                                          */
#define LVT_CONV_WEIGHT_IMAGE (1 << 21)
typedef struct { const float *wp; int rows, cols; const float *amax; } lvt_weight_image_entry;
size_t lvt_conv3d_weight_image_bytes(int rows, int cols);     /* 0 when rows % 32 or cols % 128 */
int lvt_conv3d_weight_images(const lvt_weight_image_entry *entries, int n, void *stream);
/* Packed weights of the convolution that IS the backward-data pass of a stride-1 convolution: channels swapped, taps
 * reversed: wt[taps-1-tap][co][ci] = w[co][ci][tap].  `lvt_conv3d_fwd` on the swapped geometry (Ci <-> Co, same kernel
 * and padding k-1-p) with these weights computes dx; for 3x3 / pad 1 layers of 16x16 frames that launch runs on the
 * frame-resident kernel (lvt_conv3d_uses_patch_kernel), which stages every input element once instead of once per tap. */
int lvt_conv3d_pack_weight_t(const lvt_conv_geom *g, const float *w, int Ci_real, int Co_real,
                             float *wt, void *stream);
int lvt_conv3d_uses_patch_kernel(const lvt_conv_geom *g, int flags);
/* The 4x4 / stride 2 / pad 1 convolution 32x32 -> 16x16 (Ci % 32 == 0, Co % 128 == 0) on the frame-resident kernel: the 16
 * taps are 4 parity classes x 4 taps, each class a 2x2 / stride 1 convolution on a 17x17 sub-image that is staged once per
 * 32-channel chunk.  Weights: lvt_conv3d_pack_weight_parity, wq[(py,px)][(a,b)][ci][co] = w[co][ci][2a+py][2b+px].        */
int lvt_conv3d_fwd_uses_parity_kernel(const lvt_conv_geom *g, int flags);
int lvt_conv3d_pack_weight_parity(const lvt_conv_geom *g, const float *w, int Ci_real, int Co_real,
                                  float *wq, void *stream);
int lvt_conv3d_fwd_parity(const lvt_conv_geom *g, const float *x, const float *wq, const float *bias,
                          const float *res, const float *mask, float *y, int flags, const lvt_amax_io *ax,
        void *stream);
/* y = epi( conv(x, wp) ); bias[Co]; res / y are (N,To,Ho,Wo,Co).  flags: BIAS|RESIDUAL|RELU|TANH|MASK */
int lvt_conv3d_fwd(const lvt_conv_geom *g, const float *x, const float *wp, const float *bias,
                   const float *res, const float *mask, float *y, int flags, const lvt_amax_io *ax,
        void *stream);
/* The same pass for the 4x4 / stride 2 / pad 1 layers between 32x32 and 16x16 frames (Co % 32 == 0, Ci % 128 == 0) on the
 * frame-resident kernel: every output phase (py, px) reads four taps of the ONE staged 18x18 patch of dy.  Weights packed
 * per phase by lvt_conv3d_pack_weight_phases: wph[(py,px)][(a,b)][co][ci] = w[co][ci][3-py-2a][3-px-2b].               */
int lvt_conv3d_bwd_data_uses_phase_kernel(const lvt_conv_geom *g, int flags);
int lvt_conv3d_pack_weight_phases(const lvt_conv_geom *g, const float *w, int Ci_real, int Co_real,
                                  float *wph, void *stream);
int lvt_conv3d_bwd_data_phases(const lvt_conv_geom *g, const float *dy, const float *wph, const float *bias,
                               const float *res, const float *mask, float *dx, int flags, const lvt_amax_io *ax,
        void *stream);
/* dx = epi( conv_transpose(dy, wp) ); res / mask / dx are (N,Ti,Hi,Wi,Ci).
 * flags: BIAS (bias[Ci], used when this IS a ConvTranspose forward) | RESIDUAL | RELU | TANH | MASK.
 * Requires Kt % st == 0 etc. and To*st == Ti-ish geometries produced by lvt_conv geometry helpers.  */
int lvt_conv3d_bwd_data(const lvt_conv_geom *g, const float *dy, const float *wp, const float *bias,
                        const float *res, const float *mask, float *dx, int flags, const lvt_amax_io *ax,
        void *stream);
/* dw[Co_real][Ci_real][Kt][Kh][Kw] = sum_pixels x (*) dy, deterministic split-K through workspace.
 * db (nullable, Co_real floats) = sum_pixels dy: the bias gradient, accumulated from the dy tiles the kernel streams
 * anyway (no second pass over dy).                                                                         */
size_t lvt_conv3d_bwd_weight_workspace_bytes(const lvt_conv_geom *g);
/* 1 when lvt_conv3d_bwd_weight can also produce db for this geometry: every served geometry (since ABI 400 the frame-
 * resident weight-gradient kernels of csrc/conv_wgrad.hip -- 3x3 / pad 1 layers of 16x16 frames and the 4x4 / stride 2 layer
 * with 256 channels on one side -- sum the dy rows / patches they stage as well).  Kept for callers written against 300.  */
int lvt_conv3d_bwd_weight_fuses_bias(const lvt_conv_geom *g, int flags);
/* In `flags` of lvt_conv3d_bwd_weight / lvt_conv3d_bwd_weight_fuses_bias: db (Ci_real floats) = column sums of X instead of dy --
 * the bias gradient of a TRANSPOSED convolution, whose weight gradient is this call with the operands swapped (x = the gradient of
 * the layer's output).  Served where the query answers 1 (the 4x4 / stride 2 frame-resident kernel: 32x32 <-> 16x16 frames, 256
 * channels on the small side); elsewhere use lvt_colsum(x).                                                                  */
#define LVT_WGRAD_DB_OF_X (1 << 20)
/* flags: 0, LVT_MATH_F32 or LVT_MATH_F16X2 (ax->a = max |x|, ax->b = max |dy|; ax may be NULL otherwise).  The
 * frame-resident kernel keeps the fp16 low term UNSCALED in f16x2 mode (one accumulator set for nine taps): full 22-bit
 * operands within 2^-16 of the operand's max, gradually fewer bits below (csrc/conv_wgrad.hip).                          */
int lvt_conv3d_bwd_weight(const lvt_conv_geom *g, const float *x, const float *dy, float *dw, float *db,
                          int Ci_real, int Co_real, int flags, const lvt_amax_io *ax, void *workspace,
                          size_t workspace_bytes, void *stream);
/* Image-side layer (3 channels carried as 4).
 * lvt_convt4_fwd: ConvTranspose2d(Ci -> Cr<=3, k4 s2 p1) forward, x (N,Hi,Wi,Ci) -> y (N,2Hi,2Wi,4) (+tanh);
 *                 w in torch layout (Ci, Cr, 4, 4), bias (Cr).   (K6 of resdecoder.py:56,68)
 * flags: 0 / LVT_MATH_F32: an LDS-tiled fp32 FMA kernel that reads the activation exactly once instead of spending 32-wide
 * MFMA tiles on padding columns (ax may be NULL).  LVT_MATH_F16X2 (ax->a = max |x|, ax->b = max |w| required) with Ci == 128,
 * Hi % 8 == 0, Wi % 32 == 0: the matrix cores -- rows = input positions, 16 columns = (output phase, channel), reduction over
 * the 3 x 3 neighbours x Ci on v_mfma_f32_16x16x32_f16, persistent workgroups, weights split once per workgroup.          */
int lvt_convt4_fwd(const float *x, const float *w, const float *bias, int N, int Hi, int Wi, int Ci, int Cr,
                   int act_tanh, float *y, int flags, const lvt_amax_io *ax, void *stream);
/* out[n] (+)= sum_m g[m*ld + n]  (bias gradients).  workspace >= lvt_colsum_workspace_bytes.        */
size_t lvt_colsum_workspace_bytes(long long M, int N);
int lvt_colsum(const float *g, long long M, int N, long long ld, float *out, void *workspace,
               size_t workspace_bytes, void *stream);

/* ---- product vector quantiser (vidgen/modeling/vq/vq_utils.py:5-65, vq_embedding.py:9-99; K7-K9) ----
 * z: [rows][ldz] channels-last activations, group g owns columns [g*D, (g+1)*D).  codebooks: [num][KC][D].
 * idx: int64 [rows/P][num][P]  (== the reference's (N, num, H, W) layout with P = H*W).
 * lvt_vq_nearest: idx = argmin_k |e_k|^2 + |x|^2 - 2 x.e_k, lowest k on ties (torch.min).  Three arithmetics, by `flags`:
 *   LVT_MATH_F16X2 (the default mode of the Python side): argmax_k x.e_k - |e_k|^2 / 2 -- the same argmin, |x|^2 dropped --
 *     on two fp16 planes of the codebook group (one power-of-two scale per group) and of every row (one scale per row), three
 *     fp16 MFMAs per 16 dims, the whole group LDS-resident, indices written directly; no workspace needed;
 *   neither math flag (bf16x3): the reference's distance form on the bf16 matrix cores, one workgroup per codebook half, the
 *     per-half (distance, index) candidates go through `workspace`;
 *   LVT_MATH_F32, or bf16x3 without a workspace: the fp32-MFMA kernel with the whole codebook LDS-resident.
 * All three agree with an fp64 search wherever the top-2 distance gap exceeds 1e-5 (|x|^2 + max |e|^2); which code a closer
 * pair resolves to is an artefact of the fp32 evaluation order in the reference as well.                       */
size_t lvt_vq_nearest_workspace_bytes(long long rows, int num, int KC);
int lvt_vq_nearest(const float *z, long long rows, int ldz, int num, int D, int KC,
                   const float *codebooks, long long *idx, int P, int flags, void *workspace, size_t workspace_bytes,
                   void *stream);
/* out[row][g*D+d] = codebooks[g][idx][d]   (index_select / embedding: z_q_st, z_q_bar, mode "emb")   */
int lvt_vq_gather(const long long *idx, const float *codebooks, long long rows, int num, int D, int KC,
                  int P, float *out, int ldo, void *stream);
/* stats[num][KC][D+1]: per-code sum of assigned rows and (last column) their count.  LDS-private
 * accumulation per (group, row chunk) + fixed-order chunk reduction: no atomics, bit-reproducible.
 * Kept separate from finalize so that a data-parallel all-reduce of `stats` can sit in between.      */
size_t lvt_vq_ema_workspace_bytes(long long rows, int num, int D, int KC);
int lvt_vq_ema_accumulate(const long long *idx, const float *z, long long rows, int ldz, int num, int D,
                          int KC, int P, float *stats, void *workspace, size_t workspace_bytes,
                          void *stream);
/* running_size[num][KC], running_sum/weight[num][KC][D] updated in place (vq_embedding.py:48-59).    */
int lvt_vq_ema_finalize(const float *stats, int num, int D, int KC, float decay, float eps,
                        float *running_size, float *running_sum, float *weight, void *stream);

/* ---- boundary layout conversion + input (de)normalisation (ae.py:32-37,151-168; K12) --------------
 * to_channels_last : in [B][C][R] -> out [B][R][ldo] (columns >= C zero); mode 1: (x - a[c]) / s[c]
 * to_channels_first: in [B][R][ldi] -> out [B][C][R];  mode 2: clamp(x * s[c] + a[c], lo, hi)          */
int lvt_to_channels_last(const float *in, int B, int C, long long R, int ldo, int mode, const float *a,
                         const float *s, float *out, float *out_amax, void *stream);
/* (out_amax, here and on lvt_mse_bwd / lvt_tanh_bwd: nullable, max |out| folded into a device scalar as lvt_amax_io.c is --
 *  these tensors are operands of engine launches.)                                                                        */
int lvt_to_channels_first(const float *in, int B, int C, long long R, int ldi, int mode, const float *a,
                          const float *s, float lo, float hi, float *out, void *stream);

/* ---- losses / glue (F.mse_loss: K11) ----------------------------------------------------------------*/
size_t lvt_reduce_workspace_bytes(void);
/* out[0] = scale * sum((a-b)^2) / denom, fixed summation order                                        */
int lvt_mse_fwd(const float *a, const float *b, long long n, double denom, float scale, float *out,
                void *workspace, size_t workspace_bytes, void *stream);
/* out = add + gout[0] * (2*scale/denom) * (a-b) [* (1-a^2) if tanh_of_a]; gout/add may be NULL        */
int lvt_mse_bwd(const float *a, const float *b, long long n, double denom, float scale,
                const float *gout_dev, const float *add, int tanh_of_a, float *out, float *out_amax, void *stream);
/* the same pair for F.l1_loss (ABI 600; LOSS.PIXEL.MODE "l1", vidgen/modeling/loss/loss.py:11-12): scale * sum |a - b| / denom, and
 * out = add + g (scale / denom) sign(a - b) (0 where a == b, as torch)                                    */
int lvt_l1_fwd(const float *a, const float *b, long long n, double denom, float scale, float *out,
               void *workspace, size_t workspace_bytes, void *stream);
int lvt_l1_bwd(const float *a, const float *b, long long n, double denom, float scale,
               const float *gout_dev, const float *add, int tanh_of_a, float *out, float *out_amax, void *stream);
int lvt_tanh_bwd(const float *g, const float *y, long long n, float *out, float *out_amax, void *stream);
/* out = alpha * alpha_dev[0] * x (+ add)                                                              */
int lvt_axpy(const float *x, const float *add, long long n, const float *alpha_dev, float alpha,
             float *out, void *stream);
/* x[r][:] += table[r % P][:]  (3-D sinusoidal position signal, vt_attention.py:25-50; K17)            */
int lvt_add_periodic(float *x, const float *table, long long rows, int P, int d, void *stream);

/* ---- LayerNorm over the last dim, eps inside the sqrt (F.layer_norm; K19) ---------------------------*/
int lvt_layernorm_fwd(const float *x, long long rows, int d, float eps, const float *w, const float *b,
                      float *y, float *mean, float *rstd, float *y_amax, const float *w_amax, const float *b_amax,
                      void *stream);
/* (y_amax / dx_amax, nullable: max |y| resp. max |dx| folded into a device scalar as lvt_amax_io.c is.  With w_amax and
 *  b_amax -- device scalars >= max |w|, max |b| -- the forward STORES the bound max |w| sqrt(d - 1) + max |b| into *y_amax
 *  instead of reducing: |(x - mean) rstd| <= sqrt(d - 1) on every row.)                                                   */
/* The same forward that ALSO writes y as a P2 image `yp` (ABI 600; row pitch d floats, d % 32 == 0) under the scale of the
 * a-priori bound it stores into *y_amax: w_amax, b_amax and y_amax are required.  The image feeds lvt_gemm_p2_f32 (a_planes). */
int lvt_layernorm_fwd_p2(const float *x, long long rows, int d, float eps, const float *w, const float *b,
                         float *y, void *yp, float *mean, float *rstd, float *y_amax, const float *w_amax,
                         const float *b_amax, void *stream);
size_t lvt_layernorm_bwd_workspace_bytes(int d);
/* dx = LN'(dy) (+ add); dw[d], db[d] reduced in a fixed order                                        */
int lvt_layernorm_bwd(const float *dy, const float *x, const float *mean, const float *rstd,
                      const float *w, long long rows, int d, const float *add, float *dx, float *dw,
                      float *db, float *dx_amax, void *workspace, size_t workspace_bytes, void *stream);

/* ---- attention softmax with learned relative-position bias (vt_attention.py:59-81,142-174; K21,K22) ----
 * scores (B,H,S,S) in place:  softmax_j( s/temper + (dt[h][di_t] + dh[h][di_h]) + dw[h][di_w] ), with the
 * causal fill (`fill` where j > i, the reference uses -1e4) when masked != 0.  Banks: dt (H,2bt-1) etc.,
 * S == bt*bh*bw, S % 64 == 0, S <= 1024.                                                              */
int lvt_attn_softmax_fwd(float *scores, int B, int H, int S, float temper, const float *dt,
                         const float *dh, const float *dw, int bt, int bh, int bw, int masked, float fill,
                         void *stream);
/* dP (B,H,S,S) is overwritten with dS = P*(dP - sum_j P dP)/temper; ddt/ddh/ddw receive the bank
 * gradients (batch-reduced in a fixed order through the (H,S,S) scratch G).                           */
int lvt_attn_softmax_bwd(const float *P, float *dP, int B, int H, int S, float temper, int bt, int bh,
                         int bw, float *G, float *ddt, float *ddh, float *ddw, void *stream);

/* fused attention of one 256-token block (ScaledDotProductAttention, vt_attention.py:52-81):
 * P = lvt_attn_softmax_fwd(q k^T) and o = P v in one launch; q/k/v/o token-major (B*S rows, H*da columns, head h in
 * columns h*da..), P (B,H,S,S) is written for the backward pass.  S == 256, da == 128.                    */
int lvt_attn_fwd(const float *q, const float *k, const float *v, int B, int H, int S, int da, float temper,
                 const float *dt, const float *dh, const float *dw, int bt, int bh, int bw, int masked, float fill,
                 float *P, float *o, void *stream);

/* ---- fused attention on PRE-SPLIT operands (csrc/attention_pipe.hip) ------------------------------------------------
 * q, k, v (and dO) arrive as the exact 3-way bf16 split written by lvt_gemm_f32 with LVT_EPI_PLANES: operand x, plane j,
 * element (token row m, column c) at ((uint16_t *)qkv_planes)[x*operand_stride + j*plane_stride + m*(H*da) + c], x = 0 (q),
 * 1 (k), 2 (v); do_planes has the layout of one operand.  Same arithmetic as lvt_attn_fwd (the planes ARE the fp32 values),
 * software-pipelined: staging is a copy, transposed operands come from ds_read_b64_tr_b16, the softmax is online per key
 * chunk beside the MFMAs of the next one.  S == 256, da == 128 and a block geometry with an instantiation
 * (lvt_attn_planes_supported: (1,16,16) and (4,8,8)).
 * forward : P (B,H,S,S) and o (B*S, H*da) fp32, as lvt_attn_fwd.
 * backward: two launches over the saved P -- (A) dS = P o (dO V^T - rowsum(dO o O)) / temper, dQ = dS K and the per-(sample,
 *           head, query half) bias-bank sums; (B) dV = P^T dO, dK = dS^T Q -- and a fixed-order reduction of the bank sums:
 *           dq / dk / dv (B*S, H*da) fp32, ddt (H, 2bt-1), ddh (H, 2bh-1), ddw (H, 2bw-1).  Replaces the four batched
 *           GEMMs + lvt_attn_softmax_bwd of the unfused path (vt_attention.py:59-81 under autograd).  The workspace holds dS. */
int lvt_attn_planes_supported(int S, int da, int bt, int bh, int bw);
int lvt_attn_fwd_planes(const void *qkv_planes, long long plane_stride, long long operand_stride, int B, int H, int S, int da,
                        float temper, const float *dt, const float *dh, const float *dw, int bt, int bh, int bw, int masked,
                        float fill, float *P, float *o, float *o_amax, void *stream);
/* (o_amax / d_amax, nullable: max |o| resp. max over dq, dk, dv folded into a device scalar as lvt_amax_io.c is -- the
 *  outputs feed engine launches.)                                                                                          */
size_t lvt_attn_bwd_planes_workspace_bytes(int B, int H, int S, int bt, int bh, int bw);
int lvt_attn_bwd_planes(const void *qkv_planes, long long plane_stride, long long operand_stride, const void *do_planes,
                        const float *P, const float *o, int B, int H, int S, int da, float temper, int bt, int bh, int bw,
                        int masked, float *dq, float *dk, float *dv, float *ddt, float *ddh, float *ddw, float *d_amax,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- flash-style fused attention on fp32 operands, f16x2 arithmetic (csrc/attention_flash.hip; ABI 500) --------------
 * Replaces bmm(q, k^T) / temper + B (+ masked_fill) -> softmax -> bmm(attn, v) of ScaledDotProductAttention
 * (vt_attention.py:59-81, bias B of BlockLocalAttention.get_B, :169-174) and its autograd backward WITHOUT materialising
 * the (B,H,S,S) attention matrix: q / k / v / d_o / o / dq / dk / dv are token-major fp32 (B*S rows, row stride `ld` floats,
 * head h in columns h*da ..; q, k, v may be the three (B*S, H*da) slabs of the packed projection output).
 * forward : o, and `stats` (2, B*H*S) fp32: row max m of the biased (masked) scores, then 1 / sum_j exp(score - m).
 * backward: dq, dk, dv and the bank gradients ddt (H, 2bt-1), ddh (H, 2bh-1), ddw (H, 2bw-1); P and dS are recomputed from
 *           q, k, v, d_o and `stats` in both backward launches (workspace: one float per row + the per-workgroup bank sums).
 *           `o` (ABI 510): the forward's output (row stride ld).  With it the softmax-backward row term delta_i = sum_j P_ij dP_ij
 *           is taken as dO_i . O_i and the query-stationary launch makes ONE pass over the keys (three score-sized products
 *           instead of five); o == NULL keeps the two-pass form, which derives delta from the dP values it recomputes.
 * Operands are split in-kernel into two fp16 terms under an exact power-of-two scale PER ROW (token x head, 128 values):
 * 22 bits + sign for every element within 2^-16 of its row's max |.|, absolute error <= 2^-39 of the row max below that;
 * three fp16 MFMAs per product, fp32 accumulation.  S == 256, da == 128, (bt,bh,bw) in {(1,16,16), (4,8,8)}, B*H % 8 == 0.
 * o_amax / d_amax: nullable, as for lvt_attn_fwd_planes.                                                                 */
int lvt_attn_flash_supported(int S, int da, int bt, int bh, int bw);
int lvt_attn_fwd_flash(const float *q, const float *k, const float *v, long long ld, int B, int H, int S, int da,
                       float temper, const float *dt, const float *dh, const float *dw, int bt, int bh, int bw,
                       int masked, float fill, float *o, float *stats, float *o_amax, void *stream);
size_t lvt_attn_bwd_flash_workspace_bytes(int B, int H, int S, int bt, int bh, int bw);
int lvt_attn_bwd_flash(const float *q, const float *k, const float *v, const float *d_o, long long ld, const float *stats,
                       const float *o, int B, int H, int S, int da, float temper, const float *dt, const float *dh, const float *dw,
                       int bt, int bh, int bw, int masked, float fill, float *dq, float *dk, float *dv, float *ddt,
                       float *ddh, float *ddw, float *d_amax, void *workspace, size_t workspace_bytes, void *stream);

/* single-query attention against a token-major K/V cache (incremental sampling: the reference re-runs the
 * whole causal decoder for every generated pixel, vt.py:121-131).  q (B rows of H*da, row stride ldq), o (B, H*da), caches (B, S, H*da);
 * attends keys 0..qi with the same scale / bias-bank rule as lvt_attn_softmax_fwd.  da == 128.
 * With `pos` != NULL the query position is the device int pos[0] (clamped to [0, S)) instead of `qi`, and the query
 * rows start at q + pos[0]*q_pos: position-independent launch arguments for hipGraph replay.            */
int lvt_attn_decode(const float *q, long long ldq, const float *Kc, const float *Vc, int B, int H, int S, int da, int qi,
                    float temper, const float *dt, const float *dh, const float *dw, int bt, int bh, int bw,
                    float *o, const int *pos, long long q_pos, void *stream);

/* integer plumbing of a decode step driven by a device-side cursor (the reference indexes python ints: vt.py:121-131).
 * codes (rows, S1) int64 with S1 = S + 1: the slice being decoded, one always-padded extra slot per row.
 * gather: out[r][j] = codes[r][nb[pos[0]][j]] for the `taps` causal-conv neighbours of the current position (nb is
 *         (S, taps) int64, entries in [0, S]); commit: codes[r][pos[0]] = drawn[r] (drawn may be NULL), then pos[0] += 1. */
int lvt_decode_gather_codes(const long long *codes, const long long *nb, const int *pos, int rows, int S1, int taps,
                            long long *out, void *stream);
int lvt_decode_commit(const long long *drawn, int rows, int S1, long long *codes, int *pos, void *stream);

/* categorical draw per row from logits / temp with caller-supplied uniforms u[row] in [0,1) (the reference draws
 * with torch.multinomial on softmax(logit / temp), videotransformer.py:176-181): code = #{ j : cdf_j <= u * total },
 * clamped to V-1, written as int64 at out[row * out_stride]; `probs` (rows, V) is optional.  V <= 1024.
 * With `pos` != NULL the uniforms are read at u + pos[0]*u_pos: a table of draws for every position of a slice,
 * filled once per slice, indexed by the device-side cursor of the decode graphs (no generator inside a graph).  */
int lvt_sample_categorical(const float *logits, long long rows, int V, float temp, const float *u,
                           long long *out, long long out_stride, float *probs, const int *pos, long long u_pos,
                           void *stream);

/* ---- embedding bags: the one-hot Conv3d / Embedding sums / one-hot Linear inputs as gathers (K13,K15,K25)
 * out[b*P+pos][:] = bias + btable[bindex[b]] + sum_s table[tab_row[s] + idx[b*bstride + off[s] + pos]][:]
 * (negative indices, i.e. PAD_VALUE, contribute nothing).                                             */
int lvt_embbag_fwd(const long long *idx, long long bstride, int P, long long rows, int nslots,
                   const int *slot_off, const int *tab_row, const float *table, int D, const float *bias,
                   const float *btable, const long long *bindex, float *out, void *stream);
/* gradient of the tables: out[(s*V + code)][n] = sum_rows [idx(row,s) == code] * dout[row][n], idx(row,s) =
 * idx[b*bstride + off[s] + pos*pstride], row = b*P + pos.  Tables of >= 512 rows with N in {64,128,256,512} are
 * summed by a gather (one wave per output row adds its rows in ascending order: exact fp32 sums in a fixed order, no
 * atomics, flags' math mode and dout_amax unused); the others, and every call carrying LVT_ONEHOT_DENSE, run as a
 * transposed one-hot GEMM on the matrix cores (deterministic split-K).  Indices outside [0, V) contribute nothing. */
#define LVT_ONEHOT_DENSE (1 << 19)
size_t lvt_onehot_tn_workspace_bytes(int nslots, int V, int N, long long rows);
/* 1 when lvt_onehot_tn_gemm sums this shape by the gather (callers that account matrix-core work ask)  */
int lvt_onehot_tn_is_gather(int nslots, int V, int N, long long ldb, const float *dout, int flags);
int lvt_onehot_tn_gemm(const long long *idx, int nslots, int V, const int *slot_off, long long bstride,
                       long long pstride, int P, long long rows, const float *dout, long long ldb, int N,
                       float *out, int flags, const float *dout_amax, void *workspace, size_t workspace_bytes,
                       void *stream);
/* (dout_amax: max |dout| as a device scalar, LVT_MATH_F16X2 only; the one-hot operand is exact in fp16 as it is.)     */
/* out (n0,n1,n2) contiguous <- in[i0*s0 + i1*s1 + i2*s2]  (weight re-layouts)                         */
int lvt_permute3(const float *in, long long s0, long long s1, long long s2, int n0, int n1, int n2,
                 float *out, void *stream);

/* out[b][i][:] = x[b][perm[i]][:] for (B, S, d) tokens, perm (S) int64 on the device: the block-split regrouping of
 * BlockLocalAttention (vt_attention.py:189-200) and, with the inverse permutation, its backward.  d % 4 == 0.        */
int lvt_row_gather(const float *x, const long long *perm, long long B, int S, int d, float *out, void *stream);

/* ---- subscale slice / context builder for a batch of code clips (DatasetMapper.prepare_slices,
 * vidgen/data/dataset_mapper.py:113-149 with vt_utils.py:24-57,104-128; the reference runs it per sample in CPU
 * data-loader workers).  video (B,T,nc,H,W) int64, abc (B,3) int32 slice offsets on the device.  Outputs:
 * ctx (B,nc,Tc,Hc,Wc) with Xc = 2*(kx/2) + (X/sx - 1)*sx + 1 -- the clip masked to the slices generated before (a,b,c)
 * and shifted so that a (kt,kh,kw)/(st,sh,sw) conv is centred on the slice's first element, `pad_value` elsewhere;
 * slice (B,nc,T/st,H/sh,W/sw); slice_idx (B) = raster index of (a,b,c); ignore (B,1,T/st,H/sh,W/sw) bytes =
 * frame < n_prime.                                                                                       */
int lvt_slice_context(const long long *video, int B, int T, int nc, int H, int W, const int *abc, int st, int sh,
                      int sw, int kt, int kh, int kw, int n_prime, long long pad_value, long long *ctx,
                      long long *slice, long long *slice_idx, unsigned char *ignore, void *stream);

/* ---- cross entropy with ignore_index (F.cross_entropy; vt.py:305-313; K26) ---------------------------
 * rows = B*P rows of V logits; target(b,pos) = target[b*tstride_b + pos*tstride_pos].
 * loss[0] = scale * mean_{non-ignored}(lse - logit[target]); count[0] = #non-ignored.                  */
size_t lvt_xent_workspace_bytes(void);
int lvt_xent_fwd(const float *logits, const long long *target, long long tstride_b, long long tstride_pos,
                 int P, long long rows, int V, long long ignore, float scale, float *row_loss, float *lse,
                 float *loss, float *count, void *workspace, size_t workspace_bytes, void *stream);
int lvt_xent_bwd(const float *logits, const long long *target, long long tstride_b, long long tstride_pos,
                 int P, long long rows, int V, long long ignore, const float *lse, const float *count,
                 const float *gout, float scale, float *dlogits, float *dl_amax, void *stream);
/* (dl_amax, nullable: receives |gout * scale / count|, an a-priori bound of max |dlogits| since |softmax - onehot| <= 1) */

/* ---- fused multi-tensor optimizer steps (torch.optim.Adam / RMSprop as configured by
 * vidgen/solver/build.py:46-74; the reference's one-param-group-per-parameter layout costs hundreds of
 * launches per step).  `entries` is a HOST array; up to 64 tensors travel by value per launch.          */
typedef struct {
    float *p; const float *g; float *s0; float *s1;   /* param, grad, state 0, state 1 (device pointers) */
    long long n; float lr; float wd;
} lvt_opt_entry;
/* s0 = exp_avg, s1 = exp_avg_sq; step = 1-based step count (bias correction).                           */
int lvt_adam_step(const lvt_opt_entry *entries, int n, float beta1, float beta2, float eps, int step,
                  void *stream);
/* s0 = square_avg, s1 = momentum_buffer.                                                                */
int lvt_rmsprop_step(const lvt_opt_entry *entries, int n, float alpha, float eps, float momentum,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LVT_HIP_H */
