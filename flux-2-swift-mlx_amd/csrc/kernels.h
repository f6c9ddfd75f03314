// kernels.h — launchers of the gfx950 kernels (defined in *.hip).  Every launcher
// enqueues on `stream` and returns 0 on success (1 = bad argument, 2 = launch error).
#pragma once
// Development switches (A/B runs, ablations, diagnostics) are environment variables read through F2_DEV_ENV.  They
// exist only in the development library (`make dev` -> libflux2mi_dev.so, -DFLUX2MI_DEV); in the product library the
// macro is a null pointer constant, every switch folds to its default and no environment variable can change a result.
#ifdef FLUX2MI_DEV
#include <stdlib.h>
#define F2_DEV_ENV(name) getenv(name)
#else
#define F2_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------- GEMM (gemm.hip)
enum GemmEpilogue {
    EPI_F32 = 0,        // C f32  = alpha*acc
    EPI_BF16 = 1,       // C bf16 = alpha*acc
    EPI_GATE_RESID = 2, // C f32 += vec[col]*acc                (x += gate * f(y))
    EPI_SWIGLU = 3,     // C2 bf16[row, h] = silu(gate)*value   (interleaved weight rows)
    EPI_QKV_MLP = 4,    // cols < n_split: EPI_BF16 into C; cols >= n_split: EPI_SWIGLU into C2
    EPI_BIAS_BF16 = 5,  // C bf16 = alpha*acc + vec[col] (+ resid[row,col])
    EPI_CONV = 6        // implicit-GEMM convolution (GemmProblem::conv_*): C bf16 NHWC = acc + vec[col] (+ resid), rows remapped to pixels
};

struct GemmProblem {
    const uint16_t* A;      // bf16 [M,K], row stride lda
    const uint16_t* W;      // bf16 [N,K], row stride ldw
    void* C;
    void* C2;
    const float* vec;       // gate or bias (f32 [N]) or nullptr
    const uint16_t* resid;  // EPI_BIAS_BF16: optional bf16 residual [M,N], row stride ldr
    int M, N, K;
    int lda, ldw, ldc, ldc2, ldr;
    int n_split;
    float alpha;
    int epi;
    // fp8 (OCP e4m3) operands: A and W are bytes (lda/ldw in elements = bytes), K % 128 == 0,
    // acc is multiplied by a_scale[row] * w_scale[col] in the epilogue.  nullptr/0 for bf16.
    const float* a_scale;
    const float* w_scale;
    int fp8;
    // fp8, gemm256s: A quantised in 2-D blocks instead of per row — one e8m0 byte (value = 2^(byte - 127)) per 16 consecutive rows x
    // 128 consecutive k (= one K-tile): a_bscale[kt * a_bscale_ld + (row >> 4)], row = the problem's A row.  The bytes ride in the
    // block-scaled MFMA's scale operand (no epilogue factor; a_scale is not read).  Written by the producing epilogues (SwiGLU hidden,
    // attention output) or by launch_quantize_blocks_fp8.  a_bscale 4-byte aligned, a_bscale_ld % 4 == 0 and >= ceil(M / 16) + 12.
    const uint8_t* a_bscale;
    int a_bscale_ld;
    // fp8, gemm256s, EPI_SWIGLU / EPI_QKV_MLP: the SwiGLU hidden leaves in that block-scaled form (the A operand of mlp_out / linear2)
    // instead of bf16 C2: bytes c2_q8[row * c2_q8_ld + h], scale bytes c2_bs[(h / 128) * c2_bs_ld + row / 16] (row, h: this problem's
    // row and hidden column; a tile's 128 hidden columns x 16 rows are one block, its exponent taken from max |gate x value| >= the
    // block's maximum).  N - n_split % 256 == 0, c2_q8 8-byte aligned, c2_q8_ld % 8 == 0.  nullptr = bf16 C2.
    uint8_t* c2_q8;
    int c2_q8_ld;
    uint8_t* c2_bs;
    int c2_bs_ld;
    // 256x256 kernel, EPI_BF16 columns [0, qk_cols) (= q | k of a fused qkv projection, qk_cols = 2*dim, dim % 256 == 0):
    // per-head RMSNorm (eps 1e-6, weights qk_qw / qk_kw bf16 [128]) + RoPE applied to the f32 accumulators before
    // the store — what launch_qk_norm_rope_vt does in a separate pass over the stored bf16 values; q additionally
    // times qk_qmul.  The rotation angles are computed in the epilogue from the positions qk_pos f32 [rows, 4]
    // (launch_rope_tables' third output; table row = qk_row0 + row) and qk_l2t16 = log2(theta) / 16: v_sin / v_cos
    // on pos * theta^(-k/16) / 2pi, no table traffic.  nullptr = off.
    // 256x256 kernel, EPI_BF16: inside every aligned group of 16 output columns, column 8a+4h+b is stored at 8h+4a+b
    // (the key order of the attention's V^T operand, attention.hip).  With A = W_v [dim, K], W = the activation
    // [S, K] and C = vt [dim, s_pad] the GEMM writes V^T directly: no V round trip, no transpose kernel.
    int col_perm16;
    // 128x128 kernel, single-problem launches with few tiles and a long K (the text encoders' M = 512 projections): split-K.
    // splitk_ws (>= gemm_splitk_ws_bytes(M, N, K) bytes of device scratch) lets launch_gemm cut K into `splits` ranges, one
    // workgroup each per tile; the partial accumulators go to the workspace and a second kernel adds them IN SPLIT ORDER and
    // applies the epilogue (EPI_F32 / EPI_BF16 / EPI_GATE_RESID / EPI_BIAS_BF16): deterministic, no atomics.  nullptr = off.
    float* splitk_ws;
    size_t splitk_ws_bytes;
    const uint16_t* qk_qw;
    const uint16_t* qk_kw;
    const float* qk_pos;
    int qk_cols, qk_row0;
    float qk_qmul, qk_l2t16;
    // fused QK norm + RoPE epilogue, fp8 models: instead of the bf16 q | k columns of C, write their e4m3 copies for the attention's
    // block-scaled Q K^T (attention64b.hip, QK8): qk8 bytes [rows][qk8_ld], column = the GEMM column (q | k), row = qk_row0 + row; values
    // times qk8_qmul / qk8_kmul (powers of two).  nullptr = the bf16 store.
    uint8_t* qk8;
    int qk8_ld;
    float qk8_qmul, qk8_kmul;
    // gemm256s, EPI_CONV: a 3x3 (or the 2x2 of an upsample phase) convolution over a ZERO-HALOED NHWC image as a plain GEMM.
    // The image is stored [(H + 2)][Wp = W + 2][Cin] with a zero border; GEMM row r = padded pixel q = Wp + 1 + r (first interior pixel
    // on), M = H*Wp - 2, lda = Cin, A = the padded image's base (+ the phase's (py*Wp + px) pixels).  For tap (ky, kx) every row reads
    // the SAME byte offset (ky*Wp + kx)*Cin*2 from its own address, so with K ordered (ky, kx, c) — the OHWI weight rows as they are —
    // the A source of K-tile kt is  row + kt*128 + (kt / conv_seg) * conv_row_jump:  conv_seg = taps_x * Cin/64 K-tiles per tap row,
    // conv_row_jump = (Wp - taps_x) * Cin * 2 bytes.  Rows whose pixel is a halo column compute garbage and are masked in the epilogue.
    // Output pixel of row r: (y, x) = (q / Wp - 1, q % Wp - 1), valid when 0 <= x < conv_w; conv_up: (2y + conv_py, 2x + conv_px) of a
    // 2H x 2W image.  conv_out_padded 0: C / resid are plain [Ho][Wo][N]; 1: C is zero-haloed [(Ho + 2)][Wo + 2][N] (halo columns inside
    // the row range are written as zeros; the first / last halo row are the caller's), resid stays plain.
    int conv_seg;
    long long conv_row_jump;
    int conv_wp, conv_w, conv_up, conv_py, conv_px, conv_out_padded;
    // EPI_CONV, optional: GroupNorm statistics of the OUTPUT from the epilogue.  Row (tile_m of this problem) of gn_part f32
    // [rows][N][2] receives, per output channel, (sum, sum of squares) of the tile's valid pixels — the f32 values in front of their bf16
    // rounding, lanes / waves combined in a fixed order through LDS.  gn_part has ceil(M / gemm256s_conv_tile_rows()) rows, every one
    // written; launch_groupnorm_finalize_channels turns it into the affine: the consumer's GroupNorm then needs no pass over the
    // tensor for its statistics.  nullptr = off.
    float* gn_part;
};
// Problems of one launch: callers batch at most two (the text / image halves of a double block); the launcher's own split-K form
// (gemm.hip) expands them into up to GEMM_MAX_PROBS K ranges, each a problem of its own.
constexpr int GEMM_MAX_PROBS = 8;
struct GemmBatch {
    GemmProblem p[GEMM_MAX_PROBS];
    int nprob;
    int tile_start[GEMM_MAX_PROBS];   // first workgroup of problem i (tile_start[0] = 0)
    int flags;    // gemm256: bit 0 = 16-byte bf16 epilogue stores
    unsigned long long* trace;   // gemm256: per-wave time stamps (dev tool), nullptr normally
    int splits;   // 128x128 kernel: K ranges per tile (1 = whole K in one workgroup)
};
int launch_gemm(hipStream_t stream, const GemmProblem* probs, int nprob);
// GEMM rows per tile launch_gemm will use for these EPI_CONV problems: GemmProblem::gn_part has ceil(M / rows) rows per problem
int gemm256s_conv_tile_rows(const GemmProblem* probs, int nprob);
// Workspace launch_gemm wants in GemmProblem::splitk_ws to split the K dimension of a bf16 [M,K] x [N,K]^T product (0 = it would not split)
size_t gemm_splitk_ws_bytes(int M, int N, int K, bool fp8 = false);

// ---------------------------------------------------------------- attention (attention.hip)
// Joint (text+image) non-causal attention, head_dim 128, bf16 in/out, f32 softmax.
//   q, k: bf16, row s at (q + s*ldqk + head*128); vt: bf16 [heads][128][s_pad] with the
//   key-permuted layout produced by launch_qk_norm_rope_vt; out row s at out + s*ldo + head*128.
// Sq query rows starting at q / out, S keys starting at k / column 0 of vt (Sq == S: plain self-attention;
// the KV-cache paths launch it per query range, and with S = the key prefix a range may see).
// score_bound_log2 > 0: the caller guarantees |q.k| / sqrt(128) * log2(e) <= score_bound_log2 for every
// query/key pair (q, k out of the per-head RMSNorm); the softmax then uses it as a fixed offset instead
// of tracking row maxima (used when <= 40, or <= 80 with q_prescaled).  0 = no bound known.
// q_prescaled: Q already carries the factor log2(e)/sqrt(128) (launch_qk_norm_rope_vt's q_mul), so the scores
// come out of the MFMA in the log2 domain; with a bound the offset is then taken as ZERO and p = exp2(s)
// needs no multiply-add at all (p in [2^-80, 2^80]: inside f32 / bf16 range with room for the sums, any offset cancels in O / l).
// true when launch_attention takes the fixed-offset softmax for this score bound (log2 domain; 0 = no bound known)
bool attention_uses_fixed_offset(float score_bound_log2, bool q_prescaled = true);
// split_ws (optional, attention64b_split_ws_bytes(n_cu) bytes, n_cu = the device's CU count): lets the one-wave-per-SIMD kernel cut the
// items of a partial last round along the keys into f equal pieces each (attention64b.hip, AttnSplit: fixed-offset softmax, heads % 8
// == 0, a tail of at most half the CUs — Klein-9B's 576 items and Dev's 864 on 256 CUs, not Klein-4B's 432); nullptr = never.
size_t attention64b_split_ws_bytes(int n_cu);
int attention64b_split_factor(int Sq, int S, int heads, int n_cu);   // pieces per tail item (1 = whole items)
int launch_attention(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt,
                     int s_pad, uint16_t* out, int ldo, int Sq, int S, int heads, float score_bound_log2 = 0.f, bool q_prescaled = false,
                     float* split_ws = nullptr, int n_cu = 0);
static inline int attn_s_pad(int S) { return ((S + 63) / 64) * 64; }
// The blocks' attention with e4m3 Q and K (fp8 models; attention64b.hip, QK8): q8 / k8 bytes [S][ld8] (head h at column h*128),
// q8 = q 2^q_exp with q pre-scaled as above, k8 = k 2^k_exp (written by launch_qk_norm_rope_vt); S^T = K Q^T is one block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 block, the scales undone by its e8m0 operands; P, V, O stay bf16 / f32.
// data_offset as launch_attention's softmax forms.  1 = not this kernel's case.
// out8 != nullptr: O leaves as block-scaled e4m3 instead of bf16 (GemmProblem::a_bscale of the consuming proj / linear2): bytes
// out8[row * ld8o + head * 128 + d], scale bytes bs[head * bs_ld + row / 16] (16 queries x one head's 128 columns per block); `out` unused.
int launch_attention64b_qk8(hipStream_t stream, const uint8_t* q8, const uint8_t* k8, int ld8, int q_exp, int k_exp, const uint16_t* vt, int s_pad,
                            uint16_t* out, int ldo, int Sq, int S, int heads, bool data_offset, uint8_t* out8 = nullptr, int ld8o = 0,
                            uint8_t* bs = nullptr, int bs_ld = 0, int n_cu = 0, float* split_ws = nullptr);

// ---------------------------------------------------------------- norms / rope (norm.hip)
// y bf16[row,:] = LN(x f32[row,:]) * (1 + scale) + shift ; rows < split use (shift0, scale0),
// rows >= split use (shift1, scale1).  eps 1e-6, no affine (Flux2TransformerBlock.swift:56-61).
int launch_layernorm_modulate(hipStream_t stream, const float* x, uint16_t* y, int rows, int dim, int split,
                              const float* shift0, const float* scale0, const float* shift1,
                              const float* scale1);
// fp8 variant: y is OCP e4m3 bytes [rows, dim] with a per-row scale (y_scale[row] = amax/448).
int launch_layernorm_modulate_fp8(hipStream_t stream, const float* x, uint8_t* y, float* y_scale, int rows, int dim,
                                  int split, const float* shift0, const float* scale0, const float* shift1,
                                  const float* scale1);
// Register-only MFMA stream (mfma_stream.hip): the matrix-pipe rate this card sustains at its power cap; diagnostic.
int launch_mfma_stream(hipStream_t stream, float* out, int workgroups, bool fp8, double seconds, double* tflops);
// Row-wise dynamic quantisation bf16 [rows, cols] (row stride ld) -> e4m3 [rows, cols] (stride ldq) + scale[rows].
// x bf16 [rows, cols] -> e4m3 in 16 x 128 blocks with power-of-two scales: q [rows, cols] bytes + bscale[(col / 128) * lds + row / 16]
// (GemmProblem::a_bscale; dev.h fp8_block_exp).  cols % 128 == 0, lds >= ceil(rows / 16)
int launch_quantize_blocks_fp8(hipStream_t stream, const uint16_t* x, int ld, uint8_t* q, int ldq, uint8_t* bscale, int lds, int rows, int cols);
int launch_quantize_rows_fp8(hipStream_t stream, const uint16_t* x, int ld, uint8_t* q, int ldq, float* scale,
                             int rows, int cols);
// In place on qkv bf16 [S, 3*dim] (q | k | v): per-head RMSNorm (eps 1e-6, learned scale)
// then RoPE on q and k; V is written transposed + key-permuted into vt [heads][128][s_pad].
// Rows < split use (qscale0, kscale0) — the text stream of a double block — rows >= split
// use (qscale1, kscale1).  q_mul multiplies Q (not K) before its rounding to bf16: the DiT folds the
// attention's score scale log2(e)/sqrt(128) in here and calls launch_attention with q_prescaled.
int launch_qk_norm_rope_vt(hipStream_t stream, uint16_t* qkv, int ldqkv, int S, int heads, int split,
                           const uint16_t* qscale0, const uint16_t* kscale0, const uint16_t* qscale1,
                           const uint16_t* kscale1, const float* cos_tab, const float* sin_tab,
                           uint16_t* vt, int s_pad, float q_mul = 1.0f,
                           uint8_t* qk8 = nullptr, int q8_exp = 0, int k8_exp = 0);
// V half of the above only (standalone attention op).  v_col0: first V column of a row (default: 2 d, the fused [q | k | v] row;
// 0 for a V matrix of its own).
int launch_v_transpose_only(hipStream_t stream, const uint16_t* qkv, int ldqkv, int S, int heads, uint16_t* vt,
                            int s_pad, int v_col0 = -1);
// cos/sin f32 [S,128] from position ids i32 [S,4] (Flux2RoPE.swift:129-169); rows [0,s_txt)
// come from txt_ids, the rest from img_ids.
// Copy the V^T columns of keys [key0, key0 + nkeys) between the attention operand vt
// ([heads*128][s_pad], keys permuted inside groups of 16) and a compact key-ordered cache
// [heads*128][nkeys]; to_cache = 1: vt -> cache, 0: cache -> vt.  (TransformerKVCache.swift)
int launch_vt_keys_copy(hipStream_t stream, uint16_t* vt, int s_pad, uint16_t* cache, int heads, int key0, int nkeys,
                        int to_cache);
// ---- text encoder (Qwen3, FluxTextEncoders/Model/Qwen3/*.swift)
// y bf16 [rows, dim] = x f32 * rsqrt(mean(x^2) + eps) * w   (RMSNorm with a bf16 weight)
int launch_rmsnorm_rows(hipStream_t stream, const float* x, const uint16_t* w, uint16_t* y, int rows, int dim, float eps);
// In place on qkv bf16 [S, (hq + 2 hkv) * 128] = q | k | v: per-head RMSNorm(eps, weight) of q and k (skipped
// when q_w == nullptr: Mistral has no q/k norm), then
// RoPE in the split-half convention (MLXFast.RoPE traditional: false; pairs (i, i + 64)), position = row,
// inv_freq_i = theta^(-2i/128).  V is written transposed + key-permuted into vt [hkv][128][s_pad].
int launch_qk_norm_rope_llm_vt(hipStream_t stream, uint16_t* qkv, int S, int hq, int hkv, const uint16_t* q_w,
                               const uint16_t* k_w, float eps, float theta, uint16_t* vt, int s_pad);
// hs f32 [S, dim] = table bf16 [vocab, dim] rows ids[s] (ids on the device; out-of-range ids are an error flag)
int launch_embed_rows(hipStream_t stream, const int32_t* ids, const uint16_t* table, float* hs, int S, int dim, int vocab);
// out bf16 [S, ldo] columns [c0, c0 + dim) = hs f32 [S, dim]
int launch_f32_rows_to_bf16_cols(hipStream_t stream, const float* hs, uint16_t* out, int S, int dim, int ldo, int c0);
// Causal attention over a padded sequence with grouped KV heads: real tokens are rows [pad_lo, valid_len);
// a real query row i of head h attends to keys pad_lo <= j <= i, j < valid_len of KV head
// h / (heads / kv_heads); rows >= valid_len (right padding) see the real keys, rows < pad_lo (left
// padding) attend uniformly to keys 0..i (see attention.hip).  Operand layouts as launch_attention.
int launch_attention_causal(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt,
                            int s_pad, uint16_t* out, int ldo, int S, int pad_lo, int valid_len, int heads, int kv_heads);
// pos_tab (optional): f32 [S, 4], the ids as floats (the fused GEMM epilogue's input)
int launch_rope_tables(hipStream_t stream, const int32_t* txt_ids, int s_txt, const int32_t* img_ids, int s_img,
                       float* cos_tab, float* sin_tab, float* pos_tab = nullptr);

// ---------------------------------------------------------------- small ops (elementwise.hip)
// out f32[n] (+)= sum_k W[n,k] * f(x[k]); f = silu when silu_in; one wave per output row.
int launch_gemv(hipStream_t stream, const uint16_t* W, const float* x, float* out, int N, int K, int silu_in,
                int accumulate);
// emb f32[256] = [cos(1000 t f_i) | sin(1000 t f_i)] (Flux2Embeddings.swift:27-44)
// V <= 32 vectors x + v xs -> out + v os through launch_gemv's arithmetic (identical results), the weights read once
int launch_gemv_multi(hipStream_t stream, const uint16_t* W, const float* x, long xs, float* out, long os, int N, int K, int V, int silu_in,
                      int accumulate);
int launch_sinusoid(hipStream_t stream, float value, float* emb);
int launch_silu_inplace(hipStream_t stream, float* x, int n);
int launch_f32_to_bf16(hipStream_t stream, const float* in, uint16_t* out, int64_t n);
int launch_convert_to_bf16(hipStream_t stream, const void* in, int dtype, uint16_t* out, int64_t n);
int launch_convert_to_f32(hipStream_t stream, const void* in, int dtype, float* out, int64_t n);
// out rows permuted: out[r,:] = in[perm(r),:], SwiGLU interleave of a [2h,K] matrix (see gemm.hip)
int launch_swiglu_interleave_rows(hipStream_t stream, const uint16_t* in, uint16_t* out, int hidden, int K);
// w[i] = bf16(f32(w[i]) + f32(delta[i]))  (LoRA merge, WeightLoader.swift:823-838)
// MLX random stream (Threefry-2x32): mode 0 raw words, 1 uniform [lo,hi), 2 standard normal
int launch_mlx_random(hipStream_t stream, uint32_t k0, uint32_t k1, float* out, int64_t n, int mode, float lo, float hi);
int launch_bf16_add_inplace(hipStream_t stream, uint16_t* w, const uint16_t* delta, int64_t n);
// quanto qint8: out_bf16[r, c] = bf16(f32(data[r, c]) * scale[r])  (WeightLoader.swift:347-363)
int launch_dequant_i8_rows(hipStream_t stream, const int8_t* data, const float* scale, uint16_t* out, int rows, int cols);
int launch_axpy(hipStream_t stream, float* x, const float* v, float a, int64_t n);                 // x += a*v
int launch_lerp(hipStream_t stream, float* out, const float* a, const float* b, float t, int64_t n); // (1-t)a + t b
int launch_cfg_combine(hipStream_t stream, float* vc, const float* vu, float g, int64_t n);
int launch_repaint_blend(hipStream_t stream, float* x, const float* img, const float* noise, const float* mask,
                         float sigma_next, int seq, int c);
// out[dims] = a (op) b, op 0 add / 1 sub / 2 mul; sa / sb: element strides per output dim, 0 where the operand broadcasts
int launch_bcast_binary(hipStream_t stream, int op, const float* a, const int64_t* sa, const float* b, const int64_t* sb, float* out,
                        const int* dims);
int launch_transpose_2d(hipStream_t stream, const float* in, float* out, int rows, int cols);      // [rows,cols]->[cols,rows]
int launch_latents_to_vae_input(hipStream_t stream, const float* packed, const float* mean, const float* var,
                                float* z, int h16, int w16);
int launch_vae_latents_to_packed(hipStream_t stream, const float* z, const float* mean, const float* var,
                                 float* packed, int h16, int w16);

// ---------------------------------------------------------------- VAE (vae_kernels.hip)
struct ConvArgs {
    const uint16_t* x;      // NHWC bf16 [Hin, Win, Cin]
    const uint16_t* w;      // bf16 [Cout_pad][ks*ks][Cin]  (OHWI)
    const float* bias;      // f32 [Cout]
    const uint16_t* resid;  // optional NHWC bf16 [Ho, Wo, Cout]
    uint16_t* y;            // NHWC bf16 [Ho, Wo, Cout] or nullptr
    float* y_f32_nchw;      // optional f32 [Cout, Ho, Wo]
    uint8_t* y_u8_hwc;      // optional u8 [Ho, Wo, Cout]: clip((v+1)*127.5, 0, 255) truncated
    int Hin, Win, Cin, Cout;
    int ks;                 // 1 or 3 (pad = ks/2)
    int upsample;           // 1: the conv reads a nearest-2x upsampled view of x (Ho = 2*Hin)
    int downsample;         // 1: stride 2, no top/left padding, one zero row/column at the bottom/right
                            //    (Downsample2D, ResnetBlock.swift:189-214): Ho = Hin/2, ks must be 3
    const uint16_t* w_phase;  // optional, with upsample = 1 and ks = 3: bf16 [4 phases][Cout_pad][2*2][Cin] from
                              // launch_upsample_phase_weights — the same convolution as four 2x2 convolutions on x (4/9 of the MACs)
};
int launch_conv2d(hipStream_t stream, const ConvArgs& a);
// A 3x3 convolution of a nearest-2x upsampled image is, per output phase (py, px) = (row & 1, col & 1), a 2x2 convolution of the
// image itself: the taps that land on the same source pixel add up.  w bf16 OHWI [cout_pad][3][3][cin] -> w_phase bf16
// [py*2+px][cout_pad][a*2+b][cin], tap (a, b) on source pixel (y + py - 1 + a, x + px - 1 + b); sums in f32, one rounding.
int launch_upsample_phase_weights(hipStream_t stream, const uint16_t* w, int cout_pad, int cin, uint16_t* w_phase);
// GroupNorm statistics of NHWC bf16 x -> per-channel affine (a_c, b_c) f32 with
// y = a_c*x + b_c == GN(x)*gamma + beta  (32 groups, eps 1e-6, f32 statistics; ResnetBlock.swift:9-55).
int launch_groupnorm_stats(hipStream_t stream, const uint16_t* x, int64_t pixels, int C, int groups,
                           const float* gamma, const float* beta, float eps, float* partial /* scratch */,
                           float* ab /* [2*C] */);
int groupnorm_partial_floats(int64_t pixels, int groups);
// The same affine from per-channel partial sums f32 [rows][C][2] (GemmProblem::gn_part, written by the producing convolution's
// epilogue): one workgroup per group, rows combined in f64 in a fixed order.  pixels = the tensor's pixel count.
int launch_groupnorm_finalize_channels(hipStream_t stream, const float* part, int rows, int64_t pixels, int C, int groups,
                                       const float* gamma, const float* beta, float eps, float* ab /* [2*C] */);
int launch_groupnorm_apply(hipStream_t stream, const uint16_t* x, uint16_t* y, int64_t pixels, int C,
                           const float* ab, int silu);
// the same, written as a zero-haloed image [(H + 2)][(W + 2)][C] (every pixel of the padded extent, the halo as zeros): the operand of
// the implicit-GEMM convolution (GemmProblem::conv_*)
int launch_groupnorm_apply_padded(hipStream_t stream, const uint16_t* x, uint16_t* y_padded, int H, int W, int C, const float* ab, int silu);
// The decoder's tail in one kernel: y = conv3x3(SiLU(a_c x + b_c)) to <= 3 channels (C = 128 or 96; w OHWI bf16 with >= 16 rows, rows >=
// cout zero), f32 NCHW and / or u8 HWC (clip((v + 1) 127.5), truncated).  Returns 1 when the shape is not this kernel's.
int launch_conv_out_fused(hipStream_t stream, const uint16_t* x, const float* ab, const uint16_t* w, const float* bias, int H, int W, int C,
                          int cout, float* y_f32_nchw, uint8_t* y_u8_hwc);
int launch_nchw_f32_to_nhwc_bf16(hipStream_t stream, const float* in, uint16_t* out, int C, int H, int W);
// same, output channel count C_pad >= C with zeros in the padding (conv_in of the encoder: 3 -> 32)
int launch_nchw_f32_to_nhwc_bf16_padded(hipStream_t stream, const float* in, uint16_t* out, int C, int C_pad, int H, int W);
// NHWC bf16 [P, C] channels [c0, c0+n) -> f32 [n, P] (NCHW)
int launch_nhwc_bf16_to_nchw_f32(hipStream_t stream, const uint16_t* in, float* out, int C, int c0, int n, int64_t P);
// OHWI [cout_pad][3][3][cin] -> [cout_pad][3][seg] (a kernel row's 3 * cin values, zero-padded to seg % 64 == 0): vae.cpp, convolutions with cin % 64 != 0
int launch_conv_w_taprow(hipStream_t stream, const uint16_t* w, int cout_pad, int cin, int seg, uint16_t* out);
int launch_softmax_rows(hipStream_t stream, const float* in, uint16_t* out, int rows, int cols);
// vae_attention.hip — the VAE mid block's single-head attention as ONE streaming kernel (no P x P scores in HBM); head dim 512 / 384.
// O [P, C] bf16 = softmax(Q K^T * scale) V + vbias; Q [P, C], K [ceil32(P), C] bf16 row-major (K's rows behind P: readable, any values),
// V^T [C][ldv] bf16 with ldv >= ceil32(P), ldv % 8 == 0 and
// FINITE values in the columns P .. ceil32(P) - 1; vbias f32 [C] or NULL.  0 ok / 1 unsupported / 2 launch error.
bool vae_attention_supported(int C);
int launch_vae_attention(hipStream_t stream, const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, int ldv, const float* vbias,
                         uint16_t* O, int P, int C, float scale);
int launch_transpose_bf16(hipStream_t stream, const uint16_t* in, uint16_t* out, int rows, int cols);
int launch_oihw_to_ohwi_bf16(hipStream_t stream, const uint16_t* in, uint16_t* out, int O, int I, int KH, int KW,
                             int O_pad, int I_pad = 0 /* 0 = I */);
