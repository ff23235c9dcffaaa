/*
 * quanto_hip.h - C ABI of libquanto_hip.so, the MI355X (gfx950 / CDNA4) backend for the
 * optimum-quanto QLinear hot path.
 *
 * Every entry point replaces one piece of the reference's per-call work (paths are relative to
 * /root/reference/optimum/quanto).  The reference binds native code through pybind11 torch
 * extensions loaded by library/extensions/extension.py:12-55; this library is the torch-free
 * equivalent: plain pointers, sizes and a HIP stream, so it can be bound from ctypes (what
 * optimum_quanto_amd/library does), from a pybind11 shim, or from C++ directly.
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller; the library never allocates,
 *    frees or copies device memory and keeps no state between calls.
 *  - Matrices are dense row-major.  Weights are [N, K] ("out_features, in_features"),
 *    activations [M, K], outputs [M, N].
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  The
 *    device that owns the buffers must be current in the calling thread (the Python shim
 *    wraps each call in torch.cuda.device(tensor.device)).
 *  - Every function returns QUANTO_HIP_OK (0) or a negative quanto_hip_status; nothing is
 *    written when an argument is rejected.  Launch failures surface as QUANTO_HIP_ELAUNCH
 *    with hipGetLastError() consumed.  Calls are asynchronous with respect to the host.
 */
#ifndef QUANTO_HIP_H_
#define QUANTO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QUANTO_HIP_ABI_VERSION 1

typedef enum {
  QUANTO_HIP_OK = 0,
  QUANTO_HIP_EINVAL = -1,   /* bad argument (null pointer, negative size, bits not in {2,4}, ...) */
  QUANTO_HIP_ENOTSUP = -2,  /* valid but unsupported dtype / layout combination                   */
  QUANTO_HIP_ELAUNCH = -3,  /* the HIP runtime rejected the launch                                 */
  QUANTO_HIP_EALIGN = -4    /* a pointer violates the documented alignment                         */
} quanto_hip_status;

typedef enum {
  QUANTO_HIP_F32 = 0,
  QUANTO_HIP_F16 = 1,
  QUANTO_HIP_BF16 = 2,
  QUANTO_HIP_I8 = 3,
  QUANTO_HIP_U8 = 4,
  QUANTO_HIP_F8_E4M3FN = 5,   /* OCP e4m3 - gfx950's native fp8 */
  QUANTO_HIP_F8_E5M2 = 6,
  QUANTO_HIP_F8_E4M3FNUZ = 7  /* MI300-era encoding; decoded in software */
} quanto_hip_dtype;

/* Kernel selection for the *_mm entry points.  AUTO is what product code uses; the others
 * exist so tests and benchmarks can pin one implementation. */
typedef enum {
  QUANTO_HIP_KERNEL_AUTO = 0,
  QUANTO_HIP_KERNEL_NAIVE = 1, /* one thread per output element, any shape                      */
  QUANTO_HIP_KERNEL_GEMV = 2,  /* weight-streaming kernel for M <= QUANTO_HIP_GEMV_MAX_M[_QBITS] */
  QUANTO_HIP_KERNEL_MFMA = 3,  /* LDS-tiled MFMA kernel, 128x128 tile (any M)                   */
  QUANTO_HIP_KERNEL_MFMA_LARGE = 4, /* 256x256 tile, LDS-DMA pipeline (prefill-sized M and N)    */
  QUANTO_HIP_KERNEL_SKINNY = 5, /* qbits_mm (int4; group sizes 128 / 64 / 32, per-channel scales) and qbytes_mm: weight-streaming MFMA kernel for
                                 * M <= QUANTO_HIP_SKINNY_MAX_M; K split over workgroups when a workspace is given */
  QUANTO_HIP_KERNEL_NATIVE8 = 6, /* qbytes_mm with quantized activations: int8 x int8 (i32 MFMA) / fp8 x fp8 (fp8 MFMA) */
  QUANTO_HIP_KERNEL_DEQUANT_MFMA = 7, /* qbits_mm, M beyond ~1-1.5 k rows (and formats the fused kernels do not take): fused dequantize into the
                                       * workspace + dense MFMA GEMM - multiplies the weight rounded to the activation dtype, as the reference does */
  QUANTO_HIP_KERNEL_MFMA_FUSED4 = 8,  /* qbits_mm, 64 < M <= ~1-1.5 k (AUTO: its own time model): packed int4 -> MFMA operands in registers,
                                       * per-group fp32 fold, no dequantized weight; workspace only when K is split (plan / workspace_size say so) */
  QUANTO_HIP_KERNEL_MMV = 9,          /* qbits_mm, 4 < M <= 16 (AUTO; the kernel itself accepts up to 32 rows): register-streaming MFMA kernel, K split over the waves of a block (no workspace) */
  QUANTO_HIP_KERNEL_MFMA_LARGE4 = 10  /* qbits_mm (int4, group size a multiple of 32 or per-channel), prefill-sized M: packed int4 -> registers -> MFMA
                                         operands with the reference's rounding sequence (the product multiplies exactly the reference's dequantized
                                         weight), 256 x 256 tiles, no workspace, no dense weight in memory */
} quanto_hip_kernel;

/* Split-K workspaces (SKINNY and MFMA_LARGE kernels) all share ONE layout: the first QUANTO_HIP_WS_COUNTER_BYTES bytes are
 * arrival counters - zero on entry, restored to zero by the kernel - and the fp32 partial sums start right behind them,
 * whatever the problem size.  A buffer whose counter region was zeroed once can therefore serve any sequence of calls on one
 * stream; a kernel never splits a problem that needs more counters than the region holds. */
#define QUANTO_HIP_WS_COUNTER_BYTES 4096

#define QUANTO_HIP_MAX_MULTI 4           /* Linears per quanto_hip_qbits_mm_multi call                        */
#define QUANTO_HIP_GEMV_MAX_M 8         /* qbytes_mm: rows of x the GEMV kernel accepts                    */
#define QUANTO_HIP_SKINNY_MAX_M 256      /* qbits_mm: rows of x the streaming MFMA kernel accepts (passes of 64) */
#define QUANTO_HIP_GEMV_MAX_M_QBITS 64  /* qbits_mm: ditto (passes of up to 8 rows; weights re-read from MALL) */
#define QUANTO_HIP_GEMV_F32_MAX_M 8     /* fp32 activations (r6, csrc/qmm_f32.hip): rows the fp32 weight-streaming kernel takes (passes of 4); beyond: fp32 MFMA tiles */
#define QUANTO_HIP_GEMV_MAX_M_OTHER 24  /* qbits_mm, formats / shapes the streaming kernel's 64-feature blocks do not fit (group sizes 96 / 64 / 32, per-channel, qint2): passes of 4 rows */

int quanto_hip_abi_version(void);
const char* quanto_hip_status_string(int status);

/* Name of the kernel the last successful *_mm / *_conv2d call on this thread dispatched to ("naive", "gemv", "mmv", "skinny", "skinny_multi",
 * "mfma", "mfma_large", "mfma_native8", "mfma_fused4", "mfma_large4", "dequant_mfma", "conv2d_mfma", "conv2d_mfma_int4", ...).  Used by tests to
 * assert that the intended path ran. */
const char* quanto_hip_last_kernel(void);

/* 0 when `stream` is not being captured into a hipGraph, otherwise the (positive) id of the capture sequence; negative
 * status on error.  Lets a binding keep one zero-initialised split-K workspace per capture: memory allocated and zeroed
 * while capturing belongs to that graph and must not be handed to eager launches (or to another capture). */
int64_t quanto_hip_stream_capture_id(void* stream);

/*
 * quanto::unpack(Tensor self, int bits) -> Tensor
 *   replaces library/extensions/hip/unpack.cu:33-97 (bound at library/extensions/hip/__init__.py:34-36),
 *   semantics of library/unpack.py:21-54.
 * packed: uint8[packed_numel]; unpacked: uint8[packed_numel * 8 / bits].
 * Plane i (i < 8/bits) of the output, i.e. unpacked[i*packed_numel + j], is
 * (packed[j] >> (bits*i)) & ((1<<bits)-1): the planes are concatenated along dim 0.
 */
int quanto_hip_unpack(const uint8_t* packed, uint8_t* unpacked, int64_t packed_numel, int bits, void* stream);

/*
 * Fused PackedTensor.unpack + QBitsDequantizer.forward + ungroup for axis-0 weights
 *   replaces tensor/packed.py:101-104 + tensor/qbits.py:27-49 + tensor/grouped.py:39-51
 *   (three elementwise passes and a 2x intermediate in the reference).
 * packed:  uint8[ceil(R / (8/bits)), C] generic PackedTensor layout of the grouped weight, where
 *          C = group_size (or K when group_size == 0, i.e. per-channel) and R = N*K/C grouped rows.
 * scale:   dtype[R]; shift: shift_dtype[R] with shift_dtype == dtype (float shift) or U8/I8
 *          (integer zero-point).
 * out:     dtype[N, K].  The rounding sequence is the reference's: float shift ->
 *          round(round(scale*q) - shift), zero-point -> round(scale*(q - zp)); the result is
 *          bit-identical to the reference CPU dequantize() in every dtype.
 */
int quanto_hip_dequantize_qbits(const uint8_t* packed, const void* scale, const void* shift, void* out,
                                int64_t N, int64_t K, int bits, int group_size, int dtype, int shift_dtype,
                                void* stream);

/*
 * quanto::qbits_mm - the fused product behind QLinear.forward for qint4/qint2 weights
 *   replaces QuantizedLinearFunction.forward (tensor/function.py:41-47) applied to
 *   QBitsDequantizer's output (tensor/qbits.py:27-49): y = x @ dequant(W).T (+ bias).
 *   The reference has no such op; its CUDA analogs are gemm_f16i4_awq / gemm_f16i4_marlin
 *   (library/extensions/cuda/__init__.py:82-121,170-202).
 * x: dtype[M, K]; packed/scale/shift as in quanto_hip_dequantize_qbits; bias: dtype[N] or NULL;
 * y: dtype[M, N].  dtype in {F32, F16, BF16}.  Accumulation is fp32.  Every kernel except DEQUANT_MFMA and MFMA_LARGE4 multiplies
 * the stored integers exactly and applies scale / shift to the fp32 accumulator (the dequantized weight exists nowhere).  The two
 * prefill kernels multiply the reference's DEQUANTIZED weight (its two roundings to dtype, bit-identical to QBitsDequantizer), as the
 * reference does per call: DEQUANT_MFMA - AUTO's choice for large M while weight and activations sit in the caches - writes it into the
 * caller's workspace (one pass, amortised over thousands of rows) and runs a dense GEMM; MFMA_LARGE4 builds the same values as MFMA
 * operands in registers (no workspace; AUTO beyond the Infinity Cache: N*K >= 8192^2 with M >= 8192, or N*K >= 3 * 8192^2, and whenever
 * workspace == NULL at prefill sizes).
 */
int quanto_hip_qbits_mm(const void* x, const uint8_t* packed, const void* scale, const void* shift,
                        const void* bias, void* y, int64_t M, int64_t N, int64_t K, int bits, int group_size,
                        int dtype, int shift_dtype, int kernel, void* workspace, size_t workspace_bytes,
                        void* stream);

/*
 * `count` (1..QUANTO_HIP_MAX_MULTI) qbits_mm products that share the same input x and the same K, bits, group size and dtypes -
 * the q/k/v or gate/up projections of a decoder layer - in one call: y[i] = x @ dequant(W[i]).T (+ bias[i]), W[i] of N[i]
 * output features.  The reference issues one F.linear per module (nn/qlinear.py:49-50); its CUDA GEMV analog processes one
 * weight per launch as well (library/extensions/cuda/awq/v2/gemv_cuda.cu:220-308).  A decode-shaped (M <= 4) call is mostly
 * launch + first-byte latency, so when every W[i] is eligible for the GEMV kernel (quanto_hip_qbits_mm_pick says GEMV) all
 * products run in ONE kernel launch, bit-identical to the separate calls.  Batched decode (4 < M <= 64, int4 group 128, every
 * N[i] a multiple of 64) runs ONE launch of the streaming MFMA kernel over the feature blocks of all members: launch, first-byte
 * latency and the split-K tail are paid once, and the wider grid needs a smaller split (results within the same exact-math gate
 * as the separate calls; the summation order over K may differ).  Otherwise this is exactly `count` quanto_hip_qbits_mm calls
 * with KERNEL_AUTO sharing the workspace.  The pointer arrays are HOST arrays of device pointers; `bias` may be NULL or hold
 * NULL entries.  `_ws` takes the scratch buffer quanto_hip_qbits_mm_multi_workspace_size asks for (same contract as
 * quanto_hip_qbits_mm's); the form without it never splits K.
 */
int quanto_hip_qbits_mm_multi(const void* x, int count, const uint8_t* const* packed, const void* const* scale,
                              const void* const* shift, const void* const* bias, void* const* y, const int64_t* N, int64_t M,
                              int64_t K, int bits, int group_size, int dtype, int shift_dtype, void* stream);
int quanto_hip_qbits_mm_multi_ws(const void* x, int count, const uint8_t* const* packed, const void* const* scale,
                                 const void* const* shift, const void* const* bias, void* const* y, const int64_t* N, int64_t M,
                                 int64_t K, int bits, int group_size, int dtype, int shift_dtype, void* workspace,
                                 size_t workspace_bytes, void* stream);
int64_t quanto_hip_qbits_mm_multi_workspace_size(int count, const int64_t* N, int64_t M, int64_t K, int bits, int group_size,
                                                 int dtype);
/* What quanto_hip_qbits_mm_multi_ws will do: *kernel_out = QUANTO_HIP_KERNEL_GEMV or _SKINNY when all products run in one launch
 * of that kernel (with *workspace_bytes_out of zeroed-counter scratch for the latter), QUANTO_HIP_KERNEL_AUTO for separate calls. */
int quanto_hip_qbits_mm_multi_plan(int count, const int64_t* N, int64_t M, int64_t K, int bits, int group_size, int dtype,
                                   int* kernel_out, int64_t* workspace_bytes_out);

/*
 * Scratch bytes quanto_hip_qbits_mm needs for this problem (0 when the selected kernel needs none).
 * The caller allocates it (16-byte aligned), passes it as `workspace` and may reuse it for any later
 * call on the same stream.  The MFMA kernel stores the per-group row sums of x there
 * (fp32 [K/group_size][roundup(M,128)]; MFMA_FUSED4 needs none unless it splits K - K > 8192 - then zeroed counters + fp32 partial tiles); the DEQUANT_MFMA path stores the dequantized weight (dtype[N, K]).
 * The SKINNY kernel splits K across workgroups when N alone cannot occupy the chip: its workspace starts with
 * QUANTO_HIP_WS_COUNTER_BYTES bytes of arrival counters that MUST BE ZERO on entry (the kernel leaves them zero), followed by
 * fp32 partial sums; without a workspace it runs unsplit.  quanto_hip_qbits_mm_pick tells which kernel AUTO selects, so
 * that a caller can hand the zero-initialised buffer to exactly those calls.
 * Returns a negative status on invalid arguments.
 */
int64_t quanto_hip_qbits_mm_workspace_size(int64_t M, int64_t N, int64_t K, int bits, int group_size, int dtype,
                                           int kernel);
/* The quanto_hip_kernel that QUANTO_HIP_KERNEL_AUTO resolves to for this problem when a sufficient workspace is given. */
int quanto_hip_qbits_mm_pick(int64_t M, int64_t N, int64_t K, int bits, int group_size, int dtype);
/* Both answers in one call (a binding on the decode path makes one FFI round trip instead of two): resolves `kernel`
 * (AUTO -> the kernel quanto_hip_qbits_mm_pick returns) into *kernel_out and its scratch bytes into *workspace_bytes_out. */
int quanto_hip_qbits_mm_plan(int64_t M, int64_t N, int64_t K, int bits, int group_size, int dtype, int kernel, int* kernel_out,
                             int64_t* workspace_bytes_out);

/*
 * quanto::qbytes_mm(Tensor A, Tensor B, Tensor scales) -> Tensor
 *   replaces library/qbytes_mm.py:25-50,73-88: y = (A @ B.T) * scales.T, fp32 (or int32) accumulate.
 * a: a_dtype[M, K] with a_dtype in {F32,F16,BF16} (float activations) or I8 / F8_* (quantized
 *    activations, tensor/weights/qbytes.py:72-73);
 * b: b_dtype[N, K] with b_dtype in {I8, F8_E4M3FN, F8_E5M2, F8_E4M3FNUZ};
 * scales: out_dtype[N] (the reference passes (N,1) or a (1,N)-broadcastable product);
 * bias: out_dtype[N] or NULL; y: out_dtype[M, N].
 */
int quanto_hip_qbytes_mm(const void* a, const void* b, const void* scales, const void* bias, void* y,
                         int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype, int kernel,
                         void* stream);

/*
 * `count` (1..QUANTO_HIP_MAX_MULTI) qbytes_mm products that share the activation `a` and K and the dtypes - q/k/v or gate/up of an
 * int8 / fp8 model (WeightQBytesLinearFunction.forward issues one quanto::qbytes_mm per module, tensor/weights/qbytes.py:68-82) - in
 * one call, the counterpart of quanto_hip_qbits_mm_multi: ONE launch of the GEMV (M <= 2) or of the streaming MFMA kernel
 * (M <= 64, every N[i] a multiple of 64; `workspace` as quanto_hip_qbytes_mm_multi_plan asks), otherwise `count` separate
 * quanto_hip_qbytes_mm_ws calls without scratch.  HOST arrays of device pointers; `bias` may be NULL or hold NULL entries.
 */
int quanto_hip_qbytes_mm_multi_ws(const void* a, int count, const void* const* b, const void* const* scales,
                                  const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int a_dtype,
                                  int b_dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream);
/* *kernel_out = QUANTO_HIP_KERNEL_GEMV / _SKINNY (one launch, *workspace_bytes_out of zeroed-counter scratch for the latter) or
 * QUANTO_HIP_KERNEL_AUTO (separate calls). */
int quanto_hip_qbytes_mm_multi_plan(int count, const int64_t* N, int64_t M, int64_t K, int a_dtype, int b_dtype, int out_dtype,
                                    int* kernel_out, int64_t* workspace_bytes_out);

/*
 * Same, with a caller-provided scratch buffer.  The SKINNY kernel (float activations, 2 < M <= QUANTO_HIP_SKINNY_MAX_M), the
 * MFMA_LARGE kernel and (r6) the NATIVE8 kernel use it to split K across workgroups when the output tiles alone cannot occupy the
 * chip: the buffer starts with arrival counters that MUST BE ZERO on entry (the kernels leave them zero), followed by partial
 * accumulators (fp32, int32 for int8 x int8: the split result of that product stays bit-identical to the unsplit one) - the
 * contract of quanto_hip_qbits_mm's SKINNY kernel.  With workspace == NULL the call is quanto_hip_qbytes_mm.  quanto_hip_qbytes_mm_pick returns the kernel AUTO selects.
 */
int quanto_hip_qbytes_mm_ws(const void* a, const void* b, const void* scales, const void* bias, void* y,
                            int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype, int kernel,
                            void* workspace, size_t workspace_bytes, void* stream);
int64_t quanto_hip_qbytes_mm_workspace_size(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype, int kernel);
int quanto_hip_qbytes_mm_pick(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype);
int quanto_hip_qbytes_mm_plan(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype, int kernel, int* kernel_out,
                              int64_t* workspace_bytes_out);

/*
 * quanto::quantize_symmetric(Tensor base, ScalarType dtype, int? axis, Tensor scale) -> Tensor
 *   replaces library/quantize.py:26-55 (div, round, clamp, cast = four elementwise passes) with one pass; it is the
 *   per-forward step of quantized activations (tensor/activations/qbytes.py:31-39, nn/qmodule.py:281-291).
 * base: in_dtype[numel] contiguous, in_dtype in {F32, F16, BF16}; scale: in_dtype, laid out per scale_mode:
 *   QUANTO_HIP_SCALE_PER_TENSOR  one value (axis=None; activations are always per-tensor);
 *   QUANTO_HIP_SCALE_AXIS_FIRST  numel/inner values, element i uses scale[i / inner]   (axis=0);
 *   QUANTO_HIP_SCALE_AXIS_LAST   inner values,       element i uses scale[i % inner]   (axis=-1).
 * out: out_dtype[numel], out_dtype in {I8, F8_E4M3FN, F8_E5M2}.  out = cast(clamp(q)) with q = base / scale rounded
 *   to in_dtype, and additionally rounded half-to-even to an integer for I8 - bit-identical to the reference sequence.
 */
enum quanto_hip_scale_mode { QUANTO_HIP_SCALE_PER_TENSOR = 0, QUANTO_HIP_SCALE_AXIS_FIRST = 1, QUANTO_HIP_SCALE_AXIS_LAST = 2 };
int quanto_hip_quantize_symmetric(const void* base, const void* scale, void* out, int64_t numel, int64_t inner,
                                  int scale_mode, int in_dtype, int out_dtype, void* stream);

/*
 * QBytesTensor.dequantize() for a per-tensor scale (r6) - tensor/qbytes.py:23-36 `scale * data.to(dtype)`, the step the reference takes whenever a quantized
 * activation meets an op without a quantized kernel (tensor/weights/awq/qbits.py:57-58 before every int4 product) - as one pass instead of a cast kernel and
 * a multiply kernel: out[i] = T(float(q[i]) * float(scale[0])), bit-identical to the two-kernel sequence (every int8 / fp8 value is exact in T).
 *   q: int8 / OCP fp8 [numel] (16-byte aligned); scale: T[1]; out: T[numel] (16-byte aligned); T in {F32, F16, BF16}.  QUANTO_HIP_EALIGN for other views.
 */
int quanto_hip_dequantize_symmetric(const void* q, const void* scale, void* out, int64_t numel, int q_dtype, int out_dtype, void* stream);

/*
 * quanto::quantize_affine(Tensor base, int bits, int axis, int? group_size, Tensor scale, Tensor shift) -> Tensor, axis 0
 *   replaces library/quantize.py:66-78 (add/div, round, clamp, cast passes) at freeze / dynamic-quantization time.
 * base: dtype[N*K] (row-major [N, K]); scale: dtype[N*K/C]; shift: dtype[N*K/C] (float shift) or U8/I8 (zero-point), with
 * C = group_size, or K when group_size == 0 (per-channel).  out: uint8[N*K] = the grouped matrix [N*K/C, C] of values in
 * [0, 2^bits), bit-identical to the reference sequence in every dtype.
 */
int quanto_hip_quantize_affine(const void* base, const void* scale, const void* shift, uint8_t* out, int64_t N, int64_t K,
                               int bits, int group_size, int dtype, int shift_dtype, void* stream);

/*
 * quantize_affine + pack_weights fused: what freezing an int4 / int2 weight does (library/quantize.py:66-78 followed by
 * tensor/packed.py:24-69, i.e. nn/qmodule.py:301-304 -> WeightQBitsTensor.__init__ -> PackedTensor.pack), in one pass.
 * Arguments as quanto_hip_quantize_affine; packed: uint8[ceil(R / (8/bits)), C] with R = N*K/C grouped rows - the
 * PackedTensor._data of the frozen weight, bit-identical to quanto_hip_pack(quanto_hip_quantize_affine(...)).
 */
int quanto_hip_quantize_affine_packed(const void* base, const void* scale, const void* shift, uint8_t* packed, int64_t N, int64_t K,
                                      int bits, int group_size, int dtype, int shift_dtype, void* stream);

/*
 * pack_weights (tensor/packed.py:24-69): packed[r, c] = OR_i unpacked[r + i*row_dim, c] << (bits*i), row_dim = ceil(rows / (8/bits)).
 * unpacked: uint8[rows, cols]; packed: uint8[row_dim, cols].  Inverse of quanto_hip_unpack (+ the trailing-row trim).
 */
int quanto_hip_pack(const uint8_t* unpacked, uint8_t* packed, int64_t rows, int64_t cols, int bits, void* stream);

/*
 * qbits_mm with QUANTIZED activations (W4A8, r6): F.linear(ActivationQBytesTensor, WeightQBitsTensor) - the combination the reference's
 * tests/tensor/ops/test_linear_dispatch.py:22-42 exercises and every backend of the reference serves by dequantizing the activation first
 * (tensor/weights/qbits.py:262-287 -> tensor/function.py:41-47; the CUDA subclasses: tensor/weights/awq/qbits.py:57-58).  Here the stored int8 / fp8-e4m3
 * activation values meet the stored nibbles on the 8-bit matrix instructions (csrc/qbits_a8_fused.hip):
 *   y[m, n] = a_scale * sum_g ( scale[n,g] * sum_{k in g} a[m,k] q[n,k] - z[n,g] * sum_{k in g} a[m,k] ) (+ bias[n]),  z = shift, or scale * zero_point
 *   a: a_dtype[M, K] (I8 or F8_E4M3FN); a_scale: ONE element of `dtype` on the device (the per-tensor activation scale, tensor/activations/qbytes.py:28-43);
 *   packed / scale / shift / bias / y as for quanto_hip_qbits_mm; dtype in {F16, BF16}; bits = 4, group_size = 128, N % 8 == 0, K % 128 == 0.
 * QUANTO_HIP_ENOTSUP for every other format (the caller then dequantizes the activation and calls quanto_hip_qbits_mm, as the reference does).
 * workspace: optional split-K scratch of quanto_hip_qbits_mm_a8_workspace_size bytes, counter region zero on entry (QUANTO_HIP_WS_COUNTER_BYTES) and left
 * zero; without it the call runs unsplit.  For int8 activations the unsplit result is a pure function of the integers (fp32 fma chain over exact group sums).
 */
int quanto_hip_qbits_mm_a8(const void* a, const void* a_scale, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y,
                           int64_t M, int64_t N, int64_t K, int bits, int group_size, int a_dtype, int dtype, int shift_dtype, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Split-K scratch bytes of quanto_hip_qbits_mm_a8 (0: unsplit); QUANTO_HIP_ENOTSUP when the format is not served, QUANTO_HIP_EINVAL on bad arguments. */
int64_t quanto_hip_qbits_mm_a8_workspace_size(int64_t M, int64_t N, int64_t K, int bits, int group_size, int a_dtype, int dtype);

/*
 * F.conv2d with an int8 / fp8 weight - what QConv2d.forward (nn/qconv2d.py:54-55) reaches through WeightQBytesTensor's dispatch, where the
 * reference dequantizes the whole weight per call (qfallback) and runs a float convolution.  Dense convolution (groups = 1) as an IMPLICIT
 * GEMM: y[b, n, oh, ow] = scale[n] * sum_{c,i,j} x[b, c, oh*sh - ph + i*dh, ow*sw - pw + j*dw] * w[n, c, i, j] (+ bias[n]); the im2col operand
 * is gathered inside the kernel's staging loads, nothing is materialised.
 *   x: dtype[B, cin, H, W] (NCHW, contiguous); w: 8-bit [OC, cin, KH, KW] (I8 / F8_E4M3FN / F8_E5M2); scales: dtype[OC]; bias: dtype[OC] or NULL;
 *   y: dtype[B, OC, OH, OW] with OH = (H + 2 ph - dh (KH - 1) - 1) / sh + 1 (OW alike), passed by the caller.  dtype in {F16, BF16}.
 *   Any K = cin * KH * KW below 2^24 (r5: the last K-tile may be ragged), windows of up to 127 taps, B cin H W < 2^30, B OC OH OW < 2^31,
 *   OC K < 2^31; QUANTO_HIP_ENOTSUP otherwise (the caller then lowers the convolution to im2col + qbytes_mm or keeps the reference's dequantize +
 *   float convolution).  Kernel: csrc/qconv_mfma.hip.
 *   workspace / workspace_bytes: optional scratch for the K split, see quanto_hip_conv2d_workspace_size below (NULL, 0: unsplit).
 */
int quanto_hip_qbytes_conv2d(const void* x, const void* w, const void* scales, const void* bias, void* y, int64_t B, int64_t cin, int64_t H,
                             int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int stride_h, int stride_w, int pad_h,
                             int pad_w, int dil_h, int dil_w, int a_dtype, int b_dtype, int out_dtype, void* workspace, size_t workspace_bytes,
                             void* stream);

/*
 * F.conv2d with groups = in_channels (depthwise; channel multiplier OC / cin >= 1) and an int8 / fp8 weight [OC, 1, KH, KW] (r6) - the layer type for which
 * QConv2d.forward (nn/qconv2d.py:54-55) otherwise keeps the reference's dequantize + float convolution:
 *   y[b, oc, oh, ow] = scale[oc] * sum_{i,j} x[b, oc / (OC / cin), oh*sh - ph + i*dh, ow*sw - pw + j*dw] * w[oc, 0, i, j]  (+ bias[oc]).
 * No GEMM in it (KH*KW products per output): a direct stencil kernel, csrc/qconv_depthwise.hip.  Same tensor layouts, dtypes, argument checks and status
 * codes as quanto_hip_qbytes_conv2d; `cin` is the number of INPUT channels (= groups), OC a multiple of it; any window.
 */
int quanto_hip_qbytes_conv2d_depthwise(const void* x, const void* w, const void* scales, const void* bias, void* y, int64_t B, int64_t cin, int64_t H,
                                       int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int stride_h, int stride_w, int pad_h,
                                       int pad_w, int dil_h, int dil_w, int a_dtype, int b_dtype, int out_dtype, void* stream);

/*
 * Scratch bytes the convolution kernels want for their K split (0: the problem is not split): when the 128 x 128 output tiles alone cannot
 * occupy the chip the K-tiles are dealt over up to 64 workgroups per tile, whose fp32 sums a second kernel adds in split order (deterministic,
 * no atomics, nothing to zero).  K = cin * KH * KW.  Both quanto_hip_q*_conv2d entries take the buffer (16-byte aligned); with NULL / too few
 * bytes they run unsplit.  -1 on invalid arguments.
 */
int64_t quanto_hip_conv2d_workspace_size(int64_t B, int64_t OH, int64_t OW, int64_t OC, int64_t K);

/*
 * F.conv2d with an int4 / int2 weight - what QConv2d.forward (nn/qconv2d.py:54-55) reaches through WeightQBitsTensor's dispatch (qfallback: dequantize
 * the whole weight, float convolution).  The same implicit GEMM as quanto_hip_qbytes_conv2d; the packed bytes are dequantized while they are
 * staged, with the reference's own roundings (tensor/qbits.py:27-49: T(T(scale q) - shift) for float shifts, T(scale (q - zero_point)) for
 * integer zero-points), so the matrix cores multiply by exactly the dense weight the reference would have materialised.  No dense weight and no
 * im2col tensor in memory; `workspace` (optional) is the K split's scratch of quanto_hip_conv2d_workspace_size.
 *   x: dtype[B, cin, H, W]; packed: the generic PackedTensor bytes of the axis-0 quantized weight [OC, cin, KH, KW] viewed as [OC, K = cin KH KW]
 *   (byte (p, k) = q[p, k] | q[p + OC/2, k] << 4); scale / shift: [OC * K / group_size] as for quanto_hip_qbits_mm (group_size 0 = per-channel);
 *   bias: dtype[OC] or NULL; y: dtype[B, OC, OH, OW].  dtype in {F16, BF16}; shift_dtype = dtype or U8 / I8.
 *   bits = 4 (OC even) or 2 (OC a multiple of 4; r5), group_size a multiple of 8 or 0, the geometry limits of quanto_hip_qbytes_conv2d;
 *   QUANTO_HIP_ENOTSUP otherwise (the caller then lowers the convolution to im2col + qbits_mm or keeps the reference's dequantize + float convolution).
 *   r5: windows three taps wide at dilation 1 along the width (any stride; W >= 4, cin * KH a multiple of 8, KH * KW <= 31) on at least 8 tiles
 *   of 128 output pixels take another route when `workspace` holds quanto_hip_qbits_conv2d_workspace_size bytes: the weight is dequantized ONCE into
 *   the workspace (the reference's dense weight, tensor/qbits.py:27-49, bit for bit) and the row form of the implicit GEMM multiplies by it - every
 *   pixel tile of the tap kernel dequantizes the whole weight again.  With a smaller workspace (or fewer pixels) the call keeps the tap kernel.
 */
int quanto_hip_qbits_conv2d(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t B, int64_t cin,
                            int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int stride_h, int stride_w, int pad_h,
                            int pad_w, int dil_h, int dil_w, int bits, int group_size, int dtype, int shift_dtype, void* workspace,
                            size_t workspace_bytes, void* stream);

/*
 * Scratch bytes quanto_hip_qbits_conv2d can use: the K split's partial tiles (quanto_hip_conv2d_workspace_size) plus OC * K elements of the
 * activation dtype for the dense weight of the row form, rounded up to 256 bytes.  Plain scratch, nothing to zero.  -1 on invalid arguments.
 */
int64_t quanto_hip_qbits_conv2d_workspace_size(int64_t B, int64_t OH, int64_t OW, int64_t OC, int64_t K);

/*
 * The same with the window geometry (r6): the dense-weight bytes are counted only when quanto_hip_qbits_conv2d can take the row form for this
 * convolution (three taps wide at dilation 1 along the width, W >= 4, cin * KH a multiple of 8, at least 8 tiles of 128 output pixels) - a
 * caller that sizes its scratch with quanto_hip_qbits_conv2d_workspace_size reserves OC * K elements for every int4 / int2 convolution,
 * including the ones that can only run the tap kernel.  -1 on invalid arguments.
 */
int64_t quanto_hip_qbits_conv2d_workspace_size_geom(int64_t B, int64_t cin, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW,
                                                    int stride_w, int dil_w);

#ifdef __cplusplus
}
#endif
#endif /* QUANTO_HIP_H_ */
