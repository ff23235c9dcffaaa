// extern "C" surface of libquanto_hip.so: argument validation + kernel selection.  No torch, no allocation.
#include "qh_common.h"

namespace qh {

static thread_local const char* g_last_kernel = "";
void set_last_kernel(const char* name) { g_last_kernel = name; }

int launch_status() { return hipGetLastError() == hipSuccess ? QUANTO_HIP_OK : QUANTO_HIP_ELAUNCH; }

// implemented in the kernel translation units
int unpack_dispatch(const uint8_t*, uint8_t*, int64_t, int, hipStream_t);
int dequantize_qbits_dispatch(const uint8_t*, const void*, const void*, void*, const PackedGeom&, int, bool, hipStream_t);
int qbytes_mm_naive(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, int, int, hipStream_t);
int qbits_mm_naive(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool,
                   hipStream_t);
bool qbits_gemv_supported(int64_t, const PackedGeom&, int);
int qbits_mm_gemv(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool,
                  hipStream_t);
int qbits_mm_gemv_multi(const void*, int, const uint8_t* const*, const void* const*, const void* const*, const void* const*, void* const*,
                        const int64_t*, int64_t, int64_t, int, bool, hipStream_t);
bool qbytes_gemv_supported(int64_t, int64_t, int64_t, int, int, int);
int qbytes_mm_gemv(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, int, int, hipStream_t);
bool qbytes_mfma_supported(int64_t, int64_t, int64_t, int, int, int);
int qbytes_mm_mfma(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, int, int, hipStream_t);
bool qbytes_mfma_large_supported(int64_t, int64_t, int64_t, int, int, int);
int qbytes_mm_mfma_large(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, int, int, void*, size_t, hipStream_t);
size_t qbytes_mfma_large_workspace(int64_t, int64_t, int64_t);
int quantize_symmetric(const void*, const void*, void*, int64_t, int64_t, int, int, int, hipStream_t);
int dequantize_symmetric(const void*, const void*, void*, int64_t, int, int, hipStream_t);
int quantize_affine(const void*, const void*, const void*, void*, int64_t, int64_t, int, int, bool, hipStream_t);
int quantize_affine_packed(const void*, const void*, const void*, void*, int64_t, int64_t, int, int, bool, hipStream_t);
int pack_weights(const uint8_t*, uint8_t*, int64_t, int64_t, int, hipStream_t);
bool qbits_conv2d_supported(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const PackedGeom&, int);
int qbits_conv2d_mfma(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                      int64_t, int64_t, int, int, int, int, int, int, const PackedGeom&, int, bool, void*, size_t, hipStream_t);
size_t conv2d_workspace(int64_t, int64_t, int64_t);
// depthwise convolution with an 8-bit weight (r6, qconv_depthwise.hip)
int qbytes_conv2d_depthwise(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                            int64_t, int, int, int, int, int, int, int, int, int, hipStream_t);
bool conv2d_last_was_rows();
size_t conv2d_dense_weight_bytes(int64_t, int64_t);
bool conv2d_rows_eligible(int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int64_t);
int qdense_conv2d_rows(const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int,
                       int, int, int, int, int, void*, size_t, hipStream_t);
int qbytes_conv2d_mfma(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                       int64_t, int, int, int, int, int, int, int, int, int, void*, size_t, hipStream_t);
int qbytes_mm_gemv_multi(const void*, int, const void* const*, const void* const*, const void* const*, void* const*, const int64_t*, int64_t,
                         int64_t, int, int, hipStream_t);
bool qbytes_skinny_multi_supported(int, const int64_t*, int64_t, int64_t, int, int, int);
size_t qbytes_skinny_multi_workspace(int, const int64_t*, int64_t, int64_t);
int qbytes_mm_skinny_multi(const void*, int, const void* const*, const void* const*, const void* const*, void* const*, const int64_t*, int64_t,
                           int64_t, int, int, int, void*, size_t, hipStream_t);
bool qbytes_skinny_supported(int64_t, int64_t, int64_t, int, int, int);
size_t qbytes_skinny_workspace(int64_t, int64_t, int64_t);
int qbytes_mm_skinny(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, int, int, void*, size_t, hipStream_t);
bool dense_mm_large_supported(int64_t, int64_t, int64_t, int);
int dense_mm_large(const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, hipStream_t);
bool dense_mm_wd_supported(int64_t, int64_t, int64_t, int);
int dense_mm_wd(const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, hipStream_t);
bool qbytes_native8_supported(int64_t, int64_t, int64_t, int, int, int);
size_t qbytes_native8_workspace(int64_t, int64_t, int64_t, int, int, int);
int qbytes_mm_native8(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, int, int, void*, size_t, hipStream_t);
bool qbits_skinny_multi_supported(int, const int64_t*, int64_t, int64_t, int);
size_t qbits_skinny_multi_workspace(int, const int64_t*, int64_t, int64_t);
int qbits_mm_skinny_multi(const void*, int, const uint8_t* const*, const void* const*, const void* const*, const void* const*, void* const*,
                          const int64_t*, int64_t, int64_t, int, bool, void*, size_t, hipStream_t);
bool qbits_mmv_supported(int64_t, const PackedGeom&, int);
int qbits_mm_mmv(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool, hipStream_t);
bool qbits_skinny_supported(int64_t, const PackedGeom&, int);
size_t qbits_skinny_workspace(int64_t, const PackedGeom&);
int qbits_mm_skinny(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool, void*,
                    size_t, hipStream_t);
bool qbits_mfma_supported(int64_t, const PackedGeom&, int);
size_t qbits_mfma_workspace(int64_t, const PackedGeom&);
int qbits_mm_mfma(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool, void*,
                  size_t, hipStream_t);

bool qbits_mfma_fused_supported(int64_t, const PackedGeom&, int);
bool qbits_mfma_fused_needs_workspace(const PackedGeom&);
size_t qbits_mfma_fused_workspace(int64_t, const PackedGeom&);
int qbits_mm_mfma_fused(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool, void*,
                        size_t, hipStream_t);

bool qbits_mfma_large_supported(int64_t, const PackedGeom&, int);
int qbits_mm_mfma_large(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool, hipStream_t);

// fp32 activations (r6, qmm_f32.hip)
bool qbits_gemv_f32_supported(int64_t, const PackedGeom&, int);
bool qbits_mm_f32_supported(int64_t, const PackedGeom&, int);
int qbits_mm_gemv_f32(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool, hipStream_t);
int qbits_mm_f32(const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, bool, hipStream_t);
bool qbytes_gemv_f32_supported(int64_t, int64_t, int64_t, int, int, int);
bool qbytes_mm_f32_supported(int64_t, int64_t, int64_t, int, int, int);
int qbytes_mm_gemv_f32(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, int, int, hipStream_t);
int qbytes_mm_f32(const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, int, int, hipStream_t);

// quantized activations x int4 weights (r6, qbits_a8_fused.hip)
bool qbits_a8_supported(int64_t, const PackedGeom&, int, int);
size_t qbits_a8_workspace(int64_t, const PackedGeom&);
int qbits_mm_a8(const void*, const void*, const uint8_t*, const void*, const void*, const void*, void*, int64_t, const PackedGeom&, int, int, bool, void*, size_t,
                hipStream_t);

static bool is_float_dtype(int dt) { return dt == QUANTO_HIP_F32 || dt == QUANTO_HIP_F16 || dt == QUANTO_HIP_BF16; }

static int check_qbits(int64_t M, int64_t N, int64_t K, int bits, int group_size, int dtype, int shift_dtype, bool* int_shift) {
  if (M < 0 || N <= 0 || K <= 0) return QUANTO_HIP_EINVAL;
  if (bits != 2 && bits != 4) return QUANTO_HIP_EINVAL;
  if (group_size < 0) return QUANTO_HIP_EINVAL;
  if (group_size > 0 && (K % group_size) != 0) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(dtype)) return QUANTO_HIP_ENOTSUP;
  if (shift_dtype == dtype)
    *int_shift = false;
  else if (shift_dtype == QUANTO_HIP_U8 || shift_dtype == QUANTO_HIP_I8)
    *int_shift = true;
  else
    return QUANTO_HIP_ENOTSUP;
  return QUANTO_HIP_OK;
}

// M above which the weight-streaming GEMV stops being the better choice (it re-reads W once per 4 (int4) / 2 (int8) rows of x)
static bool prefer_gemv(int64_t M) { return M <= QUANTO_HIP_GEMV_MAX_M; }

// qbits_mm kernel choice (measured, int4 g128, N = K = 4096 unless noted):
//   M <= 4    dot2 GEMV: x in registers, every CU busy even for N = 4096 (4.4 / 5.4 / 7.1 us at M = 1 / 2 / 4; 11.3 at M = 8,
//             where the split-K streaming kernel needs 10.1 - and 21 vs 44 us for K = 14336);
//   M <= 256  streaming MFMA kernel in passes of 64 rows, K split across workgroups when N alone cannot occupy the chip
//             (M = 32: 12 us, M = 128: 35 us, M = 256: 68 us; the dense path needs 90 us at any of these M);
//   above     dequantize once into the workspace (one pass over the packed weight, ~N*K*2.5 bytes of HBM traffic), then a
//             dense 256x256-tile GEMM: the dequantization cost is amortised over M rows instead of being paid per tile;
//   the GEMV in passes / the register-staged 128x128 kernel / the naive kernel serve what those reject.
static bool dequant_mfma_supported(int64_t M, const PackedGeom& g, int dtype) { return dense_mm_large_supported(M, g.N, g.K, dtype); }
static size_t dequant_mfma_workspace(const PackedGeom& g) { return (size_t)g.N * g.K * 2; }

// Fused int4 GEMM (qbits_mfma_fused.hip) vs dequantize + dense GEMM vs the streaming kernel, r3 measurements (us; fused / dequantize +
// dense / streaming in passes of 64 rows):
//   N = K = 4096:        M = 72 21.6 / 58 / 24.4, 128 21.8 / 55 / 31.8, 256 25.2 / 53 / 61.6, 512 28.5 / 52.7, 1024 50.4 / 53.1,
//                        1536 79.3 / 81.9, 2048 98.4 / 84.4, 4096 191 / 108
//   N = 14336, K = 4096: M = 128 27.9 / 66 / 71, 256 49.6 / 68.5, 512 97.1 / 97.2, 1024 184 / 128
//   N = 4096, K = 14336: M = 128 40.0 / 155 / 80, 256 55.1 / 154, 512 84.8 / 155, 1024 169 / 162
//   N = 1024, K = 4096:  M = 128 18.5 / 47 / 25.7, 512 22.5 / 49.5, 1024 26.6 / 48.7
// The fused kernel's time is rounds of tiles x groups (its own model picks 64- or 128-token tiles and the K split, qbits_mfma_fused.hip);
// dequantize + dense is flat in M up to ~1 k rows (one dequantize pass + a dense GEMM that cannot fill the chip) and wins once the
// fused kernel needs more than ~1.6 rounds of 128-token tiles (1.2 until two 64-token workgroups shared a CU: (1536,4096,4096) 71.5 / 81.4).  QUANTO_HIP_FUSED4_MAX_COST (x 100) / _MIN_M override the limits in experiments.
float qbits_mfma_fused_cost(int64_t, const PackedGeom&);
// r5: the dense kernel behind dequantize + dense got faster (128-byte rows, qmm_native8.hip) and gains most on wide outputs (more tiles per round):
// off the fitted grid AUTO was 1.31 x behind at (1024, K = 11008, N = 5120) (fused 193 vs 147 us) and 1.11 x at (1024,5120,5120) (87.8 vs 79.1),
// while N = 4096 still hands over at ~1.6 rounds ((1536,4096,4096) 72.2 vs 83.9, (1536,14336,4096) 250 vs 256; profiles/r05_auto_vs_best.jsonl,
// r05_prefill_handover.jsonl): beyond 4096 output features the limit is 1.3 rounds.
static bool fused4_wins(int64_t M, const PackedGeom& g) {
  if (M <= env_int("QUANTO_HIP_FUSED4_MIN_M", 64)) return false;
  return qbits_mfma_fused_cost(M, g) * 100.f <= (float)env_int("QUANTO_HIP_FUSED4_MAX_COST", g.N > 4096 ? 130 : 160);
}

// The register-streaming kernel (K split inside the block, no split-K tail) against the LDS-streaming one, us per launch:
// (8,4096,4096) 7.67 / 9.17, (16,4096,4096) 8.89 / 9.13, (8,4096,1024) 7.61 / 8.14; but every block re-reads all of x, so it
// loses once the grid exceeds one block per CU or K grows: (8,5120,5120) 13.2 / 10.6, (8,8192,8192) 19.4 / 14.8,
// (8,4096,14336) 15.9 / 13.9, (8,14336,4096) 19.5 / 15.0, (32,4096,4096) 12.2 / 10.7
static bool mmv_wins(int64_t M, const PackedGeom& g) {
  return M <= env_int("QUANTO_HIP_MMV_MAX_M", 16) && g.N <= env_int("QUANTO_HIP_MMV_MAX_N", 4096) && g.K <= 4096;
}

// Large-tile int4 GEMM (qbits_mfma_large.hip: reference-rounded operands built in registers, no workspace, 2 x less traffic) against
// dequantize + dense GEMM.  r4 (64-byte-row dense kernel): 4096^3 138.9 / 116.8 us, but 8192^3 932 / 1339, (16384,8192,8192) 1872 / 2702,
// (4096,8192,28672) 1643 / 2167 - it won once the dense weight fell out of the Infinity Cache.  r5: with the 128-byte-row dense kernel and its grouped
// tile raster (qmm_native8.hip) dequantize + dense wins at EVERY size measured (profiles/r05_large4_vs_dequant_dense.jsonl, group sizes 128 / 96 / 32):
// 4096^3 133.7 / 110.8, 8192^3 958.9 / 791.0, (16384,8192,8192) 1959 / 1677, (4096,28672,8192) 1697 / 1517, (4096,8192,28672) 1684 / 1524 -
// its conversion (~2.9 VALU per MFMA) costs more than one dequantize pass.  It remains what AUTO takes at prefill sizes when the caller has
// no workspace for the dense weight (N * K * 2 bytes), and a forced kernel (QUANTO_HIP_LARGE4=2: whenever supported).
static bool large4_wins(int64_t M, const PackedGeom& g, bool have_workspace) {
  const int mode = env_int("QUANTO_HIP_LARGE4", 1);  // experiments: 0 never, 2 whenever supported
  if (mode == 0) return false;
  if (mode == 2) return M > 192;
  (void)g;
  return !have_workspace && M > 1024;  // otherwise: passes of the streaming kernel / the one-thread-per-output kernel
}

static int pick_qbits_kernel(int64_t M, const PackedGeom& g, int dtype, bool have_workspace) {
  if (dtype == QUANTO_HIP_F32) {
    // fp32 activations (r6): the weight stream with fp32 arithmetic up to 8 rows, fp32 MFMA tiles beyond (qmm_f32.hip); K % 16 / % 32 != 0
    // and group sizes that are not a multiple of 16 keep the one-thread-per-output kernel
    if (qbits_gemv_f32_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_GEMV;
    if (qbits_mm_f32_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_MFMA;
    return QUANTO_HIP_KERNEL_NAIVE;
  }
  if (M <= env_int("QUANTO_HIP_GEMV_MAX_M", 4) && qbits_gemv_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_GEMV;
  if (mmv_wins(M, g) && qbits_mmv_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_MMV;
  if (fused4_wins(M, g) && qbits_mfma_fused_supported(M, g, dtype) && (have_workspace || !qbits_mfma_fused_needs_workspace(g)))
    return QUANTO_HIP_KERNEL_MFMA_FUSED4;
  if (large4_wins(M, g, have_workspace) && qbits_mfma_large_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_MFMA_LARGE4;
  // the streaming kernel's time grows with M (passes of 64 rows), dequantize + dense GEMM is flat in M up to 1024 rows:
  // (M, 4096, 4096) us streaming / dequantize + GEMM: M = 128 34 / 56, M = 256 66 / 54; (256, 14336, 4096) 116 / 79; but
  // (256, 4096, 14336) 127 / 167
  const bool flat_wins = M > 192 && g.K <= 8192 && have_workspace && dequant_mfma_supported(M, g, dtype);
  if (!flat_wins && qbits_skinny_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_SKINNY;
  if (!flat_wins && qbits_gemv_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_GEMV;
  if (have_workspace && M > QUANTO_HIP_GEMV_MAX_M_QBITS && dequant_mfma_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_DEQUANT_MFMA;
  if (have_workspace && qbits_mfma_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_MFMA;
  // int2, per-channel and group sizes other than 64 / 128 at small M: still one fused dequantize + one dense GEMM rather than
  // the one-thread-per-output kernel
  if (have_workspace && dequant_mfma_supported(M, g, dtype)) return QUANTO_HIP_KERNEL_DEQUANT_MFMA;
  return QUANTO_HIP_KERNEL_NAIVE;
}

// The LDS-DMA pipelined kernel (256x256 or 128x128 tiles, picked inside) whenever its 128-tiles can occupy a good part of
// the chip; the register-staged 128x128 kernel remains for what it does not support (fp32, K % 64 != 0, K < 128)
// ... or M fills most of a 128-row tile: with the weights-direct loop a lone 128-tile workgroup streams a K-tile in 0.42 us,
// which beats the streaming kernel's passes of 64 rows (bf16 x int8, us, streaming -> tile: (128,4096,4096) 32 -> 27,
// (256,4096,4096) 64 -> 27, (128,14336,4096) 45 -> 28, (128,1024,4096) 29 -> 27; but (96,4096,4096) 26 -> 32)
// r4 (profiles/r04_auto_vs_best.jsonl, off the fitted grid): at M = 96 the tile kernel already wins from 40 tiles on while K is short enough
// for its K split to cover the chip - (96,5120,5120) 30.3 vs 34.1 us streaming, but (96,11008,5120) 69.4 vs 56.4 and (96,2048,2048) 21.0 vs 18.1
static bool prefer_large_tile(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  return tiles >= 64 || (tiles >= 40 && K <= 8192 && M >= 96) || (M > 96 && K >= 512);
}

}  // namespace qh

using namespace qh;

extern "C" {

int quanto_hip_abi_version(void) { return QUANTO_HIP_ABI_VERSION; }

const char* quanto_hip_status_string(int status) {
  switch (status) {
    case QUANTO_HIP_OK: return "ok";
    case QUANTO_HIP_EINVAL: return "invalid argument";
    case QUANTO_HIP_ENOTSUP: return "unsupported dtype/layout combination";
    case QUANTO_HIP_ELAUNCH: return "HIP kernel launch failed";
    case QUANTO_HIP_EALIGN: return "pointer is not 16-byte aligned";
  }
  return "unknown status";
}

const char* quanto_hip_last_kernel(void) { return g_last_kernel; }

int64_t quanto_hip_stream_capture_id(void* stream) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo(reinterpret_cast<hipStream_t>(stream), &status, &id) != hipSuccess) {
    (void)hipGetLastError();
    return QUANTO_HIP_EINVAL;
  }
  if (status != hipStreamCaptureStatusActive) return 0;
  return (int64_t)(id & 0x7FFFFFFFFFFFFFFFull) + 1;  // ids start at 0 on some runtimes: keep "capturing" non-zero
}

int quanto_hip_unpack(const uint8_t* packed, uint8_t* unpacked, int64_t packed_numel, int bits, void* stream) {
  if (bits != 2 && bits != 4) return QUANTO_HIP_EINVAL;
  if (packed_numel < 0) return QUANTO_HIP_EINVAL;
  if (packed_numel > 0 && (packed == nullptr || unpacked == nullptr)) return QUANTO_HIP_EINVAL;
  return unpack_dispatch(packed, unpacked, packed_numel, bits, reinterpret_cast<hipStream_t>(stream));
}

int quanto_hip_dequantize_qbits(const uint8_t* packed, const void* scale, const void* shift, void* out, int64_t N, int64_t K, int bits,
                                int group_size, int dtype, int shift_dtype, void* stream) {
  bool int_shift = false;
  const int st = check_qbits(0, N, K, bits, group_size, dtype, shift_dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  if (!packed || !scale || !shift || !out) return QUANTO_HIP_EINVAL;
  const PackedGeom g = make_geom(N, K, bits, group_size);
  return dequantize_qbits_dispatch(packed, scale, shift, out, g, dtype, int_shift, reinterpret_cast<hipStream_t>(stream));
}

int64_t quanto_hip_qbits_mm_workspace_size(int64_t M, int64_t N, int64_t K, int bits, int group_size, int dtype, int kernel) {
  bool int_shift = false;
  const int st = check_qbits(M, N, K, bits, group_size, dtype, dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  const PackedGeom g = make_geom(N, K, bits, group_size);
  if (kernel == QUANTO_HIP_KERNEL_AUTO) kernel = pick_qbits_kernel(M, g, dtype, true);
  if (kernel == QUANTO_HIP_KERNEL_SKINNY) return qbits_skinny_supported(M, g, dtype) ? (int64_t)qbits_skinny_workspace(M, g) : 0;
  if (kernel == QUANTO_HIP_KERNEL_MFMA) return qbits_mfma_supported(M, g, dtype) ? (int64_t)qbits_mfma_workspace(M, g) : 0;
  if (kernel == QUANTO_HIP_KERNEL_DEQUANT_MFMA) return dequant_mfma_supported(M, g, dtype) ? (int64_t)dequant_mfma_workspace(g) : 0;
  if (kernel == QUANTO_HIP_KERNEL_MFMA_FUSED4) return qbits_mfma_fused_supported(M, g, dtype) ? (int64_t)qbits_mfma_fused_workspace(M, g) : 0;
  return 0;
}

int quanto_hip_qbits_mm_pick(int64_t M, int64_t N, int64_t K, int bits, int group_size, int dtype) {
  bool int_shift = false;
  const int st = check_qbits(M, N, K, bits, group_size, dtype, dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  return pick_qbits_kernel(M, make_geom(N, K, bits, group_size), dtype, true);
}

int quanto_hip_qbits_mm_plan(int64_t M, int64_t N, int64_t K, int bits, int group_size, int dtype, int kernel, int* kernel_out,
                             int64_t* workspace_bytes_out) {
  if (!kernel_out || !workspace_bytes_out) return QUANTO_HIP_EINVAL;
  if (kernel == QUANTO_HIP_KERNEL_AUTO) {
    kernel = quanto_hip_qbits_mm_pick(M, N, K, bits, group_size, dtype);
    if (kernel < 0) return kernel;
  }
  const int64_t ws = quanto_hip_qbits_mm_workspace_size(M, N, K, bits, group_size, dtype, kernel);
  if (ws < 0) return (int)ws;
  *kernel_out = kernel;
  *workspace_bytes_out = ws;
  return QUANTO_HIP_OK;
}

int quanto_hip_qbits_mm(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                        int64_t N, int64_t K, int bits, int group_size, int dtype, int shift_dtype, int kernel, void* workspace,
                        size_t workspace_bytes, void* stream_) {
  bool int_shift = false;
  const int st = check_qbits(M, N, K, bits, group_size, dtype, shift_dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  if (M == 0) return QUANTO_HIP_OK;
  if (!x || !packed || !scale || !shift || !y) return QUANTO_HIP_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const PackedGeom g = make_geom(N, K, bits, group_size);
  if (kernel == QUANTO_HIP_KERNEL_AUTO) {
    // the fast kernels need 16-byte aligned x / packed (views into larger buffers may not be): AUTO then takes the kernel that copes
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) {
      const int r = qbits_mm_naive(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("naive");
      return r;
    }
    kernel = pick_qbits_kernel(M, g, dtype, workspace != nullptr);
    if (kernel == QUANTO_HIP_KERNEL_DEQUANT_MFMA && workspace_bytes < dequant_mfma_workspace(g))
      kernel = qbits_mfma_supported(M, g, dtype) ? QUANTO_HIP_KERNEL_MFMA : QUANTO_HIP_KERNEL_NAIVE;
    if (kernel == QUANTO_HIP_KERNEL_MFMA && dtype != QUANTO_HIP_F32 && workspace_bytes < qbits_mfma_workspace(M, g)) kernel = QUANTO_HIP_KERNEL_NAIVE;
  }
  int r;
  switch (kernel) {
    case QUANTO_HIP_KERNEL_NAIVE:
      r = qbits_mm_naive(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("naive");
      return r;
    case QUANTO_HIP_KERNEL_GEMV:
      if (dtype == QUANTO_HIP_F32) {
        r = qbits_mm_gemv_f32(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, stream);
        if (r == QUANTO_HIP_OK) set_last_kernel("gemv_f32");
        return r;
      }
      r = qbits_mm_gemv(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("gemv");
      return r;
    case QUANTO_HIP_KERNEL_MFMA:
      if (dtype == QUANTO_HIP_F32) {  // fp32 MFMA tiles: no workspace
        r = qbits_mm_f32(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, stream);
        if (r == QUANTO_HIP_OK) set_last_kernel("mfma_f32");
        return r;
      }
      r = qbits_mm_mfma(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, workspace, workspace_bytes, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("mfma");
      return r;
    case QUANTO_HIP_KERNEL_DEQUANT_MFMA:
      if (!dequant_mfma_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
      if (!workspace || workspace_bytes < dequant_mfma_workspace(g)) return QUANTO_HIP_EINVAL;
      r = dequantize_qbits_dispatch(packed, scale, shift, workspace, g, dtype, int_shift, stream);
      if (r != QUANTO_HIP_OK) return r;
      r = dense_mm_wd_supported(M, g.N, g.K, dtype) ? dense_mm_wd(x, workspace, bias, y, M, g.N, g.K, dtype, stream)
                                                    : dense_mm_large(x, workspace, bias, y, M, g.N, g.K, dtype, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("dequant_mfma");
      return r;
    case QUANTO_HIP_KERNEL_SKINNY:
      r = qbits_mm_skinny(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, workspace, workspace_bytes, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("skinny");
      return r;
    case QUANTO_HIP_KERNEL_MMV:
      r = qbits_mm_mmv(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("mmv");
      return r;
    case QUANTO_HIP_KERNEL_MFMA_FUSED4:
      r = qbits_mm_mfma_fused(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, workspace, workspace_bytes, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("mfma_fused4");
      return r;
    case QUANTO_HIP_KERNEL_MFMA_LARGE4:
      r = qbits_mm_mfma_large(x, packed, scale, shift, bias, y, M, g, dtype, int_shift, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("mfma_large4");
      return r;
  }
  return QUANTO_HIP_EINVAL;
}

int64_t quanto_hip_qbits_mm_a8_workspace_size(int64_t M, int64_t N, int64_t K, int bits, int group_size, int a_dtype, int dtype) {
  bool int_shift = false;
  const int st = check_qbits(M, N, K, bits, group_size, dtype, dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  const PackedGeom g = make_geom(N, K, bits, group_size);
  if (M == 0) return 0;
  if (!qbits_a8_supported(M, g, a_dtype, dtype)) return QUANTO_HIP_ENOTSUP;
  return (int64_t)qbits_a8_workspace(M, g);
}

int quanto_hip_qbits_mm_a8(const void* a, const void* a_scale, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y,
                           int64_t M, int64_t N, int64_t K, int bits, int group_size, int a_dtype, int dtype, int shift_dtype, void* workspace,
                           size_t workspace_bytes, void* stream) {
  bool int_shift = false;
  const int st = check_qbits(M, N, K, bits, group_size, dtype, shift_dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  if (M == 0) return QUANTO_HIP_OK;
  if (!a || !a_scale || !packed || !scale || !shift || !y) return QUANTO_HIP_EINVAL;
  const PackedGeom g = make_geom(N, K, bits, group_size);
  const int r = qbits_mm_a8(a, a_scale, packed, scale, shift, bias, y, M, g, a_dtype, dtype, int_shift, workspace, workspace_bytes,
                            reinterpret_cast<hipStream_t>(stream));
  if (r == QUANTO_HIP_OK) set_last_kernel(a_dtype == QUANTO_HIP_I8 ? "a8_fused_int8" : "a8_fused_fp8");
  return r;
}

// one streaming-MFMA launch for the whole group pays when the members are small enough to be dominated by per-call costs
static bool skinny_multi_applies(int count, const int64_t* N, int64_t M, int64_t K, int bits, int group_size, int dtype) {
  return count >= 2 && bits == 4 && group_size == 128 && M > 4 && M <= env_int("QUANTO_HIP_SKINNY_MULTI_MAX_M", 64) &&
         qbits_skinny_multi_supported(count, N, M, K, dtype);
}

int64_t quanto_hip_qbits_mm_multi_workspace_size(int count, const int64_t* N, int64_t M, int64_t K, int bits, int group_size, int dtype) {
  if (count < 1 || count > QUANTO_HIP_MAX_MULTI || !N) return QUANTO_HIP_EINVAL;
  if (skinny_multi_applies(count, N, M, K, bits, group_size, dtype)) return (int64_t)qbits_skinny_multi_workspace(count, N, M, K);
  int64_t need = 0;  // separate calls, one after the other on the stream: they share the buffer
  for (int i = 0; i < count; ++i) {
    const int64_t w = quanto_hip_qbits_mm_workspace_size(M, N[i], K, bits, group_size, dtype, QUANTO_HIP_KERNEL_AUTO);
    if (w < 0) return w;
    need = w > need ? w : need;
  }
  return need;
}

int quanto_hip_qbits_mm_multi_plan(int count, const int64_t* N, int64_t M, int64_t K, int bits, int group_size, int dtype, int* kernel_out,
                                   int64_t* workspace_bytes_out) {
  if (count < 1 || count > QUANTO_HIP_MAX_MULTI || !N || !kernel_out || !workspace_bytes_out) return QUANTO_HIP_EINVAL;
  bool gemv = M >= 1 && M <= 4 && group_size == 128 && bits == 4;
  for (int i = 0; i < count; ++i) {
    bool int_shift = false;
    const int st = check_qbits(M, N[i], K, bits, group_size, dtype, dtype, &int_shift);
    if (st != QUANTO_HIP_OK) return st;
    gemv = gemv && qbits_gemv_supported(M, make_geom(N[i], K, bits, group_size), dtype);
  }
  const bool skinny = !gemv && skinny_multi_applies(count, N, M, K, bits, group_size, dtype);
  *kernel_out = gemv ? QUANTO_HIP_KERNEL_GEMV : (skinny ? QUANTO_HIP_KERNEL_SKINNY : QUANTO_HIP_KERNEL_AUTO);
  *workspace_bytes_out = skinny ? (int64_t)qbits_skinny_multi_workspace(count, N, M, K) : 0;
  return QUANTO_HIP_OK;
}

int quanto_hip_qbits_mm_multi_ws(const void* x, int count, const uint8_t* const* packed, const void* const* scale, const void* const* shift,
                                 const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int bits, int group_size,
                                 int dtype, int shift_dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  if (count < 1 || count > QUANTO_HIP_MAX_MULTI || !packed || !scale || !shift || !y || !N) return QUANTO_HIP_EINVAL;
  bool int_shift = false, one_launch = M >= 1 && M <= 4 && group_size == 128 && bits == 4;  // the fused launch serves int4 g128
  for (int i = 0; i < count; ++i) {
    const int st = check_qbits(M, N[i], K, bits, group_size, dtype, shift_dtype, &int_shift);
    if (st != QUANTO_HIP_OK) return st;
    if (M > 0 && (!x || !packed[i] || !scale[i] || !shift[i] || !y[i])) return QUANTO_HIP_EINVAL;
    one_launch = one_launch && qbits_gemv_supported(M, make_geom(N[i], K, bits, group_size), dtype);
  }
  if (M == 0) return QUANTO_HIP_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (one_launch) {
    const int r = qbits_mm_gemv_multi(x, count, packed, scale, shift, bias, y, N, M, K, dtype, int_shift, stream);
    if (r == QUANTO_HIP_OK) set_last_kernel("gemv_multi");
    if (r != QUANTO_HIP_EALIGN) return r;  // misaligned views: the separate calls below pick a kernel that copes
  } else if (skinny_multi_applies(count, N, M, K, bits, group_size, dtype)) {
    const int r = qbits_mm_skinny_multi(x, count, packed, scale, shift, bias, y, N, M, K, dtype, int_shift, workspace, workspace_bytes, stream);
    if (r == QUANTO_HIP_OK) set_last_kernel("skinny_multi");
    if (r != QUANTO_HIP_EALIGN) return r;
  }
  // Separate calls that share ONE workspace.  The kernels that use it as scratch (dequantized weight, row sums) write from offset 0,
  // i.e. over the counter words the split-K kernels expect to find zero: the counter region is zeroed again behind every such
  // member, so that whichever member comes next (and the caller's next call) sees the "zero on entry" contract.
  for (int i = 0; i < count; ++i) {
    const PackedGeom g = make_geom(N[i], K, bits, group_size);
    const int picked = workspace ? pick_qbits_kernel(M, g, dtype, true) : QUANTO_HIP_KERNEL_NAIVE;
    const int r = quanto_hip_qbits_mm(x, packed[i], scale[i], shift[i], bias ? bias[i] : nullptr, y[i], M, N[i], K, bits, group_size, dtype,
                                      shift_dtype, QUANTO_HIP_KERNEL_AUTO, workspace, workspace_bytes, stream_);
    if (r != QUANTO_HIP_OK) return r;
    if (workspace && (picked == QUANTO_HIP_KERNEL_DEQUANT_MFMA || picked == QUANTO_HIP_KERNEL_MFMA)) {
      const size_t nz = workspace_bytes < (size_t)QUANTO_HIP_WS_COUNTER_BYTES ? workspace_bytes : (size_t)QUANTO_HIP_WS_COUNTER_BYTES;
      if (hipMemsetAsync(workspace, 0, nz, stream) != hipSuccess) return QUANTO_HIP_ELAUNCH;
    }
  }
  return QUANTO_HIP_OK;
}

int quanto_hip_qbits_mm_multi(const void* x, int count, const uint8_t* const* packed, const void* const* scale, const void* const* shift,
                              const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int bits, int group_size,
                              int dtype, int shift_dtype, void* stream) {
  return quanto_hip_qbits_mm_multi_ws(x, count, packed, scale, shift, bias, y, N, M, K, bits, group_size, dtype, shift_dtype, nullptr, 0, stream);
}

// qbytes_mm kernel choice (measured with bf16 x int8, hipGraph replay, N = K = 4096 unless noted):
//   M <= 2    GEMV (7.4 us at M = 1; it re-reads the weights per pair of rows: 38 us at M = 8);
//   quantized activations -> native8 (int8 x int8 / fp8 x fp8 MFMA);
//   M <= 64   streaming MFMA kernel, K split across workgroups (8.8 us at M = 8, 17 us at M = 64; the tiled kernels need 30);
//   enough 128-tiles to occupy the chip -> pipelined large-tile kernel (M = 256: 35 us vs 68 us streaming);
//   M <= 256  streaming kernel in passes of 64 rows (few, long tiles: (256, 4096, 14336) 107 us vs 127 us tiled);
//   else the large-tile kernel whenever it applies, the register-staged 128x128 kernel, the naive kernel.
static int pick_qbytes_kernel(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  if (a_dtype == QUANTO_HIP_F32 && out_dtype == QUANTO_HIP_F32) {  // fp32 activations (r6, qmm_f32.hip)
    if (qbytes_gemv_f32_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_KERNEL_GEMV;
    if (qbytes_mm_f32_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_KERNEL_MFMA;
    return QUANTO_HIP_KERNEL_NAIVE;
  }
  if (M <= 2 && qbytes_gemv_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_KERNEL_GEMV;
  if (qbytes_native8_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_KERNEL_NATIVE8;
  const bool skinny = qbytes_skinny_supported(M, N, K, a_dtype, b_dtype, out_dtype);
  const bool large = qbytes_mfma_large_supported(M, N, K, a_dtype, b_dtype, out_dtype);
  if (skinny && M <= 64) return QUANTO_HIP_KERNEL_SKINNY;
  if (large && prefer_large_tile(M, N, K)) return QUANTO_HIP_KERNEL_MFMA_LARGE;
  if (skinny) return QUANTO_HIP_KERNEL_SKINNY;
  if (prefer_gemv(M) && qbytes_gemv_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_KERNEL_GEMV;
  if (large) return QUANTO_HIP_KERNEL_MFMA_LARGE;
  if (qbytes_mfma_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_KERNEL_MFMA;
  return QUANTO_HIP_KERNEL_NAIVE;
}

int quanto_hip_qbytes_mm_pick(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  if (M < 0 || N <= 0 || K <= 0) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(out_dtype)) return QUANTO_HIP_ENOTSUP;
  return pick_qbytes_kernel(M, N, K, a_dtype, b_dtype, out_dtype);
}

int64_t quanto_hip_qbytes_mm_workspace_size(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype, int kernel) {
  if (M < 0 || N <= 0 || K <= 0) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(out_dtype)) return QUANTO_HIP_ENOTSUP;
  if (kernel == QUANTO_HIP_KERNEL_AUTO) kernel = pick_qbytes_kernel(M, N, K, a_dtype, b_dtype, out_dtype);
  if (kernel == QUANTO_HIP_KERNEL_SKINNY && qbytes_skinny_supported(M, N, K, a_dtype, b_dtype, out_dtype))
    return (int64_t)qbytes_skinny_workspace(M, N, K);
  if (kernel == QUANTO_HIP_KERNEL_MFMA_LARGE && qbytes_mfma_large_supported(M, N, K, a_dtype, b_dtype, out_dtype))
    return (int64_t)qbytes_mfma_large_workspace(M, N, K);
  if (kernel == QUANTO_HIP_KERNEL_NATIVE8) return (int64_t)qbytes_native8_workspace(M, N, K, a_dtype, b_dtype, out_dtype);
  return 0;
}

int quanto_hip_qbytes_mm_plan(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype, int kernel, int* kernel_out,
                              int64_t* workspace_bytes_out) {
  if (!kernel_out || !workspace_bytes_out) return QUANTO_HIP_EINVAL;
  if (kernel == QUANTO_HIP_KERNEL_AUTO) {
    kernel = quanto_hip_qbytes_mm_pick(M, N, K, a_dtype, b_dtype, out_dtype);
    if (kernel < 0) return kernel;
  }
  const int64_t ws = quanto_hip_qbytes_mm_workspace_size(M, N, K, a_dtype, b_dtype, out_dtype, kernel);
  if (ws < 0) return (int)ws;
  *kernel_out = kernel;
  *workspace_bytes_out = ws;
  return QUANTO_HIP_OK;
}

int quanto_hip_qbytes_mm_ws(const void* a, const void* b, const void* scales, const void* bias, void* y, int64_t M, int64_t N, int64_t K,
                            int a_dtype, int b_dtype, int out_dtype, int kernel, void* workspace, size_t workspace_bytes, void* stream_) {
  if (M < 0 || N <= 0 || K <= 0) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(out_dtype)) return QUANTO_HIP_ENOTSUP;
  if (M == 0) return QUANTO_HIP_OK;
  if (!a || !b || !scales || !y) return QUANTO_HIP_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (kernel == QUANTO_HIP_KERNEL_AUTO) {
    kernel = pick_qbytes_kernel(M, N, K, a_dtype, b_dtype, out_dtype);
    // misaligned views: the kernel that has no alignment requirement
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16) kernel = QUANTO_HIP_KERNEL_NAIVE;
  }
  int r;
  switch (kernel) {
    case QUANTO_HIP_KERNEL_NAIVE:
      r = qbytes_mm_naive(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("naive");
      return r;
    case QUANTO_HIP_KERNEL_GEMV:
      if (a_dtype == QUANTO_HIP_F32 && out_dtype == QUANTO_HIP_F32) {
        r = qbytes_mm_gemv_f32(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, stream);
        if (r == QUANTO_HIP_OK) set_last_kernel("gemv_f32");
        return r;
      }
      r = qbytes_mm_gemv(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("gemv");
      return r;
    case QUANTO_HIP_KERNEL_SKINNY:
      r = qbytes_mm_skinny(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, workspace, workspace_bytes, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("skinny");
      return r;
    case QUANTO_HIP_KERNEL_MFMA:
      if (a_dtype == QUANTO_HIP_F32 && out_dtype == QUANTO_HIP_F32) {
        r = qbytes_mm_f32(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, stream);
        if (r == QUANTO_HIP_OK) set_last_kernel("mfma_f32");
        return r;
      }
      r = qbytes_mm_mfma(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("mfma");
      return r;
    case QUANTO_HIP_KERNEL_MFMA_LARGE:
      r = qbytes_mm_mfma_large(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, workspace, workspace_bytes, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("mfma_large");
      return r;
    case QUANTO_HIP_KERNEL_NATIVE8:
      r = qbytes_mm_native8(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, workspace, workspace_bytes, stream);
      if (r == QUANTO_HIP_OK) set_last_kernel("mfma_native8");
      return r;
  }
  return QUANTO_HIP_EINVAL;
}

// ---- several qbytes_mm products that share the activation (q/k/v, gate/up of an int8 / fp8 model), see quanto_hip_qbits_mm_multi -----
// which single launch serves the group: GEMV (M <= 2, 16-bit activations), the streaming MFMA kernel (M <= 64, every N a multiple
// of 64), or none (KERNEL_AUTO: separate calls)
static int qbytes_multi_kernel(int count, const int64_t* N, int64_t M, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  if (count < 2) return QUANTO_HIP_KERNEL_AUTO;
  bool gemv = M >= 1 && M <= 2, lone_skinny = true;
  for (int i = 0; i < count; ++i) {
    gemv = gemv && qbytes_gemv_supported(M, N[i], K, a_dtype, b_dtype, out_dtype);
    lone_skinny = lone_skinny && pick_qbytes_kernel(M, N[i], K, a_dtype, b_dtype, out_dtype) == QUANTO_HIP_KERNEL_SKINNY;
  }
  if (gemv) return QUANTO_HIP_KERNEL_GEMV;
  if (lone_skinny && M <= env_int("QUANTO_HIP_SKINNY_MULTI_MAX_M", 64) && qbytes_skinny_multi_supported(count, N, M, K, a_dtype, b_dtype, out_dtype))
    return QUANTO_HIP_KERNEL_SKINNY;
  return QUANTO_HIP_KERNEL_AUTO;
}

int quanto_hip_qbytes_mm_multi_plan(int count, const int64_t* N, int64_t M, int64_t K, int a_dtype, int b_dtype, int out_dtype, int* kernel_out,
                                    int64_t* workspace_bytes_out) {
  if (count < 1 || count > QUANTO_HIP_MAX_MULTI || !N || !kernel_out || !workspace_bytes_out || M < 0 || K <= 0) return QUANTO_HIP_EINVAL;
  for (int i = 0; i < count; ++i)
    if (N[i] <= 0) return QUANTO_HIP_EINVAL;
  *kernel_out = qbytes_multi_kernel(count, N, M, K, a_dtype, b_dtype, out_dtype);
  *workspace_bytes_out = *kernel_out == QUANTO_HIP_KERNEL_SKINNY ? (int64_t)qbytes_skinny_multi_workspace(count, N, M, K) : 0;
  return QUANTO_HIP_OK;
}

int quanto_hip_qbytes_mm_multi_ws(const void* a, int count, const void* const* b, const void* const* scales, const void* const* bias,
                                  void* const* y, const int64_t* N, int64_t M, int64_t K, int a_dtype, int b_dtype, int out_dtype,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
  if (count < 1 || count > QUANTO_HIP_MAX_MULTI || !b || !scales || !y || !N || M < 0 || K <= 0) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(out_dtype)) return QUANTO_HIP_ENOTSUP;
  for (int i = 0; i < count; ++i)
    if (N[i] <= 0 || (M > 0 && (!a || !b[i] || !scales[i] || !y[i]))) return QUANTO_HIP_EINVAL;
  if (M == 0) return QUANTO_HIP_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int kernel = qbytes_multi_kernel(count, N, M, K, a_dtype, b_dtype, out_dtype);
  if (kernel == QUANTO_HIP_KERNEL_GEMV) {
    const int r = qbytes_mm_gemv_multi(a, count, b, scales, bias, y, N, M, K, b_dtype, out_dtype, stream);
    if (r == QUANTO_HIP_OK) set_last_kernel("gemv_multi");
    if (r != QUANTO_HIP_EALIGN) return r;  // misaligned views: the separate calls below cope
  } else if (kernel == QUANTO_HIP_KERNEL_SKINNY) {
    const int r = qbytes_mm_skinny_multi(a, count, b, scales, bias, y, N, M, K, a_dtype, b_dtype, out_dtype, workspace, workspace_bytes, stream);
    if (r == QUANTO_HIP_OK) set_last_kernel("skinny_multi");
    if (r != QUANTO_HIP_EALIGN) return r;
  }
  for (int i = 0; i < count; ++i) {  // separate calls (no shared scratch: kernels that split K run unsplit)
    const int r = quanto_hip_qbytes_mm_ws(a, b[i], scales[i], bias ? bias[i] : nullptr, y[i], M, N[i], K, a_dtype, b_dtype, out_dtype,
                                          QUANTO_HIP_KERNEL_AUTO, nullptr, 0, stream_);
    if (r != QUANTO_HIP_OK) return r;
  }
  return QUANTO_HIP_OK;
}

int quanto_hip_qbytes_mm(const void* a, const void* b, const void* scales, const void* bias, void* y, int64_t M, int64_t N, int64_t K,
                         int a_dtype, int b_dtype, int out_dtype, int kernel, void* stream) {
  return quanto_hip_qbytes_mm_ws(a, b, scales, bias, y, M, N, K, a_dtype, b_dtype, out_dtype, kernel, nullptr, 0, stream);
}

int quanto_hip_quantize_symmetric(const void* base, const void* scale, void* out, int64_t numel, int64_t inner, int scale_mode,
                                  int in_dtype, int out_dtype, void* stream) {
  if (numel < 0) return QUANTO_HIP_EINVAL;
  if (scale_mode != QUANTO_HIP_SCALE_PER_TENSOR && scale_mode != QUANTO_HIP_SCALE_AXIS_FIRST && scale_mode != QUANTO_HIP_SCALE_AXIS_LAST)
    return QUANTO_HIP_EINVAL;
  if (scale_mode != QUANTO_HIP_SCALE_PER_TENSOR && (inner <= 0 || numel % inner != 0)) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(in_dtype)) return QUANTO_HIP_ENOTSUP;
  if (out_dtype != QUANTO_HIP_I8 && out_dtype != QUANTO_HIP_F8_E4M3FN && out_dtype != QUANTO_HIP_F8_E5M2) return QUANTO_HIP_ENOTSUP;
  if (numel == 0) return QUANTO_HIP_OK;
  if (!base || !scale || !out) return QUANTO_HIP_EINVAL;
  return quantize_symmetric(base, scale, out, numel, inner, scale_mode, in_dtype, out_dtype, reinterpret_cast<hipStream_t>(stream));
}

int quanto_hip_dequantize_symmetric(const void* q, const void* scale, void* out, int64_t numel, int q_dtype, int out_dtype, void* stream) {
  if (numel < 0) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(out_dtype)) return QUANTO_HIP_ENOTSUP;
  if (q_dtype != QUANTO_HIP_I8 && q_dtype != QUANTO_HIP_F8_E4M3FN && q_dtype != QUANTO_HIP_F8_E5M2) return QUANTO_HIP_ENOTSUP;
  if (numel == 0) return QUANTO_HIP_OK;
  if (!q || !scale || !out) return QUANTO_HIP_EINVAL;
  return dequantize_symmetric(q, scale, out, numel, q_dtype, out_dtype, reinterpret_cast<hipStream_t>(stream));
}

int quanto_hip_quantize_affine(const void* base, const void* scale, const void* shift, uint8_t* out, int64_t N, int64_t K, int bits,
                               int group_size, int dtype, int shift_dtype, void* stream) {
  bool int_shift = false;
  const int st = check_qbits(1, N, K, bits, group_size, dtype, shift_dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  if (!base || !scale || !shift || !out) return QUANTO_HIP_EINVAL;
  const int64_t C = group_size > 0 ? group_size : K;
  return quantize_affine(base, scale, shift, out, N * K, C, bits, dtype, int_shift, reinterpret_cast<hipStream_t>(stream));
}

int quanto_hip_quantize_affine_packed(const void* base, const void* scale, const void* shift, uint8_t* packed, int64_t N, int64_t K,
                                      int bits, int group_size, int dtype, int shift_dtype, void* stream) {
  bool int_shift = false;
  const int st = check_qbits(1, N, K, bits, group_size, dtype, shift_dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  if (!base || !scale || !shift || !packed) return QUANTO_HIP_EINVAL;
  const int64_t C = group_size > 0 ? group_size : K;
  return quantize_affine_packed(base, scale, shift, packed, N * K / C, C, bits, dtype, int_shift, reinterpret_cast<hipStream_t>(stream));
}

int quanto_hip_pack(const uint8_t* unpacked, uint8_t* packed, int64_t rows, int64_t cols, int bits, void* stream) {
  if (bits != 2 && bits != 4) return QUANTO_HIP_EINVAL;
  if (rows < 0 || cols < 0) return QUANTO_HIP_EINVAL;
  if (rows == 0 || cols == 0) return QUANTO_HIP_OK;
  if (!unpacked || !packed) return QUANTO_HIP_EINVAL;
  return pack_weights(unpacked, packed, rows, cols, bits, reinterpret_cast<hipStream_t>(stream));
}

int64_t quanto_hip_conv2d_workspace_size(int64_t B, int64_t OH, int64_t OW, int64_t OC, int64_t K) {
  if (B < 0 || OH < 0 || OW < 0 || OC <= 0 || K <= 0) return -1;
  if (B == 0 || OH == 0 || OW == 0) return 0;
  return (int64_t)conv2d_workspace(B * OH * OW, OC, K);
}

int64_t quanto_hip_qbits_conv2d_workspace_size(int64_t B, int64_t OH, int64_t OW, int64_t OC, int64_t K) {
  const int64_t split = quanto_hip_conv2d_workspace_size(B, OH, OW, OC, K);
  if (split < 0) return split;
  if (B == 0 || OH == 0 || OW == 0) return 0;
  return split + (int64_t)conv2d_dense_weight_bytes(OC, K);
}

static bool qbits_conv2d_takes_rows(int64_t B, int64_t cin, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int stride_w, int dil_w) {
  // (the tap kernel dequantizes the whole weight once per 128-pixel tile; one dequantize launch pays from ~8 pixel tiles on: (8,512,7,7) -> 512, 4 of
  // them, 28.8 us on the tap kernel against 30.2 this way, (32,512,7,7), 13 tiles, 44.3 against 40.3 - profiles/r05_qconv2d_rows_one_pixel_ab.jsonl)
  const int64_t pixel_tiles = (B * OH * OW + 127) / 128;
  return conv2d_rows_eligible(cin, KH, KW, W, OW, stride_w, dil_w, OC) && pixel_tiles >= env_int("QUANTO_HIP_CONV_DENSE_MIN_TILES", 8);
}

int64_t quanto_hip_qbits_conv2d_workspace_size_geom(int64_t B, int64_t cin, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW,
                                                    int stride_w, int dil_w) {
  if (cin <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride_w <= 0 || dil_w <= 0) return -1;
  const int64_t K = cin * KH * KW;
  const int64_t split = quanto_hip_conv2d_workspace_size(B, OH, OW, OC, K);
  if (split < 0) return split;
  if (B == 0 || OH == 0 || OW == 0) return 0;
  return split + (qbits_conv2d_takes_rows(B, cin, W, OC, KH, KW, OH, OW, stride_w, dil_w) ? (int64_t)conv2d_dense_weight_bytes(OC, K) : 0);
}

int quanto_hip_qbytes_conv2d(const void* x, const void* w, const void* scales, const void* bias, void* y, int64_t B, int64_t cin, int64_t H,
                             int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int stride_h, int stride_w, int pad_h,
                             int pad_w, int dil_h, int dil_w, int a_dtype, int b_dtype, int out_dtype, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (B < 0 || cin <= 0 || H <= 0 || W <= 0 || OC <= 0 || KH <= 0 || KW <= 0 || OH < 0 || OW < 0) return QUANTO_HIP_EINVAL;
  if (stride_h <= 0 || stride_w <= 0 || pad_h < 0 || pad_w < 0 || dil_h <= 0 || dil_w <= 0) return QUANTO_HIP_EINVAL;
  if (OH != (H + 2 * pad_h - dil_h * (KH - 1) - 1) / stride_h + 1 || OW != (W + 2 * pad_w - dil_w * (KW - 1) - 1) / stride_w + 1) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(out_dtype)) return QUANTO_HIP_ENOTSUP;
  if (B == 0 || OH == 0 || OW == 0) return QUANTO_HIP_OK;
  if (!x || !w || !scales || !y) return QUANTO_HIP_EINVAL;
  const int r = qbytes_conv2d_mfma(x, w, scales, bias, y, B, cin, H, W, OC, KH, KW, OH, OW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, a_dtype,
                                   b_dtype, out_dtype, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
  if (r == QUANTO_HIP_OK) set_last_kernel(conv2d_last_was_rows() ? "conv2d_mfma_rows" : "conv2d_mfma");
  return r;
}

int quanto_hip_qbytes_conv2d_depthwise(const void* x, const void* w, const void* scales, const void* bias, void* y, int64_t B, int64_t cin, int64_t H,
                                       int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int stride_h, int stride_w, int pad_h,
                                       int pad_w, int dil_h, int dil_w, int a_dtype, int b_dtype, int out_dtype, void* stream) {
  if (B < 0 || cin <= 0 || H <= 0 || W <= 0 || OC <= 0 || KH <= 0 || KW <= 0 || OH < 0 || OW < 0 || OC % cin != 0) return QUANTO_HIP_EINVAL;
  if (stride_h <= 0 || stride_w <= 0 || pad_h < 0 || pad_w < 0 || dil_h <= 0 || dil_w <= 0) return QUANTO_HIP_EINVAL;
  if (OH != (H + 2 * pad_h - dil_h * (KH - 1) - 1) / stride_h + 1 || OW != (W + 2 * pad_w - dil_w * (KW - 1) - 1) / stride_w + 1) return QUANTO_HIP_EINVAL;
  if (!is_float_dtype(out_dtype)) return QUANTO_HIP_ENOTSUP;
  if (B == 0 || OH == 0 || OW == 0) return QUANTO_HIP_OK;
  if (!x || !w || !scales || !y) return QUANTO_HIP_EINVAL;
  const int r = qbytes_conv2d_depthwise(x, w, scales, bias, y, B, cin, H, W, OC, KH, KW, OH, OW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, a_dtype,
                                        b_dtype, out_dtype, reinterpret_cast<hipStream_t>(stream));
  // (the form that ran names itself: "conv2d_depthwise" = quads, "conv2d_depthwise_strip" = 16-byte row chunks)
  return r;
}

int quanto_hip_qbits_conv2d(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t B, int64_t cin,
                            int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int stride_h, int stride_w, int pad_h,
                            int pad_w, int dil_h, int dil_w, int bits, int group_size, int dtype, int shift_dtype, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (B < 0 || cin <= 0 || H <= 0 || W <= 0 || OC <= 0 || KH <= 0 || KW <= 0 || OH < 0 || OW < 0) return QUANTO_HIP_EINVAL;
  if (stride_h <= 0 || stride_w <= 0 || pad_h < 0 || pad_w < 0 || dil_h <= 0 || dil_w <= 0) return QUANTO_HIP_EINVAL;
  if (OH != (H + 2 * pad_h - dil_h * (KH - 1) - 1) / stride_h + 1 || OW != (W + 2 * pad_w - dil_w * (KW - 1) - 1) / stride_w + 1) return QUANTO_HIP_EINVAL;
  bool int_shift = false;
  const int64_t K = cin * KH * KW;
  const int st = check_qbits(B * OH * OW, OC, K, bits, group_size, dtype, shift_dtype, &int_shift);
  if (st != QUANTO_HIP_OK) return st;
  if (B == 0 || OH == 0 || OW == 0) return QUANTO_HIP_OK;
  if (!x || !packed || !scale || !shift || !y) return QUANTO_HIP_EINVAL;
  const PackedGeom g = make_geom(OC, K, bits, group_size);
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  const size_t dense = conv2d_dense_weight_bytes(OC, K);
  if (qbits_conv2d_takes_rows(B, cin, W, OC, KH, KW, OH, OW, stride_w, dil_w) && workspace &&
      workspace_bytes >= dense && reinterpret_cast<uintptr_t>(workspace) % 16 == 0 && is_float_dtype(dtype) && dtype != QUANTO_HIP_F32) {
    // three-tap-wide windows at stride 1: dequantize once (the reference's own dense weight), then the row form of the convolution on it
    int r = dequantize_qbits_dispatch(packed, scale, shift, workspace, g, dtype, int_shift, hs);
    if (r == QUANTO_HIP_OK)
      r = qdense_conv2d_rows(x, workspace, bias, y, B, cin, H, W, OC, KH, KW, OH, OW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, dtype,
                             reinterpret_cast<uint8_t*>(workspace) + dense, workspace_bytes - dense, hs);
    if (r == QUANTO_HIP_OK) {
      set_last_kernel(bits == 4 ? "conv2d_rows_dequant_int4" : "conv2d_rows_dequant_int2");
      return r;
    }
    if (r != QUANTO_HIP_ENOTSUP) return r;
  }
  const int r = qbits_conv2d_mfma(x, packed, scale, shift, bias, y, B, cin, H, W, OC, KH, KW, OH, OW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, g, dtype,
                                  int_shift, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
  if (r == QUANTO_HIP_OK) set_last_kernel(bits == 4 ? "conv2d_mfma_int4" : "conv2d_mfma_int2");
  return r;
}

}  // extern "C"
