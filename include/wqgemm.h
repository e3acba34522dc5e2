/*
 * wqgemm -- C ABI of the MI355X (gfx950) weight-quantized GEMM engine.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  The C++
 * facade under include/ck/ (the DeviceGemm_dequantB / DeviceGemmMultipleD / ... interfaces of
 * Composable Kernel) and the Python ctypes binding both sit on these entry points.  Every device
 * pointer is a raw, non-owning HIP device pointer; every call is asynchronous on the given
 * stream unless stated otherwise; no entry point throws -- status codes only.
 *
 * Reference paths below are relative to the Composable Kernel tree.
 */
#ifndef WQGEMM_H
#define WQGEMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum wq_status
{
    WQ_OK              = 0,
    WQ_ERR_UNSUPPORTED = 1, /* valid problem the engine has no kernel for (IsSupportedArgument == false) */
    WQ_ERR_INVALID     = 2, /* malformed problem (null pointer, bad stride, bad enum)                     */
    WQ_ERR_HIP         = 3, /* a HIP runtime call failed; see wqgemm_last_error()                         */
    WQ_ERR_WORKSPACE   = 4  /* split-K requested without a workspace                                      */
} wq_status;

typedef enum wq_dtype
{
    WQ_F16       = 0, /* IEEE binary16                                                             */
    WQ_BF16      = 1, /* bfloat16; float -> bf16 is TRUNCATION (include/ck/utility/type_convert.hpp:53-64) */
    WQ_F8_FNUZ   = 2, /* E4M3 FNUZ bytes, bias 8, 0x80 = NaN (include/ck/utility/f8_utils.hpp:186-267)    */
    WQ_F32       = 3,
    WQ_I8        = 4, /* signed int8, one byte per element                                           */
    WQ_U8_OFFSET = 5, /* uint8 = int8 + 128 (example/64_fpAintB_gemm/common.hpp:51-83)               */
    WQ_I4_PACKED = 6, /* two offset-binary nibbles per byte, layout documented at wqgemm_pack_int4   */
    WQ_I32       = 7,
    WQ_F8_E4M3FN = 8, /* OCP E4M3 ("e4m3fn") bytes, bias 7, 0x7F/0xFF = NaN: the fp8 the gfx950 matrix core decodes; no counterpart in the reference, whose f8_t is FNUZ only */
    WQ_I4_PRESHUFFLED = 9, /* WQ_I4_PACKED's nibbles with its 16-byte units (32 consecutive k of one column) stored in the order the matrix-core kernels fetch them; layout documented at wqgemm_preshuffle_int4 */
    WQ_I8_PRESHUFFLED = 10 /* WQ_I8 bytes with their 16-byte units (16 consecutive k of one column) in that order; layout documented at wqgemm_preshuffle_int8 */
} wq_dtype;

typedef enum wq_layout
{
    WQ_ROW_MAJOR = 0, /* tensor_layout::gemm::RowMajor    (include/ck/tensor_operation/gpu/device/tensor_layout.hpp:13-24) */
    WQ_COL_MAJOR = 1  /* tensor_layout::gemm::ColumnMajor */
} wq_layout;

/* Epilogue functors (include/ck/tensor_operation/gpu/element/...) applied to the fp32/int32
 * accumulator before the store. */
typedef enum wq_epilogue
{
    WQ_EPI_PASSTHROUGH                = 0, /* unary_element_wise_operation.hpp:28-225           */
    WQ_EPI_FASTGELU                   = 1, /* unary_element_wise_operation.hpp:435-546          */
    WQ_EPI_MULTIPLY_ADD_FASTGELU      = 2, /* element_wise_operation.hpp:277-295; d0[n], d1[n] of C's type */
    WQ_EPI_MULTIPLY_MULTIPLY          = 3, /* element_wise_operation.hpp:252-275; d0[n], d1[m] float       */
    WQ_EPI_MULTIPLY_MULTIPLY_FASTGELU = 4, /* FastGelu(c*d0[n]*d1[m]); d0, d1 float                        */
    WQ_EPI_ACT_MUL_CLAMP              = 5, /* quantization_operation.hpp:24-66; requant_scale, int8 out    */
    WQ_EPI_ADD                        = 6, /* binary_element_wise_operation.hpp:78-85: c + d0[n]; d0 of C's type */
    WQ_EPI_ADD_FASTGELU               = 7, /* binary_element_wise_operation.hpp:511-522: FastGelu(c + d0[n])     */
    WQ_EPI_MULTIPLY                   = 8, /* binary_element_wise_operation.hpp Multiply<C>(float, C): c * d0[n] */
    WQ_EPI_MULTIPLY_FASTGELU          = 9, /* binary_element_wise_operation.hpp:578-590: FastGelu(c * d0[n])     */
    /* int8 requantisation with Activation = Relu and/or an int32 per-column bias d0[n]
     * (quantization_operation.hpp:24-66,131-171; unary_element_wise_operation.hpp:409-428;
     *  example/14_gemm_quantization/gemm_xdl_bias_relu_quantization_int8.cpp:30-32,223-229) */
    WQ_EPI_RELU_MUL_CLAMP             = 10, /* Activation_Mul_Clamp<Relu>                                  */
    WQ_EPI_ADD_ACT_MUL_CLAMP          = 11, /* Add_Activation_Mul_Clamp<PassThrough>: int32 d0[n]          */
    WQ_EPI_ADD_RELU_MUL_CLAMP         = 12, /* Add_Activation_Mul_Clamp<Relu>: int32 d0[n]                 */
    /* element_wise_operation.hpp:200-231, the bhalf_t specialisation MultiplyAdd<bhalf_t, float, bhalf_t, bhalf_t> of
     * the multi-ABD "multiply ... bias" instance (gemm_multi_abd.hpp:198-209): "type_convert<bhalf_t>(c) * d0 + d1" with
     * bhalf_t = unsigned short, i.e. integer arithmetic on the three 16-bit patterns, low 16 bits kept.  Reproduced
     * bit for bit; bf16 C only; d0, d1 of C's type. */
    WQ_EPI_MULTIPLY_ADD               = 13,
    /* unary_element_wise_operation.hpp:236-251 ConvertBF16RTN -> type_convert.hpp:507-549 bf16_convert_rtn<bhalf_t, float>:
     * the accumulator leaves as bf16 rounded to nearest even instead of truncated (Inf / NaN patterns kept as the
     * reference keeps them); the C operation of example/01_gemm/gemm_xdl_bf16_rtn.cpp:21.  bf16 C only; no D operand. */
    WQ_EPI_CONVERT_BF16_RTN           = 14
} wq_epilogue;

/* wqgemm_problem.flags */
#define WQ_FLAG_TILE_128 1 /* force the 128 x 128 x 64 MFMA kernel */
#define WQ_FLAG_TILE_256 2 /* force the 256 x 256 x 64 MFMA kernel */
#define WQ_FLAG_LOCKSTEP 4   /* 256-tile kernel without the ping-pong wave schedule (tuning A/B) */
#define WQ_FLAG_WAVES_2X4 8  /* 256-tile kernel with a 2 x 4 wave grid instead of 1 x 8 (tuning A/B) */
#define WQ_FLAG_DIRECT_STORE 32 /* 256-tile kernel storing accumulator fragments directly, no LDS-staged epilogue (tuning A/B) */
#define WQ_FLAG_DEQUANT_IN_L 512 /* 256-tile ping-pong kernel dequantising in the load sections, LDS-DMA in the second MFMA section (tuning A/B) */
#define WQ_FLAG_GENERIC 1024     /* force the generic (one thread per output) kernel */
#define WQ_FLAG_F8_NO_SPECIALS 2048 /* WQ_F8_FNUZ activations hold none of the bytes 0x7F, 0xFF (+-240) and 0x80 (NaN): allows the fp8 matrix-core kernel, which reads FNUZ bytes as E4M3FN / 2 and cannot express those three */
#define WQ_FLAG_MFMA_32X32 4096 /* 256-tile kernel built on v_mfma_f32_32x32x16 (packed int4 weights; tuning A/B) */
#define WQ_FLAG_PINGPONG 8192 /* fp8 x int4 kernel in its four-section ping-pong form instead of the software-pipelined default (tuning A/B) */
#define WQ_FLAG_GROUP8 64       /* walk tiles in groups of 8 tile rows instead of the cost-minimising group (tuning A/B) */

/* One GEMM problem: C[M,N] = epilogue( sum_k A[m,k] * dequant(B[k,n], scale[n]) ).
 * Mirrors what Argument objects capture in the reference
 * (include/ck/tensor_operation/gpu/device/impl/device_fpAintB_gemm_wmma.hpp:343-417):
 * raw pointers, sizes, strides in ELEMENTS of the named layout, functor state.
 * B strides count logical elements; a packed-int4 column therefore occupies ldb/2 bytes. */
typedef struct wqgemm_problem
{
    int32_t a_dtype, b_dtype, scale_dtype, c_dtype; /* wq_dtype                        */
    int32_t a_layout, b_layout, c_layout;           /* wq_layout                       */
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    const void* a;
    const void* b;
    const void* scale; /* N values of scale_dtype (lengths (K,N), strides (0,1)); NULL = unit scale: the epilogue carries the scales */
    void* c;
    int32_t epilogue; /* wq_epilogue */
    float requant_scale;
    const void* d0;
    const void* d1;
    /* Split-K.  k_batch > 1 splits K inside one GPU: slice z reads [z*KRead, min(K,(z+1)*KRead)),
     * KRead = ceil(K/(k_batch*64))*64, 128 in place of 64 for fp8 activations (gridwise_gemm_xdl_cshuffle_v3.hpp:192-197,583-590), fp32
     * partials go to the caller's workspace and a reduce kernel applies the epilogue
     * (device_gemm_xdl_cshuffle_v3r1.hpp:209-270).
     * k_begin/k_end (k_end > k_begin) restrict the whole call to one K slice and make it store
     * the raw fp32 partial to c (c_dtype must be WQ_F32, ldc in floats): the per-GPU step of
     * multi-GPU split-K.  Both zero = the whole K range. */
    int32_t k_batch;
    int32_t k_begin, k_end;
    int32_t flags; /* 0, or one of WQ_FLAG_TILE_* to pin the tile shape (testing / tuning) */
    /* Leading dimensions of d0 / d1 in elements; 0 (the default) = the broadcast forms named at wq_epilogue.
     * Non-zero: a full (M, N) D tensor as CK's descriptors address it
     * (include/ck/tensor_operation/gpu/device/impl/device_gemm_multiple_d_xdl_cshuffle.hpp:206-229) -- the 16-bit
     * epilogues (2, 6 - 9) read row-major d[m*ldd + n]; MultiplyMultiply (3, 4) reads d0 row-major d0[m*ldd0 + n]
     * and d1 column-major d1[m + n*ldd1] (the DsLayout Tuple<Row, Col> of
     * example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8.cpp:43-47).  Ignored by the int32 bias of the
     * requantisation epilogues (always one value per column). */
    int32_t ldd0, ldd1;
    /* Group-wise weight scales.  scale_group_k = 0 (the default): one scale per column, `scale` holds N values.
     * scale_group_k = g > 0: one scale per column and per group of g consecutive k -- `scale` is a row-major
     * (ceil(K / g), N) tensor with row stride ld_scale >= N (0 = N) and the weight at (k, n) is dequantised with
     * scale[(k / g) * ld_scale + n].  In the reference's terms this is the (K, N) scale tensor of
     * ReferencefpAintBGemm (library/include/ck/library/reference_tensor_operation/cpu/reference_fpAintB_gemm.hpp:62-127)
     * with rows repeating g times, per-channel scales being its strides-(0,1) case; the (K / block, N) shape follows
     * the B1 scale tensor of example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:126-152.
     * With bf16 activations it is the B1 operand of the Multiply-on-B form
     * (example/60_gemm_multi_ABD/gemm_multi_ABD_xdl_fastgelu_bf16_i8.cpp:137-139) with rows that repeat per group instead
     * of a zero stride.  fp16 and bf16 activations.  Any g >= 1 runs on the general kernel; g a multiple of 64 (64, 128,
     * ...) on the 128- and 256-tile matrix-core kernels; g a power of two >= 32 (packed int4) or >= 16 (int8) on the
     * small-M kernels.  k (and so the group index) counts from the start of the K range also for K slices and split-K. */
    int32_t scale_group_k, ld_scale;
    /* Block-wise scales of BOTH operands, for fp8 activations (a_dtype WQ_F8_FNUZ / WQ_F8_E4M3FN; `scale` stays NULL):
     * the problem of example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:58-75,126-152,262-283
     * (DeviceGemmMultiD_ABScale, include/ck/tensor_operation/gpu/device/device_gemm_multiple_d_ab_scale.hpp) with int4 / int8
     * weights in place of its fp8 weights.  Both non-NULL selects it:
     *   a'(m, k) = float(a(m, k)) * a_block_scale[(m / scale_block_m) * ld_a_scale + k / scale_block_k]     (row-major)
     *   b'(k, n) = float(q(k, n)) * b_block_scale[k / scale_block_k + (n / scale_block_n) * ld_b_scale]     (column-major,
     *              the descriptor the example gives its B1 tensor under B0Layout = ColumnMajor)
     *   C = epilogue( sum_k a'(m, k) * b'(k, n) ),  fp32 products and sums, k ascending.
     * The general kernel evaluates exactly that (bit-exact with the reference's host loop).  The fp8 matrix-core kernel
     * (packed int4 weights, scale_block_k = 128, scale_block_n a multiple of 32, any scale_block_m -- 1 is one scale per
     * token --, no more than 640 K tiles) sums each 128-k block exactly on the matrix core and applies a_scale * b_scale to the block's sum: the same value up
     * to fp32 rounding (identical when the scales are powers of two).  ld_a_scale >= ceil(K / scale_block_k),
     * ld_b_scale >= ceil(K / scale_block_k); 0 = exactly that.  k_batch and K slices (k_begin / k_end) combine with it: the tables
     * are indexed with the absolute k, and the fp32 partials carry the scales. */
    const void* a_block_scale;
    const void* b_block_scale;
    int32_t scale_block_m, scale_block_n, scale_block_k;
    int32_t ld_a_scale, ld_b_scale;
} wqgemm_problem;

/* Library version string, e.g. "wqgemm 0.1 (gfx950)". */
const char* wqgemm_version(void);

/* sizeof(wqgemm_problem) as the library was built with.  wqgemm_problem grows at its end from release to release;
 * a caller compiled against another header must not hand its struct over (the library would read fields the caller
 * never wrote).  Bindings compare this with their own size when they load the library.  No counterpart in the
 * reference, whose Argument objects never cross a library boundary. */
size_t wqgemm_problem_size(void);

/* Message of the last failure on the calling thread ("" if none). */
const char* wqgemm_last_error(void);

/* 1 if wqgemm_launch can run the problem, else 0.
 * Replaces BaseOperator::IsSupportedArgument
 * (include/ck/tensor_operation/gpu/device/device_base.hpp:46;
 *  impl/device_fpAintB_gemm_wmma.hpp:510-597). */
int wqgemm_is_supported(const wqgemm_problem* p);

/* Bytes of device workspace wqgemm_launch can use:
 *  - k_batch*M*N*4 when the caller sets k_batch > 1 (required: the launch fails with WQ_ERR_WORKSPACE
 *    without it), plus K*N for tile-aligned row-major (K x N) 8-bit weights (their K-contiguous copy follows the slabs);
 *  - kb*M*N*4 when the planner itself splits K (small-M problems; grids shorter than the chip on the 128- and 256-tile
 *    kernels, the fp8 kernel and the int8 x int8 128-tile and small-M kernels, whose partials are int32; kb chosen so the
 *    grid fills the chip;
 *    optional: without it the launch runs the whole K range as one slice);
 *  - (M*K + N)*2 for tile-aligned WQ_F8_FNUZ activations without WQ_FLAG_F8_NO_SPECIALS (optional: holds the
 *    fp16 image of A for the fp16 matrix-core kernel), and for fp8 activations of either format at M <= 16 when the
 *    image is too large for the small-M streaming kernel to build in LDS by itself (optional: without it the fp8
 *    matrix-core kernel runs);
 *  - K*N for tile-aligned row-major (K x N) 8-bit weights (optional: holds the K-contiguous transpose the
 *    matrix-core kernels read);
 *  - else 0.
 * Without an optional workspace the launch uses the general kernel, which gives the same results far slower.
 * Replaces BaseOperator::GetWorkSpaceSize
 * (device_base.hpp:60; impl/device_gemm_xdl_cshuffle_v3r1.hpp:686-694). */
size_t wqgemm_workspace_bytes(const wqgemm_problem* p);

/* Enqueue the GEMM on `stream` (a hipStream_t; NULL = default stream).  Asynchronous, no
 * synchronisation, no allocation.  Replaces Invoker::Run with time_kernel == false
 * (impl/device_fpAintB_gemm_wmma.hpp:424-493; include/ck/host_utility/kernel_launch.hpp:71-77). */
int wqgemm_launch(const wqgemm_problem* p, void* workspace, void* stream);

/* wqgemm_launch for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): no workspace argument -- what wqgemm_launch would take from the caller's
 * workspace (split-K slabs, the fp16 image of FNUZ activations, the K-contiguous copy of row-major weights) comes
 * from a scratch array built into the library, WQGEMM_NOALLOC_SCRATCH_BYTES (64 MiB) per GPU, resident on every GPU the
 * library runs on until the process ends.  No counterpart in the reference.  Calls hipMalloc / hipFree never.
 * A problem with k_batch > 1 whose wqgemm_workspace_bytes exceeds the scratch returns WQ_ERR_WORKSPACE with a message
 * and launches nothing; an OPTIONAL workspace use that does not fit is dropped exactly as wqgemm_launch drops it
 * for a NULL workspace.  Results are those of wqgemm_launch with a workspace, bit for bit.  All calls on one GPU
 * share that GPU's scratch: nothing in it needs initialising (each launch writes every byte it later reads), but two
 * calls that overlap on one GPU -- different streams or threads -- race on it; serialise them.  The first call on a
 * device that uses the scratch looks its address up (hipGetSymbolAddress, once per device): make one such call
 * outside stream capture before capturing. */
#define WQGEMM_NOALLOC_SCRATCH_BYTES (64u << 20)
int wqgemm_launch_noalloc(const wqgemm_problem* p, void* stream);

/* cold_iters untimed launches, then nrepeat launches between one pair of HIP events on
 * `stream`; *avg_ms receives the mean.  Synchronises the stream.  Replaces Invoker::Run with
 * time_kernel == true (include/ck/host_utility/kernel_launch.hpp:20-70;
 * StreamConfig defaults cold 5 / repeat 50: include/ck/stream_config.hpp:9-19). */
int wqgemm_run_timed(const wqgemm_problem* p, void* workspace, void* stream, int cold_iters,
                     int nrepeat, float* avg_ms);

/* The timing mode that keeps cache reuse across iterations out of the figure: rotating_count copies of A and B
 * (the originals plus rotating_count - 1 device copies this call allocates and frees; at most 64) are used in
 * turn, each of the nrepeat launches is bracketed by its own pair of HIP events after a device synchronisation,
 * and *median_ms receives the median.  Replaces launch_and_time_kernel_with_preprocess + RotatingMemWrapper with
 * StreamConfig::flush_cache set (include/ck/host_utility/flush_cache.hpp:17-95,107-235; used from
 * impl/device_gemm_xdl_cshuffle_v3r1.hpp:339-374).  The reference's instruction-cache invalidation kernel
 * between iterations has no counterpart here. */
int wqgemm_run_timed_rotating(const wqgemm_problem* p, void* workspace, void* stream, int cold_iters,
                              int nrepeat, int rotating_count, float* median_ms);

/* Instance name, written to buf (NUL-terminated, truncated to n).  Replaces
 * BaseOperator::GetTypeString (device_base.hpp:47; impl/device_fpAintB_gemm_wmma.hpp:662-706). */
int wqgemm_type_string(const wqgemm_problem* p, char* buf, size_t n);

/* Name of the __global__ function wqgemm_launch dispatches the main GEMM to (as it appears in
 * rocprofv3 kernel traces).  No counterpart in the reference. */
int wqgemm_kernel_name(const wqgemm_problem* p, char* buf, size_t n);

/* Pack int4 weights held one per int8 byte (values -8..7, the reference's int4 convention:
 * include/ck/utility/data_type.hpp:10-14, example/64_fpAintB_gemm/run_gemm_example.inc:8-10)
 * into WQ_I4_PACKED.  Source: (K,N) column-major, element (k,n) at q[n*ldq + k] (device memory).
 * Destination: column n at out + n*ld_bytes, K/2 bytes; nibble = q + 8 (offset binary, the int4
 * analogue of example/64_fpAintB_gemm/common.hpp:66-74); inside each little-endian 32-bit word
 * (8 consecutive k starting at 8g) nibble p (bits 4p..4p+3) holds k = 8g + 2p for p < 4 and
 * k = 8g + 2(p-4) + 1 for p >= 4.  K must be a multiple of 8.  Values are not range-checked on
 * the device (they are masked to 4 bits after the +8). */
int wqgemm_pack_int4(const int8_t* q, int32_t ldq, int32_t K, int32_t N, uint8_t* out,
                     int32_t ld_bytes, void* stream);

/* Host-memory variant of wqgemm_pack_int4 for offline weight preparation; additionally returns
 * WQ_ERR_INVALID if a value lies outside -8..7. */
int wqgemm_pack_int4_host(const int8_t* q, int32_t ldq, int32_t K, int32_t N, uint8_t* out,
                          int32_t ld_bytes);

/* Re-order WQ_I4_PACKED weights (column n at packed + n*ld_bytes) into WQ_I4_PRESHUFFLED, the offline layout in
 * which every wave-wide weight fetch of the tile kernels is one contiguous kilobyte (the role of the pre-shuffled B of
 * example/65_gemm_multiply_multiply's newer drivers; the reference tree here has no int4 form of it).  The unit that
 * moves is 16 bytes = the 32 consecutive k [32j, 32j+32) of one column, nibble order inside it as in WQ_I4_PACKED.
 * Block (cb, kb) = columns [32cb, 32cb+32) x k [64kb, 64kb+64) = 64 units = 1024 bytes at
 *     out + cb * (ldk * 16) + kb * 1024,
 * and inside a block the unit of column 32cb + 16t + r (t = 0, 1; r = 0..15), k half h (k = 64kb + 32h ..) sits at
 * byte (32t + 16h + r) * 16.  ldk (elements, a multiple of 64, >= K) is the K pitch: wqgemm_problem.ldb of the GEMM
 * that consumes the result.  Needs N % 32 == 0 and K % 64 == 0.  Device memory, asynchronous on `stream`. */
int wqgemm_preshuffle_int4(const uint8_t* packed, int32_t ld_bytes, int32_t K, int32_t N, uint8_t* out, int32_t ldk,
                           void* stream);

/* The same for 8-bit weights: WQ_I8 (K contiguous per column, column n at b + n*ldb) -> WQ_I8_PRESHUFFLED.  The unit
 * is 16 bytes = the 16 consecutive k [16j, 16j+16) of one column; block (cb, kb) = columns [16cb, 16cb+16) x
 * k [64kb, 64kb+64) = 64 units = 1024 bytes at out + cb * (ldk * 16) + kb * 1024, and inside a block the unit of column
 * 16cb + c, k quarter j (k = 64kb + 16j ..) sits at byte (4c + j) * 16.  ldk (a multiple of 64, >= K) is the K pitch:
 * wqgemm_problem.ldb of the consuming GEMM.  Needs N % 16 == 0 and K % 64 == 0. */
int wqgemm_preshuffle_int8(const int8_t* b, int32_t ldb, int32_t K, int32_t N, int8_t* out, int32_t ldk, void* stream);
int wqgemm_preshuffle_int8_host(const int8_t* b, int32_t ldb, int32_t K, int32_t N, int8_t* out, int32_t ldk);

/* Host-memory variant of wqgemm_preshuffle_int4 for offline weight preparation. */
int wqgemm_preshuffle_int4_host(const uint8_t* packed, int32_t ld_bytes, int32_t K, int32_t N, uint8_t* out,
                                int32_t ldk);

/* Sum k_batch fp32 partials [k_batch][M][N] (row-major, ld = N) in ascending batch order, apply
 * p's epilogue and store p->c.  With k_batch == 1 this is the final convert/epilogue step after
 * a multi-GPU reduction.  Replaces RunReduce
 * (impl/device_gemm_xdl_cshuffle_v3r1.hpp:209-270). */
int wqgemm_reduce_finalize(const wqgemm_problem* p, const float* partials, int32_t k_batch,
                           void* stream);

/* Multi-GPU split-K, one step on ONE rank (one process per GPU; every rank of the communicator makes the same call
 * on its own stream).  The reference has no communication code; the flow keeps the shape of its in-GPU split-K --
 * partial GEMMs, fp32 slabs, sum, convert (impl/device_gemm_xdl_cshuffle_v3r1.hpp:209-270) -- with the sum done by
 * RCCL over xGMI:
 *   1. this rank's K slice -> fp32 partial [M][N]         (p describes the slice: a / b point at it, K = its length;
 *                                                          K = 0 or a = NULL: an empty slice, zeros)
 *   2. ncclReduceScatter(fp32, sum): rank r keeps rows [r M/G, (r+1) M/G) of the sum
 *   3. wqgemm_reduce_finalize of those rows into p->c     (p's epilogue; D operands addressed by global row)
 *   4. ncclAllGather in place: p->c [M][N] complete on every rank
 * p->c must be contiguous (ldc == N), fp16 or bf16; M a multiple of nranks; k_batch / k_begin / k_end unset.
 * nccl_comm: an ncclComm_t of nranks ranks (RCCL is bound at run time from the copy already loaded in the process,
 * else librccl.so.1); NULL = no exchange, the rank's own partial stands for the sum (single-process rehearsals).
 * workspace: wqgemm_splitk_multi_workspace_bytes(p, nranks) = (M + M/nranks) * N * 4 bytes of device memory.
 * Slices follow wqgemm_splitk_kread(K_total, nranks, 64): rank g owns [g KRead, min(K_total, (g+1) KRead)). */
size_t wqgemm_splitk_multi_workspace_bytes(const wqgemm_problem* p, int32_t nranks);
int wqgemm_splitk_multi_step(const wqgemm_problem* p, void* nccl_comm, int32_t rank, int32_t nranks, void* workspace,
                             void* stream);

/* KRead of the split-K partition: ceil(K/(k_batch*granule))*granule
 * (gridwise_gemm_xdl_cshuffle_v3.hpp:192-197). */
int32_t wqgemm_splitk_kread(int32_t K, int32_t k_batch, int32_t granule);

#ifdef __cplusplus
}
#endif
#endif /* WQGEMM_H */
