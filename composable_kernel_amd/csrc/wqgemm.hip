// C ABI of the gfx950 weight-quantized GEMM engine (declared in include/wqgemm.h): problem
// validation, kernel selection, launches and timing.  No allocation and no synchronisation on the
// launch path; errors are reported by status code plus a thread-local message.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <type_traits>

#define WQ_MAIN_TU
#include "wqgemm.h"
#include "wq_kernels.h"

// The 256-tile kernels are compiled one translation unit per (activation, weight format) pair
// (wq_inst_256.hip) so that the build runs in parallel.
namespace wq {
#define WQ_DECL_256(AK, BK) int launch_256_##AK##_##BK(const GemmParams& gp, dim3 grid, hipStream_t s, int variant);
WQ_DECL_256(0, 0) WQ_DECL_256(0, 1) WQ_DECL_256(0, 2) WQ_DECL_256(1, 0) WQ_DECL_256(1, 1) WQ_DECL_256(1, 2)
#undef WQ_DECL_256
int launch_256_f8(const GemmParams& gp, dim3 grid, hipStream_t s, int variant); // wq_inst_256_f8.hip
int launch_256_w32(const GemmParams& gp, dim3 grid, hipStream_t s, int ak);    // wq_inst_256_w32.hip
int launch_skinny(int ak, int bk, const GemmParams& gp, dim3 grid, hipStream_t s, int nmb, int p); // wq_inst_skinny.hip
int launch_skinny_ws(int ak, int bk, const GemmParams& gp, dim3 grid, int lds_bytes, hipStream_t s, int p);
} // namespace wq

// Scratch of wqgemm_launch_noalloc: lives in the code object, so every GPU the library's kernels are loaded on holds
// its own copy, at its own address, until the process ends.  `scratch` itself is only a host-side handle for
// hipGetSymbolAddress; the kernels receive the address that call returns.
static __device__ __align__(256) unsigned char scratch[WQGEMM_NOALLOC_SCRATCH_BYTES];

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define WQ_HIP(expr)                                                                         \
    do                                                                                       \
    {                                                                                        \
        hipError_t e_ = (expr);                                                              \
        if(e_ != hipSuccess)                                                                 \
            return fail(WQ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));          \
    } while(0)

thread_local void* g_dbg = nullptr; // diagnostic stamp buffer, set only inside wqgemm_debug_section_cycles on the calling thread

enum Route
{
    ROUTE_NONE = 0,
    ROUTE_MFMA128,
    ROUTE_MFMA256,
    ROUTE_I8_MFMA128,
    ROUTE_I8_SKINNY,
    ROUTE_I8_MFMA256,
    ROUTE_F8_MFMA256,
    ROUTE_SKINNY,
    ROUTE_SKINNY_WS,
    ROUTE_GENERIC
};

struct Plan
{
    Route route   = ROUTE_NONE;
    int ak        = 0; // wq::AKind
    int bk        = 0; // wq::BKind
    int k_begin   = 0;
    int k_end     = 0;
    int k_batch   = 1;
    int k_per_z   = 0;
    bool partial  = false; // main kernel writes fp32 partial slabs
    bool to_c     = false; // ... directly into p->c (multi-GPU slice) instead of the workspace
    bool auto_split = false; // split-K chosen by the planner (small-M kernel, short grids of the 128-tile kernel): needs, and is dropped without, a workspace
    bool f8_fused   = false; // fp8 activations widened inside the small-M whole-image kernel (no image in the workspace)
    bool split_256  = false; // the planner's split-K is what put this problem on the 256-tile kernel
    bool via_colb = false; // row-major 8-bit B: transpose into the workspace, then the K-contiguous matrix-core kernels
    bool via_f16  = false; // fp8 FNUZ activations: convert to fp16 in the workspace, then the fp16 MFMA kernel
};

bool is16(int dt) { return dt == WQ_F16 || dt == WQ_BF16; }
bool is_i4(int dt) { return dt == WQ_I4_PACKED || dt == WQ_I4_PRESHUFFLED; } // two nibbles per byte, either unit order
bool is_preshuf(int dt) { return dt == WQ_I4_PRESHUFFLED || dt == WQ_I8_PRESHUFFLED; }
bool has_block_scales(const wqgemm_problem* p) { return p->a_block_scale != nullptr || p->b_block_scale != nullptr; }

// Epilogues of the floating-point family and the D vectors each one reads.
int check_fp_epilogue(const wqgemm_problem* p)
{
    switch(p->epilogue)
    {
    case WQ_EPI_PASSTHROUGH:
    case WQ_EPI_FASTGELU: break;
    case WQ_EPI_CONVERT_BF16_RTN:
        if(p->c_dtype != WQ_BF16)
            return fail(WQ_ERR_UNSUPPORTED, "ConvertBF16RTN produces bf16 outputs only");
        break;
    case WQ_EPI_MULTIPLY_ADD:
        if(p->c_dtype != WQ_BF16)
            return fail(WQ_ERR_UNSUPPORTED, "the bit-pattern MultiplyAdd epilogue exists for bf16 outputs only");
        [[fallthrough]];
    case WQ_EPI_MULTIPLY_ADD_FASTGELU:
    case WQ_EPI_MULTIPLY_MULTIPLY:
    case WQ_EPI_MULTIPLY_MULTIPLY_FASTGELU:
        if(!p->d0 || !p->d1)
            return fail(WQ_ERR_INVALID, "epilogue %d needs d0 and d1", p->epilogue);
        break;
    case WQ_EPI_ADD:
    case WQ_EPI_ADD_FASTGELU:
    case WQ_EPI_MULTIPLY:
    case WQ_EPI_MULTIPLY_FASTGELU:
        if(!p->d0)
            return fail(WQ_ERR_INVALID, "epilogue %d needs d0", p->epilogue);
        break;
    default: return fail(WQ_ERR_UNSUPPORTED, "epilogue %d is not valid for this dtype", p->epilogue);
    }
    if(p->ldd0 < 0 || p->ldd1 < 0)
        return fail(WQ_ERR_INVALID, "ldd0 and ldd1 must be 0 (broadcast) or a positive leading dimension");
    // A full D tensor whose leading dimension is shorter than its rows (columns for MultiplyMultiply's column-major d1)
    // would alias: refused rather than read as given.
    const bool mm = p->epilogue == WQ_EPI_MULTIPLY_MULTIPLY || p->epilogue == WQ_EPI_MULTIPLY_MULTIPLY_FASTGELU;
    if(p->ldd0 > 0 && p->ldd0 < p->N)
        return fail(WQ_ERR_INVALID, "ldd0 = %d is shorter than a row of d0 (N = %d)", p->ldd0, p->N);
    if(p->ldd1 > 0 && p->ldd1 < (mm ? p->M : p->N))
        return fail(WQ_ERR_INVALID, "ldd1 = %d is shorter than a %s of d1 (%d)", p->ldd1, mm ? "column" : "row", mm ? p->M : p->N);
    return WQ_OK;
}

// Validate and choose a kernel.  Returns WQ_OK and fills plan, or an error status with message.
int make_plan(const wqgemm_problem* p, Plan& plan, bool have_workspace = true)
{
    if(!p)
        return fail(WQ_ERR_INVALID, "null problem");
    if(p->M <= 0 || p->N <= 0 || p->K <= 0)
        return fail(WQ_ERR_INVALID, "M, N, K must be positive (got %d, %d, %d)", p->M, p->N, p->K);
    if(!p->a || !p->b || !p->c)
        return fail(WQ_ERR_INVALID, "a, b and c must be non-null device pointers");
    if(p->flags & ~(WQ_FLAG_TILE_128 | WQ_FLAG_TILE_256 | WQ_FLAG_LOCKSTEP | WQ_FLAG_WAVES_2X4 | WQ_FLAG_DIRECT_STORE | WQ_FLAG_GROUP8 | WQ_FLAG_DEQUANT_IN_L | WQ_FLAG_GENERIC | WQ_FLAG_F8_NO_SPECIALS | WQ_FLAG_MFMA_32X32 | WQ_FLAG_PINGPONG))
        return fail(WQ_ERR_INVALID, "unknown bits in flags");
    if(p->c_layout != WQ_ROW_MAJOR)
        return fail(WQ_ERR_UNSUPPORTED, "only row-major C is implemented");
    if(p->a_layout != WQ_ROW_MAJOR && p->a_layout != WQ_COL_MAJOR)
        return fail(WQ_ERR_INVALID, "bad a_layout");
    if(p->b_layout != WQ_ROW_MAJOR && p->b_layout != WQ_COL_MAJOR)
        return fail(WQ_ERR_INVALID, "bad b_layout");
    // Input strides below the row length alias rows; the reference's descriptors accept them
    // (client_example/30_gemm_bf16Aint8B/gemm_xdl_bf16_i8.cpp:82 passes StrideB = K for a row-major (K, N) B), so
    // they are addressed as given.  The output stride must keep rows apart.
    if(p->lda <= 0 || p->ldb <= 0)
        return fail(WQ_ERR_INVALID, "lda and ldb must be positive");
    if(is_i4(p->b_dtype) && p->ldb < p->K)
        return fail(WQ_ERR_INVALID, "ldb too small for packed int4 columns");
    if(p->ldc < p->N)
        return fail(WQ_ERR_INVALID, "ldc too small");
    const int kb = p->k_batch <= 0 ? 1 : p->k_batch;
    plan.k_batch = kb;
    const bool slice = p->k_end > p->k_begin;
    if(!slice && (p->k_begin != 0 || p->k_end != 0))
        return fail(WQ_ERR_INVALID, "k_begin/k_end must both be 0 or satisfy k_end > k_begin");
    plan.k_begin = slice ? p->k_begin : 0;
    plan.k_end   = slice ? p->k_end : p->K;
    if(plan.k_begin < 0 || plan.k_end > p->K)
        return fail(WQ_ERR_INVALID, "K slice [%d,%d) outside [0,%d)", plan.k_begin, plan.k_end, p->K);
    if(slice && kb != 1)
        return fail(WQ_ERR_UNSUPPORTED, "k_batch > 1 cannot be combined with a K slice");
    if(slice && p->c_dtype != WQ_F32)
        return fail(WQ_ERR_INVALID, "a K-slice launch stores fp32 partials: c_dtype must be WQ_F32");
    plan.partial = slice || kb > 1;
    plan.to_c    = slice;
    const int klen = plan.k_end - plan.k_begin;
    // (slices in whole K tiles of the kernel that will run them: 128 k on the fp8 matrix core, 64 k elsewhere)
    const int k_tile = p->a_dtype == WQ_F8_FNUZ || p->a_dtype == WQ_F8_E4M3FN ? 128 : 64;
    plan.k_per_z     = kb > 1 ? wqgemm_splitk_kread(klen, kb, k_tile) : klen;
    // (trailing slices may be empty when K is small: they store zero partials)
    if(plan.partial && (p->N % 4 != 0 || p->ldc % 4 != 0))
        return fail(WQ_ERR_UNSUPPORTED, "split-K needs N and ldc to be multiples of 4");

    // ---- int8 x int8 requantised GEMM (example/14, client_example/09) ------------------------
    if(p->a_dtype == WQ_I8 && has_block_scales(p))
        return fail(WQ_ERR_UNSUPPORTED, "block-wise scales of both operands go with fp8 activations");
    if(p->a_dtype == WQ_I8)
    {
        const bool requant_epi = p->epilogue == WQ_EPI_ACT_MUL_CLAMP || p->epilogue == WQ_EPI_RELU_MUL_CLAMP ||
                                 p->epilogue == WQ_EPI_ADD_ACT_MUL_CLAMP || p->epilogue == WQ_EPI_ADD_RELU_MUL_CLAMP;
        if(p->b_dtype != WQ_I8 || p->c_dtype != WQ_I8 || !requant_epi)
            return fail(WQ_ERR_UNSUPPORTED, "int8 A needs int8 B, int8 C and an (Add_)Activation_Mul_Clamp epilogue");
        if((p->epilogue == WQ_EPI_ADD_ACT_MUL_CLAMP || p->epilogue == WQ_EPI_ADD_RELU_MUL_CLAMP) && !p->d0)
            return fail(WQ_ERR_INVALID, "epilogue %d needs the int32 bias d0", p->epilogue);
        if(plan.partial)
            return fail(WQ_ERR_UNSUPPORTED, "the int8 requant GEMM takes no k_batch or K slice from the caller (the planner splits short grids itself)");
        const bool tiled_i8 = p->a_layout == WQ_ROW_MAJOR && p->b_layout == WQ_COL_MAJOR &&
                              p->N % 4 == 0 && p->K % 128 == 0 && p->lda % 16 == 0 && p->ldb % 16 == 0 &&
                              p->ldc % 4 == 0 && (uintptr_t)p->a % 16 == 0 && (uintptr_t)p->b % 16 == 0 &&
                              (uintptr_t)p->c % 4 == 0 && !(p->flags & WQ_FLAG_GENERIC);
        plan.route = tiled_i8 ? ROUTE_I8_MFMA128 : ROUTE_GENERIC;
        // 256 x 256 tiles once they cover at least half the chip and the 128-tiles no longer fit one round of two
        // workgroups per CU (up to 512 of them the 128-tile kernel is 5 - 20 % faster: tools/mid_m_sweep.py --i8a), and
        // on request
        const long wgs128 = (long)((p->M + 127) / 128) * ((p->N + 127) / 128);
        if(tiled_i8 && p->N % 256 == 0 && p->ldc % 16 == 0 && (uintptr_t)p->c % 16 == 0 && !(p->flags & WQ_FLAG_TILE_128) &&
           (((long)((p->M + 255) / 256) * (p->N / 256) >= 128 && wgs128 > 512) || (p->flags & WQ_FLAG_TILE_256)))
            plan.route = ROUTE_I8_MFMA256;
        // A short grid on the 128-tile kernel: the planner's own split of K across blockIdx.z (int32 partials in the
        // optional workspace, requantised by wq_reduce_requant_i8; one slice without a workspace).  Two workgroups
        // share a CU, so the slice count is the one with the shortest chain on the busiest CU, as for the 16-bit
        // 128-tile kernel below; slices in whole 128-k tiles, at least 512 k (1024 k once a CU holds two).
        // (up to 256 workgroups; two halves of K for 257 ... 511 workgroups, which pay on the 16-bit kernel, lose 15 - 30 % here)
        if(plan.route == ROUTE_I8_MFMA128 && wgs128 <= 256 && !(p->flags & WQ_FLAG_TILE_128))
        {
            int best_score = 0, kb = 1;
            for(int c = 1; c <= 16 && wgs128 * c <= 512; ++c)
            {
                const int chain = (int)((wgs128 * c + 255) / 256);
                if(p->K / c < (chain > 1 ? 1024 : 512))
                    break;
                const int score = 1000 * c / chain * (chain > 1 ? 11 : 10);
                if(score > best_score)
                    best_score = score, kb = c;
            }
            if(kb > 1)
            {
                const int per = wqgemm_splitk_kread(p->K, kb, 128);
                kb            = (p->K + per - 1) / per;
                if(kb > 1)
                {
                    plan.auto_split = true;
                    plan.k_batch    = kb;
                    plan.k_per_z    = per;
                    plan.partial    = true;
                }
            }
        }
        // M <= 16 (decode): one pass over the weights on the streaming kernel.  Its workgroups own 16 columns each; fewer
        // than 256 of them leave CUs idle, so K is split across blockIdx.z (optional workspace, as above) in slices of
        // at least 1024 k up to about 512 workgroups.
        if(tiled_i8 && p->M <= 16 && p->N % 16 == 0 && p->K >= 256 && !(p->flags & (WQ_FLAG_TILE_128 | WQ_FLAG_TILE_256)) &&
           (long)(p->M - 1) * p->lda + p->K < 0x7F000000L)
        {
            plan.route      = ROUTE_I8_SKINNY;
            plan.auto_split = false, plan.partial = false, plan.k_batch = 1, plan.k_per_z = p->K;
            const int wgs = p->N / 16;
            if(wgs < 256 && p->K % 256 == 0 && p->K >= 2048)
            {
                int kb        = std::min((512 + wgs - 1) / wgs, p->K / 1024);
                const int per = wqgemm_splitk_kread(p->K, kb, 256);
                kb            = (p->K + per - 1) / per;
                if(kb > 1)
                    plan.auto_split = true, plan.k_batch = kb, plan.k_per_z = per, plan.partial = true;
            }
        }
        if((p->flags & WQ_FLAG_TILE_256) && plan.route != ROUTE_I8_MFMA256)
            return fail(WQ_ERR_UNSUPPORTED, "WQ_FLAG_TILE_256 needs N to be a multiple of 256 and tile-aligned operands");
        if((p->flags & WQ_FLAG_TILE_128) && plan.route != ROUTE_I8_MFMA128)
            return fail(WQ_ERR_UNSUPPORTED, "WQ_FLAG_TILE_128 needs N to be a multiple of 128 and tile-aligned operands");
        return WQ_OK;
    }

    // ---- fpAintB family ------------------------------------------------------------------------
    if(p->a_dtype != WQ_F16 && p->a_dtype != WQ_BF16 && p->a_dtype != WQ_F8_FNUZ && p->a_dtype != WQ_F8_E4M3FN)
        return fail(WQ_ERR_UNSUPPORTED, "unsupported A dtype %d", p->a_dtype);
    if(!is_i4(p->b_dtype) && p->b_dtype != WQ_U8_OFFSET && p->b_dtype != WQ_I8 && p->b_dtype != WQ_I8_PRESHUFFLED)
        return fail(WQ_ERR_UNSUPPORTED, "unsupported B dtype %d", p->b_dtype);
    if(is_i4(p->b_dtype) && (p->b_layout != WQ_COL_MAJOR || p->K % 8 != 0 || p->ldb % 8 != 0))
        return fail(WQ_ERR_UNSUPPORTED,
                    "packed int4 B must be column-major (K contiguous) with K and ldb multiples of 8");
    if(is_i4(p->b_dtype) && (plan.k_begin % 8 != 0 || plan.k_per_z % 8 != 0))
        return fail(WQ_ERR_UNSUPPORTED, "packed int4 K slices must start on multiples of 8");
    const bool preshuf = is_preshuf(p->b_dtype);
    if(preshuf && (p->b_layout != WQ_COL_MAJOR || p->N % 32 != 0 || p->K % 64 != 0 || p->ldb % 64 != 0 || p->ldb < p->K ||
                   plan.k_begin % 64 != 0 || plan.k_per_z % 64 != 0))
        return fail(WQ_ERR_UNSUPPORTED, "pre-shuffled B: column-major, N a multiple of 32; K, ldb and K slices multiples of 64");
    if(is16(p->a_dtype))
    {
        if(p->scale && p->scale_dtype != p->a_dtype)
            return fail(WQ_ERR_UNSUPPORTED, "scale dtype must equal the activation dtype");
    }
    if(!is16(p->a_dtype) && p->scale)
        return fail(WQ_ERR_UNSUPPORTED, "fp8 activations take their scales through the d0/d1 epilogue: scale must be NULL");
    if(p->scale_group_k < 0 || p->ld_scale < 0)
        return fail(WQ_ERR_INVALID, "scale_group_k and ld_scale must not be negative");
    const bool grouped = p->scale_group_k > 0;
    if(grouped)
    {
        if(!p->scale)
            return fail(WQ_ERR_INVALID, "group-wise scales need the scale tensor");
        if(!is16(p->a_dtype))
            return fail(WQ_ERR_UNSUPPORTED, "group-wise scales go with fp16 or bf16 activations");
        if(p->ld_scale != 0 && p->ld_scale < p->N)
            return fail(WQ_ERR_INVALID, "ld_scale too small");
    }
    // Block-wise scales of both operands (fp8 activations)
    const bool ab_scaled = has_block_scales(p);
    if(ab_scaled)
    {
        if(!p->a_block_scale || !p->b_block_scale)
            return fail(WQ_ERR_INVALID, "block-wise scales come in pairs: a_block_scale and b_block_scale");
        if(is16(p->a_dtype))
            return fail(WQ_ERR_UNSUPPORTED, "block-wise scales of both operands go with fp8 activations (16-bit activations: scale_group_k)");
        if(p->scale_block_m < 1 || p->scale_block_n < 1 || p->scale_block_k < 1)
            return fail(WQ_ERR_INVALID, "scale_block_m / _n / _k must be positive");
        const int kblocks = (p->K + p->scale_block_k - 1) / p->scale_block_k;
        if(p->ld_a_scale < 0 || p->ld_b_scale < 0 || (p->ld_a_scale != 0 && p->ld_a_scale < kblocks) ||
           (p->ld_b_scale != 0 && p->ld_b_scale < kblocks))
            return fail(WQ_ERR_INVALID, "ld_a_scale / ld_b_scale must be 0 or at least ceil(K / scale_block_k) = %d", kblocks);
        // (K slices and k_batch: every kernel indexes the scale tables with the absolute k, and the fp32 partials already
        // carry the scales, so the slabs sum like any others)
        if((uintptr_t)p->a_block_scale % 4 != 0 || (uintptr_t)p->b_block_scale % 4 != 0)
            return fail(WQ_ERR_INVALID, "block scales are fp32: 4-byte aligned pointers");
    }
    if(!slice)
    {
        // (a K-slice launch stores fp32 partials; its epilogue runs in wqgemm_reduce_finalize)
        if(!is16(p->c_dtype))
            return fail(WQ_ERR_UNSUPPORTED, "C must be fp16 or bf16");
        if(is16(p->a_dtype) && p->c_dtype != p->a_dtype)
            return fail(WQ_ERR_UNSUPPORTED, "C dtype must equal the activation dtype");
    }
    if(const int rc = check_fp_epilogue(p); rc != WQ_OK)
        return rc;

    plan.ak = p->a_dtype == WQ_BF16 ? wq::A_BF16 : wq::A_F16;
    plan.bk = is_i4(p->b_dtype) ? wq::B_I4P : (p->b_dtype == WQ_U8_OFFSET ? wq::B_U8O : wq::B_I8);

    // The tiled MFMA kernel needs whole tiles along K, N % 4 == 0 (its last row and column tiles may be partial),
    // K-contiguous operands and 16-byte aligned rows; everything else goes to the generic kernel.
    const int b_row_align = is_i4(p->b_dtype) ? 32 : 16; // ldb elements per 16 bytes
    const bool tiled = is16(p->a_dtype) && p->a_layout == WQ_ROW_MAJOR && p->b_layout == WQ_COL_MAJOR &&
                       p->N % 4 == 0 && plan.k_begin % 64 == 0 &&
                       plan.k_per_z % 64 == 0 && (plan.k_end - plan.k_begin) % 64 == 0 &&
                       p->lda % 8 == 0 && p->ldb % b_row_align == 0 && p->ldc % 4 == 0 &&
                       ((uintptr_t)p->a % 16 == 0) && ((uintptr_t)p->b % 16 == 0) &&
                       ((uintptr_t)p->c % 16 == 0) && ((uintptr_t)p->scale % 2 == 0) &&
                       (!grouped || p->scale_group_k % 64 == 0); // a K tile must lie inside one scale group
    plan.route = tiled && !(p->flags & WQ_FLAG_GENERIC) ? ROUTE_MFMA128 : ROUTE_GENERIC;
    // Small M (decode-like shapes): HBM-bound, one pass over the weights; its own kernel with split-K for occupancy.
    {
        const int ks        = is_i4(p->b_dtype) ? 128 : 64; // k per super-step of wq_gemm_skinny
        const int klen_all  = plan.k_end - plan.k_begin;
        // group-wise scales: fp16 activations, groups of a power of two that cover a lane's share of a super-step
        const bool grouped_ok = !grouped || (p->b_dtype != WQ_U8_OFFSET &&
                                             (p->scale_group_k & (p->scale_group_k - 1)) == 0 && p->scale_group_k >= ks / 4);
        const bool eligible = is16(p->a_dtype) && grouped_ok && p->M <= 64 && !(p->flags & (WQ_FLAG_GENERIC | WQ_FLAG_TILE_128 | WQ_FLAG_TILE_256)) &&
                              p->a_layout == WQ_ROW_MAJOR && p->b_layout == WQ_COL_MAJOR && p->N % 64 == 0 &&
                              klen_all % ks == 0 && plan.k_begin % ks == 0 && p->lda % 8 == 0 &&
                              p->ldb % b_row_align == 0 && p->ldc % 4 == 0 && (uintptr_t)p->a % 16 == 0 &&
                              (uintptr_t)p->b % 16 == 0 && (uintptr_t)p->c % 16 == 0 && (uintptr_t)p->scale % 2 == 0;
        // M <= 16: four waves split K inside the workgroup, one kernel, no workspace -- as long as N / 16 workgroups
        // occupy the chip and K is short enough that splitting it across the grid would not pay for a second kernel
        // (every workgroup reads all M rows of the activations: N / 16 * M * K * 2 bytes from L2 in total, four times
        // what the grid-split kernel reads; measured break-even near 96 MB)
        // fp8 activations at M <= 16 (decode): their fp16 image (exact, M * K * 2 bytes in the workspace) lets the same
        // streaming kernel read the weights once at HBM speed; the fp8 tile kernel would spend a 256-row tile on them.
        // Only with a workspace, packed int4 weights, fp16 C; the tile kernel (or, for FNUZ bytes that may hold the
        // special encodings, the image through the 128-tile kernel) otherwise.
        // (when the whole image fits the kernel's LDS budget the kernel widens the bytes itself: no workspace, one launch)
        // (every workgroup widens the whole image for itself: beyond some 8M conversions per row of activations -- 28672
        // columns x 8192 k: 32.2 against 29.3 us -- one pass into the workspace is cheaper when there is one)
        const bool f8_fused = (size_t)p->M * (klen_all * 2 + 16) <= 40 * 1024 &&
                              (!have_workspace || (long)(p->N / 16) * klen_all <= 8L * 1000 * 1000);
        const bool f8_image = !is16(p->a_dtype) && !ab_scaled && (have_workspace || f8_fused) && is_i4(p->b_dtype) && p->c_dtype == WQ_F16 &&
                              p->K % 8 == 0 && !(p->flags & WQ_FLAG_PINGPONG);
        const bool eligible_f8 = f8_image && p->M <= 16 && !(p->flags & (WQ_FLAG_GENERIC | WQ_FLAG_TILE_128 | WQ_FLAG_TILE_256)) &&
                                 p->a_layout == WQ_ROW_MAJOR && p->b_layout == WQ_COL_MAJOR && p->N % 64 == 0 && klen_all % ks == 0 &&
                                 p->ldb % b_row_align == 0 && p->ldc % 4 == 0 && (uintptr_t)p->a % 8 == 0 && p->lda % 8 == 0 &&
                                 (uintptr_t)p->b % 16 == 0 && (uintptr_t)p->c % 16 == 0;
        if(eligible_f8 && !slice && plan.k_batch == 1 && klen_all / ks >= 4 && p->N / 16 >= 128 &&
           (klen_all <= 8192 || (size_t)p->M * (klen_all * 2 + 16) <= 40 * 1024) && (long)(p->N / 16) * p->M * klen_all * 2 <= 96L * 1000 * 1000)
        {
            plan.via_f16  = !f8_fused;
            plan.f8_fused = f8_fused;
            plan.ak       = wq::A_F16;
            plan.route    = ROUTE_SKINNY_WS;
            return WQ_OK;
        }
        if(eligible && p->M <= 16 && !slice && plan.k_batch == 1 && klen_all / ks >= 4 && p->N % 16 == 0 &&
           p->N / 16 >= 128 && (klen_all <= 8192 || (size_t)p->M * (klen_all * 2 + 16) <= 40 * 1024) && // (chunked variant: short K only)
           (long)(p->N / 16) * p->M * klen_all * 2 <= 96L * 1000 * 1000)
        {
            plan.route = ROUTE_SKINNY_WS;
            return WQ_OK;
        }
        if(eligible)
        {
            int kb = plan.k_batch, per = klen_all;
            if(kb > 1)
                per = wqgemm_splitk_kread(klen_all, kb, ks);
            else if(!slice)
            {
                // Split K for occupancy (about four workgroups per CU), but not beyond the point where writing and
                // re-reading the fp32 partials (8 kb M N bytes) costs half of what streaming the weights does, and
                // in slices of whole 4-super-step groups when K allows (the kernel then keeps 4 loads in flight).
                const int gran       = klen_all % (4 * ks) == 0 ? 4 * ks : ks;
                const int wgs        = p->N / 64;
                const long wbytes_k  = is_i4(p->b_dtype) ? klen_all / 2 : klen_all; // weight bytes per column
                const int kb_occ     = (1024 + wgs - 1) / wgs;
                const long kb_bytes  = wbytes_k / (16L * p->M);
                kb                   = kb_occ;
                if(kb > kb_bytes)
                    kb = (int)kb_bytes;
                if(kb > klen_all / gran)
                    kb = klen_all / gran;
                if(kb < 1)
                    kb = 1;
                per = wqgemm_splitk_kread(klen_all, kb, gran);
                kb  = (klen_all + per - 1) / per; // no empty slices
            }
            plan.route      = ROUTE_SKINNY;
            plan.auto_split = !slice && plan.k_batch == 1 && kb > 1;
            plan.k_batch    = kb;
            plan.k_per_z    = per;
            plan.partial    = slice || kb > 1;
            return WQ_OK;
        }
    }
    // 8-bit weights stored (K, N) row-major (the B layout of client_example/30_gemm_bf16Aint8B): the matrix-core
    // kernels want K contiguous, so the launch first transposes B into the workspace (HBM-bound, 2 K N bytes);
    // without a workspace the general kernel runs.
    // Split-K by the caller (k_batch > 1) combines with it: the slabs come first in the workspace, the copy after them.
    if(plan.route == ROUTE_GENERIC && !(p->flags & WQ_FLAG_GENERIC) && is16(p->a_dtype) && !slice &&
       plan.k_per_z % 64 == 0 && (!grouped || p->scale_group_k % 64 == 0) &&
       p->a_layout == WQ_ROW_MAJOR && p->b_layout == WQ_ROW_MAJOR && !is_i4(p->b_dtype) &&
       p->N % 128 == 0 && p->K % 64 == 0 && p->lda % 8 == 0 && p->ldb % 16 == 0 && p->ldc % 4 == 0 &&
       (uintptr_t)p->a % 16 == 0 && (uintptr_t)p->b % 16 == 0 && (uintptr_t)p->c % 16 == 0 && (uintptr_t)p->scale % 2 == 0)
    {
        plan.via_colb = true;
        plan.route    = ROUTE_MFMA128;
    }
    // fp8 activations x packed int4 weights on the fp8 matrix core (scales arrive through the epilogue).
    // FNUZ bytes qualify only when the caller excludes the three encodings E4M3FN cannot express.
    const bool is_f8 = p->a_dtype == WQ_F8_FNUZ || p->a_dtype == WQ_F8_E4M3FN;
    // The fp8 kernel and the 32x32-MFMA kernel address A and B through buffer resources with 32-bit byte offsets: at most
    // 2 GiB per operand there (the bound the reference puts on every operand,
    // include/ck/tensor_operation/gpu/grid/gridwise_fpAintB_gemm_wmma.hpp:501-507); the other kernels index with 64 bits.
    const uint64_t a_span = ((uint64_t)(p->a_layout == WQ_ROW_MAJOR ? p->M : p->K) - 1) * (uint64_t)p->lda +
                            (uint64_t)(p->a_layout == WQ_ROW_MAJOR ? p->K : p->M);
    const uint64_t b_div  = is_i4(p->b_dtype) ? 2 : 1; // elements per byte
    const uint64_t b_span = preshuf                         ? (uint64_t)p->N * (uint64_t)p->ldb / b_div
                            : p->b_layout == WQ_COL_MAJOR ? (((uint64_t)p->N - 1) * (uint64_t)p->ldb + (uint64_t)p->K) / b_div
                                                          : (((uint64_t)p->K - 1) * (uint64_t)p->ldb + (uint64_t)p->N) / b_div;
    const bool fits_32bit = a_span * (is_f8 || p->a_dtype == WQ_I8 ? 1 : 2) <= 0x7FFFFFFFull && b_span <= 0x7FFFFFFFull;
    if(is_f8 && fits_32bit && !(p->flags & WQ_FLAG_GENERIC) && (p->a_dtype == WQ_F8_E4M3FN || (p->flags & WQ_FLAG_F8_NO_SPECIALS)) &&
       is_i4(p->b_dtype) && !(preshuf && (p->flags & WQ_FLAG_PINGPONG)) && !p->scale && p->a_layout == WQ_ROW_MAJOR && p->N % 256 == 0 &&
       (!ab_scaled || (p->scale_block_k == 128 && p->scale_block_n % 32 == 0 && p->K / 128 <= 640 &&
                       !(p->flags & WQ_FLAG_PINGPONG))) &&
       plan.k_begin % 128 == 0 && plan.k_per_z % 128 == 0 && (plan.k_end - plan.k_begin) % 128 == 0 &&
       p->lda % 16 == 0 && p->ldb % 32 == 0 && p->ldc % 8 == 0 && (uintptr_t)p->a % 16 == 0 &&
       (uintptr_t)p->b % 16 == 0 && (uintptr_t)p->c % 16 == 0)
    {
        plan.route = ROUTE_F8_MFMA256;
        // A grid shorter than the chip (M up to a thousand rows or so): split K across blockIdx.z when the caller has
        // not.  Needs (and is dropped without) a workspace.  The slice count minimises a cost model in microseconds
        // fitted to tools/mid_m_sweep.py --f8 (within 10 % of the best measured candidate at all 55 points, 0.5 % on
        // average): per workgroup a fixed 10 us (prologue, epilogue) plus 0.9 - 1.4 us per 128-k tile depending on how
        // much of the chip is busy, times the number of workgroups the busiest CU runs; the fp32 partials cost their
        // 8 bytes per slice and output at 8 TB/s (they mostly stay in the last-level cache) plus 3 us for the reduction.
        const int tiles = ((p->M + 255) / 256) * (p->N / 256);
        const int klen  = plan.k_end - plan.k_begin;
        if(!plan.partial && tiles < 256)
        {
            double best = 1e30;
            int kb      = 1;
            for(int c = 1; c <= 8 && (c == 1 || klen / c >= 512); ++c)
            {
                const int wgs = tiles * c, chain = (wgs + 255) / 256, kt = (klen / c + 127) / 128;
                double t = chain * (10.0 + kt * (0.9 + 0.5 * std::min(1.0, wgs / 256.0)));
                if(c > 1)
                    t += (double)c * p->M * p->N * 8.0 / 8e6 + 3.0;
                if(t < best)
                    best = t, kb = c;
            }
            if(kb > 1)
            {
                const int per = wqgemm_splitk_kread(klen, kb, 128);
                kb            = (klen + per - 1) / per;
                if(kb > 1)
                {
                    plan.auto_split = true;
                    plan.k_batch    = kb;
                    plan.k_per_z    = per;
                    plan.partial    = true;
                }
            }
        }
    }
    // FNUZ bytes without that promise: exact in every encoding through an fp16 image of A in the workspace and
    // the fp16 kernels with a unit weight scale (falls back to the general kernel when no workspace is given).
    if(p->a_dtype == WQ_F8_FNUZ && !ab_scaled && plan.route == ROUTE_GENERIC && !(p->flags & WQ_FLAG_GENERIC) &&
       !plan.partial && !p->scale && p->a_layout == WQ_ROW_MAJOR && p->b_layout == WQ_COL_MAJOR &&
       p->N % 128 == 0 && p->K % 64 == 0 && p->lda % 8 == 0 && p->ldb % b_row_align == 0 && p->ldc % 4 == 0 &&
       (uintptr_t)p->a % 8 == 0 && (uintptr_t)p->b % 16 == 0 && (uintptr_t)p->c % 16 == 0)
    {
        plan.via_f16 = true;
        plan.ak      = wq::A_F16;
        plan.route   = ROUTE_MFMA128;
    }
    // (the last row tile may be partial, as in the 128-tile kernel; the 32x32-MFMA variant keeps whole tiles)
    // (group-wise scales reach the 256-tile kernel by 4-byte LDS-DMA: 4-byte aligned scale rows, default variant only)
    const bool grouped_256_ok =
        !grouped || ((uintptr_t)p->scale % 4 == 0 && p->ld_scale % 2 == 0 &&
                     !(p->flags & (WQ_FLAG_LOCKSTEP | WQ_FLAG_WAVES_2X4 | WQ_FLAG_DEQUANT_IN_L | WQ_FLAG_MFMA_32X32)));
    if(plan.route == ROUTE_MFMA128 && grouped_256_ok && (p->M % 256 == 0 || !(p->flags & WQ_FLAG_MFMA_32X32)) && p->N % 256 == 0 &&
       p->ldc % 8 == 0 && !(p->flags & WQ_FLAG_TILE_128))
    {
        // 256 x 256 tiles, one workgroup per CU: worth it once the grid covers at least half the chip
        const long wgs = (long)((p->M + 255) / 256) * (p->N / 256) * plan.k_batch;
        if(wgs >= 128 || (p->flags & WQ_FLAG_TILE_256))
            plan.route = ROUTE_MFMA256;
        // Between 40 (32 once K reaches 8192) and 128 tiles (M of a few hundred rows against a few thousand columns) the 256-tile kernel still
        // wins once K is split across blockIdx.z up to one workgroup per CU, in slices of at least 768 k: 10 - 40 % over
        // the 128-tile kernel's best split at M = 192 ... 2048 against 4096 ... 28672 columns (tools/mid_m_sweep.py).
        // Planner's own split: needs a workspace, and without one the launch goes back to the 128-tile kernel.
        const int klen = plan.k_end - plan.k_begin;
        if(wgs >= (klen >= 8192 ? 32 : 40) && wgs <= 128 && !plan.partial && !plan.via_f16 && !plan.via_colb && p->M > 128 &&
           !(p->flags & (WQ_FLAG_TILE_256 | WQ_FLAG_MFMA_32X32)))
        {
            int kb = (int)(256 / wgs);
            if(kb > klen / 768)
                kb = klen / 768;
            if(kb > 1)
            {
                const int per = wqgemm_splitk_kread(klen, kb, 64);
                kb            = (klen + per - 1) / per;
                if(kb > 1)
                {
                    plan.route      = ROUTE_MFMA256;
                    plan.auto_split = true;
                    plan.split_256  = wgs < 128;
                    plan.k_batch    = kb;
                    plan.k_per_z    = per;
                    plan.partial    = true;
                }
            }
        }
    }
    // Few tiles (a short M against a narrow N) leave most CUs idle: split K across blockIdx.z when the caller has not,
    // up to about one workgroup per CU and at least four K tiles per slice.  Needs (and is dropped without) a workspace.
    if(plan.route == ROUTE_MFMA128 && !plan.partial && !plan.via_f16 && !plan.via_colb && !(p->flags & WQ_FLAG_TILE_128))
    {
        const int wgs  = ((p->M + 127) / 128) * ((p->N + 127) / 128);
        const int klen = plan.k_end - plan.k_begin;
        // (up to 128 workgroups: about one per CU after the split; up to two per CU, which the 128-tile kernel runs
        // side by side, halves of a long K still pay: measured 15 - 20 % at 4096 <= K)
        int kb = wgs < 512 && klen >= 4096 ? 2 : 1;
        if(wgs <= 128)
        {
            // The busiest CU runs ceil(wgs * kb / 256) workgroups of klen / kb each: pick the kb with the shortest such
            // chain, a second workgroup per CU (slices of at least 1024 k then, 256 k otherwise) counting as a tenth
            // faster than its share; ties go to the smaller kb (less partial traffic).  Workgroup counts between 256
            // and 512 that are not near 512 lose: 320 workgroups run as long as 512 of the same slice length.
            int best_score = 0;
            kb             = 1;
            for(int c = 1; c <= 16 && wgs * c <= 512; ++c)
            {
                const int chain = (wgs * c + 255) / 256;
                if(klen / c < (chain > 1 ? 1024 : 256))
                    break;
                const int score = 1000 * c / chain * (chain > 1 ? 11 : 10);
                if(score > best_score)
                    best_score = score, kb = c;
            }
        }
        if(kb > 1)
        {
            const int per = wqgemm_splitk_kread(klen, kb, 64);
            kb            = (klen + per - 1) / per;
            if(kb > 1)
            {
                plan.auto_split = true;
                plan.k_batch    = kb;
                plan.k_per_z    = per;
                plan.partial    = true;
            }
        }
    }
    // Pre-shuffled weights are read by the tile kernels and the small-M kernels (and by the general kernel when it is asked
    // for by name); there is no slow path to fall onto silently.
    if(preshuf && plan.route == ROUTE_GENERIC && !(p->flags & WQ_FLAG_GENERIC))
        return fail(WQ_ERR_UNSUPPORTED,
                    "pre-shuffled B runs on the matrix-core kernels only: K-contiguous 16-byte aligned operands, K in whole "
                    "64-k tiles (128 for fp8 activations, which also need N %% 256 == 0), scale groups of a multiple of 64 k");
    if(plan.route == ROUTE_GENERIC && (p->M + 15) / 16 > 65535)
        return fail(WQ_ERR_UNSUPPORTED, "the general kernel takes at most 1048560 rows (M = %d)", p->M);
    if((p->flags & WQ_FLAG_MFMA_32X32) && plan.route == ROUTE_MFMA256 && !fits_32bit)
        return fail(WQ_ERR_UNSUPPORTED, "the 32x32-MFMA kernel addresses at most 2 GiB of A and of B");
    if(preshuf && (p->flags & WQ_FLAG_MFMA_32X32))
        return fail(WQ_ERR_UNSUPPORTED, "pre-shuffled B: the 32x32-MFMA kernel does not read this layout");
    if((p->flags & WQ_FLAG_TILE_256) && plan.route != ROUTE_MFMA256)
        return fail(WQ_ERR_UNSUPPORTED, "WQ_FLAG_TILE_256 needs N to be a multiple of 256 and tile-aligned operands");
    if((p->flags & WQ_FLAG_TILE_128) && plan.route != ROUTE_MFMA128)
        return fail(WQ_ERR_UNSUPPORTED, "WQ_FLAG_TILE_128 needs N to be a multiple of 128 and tile-aligned operands");
    return WQ_OK;
}

void set_scale_groups(wq::GemmParams& gp, const wqgemm_problem* p)
{
    gp.scale_group_k     = p->scale_group_k > 0 ? p->scale_group_k : 0;
    gp.ld_scale          = p->ld_scale > 0 ? p->ld_scale : p->N;
    gp.scale_group_shift = -1;
    if(gp.scale_group_k > 0 && (gp.scale_group_k & (gp.scale_group_k - 1)) == 0)
        gp.scale_group_shift = __builtin_ctz((unsigned)gp.scale_group_k);
}

template <typename Params>
void set_block_scales(Params& gp, const wqgemm_problem* p)
{
    const bool on = p->a_block_scale && p->b_block_scale;
    gp.sa         = on ? (const float*)p->a_block_scale : nullptr;
    gp.sb         = on ? (const float*)p->b_block_scale : nullptr;
    gp.bm = on ? p->scale_block_m : 1, gp.bn = on ? p->scale_block_n : 1;
    const int kblocks = on ? (p->K + p->scale_block_k - 1) / p->scale_block_k : 1;
    gp.ld_sa = on && p->ld_a_scale > 0 ? p->ld_a_scale : kblocks;
    gp.ld_sb = on && p->ld_b_scale > 0 ? p->ld_b_scale : kblocks;
    if constexpr(std::is_same_v<Params, wq::GenericParams>)
        gp.bk = on ? p->scale_block_k : 1;
}

wq::EpilogueArgs epilogue_of(const wqgemm_problem* p)
{
    wq::EpilogueArgs e;
    e.kind    = p->epilogue;
    e.c_dtype = p->c_dtype;
    e.requant = p->requant_scale;
    e.d0      = p->d0;
    e.d1      = p->d1;
    e.ldd0    = p->ldd0;
    e.ldd1    = p->ldd1;
    return e;
}

template <int AK, int BKIND>
void launch_mfma128(const wq::GemmParams& gp, dim3 grid, hipStream_t s)
{
    if(gp.scale_group_k > 0)
    {
        hipLaunchKernelGGL((wq::wq_gemm_mfma_128<AK, BKIND, true>), grid, dim3(256), 0, s, gp);
        return;
    }
    hipLaunchKernelGGL((wq::wq_gemm_mfma_128<AK, BKIND>), grid, dim3(256), 0, s, gp);
}

// fp8 activations through the 16-bit kernels: workspace = [M][K] fp16 image of A, then N fp16 ones (the unit weight scale)
void f16_image_of_a(const wqgemm_problem* p, void* workspace, hipStream_t s, wq::GemmParams& gp)
{
    uint16_t* a16     = (uint16_t*)workspace;
    uint16_t* ones    = a16 + (size_t)p->M * p->K;
    const long groups = (long)p->M * (p->K / 8);
    int blocks        = (int)((groups + 255) / 256);
    if(blocks > 8192)
        blocks = 8192;
    if(blocks < 1)
        blocks = 1;
    hipLaunchKernelGGL(wq::wq_f8_fnuz_to_f16, dim3(blocks), dim3(256), 0, s, (const uint8_t*)p->a, p->lda, p->M, p->K, a16, ones,
                       p->N, p->a_dtype == WQ_F8_E4M3FN ? 1 : 0);
    gp.a     = a16;
    gp.scale = ones;
    gp.lda   = p->K;
}

int launch_main(const wqgemm_problem* p, const Plan& plan, void* workspace, hipStream_t s)
{
    void* out  = plan.partial && !plan.to_c ? workspace : p->c;
    const int ld_out = plan.partial && !plan.to_c ? p->N : p->ldc;
    if(plan.route == ROUTE_MFMA128 || plan.route == ROUTE_MFMA256 || plan.route == ROUTE_F8_MFMA256)
    {
        const int tile = plan.route == ROUTE_MFMA128 ? 128 : 256;
        wq::GemmParams gp;
        gp.a_f8      = 0;
        gp.a         = p->a;
        gp.b         = p->b;
        gp.scale     = p->scale;
        gp.lda       = p->lda;
        if(plan.via_colb)
        {
            uint8_t* colb = (uint8_t*)workspace + (plan.partial ? (size_t)plan.k_batch * p->M * p->N * sizeof(float) : 0);
            hipLaunchKernelGGL(wq::wq_transpose_b8, dim3(p->N / 64, p->K / 64), dim3(256), 0, s, (const uint8_t*)p->b, p->ldb,
                               p->K, p->N, colb);
            gp.b         = colb;
            gp.ldb_bytes = p->K;
        }
        if(plan.via_f16)
            f16_image_of_a(p, workspace, s, gp);
        gp.c         = out;
        gp.M         = p->M;
        gp.N         = p->N;
        gp.K         = p->K;
        if(!plan.via_colb)
            gp.ldb_bytes = is_i4(p->b_dtype) ? p->ldb / 2 : p->ldb;
        gp.b_preshuf = is_preshuf(p->b_dtype) ? 1 : 0;
        gp.ldc       = ld_out;
        gp.k_begin   = plan.k_begin;
        gp.k_end     = plan.k_end;
        gp.k_per_z   = plan.k_per_z;
        gp.partial   = plan.partial ? 1 : 0;
        gp.tiles_m   = (p->M + tile - 1) / tile;
        gp.tiles_n   = (p->N + tile - 1) / tile;
        gp.epi       = epilogue_of(p);
        gp.direct_store = (p->flags & WQ_FLAG_DIRECT_STORE) ? 1 : 0;
        gp.dbg          = g_dbg;
        gp.a_scale      = p->a_dtype == WQ_F8_FNUZ ? 0x7E7E7E7E : 0x7F7F7F7F;
        set_scale_groups(gp, p);
        set_block_scales(gp, p);
        {
            // Tiles are dealt to the 8 XCDs in contiguous runs of nwg/8 (xcd_remap); inside a run the
            // walk covers group_m tile rows x (run / group_m) tile columns.  Pick the group that
            // minimises the bytes one XCD pulls through its L2: g rows of A plus run/g columns of B.
            const double a_bytes = (plan.route == ROUTE_F8_MFMA256 ? 1.0 : 2.0) * tile, b_bytes = (is_i4(p->b_dtype) ? 0.5 : 1.0) * tile;
            const int run        = (gp.tiles_m * gp.tiles_n + 7) / 8;
            int best = 1;
            double best_cost = 1e30;
            for(int g = 1; g <= 16 && g <= gp.tiles_m; g *= 2)
            {
                const double cost = g * a_bytes + ((run + g - 1) / g) * b_bytes;
                if(cost <= best_cost)
                    best_cost = cost, best = g;
            }
            gp.group_m = (p->flags & WQ_FLAG_GROUP8) ? 8 : best;
        }
        const dim3 grid(gp.tiles_m * gp.tiles_n, 1, plan.k_batch);
        if(plan.route == ROUTE_F8_MFMA256)
        {
            const int rc = wq::launch_256_f8(gp, grid, s, p->flags);
            if(rc == WQ_ERR_HIP)
                return fail(WQ_ERR_HIP, "fp8 256-tile kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        }
        else if(plan.route == ROUTE_MFMA256)
        {
            const int ls = p->flags; // tuning A/B switches
            int rc;
            if((ls & WQ_FLAG_MFMA_32X32) && plan.bk == wq::B_I4P)
                rc = wq::launch_256_w32(gp, grid, s, plan.ak);
            else if(plan.ak == wq::A_F16)
                rc = plan.bk == wq::B_I4P   ? wq::launch_256_0_0(gp, grid, s, ls)
                     : plan.bk == wq::B_U8O ? wq::launch_256_0_1(gp, grid, s, ls)
                                            : wq::launch_256_0_2(gp, grid, s, ls);
            else
                rc = plan.bk == wq::B_I4P   ? wq::launch_256_1_0(gp, grid, s, ls)
                     : plan.bk == wq::B_U8O ? wq::launch_256_1_1(gp, grid, s, ls)
                                            : wq::launch_256_1_2(gp, grid, s, ls);
            if(rc == WQ_ERR_HIP)
                return fail(WQ_ERR_HIP, "256-tile kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
            if(rc != WQ_OK)
                return rc;
        }
        else if(plan.ak == wq::A_F16)
        {
            if(plan.bk == wq::B_I4P)
                launch_mfma128<wq::A_F16, wq::B_I4P>(gp, grid, s);
            else if(plan.bk == wq::B_U8O)
                launch_mfma128<wq::A_F16, wq::B_U8O>(gp, grid, s);
            else
                launch_mfma128<wq::A_F16, wq::B_I8>(gp, grid, s);
        }
        else
        {
            if(plan.bk == wq::B_I4P)
                launch_mfma128<wq::A_BF16, wq::B_I4P>(gp, grid, s);
            else if(plan.bk == wq::B_U8O)
                launch_mfma128<wq::A_BF16, wq::B_U8O>(gp, grid, s);
            else
                launch_mfma128<wq::A_BF16, wq::B_I8>(gp, grid, s);
        }
    }
    else if(plan.route == ROUTE_SKINNY)
    {
        wq::GemmParams gp{};
        gp.a = p->a, gp.b = p->b, gp.scale = p->scale, gp.c = out;
        gp.M = p->M, gp.N = p->N, gp.K = p->K;
        gp.lda = p->lda, gp.ldb_bytes = is_i4(p->b_dtype) ? p->ldb / 2 : p->ldb, gp.ldc = ld_out;
        gp.k_begin = plan.k_begin, gp.k_end = plan.k_end, gp.k_per_z = plan.k_per_z;
        gp.partial = plan.partial ? 1 : 0;
        gp.epi     = epilogue_of(p);
        gp.b_preshuf = is_preshuf(p->b_dtype) ? 1 : 0;
        set_scale_groups(gp, p);
        const int nmb       = p->M <= 16 ? 1 : (p->M <= 32 ? 2 : 4);
        const int ks        = is_i4(p->b_dtype) ? 128 : 64;
        // 4 loads in flight when every slice is a whole number of 4-super-step groups
        const int klen      = plan.k_end - plan.k_begin;
        const bool by4      = plan.k_per_z % (4 * ks) == 0 && klen % (4 * ks) == 0;
        const dim3 grid(p->N / 64, 1, plan.k_batch);
        const int rc = wq::launch_skinny(plan.ak, plan.bk, gp, grid, s, nmb, by4 ? 4 : 1);
        if(rc != WQ_OK)
            return fail(WQ_ERR_HIP, "small-M kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    else if(plan.route == ROUTE_SKINNY_WS)
    {
        wq::GemmParams gp{};
        gp.a = p->a, gp.b = p->b, gp.scale = p->scale, gp.c = p->c;
        gp.M = p->M, gp.N = p->N, gp.K = p->K;
        gp.lda = p->lda, gp.ldb_bytes = is_i4(p->b_dtype) ? p->ldb / 2 : p->ldb, gp.ldc = p->ldc;
        gp.k_begin = plan.k_begin, gp.k_end = plan.k_end;
        gp.epi     = epilogue_of(p);
        gp.b_preshuf = is_preshuf(p->b_dtype) ? 1 : 0;
        set_scale_groups(gp, p);
        if(plan.via_f16)
            f16_image_of_a(p, workspace, s, gp);
        if(plan.f8_fused)
            gp.a_f8 = p->a_dtype == WQ_F8_E4M3FN ? 2 : 1;
        const int ks = is_i4(p->b_dtype) ? 128 : 64, klen = plan.k_end - plan.k_begin;
        // loads in flight = super-steps per unit of K dealt to a wave: as many as leave every wave at least one unit
        const int steps   = klen / ks;
        const int p_loads = steps % 4 == 0 && steps / 4 >= 4 ? 4 : (steps % 2 == 0 && steps / 2 >= 4 ? 2 : 1);
        // the whole activation image in LDS when it is small (no synchronisation in the loop), else chunk by chunk
        int lds_bytes = p->M * (klen * 2 + 16);
        if(lds_bytes > 40 * 1024)
            lds_bytes = 0;
        else if(lds_bytes < 4096)
            lds_bytes = 4096; // the cross-wave sum needs 4 x 64 x 16 bytes
        const int rc = wq::launch_skinny_ws(plan.ak, plan.bk, gp, dim3(p->N / 16), lds_bytes, s,
                                            p_loads);
        if(rc != WQ_OK)
            return fail(WQ_ERR_HIP, "small-M kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    else if(plan.route == ROUTE_I8_MFMA256)
    {
        wq::I8GemmParams gp;
        gp.a = p->a, gp.b = p->b, gp.c = p->c;
        gp.M = p->M, gp.N = p->N, gp.K = p->K;
        gp.lda = p->lda, gp.ldb = p->ldb, gp.ldc = p->ldc;
        gp.tiles_m = (p->M + 255) / 256, gp.tiles_n = p->N / 256;
        gp.group_m = gp.tiles_m >= 8 ? 8 : gp.tiles_m;
        gp.requant = p->requant_scale;
        gp.kind    = p->epilogue;
        gp.bias    = (const int32_t*)p->d0;
        gp.partial = nullptr, gp.k_tiles_per_z = 0;
        static std::atomic<bool> opted_in[64]; // more than 64 KiB of dynamic LDS must be opted into once per device
        int dev                  = 0;
        WQ_HIP(hipGetDevice(&dev));
        if(dev >= 0 && dev < 64 && !opted_in[dev])
        {
            WQ_HIP(hipFuncSetAttribute((const void*)wq::wq_gemm_i8_mfma_256, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       wq::kI8Tile256Lds));
            opted_in[dev] = true;
        }
        hipLaunchKernelGGL(wq::wq_gemm_i8_mfma_256, dim3(gp.tiles_m * gp.tiles_n), dim3(512), wq::kI8Tile256Lds, s, gp);
    }
    else if(plan.route == ROUTE_I8_SKINNY)
    {
        wq::I8GemmParams gp;
        gp.a = p->a, gp.b = p->b, gp.c = p->c;
        gp.M = p->M, gp.N = p->N, gp.K = p->K;
        gp.lda = p->lda, gp.ldb = p->ldb, gp.ldc = p->ldc;
        gp.tiles_m = 1, gp.tiles_n = p->N / 16, gp.group_m = 1;
        gp.requant = p->requant_scale;
        gp.kind    = p->epilogue;
        gp.bias    = (const int32_t*)p->d0;
        gp.partial = plan.partial ? (int32_t*)workspace : nullptr;
        gp.k_tiles_per_z = plan.k_per_z / 128;
        // NB blocks of 16 columns per workgroup share each activation fetch (two, four past 8 rows: the activation reads
        // weigh M / 16 of the weight reads) as long as that leaves a workgroup per CU; about eight loads in flight per lane,
        // and every slice a multiple of 64 P (the last slice is K minus whole slices)
        const int per = plan.k_per_z;
        int nb        = p->M <= 8 ? 2 : 4;
        while(nb > 1 && (p->N % (16 * nb) != 0 || p->N / (16 * nb) * (plan.partial ? plan.k_batch : 1) < 256))
            nb >>= 1;
        const bool k512 = per % 512 == 0 && p->K % 512 == 0 && per >= 2048, k256 = per % 256 == 0 && p->K % 256 == 0 && per >= 1024;
        const dim3 grid(p->N / (16 * nb), 1, plan.partial ? plan.k_batch : 1);
#define WQ_I8_SKINNY(P_, NB_) hipLaunchKernelGGL((wq::wq_gemm_i8_skinny<P_, NB_>), grid, dim3(256), 0, s, gp)
        if(nb == 4)
        {
            if(k256) WQ_I8_SKINNY(2, 4); else WQ_I8_SKINNY(1, 4);
        }
        else if(nb == 2)
        {
            if(k256) WQ_I8_SKINNY(4, 2); else WQ_I8_SKINNY(1, 2);
        }
        else
        {
            if(k512) WQ_I8_SKINNY(8, 1); else if(k256) WQ_I8_SKINNY(4, 1); else WQ_I8_SKINNY(1, 1);
        }
#undef WQ_I8_SKINNY
    }
    else if(plan.route == ROUTE_I8_MFMA128)
    {
        wq::I8GemmParams gp;
        gp.a = p->a, gp.b = p->b, gp.c = p->c;
        gp.M = p->M, gp.N = p->N, gp.K = p->K;
        gp.lda = p->lda, gp.ldb = p->ldb, gp.ldc = p->ldc;
        gp.tiles_m = (p->M + 127) / 128, gp.tiles_n = (p->N + 127) / 128;
        gp.group_m = gp.tiles_m >= 8 ? 8 : gp.tiles_m;
        gp.requant = p->requant_scale;
        gp.kind    = p->epilogue;
        gp.bias    = (const int32_t*)p->d0;
        gp.partial = plan.partial ? (int32_t*)workspace : nullptr;
        gp.k_tiles_per_z = plan.k_per_z / 128;
        hipLaunchKernelGGL(wq::wq_gemm_i8_mfma_128, dim3(gp.tiles_m * gp.tiles_n, 1, plan.partial ? plan.k_batch : 1), dim3(256), 0,
                           s, gp);
    }
    else
    {
        wq::GenericParams gp;
        gp.a     = p->a;
        gp.b     = p->b;
        gp.scale = p->scale;
        gp.c     = out;
        gp.M     = p->M;
        gp.N     = p->N;
        gp.K     = p->K;
        gp.a_sm  = p->a_layout == WQ_ROW_MAJOR ? p->lda : 1;
        gp.a_sk  = p->a_layout == WQ_ROW_MAJOR ? 1 : p->lda;
        if(is_i4(p->b_dtype))
        {
            gp.b_sk = 0;
            gp.b_sn = p->ldb / 2;
        }
        else
        {
            gp.b_sk = p->b_layout == WQ_COL_MAJOR ? 1 : p->ldb;
            gp.b_sn = p->b_layout == WQ_COL_MAJOR ? p->ldb : 1;
        }
        gp.ldc     = ld_out;
        gp.k_begin = plan.k_begin;
        gp.k_end   = plan.k_end;
        gp.k_per_z = plan.k_per_z;
        gp.partial = plan.partial ? 1 : 0;
        gp.a_dtype = p->a_dtype;
        gp.b_dtype = p->b_dtype;
        gp.epi     = epilogue_of(p);
        gp.scale_group_k = p->scale_group_k > 0 ? p->scale_group_k : 0;
        gp.ld_scale      = p->ld_scale > 0 ? p->ld_scale : p->N;
        set_block_scales(gp, p);
        const dim3 grid((p->N + 15) / 16, (p->M + 15) / 16, plan.k_batch);
        hipLaunchKernelGGL(wq::wq_gemm_generic, grid, dim3(256), 0, s, gp);
    }
    WQ_HIP(hipGetLastError());
    return WQ_OK;
}

int launch_reduce(const wqgemm_problem* p, const float* partials, int k_batch, hipStream_t s)
{
    const long work = (long)p->M * (p->N / 4);
    int blocks      = (int)((work + 255) / 256);
    if(blocks > 4096)
        blocks = 4096;
    if(p->a_dtype == WQ_I8)
    {
        hipLaunchKernelGGL(wq::wq_reduce_requant_i8, dim3(blocks), dim3(256), 0, s, (const int32_t*)partials, k_batch, p->M, p->N,
                           (int8_t*)p->c, p->ldc, p->epilogue, p->requant_scale, (const int32_t*)p->d0);
        WQ_HIP(hipGetLastError());
        return WQ_OK;
    }
    hipLaunchKernelGGL(wq::wq_reduce_finalize, dim3(blocks), dim3(256), 0, s, partials, k_batch,
                       p->M, p->N, p->N, p->c, p->ldc, epilogue_of(p));
    WQ_HIP(hipGetLastError());
    return WQ_OK;
}

const char* dt_name(int dt)
{
    switch(dt)
    {
    case WQ_F16: return "F16";
    case WQ_BF16: return "BF16";
    case WQ_F8_FNUZ: return "F8";
    case WQ_F8_E4M3FN: return "F8_E4M3FN";
    case WQ_F32: return "F32";
    case WQ_I8: return "I8";
    case WQ_U8_OFFSET: return "U8";
    case WQ_I4_PACKED: return "I4";
    case WQ_I4_PRESHUFFLED: return "I4_PRESHUFFLED";
    case WQ_I8_PRESHUFFLED: return "I8_PRESHUFFLED";
    case WQ_I32: return "I32";
    default: return "?";
    }
}

} // namespace

// ---- multi-GPU split-K over RCCL -------------------------------------------------------------------------------
// RCCL is bound at first use from the copy the process already holds (a PyTorch process ships its own), else from
// the system one; libwqgemm.so itself carries no link-time dependency on it.
namespace {
struct RcclApi
{
    int (*reduce_scatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t)          = nullptr;
    const char* (*error_string)(int)                                                = nullptr;
    bool ok = false;
};
// Built once, on first use, by whichever thread gets there first (function-local static: the C++ runtime serialises
// the initialisation and publishes the finished table).
const RcclApi& rccl()
{
    static const RcclApi api = [] {
        RcclApi t;
        void* h = nullptr;
        for(const char* name : {"librccl.so.1", "librccl.so"})
            if((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)))
                break;
        if(!h)
            for(const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                    break;
        if(!h)
            return t;
        t.reduce_scatter = (decltype(t.reduce_scatter))dlsym(h, "ncclReduceScatter");
        t.all_gather     = (decltype(t.all_gather))dlsym(h, "ncclAllGather");
        t.error_string   = (decltype(t.error_string))dlsym(h, "ncclGetErrorString");
        t.ok             = t.reduce_scatter && t.all_gather;
        return t;
    }();
    return api;
}
constexpr int kNcclInt8 = 0, kNcclFloat32 = 7, kNcclSum = 0; // rccl.h: ncclDataType_t / ncclRedOp_t

size_t out_elt_bytes(int dt) { return dt == WQ_F32 ? 4 : 2; }
} // namespace

extern "C" {

const char* wqgemm_version(void) { return "wqgemm 0.1 (gfx950)"; }

size_t wqgemm_problem_size(void) { return sizeof(wqgemm_problem); }

const char* wqgemm_last_error(void) { return g_err; }

int32_t wqgemm_splitk_kread(int32_t K, int32_t k_batch, int32_t granule)
{
    if(k_batch < 1)
        k_batch = 1;
    if(granule < 1)
        granule = 1;
    const long per = (long)k_batch * granule;
    return (int32_t)(((long)K + per - 1) / per * granule);
}

int wqgemm_is_supported(const wqgemm_problem* p)
{
    Plan plan;
    return make_plan(p, plan) == WQ_OK ? 1 : 0;
}

static size_t plan_workspace_bytes(const wqgemm_problem* p, const Plan& plan)
{
    if(plan.partial && !plan.to_c)
        return (size_t)plan.k_batch * p->M * p->N * sizeof(float) + (plan.via_colb ? (size_t)p->K * p->N : 0);
    if(plan.via_f16)
        return ((size_t)p->M * p->K + p->N) * sizeof(uint16_t);
    if(plan.via_colb)
        return (size_t)p->K * p->N;
    return 0;
}

size_t wqgemm_workspace_bytes(const wqgemm_problem* p)
{
    Plan plan;
    if(make_plan(p, plan) != WQ_OK)
        return 0;
    return plan_workspace_bytes(p, plan);
}

// Run a plan made for `workspace != nullptr` (or without one: every optional use of it is dropped first).
static int launch_planned(const wqgemm_problem* p, Plan plan, void* workspace, hipStream_t s)
{
    if(plan.auto_split && !workspace)
    {
        // the planner's own split-K is an optimisation: without a workspace the kernel runs the whole K range as
        // one slice
        plan.auto_split = false;
        plan.k_batch    = 1;
        plan.partial    = false;
        plan.k_per_z    = plan.k_end - plan.k_begin;
        if(plan.split_256) // too few 256-tiles for one slice each: the 128-tile kernel's grid is four times as long
            plan.route = ROUTE_MFMA128;
    }
    if(plan.partial && !plan.to_c && !workspace)
        return fail(WQ_ERR_WORKSPACE, "k_batch = %d needs a workspace of %zu bytes", plan.k_batch,
                    (size_t)plan.k_batch * p->M * p->N * sizeof(float));
    if((plan.via_f16 || plan.via_colb) && !workspace)
    {
        plan.via_f16 = plan.via_colb = false; // no room for the re-laid-out operand: the general kernel takes it as it is
        plan.route                   = ROUTE_GENERIC;
        if((p->M + 15) / 16 > 65535)
            return fail(WQ_ERR_UNSUPPORTED, "the general kernel takes at most 1048560 rows (M = %d); with a workspace of %zu bytes "
                        "the matrix-core kernels run this problem", p->M, wqgemm_workspace_bytes(p));
    }
    const int rm = launch_main(p, plan, workspace, s);
    if(rm != WQ_OK)
        return rm;
    if(plan.partial && !plan.to_c)
        return launch_reduce(p, (const float*)workspace, plan.k_batch, s);
    return WQ_OK;
}

int wqgemm_launch(const wqgemm_problem* p, void* workspace, void* stream)
{
    Plan plan;
    const int rc = make_plan(p, plan, workspace != nullptr);
    if(rc != WQ_OK)
        return rc;
    return launch_planned(p, plan, workspace, (hipStream_t)stream);
}

// Device address of this GPU's copy of `scratch`, looked up once per device with that device current.
static int noalloc_scratch(void** out)
{
    static std::atomic<void*> per_device[64];
    int dev = 0;
    WQ_HIP(hipGetDevice(&dev));
    if(dev < 0 || dev >= 64)
        return fail(WQ_ERR_HIP, "device ordinal %d outside the table of 64", dev);
    void* ptr = per_device[dev].load(std::memory_order_acquire);
    if(!ptr)
    {
        // (two threads racing here both get the same answer for the same device)
        WQ_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if(!ptr)
            return fail(WQ_ERR_HIP, "hipGetSymbolAddress returned a null scratch address");
        per_device[dev].store(ptr, std::memory_order_release);
    }
    *out = ptr;
    return WQ_OK;
}

int wqgemm_launch_noalloc(const wqgemm_problem* p, void* stream)
{
    Plan plan;
    if(const int rc = make_plan(p, plan, true); rc != WQ_OK)
        return rc;
    const size_t need = plan_workspace_bytes(p, plan);
    if(need > WQGEMM_NOALLOC_SCRATCH_BYTES)
    {
        if(plan.partial && !plan.auto_split)
            return fail(WQ_ERR_WORKSPACE, "launch_noalloc: the problem needs %zu bytes of scratch, the built-in scratch holds %u",
                        need, WQGEMM_NOALLOC_SCRATCH_BYTES);
        // an optional scratch use (planner split-K, operand re-layout) that does not fit is dropped, as with no workspace
        return wqgemm_launch(p, nullptr, stream);
    }
    void* ws = nullptr;
    if(need != 0)
        if(const int rc = noalloc_scratch(&ws); rc != WQ_OK)
            return rc;
    if(!ws) // (nothing to hold: the plan for "no workspace" can differ from the one made above)
        return wqgemm_launch(p, nullptr, stream);
    return launch_planned(p, plan, ws, (hipStream_t)stream);
}

int wqgemm_run_timed(const wqgemm_problem* p, void* workspace, void* stream, int cold_iters,
                     int nrepeat, float* avg_ms)
{
    if(!avg_ms || nrepeat <= 0)
        return fail(WQ_ERR_INVALID, "avg_ms must be non-null and nrepeat positive");
    hipStream_t s = (hipStream_t)stream;
    for(int i = 0; i < cold_iters; ++i)
    {
        const int rc = wqgemm_launch(p, workspace, stream);
        if(rc != WQ_OK)
            return rc;
    }
    hipEvent_t e0, e1;
    WQ_HIP(hipEventCreate(&e0));
    WQ_HIP(hipEventCreate(&e1));
    WQ_HIP(hipEventRecord(e0, s));
    for(int i = 0; i < nrepeat; ++i)
    {
        const int rc = wqgemm_launch(p, workspace, stream);
        if(rc != WQ_OK)
        {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return rc;
        }
    }
    WQ_HIP(hipEventRecord(e1, s));
    WQ_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    WQ_HIP(hipEventElapsedTime(&ms, e0, e1));
    WQ_HIP(hipEventDestroy(e0));
    WQ_HIP(hipEventDestroy(e1));
    *avg_ms = ms / (float)nrepeat;
    return WQ_OK;
}

// Bytes an operand occupies as the kernels address it (last row/column ends at its last element).
static size_t a_span_bytes(const wqgemm_problem* p)
{
    const size_t sz   = (p->a_dtype == WQ_F16 || p->a_dtype == WQ_BF16) ? 2 : 1;
    const size_t rows = p->a_layout == WQ_ROW_MAJOR ? p->M : p->K, cols = p->a_layout == WQ_ROW_MAJOR ? p->K : p->M;
    return ((rows - 1) * (size_t)p->lda + cols) * sz;
}
static size_t b_span_bytes(const wqgemm_problem* p)
{
    const size_t rows = p->b_layout == WQ_COL_MAJOR ? p->N : p->K, cols = p->b_layout == WQ_COL_MAJOR ? p->K : p->N;
    const size_t elems = (rows - 1) * (size_t)p->ldb + cols;
    return is_i4(p->b_dtype) ? (elems + 1) / 2 : elems;
}

int wqgemm_run_timed_rotating(const wqgemm_problem* p, void* workspace, void* stream, int cold_iters, int nrepeat,
                              int rotating_count, float* median_ms)
{
    if(!median_ms || nrepeat <= 0)
        return fail(WQ_ERR_INVALID, "median_ms must be non-null and nrepeat positive");
    Plan plan;
    if(const int rc = make_plan(p, plan); rc != WQ_OK)
        return rc;
    if(rotating_count < 1)
        rotating_count = 1;
    if(rotating_count > 64)
        return fail(WQ_ERR_INVALID, "rotating_count %d: at most 64 copies", rotating_count);
    hipStream_t s = (hipStream_t)stream;
    const size_t na = a_span_bytes(p), nb = b_span_bytes(p);
    const void* a_copies[64];
    const void* b_copies[64];
    a_copies[0] = p->a, b_copies[0] = p->b;
    int made = 1, status = WQ_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float times[4096];
    if(nrepeat > 4096)
        nrepeat = 4096;
#define WQ_TRY(expr)                                                                      \
    if(status == WQ_OK)                                                                   \
    {                                                                                     \
        hipError_t e_ = (expr);                                                           \
        if(e_ != hipSuccess)                                                              \
            status = fail(WQ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    }
    for(; made < rotating_count && status == WQ_OK; ++made)
    {
        void *pa = nullptr, *pb = nullptr;
        WQ_TRY(hipMalloc(&pa, na));
        WQ_TRY(hipMalloc(&pb, nb));
        a_copies[made] = pa, b_copies[made] = pb;
        WQ_TRY(hipMemcpy(pa, p->a, na, hipMemcpyDeviceToDevice));
        WQ_TRY(hipMemcpy(pb, p->b, nb, hipMemcpyDeviceToDevice));
        if(status != WQ_OK)
        {
            (void)hipFree(pa);
            (void)hipFree(pb);
            break;
        }
    }
    wqgemm_problem q = *p;
    for(int i = 0; i < cold_iters && status == WQ_OK; ++i)
        status = wqgemm_launch(&q, workspace, stream);
    WQ_TRY(hipEventCreate(&e0));
    WQ_TRY(hipEventCreate(&e1));
    for(int i = 0; i < nrepeat && status == WQ_OK; ++i)
    {
        // untimed preprocessing of the iteration: step to the next copy of A and B
        q.a = a_copies[i % made], q.b = b_copies[i % made];
        WQ_TRY(hipDeviceSynchronize());
        WQ_TRY(hipEventRecord(e0, s));
        if(status == WQ_OK)
            status = wqgemm_launch(&q, workspace, stream);
        WQ_TRY(hipEventRecord(e1, s));
        WQ_TRY(hipEventSynchronize(e1));
        WQ_TRY(hipEventElapsedTime(&times[i], e0, e1));
    }
#undef WQ_TRY
    if(e0)
        (void)hipEventDestroy(e0);
    if(e1)
        (void)hipEventDestroy(e1);
    for(int i = 1; i < made; ++i)
    {
        (void)hipFree(const_cast<void*>(a_copies[i]));
        (void)hipFree(const_cast<void*>(b_copies[i]));
    }
    if(status != WQ_OK)
        return status;
    std::sort(times, times + nrepeat);
    *median_ms = nrepeat % 2 == 1 ? times[(nrepeat - 1) / 2] : 0.5f * (times[(nrepeat - 1) / 2] + times[(nrepeat - 1) / 2 + 1]);
    return WQ_OK;
}

int wqgemm_type_string(const wqgemm_problem* p, char* buf, size_t n)
{
    if(!buf || n == 0)
        return fail(WQ_ERR_INVALID, "null buffer");
    Plan plan;
    if(make_plan(p, plan) != WQ_OK)
    {
        snprintf(buf, n, "DeviceWqGemm_gfx950<unsupported>");
        return WQ_ERR_UNSUPPORTED;
    }
    if(plan.route == ROUTE_I8_MFMA128)
        snprintf(buf, n, "DeviceGemmQuantization_Mfma_gfx950<256, 128, 128, 128, 16x16x64, I8, I8, I8, LdsDma>");
    else if(plan.route == ROUTE_I8_SKINNY)
        snprintf(buf, n, "DeviceGemmQuantization_SmallM_gfx950<256, 16, 16, 16x16x64, I8, I8, I8, StreamB, WavesSplitK, KBatch%d>", plan.k_batch);
    else if(plan.route == ROUTE_I8_MFMA256)
        snprintf(buf, n, "DeviceGemmQuantization_Mfma_gfx950<512, 256, 256, 128, 16x16x64, I8, I8, I8, LdsDma>");
    else if(plan.route == ROUTE_F8_MFMA256)
        snprintf(buf, n,
                 "DeviceGemmMultiplyMultiply_WeightQuant_Mfma_gfx950<512, 256, 256, 128, 16x16x128_f8f6f4, %s, %s, %s, "
                 "LdsDmaA, DirectB, ConvertInRegister, KBatch%d%s>",
                 dt_name(p->a_dtype), dt_name(p->b_dtype), dt_name(p->c_dtype), plan.k_batch,
                 !has_block_scales(p) ? "" : p->scale_block_m % 128 == 0 ? ", ABScalePer128K" : ", ABScalePer128K_RowScaleA");
    else if(plan.route == ROUTE_SKINNY_WS)
        snprintf(buf, n, "DeviceFpAintBGemm_SmallM_gfx950<256, 16, 16, 16x16x32, %s, %s, %s, StreamB, WavesSplitK>",
                 dt_name(p->a_dtype), dt_name(p->b_dtype), dt_name(p->c_dtype));
    else if(plan.route == ROUTE_SKINNY)
        snprintf(buf, n, "DeviceFpAintBGemm_SmallM_gfx950<256, 64, 64, 16x16x32, %s, %s, %s, StreamB, KBatch%d>",
                 dt_name(p->a_dtype), dt_name(p->b_dtype), dt_name(p->c_dtype), plan.k_batch);
    else if(plan.route == ROUTE_MFMA128 || plan.route == ROUTE_MFMA256)
    {
        char grp[32] = "";
        if(p->scale_group_k > 0)
            snprintf(grp, sizeof(grp), ", ScaleGroupK%d", p->scale_group_k);
        snprintf(buf, n,
                 "DeviceFpAintBGemm_Mfma_gfx950<%s, 64, 16x16x32, %s, %s, %s, LdsDma, "
                 "DequantInRegister, KBatch%d%s>",
                 plan.route == ROUTE_MFMA256 ? "512, 256, 256" : "256, 128, 128", dt_name(p->a_dtype),
                 dt_name(p->b_dtype), dt_name(p->c_dtype), plan.k_batch, grp);
    }
    else
        snprintf(buf, n, "DeviceWqGemm_Generic_gfx950<256, 16, 16, %s, %s, %s, KBatch%d>",
                 dt_name(p->a_dtype), dt_name(p->b_dtype), dt_name(p->c_dtype), plan.k_batch);
    return WQ_OK;
}

int wqgemm_kernel_name(const wqgemm_problem* p, char* buf, size_t n)
{
    if(!buf || n == 0)
        return fail(WQ_ERR_INVALID, "null buffer");
    Plan plan;
    const int rc = make_plan(p, plan);
    if(rc != WQ_OK)
        return rc;
    snprintf(buf, n, "%s%s", plan.via_f16 ? "wq_f8_fnuz_to_f16+" : (plan.via_colb ? "wq_transpose_b8+" : ""),
             plan.route == ROUTE_MFMA256      ? "wq_gemm_mfma_256"
             : plan.route == ROUTE_I8_MFMA128 ? "wq_gemm_i8_mfma_128"
             : plan.route == ROUTE_I8_SKINNY  ? "wq_gemm_i8_skinny"
             : plan.route == ROUTE_I8_MFMA256 ? "wq_gemm_i8_mfma_256"
             : plan.route == ROUTE_F8_MFMA256 ? (!has_block_scales(p) ? "wq_gemm_f8_mfma_256" : p->scale_block_m % 128 == 0 ? "wq_gemm_f8_mfma_256_abscale" : "wq_gemm_f8_mfma_256_abscale_rows")
             : plan.route == ROUTE_SKINNY ? "wq_gemm_skinny"
             : plan.route == ROUTE_SKINNY_WS
                 ? ((size_t)p->M * ((plan.k_end - plan.k_begin) * 2 + 16) <= 40 * 1024 ? "wq_gemm_skinny_ws" : "wq_gemm_skinny_wsc")
             : plan.route == ROUTE_MFMA128 ? "wq_gemm_mfma_128"
                                           : "wq_gemm_generic");
    return WQ_OK;
}

int wqgemm_pack_int4(const int8_t* q, int32_t ldq, int32_t K, int32_t N, uint8_t* out,
                     int32_t ld_bytes, void* stream)
{
    if(!q || !out)
        return fail(WQ_ERR_INVALID, "null pointer");
    if(K <= 0 || N <= 0 || K % 8 != 0 || ldq < K || ld_bytes < K / 2 || ld_bytes % 4 != 0)
        return fail(WQ_ERR_INVALID, "pack_int4: need K %% 8 == 0, ldq >= K, ld_bytes >= K/2, ld_bytes %% 4 == 0");
    const long work = (long)N * (K / 8);
    int blocks      = (int)((work + 255) / 256);
    if(blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(wq::wq_pack_int4, dim3(blocks), dim3(256), 0, (hipStream_t)stream, q, ldq, K,
                       N, out, ld_bytes);
    WQ_HIP(hipGetLastError());
    return WQ_OK;
}

int wqgemm_preshuffle_int4(const uint8_t* packed, int32_t ld_bytes, int32_t K, int32_t N, uint8_t* out, int32_t ldk,
                           void* stream)
{
    if(!packed || !out)
        return fail(WQ_ERR_INVALID, "null pointer");
    if(K <= 0 || N <= 0 || K % 64 != 0 || N % 32 != 0 || ld_bytes < K / 2 || ld_bytes % 16 != 0 || ldk < K || ldk % 64 != 0 ||
       (uintptr_t)packed % 16 != 0 || (uintptr_t)out % 16 != 0)
        return fail(WQ_ERR_INVALID, "preshuffle_int4: need K %% 64 == 0, N %% 32 == 0, 16-byte aligned columns (ld_bytes %% 16 == 0, "
                                    "ld_bytes >= K/2), ldk >= K, ldk %% 64 == 0");
    const long work = (long)N * (K / 32);
    int blocks      = (int)((work + 255) / 256);
    if(blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(wq::wq_preshuffle_int4, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, ld_bytes, K, N, out,
                       ldk / 2);
    WQ_HIP(hipGetLastError());
    return WQ_OK;
}

int wqgemm_preshuffle_int8(const int8_t* b, int32_t ldb, int32_t K, int32_t N, int8_t* out, int32_t ldk, void* stream)
{
    if(!b || !out)
        return fail(WQ_ERR_INVALID, "null pointer");
    if(K <= 0 || N <= 0 || K % 64 != 0 || N % 16 != 0 || ldb < K || ldb % 16 != 0 || ldk < K || ldk % 64 != 0 ||
       (uintptr_t)b % 16 != 0 || (uintptr_t)out % 16 != 0)
        return fail(WQ_ERR_INVALID, "preshuffle_int8: need K %% 64 == 0, N %% 16 == 0, 16-byte aligned columns (ldb %% 16 == 0, "
                                    "ldb >= K), ldk >= K, ldk %% 64 == 0");
    const long work = (long)N * (K / 16);
    int blocks      = (int)((work + 255) / 256);
    if(blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(wq::wq_preshuffle_int8, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)b, ldb, K, N,
                       (uint8_t*)out, ldk);
    WQ_HIP(hipGetLastError());
    return WQ_OK;
}

int wqgemm_preshuffle_int8_host(const int8_t* b, int32_t ldb, int32_t K, int32_t N, int8_t* out, int32_t ldk)
{
    if(!b || !out)
        return fail(WQ_ERR_INVALID, "null pointer");
    if(K <= 0 || N <= 0 || K % 64 != 0 || N % 16 != 0 || ldb < K || ldk < K || ldk % 64 != 0)
        return fail(WQ_ERR_INVALID, "preshuffle_int8_host: need K %% 64 == 0, N %% 16 == 0, ldb >= K, ldk >= K, ldk %% 64 == 0");
    for(int n = 0; n < N; ++n)
        for(int k16 = 0; k16 < K / 16; ++k16)
            memcpy(out + wq::preshuf8_unit_offset(n, k16, ldk), b + (size_t)n * ldb + (size_t)k16 * 16, 16);
    return WQ_OK;
}

int wqgemm_preshuffle_int4_host(const uint8_t* packed, int32_t ld_bytes, int32_t K, int32_t N, uint8_t* out, int32_t ldk)
{
    if(!packed || !out)
        return fail(WQ_ERR_INVALID, "null pointer");
    if(K <= 0 || N <= 0 || K % 64 != 0 || N % 32 != 0 || ld_bytes < K / 2 || ldk < K || ldk % 64 != 0)
        return fail(WQ_ERR_INVALID, "preshuffle_int4_host: need K %% 64 == 0, N %% 32 == 0, ld_bytes >= K/2, ldk >= K, ldk %% 64 == 0");
    for(int n = 0; n < N; ++n)
        for(int k32 = 0; k32 < K / 32; ++k32)
            memcpy(out + wq::preshuf_unit_offset(n, k32, ldk / 2), packed + (size_t)n * ld_bytes + (size_t)k32 * 16, 16);
    return WQ_OK;
}

int wqgemm_pack_int4_host(const int8_t* q, int32_t ldq, int32_t K, int32_t N, uint8_t* out,
                          int32_t ld_bytes)
{
    if(!q || !out)
        return fail(WQ_ERR_INVALID, "null pointer");
    if(K <= 0 || N <= 0 || K % 8 != 0 || ldq < K || ld_bytes < K / 2)
        return fail(WQ_ERR_INVALID, "pack_int4_host: need K %% 8 == 0, ldq >= K, ld_bytes >= K/2");
    for(int n = 0; n < N; ++n)
        for(int g = 0; g < K / 8; ++g)
        {
            uint32_t w = 0;
            for(int pos = 0; pos < 8; ++pos)
            {
                const int k = 8 * g + (pos < 4 ? 2 * pos : 2 * (pos - 4) + 1);
                const int v = q[(size_t)n * ldq + k];
                if(v < -8 || v > 7)
                    return fail(WQ_ERR_INVALID, "pack_int4_host: value %d at (k=%d, n=%d) outside -8..7", v, k, n);
                w |= (uint32_t)(v + 8) << (4 * pos);
            }
            memcpy(out + (size_t)n * ld_bytes + 4 * g, &w, 4);
        }
    return WQ_OK;
}

size_t wqgemm_splitk_multi_workspace_bytes(const wqgemm_problem* p, int32_t nranks)
{
    if(!p || nranks < 1 || p->M <= 0 || p->N <= 0 || p->M % nranks != 0)
        return 0;
    return ((size_t)p->M + (size_t)p->M / nranks) * p->N * sizeof(float);
}

int wqgemm_splitk_multi_step(const wqgemm_problem* p, void* nccl_comm, int32_t rank, int32_t nranks, void* workspace,
                             void* stream)
{
    if(!p || !workspace)
        return fail(WQ_ERR_INVALID, "splitk_multi_step: null problem or workspace");
    if(nranks < 1 || rank < 0 || rank >= nranks)
        return fail(WQ_ERR_INVALID, "splitk_multi_step: rank %d outside [0, %d)", rank, nranks);
    if(p->M <= 0 || p->N <= 0 || p->M % nranks != 0)
        return fail(WQ_ERR_UNSUPPORTED, "splitk_multi_step: M (%d) must be a positive multiple of the rank count %d", p->M, nranks);
    if(p->ldc != p->N || !p->c)
        return fail(WQ_ERR_UNSUPPORTED, "splitk_multi_step: C must be contiguous (ldc == N): the row blocks are gathered in place");
    if(p->k_batch > 1 || p->k_begin != 0 || p->k_end != 0)
        return fail(WQ_ERR_INVALID, "splitk_multi_step: the problem describes this rank's whole K slice (k_batch, k_begin, k_end unset)");
    if(p->c_dtype != WQ_F16 && p->c_dtype != WQ_BF16)
        return fail(WQ_ERR_UNSUPPORTED, "splitk_multi_step: C must be fp16 or bf16");
    hipStream_t s    = (hipStream_t)stream;
    const int rows   = p->M / nranks, row0 = rank * rows;
    float* partial   = (float*)workspace;
    float* sum_rows  = partial + (size_t)p->M * p->N;
    // 1. this rank's K slice -> fp32 partial (a rank without a slice passes K = 0 or a = NULL and contributes zeros)
    if(p->K <= 0 || !p->a)
        WQ_HIP(hipMemsetAsync(partial, 0, (size_t)p->M * p->N * sizeof(float), s));
    else
    {
        wqgemm_problem g = *p;
        g.c = partial, g.c_dtype = WQ_F32, g.ldc = p->N;
        g.k_begin = 0, g.k_end = p->K, g.k_batch = 1;
        if(const int rc = wqgemm_launch(&g, nullptr, stream); rc != WQ_OK)
            return rc;
    }
    // 2. sum over the ranks; rank r keeps rows [r rows, (r + 1) rows)
    const float* summed = partial + (size_t)row0 * p->N;
    if(nccl_comm)
    {
        const RcclApi& api = rccl();
        if(!api.ok)
            return fail(WQ_ERR_HIP, "splitk_multi_step: librccl.so.1 (ncclReduceScatter / ncclAllGather) could not be loaded");
        const int rc = api.reduce_scatter(partial, sum_rows, (size_t)rows * p->N, kNcclFloat32, kNcclSum, nccl_comm, s);
        if(rc != 0)
            return fail(WQ_ERR_HIP, "ncclReduceScatter failed: %s", api.error_string ? api.error_string(rc) : "?");
        summed = sum_rows;
    }
    // 3. epilogue + convert of this rank's rows, written in place into C; per-row / full-matrix D operands move with
    //    the row offset
    wqgemm_problem f = *p;
    f.M              = rows;
    f.K              = p->K > 0 ? p->K : 64;
    f.c              = (uint8_t*)p->c + (size_t)row0 * p->ldc * out_elt_bytes(p->c_dtype);
    const bool mm    = p->epilogue == WQ_EPI_MULTIPLY_MULTIPLY || p->epilogue == WQ_EPI_MULTIPLY_MULTIPLY_FASTGELU;
    if(f.d0)
        f.d0 = (const uint8_t*)p->d0 + (size_t)row0 * p->ldd0 * (mm ? 4 : 2);
    if(f.d1)
        f.d1 = (const uint8_t*)p->d1 + (mm ? (size_t)row0 * 4 : (size_t)row0 * p->ldd1 * 2);
    if(const int rc = wqgemm_reduce_finalize(&f, summed, 1, stream); rc != WQ_OK)
        return rc;
    // 4. every rank receives every row block
    if(nccl_comm)
    {
        const RcclApi& api = rccl();
        const size_t bytes = (size_t)rows * p->N * out_elt_bytes(p->c_dtype);
        const int rc       = api.all_gather((const uint8_t*)p->c + (size_t)rank * bytes, p->c, bytes, kNcclInt8, nccl_comm, s);
        if(rc != 0)
            return fail(WQ_ERR_HIP, "ncclAllGather failed: %s", api.error_string ? api.error_string(rc) : "?");
    }
    return WQ_OK;
}

// Diagnostic (not part of the public header): run p once with the stamped build of the default
// 256-tile fp16 x int4 kernel; dev_buf receives 8 waves x {work[L0,M0,L1,M1], wait[L0,M0,L1,M1]} cycles.
int wqgemm_debug_section_cycles(const wqgemm_problem* p, void* dev_buf, void* stream)
{
    g_dbg        = dev_buf;
    const int rc = wqgemm_launch(p, nullptr, stream);
    g_dbg        = nullptr;
    return rc;
}

int wqgemm_reduce_finalize(const wqgemm_problem* p, const float* partials, int32_t k_batch,
                           void* stream)
{
    if(!p || !partials || !p->c)
        return fail(WQ_ERR_INVALID, "null pointer");
    if(k_batch < 1 || p->M <= 0 || p->N <= 0 || p->N % 4 != 0 || p->ldc % 4 != 0 || p->ldc < p->N)
        return fail(WQ_ERR_INVALID, "reduce_finalize: need k_batch >= 1, N %% 4 == 0, ldc %% 4 == 0");
    if(p->c_dtype != WQ_F16 && p->c_dtype != WQ_BF16 && p->c_dtype != WQ_F32)
        return fail(WQ_ERR_UNSUPPORTED, "reduce_finalize: C must be fp16, bf16 or fp32");
    if(const int rc = check_fp_epilogue(p); rc != WQ_OK) // (the int8 requantisation kinds have no split-K form)
        return rc;
    return launch_reduce(p, partials, k_batch, (hipStream_t)stream);
}

} // extern "C"
