// Multi-GPU split-K from C++: one process per GPU, K partitioned across them, fp32 partials summed by RCCL inside
// wqgemm_splitk_multi_step (include/wqgemm.h).  The reference has no multi-GPU code; the flow keeps the shape of its
// in-GPU split-K (include/ck/tensor_operation/gpu/device/impl/device_gemm_xdl_cshuffle_v3r1.hpp:209-270: partial
// GEMMs into fp32 slabs, sum, convert) and its K partition (gridwise_gemm_xdl_cshuffle_v3.hpp:192-197,583-590).
//
//   example_splitk_multi_gpu [nranks = 1] [M = 1024] [N = 1024] [K = 8192] [timed_iters = 0]
//
// The parent forks the ranks BEFORE anything touches the GPU runtime (it never does itself), relays rank 0's RCCL unique
// id to the others over pipes, and returns 0 only if every rank verified its result.  Inputs are small integers from a
// stateless hash of (row, global k), so every rank can recompute any element of any rank's slice: the expected C is an
// exact integer sum, rounded to fp16 once -- the device result must match it bit for bit.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wqgemm.h"

namespace {

uint32_t mix(uint32_t a, uint32_t b, uint32_t seed)
{
    uint32_t h = seed ^ (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u);
    h ^= h >> 15, h *= 0x2C1B3C6Du, h ^= h >> 12, h *= 0x297A2D39u, h ^= h >> 15;
    return h;
}
int a_val(int m, int k) { return (int)(mix(m, k, 1u) % 5u) - 2; }      // activations in -2 .. 2
int q_val(int n, int k) { return (int)(mix(n, k, 2u) % 16u) - 8; }     // int4 weights in -8 .. 7
int s_val(int n) { return 1 + (int)(mix(n, 0, 3u) & 1u); }             // per-channel scale 1 or 2

uint16_t half_bits(float v)
{
    const _Float16 h = (_Float16)v;
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}

bool read_all(int fd, void* buf, size_t n)
{
    char* p = (char*)buf;
    while(n)
    {
        const ssize_t r = read(fd, p, n);
        if(r <= 0)
            return false;
        p += r, n -= (size_t)r;
    }
    return true;
}

#define HIP_OK(expr)                                                                                   \
    do                                                                                                 \
    {                                                                                                  \
        const hipError_t e_ = (expr);                                                                  \
        if(e_ != hipSuccess)                                                                           \
        {                                                                                              \
            fprintf(stderr, "rank %d: %s failed: %s\n", rank, #expr, hipGetErrorString(e_));         \
            return 1;                                                                                  \
        }                                                                                              \
    } while(0)
#define NCCL_OK(expr)                                                                                  \
    do                                                                                                 \
    {                                                                                                  \
        const ncclResult_t r_ = (expr);                                                                \
        if(r_ != ncclSuccess)                                                                          \
        {                                                                                              \
            fprintf(stderr, "rank %d: %s failed: %s\n", rank, #expr, ncclGetErrorString(r_));        \
            return 1;                                                                                  \
        }                                                                                              \
    } while(0)
#define WQ_OK_(expr)                                                                                   \
    do                                                                                                 \
    {                                                                                                  \
        const int r_ = (expr);                                                                         \
        if(r_ != WQ_OK)                                                                                \
        {                                                                                              \
            fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, #expr, r_, wqgemm_last_error());   \
            return 1;                                                                                  \
        }                                                                                              \
    } while(0)

int run_rank(int rank, int nranks, int M, int N, int K, int iters, int uid_out, int uid_in)
{
    int ngpu = 0;
    if(hipGetDeviceCount(&ngpu) != hipSuccess)
        ngpu = 0;
    if(ngpu < nranks)
    {
        if(rank == 0)
            fprintf(stderr, "example_splitk_multi_gpu: needs %d GPUs, found %d\n", nranks, ngpu);
        return 3;
    }
    HIP_OK(hipSetDevice(rank));
    ncclUniqueId uid;
    if(rank == 0)
    {
        NCCL_OK(ncclGetUniqueId(&uid));
        if(write(uid_out, &uid, sizeof(uid)) != (ssize_t)sizeof(uid))
            return 1;
    }
    if(!read_all(uid_in, &uid, sizeof(uid)))
    {
        fprintf(stderr, "rank %d: no unique id arrived\n", rank);
        return 1;
    }
    ncclComm_t comm;
    NCCL_OK(ncclCommInitRank(&comm, nranks, uid, rank));

    // this rank's K slice
    const int kread = wqgemm_splitk_kread(K, nranks, 64);
    const int k0 = rank * kread < K ? rank * kread : K, k1 = (rank + 1) * kread < K ? (rank + 1) * kread : K;
    const int klen = k1 - k0;
    std::vector<uint16_t> a((size_t)M * (klen ? klen : 1));
    std::vector<int8_t> q((size_t)N * (klen ? klen : 1));
    std::vector<uint16_t> scale(N);
    for(int m = 0; m < M; ++m)
        for(int k = 0; k < klen; ++k)
            a[(size_t)m * klen + k] = half_bits((float)a_val(m, k0 + k));
    for(int n = 0; n < N; ++n)
    {
        scale[n] = half_bits((float)s_val(n));
        for(int k = 0; k < klen; ++k)
            q[(size_t)n * klen + k] = (int8_t)q_val(n, k0 + k);
    }
    std::vector<uint8_t> packed((size_t)N * (klen ? klen / 2 : 1)), shuffled(packed.size());
    int b_dtype = WQ_I4_PACKED;
    if(klen > 0)
    {
        WQ_OK_(wqgemm_pack_int4_host(q.data(), klen, klen, N, packed.data(), klen / 2));
        if(klen % 64 == 0 && N % 32 == 0) // the offline order the tile kernels fetch whole kilobytes from
        {
            WQ_OK_(wqgemm_preshuffle_int4_host(packed.data(), klen / 2, klen, N, shuffled.data(), klen));
            packed.swap(shuffled);
            b_dtype = WQ_I4_PRESHUFFLED;
        }
    }
    void *d_a = nullptr, *d_b = nullptr, *d_s = nullptr, *d_c = nullptr, *d_ws = nullptr;
    HIP_OK(hipMalloc(&d_a, a.size() * 2));
    HIP_OK(hipMalloc(&d_b, packed.size()));
    HIP_OK(hipMalloc(&d_s, scale.size() * 2));
    HIP_OK(hipMalloc(&d_c, (size_t)M * N * 2));
    HIP_OK(hipMemcpy(d_a, a.data(), a.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_b, packed.data(), packed.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_s, scale.data(), scale.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_c, 0xFF, (size_t)M * N * 2));

    wqgemm_problem p;
    memset(&p, 0, sizeof(p));
    p.a_dtype = WQ_F16, p.b_dtype = b_dtype, p.scale_dtype = WQ_F16, p.c_dtype = WQ_F16;
    p.a_layout = WQ_ROW_MAJOR, p.b_layout = WQ_COL_MAJOR, p.c_layout = WQ_ROW_MAJOR;
    p.M = M, p.N = N, p.K = klen;
    p.lda = klen ? klen : 1, p.ldb = klen ? klen : 64, p.ldc = N;
    p.a = klen ? d_a : nullptr, p.b = d_b, p.scale = d_s, p.c = d_c;
    p.epilogue = WQ_EPI_PASSTHROUGH;
    const size_t ws_bytes = wqgemm_splitk_multi_workspace_bytes(&p, nranks);
    if(ws_bytes == 0)
    {
        fprintf(stderr, "rank %d: M = %d must be a positive multiple of the rank count %d\n", rank, M, nranks);
        return 1;
    }
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    WQ_OK_(wqgemm_splitk_multi_step(&p, comm, rank, nranks, d_ws, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<uint16_t> c((size_t)M * N);
    HIP_OK(hipMemcpy(c.data(), d_c, c.size() * 2, hipMemcpyDeviceToHost));

    // verification: a block in this rank's own rows, one in the next rank's rows (it arrived by all-gather) and a
    // scattered sample, each against the exact integer sum over the WHOLE K
    const int rows = M / nranks;
    long checked = 0, bad = 0;
    auto check = [&](int m, int n) {
        long sum = 0;
        for(int k = 0; k < K; ++k)
            sum += (long)a_val(m, k) * q_val(n, k);
        const uint16_t exp = half_bits((float)(sum * s_val(n)));
        ++checked;
        if(c[(size_t)m * N + n] != exp && ++bad <= 5)
            fprintf(stderr, "rank %d: C[%d][%d] = 0x%04x, expected 0x%04x (sum %ld)\n", rank, m, n, c[(size_t)m * N + n], exp,
                    sum * s_val(n));
    };
    for(int blk = 0; blk < 2; ++blk)
    {
        const int r0 = ((rank + blk) % nranks) * rows;
        for(int i = 0; i < 4 && i < rows; ++i)
            for(int n = 0; n < 32 && n < N; ++n)
                check(r0 + (i * 37) % rows, (n * 53 + 11 * rank) % N);
    }
    for(int i = 0; i < 64; ++i)
        check((int)(mix(i, rank, 9u) % (uint32_t)M), (int)(mix(i, rank, 10u) % (uint32_t)N));
    char kernel[128] = "";
    {
        wqgemm_problem g = p;
        if(klen > 0)
        {
            g.c_dtype = WQ_F32, g.k_begin = 0, g.k_end = klen;
            wqgemm_kernel_name(&g, kernel, sizeof(kernel));
        }
    }
    printf("rank %d/%d: K slice [%d, %d) on %s, %ld elements checked, %ld wrong\n", rank, nranks, k0, k1,
           klen ? kernel : "(empty slice)", checked, bad);

    if(iters > 0 && bad == 0)
    {
        for(int i = 0; i < 3; ++i)
            WQ_OK_(wqgemm_splitk_multi_step(&p, comm, rank, nranks, d_ws, stream));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipEventRecord(e0, stream));
        for(int i = 0; i < iters; ++i)
            WQ_OK_(wqgemm_splitk_multi_step(&p, comm, rank, nranks, d_ws, stream));
        HIP_OK(hipEventRecord(e1, stream));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        ms /= (float)iters;
        if(rank == 0)
            printf("Perf: %.4f ms per split-K GEMM over %d GPUs, %.1f TFlops (M = %d, N = %d, K = %d)\n", ms, nranks,
                   2.0 * M * N * K / 1e9 / ms, M, N, K);
    }
    HIP_OK(hipStreamSynchronize(stream));
    NCCL_OK(ncclCommDestroy(comm));
    fflush(stdout);
    return bad == 0 ? 0 : 1;
}

} // namespace

int main(int argc, char** argv)
{
    const int nranks = argc > 1 ? atoi(argv[1]) : 1;
    const int M = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 1024, K = argc > 4 ? atoi(argv[4]) : 8192;
    const int iters = argc > 5 ? atoi(argv[5]) : 0;
    if(nranks < 1 || nranks > 16 || M < 1 || N < 1 || K < 1)
    {
        fprintf(stderr, "usage: %s [nranks 1..16] [M] [N] [K] [timed_iters]\n", argv[0]);
        return 2;
    }
    // pipes: rank 0 -> parent (the unique id), parent -> every rank
    int up[2];
    std::vector<int> down_r(nranks), down_w(nranks);
    std::vector<pid_t> pids(nranks);
    if(pipe(up) != 0)
        return 2;
    for(int r = 0; r < nranks; ++r)
    {
        int d[2];
        if(pipe(d) != 0)
            return 2;
        down_r[r] = d[0], down_w[r] = d[1];
    }
    fflush(stdout);
    for(int r = 0; r < nranks; ++r)
    {
        pids[r] = fork(); // (no GPU call has been made in this process, and none is made in it afterwards)
        if(pids[r] < 0)
            return 2;
        if(pids[r] == 0)
        {
            close(up[0]);
            for(int o = 0; o < nranks; ++o)
            {
                close(down_w[o]);
                if(o != r)
                    close(down_r[o]);
            }
            const int rc = run_rank(r, nranks, M, N, K, iters, up[1], down_r[r]);
            fflush(stdout);
            fflush(stderr);
            _exit(rc);
        }
    }
    close(up[1]);
    for(int r = 0; r < nranks; ++r)
        close(down_r[r]);
    ncclUniqueId uid;
    const bool have_uid = read_all(up[0], &uid, sizeof(uid));
    for(int r = 0; r < nranks; ++r)
    {
        if(have_uid && write(down_w[r], &uid, sizeof(uid)) != (ssize_t)sizeof(uid))
            fprintf(stderr, "could not hand the unique id to rank %d\n", r);
        close(down_w[r]); // (without an id the ranks see end-of-file and leave)
    }
    int worst = 0;
    for(int r = 0; r < nranks; ++r)
    {
        int st = 0;
        waitpid(pids[r], &st, 0);
        const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if(rc > worst)
            worst = rc;
    }
    printf("example_splitk_multi_gpu: %s (%d rank%s, M = %d, N = %d, K = %d)\n", worst == 0 ? "verification: PASS" : "FAILED",
           nranks, nranks == 1 ? "" : "s", M, N, K);
    return worst;
}
