// tools/h2d_probe.hip -- what the legacy host-pointer entry points (search_ac & co.: text in pageable host memory)
// can reach on this box: pageable hipMemcpy, pinned staging with 1..16 copy threads, hipHostRegister of the caller's
// buffer, and a kernel reading registered host memory directly.  Build: hipcc -O2 --offload-arch=gfx950 -o h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void xor_read(const uint4 *p, uint64_t n16, unsigned long long *out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 a = p[i];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x12345u) atomicXor(out, 1ull);
}

static void par_copy(char *dst, const char *src, size_t n, int threads)
{
    std::vector<std::thread> th;
    const size_t per = (n / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; ++t) {
        const size_t b = (size_t)t * per, e = b + per > n ? n : b + per;
        if (b >= n) break;
        th.emplace_back([=] { memcpy(dst + b, src + b, e - b); });
    }
    for (auto &t : th) t.join();
}

int main()
{
    const size_t N = (size_t)1 << 30, PIECE = (size_t)64 << 20;
    char *host = (char *)malloc(N);
    for (size_t i = 0; i < N; i += 4096) host[i] = (char)i; // fault the pages in
    memset(host, 1, N);
    char *dev = nullptr;
    CK(hipMalloc((void **)&dev, N));
    unsigned long long *d_out;
    CK(hipMalloc((void **)&d_out, 8));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now();
        CK(hipMemcpy(dev, host, N, hipMemcpyHostToDevice));
        printf("pageable hipMemcpy 1 GiB: %.1f GB/s\n", N / (now() - t0) / 1e9);
    }
    char *pin[2];
    CK(hipHostMalloc((void **)&pin[0], PIECE, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&pin[1], PIECE, hipHostMallocDefault));
    {
        memset(pin[0], 2, PIECE);
        const double t0 = now();
        for (size_t o = 0; o < N; o += PIECE) CK(hipMemcpyAsync(dev + o, pin[0], PIECE, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        printf("pinned -> device, 64 MiB pieces: %.1f GB/s\n", N / (now() - t0) / 1e9);
    }
    for (int threads : {1, 2, 4, 8, 12, 16, 24, 32}) {
        const double t0 = now();
        for (size_t o = 0; o < N; o += PIECE) par_copy(pin[0], host + o, PIECE, threads);
        printf("pageable -> pinned memcpy, %2d threads: %.1f GB/s\n", threads, N / (now() - t0) / 1e9);
    }
    for (int threads : {4, 8, 16}) {
        for (size_t piece : {(size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20}) {
            // the pipeline: piece k is staged by the threads while piece k-1 crosses PCIe
            hipEvent_t ev[2];
            CK(hipEventCreate(&ev[0]));
            CK(hipEventCreate(&ev[1]));
            const double t0 = now();
            int k = 0;
            for (size_t o = 0; o < N; o += piece, ++k) {
                const int b = k & 1;
                if (k >= 2) CK(hipEventSynchronize(ev[b]));
                par_copy(pin[b], host + o, piece, threads);
                CK(hipMemcpyAsync(dev + o, pin[b], piece, hipMemcpyHostToDevice, s0));
                CK(hipEventRecord(ev[b], s0));
            }
            CK(hipStreamSynchronize(s0));
            printf("pipeline %2d threads, %2zu MiB pieces: %.1f GB/s\n", threads, piece >> 20, N / (now() - t0) / 1e9);
        }
    }
    {
        double t0 = now();
        CK(hipHostRegister(host, N, hipHostRegisterDefault));
        const double treg = now() - t0;
        t0 = now();
        CK(hipMemcpyAsync(dev, host, N, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        const double tcopy = now() - t0;
        void *dp = nullptr;
        CK(hipHostGetDevicePointer(&dp, host, 0));
        CK(hipMemset(d_out, 0, 8));
        xor_read<<<1024, 256, 0, s0>>>((const uint4 *)dp, N / 16, d_out);
        CK(hipStreamSynchronize(s0));
        t0 = now();
        xor_read<<<1024, 256, 0, s0>>>((const uint4 *)dp, N / 16, d_out);
        CK(hipStreamSynchronize(s0));
        const double tk = now() - t0;
        t0 = now();
        CK(hipHostUnregister(host));
        const double tun = now() - t0;
        printf("hipHostRegister 1 GiB: %.3f s (%.1f GB/s), copy from it %.1f GB/s, kernel reading it in place %.1f GB/s, unregister %.3f s\n",
               treg, N / treg / 1e9, N / tcopy / 1e9, N / tk / 1e9, tun);
        printf("  register + copy: %.1f GB/s; register + in-place kernel + unregister: %.1f GB/s\n", N / (treg + tcopy + tun) / 1e9,
               N / (treg + tk + tun) / 1e9);
    }
    {
        // registering in pieces, pipelined against the copy of the piece before
        const size_t piece = (size_t)128 << 20;
        const double t0 = now();
        for (size_t o = 0; o < N; o += piece) {
            CK(hipHostRegister(host + o, piece, hipHostRegisterDefault));
            CK(hipMemcpyAsync(dev + o, host + o, piece, hipMemcpyHostToDevice, s0));
        }
        CK(hipStreamSynchronize(s0));
        const double t1 = now();
        for (size_t o = 0; o < N; o += piece) CK(hipHostUnregister(host + o));
        printf("register 128 MiB pieces + async copy: %.1f GB/s (with unregister %.1f GB/s)\n", N / (t1 - t0) / 1e9, N / (now() - t0) / 1e9);
    }
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    return 0;
}
