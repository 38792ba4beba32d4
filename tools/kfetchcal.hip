// kfetchcal.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the ACCESS SHAPES of the marching kernels
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// Every kernel moves a KNOWN number of bytes of a 9-plane lattice laid out like the library's (plane = nx columns x pitch rows,
// y fastest), each byte exactly once:
//   read16 / write16   one wave = 64 lanes x 16 B of one column (k_step's shape: 1 KB per wave instruction)
//   read8  / write8    one wave = 64 lanes x  8 B, a 128-row window marched over a range of columns, nine planes per column
//                      (k_march3<float,2,..>'s shape: 512 B per wave instruction), the next column requested one iteration ahead
//   read4              64 lanes x 4 B (256 B per wave instruction)
//   read_rec           the halo-table shape: lanes 0..5 read 4 B each of a 32-byte record per (seam, column), consecutive columns in
//                      consecutive iterations
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kfetchcal tools/kfetchcal.hip
// Run:   rocprofv3 --pmc FETCH_SIZE -d out -o f -- ./tools/kfetchcal      (and WRITE_SIZE, TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ...)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr int NX = 4096, NY = 4096, PITCH = 4096;
static constexpr long PLANE = (long)NX * PITCH;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// 64 lanes x 16 B: wave = (column, 256-row tile); 9 planes
__global__ __launch_bounds__(256) void read16(const float *__restrict__ f, float *__restrict__ sink)
{
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int tiles = NY / 256;
    const long col = wave / tiles, t = wave % tiles;
    if (col >= NX) return;
    const float *p = f + col * PITCH + t * 256 + lane * 4;
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 9; k++) acc += *reinterpret_cast<const f4 *>(p + k * PLANE);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void write16(float *__restrict__ f)
{
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int tiles = NY / 256;
    const long col = wave / tiles, t = wave % tiles;
    if (col >= NX) return;
    float *p = f + col * PITCH + t * 256 + lane * 4;
    const f4 v = {1.f, 2.f, 3.f, (float)lane};
#pragma unroll
    for (int k = 0; k < 9; k++) *reinterpret_cast<f4 *>(p + k * PLANE) = v;
}
// 64 lanes x 8 B: wave = (128-row window, range of `len` columns), marching; next column requested one iteration ahead
__global__ __launch_bounds__(256) void read8(const float *__restrict__ f, float *__restrict__ sink, int len)
{
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int per_win = NX / len;
    const long w = wave / per_win, u = wave % per_win;
    if (w >= NY / 128) return;
    const float *p = f + (u * len) * PITCH + w * 128 + lane * 2;
    f2 cur[9], acc = {0, 0};
#pragma unroll
    for (int k = 0; k < 9; k++) cur[k] = *reinterpret_cast<const f2 *>(p + k * PLANE);
    for (int x = 0; x < len; x++) {
        f2 nxt[9];
        const float *q = p + (long)((x + 1 < len) ? x + 1 : x) * PITCH;
#pragma unroll
        for (int k = 0; k < 9; k++) nxt[k] = *reinterpret_cast<const f2 *>(q + k * PLANE);
#pragma unroll
        for (int k = 0; k < 9; k++) acc = acc * 0.999f + cur[k];
#pragma unroll
        for (int k = 0; k < 9; k++) cur[k] = nxt[k];
    }
    if (acc.x + acc.y == 12345.678f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void write8(float *__restrict__ f, int len)
{
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int per_win = NX / len;
    const long w = wave / per_win, u = wave % per_win;
    if (w >= NY / 128) return;
    float *p = f + (u * len) * PITCH + w * 128 + lane * 2;
    for (int x = 0; x < len; x++) {
        const f2 v = {(float)x, (float)lane};
#pragma unroll
        for (int k = 0; k < 9; k++) *reinterpret_cast<f2 *>(p + (long)x * PITCH + k * PLANE) = v;
    }
}
// both at once, as the marching kernel: read lattice a, write lattice b
__global__ __launch_bounds__(256) void copy8(const float *__restrict__ f, float *__restrict__ o, int len)
{
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int per_win = NX / len;
    const long w = wave / per_win, u = wave % per_win;
    if (w >= NY / 128) return;
    const long base = (u * len) * PITCH + w * 128 + lane * 2;
    f2 cur[9];
#pragma unroll
    for (int k = 0; k < 9; k++) cur[k] = *reinterpret_cast<const f2 *>(f + base + k * PLANE);
    for (int x = 0; x < len; x++) {
        f2 nxt[9];
        const long q = base + (long)((x + 1 < len) ? x + 1 : x) * PITCH;
#pragma unroll
        for (int k = 0; k < 9; k++) nxt[k] = *reinterpret_cast<const f2 *>(f + q + k * PLANE);
#pragma unroll
        for (int k = 0; k < 9; k++) *reinterpret_cast<f2 *>(o + base + (long)x * PITCH + k * PLANE) = cur[k] * 1.0001f;
#pragma unroll
        for (int k = 0; k < 9; k++) cur[k] = nxt[k];
    }
}
// 64 lanes x 4 B: wave = (64-row window, range of columns)
__global__ __launch_bounds__(256) void read4(const float *__restrict__ f, float *__restrict__ sink, int len)
{
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int per_win = NX / len;
    const long w = wave / per_win, u = wave % per_win;
    if (w >= NY / 64) return;
    const float *p = f + (u * len) * PITCH + w * 64 + lane;
    float acc = 0;
    for (int x = 0; x < len; x++) {
#pragma unroll
        for (int k = 0; k < 9; k++) acc = acc * 0.999f + p[(long)x * PITCH + k * PLANE];
    }
    if (acc == 12345.678f) sink[0] = 1.f;
}
// the halo-table shape: table[(seam * (NX + 2) + x + 1) * 8 + lane], lanes 0..5, one record per iteration; three tables
__global__ __launch_bounds__(256) void read_rec(const float *__restrict__ t1, const float *__restrict__ t2, const float *__restrict__ t3,
                                                float *__restrict__ sink, int len)
{
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int per_win = NX / len;
    const long w = wave / per_win, u = wave % per_win;
    if (w >= NY / 128) return;
    float acc = 0;
    for (int x = 0; x < len; x++) {
        const long r = (w * (NX + 2) + u * len + x + 1) * 8 + (lane < 6 ? lane : 0);
        if (lane < 6) acc += t1[r] + t2[r] + t3[r];
    }
    if (acc == 12345.678f) sink[0] = 1.f;
}

// k_halo4's level-1 gather: one block = one seam x 66 columns, item (column, row) reads nine dwords out of the seam-buffer records of columns
// x-1, x, x+1 (320-byte records, [seam][column + 1][2 halves x 40]); every 128-byte line of the 31 x 4098 records is touched
__constant__ int c_ex[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}, c_ey[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
__global__ __launch_bounds__(256) void read_gather(const float *__restrict__ s3, float *__restrict__ sink)
{
    const int nblk_x = (NX + 59) / 60;
    const int b = 1 + blockIdx.x / nblk_x, x0 = (blockIdx.x % nblk_x) * 60;
    float acc = 0;
    for (int w = threadIdx.x; w < 66 * 6; w += 256) {
        const int cl = w / 6, r = w % 6, x = x0 - 3 + cl;
        if (x < 0 || x >= NX) continue;
        const float *rec = s3 + ((long)b * (NX + 2) + x + 1) * 80;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int q = 1 + r - c_ey[k];
            acc += rec[-(long)c_ex[k] * 80 + (q >> 2) * 40 + 4 * k + (q & 3)];
        }
    }
    if (acc == 12345.678f) sink[0] = 1.f;
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    float *a, *b, *sink, *t1, *t2, *t3;
    const size_t lat = (size_t)9 * PLANE * sizeof(float);
    const size_t tab = (size_t)(NY / 128) * (NX + 2) * 8 * sizeof(float);
    CK(hipMalloc(&a, lat)); CK(hipMalloc(&b, lat)); CK(hipMalloc(&sink, 256));
    CK(hipMalloc(&t1, tab)); CK(hipMalloc(&t2, tab)); CK(hipMalloc(&t3, tab));
    CK(hipMemset(a, 0, lat)); CK(hipMemset(b, 0, lat)); CK(hipMemset(t1, 0, tab)); CK(hipMemset(t2, 0, tab)); CK(hipMemset(t3, 0, tab));
    const int len = 64;                                                  // 64 columns per wave: 2048 waves = one resident round at 2 waves per SIMD
    const long waves8 = (long)(NY / 128) * (NX / len), waves4 = (long)(NY / 64) * (NX / len), waves16 = (long)NX * (NY / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double bytes, auto launch) {
        launch();                                                        // warm
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s %8.1f MB per launch  %8.1f us  %6.2f TB/s\n", name, bytes / 1e6, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e12);
    };
    // a 1.2 GB memset between kernels would keep the Infinity Cache (256 MiB) cold; the lattice itself (604 MB) is larger than it
    timed("read16", (double)lat, [&] { hipLaunchKernelGGL(read16, dim3(waves16 / 4), dim3(256), 0, 0, a, sink); });
    timed("write16", (double)lat, [&] { hipLaunchKernelGGL(write16, dim3(waves16 / 4), dim3(256), 0, 0, b); });
    timed("read8", (double)lat, [&] { hipLaunchKernelGGL(read8, dim3(waves8 / 4), dim3(256), 0, 0, a, sink, len); });
    timed("write8", (double)lat, [&] { hipLaunchKernelGGL(write8, dim3(waves8 / 4), dim3(256), 0, 0, b, len); });
    timed("copy8", 2.0 * lat, [&] { hipLaunchKernelGGL(copy8, dim3(waves8 / 4), dim3(256), 0, 0, a, b, len); });
    timed("read4", (double)lat, [&] { hipLaunchKernelGGL(read4, dim3(waves4 / 4), dim3(256), 0, 0, a, sink, len); });
    timed("read_rec", 3.0 * (NY / 128) * NX * 24, [&] { hipLaunchKernelGGL(read_rec, dim3(waves8 / 4), dim3(256), 0, 0, t1, t2, t3, sink, len); });
    {
        float *s3;
        const size_t sb = (size_t)33 * (NX + 2) * 80 * sizeof(float);
        CK(hipMalloc(&s3, sb)); CK(hipMemset(s3, 0, sb));
        // between launches the 604 MB lattice is read once (read16 above in the loop below) so that the 43 MB table is not served by the Infinity Cache
        const int nb = 31 * ((NX + 59) / 60);
        timed("read_gather", 31.0 * (NX + 2) * 320.0, [&] { hipLaunchKernelGGL(read16, dim3(waves16 / 4), dim3(256), 0, 0, a, sink);
                                                            hipLaunchKernelGGL(read_gather, dim3(nb), dim3(256), 0, 0, s3, sink); });
    }
    CK(hipDeviceSynchronize());
    return 0;
}
