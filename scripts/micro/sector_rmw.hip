// Micro-benchmark (round 3): what does the memory system deliver for k_solve's rank-1 update pattern?
// Every workgroup owns one dense fp64 "dictionary" (ROWS x LD) in HBM and repeats: pick NR rows (pseudo-random), and on each of them
// read-modify-write the ACTIVE sectors of the row (sector = SEC doubles; a fraction `act` of the sectors is active, drawn per iteration,
// the same for all rows of the iteration -- the pivot row's non-zero pattern).  Lane layout as in s_update_rows: 16 bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 -o sector_rmw sector_rmw.hip ;  ./sector_rmw <wgs> <threads> <sec_doubles> <act_permille> <iters> [mode] [rows of the private matrix, default 600]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ROWS_MAX = 600, LD = 576, NR = 107;
typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int SEC>
__global__ void __launch_bounds__(512) k_rmw(double *base, int iters, int act_permille, unsigned long long *bytes_out, int mode, int ROWS)
{
    extern __shared__ unsigned short lds[];
    unsigned short *seclist = lds, *rowlist = lds + 128;
    __shared__ int s_nact;
    const int tid = threadIdx.x, nt = blockDim.x, nw = nt / 64, wave = tid >> 6, lane = tid & 63;
    double *D = base + (size_t)blockIdx.x * ROWS * LD;      // the workgroup's private matrix: ROWS x LD doubles (the footprint knob)
    constexpr int LPS = SEC / 2, SPW = 64 / LPS, NSEC = LD / SEC;
    unsigned long long moved = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (tid == 0) {
            int na = 0;
            for (int s = 0; s < NSEC; ++s) if (hash32(blockIdx.x * 7919u + it * 104729u + s) % 1000u < (unsigned)act_permille) seclist[na++] = (unsigned short)s;
            s_nact = na;
            for (int r = 0; r < NR; ++r) rowlist[r] = (mode & 1) ? (unsigned short)((it * NR + r) % ROWS) : (unsigned short)(hash32(blockIdx.x * 31u + it * 65537u + r * 2654435761u) % ROWS);
        }
        __syncthreads();
        const int nact = s_nact;
        const int sub = lane / LPS, pr = (lane % LPS) * 2;
        const int ngrp = (nact + SPW - 1) / SPW;
        for (int g0 = 0; g0 < ngrp; g0 += 2) {
            int koff[2]; bool ok[2];
            for (int g = 0; g < 2; ++g) { const int si = (g0 + g) * SPW + sub; ok[g] = si < nact; koff[g] = ok[g] ? seclist[si] * SEC + pr : 0; }
            constexpr int RB = 16;
            for (int b = wave * RB; b < NR; b += nw * RB) {
                double2_t v[RB][2];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    if (b + q < NR) { const double *row = D + (size_t)rowlist[b + q] * LD;
#pragma unroll
                        for (int g = 0; g < 2; ++g) if (ok[g]) v[q][g] = (mode & 4) ? (double2_t){1.0, 2.0} : *(const double2_t *)(row + koff[g]); }
                }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    if (b + q < NR) { double *row = D + (size_t)rowlist[b + q] * LD;
#pragma unroll
                        for (int g = 0; g < 2; ++g) if (ok[g]) { double2_t o = v[q][g]; o = o * 0.999 + 1e-3; if (!(mode & 2)) *(double2_t *)(row + koff[g]) = o; else if (o.x == 123.456) row[koff[g]] = o.y; } }
                }
            }
        }
        moved += (unsigned long long)nact * SEC * 8 * ((mode & 6) ? 1 : 2) * NR;
    }
    if (tid == 0) atomicAdd(bytes_out, moved);
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 512, sec = argc > 3 ? atoi(argv[3]) : 8;
    const int act = argc > 4 ? atoi(argv[4]) : 380, iters = argc > 5 ? atoi(argv[5]) : 2000, mode = argc > 6 ? atoi(argv[6]) : 0, rows = argc > 7 ? atoi(argv[7]) : ROWS_MAX;
    double *d; unsigned long long *db, hb = 0;
    const size_t bytes = (size_t)wgs * rows * LD * sizeof(double);
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMemset(d, 0, bytes)); CHECK(hipMalloc(&db, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(db, 0, 8));
        CHECK(hipEventRecord(e0));
        const size_t lds = 2 * (128 + 128);
        if (sec == 8) hipLaunchKernelGGL(k_rmw<8>, dim3(wgs), dim3(threads), lds, 0, d, iters, act, db, mode, rows);
        else if (sec == 16) hipLaunchKernelGGL(k_rmw<16>, dim3(wgs), dim3(threads), lds, 0, d, iters, act, db, mode, rows);
        else if (sec == 32) hipLaunchKernelGGL(k_rmw<32>, dim3(wgs), dim3(threads), lds, 0, d, iters, act, db, mode, rows);
        else hipLaunchKernelGGL(k_rmw<4>, dim3(wgs), dim3(threads), lds, 0, d, iters, act, db, mode, rows);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&hb, db, 8, hipMemcpyDeviceToHost));
        if (rep) printf("footprint %4zu MB mode %d (1 = rows in sequence, 2 = read only, 4 = write only) wgs %4d threads %3d sector %3d B active %4.1f%%  iters %d: %8.2f ms  %7.1f GB/s (bytes moved)  %6.2f us/iteration/wg\n", bytes >> 20, mode, wgs, threads, sec * 8, act / 10.0, iters, ms, hb / (ms * 1e6), ms * 1e3 / iters);
    }
    return 0;
}
