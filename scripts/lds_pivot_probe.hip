// Feasibility probe: what would one dual-simplex pivot cost if the LP lived in LDS?
//
// Revised dual simplex on the condensed MLD problem with
//   * the compact block-Toeplitz constraint matrix  G  (N lag blocks of nc x nv doubles: cfg3 25 x 20 x 23 x 8 B = 92 KB) in LDS,
//   * the explicit inverse of the WORKING basis  W  (k x k, k = basic structurals = tight rows; measured on the bench workload with
//     the oracle: mean 53.5, max 75 at the root after cuts) in LDS,
// one workgroup (512 threads) per instance, all CUs busy.  Per pivot the kernel performs the data movement and arithmetic of
//   (1) btran:      y = e_r' W^-1                                   k
//   (2) pivot row:  alpha_c = sum_t y_t G[tight_t, c]  for all n columns  (walks the Toeplitz blocks of each tight row)   ~ k * n / 2
//   (3) ratio test: block min-reduction over n
//   (4) ftran:      w = W^-1 (G[tight, q])                          k * k   and the slack part  G[:, basic] w   m * k (Toeplitz)
//   (5) update:     W^-1 rank-1 update k * k,  x_B update m
// on synthetic data (the numbers are not a solver's; the op counts, LDS footprint and barriers are).  Printed next to it: the
// rank-1 update of the dense dictionary the production kernel does per pivot (HBM).  Build / run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_pivot_probe scripts/lds_pivot_probe.hip && /tmp/lds_pivot_probe [k=64] [pivots=2000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define NT 512
#define N_STEPS 25
#define NC 20
#define NV 23
#define NN (N_STEPS * NV)      // 575 columns
#define MM (N_STEPS * NC)      // 500 rows

__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }

// G[(i, c), (j, p)] = blk[i - j][c][p] for j <= i, else 0
__global__ void __launch_bounds__(NT) k_revised(int k, int pivots, const double *g_blk, double *sink)
{
    extern __shared__ double lds[];
    double *blk = lds;                                   // N_STEPS * NC * NV
    double *Winv = blk + N_STEPS * NC * NV;              // k * k
    double *y = Winv + k * k;                            // k
    double *alpha = y + k;                               // NN
    double *w = alpha + NN;                              // k
    double *xB = w + k;                                  // MM
    int *tight = (int *)(xB + MM);                       // k row ids
    int *basic = tight + k;                              // k column ids
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < N_STEPS * NC * NV; e += NT) blk[e] = g_blk[e];
    for (int e = tid; e < k * k; e += NT) Winv[e] = (e / k == e % k) ? 1.0 : 1e-3 * ((e * 7) % 13);
    for (int t = tid; t < k; t += NT) { tight[t] = (t * 37 + blockIdx.x) % MM; basic[t] = (t * 53 + blockIdx.x) % NN; }
    for (int i = tid; i < MM; i += NT) xB[i] = 1.0 + i;
    __syncthreads();
    double acc = 0.0;
    for (int it = 0; it < pivots; ++it) {
        const int r = it % k, q = (it * 31) % NN;
        // (1) btran: row r of W^-1
        for (int t = tid; t < k; t += NT) y[t] = Winv[r * k + t];
        __syncthreads();
        // (2) pivot row over all structural columns: thread per column, walks the tight rows (causal: row step >= column step)
        for (int c = tid; c < NN; c += NT) {
            const int j = c / NV, p = c - j * NV;
            double s = 0.0;
            for (int t = 0; t < k; ++t) {
                const int row = tight[t], i = row / NC, cc = row - i * NC;
                if (i >= j) s += y[t] * blk[((i - j) * NC + cc) * NV + p];
            }
            alpha[c] = s;
        }
        __syncthreads();
        // (3) ratio test: min over columns of |alpha| (stand-in for d_c / |alpha_c|)
        double mn = 1e300;
        for (int c = tid; c < NN; c += NT) { const double a = fabs(alpha[c]) + 1e-9; mn = fmin(mn, 1.0 / a); }
        for (int o = 32; o > 0; o >>= 1) mn = fmin(mn, __shfl_down(mn, o, 64));
        if (lane == 0) red[wave] = mn;
        __syncthreads();
        double theta = red[0];
        for (int v = 1; v < NT / 64; ++v) theta = fmin(theta, red[v]);
        // (4) ftran: w = W^-1 a_q restricted to the tight rows (a wave per output entry), then the slack rows' column
        {
            const int j = q / NV, p = q - j * NV;
            for (int t = wave; t < k; t += NT / 64) {
                double s = 0.0;
                for (int u = lane; u < k; u += 64) {
                    const int row = tight[u], i = row / NC, cc = row - i * NC;
                    const double a = i >= j ? blk[((i - j) * NC + cc) * NV + p] : 0.0;
                    s += Winv[t * k + u] * a;
                }
                s = wave_sum(s);
                if (lane == 0) w[t] = s;
            }
        }
        __syncthreads();
        // (5a) x_B update over all rows: x_B[i] -= theta * (G[i, q] - sum_t G[i, basic_t] w_t): thread per row, k-term Toeplitz gather
        for (int i = tid; i < MM; i += NT) {
            const int si = i / NC, cc = i - si * NC;
            double s = 0.0;
            for (int t = 0; t < k; ++t) {
                const int col = basic[t], j = col / NV, p = col - j * NV;
                if (si >= j) s += blk[((si - j) * NC + cc) * NV + p] * w[t];
            }
            xB[i] -= 1e-9 * theta * s;
        }
        // (5b) rank-1 update of W^-1
        const double piv = 1.0 / (fabs(w[r]) + 1.0);
        for (int e = tid; e < k * k; e += NT) { const int a = e / k, b = e - a * k; if (a != r) Winv[e] -= 1e-6 * w[a] * piv * y[b]; }
        __syncthreads();
        acc += theta;
    }
    if (tid == 0) sink[blockIdx.x] = acc + xB[3] + Winv[5];
}

// the production kernel's rank-1 update of a dense (m + cuts) x (n + 1) dictionary in HBM: rows x sectors as measured (278 of 560 rows, 34 of
// 73 sectors per pivot would be the sparse version; this is the plain dense bound: every row, every column)
__global__ void __launch_bounds__(NT) k_dense(int rows, int ld, int pivots, double *D, int touch_rows, int touch_cols)
{
    double *d = D + (size_t)blockIdx.x * rows * ld;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int it = 0; it < pivots; ++it) {
        const int r0 = (it * 17) % (rows - touch_rows + 1);
        for (int i = wave; i < touch_rows; i += NT / 64) {
            double *row = d + (size_t)(r0 + i) * ld;
            for (int c = lane * 2; c + 1 < touch_cols; c += 128) { double2 v = *(double2 *)(row + c); v.x -= 1e-9 * v.y; v.y -= 1e-9; *(double2 *)(row + c) = v; }
        }
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    const int k = argc > 1 ? atoi(argv[1]) : 64, pivots = argc > 2 ? atoi(argv[2]) : 2000;
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int grid = prop.multiProcessorCount;
    std::vector<double> blk((size_t)N_STEPS * NC * NV);
    for (size_t e = 0; e < blk.size(); ++e) blk[e] = ((e * 2654435761u) % 1000) / 1000.0 - 0.5;
    double *d_blk, *d_sink, *d_D;
    hipMalloc(&d_blk, blk.size() * 8); hipMemcpy(d_blk, blk.data(), blk.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&d_sink, grid * 8);
    const size_t lds = 8 * ((size_t)N_STEPS * NC * NV + (size_t)k * k + k + NN + k + MM) + 4 * 2 * (size_t)k + 64;
    hipFuncSetAttribute((const void *)k_revised, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_revised, dim3(grid), dim3(NT), lds, 0, k, 50, d_blk, d_sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_revised, dim3(grid), dim3(NT), lds, 0, k, pivots, d_blk, d_sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("revised, LDS resident: k = %d, LDS %.1f KB per workgroup, %d workgroups: %.2f us per pivot (%s)\n", k, lds / 1024.0, grid, ms * 1e3 / pivots, hipGetErrorString(hipGetLastError()));
    const int rows = 701, ld = 576;
    hipMalloc(&d_D, (size_t)grid * rows * ld * 8); hipMemset(d_D, 0, (size_t)grid * rows * ld * 8);
    const int cfgs[3][2] = {{560, 576}, {278, 34 * 8}, {278, 576}};
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(k_dense, dim3(grid), dim3(NT), 0, 0, rows, ld, 20, d_D, cfgs[c][0], cfgs[c][1]);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_dense, dim3(grid), dim3(NT), 0, 0, rows, ld, 400, d_D, cfgs[c][0], cfgs[c][1]);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 2.0 * cfgs[c][0] * cfgs[c][1] * 8.0;
        printf("dense dictionary in HBM: %d rows x %d columns per pivot (%.0f KB moved): %.2f us per pivot, %.2f TB/s over %d workgroups\n", cfgs[c][0], cfgs[c][1], bytes / 1024,
               ms * 1e3 / 400, bytes * 400 * grid / (ms * 1e-3) / 1e12, grid);
    }
    return 0;
}
