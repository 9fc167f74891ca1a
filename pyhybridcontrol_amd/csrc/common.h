// libmldgpu internal definitions (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/mldgpu.h"

#define MLD_WAVE 64

// ---- error plumbing -----------------------------------------------------------------------------
void mld_set_error(const char *fmt, ...);
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            mld_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MLD_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

// ---- per-model block store produced by k_condense_blocks ------------------------------------------
// family f in {0: state (rows nx), 1: output (rows ny), 2: constraint (rows nc)}
// blkV[f][k][r][c]  k = i-j (0 = diagonal block) ; blkW same with nw columns ; blkX[f][i][r][nx] ; blk5[f][i][r]
struct CondLayout {
    int nx, nu, nd, nz, nmu, nw, ny, nc, nv, N;
    int rows[3];          // nx, ny, nc
    size_t offV[3], offW[3], offX[3], off5[3];   // offsets (in doubles) inside the per-model block store
    size_t blk_stride;    // doubles per model
    // scratch inside the same store
    size_t offAk, offABv, offABw, offAb5, offS5, offBv, offDv, offFv;
    // materialised outputs: per model sizes in doubles
    size_t out_size[12];  // Phi_x, Gamma_v, Gamma_w, Gamma_5, L_x, L_v, L_w, L_5, H_x, H_v, H_w, H_5
    size_t out_off[12];   // offset of each matrix family buffer base (each buffer holds n_models copies)
};

struct mld_model {
    mld_dims dims;
    int n_models;         // horizons: with tv_N > 0 every horizon is tv_N consecutive step models in d_mats / h_mats / d_pack
    int tv_N;             // 0 = time-invariant (one model per horizon)
    int nv;
    // device copies of the 20 system matrices, each n_models x rows x cols (NULL if zero-sized)
    double *d_mats[20];
    size_t mat_size[20];
    int mat_rows[20], mat_cols[20];
    std::vector<std::vector<double>> h_mats;   // host copies (needed by the big-M tightening)
    double *d_pack;       // per model, packed once at creation in the LDS order of k_condense_blocks:
                          // A, B4, b5, C, D4, d5, E, F4, f5, G, [B1 B2 B3 0], [D1 D2 D3 0], [F1 F2 F3 Psi]
    size_t pack_len;
    // condensing results (device resident)
    int cond_N;
    CondLayout lay;
    double *d_blocks;     // n_models x blk_stride
    double *d_out[12];    // materialised matrices, each n_models x out_size[k]
    float *d_out32[12];   // the same in fp32 (mld_condense_f32), allocated on first use
    double *d_tvQ, *d_tvS; // time-varying horizons: products Q(i,j) (triangular) and the affine chain, written by k_tv_chain
};

// matrix order in d_mats
enum { MT_A = 0, MT_B1, MT_B2, MT_B3, MT_B4, MT_b5, MT_C, MT_D1, MT_D2, MT_D3, MT_D4, MT_d5, MT_E, MT_F1, MT_F2, MT_F3, MT_F4, MT_f5, MT_G, MT_Psi };
// output order in d_out
enum { O_PhiX = 0, O_GamV, O_GamW, O_Gam5, O_LX, O_LV, O_LW, O_L5, O_HX, O_HV, O_HW, O_H5 };

int condense_model_device(mld_model *m, int N, double *kernel_ms, hipStream_t stream, bool f32 = false);
void compute_layout(const mld_dims &d, int N, CondLayout *L);
