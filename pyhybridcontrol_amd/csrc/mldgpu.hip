// libmldgpu: C-ABI entry points (include/mldgpu.h).  Single translation unit; kernels live in the .inc files.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"
#include "condense.inc"
#include "problem.inc"

// ---- errors ------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void mld_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char *mld_last_error(void) { return g_err; }
const char *mld_version(void) { return "mldgpu 0.4 (gfx950, fp64 dense-dictionary cut-and-branch, in-kernel sub-tree hand-off; sizeof(mld_opts) = 64)"; }

int mld_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mld_set_device(int device)
{
    if (mld_device_count() <= 0) { mld_set_error("no HIP device"); return MLD_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device));
    return MLD_OK;
}

int mld_device_info(char *name, int name_len, int *n_cu, int64_t *hbm_bytes, int *lds_bytes)
{
    if (mld_device_count() <= 0) { mld_set_error("no HIP device"); return MLD_ERR_NO_DEVICE; }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
    return MLD_OK;
}

// ---- model -------------------------------------------------------------------------------------
static void mat_shape(const mld_dims &d, int id, int *r, int *c)
{
    const int rows[20] = {d.nx, d.nx, d.nx, d.nx, d.nx, d.nx, d.ny, d.ny, d.ny, d.ny, d.ny, d.ny,
                          d.nc, d.nc, d.nc, d.nc, d.nc, d.nc, d.nc, d.nc};
    const int cols[20] = {d.nx, d.nu, d.ndelta, d.nz, d.nomega, 1, d.nx, d.nu, d.ndelta, d.nz, d.nomega, 1,
                          d.nx, d.nu, d.ndelta, d.nz, d.nomega, 1, d.ny, d.nmu};
    *r = rows[id]; *c = cols[id];
}

static int model_create_impl(mld_model_t **out, const mld_dims *dims, int n_sets, int tv_N, const double *const *mats);

int mld_model_create(mld_model_t **out, const mld_dims *dims, int n_models, const double *const *mats)
{
    return model_create_impl(out, dims, n_models, 0, mats);
}

int mld_model_create_tv(mld_model_t **out, const mld_dims *dims, int n_horizons, int N_tilde, const double *const *mats)
{
    if (N_tilde < 1) { mld_set_error("mld_model_create_tv: N_tilde must be >= 1"); return MLD_ERR_INVALID; }
    return model_create_impl(out, dims, n_horizons, N_tilde, mats);
}

static int model_create_impl(mld_model_t **out, const mld_dims *dims, int n_sets, int tv_N, const double *const *mats)
{
    const int n_models = n_sets * (tv_N > 0 ? tv_N : 1);      // step models uploaded
    if (!out || !dims || n_sets < 1 || !mats) { mld_set_error("mld_model_create: bad arguments"); return MLD_ERR_INVALID; }
    const mld_dims &d = *dims;
    if (d.nx < 0 || d.nu < 0 || d.ndelta < 0 || d.nz < 0 || d.nmu < 0 || d.nomega < 0 || d.ny < 0 || d.nc < 0 ||
        d.nu_l < 0 || d.nu_l > d.nu || d.nmu_l < 0 || d.nmu_l > d.nmu) {
        mld_set_error("mld_model_create: invalid dimensions"); return MLD_ERR_INVALID;
    }
    if (mld_device_count() <= 0) { mld_set_error("no HIP device (libmldgpu has no CPU fallback)"); return MLD_ERR_NO_DEVICE; }
    mld_model *m = new mld_model();
    m->dims = d; m->n_models = n_sets; m->tv_N = tv_N; m->nv = d.nu + d.ndelta + d.nz + d.nmu;
    m->cond_N = -1; m->d_blocks = nullptr; m->d_pack = nullptr; m->pack_len = 0; m->d_tvQ = m->d_tvS = nullptr;
    for (int k = 0; k < 12; ++k) { m->d_out[k] = nullptr; m->d_out32[k] = nullptr; }
    m->h_mats.resize(20);
    for (int k = 0; k < 20; ++k) {
        int r, c; mat_shape(d, k, &r, &c);
        m->mat_rows[k] = r; m->mat_cols[k] = c; m->mat_size[k] = (size_t)r * c; m->d_mats[k] = nullptr;
        const size_t tot = m->mat_size[k] * n_models;
        m->h_mats[k].assign(tot, 0.0);
        if (tot == 0) continue;
        if (mats[k]) memcpy(m->h_mats[k].data(), mats[k], sizeof(double) * tot);
        bool nonzero = false;
        for (size_t t = 0; t < tot && !nonzero; ++t) nonzero = m->h_mats[k][t] != 0.0;
        if (!nonzero) continue;   // all-zero matrices are skipped like the reference's _all_zero_mats short-cuts
        hipError_t e = hipMalloc(&m->d_mats[k], sizeof(double) * tot);
        if (e == hipSuccess) e = hipMemcpy(m->d_mats[k], m->h_mats[k].data(), sizeof(double) * tot, hipMemcpyHostToDevice);
        if (e != hipSuccess) { mld_set_error("mld_model_create: %s", hipGetErrorString(e)); mld_model_destroy(m); return MLD_ERR_HIP; }
    }
    {   // every per-step matrix the block kernel needs, packed per model in its LDS order (one coalesced sweep instead of
        // thirteen dependent small copies); the input matrices are stacked horizontally as in mld_evolution_matrices.py:291,355,411
        const int nv = m->nv;
        const int plain[10] = {MT_A, MT_B4, MT_b5, MT_C, MT_D4, MT_d5, MT_E, MT_F4, MT_f5, MT_G};
        const int rws[3] = {d.nx, d.ny, d.nc};
        const int ids[3][4] = {{MT_B1, MT_B2, MT_B3, -1}, {MT_D1, MT_D2, MT_D3, -1}, {MT_F1, MT_F2, MT_F3, MT_Psi}};
        const int cw[4] = {d.nu, d.ndelta, d.nz, d.nmu};
        size_t len = 0;
        for (int k = 0; k < 10; ++k) len += m->mat_size[plain[k]];
        for (int f = 0; f < 3; ++f) len += (size_t)rws[f] * nv;
        m->pack_len = len;
        if (len) {
            std::vector<double> pack(len * n_models, 0.0);
            for (int mdl = 0; mdl < n_models; ++mdl) {
                double *dst = pack.data() + (size_t)mdl * len;
                for (int k = 0; k < 10; ++k) {
                    const size_t sz = m->mat_size[plain[k]];
                    if (sz) memcpy(dst, m->h_mats[plain[k]].data() + (size_t)mdl * sz, sizeof(double) * sz);
                    dst += sz;
                }
                for (int f = 0; f < 3; ++f) {
                    int o = 0;
                    for (int g = 0; g < 4; ++g) {
                        if (ids[f][g] >= 0 && cw[g])
                            for (int i = 0; i < rws[f]; ++i)
                                for (int c = 0; c < cw[g]; ++c)
                                    dst[(size_t)i * nv + o + c] = m->h_mats[ids[f][g]][(size_t)mdl * rws[f] * cw[g] + (size_t)i * cw[g] + c];
                        o += cw[g];
                    }
                    dst += (size_t)rws[f] * nv;
                }
            }
            hipError_t e = hipMalloc(&m->d_pack, sizeof(double) * pack.size());
            if (e == hipSuccess) e = hipMemcpy(m->d_pack, pack.data(), sizeof(double) * pack.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) { mld_set_error("mld_model_create: %s", hipGetErrorString(e)); mld_model_destroy(m); return MLD_ERR_HIP; }
        }
    }
    *out = m;
    return MLD_OK;
}

int mld_model_destroy(mld_model_t *m)
{
    if (!m) return MLD_OK;
    for (int k = 0; k < 20; ++k) if (m->d_mats[k]) (void)hipFree(m->d_mats[k]);
    if (m->d_blocks) (void)hipFree(m->d_blocks);
    if (m->d_pack) (void)hipFree(m->d_pack);
    if (m->d_tvQ) (void)hipFree(m->d_tvQ);
    if (m->d_tvS) (void)hipFree(m->d_tvS);
    for (int k = 0; k < 12; ++k) if (m->d_out[k]) (void)hipFree(m->d_out[k]);
    for (int k = 0; k < 12; ++k) if (m->d_out32[k]) (void)hipFree(m->d_out32[k]);
    delete m;
    return MLD_OK;
}

int mld_condense_device(mld_model_t *m, int N_tilde, int flags, double *kernel_ms)
{
    (void)flags;
    if (!m) { mld_set_error("null model"); return MLD_ERR_INVALID; }
    return condense_model_device(m, N_tilde, kernel_ms, 0);
}

int mld_condense(mld_model_t *m, int N_tilde, int flags, double *Phi_x, double *Gamma_v, double *Gamma_w,
                 double *Gamma_5, double *L_x, double *L_v, double *L_w, double *L_5, double *H_x, double *H_v,
                 double *H_w, double *H_5)
{
    int rc = mld_condense_device(m, N_tilde, flags, nullptr);
    if (rc) return rc;
    double *outs[12] = {Phi_x, Gamma_v, Gamma_w, Gamma_5, L_x, L_v, L_w, L_5, H_x, H_v, H_w, H_5};
    for (int k = 0; k < 12; ++k) {
        const size_t bytes = sizeof(double) * m->lay.out_size[k] * m->n_models;
        if (outs[k] && bytes) HIP_TRY(hipMemcpy(outs[k], m->d_out[k], bytes, hipMemcpyDeviceToHost));
    }
    return MLD_OK;
}

int mld_condense_device_f32(mld_model_t *m, int N_tilde, int flags, double *kernel_ms)
{
    (void)flags;
    if (!m) { mld_set_error("null model"); return MLD_ERR_INVALID; }
    return condense_model_device(m, N_tilde, kernel_ms, 0, true);
}

int mld_condense_f32(mld_model_t *m, int N_tilde, int flags, float *Phi_x, float *Gamma_v, float *Gamma_w, float *Gamma_5, float *L_x, float *L_v,
                     float *L_w, float *L_5, float *H_x, float *H_v, float *H_w, float *H_5)
{
    int rc = mld_condense_device_f32(m, N_tilde, flags, nullptr);
    if (rc) return rc;
    float *outs[12] = {Phi_x, Gamma_v, Gamma_w, Gamma_5, L_x, L_v, L_w, L_5, H_x, H_v, H_w, H_5};
    for (int k = 0; k < 12; ++k) {
        const size_t bytes = sizeof(float) * m->lay.out_size[k] * m->n_models;
        if (outs[k] && bytes) HIP_TRY(hipMemcpy(outs[k], m->d_out32[k], bytes, hipMemcpyDeviceToHost));
    }
    return MLD_OK;
}

} // extern "C"

#include "api_problem.inc"
