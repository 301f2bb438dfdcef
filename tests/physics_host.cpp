// tests/physics_host.cpp — the PRODUCT's per-cell physics (aerobulk_amd/csrc/ab_physics.hpp), compiled for the host, on a batch of
// cells: the cell-level sequence of flux_kernel (ab_kernels.hip compute_cell: pre-processing, TURB_<algo>, BULK_FORMULA, stress
// vector, warm-layer state between records).  TEST INFRASTRUCTURE: lets the CPU suite (`-m "not gpu"`) check every change to the
// physics header against the golden vectors of the unmodified reference before a GPU is spent on it.  The hardware seeds
// (v_rcp_f64, v_rsq_f64, v_log_f32 ...) are emulated at their accuracy (AB_FASTMATH_HOST), so results agree with the GPU to
// rounding, not to the bit.  Never part of the library: the product has no CPU path.
//
//   physics_host <in.bin> <out.bin> [f64|f32|mixed]
//   (f32 / mixed: the arithmetic of the AB_F32 / AB_F32_MIXED sessions on the same numbers; the caller rounds the inputs to fp32)
//   in : int32 algo, skin, niter, nt, hum_type ; int64 n ; double zt, zu ; 8 x n doubles (sst t_zt hum u v slp rad_sw rad_lw)
//   out: nt x 6 x n doubles (QL QH Tau_x Tau_y Evap T_s)
#define AB_FASTMATH_HOST 1
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../aerobulk_amd/csrc/ab_physics.hpp"
#include "../aerobulk_amd/csrc/ab_launch.hpp"

using namespace ab;

// R: arithmetic type, A: anchor type (ab_physics.hpp "ANCHORS"): <double,double> fp64, <float,float> AB_F32, <float,double> AB_F32_MIXED
template <int ALGO, bool SKIN, class R = double, class A = R>
static void cell(double zt, double zu, int nb_iter, int hum_type, const double *f[8], long k, double (&wls)[4], bool wl_load, double out[6])
{
    const Heights<R> h = make_heights<R>(zt, zu);
    const R sst = (R)f[0][k], t_zt = (R)f[1][k], hum = (R)f[2][k], uu = (R)f[3][k], vv = (R)f[4][k], slp = (R)f[5][k];
    A q_zt;
    if (hum_type == 0) q_zt = A(hum);
    else if (hum_type == 1) q_zt = q_air_dp<A>(A(hum), A(vmax(slp, R(50000.))));
    else q_zt = q_air_rh<A>(A(hum), A(t_zt), A(vmax(slp, R(50000.))));
    CellIn<R, A> in;
    in.sst = A(sst);
    in.theta_zt = theta_from_z_p0_t_q<A>(A(h.zt), A(slp), A(t_zt), q_zt);
    in.q_zt = q_zt;
    if (sizeof(R) != sizeof(A)) in.q_zt = A(R(q_zt));      // the mixed kernels park q as a float
    in.slp = slp;
    in.wnd = Mth<R>::sqrt(uu * uu + vv * vv);
    in.ssq = rounded(K<A>::rdct_qsat_salt * q_sat<A>(A(sst), A(slp)));
    in.qsw = SKIN ? (R(1.) - K<R>::roce_alb0) * (R)f[6][k] : R(0.);
    in.rlw = SKIN ? (R)f[7][k] : R(0.);
    R wl[4] = {(R)wls[0], (R)wls[1], (R)wls[2], (R)wls[3]};
    if (SKIN && !wl_load) { wl[0] = 0.; wl[1] = (ALGO == 4) ? 3. : 20.; wl[2] = 0.; wl[3] = 0.; }
    const bool dawn = dawn_at_lon0(12) != 0;
    CellOut<R, A> o;
    constexpr int kSkin = SKIN ? kSkinBoth : 0;
    if (ALGO == 1) turb_coare<R, false, kSkin, false, A>(h, in, nb_iter, wl, dawn, o);
    else if (ALGO == 2) turb_coare<R, true, kSkin, false, A>(h, in, nb_iter, wl, dawn, o);
    else if (ALGO == 3) turb_ncar<R, false, A>(h, in, nb_iter, o);
    else if (ALGO == 4) turb_ecmwf<R, kSkin, false, A>(h, in, nb_iter, wl, o);
    else turb_andreas<R, false, A>(h, in, nb_iter, o);
    R tau, qh, ql, ev;
    bulk_formula<R, A>(h.zu, o.T_s, o.q_s, o.t_zu, o.q_zu, o.Cd, o.Ch, o.Ce, in.wnd, o.Ubzu, slp, tau, qh, ql, ev);
    R tx = 0., ty = 0.;
    if (in.wnd > R(1.E-3)) { const R s = Mth<R>::div(tau, in.wnd); tx = s * uu; ty = s * vv; }
    for (int p = 0; p < 4; ++p) wls[p] = (double)wl[p];
    out[0] = ql; out[1] = qh; out[2] = tx; out[3] = ty; out[4] = ev; out[5] = (double)o.T_s;
}

typedef void (*cell_fn)(double, double, int, int, const double *[8], long, double (&)[4], bool, double[6]);
template <class R, class A> static cell_fn pick(int algo, int skin)
{
    switch (algo * 2 + (skin ? 1 : 0)) {
    case 2: return cell<1, false, R, A>;
    case 3: return cell<1, true, R, A>;
    case 4: return cell<2, false, R, A>;
    case 5: return cell<2, true, R, A>;
    case 6: return cell<3, false, R, A>;
    case 8: return cell<4, false, R, A>;
    case 9: return cell<4, true, R, A>;
    case 10: return cell<5, false, R, A>;
    default: return nullptr;
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) return 3;
    int32_t hdr[5];
    int64_t n;
    double z[2];
    if (fread(hdr, 4, 5, fi) != 5 || fread(&n, 8, 1, fi) != 1 || fread(z, 8, 2, fi) != 2) return 4;
    std::vector<double> buf((size_t)8 * n);
    if (fread(buf.data(), 8, (size_t)8 * n, fi) != (size_t)8 * n) return 5;
    fclose(fi);
    const int algo = hdr[0], skin = hdr[1], niter = hdr[2], nt = hdr[3], hum = hdr[4];
    const double *f[8];
    for (int i = 0; i < 8; ++i) f[i] = buf.data() + (size_t)i * n;
    const std::string prec = argc > 3 ? argv[3] : "f64";
    cell_fn fn = prec == "mixed" ? pick<float, double>(algo, skin) : (prec == "f32" ? pick<float, float>(algo, skin) : pick<double, double>(algo, skin));
    if (!fn) return 6;
    std::vector<double> out((size_t)nt * 6 * n), state((size_t)4 * n, 0.);
    for (int jt = 1; jt <= nt; ++jt)
        for (long k = 0; k < n; ++k) {
            double wl[4] = {state[k], state[n + k], state[2 * n + k], state[3 * n + k]}, o[6];
            fn(z[0], z[1], niter, hum, f, k, wl, jt > 1, o);
            for (int p = 0; p < 4; ++p) state[(size_t)p * n + k] = wl[p];
            for (int p = 0; p < 6; ++p) out[((size_t)(jt - 1) * 6 + p) * n + k] = o[p];
        }
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 7;
    fwrite(out.data(), 8, out.size(), fo);
    fclose(fo);
    return 0;
}
